// csim_sanitize_check -- CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer (GPU ASan is not
// available on this pool): the front-end (parser, circuit, flatten), the assembly plan, all three kernel generators
// with the shipped schedules, the sixteen-lane plan's host interpreter, and the CPU oracle (DC + a few transient
// steps, the threaded batch driver).  Test infrastructure (it links the oracle): built by tests/sanitize/Makefile, run by
// tests/test_sanitizers.py.
//
//   csim_sanitize_check <netlist.sp> <schedule-file|->
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "api/circuit.hpp"
#include "api/parser.hpp"
#include "engine/codegen.hpp"
#include "engine/group_plan.hpp"
#include "engine/plan.hpp"

extern "C" {
#include "mna_oracle.h"
}

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: csim_sanitize_check <netlist.sp> <schedule|->\n"); return 1; }
    Circuit ckt;
    SimulationConfig sim;
    if (!parseNetlist(argv[1], ckt, sim)) return 2;
    ckt.assignEquationIndices();
    const csim::CircuitIR cir = csim::flatten(ckt);
    const csim_ir* ir = cir.view();
    const int N = ir->n_unknowns, P = ir->n_params;
    const csim::AssemblyPlan ap = csim::buildAssemblyPlan(*ir);

    std::string text = "-";
    if (std::string(argv[2]) != "-") {
        std::ifstream f(argv[2]);
        if (!f) return 2;
        std::string line;
        text.clear();
        while (std::getline(f, line)) text += line + "\n";
    }
    csim::ScheduleSet sch;
    if (!csim::ScheduleSet::parse(text, N, sch)) return 2;
    csim::CodegenStats st;
    const std::string src = csim::generateTranKernelSource(*ir, ap, sch, "sanitize", &st);
    csim::GeneratorOptions wide;
    wide.set("near_band=0.1");
    wide.set("stage_ahead=-1");
    wide.set("pipeline_mos=0");
    const std::string src2 = csim::generateTranKernelSource(*ir, ap, sch, "sanitize", &st, wide);
    std::printf("generated %zu + %zu bytes of HIP for N=%d\n", src.size(), src2.size(), N);
    if (src.empty()) return 3;

    // sixteen-lane plan through its host interpreter on one set of term values
    if (ir->has_nonlinear) {
        csim::GroupPlan gp;
        if (csim::buildGroupPlan(*ir, ap, sch.alts[0], gp)) {
            std::vector<double> T(static_cast<std::size_t>(ap.nTerms), 0.75), x(static_cast<std::size_t>(N));
            T[static_cast<std::size_t>(ap.termOne)] = 1.0;
            bool viol = false, planError = false;
            csim::interpretGroupPlan(gp, ap, *ir, T.data(), 1e-15, x.data(), &viol, &planError);
            std::printf("group plan interpreted: violated=%d planError=%d\n", (int)viol, (int)planError);
        }
    }

    // the oracle: DC, 20 transient steps, and four instances on three threads
    std::vector<double> x(static_cast<std::size_t>(N));
    int32_t dcIt = 0;
    uint32_t status = 0;
    if (oracle_dc(ir, cir.nominal.data(), 1, x.data(), &dcIt, &status) != 0) return 4;
    const double dt = sim.tran.enabled ? sim.tran.tstep : 1e-9;
    int64_t its = 0, nRows = 0;
    std::vector<int32_t> per(20);
    std::vector<double> rows(21 * static_cast<std::size_t>(N + 1));
    if (oracle_tran(ir, cir.nominal.data(), 1, dt, dt * 20, 0.0, nullptr, rows.data(), 21, &nRows, x.data(), &its, per.data(), &status) < 0) return 4;
    std::vector<double> table(static_cast<std::size_t>(P) * 4);
    for (int p = 0; p < P; ++p)
        for (int b = 0; b < 4; ++b) table[static_cast<std::size_t>(p) * 4 + static_cast<std::size_t>(b)] = cir.nominal[static_cast<std::size_t>(p)] * (1.0 + 0.01 * b * (cir.mcKind[static_cast<std::size_t>(p)] == 1));
    int64_t its4[4] = {0, 0, 0, 0};
    if (oracle_tran_batch_mt(ir, table.data(), 4, 0, 4, dt, dt * 10, 3, its4) < 1) return 4;
    std::printf("oracle: dc %d passes, 20 steps %lld passes, rows %lld; batch %lld %lld %lld %lld\n", dcIt, (long long)its, (long long)nRows,
                (long long)its4[0], (long long)its4[1], (long long)its4[2], (long long)its4[3]);
    return 0;
}
