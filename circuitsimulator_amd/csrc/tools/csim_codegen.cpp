// csim_codegen -- emit the circuit-specialised transient kernel of a netlist.
//
//   csim_codegen <netlist.sp> <schedule-file|-> <out.hip>      writes HIP source, prints the hash
//   csim_codegen --hash <netlist.sp> <schedule-file|->          prints the hash only
//   csim_codegen --opt key=value ... (before the other arguments): generator options
//     (barrier_every=3, sweep=0,16,32 -- tuning aids, part of the library's hash)
//
// The schedule file holds the partial-pivot row swaps of the transient
// factorisation as "column:row,column:row,..." ('-' = no swaps).  It is
// recorded from the general kernel's planner (tools/record_schedule.py) and is
// verified again at run time on every factorisation, so a stale schedule can
// cost speed (fallback to the general kernel) but never correctness.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "../api/circuit.hpp"
#include "../api/parser.hpp"
#include "../engine/codegen.hpp"
#include "../engine/group_plan.hpp"
#include "../engine/plan.hpp"

static std::string readSchedule(const std::string& path)
{
    if (path == "-") return "-";
    std::ifstream f(path);
    if (!f) { std::cerr << "cannot open schedule file " << path << "\n"; std::exit(2); }
    std::string line, all;
    while (std::getline(f, line)) all += line + "\n";      // one schedule per line; '#' comments
    return all;
}

// --selftest-group: run the sixteen-lanes-per-instance plan through its host interpreter on random term
// values and compare with a plain dense elimination that uses the same pivot order (no GPU needed)
static int selftestGroup(const csim_ir* ir, const csim::AssemblyPlan& ap, const csim::ScheduleSet& sch, int lanes,
                         const csim::GeneratorOptions& gopt)
{
    int worstAlt = -1;
    double worst = 0.0;
    csim::GroupPlan first;
    for (std::size_t alt = 0; alt < sch.alts.size(); ++alt) {
        csim::GroupPlan gp;
        // as the emitter does: alternatives are planned over the first schedule's row placement
        // (and the first over the placement found by the search, where the generator options ask for it)
        csim::GroupPlan placed;
        const bool search = alt == 0 && (gopt.placeSearch & (lanes == csim::kGroupLanes ? 2 : 1)) != 0;
        if (search) {
            double c0 = 0.0, c1 = 0.0;
            if (!csim::optimizeGroupPlacement(*ir, ap, sch.alts, lanes, placed, &c0, &c1)) { std::printf("group plan: circuit does not fit\n"); return 3; }
            std::printf("placement search: planned instructions per solve (weighted over the schedules) %.1f -> %.1f\n", c0, c1);
        }
        if (!csim::buildGroupPlan(*ir, ap, sch.alts[alt], gp, alt ? &first : (search ? &placed : nullptr), lanes)) { std::printf("group plan: circuit does not fit\n"); return 3; }
        if (alt == 0) first = gp;
        const int N = ir->n_unknowns, LD = ap.LD;
        unsigned long long seed = 0x9E3779B97F4A7C15ull + alt;
        auto rnd = [&seed]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (double)((seed >> 11) & 0xFFFFFFFFull) / 4294967296.0; };
        for (int trial = 0; trial < 20; ++trial) {
            std::vector<double> T((std::size_t)ap.nTerms);
            for (double& t : T) t = 0.5 + rnd();
            T[(std::size_t)ap.termOne] = 1.0;
            T[(std::size_t)ap.termGmin] = 1e-3;
            for (int e = 0; e < ir->n_elems; ++e) {
                const int tb = ap.termBase[(std::size_t)e];
                if (ir->kind[e] == CSIM_L) T[(std::size_t)(tb + csim::T_L_ONE)] = 1.0;
                if (ir->kind[e] == CSIM_NMOS || ir->kind[e] == CSIM_PMOS) {
                    T[(std::size_t)(tb + csim::T_M_GD)] = 0.1 * rnd();
                    T[(std::size_t)(tb + csim::T_M_GG)] = 0.1 * rnd();
                    T[(std::size_t)(tb + csim::T_M_GS)] = -(T[(std::size_t)(tb + csim::T_M_GD)] + T[(std::size_t)(tb + csim::T_M_GG)]);
                    T[(std::size_t)(tb + csim::T_M_CST)] = rnd() - 0.5;
                }
            }
            // dense reference with the same pivot order
            std::vector<double> A((std::size_t)N * (N + 1), 0.0);
            const csim::GatherPlan& g = ap.tran;
            for (int n = 0; n < g.nnzG(); ++n) {
                double acc = 0.0;
                for (int c = g.gPtr[(std::size_t)n]; c < g.gPtr[(std::size_t)n + 1]; ++c)
                    acc = (g.gCon[(std::size_t)c] & 1) ? acc - T[(std::size_t)(g.gCon[(std::size_t)c] >> 1)] : acc + T[(std::size_t)(g.gCon[(std::size_t)c] >> 1)];
                A[(std::size_t)(g.gPos[(std::size_t)n] / LD) * (N + 1) + (std::size_t)(g.gPos[(std::size_t)n] % LD)] = acc;
            }
            for (int n = 0; n < g.nnzI(); ++n) {
                double acc = 0.0;
                for (int c = g.iPtr[(std::size_t)n]; c < g.iPtr[(std::size_t)n + 1]; ++c)
                    acc = (g.iCon[(std::size_t)c] & 1) ? acc - T[(std::size_t)(g.iCon[(std::size_t)c] >> 1)] : acc + T[(std::size_t)(g.iCon[(std::size_t)c] >> 1)];
                A[(std::size_t)g.iRow[(std::size_t)n] * (N + 1) + (std::size_t)N] = acc;
            }
            for (int k = 0; k < N; ++k) {
                const int p = sch.alts[alt].pivotPos[(std::size_t)k];
                if (p != k) for (int j = 0; j <= N; ++j) std::swap(A[(std::size_t)k * (N + 1) + j], A[(std::size_t)p * (N + 1) + j]);
                for (int i = k + 1; i < N; ++i) {
                    const double f = A[(std::size_t)i * (N + 1) + k] / A[(std::size_t)k * (N + 1) + k];
                    if (f == 0.0) continue;
                    for (int j = k + 1; j <= N; ++j) A[(std::size_t)i * (N + 1) + j] -= f * A[(std::size_t)k * (N + 1) + j];
                }
            }
            std::vector<double> xr((std::size_t)N), xg((std::size_t)N);
            for (int i = N - 1; i >= 0; --i) {
                double sum = A[(std::size_t)i * (N + 1) + N];
                for (int j = i + 1; j < N; ++j) sum -= A[(std::size_t)i * (N + 1) + j] * xr[(std::size_t)j];
                xr[(std::size_t)i] = sum / A[(std::size_t)i * (N + 1) + i];
            }
            bool viol = false, planError = false;
            csim::interpretGroupPlan(gp, ap, *ir, T.data(), 1e-15, xg.data(), &viol, &planError);
            if (planError) { std::printf("group plan alt %zu: a candidate test would see a row it must not\n", alt); return 5; }
            for (int i = 0; i < N; ++i) {
                const double e = std::fabs(xg[(std::size_t)i] - xr[(std::size_t)i]) / std::max(std::fabs(xr[(std::size_t)i]), 1e-6);
                if (e > worst) { worst = e; worstAlt = (int)alt; }
            }
        }
        std::printf("group plan alt %zu: elimination instructions by dependence: exact constants %d, launch-constant %d, per step %d, per iteration %d\n",
                    alt, gp.opsByLevel[0], gp.opsByLevel[1], gp.opsByLevel[2], gp.opsByLevel[3]);
        std::printf("group plan alt %zu: N=%d slots=%d classes=%zu staging rows=%zu  per solve (wave instructions): bcast=%d fma=%d mul=%d cmp=%d recip=%d; critical path %d cycles after elimination, %d after substitution\n",
                    alt, gp.N, gp.S, gp.gClasses.size(), gp.stageRows.size(), gp.nBcast, gp.nFma, gp.nMul, gp.nCmp, gp.nRecip,
                    gp.depthElimination, gp.depthSolve);
    }
    std::printf("group plan self test: worst relative difference to the dense elimination %.3g (alternative %d)\n", worst, worstAlt);
    return worst < 1e-9 ? 0 : 4;
}

int main(int argc, char** argv)
{
    csim::GeneratorOptions gopt;
    while (argc >= 3 && std::string(argv[1]) == "--opt") {
        if (!gopt.set(argv[2])) { std::cerr << "unknown generator option " << argv[2] << "\n"; return 1; }
        argv += 2;
        argc -= 2;
    }
    const bool selftest4 = argc >= 2 && std::string(argv[1]) == "--selftest-group4";      // the plan for four lanes per instance
    const bool selftest = selftest4 || (argc >= 2 && std::string(argv[1]) == "--selftest-group");
    const bool hashOnly = argc >= 2 && (std::string(argv[1]) == "--hash" || selftest);
    if ((hashOnly && argc != 4) || (!hashOnly && argc != 4)) {
        std::cerr << "usage: csim_codegen <netlist.sp> <schedule|-> <out.hip>\n"
                     "       csim_codegen --hash <netlist.sp> <schedule|->\n";
        return 1;
    }
    const std::string netlist = hashOnly ? argv[2] : argv[1];
    const std::string schedPath = hashOnly ? argv[3] : argv[2];

    Circuit ckt;
    SimulationConfig sim;
    if (!parseNetlist(netlist, ckt, sim)) return 2;
    ckt.assignEquationIndices();
    const csim::CircuitIR cir = csim::flatten(ckt);
    const csim_ir* ir = cir.view();
    const csim::AssemblyPlan ap = csim::buildAssemblyPlan(*ir);
    csim::ScheduleSet sch;
    if (!csim::ScheduleSet::parse(readSchedule(schedPath), ir->n_unknowns, sch)) {
        std::cerr << "bad schedule\n";
        return 2;
    }
    const unsigned long long h = csim::scheduleHash(*ir, sch, gopt);
    // the library is NAMED by the topology hash (what an engine can compute before it
    // knows any schedule); the full hash is embedded for diagnostics
    const unsigned long long topo = csim::scheduleHash(*ir, csim::PivotSchedule::identity(ir->n_unknowns));
    if (selftest) return selftestGroup(ir, ap, sch, selftest4 ? 4 : 16, gopt);
    if (hashOnly) { std::printf("%016llx\n", topo); return 0; }

    std::string label = netlist;
    const std::size_t slash = label.find_last_of('/');
    if (slash != std::string::npos) label = label.substr(slash + 1);
    csim::CodegenStats st;
    const std::string src = csim::generateTranKernelSource(*ir, ap, sch, label, &st, gopt);
    std::ofstream out(argv[3]);
    if (!out) { std::cerr << "cannot write " << argv[3] << "\n"; return 2; }
    out << src;
    std::fprintf(stderr, "csim_codegen: %s N=%d hash=%016llx  first schedule, per NR iteration: fma=%d mul=%d add/sub=%d recip=%d "
                         "cmp=%d  L entries=%d dynamic U entries=%d\n",
                 label.c_str(), ir->n_unknowns, h, st.nFma, st.nMul, st.nAddSub, st.nRecip, st.nCmp, st.nLower, st.nDynU);
    std::printf("%016llx\n", topo);
    return 0;
}
