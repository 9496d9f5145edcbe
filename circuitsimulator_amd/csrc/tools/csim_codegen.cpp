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
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "../api/circuit.hpp"
#include "../api/parser.hpp"
#include "../engine/codegen.hpp"
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

int main(int argc, char** argv)
{
    csim::GeneratorOptions gopt;
    while (argc >= 3 && std::string(argv[1]) == "--opt") {
        if (!gopt.set(argv[2])) { std::cerr << "unknown generator option " << argv[2] << "\n"; return 1; }
        argv += 2;
        argc -= 2;
    }
    const bool hashOnly = argc >= 2 && std::string(argv[1]) == "--hash";
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
