// csim_cli -- a caller of the reference-shaped C++ API (api/*.hpp), end to end on the GPU:
//
//   csim_cli <netlist.sp> [waveforms.csv]
//
// parseNetlist -> Circuit::assignEquationIndices -> computeDcOperatingPoint -> (with a .TRAN card)
// runTransientAnalysisBackwardEuler, i.e. the sequence of calls the reference's driver makes.  The report
// it prints is this tool's own: a size line, one table of node voltages, one of branch currents (the
// lines tests/test_gpu_parity.py::test_cli_writes_the_reference_csv looks for), and where the CSV went.
// The upstream driver itself (src/main.cpp) compiles and links unchanged against the same headers
// (INTEGRATION.md §1); this tool is the smaller program the tests need.
#include <cstdio>
#include <exception>
#include <iostream>
#include <string>
#include <vector>

#include "../api/circuit.hpp"
#include "../api/dcanalysis.hpp"
#include "../api/element.hpp"
#include "../api/parser.hpp"
#include "../api/sim.hpp"
#include "../api/tanalisis.hpp"

namespace {

struct BranchRow {
    std::string label;      // "L2, 117 -> 118" / "VDD, +103 -> -0"
    int eq;
};

// the unknowns beyond the node voltages: one per voltage source and inductor, in element order
std::vector<BranchRow> branchRows(const Circuit& c)
{
    std::vector<BranchRow> rows;
    for (const auto& el : c.elements) {
        const auto* v = dynamic_cast<const VoltageSource*>(el.get());
        const auto* l = dynamic_cast<const Inductor*>(el.get());
        if (!v && !l) continue;
        const std::string& a = c.nodes[static_cast<std::size_t>(el->getNodeIds()[0])].name;
        const std::string& b = c.nodes[static_cast<std::size_t>(el->getNodeIds()[1])].name;
        rows.push_back({el->getName() + (v ? ", +" + a + " -> -" + b : ", " + a + " -> " + b),
                        v ? v->getBranchEqIndex() : l->getBranchEqIndex()});
    }
    return rows;
}

int report(const Circuit& c, const Eigen::VectorXd& x)
{
    std::printf("Unknowns     : %d  (nodeEq=%d, branchEq=%d)   nodes %zu, elements %zu\n", c.numUnknowns(),
                c.numNodeEquations(), c.numVoltageBranches(), c.nodes.size(), c.elements.size());
    if (x.size() != c.numUnknowns()) { std::fprintf(stderr, "csim_cli: operating point has %ld entries\n", (long)x.size()); return 1; }
    std::printf("\noperating point, node voltages\n");
    for (const Node& n : c.nodes) {
        if (n.eqIndex < 0) std::printf("V(%s) = 0.000000 V   [GND]\n", n.name.c_str());
        else std::printf("V(%s) = %.6f V   [eqIndex=%d]\n", n.name.c_str(), x(n.eqIndex), n.eqIndex);
    }
    std::printf("\noperating point, branch currents\n");
    for (const BranchRow& r : branchRows(c))
        std::printf("I(%s) = %.6f A   [branchEq=%d]\n", r.label.c_str(), (r.eq >= 0 && r.eq < x.size()) ? x(r.eq) : 0.0, r.eq);
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 2 || argc > 3) { std::fprintf(stderr, "usage: csim_cli <netlist.sp> [waveforms.csv]\n"); return 1; }
    const std::string csv = argc == 3 ? argv[2] : "tran_out.csv";
    Circuit c;
    SimulationConfig cfg;
    if (!parseNetlist(argv[1], c, cfg)) { std::fprintf(stderr, "csim_cli: cannot read %s\n", argv[1]); return 1; }
    c.assignEquationIndices();
    try {
        if (const int rc = report(c, computeDcOperatingPoint(c))) return rc;
        if (!cfg.tran.enabled) { std::printf("\nno .TRAN card: done\n"); return 0; }
        std::printf("\ntransient: tstep %.6e  tstop %.6e  tstart %.6e  ->  %s\n", cfg.tran.tstep, cfg.tran.tstop, cfg.tran.tstart,
                    csv.c_str());
        std::fflush(stdout);
        runTransientAnalysisBackwardEuler(c, cfg, csv);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "csim_cli: %s\n", e.what());
        return 1;
    }
    return 0;
}
