// csim_cli -- command-line driver with the phases of the reference's src/main.cpp:
// parse -> assignEquationIndices -> circuit summary -> DC operating point ->
// (if .TRAN) transient CSV.  Every analysis runs on the GPU through the C++
// shims (api/analysis.cpp) over the C-ABI.
//
//   csim_cli <netlist.sp> [tran_out.csv]
#include <exception>
#include <iomanip>
#include <iostream>
#include <string>

#include "../api/circuit.hpp"
#include "../api/dcanalysis.hpp"
#include "../api/element.hpp"
#include "../api/parser.hpp"
#include "../api/sim.hpp"
#include "../api/tanalisis.hpp"

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::cerr << "Usage: csim_cli <netlist.sp> [tran_out.csv]\n";
        return 1;
    }
    const std::string netlistFile = argv[1];
    const std::string tranOutFile = argc >= 3 ? argv[2] : "tran_out.csv";

    Circuit ckt;
    SimulationConfig sim;
    std::cout << "Reading netlist: " << netlistFile << "\n";
    if (!parseNetlist(netlistFile, ckt, sim)) {
        std::cerr << "parseNetlist() failed.\n";
        return 1;
    }
    ckt.assignEquationIndices();

    std::cout << "\n==== Circuit summary ====\n"
              << "Node count   : " << ckt.nodes.size() << "\n"
              << "Element count: " << ckt.elements.size() << "\n"
              << "Unknowns     : " << ckt.numUnknowns() << "  (nodeEq=" << ckt.numNodeEquations()
              << ", branchEq=" << ckt.numVoltageBranches() << ")\n";

    std::cout << "\nRunning DC operating point...\n";
    Eigen::VectorXd xdc;
    try {
        xdc = computeDcOperatingPoint(ckt);
    } catch (const std::exception& e) {
        std::cerr << "DC solve failed: " << e.what() << "\n";
        return 1;
    }
    if (xdc.size() != ckt.numUnknowns()) {
        std::cerr << "DC solution size mismatch.\n";
        return 1;
    }

    std::cout << std::fixed << std::setprecision(6) << "\n==== DC node voltages ====\n";
    for (const Node& node : ckt.nodes) {
        if (node.eqIndex >= 0)
            std::cout << "V(" << node.name << ") = " << xdc(node.eqIndex) << " V   [eqIndex=" << node.eqIndex << "]\n";
        else
            std::cout << "V(" << node.name << ") = 0.000000 V   [GND]\n";
    }
    std::cout << "\n==== DC branch currents (voltage sources / inductors) ====\n";
    for (const auto& e : ckt.elements) {
        int k = -1;
        const char* arrow = " -> ";
        if (auto* vs = dynamic_cast<const VoltageSource*>(e.get())) { k = vs->getBranchEqIndex(); arrow = " -> -"; }
        else if (auto* ind = dynamic_cast<const Inductor*>(e.get())) k = ind->getBranchEqIndex();
        else continue;
        const double I = (k >= 0 && k < xdc.size()) ? xdc(k) : 0.0;
        const bool isV = dynamic_cast<const VoltageSource*>(e.get()) != nullptr;
        std::cout << "I(" << e->getName() << ", " << (isV ? "+" : "") << ckt.nodes[e->getNodeIds()[0]].name << arrow
                  << ckt.nodes[e->getNodeIds()[1]].name << ") = " << I << " A   [branchEq=" << k << "]\n";
    }
    std::cout << "\nDC analysis finished.\n";

    if (sim.tran.enabled) {
        std::cout << "\nRunning transient analysis (Backward Euler)...\n"
                  << std::scientific << std::setprecision(6) << "  .TRAN: tstep=" << sim.tran.tstep
                  << ", tstop=" << sim.tran.tstop << ", tstart=" << sim.tran.tstart << "\n"
                  << "  output file: " << tranOutFile << "\n";
        try {
            runTransientAnalysisBackwardEuler(ckt, sim, tranOutFile);
        } catch (const std::exception& e) {
            std::cerr << "Transient failed: " << e.what() << "\n";
            return 1;
        }
    } else {
        std::cout << "\nNo .TRAN card; transient analysis skipped.\n";
    }
    return 0;
}
