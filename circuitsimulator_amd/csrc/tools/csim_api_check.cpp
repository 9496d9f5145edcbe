// csim_api_check -- the declarations of the reference's headers that main() never uses, exercised the
// way a third-party caller would (tests/test_api_shims.py):
//
//   csim_api_check stamp <netlist.sp> <scale>      host, no GPU: Element::stamp of every element at
//                                                  x(i) = 0.1*(i+1), OP context -> G rows and I as hex floats;
//                                                  then one ConvController::update on fixed vectors
//   csim_api_check gs <netlist.sp>                 GPU: dcSolveGaussSeidel(ckt) and two
//                                                  Solver::solveLinearSystemGaussSeidel calls on the system
//                                                  stamped by Element::stamp
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "../api/circuit.hpp"
#include "../api/dcanalysis.hpp"
#include "../api/parser.hpp"
#include "../api/solver.hpp"

static void printVec(const char* tag, const Eigen::VectorXd& v)
{
    std::printf("%s", tag);
    for (long i = 0; i < v.size(); ++i) std::printf(" %a", v(i));
    std::printf("\n");
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::cerr << "usage: csim_api_check stamp <netlist> <scale> | gs <netlist>\n"; return 1; }
    const std::string mode = argv[1];
    Circuit ckt;
    SimulationConfig sim;
    if (!parseNetlist(argv[2], ckt, sim)) return 1;
    ckt.assignEquationIndices();
    const int N = ckt.numUnknowns();
    Eigen::MatrixXd G = Eigen::MatrixXd::Zero(N, N);
    Eigen::VectorXd I = Eigen::VectorXd::Zero(N), x(N);
    for (int i = 0; i < N; ++i) x(i) = 0.1 * (i + 1);
    AnalysisContext ctx;
    ctx.type = AnalysisType::OP;
    ctx.sourceScale = argc > 3 ? std::atof(argv[3]) : 1.0;
    for (const auto& e : ckt.elements) e->stamp(G, I, ckt, x, ctx);          // the reference's stamping loop (dcanalysis.cpp:126-128)
    if (mode == "stamp") {
        for (int r = 0; r < N; ++r) {
            std::printf("G");
            for (int c = 0; c < N; ++c) std::printf(" %a", G(r, c));
            std::printf("\n");
        }
        printVec("I", I);
        ConvController ctrl;
        Eigen::VectorXd xr(N);
        for (int i = 0; i < N; ++i) xr(i) = 0.3 - 0.05 * i;
        const ConvStatus st = ctrl.update(x, xr, 0.5, 3, ctrl.initialAlphaGS(), 2e-6, 0.4, 1e-9);
        printVec("xNext", st.xNext);
        std::printf("ctrl %a %a %a %d\n", st.alphaNext, st.gminNext, st.error, st.converged ? 1 : 0);
        return 0;
    }
    try {
        printVec("dcgs", dcSolveGaussSeidel(ckt));
        printVec("gs0", Solver::solveLinearSystemGaussSeidel(G, I, 50, 1e-12));
        printVec("gsw", Solver::solveLinearSystemGaussSeidel(G, I, x, 7, 1e-30));
    } catch (const std::exception& e) {
        std::cerr << "csim_api_check: " << e.what() << "\n";
        return 2;
    }
    return 0;
}
