// csim_batch_demo -- the C++ batch API (api/analysis.hpp) end to end:
// parse -> index -> BatchEngine -> Monte-Carlo table -> batched DC -> batched transient.
// Prints one JSON line per instance (used by tests/test_gpu_parity.py::test_cpp_batch_api).
//
//   csim_batch_demo <netlist.sp> <B> <n_steps>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "../api/analysis.hpp"
#include "../api/parser.hpp"

int main(int argc, char** argv)
{
    if (argc != 4) { std::cerr << "usage: csim_batch_demo <netlist.sp> <B> <n_steps>\n"; return 1; }
    const int B = std::atoi(argv[2]);
    const int nSteps = std::atoi(argv[3]);
    Circuit ckt;
    SimulationConfig sim;
    if (!parseNetlist(argv[1], ckt, sim)) return 1;
    ckt.assignEquationIndices();
    try {
        csim::BatchEngine eng(ckt, 0);
        const std::vector<double> params = eng.monteCarloParams(12345, 0.05, 0, B);
        const csim::BatchDcResult dc = eng.dc(params, B);
        const int N = eng.numUnknowns();
        std::vector<int32_t> probes;
        for (int i = 0; i < N; ++i) probes.push_back(i);
        const csim::BatchTranResult tr = eng.tran(params, B, sim.tran.tstep, sim.tran.tstep * nSteps, 0.0, probes, nSteps);
        for (int b = 0; b < B; ++b) {
            std::printf("{\"b\": %d, \"dc_iters\": %d, \"dc_status\": %u, \"tran_iters\": %lld, \"status\": %u, \"rows\": %lld, "
                        "\"x_dc0\": %.17g, \"x_final\": [", b, dc.iters[b], dc.status[b], (long long)tr.iters[b], tr.status[b],
                        (long long)tr.rows, dc.x[(size_t)b * N]);
            for (int i = 0; i < N; ++i) std::printf("%s%.17g", i ? ", " : "", tr.xFinal[(size_t)b * N + i]);
            std::printf("], \"wave_last\": %.17g}\n", tr.wave[((size_t)b * tr.rows + (tr.rows - 1)) * N + (N - 1)]);
        }
    } catch (const std::exception& e) {
        std::cerr << "csim_batch_demo: " << e.what() << "\n";
        return 2;
    }
    return 0;
}
