// analysis.hpp -- the batch analysis API (C++ face of include/csim.h).
//
// Upstream's include/analysis.hpp is an empty file; here it is the home of
// what the reference does not have: DC and transient analysis of a BATCH of
// circuit instances (Monte-Carlo samples, sweep points) on the GPU.  The
// scalar entry points of dcanalysis.hpp / tanalisis.hpp are the B = 1 case of
// this class.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "circuit.hpp"
#include "csim.h"
#include "sim.hpp"

namespace csim {

struct BatchDcResult {
    std::vector<double> x;            // [B][N]
    std::vector<int32_t> iters;       // NR iterations per instance
    std::vector<uint32_t> status;     // CSIM_ST_* bits per instance
};

struct BatchTranResult {
    int64_t rows = 0;                 // rows per instance in wave
    std::vector<double> wave;         // [B][rows][n_probe] (empty if no probes)
    std::vector<double> xFinal;       // [B][N]
    std::vector<int64_t> iters;
    std::vector<uint32_t> status;
};

// One engine per (circuit, GPU).  Throws std::runtime_error when no HIP
// device is usable: there is no CPU path.
class BatchEngine {
public:
    // assignEquationIndices() must have run on ckt (as in src/main.cpp:34)
    explicit BatchEngine(const Circuit& ckt, int device = 0);
    ~BatchEngine();
    BatchEngine(const BatchEngine&) = delete;
    BatchEngine& operator=(const BatchEngine&) = delete;

    int numUnknowns() const { return ir_.view()->n_unknowns; }
    int numParams() const { return ir_.view()->n_params; }
    const CircuitIR& ir() const { return ir_; }
    csim_engine* handle() const { return eng_; }          // for the C-ABI entry points without a C++ wrapper
    const std::vector<double>& nominalParams() const { return ir_.nominal; }
    // Monte-Carlo table [B][P] (instance-major) for instances bFirst..bFirst+B-1
    std::vector<double> monteCarloParams(uint64_t seed, double sigma, int64_t bFirst, int B) const;

    // params: [B][P] instance-major; empty = nominal for every instance
    BatchDcResult dc(const std::vector<double>& params, int B);
    BatchTranResult tran(const std::vector<double>& params, int B, double tstep, double tstop, double tstart,
                         const std::vector<int32_t>& probeEq, int outStride);

    // the transient of instance `instance` of params ([B][P], empty = nominal) as the reference's CSV
    // (src/tanalisis.cpp:189-231); probeEq empty: the netlist's .PLOTNV/.PRINT probes when `sim` names any, else
    // every unknown.  Throws std::runtime_error on failure.
    void writeCsv(const std::vector<double>& params, int B, int instance, const SimulationConfig& sim,
                  const std::string& path, const std::vector<int32_t>& probeEq = {});

private:
    CircuitIR ir_;
    csim_netlist* nl_ = nullptr;      // netlist handle wrapping ir_ for the C-ABI
    csim_engine* eng_ = nullptr;
};

} // namespace csim
