// engine_internal.hpp -- what a csim_engine handle holds (library-private).
#pragma once

#include <stdint.h>
#include <string>
#include <vector>

#include "../api/circuit.hpp"
#include "device_common.hpp"
#include "plan.hpp"

// Run-time options of one engine.  The environment is read ONCE, in csim_engine_create (defaults of
// the fields below); csim_engine_set_option changes them afterwards.  Nothing on a hot path calls getenv.
struct EngineConfig {
    int hybridRounds = 4;        // hybrid_rounds  / CSIM_HYBRID_ROUNDS: hand-back rounds per transient call
    int hybridSteps = 64;        // hybrid_steps   / CSIM_HYBRID_STEPS: most steps the general kernel keeps an instance per round
    int schedVariant = 0;        // sched_variant  / CSIM_SCHED_VARIANT: 0 auto, 2 rich, 10+k sweep kernels (tuning aid)
    int lanesPerInstance = 0;    // lanes_per_instance / CSIM_LANES_PER_INSTANCE: 0 auto (by batch size), 1, 4 or 16
    bool autoJit = false;        // auto_jit       / CSIM_AUTO_JIT: host API specialises a new circuit on first use
    std::string jitDir;          // jit_dir        / CSIM_JIT_DIR (default: private per-user directory, jit.hpp)
    std::string hipcc;           // hipcc          / CSIM_HIPCC
    int jitTimeoutSec = 600;     // jit_timeout    / CSIM_JIT_TIMEOUT: wall-clock limit of one compile
    int jitDcAlts = 4;           // jit_dc_alts    / CSIM_JIT_DC_ALTS: most DC sequences a JIT kernel may carry
    bool jitDcForce = false;     // jit_dc_force   / CSIM_JIT_DC_FORCE: keep a partial DC cover (tests)
    bool hybridSync = true;      // hybrid_sync    / CSIM_HYBRID_SYNC: 1 = the host reads the device's "unfinished" flags between
                                 // the launches of the hand-over ladder and stops as soon as nothing is left (the usual
                                 // case: ONE launch per call); 0 = the whole ladder is enqueued unconditionally, every
                                 // kernel returns at once when it has nothing to do, and the call never waits on the stream
    std::string jitGenOpts;      // jit_gen_opts: generator options of this engine's JIT, "key=value,key=value" (codegen.hpp)
    bool nearTestRollback = false;   // near_test_rollback: test aid -- every verified near-threshold decision counts as a
                                 // mismatch, so the roll-back path runs (results must not change)
    bool dcFast = false;         // dc_fast        / CSIM_DC_FAST: 1 = DC operating points start on the fast generated kernel
                                 // (contraction, reciprocal pivots, guarded decisions) instead of the faithful one
};

// auxiliaries of a generated library's csim_sched_launch (same struct in the generated source, codegen.cpp)
struct csim_sched_aux {
    unsigned char* fallback; int* done; int* flags; double* work;
    double* nearX; int* nearStep; int* nearIt; long long* nearItAfter;
};

struct csim_engine {
    int device = 0;
    int kernelChoice = 0;                  // 0 auto, 1 general, 2 scheduled, 3 scheduled with the reference's arithmetic
    EngineConfig cfg;
    csim::CircuitIR cir;                   // private copy of the flattened circuit
    csim::AssemblyPlan plan;

    std::vector<void*> owned;              // every device allocation, freed on destroy

    // element tables (shared by both gather plans)
    const int32_t *dKind = nullptr, *dEq = nullptr, *dBranch = nullptr, *dSlot = nullptr,
                  *dWave = nullptr, *dWaveN = nullptr, *dTermBase = nullptr;
    csim::GenPlan gpDc{}, gpTran{};

    // Monte-Carlo recipe
    const int32_t* dMcKind = nullptr;
    const double *dNominal = nullptr, *dMu = nullptr, *dCox = nullptr, *dW = nullptr, *dL = nullptr;

    // circuit-specialised transient kernel (side library libcsim_sched_<topology>.so)
    typedef int (*SchedLaunchFn)(const double*, int, double, long long, long long, const int*, int, int,
                                 double*, double*, long long*, unsigned*, int*, const csim_sched_aux*, void*, int);
    // DC operating point of the same library (nullptr: the library carries no DC schedule)
    typedef int (*SchedDcLaunchFn)(const double*, int, double*, int*, unsigned*, unsigned char*, int*, const unsigned char*, double*, void*, int);
    void* schedLib = nullptr;
    SchedLaunchFn schedLaunch = nullptr;
    SchedDcLaunchFn schedDcLaunch = nullptr;
    std::string schedInfo;
    int schedWorkDoubles = 0;              // per-instance doubles of the work area the library's launcher wants
    double* dSchedWork = nullptr;
    int schedWorkCap = 0;                  // instances
    bool schedHasFaithful = false;         // the library carries csim_tran_faithful_kernel (launch variant 3)
    int schedGroupLanes = 0;               // 16 when the library also carries the sixteen-lanes-per-instance kernel
    int schedQuadLanes = 0;                // 4 when it carries the four-lanes-per-instance kernel too
    int schedQuadRound = 0;                // instances that kernel holds on this device at a time (one wave per SIMD)
    int numCUs = 256;                      // compute units of the device (MI355X: 256)
    int schedLinearLanes = 0;              // linear-circuit library: lanes per instance of its kernel (16 or 1), else 0
    int32_t* dKnownAlts = nullptr;         // [nKnownAlts][N] pivot sequences the loaded kernel carries
    int nKnownAlts = 0;
    unsigned char* dFallback = nullptr;    // per-instance reason an instance left a generated kernel (codegen.cpp csim_sched_aux)
    unsigned char* dFallback2 = nullptr;   // second mask of the DC chain fast -> faithful -> general
    int32_t* dDone = nullptr;              // per-instance steps of the current launch already completed
    int fallbackCap = 0;
    int32_t* dViolFlag = nullptr;          // four ints: [0] some instance unfinished, [1] near-threshold decisions to verify,
                                           // [2],[3] the same pair for the verification launch (scratch)
    int32_t* hViolFlag = nullptr;          // pinned host mirror of [0..1]
    // near-threshold verification (codegen.hpp GeneratorOptions::nearBand; kernels_verify.hip)
    double* dNearX = nullptr;              // [N][B] state at the start of an instance's first near-threshold step
    double* dVerX = nullptr;               // [N][B] the faithful kernel redoes that step here
    int32_t *dNearStep = nullptr, *dNearIt = nullptr, *dVerDone = nullptr;
    long long *dNearItAfter = nullptr, *dVerIters = nullptr;
    uint32_t* dVerStatus = nullptr;
    unsigned char* dVerFallback = nullptr;
    int nearCap = 0;
    long long nearVerified = 0, nearRolledBack = 0;   // statistics (host-visible in synchronous mode only)

    // large circuits (N > 63): dense scratch matrices in global memory, one per instance
    bool big = false;
    double* dBigScratch = nullptr;
    int bigScratchCap = 0;                 // instances

    // Gauss-Seidel DC: off-diagonal structure of the DC system per row (uploaded on first use)
    const int32_t *dGsRowPtr = nullptr, *dGsRowCol = nullptr;

    std::vector<int> netlistProbes;        // .PLOTNV / .PRINT node-voltage probes of the netlist (default CSV columns)

    // probe list of the most recent transient call
    int32_t* dProbe = nullptr;
    std::vector<int32_t> probeCache;
};
