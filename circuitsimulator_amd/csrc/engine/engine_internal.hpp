// engine_internal.hpp -- what a csim_engine handle holds (library-private).
#pragma once

#include <stdint.h>
#include <string>
#include <vector>

#include "../api/circuit.hpp"
#include "device_common.hpp"
#include "plan.hpp"

struct csim_engine {
    int device = 0;
    int kernelChoice = 0;                  // 0 auto, 1 general, 2 scheduled
    int schedVariant = 0;                  // 0 by batch size, 1 lean, 2 rich (CSIM_SCHED_VARIANT)
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
                                 double*, double*, long long*, unsigned*, int*, unsigned char*, int*, void*, int);
    // DC operating point of the same library (nullptr: the library carries no DC schedule)
    typedef int (*SchedDcLaunchFn)(const double*, int, double*, int*, unsigned*, unsigned char*, void*);
    void* schedLib = nullptr;
    SchedLaunchFn schedLaunch = nullptr;
    SchedDcLaunchFn schedDcLaunch = nullptr;
    std::string schedInfo;
    int32_t* dKnownAlts = nullptr;         // [nKnownAlts][N] pivot sequences the loaded kernel carries
    int nKnownAlts = 0;
    unsigned char* dFallback = nullptr;    // per-instance "a schedule check failed in this launch" mask
    int32_t* dDone = nullptr;              // per-instance steps of the current launch already completed
    int fallbackCap = 0;

    // large circuits (N > 63): dense scratch matrices in global memory, one per instance
    bool big = false;
    double* dBigScratch = nullptr;
    int bigScratchCap = 0;                 // instances

    // probe list of the most recent transient call
    int32_t* dProbe = nullptr;
    std::vector<int32_t> probeCache;
};
