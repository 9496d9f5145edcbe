// engine.cpp -- C-ABI of the HIP engine: handle, device-side circuit plan,
// batch entry points.  No CPU arithmetic path exists here: without a usable
// HIP device csim_engine_create fails (CSIM_ERR_NO_DEVICE).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "codegen.hpp"
#include "csim.h"
#include "engine_internal.hpp"
#include "jit.hpp"
#include "kernels.hpp"
#include "netlist_internal.hpp"
#include "plan.hpp"

using csim::setError;

#define HIPCHK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            setError(std::string(#call) + ": " + hipGetErrorString(e_));               \
            return CSIM_ERR_HIP;                                                       \
        }                                                                              \
    } while (0)

namespace {

template <class T>
int upload(csim_engine* eng, const std::vector<T>& host, const T** out)
{
    void* d = nullptr;
    const size_t bytes = sizeof(T) * (host.empty() ? 1 : host.size());
    HIPCHK(hipMalloc(&d, bytes));
    eng->owned.push_back(d);
    if (!host.empty()) HIPCHK(hipMemcpy(d, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(d);
    return CSIM_OK;
}

int fillGenPlan(csim_engine* eng, const csim::GatherPlan& g, csim::GenPlan& out)
{
    const csim_ir* ir = eng->cir.view();
    const csim::AssemblyPlan& ap = eng->plan;
    out.N = ap.N;
    out.LD = ap.LD;
    out.nNodeEq = ir->n_node_eq;
    out.nElem = ir->n_elems;
    out.P = ir->n_params;
    out.nTerms = ap.nTerms;
    out.termOne = ap.termOne;
    out.termGmin = ap.termGmin;
    out.nnzG = g.nnzG();
    out.nnzI = g.nnzI();
    out.hasNonlinear = ir->has_nonlinear;
    out.nConG = static_cast<int>(g.gCon.size());
    out.nConI = static_cast<int>(g.iCon.size());
    out.pad = 0;
    out.kind = eng->dKind; out.eq = eng->dEq; out.branch = eng->dBranch;
    out.slot = eng->dSlot; out.wave = eng->dWave; out.waveN = eng->dWaveN; out.termBase = eng->dTermBase;
    int rc;
    if ((rc = upload(eng, g.gPtr, &out.gPtr))) return rc;
    if ((rc = upload(eng, g.gPos, &out.gPos))) return rc;
    if ((rc = upload(eng, g.gCon, &out.gCon))) return rc;
    if ((rc = upload(eng, g.iPtr, &out.iPtr))) return rc;
    if ((rc = upload(eng, g.iRow, &out.iRow))) return rc;
    if ((rc = upload(eng, g.iCon, &out.iCon))) return rc;
    out.k = ir->k;
    return CSIM_OK;
}

// scratch allocation that frees itself
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

int ensureProbes(csim_engine* eng, const int32_t* probe_eq, int n_probe, const int32_t** dOut)
{
    *dOut = nullptr;
    if (n_probe <= 0) return CSIM_OK;
    if (!probe_eq) { setError("probe_eq is null but n_probe > 0"); return CSIM_ERR_ARG; }
    const int N = eng->plan.N;
    for (int i = 0; i < n_probe; ++i)
        if (probe_eq[i] < 0 || probe_eq[i] >= N) { setError("probe equation index out of range"); return CSIM_ERR_ARG; }
    if (n_probe > 1024) { setError("at most 1024 probes per launch"); return CSIM_ERR_UNSUPPORTED; }
    std::vector<int32_t> want(probe_eq, probe_eq + n_probe);
    if (want != eng->probeCache) {
        if (!eng->dProbe) {
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->dProbe), sizeof(int32_t) * 1024));
            eng->owned.push_back(eng->dProbe);
        }
        HIPCHK(hipMemcpy(eng->dProbe, want.data(), sizeof(int32_t) * want.size(), hipMemcpyHostToDevice));
        eng->probeCache = want;
    }
    *dOut = eng->dProbe;
    return CSIM_OK;
}

// zeroed N x LD scratch matrix per instance for the large-N kernels
int ensureBigScratch(csim_engine* eng, int B, hipStream_t hs)
{
    if (!eng->big || eng->bigScratchCap >= B) return CSIM_OK;
    const size_t per = csim::bigScratchBytesPerInstance(eng->gpTran);
    if (per * (size_t)B > (size_t)96 << 30) { setError("batch too large for the large-N scratch matrices (96 GiB cap)"); return CSIM_ERR_UNSUPPORTED; }
    if (eng->dBigScratch) HIPCHK(hipFree(eng->dBigScratch));
    eng->dBigScratch = nullptr;
    eng->bigScratchCap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->dBigScratch), per * (size_t)B));
    HIPCHK(hipMemsetAsync(eng->dBigScratch, 0, per * (size_t)B, hs));
    eng->bigScratchCap = B;
    return CSIM_OK;
}

// directory this shared library was loaded from
std::string ownDirectory()
{
    Dl_info info;
    if (!dladdr(reinterpret_cast<void*>(&csim_engine_destroy), &info) || !info.dli_fname) return ".";
    const std::string path(info.dli_fname);
    const std::size_t slash = path.find_last_of('/');
    return slash == std::string::npos ? std::string(".") : path.substr(0, slash);
}

// the alternatives a generated library carries -> device table for the general kernel's hand-back test
void adoptScheduleTable(csim_engine* eng, void* lib)
{
    typedef const int* (*AltsFn)(int*, int*);
    AltsFn fn = reinterpret_cast<AltsFn>(dlsym(lib, "csim_sched_alts"));
    eng->nKnownAlts = 0;
    if (eng->dKnownAlts) { (void)hipFree(eng->dKnownAlts); eng->dKnownAlts = nullptr; }
    typedef int (*LanesFn)(void);
    LanesFn lanesFn = reinterpret_cast<LanesFn>(dlsym(lib, "csim_sched_group_lanes"));
    eng->schedGroupLanes = lanesFn ? lanesFn() : 0;
    LanesFn quadFn = reinterpret_cast<LanesFn>(dlsym(lib, "csim_sched_group4_lanes"));
    eng->schedQuadLanes = quadFn ? quadFn() : 0;
    LanesFn quadCuFn = reinterpret_cast<LanesFn>(dlsym(lib, "csim_sched_group4_per_cu"));
    eng->schedQuadRound = (eng->schedQuadLanes == 4 && quadCuFn) ? quadCuFn() * eng->numCUs : 0;
    LanesFn linFn = reinterpret_cast<LanesFn>(dlsym(lib, "csim_sched_linear_lanes"));
    eng->schedLinearLanes = linFn ? linFn() : 0;
    LanesFn faithFn = reinterpret_cast<LanesFn>(dlsym(lib, "csim_sched_has_faithful"));
    eng->schedHasFaithful = faithFn && faithFn() != 0;
    if (eng->kernelChoice == 3 && !eng->schedHasFaithful) eng->kernelChoice = 0;    // a reload dropped the faithful kernel
    LanesFn workFn = reinterpret_cast<LanesFn>(dlsym(lib, "csim_sched_work_doubles"));
    {
        const int need = workFn ? workFn() : 0;
        if (need != eng->schedWorkDoubles) eng->schedWorkCap = 0;        // another library: its work area has another size
        eng->schedWorkDoubles = need;
    }
    // DC operating-point kernel, present when the library was generated with "dc" schedules
    eng->schedDcLaunch = nullptr;
    if (AltsFn dcAlts = reinterpret_cast<AltsFn>(dlsym(lib, "csim_sched_dc_alts"))) {
        int nDc = 0, nn = 0;
        (void)dcAlts(&nDc, &nn);
        if (nDc > 0 && nn == eng->plan.N)
            eng->schedDcLaunch = reinterpret_cast<csim_engine::SchedDcLaunchFn>(dlsym(lib, "csim_sched_dc_launch"));
    }
    if (!fn) return;
    int nAlts = 0, n = 0;
    const int* table = fn(&nAlts, &n);
    if (!table || nAlts <= 0 || n != eng->plan.N) return;
    if (hipMalloc(reinterpret_cast<void**>(&eng->dKnownAlts), sizeof(int32_t) * (size_t)nAlts * n) != hipSuccess) return;
    if (hipMemcpy(eng->dKnownAlts, table, sizeof(int32_t) * (size_t)nAlts * n, hipMemcpyHostToDevice) != hipSuccess) return;
    eng->nKnownAlts = nAlts;
}

int envInt(const char* name, int dflt)
{
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

// the only place the engine reads the environment: once per csim_engine_create
EngineConfig configFromEnvironment()
{
    EngineConfig c;
    c.hybridRounds = std::max(0, envInt("CSIM_HYBRID_ROUNDS", c.hybridRounds));
    c.hybridSteps = std::max(1, envInt("CSIM_HYBRID_STEPS", c.hybridSteps));
    c.schedVariant = envInt("CSIM_SCHED_VARIANT", c.schedVariant);
    c.lanesPerInstance = envInt("CSIM_LANES_PER_INSTANCE", c.lanesPerInstance);
    c.autoJit = std::getenv("CSIM_AUTO_JIT") != nullptr;
    c.jitDir = csim::jitDefaultDir();
    const char* cc = std::getenv("CSIM_HIPCC");
    c.hipcc = (cc && *cc) ? cc : "/opt/rocm/bin/hipcc";
    c.jitTimeoutSec = std::max(1, envInt("CSIM_JIT_TIMEOUT", c.jitTimeoutSec));
    c.jitDcAlts = std::max(0, std::min(8, envInt("CSIM_JIT_DC_ALTS", c.jitDcAlts)));
    c.jitDcForce = std::getenv("CSIM_JIT_DC_FORCE") != nullptr;
    c.hybridSync = envInt("CSIM_HYBRID_SYNC", 1) != 0;
    c.dcFast = envInt("CSIM_DC_FAST", 0) != 0;
    if (c.lanesPerInstance != 0 && c.lanesPerInstance != 1 && c.lanesPerInstance != 4 && c.lanesPerInstance != 16) c.lanesPerInstance = 0;
    return c;
}

// look for a generated kernel of this topology next to libcsim.so
void loadScheduledKernel(csim_engine* eng)
{
    const csim_ir* ir = eng->cir.view();
    const unsigned long long topo = csim::scheduleHash(*ir, csim::PivotSchedule::identity(ir->n_unknowns));
    char name[64];
    std::snprintf(name, sizeof name, "/libcsim_sched_%016llx.so", topo);
    const std::string path = ownDirectory() + name;
    void* lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib) return;
    typedef unsigned long long (*HashFn)(void);
    typedef const char* (*InfoFn)(void);
    HashFn topoFn = reinterpret_cast<HashFn>(dlsym(lib, "csim_sched_topology"));
    InfoFn infoFn = reinterpret_cast<InfoFn>(dlsym(lib, "csim_sched_info"));
    auto launch = reinterpret_cast<csim_engine::SchedLaunchFn>(dlsym(lib, "csim_sched_launch"));
    if (!topoFn || !launch || topoFn() != topo) { dlclose(lib); return; }
    eng->schedLib = lib;
    eng->schedLaunch = launch;
    eng->schedInfo = infoFn ? infoFn() : "";
    adoptScheduleTable(eng, lib);
}

} // namespace

extern "C" {

int csim_engine_create(const csim_netlist* nl, int32_t device, csim_engine** out)
{
    if (!nl || !out) { setError("csim_engine_create: null argument"); return CSIM_ERR_ARG; }
    *out = nullptr;
    int count = 0;
    const hipError_t ce = hipGetDeviceCount(&count);
    if (ce != hipSuccess || count <= 0 || device < 0 || device >= count) {
        setError("csim_engine_create: no usable HIP device (this library has no CPU path)");
        return CSIM_ERR_NO_DEVICE;
    }
    const csim_ir* ir = nl->cir.view();
    if (ir->n_unknowns <= 0) { setError("circuit has no unknowns"); return CSIM_ERR_EMPTY; }
    {
        // the reference takes any N (src/circuit.cpp:38-40); here the general kernels bound it: N <= 63 runs
        // LDS-resident, larger circuits need their structure bit matrix, terms and three N-vectors in one
        // CU's LDS (kernels_big.hip): N = 1024 for sparse netlists of that size, checked exactly below
        const csim::AssemblyPlan probe = csim::buildAssemblyPlan(*ir);
        if (ir->n_unknowns > 63 && !csim::bigSupports(ir->n_unknowns, probe.nTerms, ir->n_params)) {
            setError("circuit too large: the general kernels need N <= 1024 and the structure in 160 KB of LDS");
            return CSIM_ERR_UNSUPPORTED;
        }
    }
    HIPCHK(hipSetDevice(device));

    auto* eng = new csim_engine();
    eng->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) eng->numCUs = cus; }
    eng->cfg = configFromEnvironment();
    eng->cir = nl->cir;
    eng->cir.view();
    eng->netlistProbes = nl->probeEq;
    eng->plan = csim::buildAssemblyPlan(*eng->cir.view());
    eng->big = ir->n_unknowns > 63;

    int rc = CSIM_OK;
    const csim::CircuitIR& c = eng->cir;
    if (!rc) rc = upload(eng, c.kind, &eng->dKind);
    if (!rc) rc = upload(eng, c.eq, &eng->dEq);
    if (!rc) rc = upload(eng, c.branchEq, &eng->dBranch);
    if (!rc) rc = upload(eng, c.paramSlot, &eng->dSlot);
    if (!rc) rc = upload(eng, c.wave, &eng->dWave);
    if (!rc) rc = upload(eng, c.waveN, &eng->dWaveN);
    if (!rc) rc = upload(eng, eng->plan.termBase, &eng->dTermBase);
    if (!rc) rc = upload(eng, c.mcKind, &eng->dMcKind);
    if (!rc) rc = upload(eng, c.nominal, &eng->dNominal);
    if (!rc) rc = upload(eng, c.mcMu, &eng->dMu);
    if (!rc) rc = upload(eng, c.mcCox, &eng->dCox);
    if (!rc) rc = upload(eng, c.mcW, &eng->dW);
    if (!rc) rc = upload(eng, c.mcL, &eng->dL);
    if (!rc) rc = fillGenPlan(eng, eng->plan.dc, eng->gpDc);
    if (!rc) rc = fillGenPlan(eng, eng->plan.tran, eng->gpTran);
    if (rc) { csim_engine_destroy(eng); return rc; }
    loadScheduledKernel(eng);
    *out = eng;
    return CSIM_OK;
}

void csim_engine_destroy(csim_engine* eng)
{
    if (!eng) return;
    (void)hipSetDevice(eng->device);
    for (void* p : eng->owned) (void)hipFree(p);
    if (eng->dFallback) (void)hipFree(eng->dFallback);
    if (eng->dFallback2) (void)hipFree(eng->dFallback2);
    if (eng->dDone) (void)hipFree(eng->dDone);
    for (void* p : {(void*)eng->dNearX, (void*)eng->dVerX, (void*)eng->dNearStep, (void*)eng->dNearIt, (void*)eng->dVerDone,
                    (void*)eng->dNearItAfter, (void*)eng->dVerIters, (void*)eng->dVerStatus, (void*)eng->dVerFallback})
        if (p) (void)hipFree(p);
    if (eng->dSchedWork) (void)hipFree(eng->dSchedWork);
    if (eng->dViolFlag) (void)hipFree(eng->dViolFlag);
    if (eng->hViolFlag) (void)hipHostFree(eng->hViolFlag);
    if (eng->dKnownAlts) (void)hipFree(eng->dKnownAlts);
    if (eng->dBigScratch) (void)hipFree(eng->dBigScratch);
    if (eng->schedLib) dlclose(eng->schedLib);
    delete eng;
}

const char* csim_engine_tran_kernel(const csim_engine* eng)
{
    if (!eng) return "";
    if (!(eng->schedLaunch && eng->kernelChoice != 1)) return "general";
    return (eng->kernelChoice == 3 && eng->schedHasFaithful) ? "faithful" : "scheduled";
}

const char* csim_engine_sched_info(const csim_engine* eng)
{
    return (eng && eng->schedLaunch) ? eng->schedInfo.c_str() : "";
}

int csim_engine_set_kernel(csim_engine* eng, int32_t which)
{
    if (!eng || which < 0 || which > 3) return CSIM_ERR_ARG;
    if (which == 3 && !(eng->schedLaunch && eng->schedHasFaithful)) {
        setError("no faithful scheduled kernel is available for this circuit");
        return CSIM_ERR_UNSUPPORTED;
    }
    if (which == 2 && !eng->schedLaunch) {
        setError("no scheduled kernel is available for this circuit (libcsim_sched_<topology>.so not found)");
        return CSIM_ERR_UNSUPPORTED;
    }
    eng->kernelChoice = which;
    return CSIM_OK;
}

int csim_engine_set_option(csim_engine* eng, const char* key, const char* value)
{
    if (!eng || !key || !value) { setError("csim_engine_set_option: null argument"); return CSIM_ERR_ARG; }
    const std::string k(key), v(value);
    EngineConfig& c = eng->cfg;
    const int iv = std::atoi(value);
    if (k == "hybrid_rounds") c.hybridRounds = std::max(0, iv);
    else if (k == "hybrid_steps") c.hybridSteps = std::max(1, iv);
    else if (k == "sched_variant") c.schedVariant = iv;
    else if (k == "lanes_per_instance") {
        if (iv != 0 && iv != 1 && iv != 4 && iv != 16) { setError("lanes_per_instance must be 0 (auto), 1, 4 or 16"); return CSIM_ERR_ARG; }
        if (iv == 4 && eng->schedLaunch && eng->schedQuadLanes != 4) {
            setError("lanes_per_instance=4: the loaded kernel library has no four-lanes-per-instance kernel");
            return CSIM_ERR_ARG;
        }
        if (iv == 16 && eng->schedLaunch && eng->schedGroupLanes != 16) {
            setError("lanes_per_instance=16: the loaded kernel library has no sixteen-lanes-per-instance kernel");
            return CSIM_ERR_UNSUPPORTED;
        }
        c.lanesPerInstance = iv;
    }
    else if (k == "auto_jit") c.autoJit = iv != 0;
    else if (k == "jit_dir") c.jitDir = v;
    else if (k == "hipcc") c.hipcc = v;
    else if (k == "jit_timeout") c.jitTimeoutSec = std::max(1, iv);
    else if (k == "jit_dc_alts") c.jitDcAlts = std::max(0, std::min(8, iv));
    else if (k == "jit_dc_force") c.jitDcForce = iv != 0;
    else if (k == "jit_gen_opts") c.jitGenOpts = v;
    else if (k == "near_test_rollback") c.nearTestRollback = iv != 0;
    else if (k == "hybrid_sync") c.hybridSync = iv != 0;
    else if (k == "dc_fast") c.dcFast = iv != 0;
    else { setError("csim_engine_set_option: unknown option '" + k + "'"); return CSIM_ERR_ARG; }
    return CSIM_OK;
}

int64_t csim_engine_stat(const csim_engine* eng, const char* key)
{
    if (!eng || !key) return -1;
    const std::string k(key);
    if (k == "near_verified") return eng->nearVerified;
    if (k == "near_rolled_back") return eng->nearRolledBack;
    return -1;
}

int csim_mc_params_dev(csim_engine* eng, uint64_t seed, double sigma, int64_t b_first,
                       int32_t B, double* d_params, void* stream)
{
    if (!eng || !d_params || B < 0 || b_first < 0) { setError("csim_mc_params_dev: bad argument"); return CSIM_ERR_ARG; }
    HIPCHK(hipSetDevice(eng->device));
    HIPCHK(csim::launchMcParams(eng->cir.ir.n_params, B, b_first, seed, sigma, eng->dMcKind, eng->dNominal,
                                eng->dMu, eng->dCox, eng->dW, eng->dL, d_params,
                                static_cast<hipStream_t>(stream)));
    return CSIM_OK;
}

// The device's flag words of the hand-over protocol -> host.  Synchronous mode (cfg.hybridSync) only: the call
// waits for the stream here, because which launches follow depends on the answer.
static int readFlags(csim_engine* eng, hipStream_t hs, bool* unfinished, bool* toVerify)
{
    HIPCHK(hipMemcpyAsync(eng->hViolFlag, eng->dViolFlag, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, hs));
    HIPCHK(hipStreamSynchronize(hs));
    *unfinished = eng->hViolFlag[0] != 0;
    if (toVerify) *toVerify = eng->hViolFlag[1] != 0;
    return CSIM_OK;
}

// which kernel of the generated library a launch of B instances uses (csim_sched_launch's `variant`):
// 0 = lane per instance, 16 = sixteen lanes per instance, 4 = four lanes per instance, other values = tuning kernels
static int schedVariantFor(const csim_engine* eng, int32_t B)
{
    if (eng->cfg.schedVariant != 0) return eng->cfg.schedVariant;
    // (a library without the sixteen-lane kernel -- a linear circuit's, or one loaded after the option was set --
    // runs its lane-per-instance kernel whatever the option says)
    if (eng->cfg.lanesPerInstance == 16 && eng->schedGroupLanes == 16) return 16;
    if (eng->cfg.lanesPerInstance == 4 && eng->schedQuadLanes == 4) return 4;
    if (eng->cfg.lanesPerInstance != 0) return 0;
    // auto: sixteen lanes per instance while the batch is too small to give every SIMD a wave of 64
    // instances.  Measured on dbmixer (gpurun_out/sw9): 16 lanes 2.4-2.5e9 NR-iter*inst/s from B = 4096 up (one
    // wave per SIMD, further instances run as further rounds), one lane 8.6e8 at B = 4096 growing linearly --
    // 1.70e9 at 8192, 2.55e9 at 12 288 -- to 1.25e10 at 65 536: they cross near B = 11 400.  Both kernels carry the
    // same set of pivot schedules.
    // Four lanes per instance (circuits of up to 32 unknowns) sit between the two: 16 instances per wavefront, one
    // wave per SIMD, so a round of 16 384 instances takes what one wave takes -- on dbmixer 30.6 ms per 1000 steps against
    // 17.5 ms per round of 4096 for sixteen lanes and 52.4 ms for one lane (gpurun_out/r03q/probe3.log): the best of
    // the three from 4097 to 16 384 instances (5.83e9 at 16 384, where one lane gives 3.38e9), never beyond (two
    // rounds take longer than the lane-per-instance kernel's one).
    // In general: beyond one round of the sixteen-lane kernel (4 instances x 4 SIMDs per CU) and up to one round of the
    // four-lane kernel -- 64 instances per CU when four of its workgroups fit a CU's LDS, fewer otherwise (the library
    // says: csim_sched_group4_per_cu).
    if (eng->schedQuadLanes == 4 && B > 16 * eng->numCUs && B <= eng->schedQuadRound) return 4;
    return (eng->schedGroupLanes == 16 && B <= 44 * eng->numCUs) ? 16 : 0;
}

extern "C" int csim_engine_lanes_for_batch(const csim_engine* eng, int32_t B)
{
    if (!eng || !eng->schedLaunch || eng->kernelChoice == 1) return 0;
    if (eng->schedLinearLanes) return eng->schedLinearLanes;       // a linear circuit's library has one transient kernel
    const int v = schedVariantFor(eng, B);
    return v == 16 ? 16 : (v == 4 ? 4 : 1);
}

// per-instance hand-over masks, progress counters and flag words of the generated kernels
static int ensureFallbackBuffers(csim_engine* eng, int32_t B)
{
    if (!eng->dViolFlag) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->dViolFlag), 8 * sizeof(int32_t)));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&eng->hViolFlag), 4 * sizeof(int32_t), hipHostMallocDefault));
    }
    if (eng->fallbackCap >= B) return CSIM_OK;
    if (eng->dFallback) HIPCHK(hipFree(eng->dFallback));
    if (eng->dFallback2) HIPCHK(hipFree(eng->dFallback2));
    if (eng->dDone) HIPCHK(hipFree(eng->dDone));
    eng->dFallback = eng->dFallback2 = nullptr;
    eng->dDone = nullptr;
    eng->fallbackCap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->dFallback), (size_t)B));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->dFallback2), (size_t)B));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->dDone), sizeof(int32_t) * (size_t)B));
    eng->fallbackCap = B;
    return CSIM_OK;
}

// work area of a linear circuit's generated kernels (their factor tape)
static int ensureSchedWork(csim_engine* eng, int32_t B)
{
    if (eng->schedWorkDoubles <= 0 || eng->schedWorkCap >= B) return CSIM_OK;
    if (eng->dSchedWork) HIPCHK(hipFree(eng->dSchedWork));
    eng->dSchedWork = nullptr;
    eng->schedWorkCap = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&eng->dSchedWork), sizeof(double) * (size_t)eng->schedWorkDoubles * ((size_t)B + 64)));   // whole workgroups
    eng->schedWorkCap = B;
    return CSIM_OK;
}

// buffers of the near-threshold verification (kernels_verify.hip)
static int ensureNearBuffers(csim_engine* eng, int32_t B)
{
    if (eng->nearCap >= B) return CSIM_OK;
    void** all[] = {reinterpret_cast<void**>(&eng->dNearX), reinterpret_cast<void**>(&eng->dVerX),
                    reinterpret_cast<void**>(&eng->dNearStep), reinterpret_cast<void**>(&eng->dNearIt),
                    reinterpret_cast<void**>(&eng->dVerDone), reinterpret_cast<void**>(&eng->dNearItAfter),
                    reinterpret_cast<void**>(&eng->dVerIters), reinterpret_cast<void**>(&eng->dVerStatus),
                    reinterpret_cast<void**>(&eng->dVerFallback)};
    for (void** p : all) { if (*p) HIPCHK(hipFree(*p)); *p = nullptr; }
    eng->nearCap = 0;
    const size_t N = (size_t)eng->plan.N, n = (size_t)B;
    HIPCHK(hipMalloc(all[0], sizeof(double) * N * n));
    HIPCHK(hipMalloc(all[1], sizeof(double) * N * n));
    HIPCHK(hipMalloc(all[2], sizeof(int32_t) * n));
    HIPCHK(hipMalloc(all[3], sizeof(int32_t) * n));
    HIPCHK(hipMalloc(all[4], sizeof(int32_t) * n));
    HIPCHK(hipMalloc(all[5], sizeof(long long) * n));
    HIPCHK(hipMalloc(all[6], sizeof(long long) * n));
    HIPCHK(hipMalloc(all[7], sizeof(uint32_t) * n));
    HIPCHK(hipMalloc(all[8], n));
    eng->nearCap = B;
    return CSIM_OK;
}

int csim_dc_batch_dev(csim_engine* eng, const double* d_params, int32_t B, double* d_x,
                      int32_t* d_iters, uint32_t* d_status, void* stream)
{
    if (!eng || !d_params || !d_x || !d_iters || !d_status || B < 0) {
        setError("csim_dc_batch_dev: bad argument");
        return CSIM_ERR_ARG;
    }
    if (B == 0) return CSIM_OK;
    HIPCHK(hipSetDevice(eng->device));
    hipStream_t hs = static_cast<hipStream_t>(stream);
    // the kernels with run-time pivoting: every instance, or the ones a mask names
    auto generalDc = [&](const unsigned char* only) -> int {
        if (eng->big) {
            const int rc = ensureBigScratch(eng, B, hs);
            if (rc) return rc;
            HIPCHK(csim::launchDcBig(eng->gpDc, d_params, B, eng->dBigScratch, d_x, d_iters, d_status, hs, only));
        } else {
            HIPCHK(csim::launchDcGeneral(eng->gpDc, d_params, B, d_x, d_iters, d_status, hs, only));
        }
        return CSIM_OK;
    };
    if (eng->schedDcLaunch && eng->kernelChoice != 1) {
        // Generated lane-per-instance kernels on the recorded DC pivot sequences, as a chain in which every
        // kernel replays, from x = 0, exactly the instances the one before could not finish (mask):
        //   [fast kernel (option dc_fast): contraction, reciprocal pivots; leaves an instance on a failed pivot
        //    check, a non-finite solve, a final ramp step at the NR cap, or a controller decision within its
        //    rounding noise (src/dcanalysis.cpp:150,285-296) ->]
        //   faithful kernel: the reference's operations, bit for bit the general kernel's operating points;
        //    leaves an instance on a failed pivot check or a non-finite solve ->
        //   general kernel (run-time pivoting).
        int rc = ensureFallbackBuffers(eng, B);
        if (!rc) rc = ensureSchedWork(eng, B);
        if (rc) return rc;
        const bool sync = eng->cfg.hybridSync;
        const bool fast = eng->cfg.dcFast && eng->kernelChoice != 3 && eng->schedLinearLanes == 0;
        auto stage = [&](int variant, unsigned char* leaves, const unsigned char* only, bool* any) -> int {
            HIPCHK(hipMemsetAsync(leaves, 0, (size_t)B, hs));
            HIPCHK(hipMemsetAsync(eng->dViolFlag, 0, 4 * sizeof(int32_t), hs));
            if (eng->schedDcLaunch(d_params, B, d_x, d_iters, d_status, leaves, eng->dViolFlag, only, eng->dSchedWork, stream, variant) != 0) {
                setError("scheduled DC kernel launch failed");
                return CSIM_ERR_HIP;
            }
            *any = true;
            return sync ? readFlags(eng, hs, any, nullptr) : CSIM_OK;
        };
        bool any = false;
        const unsigned char* left = nullptr;
        if (fast) {
            if (const int frc = stage(2, eng->dFallback, nullptr, &any)) return frc;
            if (!any) return CSIM_OK;
            left = eng->dFallback;
        }
        unsigned char* mine = fast ? eng->dFallback2 : eng->dFallback;
        if (const int frc = stage(0, mine, left, &any)) return frc;
        if (!any) return CSIM_OK;
        return generalDc(mine);
    }
    return generalDc(nullptr);
}

int csim_tran_batch_dev(csim_engine* eng, const double* d_params, int32_t B, double tstep,
                        int64_t step_first, int64_t n_steps, const int32_t* probe_eq, int32_t n_probe,
                        int32_t out_stride, double* d_wave, double* d_x, int64_t* d_iters,
                        uint32_t* d_status, int32_t* d_step_iters, void* stream)
{
    if (!eng || !d_params || !d_x || !d_iters || !d_status || B < 0 || step_first < 0 || n_steps < 0) {
        setError("csim_tran_batch_dev: bad argument");
        return CSIM_ERR_ARG;
    }
    if (!(tstep > 0.0)) { setError("tstep must be > 0"); return CSIM_ERR_CONFIG; }
    if (step_first + n_steps > 2147483647LL) { setError("step index exceeds int range of the reference"); return CSIM_ERR_CONFIG; }
    if (B == 0) return CSIM_OK;
    HIPCHK(hipSetDevice(eng->device));
    const int32_t* dProbe = nullptr;
    if (d_wave) {
        if (out_stride <= 0) { setError("out_stride must be >= 1"); return CSIM_ERR_ARG; }
        const int rc = ensureProbes(eng, probe_eq, n_probe, &dProbe);
        if (rc) return rc;
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const int np = d_wave ? n_probe : 0, os = d_wave ? out_stride : 1;
    // handBack: stop an instance after the first step that ran on recorded sequences only
    auto general = [&](int32_t* dDone, int maxSteps, bool handBack = false) -> int {
        if (eng->big) {
            const int rc = ensureBigScratch(eng, B, hs);
            if (rc) return rc;
            HIPCHK(csim::launchTranBig(eng->gpTran, d_params, B, tstep, step_first, n_steps, dProbe, np, os, d_wave, d_x,
                                       reinterpret_cast<long long*>(d_iters), d_status, d_step_iters, nullptr,
                                       eng->dBigScratch, nullptr, hs, nullptr, -1, dDone, maxSteps));
        } else {
            HIPCHK(csim::launchTranGeneral(eng->gpTran, d_params, B, tstep, step_first, n_steps, dProbe, np, os, d_wave, d_x,
                                           reinterpret_cast<long long*>(d_iters), d_status, d_step_iters, nullptr, hs,
                                           nullptr, -1, dDone, maxSteps, handBack ? eng->dKnownAlts : nullptr,
                                           handBack ? eng->nKnownAlts : 0));
        }
        return CSIM_OK;
    };
    // n_steps == 0 (a run shorter than one step is legal upstream and yields the t = 0 row only): the
    // general kernel writes that row; the hybrid sequence below would return before it
    if (!(eng->schedLaunch && eng->kernelChoice != 1) || n_steps == 0) return general(nullptr, 0);

    // Fast path with a hand-over ladder.  The generated kernel advances every instance until the launch is
    // complete or one of its checks fails (no recorded schedule fits that factorisation, or the Newton
    // iteration of a step is slow, plan.hpp slowStepIters); it keeps the state at the start of the failing
    // step and records per-instance progress in dDone.  Device flag [0] says whether any instance is
    // unfinished: if none is -- the usual case -- the call ends after ONE launch.
    //
    // Near-threshold decisions (codegen.hpp GeneratorOptions::nearBand).  A fast kernel that takes an
    // `err < tol` decision (src/tanalisis.cpp:369) within its own rounding noise goes on speculatively and
    // leaves the state at the start of that step; flag [1] then asks for a verification: the FAITHFUL
    // generated kernel (the reference's arithmetic on the recorded sequences) redoes exactly that step for
    // exactly those instances (kernels_verify.hip), and an instance whose pass count differs is rolled back
    // to that step and handed to the ladder with reason 2: the faithful kernel runs that one step, the fast
    // kernel has the instance back.
    //
    // Ladder for unfinished instances: (1) the faithful kernel -- slow / non-convergent steps are redone there
    // bit-faithfully at lane-per-instance speed; it leaves an instance only when no recorded sequence fits;
    // (2) the fast kernel again (instances handed back after their one faithful step); (3) the general
    // kernel with run-time pivoting, which hands an instance back once a whole step ran on recorded
    // sequences again; at most cfg.hybridRounds rounds, then the general kernel runs what is left.
    //
    // cfg.hybridSync (default): the host reads the flags after each stage and returns as soon as nothing is
    // left.  With hybrid_sync = 0 the call never waits: the same sequence of launches is enqueued
    // unconditionally (every kernel returns at once when it finds nothing to do; 10 launches per hybrid round,
    // a few microseconds each) -- bit for bit the synchronous results, for callers that overlap streams.
    {
        const int rc = ensureFallbackBuffers(eng, B);
        if (rc) return rc;
    }
    const bool linearLib = eng->schedWorkDoubles > 0;
    const bool canVerify = !linearLib && eng->schedHasFaithful;
    if (!linearLib) {
        const int rc = ensureNearBuffers(eng, B);
        if (rc) return rc;
    }
    HIPCHK(hipMemsetAsync(eng->dFallback, 0, (size_t)B, hs));
    HIPCHK(hipMemsetAsync(eng->dDone, 0, sizeof(int32_t) * (size_t)B, hs));
    HIPCHK(hipMemsetAsync(eng->dViolFlag, 0, 8 * sizeof(int32_t), hs));
    if (!linearLib) HIPCHK(hipMemsetAsync(eng->dNearStep, 0, sizeof(int32_t) * (size_t)B, hs));
    if (linearLib) {                               // factor store of the linear-circuit kernels
        const int rc = ensureSchedWork(eng, B);
        if (rc) return rc;
    }
    const bool sync = eng->cfg.hybridSync;
    csim_sched_aux aux{eng->dFallback, eng->dDone, eng->dViolFlag, eng->dSchedWork,
                       eng->dNearX, eng->dNearStep, eng->dNearIt, eng->dNearItAfter};
    auto launchSched = [&](int variant, const csim_sched_aux& a, double* x, long long* it, uint32_t* st, bool outputs) -> int {
        const int lrc = eng->schedLaunch(d_params, B, tstep, step_first, n_steps, outputs ? dProbe : nullptr, outputs ? np : 0,
                                         outputs ? os : 1, outputs ? d_wave : nullptr, x, it, st,
                                         outputs ? d_step_iters : nullptr, &a, stream, variant);
        if (lrc != 0) { setError(std::string("scheduled kernel launch: ") + hipGetErrorString((hipError_t)lrc)); return CSIM_ERR_HIP; }
        return CSIM_OK;
    };
    // the faithful kernel redoes the flagged steps on scratch copies; the pass counts decide (kernels_verify.hip)
    auto verify = [&]() -> int {
        HIPCHK(csim::launchNearPrep(B, eng->plan.N, n_steps, eng->dNearStep, eng->dNearX, eng->dVerX, eng->dVerDone,
                                    eng->dVerIters, eng->dVerStatus, eng->dVerFallback, hs));
        const csim_sched_aux va{eng->dVerFallback, eng->dVerDone, eng->dViolFlag + 4, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (const int rc = launchSched(3, va, eng->dVerX, eng->dVerIters, eng->dVerStatus, false)) return rc;
        HIPCHK(csim::launchNearResolve(B, eng->plan.N, eng->dNearStep, eng->dNearIt, eng->dNearItAfter, eng->dNearX,
                                       eng->dVerDone, eng->dVerIters, d_x, eng->dDone,
                                       reinterpret_cast<long long*>(d_iters), eng->dFallback, eng->dViolFlag,
                                       eng->cfg.nearTestRollback, hs));
        return CSIM_OK;
    };
    // one launch of a generated kernel on the caller's buffers (+ its verification); *unfinished: may any
    // instance be unfinished afterwards (asynchronous mode: always "maybe")
    auto scheduled = [&](int variant, bool* unfinished) -> int {
        HIPCHK(hipMemsetAsync(eng->dViolFlag, 0, 2 * sizeof(int32_t), hs));
        if (const int rc = launchSched(variant, aux, d_x, reinterpret_cast<long long*>(d_iters), d_status, true)) return rc;
        const bool speculates = canVerify && variant != 3;
        *unfinished = true;
        if (!sync) return speculates ? verify() : CSIM_OK;
        bool toVerify = false;
        if (const int rc = readFlags(eng, hs, unfinished, &toVerify)) return rc;
        if (speculates && toVerify) {
            if (const int rc = verify()) return rc;
            if (const int rc = readFlags(eng, hs, unfinished, nullptr)) return rc;
            eng->nearVerified += eng->hViolFlag[2];
            eng->nearRolledBack += eng->hViolFlag[3];
            HIPCHK(hipMemsetAsync(eng->dViolFlag + 2, 0, 2 * sizeof(int32_t), hs));
        }
        return CSIM_OK;
    };
    const bool faithfulOnly = eng->kernelChoice == 3 && eng->schedHasFaithful;
    const int fast = faithfulOnly ? 3 : schedVariantFor(eng, B);
    bool unfinished = false;
    int rc = scheduled(fast, &unfinished);
    if (rc || !unfinished) return rc;
    const int rounds = eng->cfg.hybridRounds;
    for (int r = 0; r <= rounds; ++r) {
        if (eng->schedHasFaithful && !faithfulOnly) {
            if ((rc = scheduled(3, &unfinished)) || !unfinished) return rc;
            // instances that had one faithful step go on here; one that no recorded sequence fits stops again at once
            if ((rc = scheduled(fast, &unfinished)) || !unfinished) return rc;
        }
        if (r == rounds) break;
        if ((rc = general(eng->dDone, eng->cfg.hybridSteps, true))) return rc;
        if ((rc = scheduled(fast, &unfinished)) || !unfinished) return rc;
    }
    return general(eng->dDone, 2147483647);      // whatever is still unfinished runs to the end of the launch
}

int64_t csim_tran_num_steps(double tstep, double tstop)
{
    if (!(tstep > 0.0) || !(tstop > 0.0)) return -1;
    return static_cast<int64_t>(static_cast<int>(std::floor(tstop / tstep + 1e-12)));   // tanalisis.cpp:238
}

int64_t csim_tran_num_rows(double tstep, double tstop, double tstart, int32_t out_stride)
{
    const int64_t ns = csim_tran_num_steps(tstep, tstop);
    if (ns < 0 || out_stride <= 0) return -1;
    int64_t rows = 0;
    for (int64_t r = 0; r <= ns / out_stride; ++r) {
        const double t = (r == 0) ? 0.0 : static_cast<double>(static_cast<int>(r * out_stride)) * tstep;
        if (!(t < tstart)) ++rows;                                                       // tanalisis.cpp:209
    }
    return rows;
}

// ---- host-pointer forms ---------------------------------------------------

static int stageParams(csim_engine* eng, const double* params, int B, DevBuf& dParams)
{
    const int P = eng->cir.ir.n_params;
    HIPCHK(dParams.alloc(sizeof(double) * (size_t)P * (size_t)B));
    if (params) {
        DevBuf tmp;
        HIPCHK(tmp.alloc(sizeof(double) * (size_t)P * (size_t)B));
        HIPCHK(hipMemcpy(tmp.p, params, sizeof(double) * (size_t)P * (size_t)B, hipMemcpyHostToDevice));
        HIPCHK(csim::launchTranspose(tmp.as<double>(), dParams.as<double>(), B, P, nullptr));   // [B][P] -> [P][B]
        HIPCHK(hipDeviceSynchronize());
    } else {
        std::vector<double> rep((size_t)P * (size_t)B);
        for (int p = 0; p < P; ++p)
            for (int b = 0; b < B; ++b) rep[(size_t)p * B + b] = eng->cir.nominal[(size_t)p];
        HIPCHK(hipMemcpy(dParams.p, rep.data(), sizeof(double) * rep.size(), hipMemcpyHostToDevice));
    }
    return CSIM_OK;
}

int csim_dc_batch(csim_engine* eng, const double* params, int32_t B, double* x_out,
                  int32_t* nr_iters, uint32_t* status)
{
    if (!eng || B < 0) { setError("csim_dc_batch: bad argument"); return CSIM_ERR_ARG; }
    if (B == 0) return CSIM_OK;
    HIPCHK(hipSetDevice(eng->device));
    const int N = eng->plan.N;
    DevBuf dParams, dX, dXt, dIt, dSt;
    int rc = stageParams(eng, params, B, dParams);
    if (rc) return rc;
    HIPCHK(dX.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dXt.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dIt.alloc(sizeof(int32_t) * (size_t)B));
    HIPCHK(dSt.alloc(sizeof(uint32_t) * (size_t)B));
    rc = csim_dc_batch_dev(eng, dParams.as<double>(), B, dX.as<double>(), dIt.as<int32_t>(), dSt.as<uint32_t>(), nullptr);
    if (rc) return rc;
    HIPCHK(csim::launchTranspose(dX.as<double>(), dXt.as<double>(), N, B, nullptr));            // [N][B] -> [B][N]
    HIPCHK(hipDeviceSynchronize());
    if (x_out)    HIPCHK(hipMemcpy(x_out, dXt.p, sizeof(double) * (size_t)N * B, hipMemcpyDeviceToHost));
    if (nr_iters) HIPCHK(hipMemcpy(nr_iters, dIt.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost));
    if (status)   HIPCHK(hipMemcpy(status, dSt.p, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

int csim_tran_batch(csim_engine* eng, const double* params, int32_t B, double tstep, double tstop,
                    double tstart, const int32_t* probe_eq, int32_t n_probe, int32_t out_stride,
                    double* wave_out, double* x_final, int64_t* nr_iters, uint32_t* status)
{
    if (!eng || B < 0) { setError("csim_tran_batch: bad argument"); return CSIM_ERR_ARG; }
    if (!(tstep > 0.0) || !(tstop > 0.0)) {                     // tanalisis.cpp:94-97
        setError("Invalid .TRAN card: tstep and tstop must be > 0");
        return CSIM_ERR_CONFIG;
    }
    if (wave_out && (out_stride <= 0 || n_probe <= 0 || !probe_eq)) { setError("csim_tran_batch: bad probe arguments"); return CSIM_ERR_ARG; }
    if (B == 0) return CSIM_OK;
    HIPCHK(hipSetDevice(eng->device));
    const int N = eng->plan.N;
    const int64_t nSteps = csim_tran_num_steps(tstep, tstop);
    const int64_t rowsAll = wave_out ? nSteps / out_stride + 1 : 0;

    DevBuf dParams, dX, dXt, dIt32, dIt, dSt, dChunk[2], dChunkT[2];
    int rc = stageParams(eng, params, B, dParams);
    if (rc) return rc;
    HIPCHK(dX.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dXt.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dIt32.alloc(sizeof(int32_t) * (size_t)B));
    HIPCHK(dIt.alloc(sizeof(int64_t) * (size_t)B));
    HIPCHK(dSt.alloc(sizeof(uint32_t) * (size_t)B));
    HIPCHK(hipMemset(dIt.p, 0, sizeof(int64_t) * (size_t)B));

    // Waveforms leave the device in chunks of whole output rows: the kernel writes a chunk
    // [rows][probe][B], it is transposed on the device to the caller's [B][rows][probe] and copied out
    // with a strided 2-D copy on a second stream while the next chunk is being computed (two buffers).
    // Device memory for waveforms is bounded by the chunk, not by the run.
    const int64_t chunkSteps = wave_out ? (out_stride >= 4096 ? (int64_t)out_stride : (4096 / out_stride) * (int64_t)out_stride) : 4096;
    const int64_t rowsPerChunk = wave_out ? chunkSteps / out_stride + 1 : 0;          // +1: the t = 0 row of the first chunk
    int64_t firstKept = 0;                                                            // rows with t < tstart are dropped
    for (int64_t r = 0; wave_out && r < rowsAll; ++r) {
        const double t = (r == 0) ? 0.0 : static_cast<double>(static_cast<int>(r * out_stride)) * tstep;
        if (t < tstart) firstKept = r + 1; else break;
    }
    const int64_t keep = rowsAll - firstKept;
    const size_t chunkElems = (size_t)rowsPerChunk * (size_t)(wave_out ? n_probe : 0) * (size_t)B;
    hipStream_t computeStream = nullptr, copyStream = nullptr;
    hipEvent_t computed[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    struct StreamGuard {
        hipStream_t *a, *b; hipEvent_t* e1; hipEvent_t* e2;
        ~StreamGuard() {
            for (int i = 0; i < 2; ++i) { if (e1[i]) (void)hipEventDestroy(e1[i]); if (e2[i]) (void)hipEventDestroy(e2[i]); }
            if (*a) (void)hipStreamDestroy(*a);
            if (*b) (void)hipStreamDestroy(*b);
        }
    } guard{&computeStream, &copyStream, computed, copied};
    if (wave_out) {
        HIPCHK(hipStreamCreateWithFlags(&computeStream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&copyStream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(dChunk[i].alloc(sizeof(double) * chunkElems));
            HIPCHK(dChunkT[i].alloc(sizeof(double) * chunkElems));
            HIPCHK(hipEventCreateWithFlags(&computed[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
        }
    }

    // CSIM_AUTO_JIT=1: a circuit without a prebuilt kernel is specialised on first use (what bench.py does
    // explicitly), so that callers of the reference-shaped API get the fast path without new code.  Worth it
    // for long runs only (the compile takes seconds); a failure leaves the general kernel in place.
    if (!eng->schedLaunch && eng->kernelChoice != 1 && nSteps >= 1000 && eng->cfg.autoJit) {
        if (csim_engine_jit_scheduled(eng, dParams.as<double>(), B, tstep, nSteps < 200 ? nSteps : 200) != CSIM_OK)
            std::fprintf(stderr, "csim: CSIM_AUTO_JIT: %s -- staying on the general kernel\n", csim_last_error());
    }
    HIPCHK(hipDeviceSynchronize());                       // uploads and planner work done before the streams start
    // t = 0 state: the DC operating point (tanalisis.cpp:112); its status bits stay in dSt
    rc = csim_dc_batch_dev(eng, dParams.as<double>(), B, dX.as<double>(), dIt32.as<int32_t>(), dSt.as<uint32_t>(), computeStream);
    if (rc) return rc;
    // time stepping in bounded launches (state carried in dX)
    int c = 0;
    for (int64_t s0 = 0; s0 < nSteps || s0 == 0; s0 += chunkSteps, ++c) {
        const int64_t n = (nSteps - s0) < chunkSteps ? (nSteps - s0) : chunkSteps;
        const int buf = c & 1;
        double* dW = nullptr;
        // global output rows of this chunk: (s0/stride, (s0+n)/stride], plus row 0 in the first chunk
        const int64_t rLo = (s0 == 0) ? 0 : s0 / out_stride + 1;
        const int64_t rHi = wave_out ? (s0 + n) / out_stride : -1;               // inclusive
        const int64_t rowsC = wave_out ? rHi - rLo + 1 : 0;
        if (wave_out && rowsC > 0) {
            if (c >= 2) HIPCHK(hipStreamWaitEvent(computeStream, copied[buf], 0));  // buffer free again?
            HIPCHK(hipMemsetAsync(dChunk[buf].p, 0, sizeof(double) * (size_t)rowsC * n_probe * B, computeStream));
            // the kernels index rows globally: shift the base so that row rLo lands on the chunk's first row
            dW = dChunk[buf].as<double>() - (ptrdiff_t)rLo * n_probe * B;
        }
        rc = csim_tran_batch_dev(eng, dParams.as<double>(), B, tstep, s0, n, probe_eq, n_probe, out_stride,
                                 dW, dX.as<double>(), dIt.as<int64_t>(), dSt.as<uint32_t>(), nullptr, computeStream);
        if (rc) return rc;
        if (wave_out && rowsC > 0) {
            HIPCHK(csim::launchTranspose(dChunk[buf].as<double>(), dChunkT[buf].as<double>(), (int)(rowsC * n_probe), B, computeStream));
            HIPCHK(hipEventRecord(computed[buf], computeStream));
            const int64_t skip = firstKept > rLo ? (firstKept - rLo < rowsC ? firstKept - rLo : rowsC) : 0;   // leading rows before tstart
            if (rowsC - skip > 0) {
                HIPCHK(hipStreamWaitEvent(copyStream, computed[buf], 0));
                HIPCHK(hipMemcpy2DAsync(wave_out + (size_t)(rLo + skip - firstKept) * n_probe, sizeof(double) * (size_t)keep * n_probe,
                                        dChunkT[buf].as<double>() + (size_t)skip * n_probe, sizeof(double) * (size_t)rowsC * n_probe,
                                        sizeof(double) * (size_t)(rowsC - skip) * n_probe, (size_t)B, hipMemcpyDeviceToHost, copyStream));
            }
            HIPCHK(hipEventRecord(copied[buf], copyStream));
        }
        if (nSteps == 0) break;
    }
    HIPCHK(csim::launchTranspose(dX.as<double>(), dXt.as<double>(), N, B, computeStream));
    HIPCHK(hipDeviceSynchronize());
    if (x_final)  HIPCHK(hipMemcpy(x_final, dXt.p, sizeof(double) * (size_t)N * B, hipMemcpyDeviceToHost));
    if (nr_iters) HIPCHK(hipMemcpy(nr_iters, dIt.p, sizeof(int64_t) * (size_t)B, hipMemcpyDeviceToHost));
    if (status)   HIPCHK(hipMemcpy(status, dSt.p, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

// one instance of a batch description as the reference's CSV (src/tanalisis.cpp:189-231)
int csim_tran_write_csv(csim_engine* eng, const double* params, int32_t B, int32_t instance, double tstep, double tstop,
                        double tstart, const int32_t* probe_eq, int32_t n_probe, const char* path)
{
    if (!eng || !path || B <= 0 || instance < 0 || instance >= B || n_probe < 0 || (n_probe > 0 && !probe_eq)) {
        setError("csim_tran_write_csv: bad argument");
        return CSIM_ERR_ARG;
    }
    const csim_ir* ir = eng->cir.view();
    const int N = ir->n_unknowns, P = ir->n_params;
    std::vector<int32_t> cols;
    if (n_probe > 0) cols.assign(probe_eq, probe_eq + n_probe);
    else if (!eng->netlistProbes.empty()) cols.assign(eng->netlistProbes.begin(), eng->netlistProbes.end());
    else for (int i = 0; i < N; ++i) cols.push_back(i);
    for (int32_t c : cols)
        if (c < 0 || c >= N) { setError("csim_tran_write_csv: probe equation index out of range"); return CSIM_ERR_ARG; }
    const int64_t rows = csim_tran_num_rows(tstep, tstop, tstart, 1);
    if (rows < 0) { setError("Invalid .TRAN card: tstep and tstop must be > 0"); return CSIM_ERR_CONFIG; }
    const int np = static_cast<int>(cols.size());
    std::vector<double> wave(static_cast<std::size_t>(rows) * static_cast<std::size_t>(np), 0.0), xf(static_cast<std::size_t>(N), 0.0);
    int64_t iters = 0;
    uint32_t status = 0;
    const int rc = csim_tran_batch(eng, params ? params + static_cast<std::size_t>(instance) * static_cast<std::size_t>(P) : nullptr, 1,
                                   tstep, tstop, tstart, cols.data(), np, 1, wave.data(), xf.data(), &iters, &status);
    if (rc != CSIM_OK) return rc;
    if (status & CSIM_ST_TRAN_NONFINITE) { setError("Transient: LU produced NaN/Inf."); return CSIM_ERR_UNSUPPORTED; }   // tanalisis.cpp:361 throws
    FILE* f = std::fopen(path, "w");
    if (!f) { setError(std::string("Cannot open transient output file '") + path + "'."); return CSIM_ERR_IO; }
    std::fputs("time", f);
    for (int32_t c : cols)
        std::fprintf(f, ",%s(%s)", c < ir->n_node_eq ? "V" : "I", eng->cir.eqNames[static_cast<std::size_t>(c)].c_str());
    std::fputc('\n', f);
    const int64_t nSteps = csim_tran_num_steps(tstep, tstop);
    const int64_t first = nSteps + 1 - rows;                 // rows with t < tstart are suppressed (tanalisis.cpp:208-209)
    for (int64_t k = 0; k < rows; ++k) {
        const int64_t row = first + k;
        const double t = row == 0 ? 0.0 : static_cast<double>(static_cast<int>(row)) * tstep;
        std::fprintf(f, "%.9e", t);
        for (int i = 0; i < np; ++i) std::fprintf(f, ",%.9e", wave[static_cast<std::size_t>(k) * static_cast<std::size_t>(np) + static_cast<std::size_t>(i)]);
        std::fputc('\n', f);
    }
    if (std::fclose(f) != 0) { setError(std::string("write error on '") + path + "'"); return CSIM_ERR_IO; }
    return CSIM_OK;
}

// ---- Gauss-Seidel variant (reference: include/solver.hpp:139-204, src/dcanalysis.cpp:71-92,166-237)

int csim_dc_gs_batch_dev(csim_engine* eng, const double* d_params, int32_t B, double* d_x,
                         int32_t* d_iters, uint32_t* d_status, void* stream)
{
    if (!eng || !d_params || !d_x || !d_iters || !d_status || B < 0) { setError("csim_dc_gs_batch_dev: bad argument"); return CSIM_ERR_ARG; }
    if (B == 0) return CSIM_OK;
    if (eng->big) { setError("the Gauss-Seidel DC kernel covers circuits of up to 63 unknowns"); return CSIM_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(eng->device));
    if (!eng->dGsRowPtr) {
        const int N = eng->plan.N;
        std::vector<int32_t> ptr(1, 0), col;
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j)
                if (j != i && eng->plan.patDc[(size_t)i * N + j]) col.push_back(j);
            ptr.push_back((int32_t)col.size());
        }
        int rc = upload(eng, ptr, &eng->dGsRowPtr);
        if (!rc) rc = upload(eng, col, &eng->dGsRowCol);
        if (rc) return rc;
    }
    HIPCHK(csim::launchDcGs(eng->gpDc, eng->dGsRowPtr, eng->dGsRowCol, d_params, B, d_x, d_iters, d_status,
                            static_cast<hipStream_t>(stream)));
    return CSIM_OK;
}

int csim_dc_gs_batch(csim_engine* eng, const double* params, int32_t B, double* x_out,
                     int32_t* nr_iters, uint32_t* status)
{
    if (!eng || B < 0) { setError("csim_dc_gs_batch: bad argument"); return CSIM_ERR_ARG; }
    if (B == 0) return CSIM_OK;
    HIPCHK(hipSetDevice(eng->device));
    const int N = eng->plan.N;
    DevBuf dParams, dX, dXt, dIt, dSt;
    int rc = stageParams(eng, params, B, dParams);
    if (rc) return rc;
    HIPCHK(dX.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dXt.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dIt.alloc(sizeof(int32_t) * (size_t)B));
    HIPCHK(dSt.alloc(sizeof(uint32_t) * (size_t)B));
    rc = csim_dc_gs_batch_dev(eng, dParams.as<double>(), B, dX.as<double>(), dIt.as<int32_t>(), dSt.as<uint32_t>(), nullptr);
    if (rc) return rc;
    HIPCHK(csim::launchTranspose(dX.as<double>(), dXt.as<double>(), N, B, nullptr));
    HIPCHK(hipDeviceSynchronize());
    if (x_out)    HIPCHK(hipMemcpy(x_out, dXt.p, sizeof(double) * (size_t)N * B, hipMemcpyDeviceToHost));
    if (nr_iters) HIPCHK(hipMemcpy(nr_iters, dIt.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost));
    if (status)   HIPCHK(hipMemcpy(status, dSt.p, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

int csim_gs_solve_batch(int32_t device, int32_t n, int32_t B, const double* A, const double* b, const double* x0,
                        int32_t max_iters, double tol, double* x, int32_t* sweeps)
{
    if (n < 0 || B < 0 || !x || (n > 0 && B > 0 && (!A || !b))) { setError("csim_gs_solve_batch: bad argument"); return CSIM_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        setError("csim_gs_solve_batch: no usable HIP device (this library has no CPU path)");
        return CSIM_ERR_NO_DEVICE;
    }
    if (n == 0 || B == 0) return CSIM_OK;                 // solver.hpp:146: empty system -> x0 unchanged (empty)
    if (n > 4096) { setError("csim_gs_solve_batch covers n <= 4096"); return CSIM_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(device));
    // instance-major in, slot-major on the device (lanes = systems read consecutive doubles)
    DevBuf dA, dAt, dB, dBt, dX0, dX0t, dX, dXt, dXo, dSw;
    const size_t nn = (size_t)n * n;
    HIPCHK(dA.alloc(sizeof(double) * nn * B));
    HIPCHK(dAt.alloc(sizeof(double) * nn * B));
    HIPCHK(dB.alloc(sizeof(double) * (size_t)n * B));
    HIPCHK(dBt.alloc(sizeof(double) * (size_t)n * B));
    HIPCHK(dX.alloc(sizeof(double) * (size_t)n * B));
    HIPCHK(dXt.alloc(sizeof(double) * (size_t)n * B));
    HIPCHK(dXo.alloc(sizeof(double) * (size_t)n * B));
    HIPCHK(dSw.alloc(sizeof(int32_t) * (size_t)B));
    HIPCHK(hipMemcpy(dA.p, A, sizeof(double) * nn * B, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB.p, b, sizeof(double) * (size_t)n * B, hipMemcpyHostToDevice));
    HIPCHK(csim::launchTranspose(dA.as<double>(), dAt.as<double>(), B, (int)nn, nullptr));      // [B][n*n] -> [n*n][B]
    HIPCHK(csim::launchTranspose(dB.as<double>(), dBt.as<double>(), B, n, nullptr));
    if (x0) {
        HIPCHK(dX0.alloc(sizeof(double) * (size_t)n * B));
        HIPCHK(dX0t.alloc(sizeof(double) * (size_t)n * B));
        HIPCHK(hipMemcpy(dX0.p, x0, sizeof(double) * (size_t)n * B, hipMemcpyHostToDevice));
        HIPCHK(csim::launchTranspose(dX0.as<double>(), dX0t.as<double>(), B, n, nullptr));
    }
    HIPCHK(csim::launchGsSolve(n, B, dAt.as<double>(), dBt.as<double>(), x0 ? dX0t.as<double>() : nullptr, max_iters, tol,
                               dXt.as<double>(), dXo.as<double>(), dSw.as<int32_t>(), nullptr));
    HIPCHK(csim::launchTranspose(dXt.as<double>(), dX.as<double>(), n, B, nullptr));            // [n][B] -> [B][n]
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(x, dX.p, sizeof(double) * (size_t)n * B, hipMemcpyDeviceToHost));
    if (sweeps) HIPCHK(hipMemcpy(sweeps, dSw.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

int csim_lu_solve_batch(int32_t device, int32_t n, int32_t B, const double* A, const double* b,
                        double* x, uint32_t* flags)
{
    if (n < 0 || B < 0 || !x || (n > 0 && B > 0 && (!A || !b))) { setError("csim_lu_solve_batch: bad argument"); return CSIM_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        setError("csim_lu_solve_batch: no usable HIP device (this library has no CPU path)");
        return CSIM_ERR_NO_DEVICE;
    }
    if (n == 0 || B == 0) return CSIM_OK;              // solver.hpp:86: empty system -> empty vector
    if (n > 1024) { setError("csim_lu_solve_batch covers n <= 1024"); return CSIM_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(device));
    DevBuf dA, dB, dX, dF;
    HIPCHK(dA.alloc(sizeof(double) * (size_t)n * n * B));
    HIPCHK(dB.alloc(sizeof(double) * (size_t)n * B));
    HIPCHK(dX.alloc(sizeof(double) * (size_t)n * B));
    HIPCHK(dF.alloc(sizeof(uint32_t) * (size_t)B));
    HIPCHK(hipMemcpy(dA.p, A, sizeof(double) * (size_t)n * n * B, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dB.p, b, sizeof(double) * (size_t)n * B, hipMemcpyHostToDevice));
    if (n > 63)      // dense systems beyond the LDS-resident kernel: in place in global memory (kernels_dense.hip)
        HIPCHK(csim::launchLuSolveDense(n, B, dA.as<double>(), dB.as<double>(), dX.as<double>(), dF.as<uint32_t>(), 1e-15, nullptr));
    else
        HIPCHK(csim::launchLuSolve(n, B, dA.as<double>(), dB.as<double>(), dX.as<double>(), dF.as<uint32_t>(), 1e-15, nullptr));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(x, dX.p, sizeof(double) * (size_t)n * B, hipMemcpyDeviceToHost));
    if (flags) HIPCHK(hipMemcpy(flags, dF.p, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

// planner log (device_common.hpp PIVLOG_*) -> the distinct sequences, most frequent first
static void decodePivotLog(const std::vector<int32_t>& log, int N, int32_t max_alts, int32_t* pivot_pos,
                           int64_t* counts, int32_t* n_alts, int64_t* n_other)
{
    std::vector<int> order;
    for (int s2 = 0; s2 < log[0]; ++s2) order.push_back(s2);
    auto cnt = [&](int s2) { return log[(size_t)(3 + s2 * (N + 1) + N)]; };
    for (size_t i = 0; i < order.size(); ++i)
        for (size_t j = i + 1; j < order.size(); ++j)
            if (cnt(order[j]) > cnt(order[i])) std::swap(order[i], order[j]);
    int64_t other = log[2];
    int out = 0;
    for (int s2 : order) {
        if (out < max_alts) {
            for (int k = 0; k < N; ++k) pivot_pos[(size_t)out * N + k] = log[(size_t)(3 + s2 * (N + 1) + k)];
            if (counts) counts[out] = cnt(s2);
            ++out;
        } else other += cnt(s2);
    }
    *n_alts = out;
    if (n_other) *n_other = other;
}

static int recordPivotSchedulesImpl(csim_engine* eng, const double* d_params, int32_t B, int32_t instance,
                                    double tstep, int64_t n_steps, int32_t max_alts, int32_t* pivot_pos,
                                    int64_t* counts, int32_t* n_alts, int64_t* n_other, uint32_t* inst_status);

int csim_record_pivot_schedules(csim_engine* eng, const double* d_params, int32_t B, int32_t instance,
                                double tstep, int64_t n_steps, int32_t max_alts, int32_t* pivot_pos,
                                int64_t* counts, int32_t* n_alts, int64_t* n_other)
{
    return recordPivotSchedulesImpl(eng, d_params, B, instance, tstep, n_steps, max_alts, pivot_pos, counts, n_alts,
                                    n_other, nullptr);
}

// inst_status (optional): status word of the planned instance after DC + the planned steps
static int recordPivotSchedulesImpl(csim_engine* eng, const double* d_params, int32_t B, int32_t instance,
                                    double tstep, int64_t n_steps, int32_t max_alts, int32_t* pivot_pos,
                                    int64_t* counts, int32_t* n_alts, int64_t* n_other, uint32_t* inst_status)
{
    if (!eng || !d_params || !pivot_pos || !n_alts || B <= 0 || instance < 0 || instance >= B || n_steps < 0 ||
        !(tstep > 0.0) || max_alts <= 0) {
        setError("csim_record_pivot_schedules: bad argument");
        return CSIM_ERR_ARG;
    }
    HIPCHK(hipSetDevice(eng->device));
    const int N = eng->plan.N;
    const int logInts = csim::pivlog_ints(N);
    DevBuf dX, dIt32, dIt, dSt, dLog;
    HIPCHK(dX.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dIt32.alloc(sizeof(int32_t) * (size_t)B));
    HIPCHK(dIt.alloc(sizeof(int64_t) * (size_t)B));
    HIPCHK(dSt.alloc(sizeof(uint32_t) * (size_t)B));
    HIPCHK(dLog.alloc(sizeof(int32_t) * (size_t)logInts));
    HIPCHK(hipMemset(dIt.p, 0, sizeof(int64_t) * (size_t)B));
    HIPCHK(hipMemset(dLog.p, 0, sizeof(int32_t) * (size_t)logInts));
    int rc = csim_dc_batch_dev(eng, d_params, B, dX.as<double>(), dIt32.as<int32_t>(), dSt.as<uint32_t>(), nullptr);
    if (rc) return rc;
    // only the chosen instance runs (mask), with the pivot log attached
    DevBuf dOnly;
    HIPCHK(dOnly.alloc((size_t)B));
    HIPCHK(hipMemset(dOnly.p, 0, (size_t)B));
    const unsigned char one = 1;
    HIPCHK(hipMemcpy(dOnly.as<unsigned char>() + instance, &one, 1, hipMemcpyHostToDevice));
    if (eng->big) {
        rc = ensureBigScratch(eng, B, nullptr);
        if (rc) return rc;
        HIPCHK(csim::launchTranBig(eng->gpTran, d_params, B, tstep, 0, n_steps, nullptr, 0, 1, nullptr, dX.as<double>(),
                                   dIt.as<long long>(), dSt.as<uint32_t>(), nullptr, dOnly.as<uint8_t>(),
                                   eng->dBigScratch, nullptr, nullptr, dLog.as<int32_t>(), instance));
    } else {
        HIPCHK(csim::launchTranGeneral(eng->gpTran, d_params, B, tstep, 0, n_steps, nullptr, 0, 1, nullptr, dX.as<double>(),
                                       dIt.as<long long>(), dSt.as<uint32_t>(), nullptr, dOnly.as<uint8_t>(), nullptr,
                                       dLog.as<int32_t>(), instance));
    }
    HIPCHK(hipDeviceSynchronize());
    std::vector<int32_t> log((size_t)logInts);
    HIPCHK(hipMemcpy(log.data(), dLog.p, sizeof(int32_t) * log.size(), hipMemcpyDeviceToHost));
    if (inst_status) HIPCHK(hipMemcpy(inst_status, dSt.as<uint32_t>() + instance, sizeof(uint32_t), hipMemcpyDeviceToHost));
    decodePivotLog(log, N, max_alts, pivot_pos, counts, n_alts, n_other);
    return CSIM_OK;
}

int csim_record_dc_pivot_schedules(csim_engine* eng, const double* d_params, int32_t B, int32_t instance,
                                   int32_t max_alts, int32_t* pivot_pos, int64_t* counts, int32_t* n_alts,
                                   int64_t* n_other)
{
    if (!eng || !d_params || !pivot_pos || !n_alts || B <= 0 || instance < 0 || instance >= B || max_alts <= 0) {
        setError("csim_record_dc_pivot_schedules: bad argument");
        return CSIM_ERR_ARG;
    }
    HIPCHK(hipSetDevice(eng->device));
    const int N = eng->plan.N;
    const int logInts = csim::pivlog_ints(N);
    DevBuf dX, dIt32, dSt, dLog, dOnly;
    HIPCHK(dX.alloc(sizeof(double) * (size_t)N * B));
    HIPCHK(dIt32.alloc(sizeof(int32_t) * (size_t)B));
    HIPCHK(dSt.alloc(sizeof(uint32_t) * (size_t)B));
    HIPCHK(dLog.alloc(sizeof(int32_t) * (size_t)logInts));
    HIPCHK(dOnly.alloc((size_t)B));
    HIPCHK(hipMemset(dLog.p, 0, sizeof(int32_t) * (size_t)logInts));
    HIPCHK(hipMemset(dOnly.p, 0, (size_t)B));
    const unsigned char one = 1;
    HIPCHK(hipMemcpy(dOnly.as<unsigned char>() + instance, &one, 1, hipMemcpyHostToDevice));
    if (eng->big) {
        const int rc = ensureBigScratch(eng, B, nullptr);
        if (rc) return rc;
        HIPCHK(csim::launchDcBig(eng->gpDc, d_params, B, eng->dBigScratch, dX.as<double>(), dIt32.as<int32_t>(), dSt.as<uint32_t>(),
                                 nullptr, dOnly.as<uint8_t>(), dLog.as<int32_t>(), instance));
    } else {
        HIPCHK(csim::launchDcGeneral(eng->gpDc, d_params, B, dX.as<double>(), dIt32.as<int32_t>(), dSt.as<uint32_t>(), nullptr,
                                     dOnly.as<uint8_t>(), dLog.as<int32_t>(), instance));
    }
    HIPCHK(hipDeviceSynchronize());
    std::vector<int32_t> log((size_t)logInts);
    HIPCHK(hipMemcpy(log.data(), dLog.p, sizeof(int32_t) * log.size(), hipMemcpyDeviceToHost));
    decodePivotLog(log, N, max_alts, pivot_pos, counts, n_alts, n_other);
    return CSIM_OK;
}

int csim_record_pivot_schedule(csim_engine* eng, const double* d_params, int32_t B, int32_t instance,
                               double tstep, int64_t n_steps, int32_t* pivot_pos, int64_t* n_factorizations,
                               int64_t* n_differ)
{
    if (!eng || !pivot_pos) { setError("csim_record_pivot_schedule: bad argument"); return CSIM_ERR_ARG; }
    const int N = eng->plan.N;
    std::vector<int32_t> pos((size_t)8 * N);
    int64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t nAlts = 0;
    int64_t other = 0;
    const int rc = csim_record_pivot_schedules(eng, d_params, B, instance, tstep, n_steps, 8, pos.data(), counts, &nAlts, &other);
    if (rc) return rc;
    for (int k = 0; k < N; ++k) pivot_pos[k] = nAlts > 0 ? pos[(size_t)k] : k;
    int64_t total = other;
    for (int a = 0; a < nAlts; ++a) total += counts[a];
    if (n_factorizations) *n_factorizations = total;
    if (n_differ) *n_differ = total - (nAlts > 0 ? counts[0] : 0);
    return CSIM_OK;
}

// generate + compile + load the kernels of `sch` for this engine's circuit (cached by hash)
static int buildAndLoadScheduled(csim_engine* eng, const csim::ScheduleSet& sch)
{
    const csim_ir* ir = eng->cir.view();
    const int N = ir->n_unknowns;
    const unsigned long long topo = csim::scheduleHash(*ir, csim::PivotSchedule::identity(N));
    // generator options of this engine's JIT (option jit_gen_opts: "key=value,key=value"; part of the library's hash)
    csim::GeneratorOptions gopt;
    for (std::size_t i = 0; i < eng->cfg.jitGenOpts.size();) {
        std::size_t e = eng->cfg.jitGenOpts.find(',', i);
        if (e == std::string::npos) e = eng->cfg.jitGenOpts.size();
        // "sweep" takes a comma list itself and is a tuning aid of csim_codegen only
        const std::string kv = eng->cfg.jitGenOpts.substr(i, e - i);
        if (!kv.empty() && !gopt.set(kv)) { setError("jit_gen_opts: unknown generator option '" + kv + "'"); return CSIM_ERR_ARG; }
        i = e + 1;
    }
    const unsigned long long full = csim::scheduleHash(*ir, sch, gopt);
    const std::string dir = eng->cfg.jitDir;
    {
        const std::string why = csim::jitPrepareDir(dir);
        if (!why.empty()) { setError("JIT cache: " + why); return CSIM_ERR_IO; }
    }
    char stem[96];
    std::snprintf(stem, sizeof stem, "/libcsim_sched_%016llx_%016llx", topo, full);
    const std::string lib = dir + stem + ".so", hip = dir + stem + ".hip", log = dir + stem + ".log";

    typedef unsigned long long (*HashFn)(void);
    typedef const char* (*InfoFn)(void);
    // a library is loaded only if it is a regular file of the calling user inside the private directory,
    // and kept only if it reports the hash of exactly this (circuit, schedules, generator options, revision)
    auto openChecked = [&](const std::string& path) -> void* {
        if (!csim::jitFileTrusted(path)) return nullptr;
        void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) return nullptr;
        HashFn topoFn = reinterpret_cast<HashFn>(dlsym(h, "csim_sched_topology"));
        HashFn fullFn = reinterpret_cast<HashFn>(dlsym(h, "csim_sched_hash"));
        if (!topoFn || !fullFn || topoFn() != topo || fullFn() != full || !dlsym(h, "csim_sched_launch")) { dlclose(h); return nullptr; }
        return h;
    };
    void* handle = openChecked(lib);
    if (!handle) {
        const std::string src = csim::generateTranKernelSource(*ir, eng->plan, sch, "jit", nullptr, gopt);
        if (src.empty()) { setError("circuit too large for a scheduled kernel (iterate does not fit LDS)"); return CSIM_ERR_UNSUPPORTED; }
        // several ranks may specialise the same circuit at once: private temporaries, atomic rename
        const std::string tag = "." + std::to_string((long long)getpid());
        const std::string hipTmp = hip + tag, libTmp = lib + tag, logTmp = log + tag;
        FILE* f = std::fopen(hipTmp.c_str(), "w");
        if (!f) { setError("cannot write " + hipTmp); return CSIM_ERR_IO; }
        std::fwrite(src.data(), 1, src.size(), f);
        std::fclose(f);
        const std::string why = csim::jitRun({eng->cfg.hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared",
                                              "-x", "hip", hipTmp, "-o", libTmp}, logTmp, eng->cfg.jitTimeoutSec);
        if (!why.empty()) { (void)std::remove(libTmp.c_str()); setError("JIT compile: " + why); return CSIM_ERR_UNSUPPORTED; }
        if (std::rename(libTmp.c_str(), lib.c_str()) != 0) { setError("cannot move " + libTmp + " into place"); return CSIM_ERR_IO; }
        (void)std::rename(hipTmp.c_str(), hip.c_str());
        handle = openChecked(lib);
        if (!handle) { setError("generated library " + lib + " could not be loaded or does not match the circuit"); return CSIM_ERR_IO; }
    }
    InfoFn infoFn = reinterpret_cast<InfoFn>(dlsym(handle, "csim_sched_info"));
    if (eng->schedLib) dlclose(eng->schedLib);
    eng->schedLib = handle;
    eng->schedLaunch = reinterpret_cast<csim_engine::SchedLaunchFn>(dlsym(handle, "csim_sched_launch"));
    eng->schedInfo = infoFn ? infoFn() : "";
    adoptScheduleTable(eng, handle);
    return CSIM_OK;
}

int csim_engine_jit_with_schedules(csim_engine* eng, const int32_t* pivot_pos, int32_t n_alts,
                                   const int32_t* dc_pivot_pos, int32_t n_dc_alts)
{
    if (!eng || !pivot_pos || n_alts <= 0 || n_alts > 16 || n_dc_alts < 0 || n_dc_alts > 8 || (n_dc_alts > 0 && !dc_pivot_pos)) {
        setError("csim_engine_jit_with_schedules: bad argument");
        return CSIM_ERR_ARG;
    }
    HIPCHK(hipSetDevice(eng->device));
    const int N = eng->cir.view()->n_unknowns;
    csim::ScheduleSet sch;
    auto take = [&](const int32_t* pos, int n, std::vector<csim::PivotSchedule>& dst) -> bool {
        for (int a = 0; a < n; ++a) {
            csim::PivotSchedule one = csim::PivotSchedule::identity(N);
            for (int k = 0; k < N; ++k) {
                const int p = pos[(size_t)a * N + k];
                if (p < k || p >= N) return false;
                one.pivotPos[(size_t)k] = p;
            }
            bool dup = false;
            for (const csim::PivotSchedule& o : dst) dup = dup || o.pivotPos == one.pivotPos;
            if (!dup) dst.push_back(one);
        }
        return true;
    };
    if (!take(pivot_pos, n_alts, sch.alts) || !take(dc_pivot_pos, n_dc_alts, sch.dcAlts)) {
        setError("csim_engine_jit_with_schedules: pivot position out of range (need k <= pos < N)");
        return CSIM_ERR_ARG;
    }
    return buildAndLoadScheduled(eng, sch);
}

int csim_engine_jit_scheduled(csim_engine* eng, const double* d_params, int32_t B, double tstep, int64_t plan_steps)
{
    if (!eng || !d_params || B <= 0 || !(tstep > 0.0) || plan_steps <= 0) { setError("csim_engine_jit_scheduled: bad argument"); return CSIM_ERR_ARG; }
    const csim_ir* ir = eng->cir.view();
    const int N = ir->n_unknowns;
    const int maxAlts = 4;
    // Plan on instance 0 (by convention the nominal circuit) and, for circuits small enough that a planning
    // run costs milliseconds, on three more instances spread over the batch: a switching circuit's Monte-Carlo
    // samples do not all walk through the same pivot sequences.  Sequences are merged by total use.
    std::vector<int32_t> planOn = {0};
    if (!eng->big)
        for (int32_t cand : {B / 3, (2 * B) / 3, B - 1})
            if (std::find(planOn.begin(), planOn.end(), cand) == planOn.end()) planOn.push_back(cand);
    std::vector<std::pair<std::vector<int32_t>, int64_t>> merged;     // (sequence, factorisations)
    int rc = CSIM_OK;
    for (int32_t inst : planOn) {
        const int planMax = 8;
        std::vector<int32_t> pos((size_t)planMax * N);
        int64_t counts[planMax] = {0, 0, 0, 0, 0, 0, 0, 0};
        int32_t n = 0;
        int64_t other = 0;
        uint32_t instStatus = 0;
        rc = recordPivotSchedulesImpl(eng, d_params, B, inst, tstep, plan_steps, planMax, pos.data(), counts, &n, &other,
                                      &instStatus);
        if (rc) return rc;
        // An instance whose Newton iterations do not converge (or that met a failed factorisation) is not a
        // pattern worth specialising for: its steps are redone by the general kernel anyway (the generated
        // kernel hands over every step that ends at the NR cap).  Instance 0 always counts.
        const uint32_t trouble = CSIM_ST_TRAN_NONFINITE | CSIM_ST_TRAN_NONCONV | CSIM_ST_LU_TINY_PIVOT |
                                 CSIM_ST_DC_NONCONV | CSIM_ST_DC_NONFINITE;
        if (inst != 0 && (instStatus & trouble)) continue;
        for (int a = 0; a < n; ++a) {
            std::vector<int32_t> seq(pos.begin() + (size_t)a * N, pos.begin() + (size_t)(a + 1) * N);
            auto hit = std::find_if(merged.begin(), merged.end(), [&](const auto& m) { return m.first == seq; });
            if (hit == merged.end()) merged.emplace_back(seq, counts[a]);
            else hit->second += counts[a];
        }
    }
    if (merged.empty()) { setError("planner saw no successful factorisation"); return CSIM_ERR_UNSUPPORTED; }
    std::stable_sort(merged.begin(), merged.end(), [](const auto& x, const auto& y) { return x.second > y.second; });
    csim::ScheduleSet sch;
    for (size_t a = 0; a < merged.size() && a < (size_t)maxAlts; ++a) {
        csim::PivotSchedule one = csim::PivotSchedule::identity(N);
        for (int k = 0; k < N; ++k) one.pivotPos[(size_t)k] = merged[a].first[(size_t)k];
        sch.alts.push_back(one);
    }
    // The DC operating point of Newton circuits gets its own schedules (planned on the same instance),
    // but only when a few sequences cover that instance's whole ramp: a circuit that walks through many
    // (buffer.sp: 10) would fail its checks in most instances and pay for both kernels.
    // cfg.jitDcAlts = limit (default 4, at most 8); cfg.jitDcForce keeps a partial cover (tests).
    if (!ir->has_nonlinear) {
        // A linear circuit's operating point is ONE factorisation (dcSolveDirectLU).  Its generated kernel carries one
        // pivot sequence, so it is kept only when four instances spread over the batch all choose the same one: a
        // resistor chain's DC matrix (the configs[3] ladder with its capacitors open) has pivots that win by a fraction
        // of a percent, every Monte-Carlo instance swaps differently, and nearly all of them would be sent on to the
        // general kernel after paying for the generated one.
        std::vector<int32_t> first;
        bool same = true;
        for (int32_t inst : {(int32_t)0, B / 3, (2 * B) / 3, B - 1}) {
            std::vector<int32_t> dpos((size_t)N);
            int32_t nDc = 0;
            int64_t dcOther = 0;
            rc = csim_record_dc_pivot_schedules(eng, d_params, B, inst, 1, dpos.data(), nullptr, &nDc, &dcOther);
            if (rc) return rc;
            if (nDc != 1) { same = false; break; }
            if (first.empty()) first = dpos;
            else if (first != dpos) { same = false; break; }
        }
        if (same && !first.empty()) {
            csim::PivotSchedule one = csim::PivotSchedule::identity(N);
            for (int k = 0; k < N; ++k) one.pivotPos[(size_t)k] = first[(size_t)k];
            sch.dcAlts.push_back(one);
        }
    } else if (!eng->big) {
        const int planMax = 8;
        const int limit = eng->cfg.jitDcAlts;
        const bool force = eng->cfg.jitDcForce;
        std::vector<int32_t> dpos((size_t)planMax * N);
        int32_t nDc = 0;
        int64_t dcOther = 0;
        rc = csim_record_dc_pivot_schedules(eng, d_params, B, 0, planMax, dpos.data(), nullptr, &nDc, &dcOther);
        if (rc) return rc;
        const bool covered = nDc > 0 && nDc <= limit && dcOther == 0;
        if (covered || (force && nDc > 0)) {
            for (int a = 0; a < nDc && a < std::max(limit, 1); ++a) {
                csim::PivotSchedule one = csim::PivotSchedule::identity(N);
                for (int k = 0; k < N; ++k) one.pivotPos[(size_t)k] = dpos[(size_t)a * N + k];
                sch.dcAlts.push_back(one);
            }
        }
    }
    return buildAndLoadScheduled(eng, sch);
}

int csim_lu_decompose_batch(int32_t device, int32_t n, int32_t B, const double* A, double* LU,
                            int32_t* perm, uint32_t* flags)
{
    if (n < 0 || B < 0 || (n > 0 && B > 0 && (!A || !LU || !perm))) { setError("csim_lu_decompose_batch: bad argument"); return CSIM_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        setError("csim_lu_decompose_batch: no usable HIP device (this library has no CPU path)");
        return CSIM_ERR_NO_DEVICE;
    }
    if (n == 0 || B == 0) return CSIM_OK;
    if (n > 1024) { setError("csim_lu_decompose_batch covers n <= 1024"); return CSIM_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(device));
    DevBuf dA, dLU, dP, dF;
    HIPCHK(dLU.alloc(sizeof(double) * (size_t)n * n * B));
    HIPCHK(dP.alloc(sizeof(int32_t) * (size_t)n * B));
    HIPCHK(dF.alloc(sizeof(uint32_t) * (size_t)B));
    if (n > 63) {    // in place on the copy (kernels_dense.hip)
        HIPCHK(hipMemcpy(dLU.p, A, sizeof(double) * (size_t)n * n * B, hipMemcpyHostToDevice));
        HIPCHK(csim::launchLuFactorDense(n, B, dLU.as<double>(), dP.as<int32_t>(), dF.as<uint32_t>(), 1e-15, nullptr));
    } else {
        HIPCHK(dA.alloc(sizeof(double) * (size_t)n * n * B));
        HIPCHK(hipMemcpy(dA.p, A, sizeof(double) * (size_t)n * n * B, hipMemcpyHostToDevice));
        HIPCHK(csim::launchLuFactor(n, B, dA.as<double>(), dLU.as<double>(), dP.as<int32_t>(), dF.as<uint32_t>(), 1e-15, nullptr));
    }
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(LU, dLU.p, sizeof(double) * (size_t)n * n * B, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(perm, dP.p, sizeof(int32_t) * (size_t)n * B, hipMemcpyDeviceToHost));
    if (flags) HIPCHK(hipMemcpy(flags, dF.p, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

} // extern "C"
