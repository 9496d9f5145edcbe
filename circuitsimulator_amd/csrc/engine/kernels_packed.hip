// kernels_packed.hip -- the general (dynamic-pivot, bit-faithful) kernels with FOUR instances per
// wavefront for circuits with N <= 32 unknowns (tests/buffer.sp: N = 13, tests/dbmixer.sp: N = 31).
//
// kernels_general.hip gives a whole wavefront to one instance and keeps the matrix in LDS: ~9 000
// instructions per NR iteration at N = 31, most of them the scalar walks over candidate rows and non-zero
// rows (a cross-lane round trip each), 42 us per iteration for a wave alone on its SIMD.  Here one DPP row
// of 16 lanes is one instance and the augmented matrix lives in REGISTERS: lane g keeps row g (and row
// 16 + g when N > 16: "slot" 1), column j is register j.  Rows are exchanged for real, so the pivot row of
// column k always sits in lane k % 16 of slot k / 16 and reaches the other lanes with a compile-time DPP
// broadcast; the pivot is max-reduced over the row of lanes and located with a ballot; every row below the
// pivot applies its own multiplier at once.  The size is a template parameter (every loop unrolls, every
// index is a register): N itself up to 16, N rounded up to an even number from 18 to 32 (a padding row and
// column of the identity that no real row ever meets).
//
// Same algorithm, same operation order per instance as the reference (include/solver.hpp:30-131,
// src/tanalisis.cpp:238-420, src/dcanalysis.cpp:46-68,95-163,268-307): results are bit-identical to the
// one-instance-per-wave kernels and to the oracle.  Rows whose multiplier is zero are updated like any other
// (solver.hpp:70-76 does; the LDS version skips them -- equal unless the pivot row holds an Inf or NaN).
//
// Data path of one Newton iteration: element terms T (LDS, per instance) -> every structural non-zero sums
// its terms in the reference's stamping order into Gs[n] (LDS, per instance, nnz doubles -- not a dense
// matrix: four dense copies would not leave room for one wave per SIMD at N = 31) -> lane g fills its row
// registers through rowMap[row][col] = n (LDS, one per workgroup; structural zeros point at a 0.0) -> LU and
// substitution in registers.  The plan's index arrays are staged in LDS once per launch (device_common.hpp).
//
// Control flow is wave-uniform: the groups of a wave may be at different time steps (hybrid stepping),
// converge after different numbers of passes, or stop -- each of those is a per-group flag, never a branch
// around a cross-lane operation.  The planner (pivot log) stays on the one-instance-per-wave kernels.
//
// Measured (MI355X, B = 4096): buffer.sp DC operating points 20.1 ms (wave per instance) -> 8.8 ms (packed,
// LDS matrix) -> 3.9 ms (registers).
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace csim {

#pragma clang fp contract(off)

#include "packed_lu.hpp"

namespace {

// every lane returns the same sum of squares in index order (the oracle's norm)
template <int S>
__device__ __forceinline__ double grp_norm(const double (&d)[S], double* sc, int N, int g)
{
#pragma unroll
    for (int s = 0; s < S; ++s)
        if (16 * s + g < N) sc[16 * s + g] = d[s] * d[s];
    wave_sync();
    double ss = 0.0;
    for (int i = 0; i < N; ++i) ss += sc[i];
    wave_sync();
    return sqrt(ss);
}
template <int S> __device__ __forceinline__ bool grp_all_finite(const double (&v)[S], int N, int g, int q)
{
    bool bad = false;
#pragma unroll
    for (int s = 0; s < S; ++s) bad = bad || (16 * s + g < N && !isfinite(v[s]));
    return grp_mask(bad, q) == 0u;
}

__device__ __forceinline__ bool grp_sequence_known(const int32_t* curPiv, const int32_t* alts, int nAlts, int N, int g, int q)
{
    wave_sync();
    bool known = false;
    for (int a = 0; a < nAlts; ++a) {
        bool same = true;
        for (int k = g; k < N; k += G16) same = same && (curPiv[k] == alts[a * N + k]);
        known = known || grp_mask(!same, q) == 0u;
    }
    return known;
}

// launch set-up shared by both kernels: the workgroup's rowMap and the staged plan
__device__ __forceinline__ GenPlan packed_setup(const GenPlan& plArg, double* smp, const PackedLayout& L, int lane, int32_t*& rowMap)
{
    rowMap = reinterpret_cast<int32_t*>(smp + (size_t)IPW * L.total);
    const int cells = plArg.N * plArg.LD;
    for (int i = lane; i < cells; i += 64) rowMap[i] = plArg.nnzG;      // structural zero: the 0.0 behind the gathered values
    wave_sync();
    for (int n = lane; n < plArg.nnzG; n += 64) rowMap[plArg.gPos[n]] = n;
    return plan_in_lds(plArg, rowMap + cells, lane, 64);
}

} // namespace

// ------------------------------------------------------------------ DC (K2g packed)
template <int S>
__global__ void __launch_bounds__(64)
k_dc_packed(GenPlan plArg, const double* __restrict__ params, int B, double* __restrict__ xout,
            int32_t* __restrict__ iters, uint32_t* __restrict__ status, const uint8_t* __restrict__ only)
{
    extern __shared__ double smp[];
    const int lane = threadIdx.x, g = lane % G16, q = lane / G16;
    const int bRaw = blockIdx.x * IPW + q;
    const bool exists = bRaw < B;
    const int b = exists ? bRaw : B - 1;
    const bool mine = exists && !(only && !only[b]);         // fallback launches touch the flagged instances only
    if (!__any(mine)) return;
    const int N = plArg.N, LD = plArg.LD;
    const PackedLayout L = packedLayout(plArg);
    int32_t* rowMap;
    const GenPlan pl = packed_setup(plArg, smp, L, lane, rowMap);
    double* base = smp + (size_t)q * L.total;
    double* T = base + L.T;
    double* Pv = base + L.P;
    double* xs = base + L.xs;
    double* Gs = base + L.Gs;
    double* Rs = base + L.Rs;
    double* sc = base + L.sc;
    const csim_consts& K = pl.k;

    for (int p = g; p < pl.P; p += G16) Pv[p] = params[(int64_t)p * B + b];
    for (int t = g; t < pl.nTerms; t += G16) T[t] = 0.0;
    for (int i = g; i < N; i += G16) { xs[i] = 0.0; Rs[i] = 0.0; }
    if (g == 0) Gs[pl.nnzG] = 0.0;
    wave_sync();
    terms_const<false>(pl, Pv, T, 0.0, g, G16);
    wave_sync();

    unsigned st = 0;
    int itTotal = 0;
    // a linear circuit is one solve at x = 0 with full sources and NO gmin (dcanalysis.cpp:46-68): one ramp
    // step of one pass through the same code
    const bool nonlinear = pl.hasNonlinear != 0;
    const int rampSteps = nonlinear ? K.dc_ramp_steps : 1, maxIters = nonlinear ? K.dc_max_iters : 1;
    for (int step = 1; step <= rampSteps; ++step) {
        const double scale = (double)step / rampSteps;
        double gmin = nonlinear ? base_gmin_p(K, scale) : 0.0;
        double prevErr = INFINITY;
        terms_step_dc(pl, Pv, T, scale, g, G16);
        wave_sync();
        bool active = mine;                               // this group still iterates in this ramp step
        for (int iter = 0; iter < maxIters; ++iter) {
            if (!__any(active)) break;
            if (nonlinear) terms_iter_mos(pl, Pv, T, xs, g, G16);
            if (g == 0) T[pl.termGmin] = gmin;
            wave_sync();
            gather_nonzeros(pl, T, Gs, Rs, g);
            wave_sync();
            double xr[S];
            lu_solve_dispatch(Gs, Rs, rowMap, pl.nnzG, N, LD, K.lu_eps, g, q, active, st, nullptr, xr);
            if (active) ++itTotal;
            if (!nonlinear) {
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (16 * s + g < N) xs[16 * s + g] = xr[s];
                break;
            }
            const bool finite = grp_all_finite<S>(xr, N, g, q);
            // ConvController::update, computed by every group, applied by the active ones
            const double alpha = clampd_p(K.dc_alpha, K.dc_alpha_min, K.dc_alpha_max);
            double xn[S], dx[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const double xo = (16 * s + g < N) ? xs[16 * s + g] : 0.0;
                xn[s] = xo + alpha * (xr[s] - xo);
                dx[s] = xn[s] - xo;
            }
            const double err = grp_norm<S>(dx, sc, N, g);
            const double gb = base_gmin_p(K, scale);
            double gnext = gb;
            if (iter == 0 || !isfinite(prevErr)) gnext = gb;
            else if (err > prevErr * K.slow_ratio) gnext = fmin(gmin * 2.0, K.gmin_abs_max);
            else if (err < prevErr * K.fast_ratio) gnext = 0.5 * gmin + 0.5 * gb;
            else gnext = 0.7 * gmin + 0.3 * gb;
            if (active && !finite) {                                            // :135-138: raise gmin, drop the pass
                gmin = fmin(gmin * K.gmin_nonfinite_mul, K.gmin_nonfinite_cap);
                st |= CSIM_ST_DC_NONFINITE;
            } else if (active) {
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (16 * s + g < N) xs[16 * s + g] = xn[s];
                gmin = gnext;
                prevErr = err;
                if (err < K.dc_tol) active = false;                             // :150
                else if (iter == K.dc_max_iters - 1) st |= CSIM_ST_DC_NONCONV;  // :153-158
            }
            wave_sync();
        }
    }
    wave_sync();
    if (mine) {
        for (int i = g; i < N; i += G16) xout[(int64_t)i * B + b] = xs[i];
        if (g == 0) { iters[b] = itTotal; status[b] = only ? (st | CSIM_ST_SCHED_FALLBACK_DC) : st; }
    }
}

// ------------------------------------------------------------ transient (K1g packed)
template <int S>
__global__ void __launch_bounds__(64)
k_tran_packed(GenPlan plArg, const double* __restrict__ params, int B, double dt,
              long long stepFirst, long long nSteps, const int32_t* __restrict__ probeEq, int nProbe, int outStride,
              double* __restrict__ wave, double* __restrict__ xio, long long* __restrict__ iters,
              uint32_t* __restrict__ status, int32_t* __restrict__ stepIters, const uint8_t* __restrict__ only,
              int32_t* __restrict__ done, int maxSteps, const int32_t* __restrict__ knownAlts, int nKnown)
{
    extern __shared__ double smp[];
    const int lane = threadIdx.x, g = lane % G16, q = lane / G16;
    const int bRaw = blockIdx.x * IPW + q;
    const bool exists = bRaw < B;
    const int b = exists ? bRaw : B - 1;
    const long long d0 = done ? (long long)done[b] : 0;
    const bool mine = exists && !(only && !only[b]) && !(done && d0 >= nSteps);
    if (!__any(mine)) return;
    const long long sEnd = done ? (d0 + maxSteps < nSteps ? d0 + maxSteps : nSteps) : nSteps;
    const int N = plArg.N, LD = plArg.LD;
    const PackedLayout L = packedLayout(plArg);
    int32_t* rowMap;
    const GenPlan pl = packed_setup(plArg, smp, L, lane, rowMap);
    double* base = smp + (size_t)q * L.total;
    double* T = base + L.T;
    double* Pv = base + L.P;
    double* xs = base + L.xs;
    double* xp = base + L.xp;
    double* Gs = base + L.Gs;
    double* Rs = base + L.Rs;
    double* sc = base + L.sc;
    int32_t* curPiv = (done && knownAlts) ? reinterpret_cast<int32_t*>(base + L.piv) : nullptr;
    const csim_consts& K = pl.k;
    const int slowIters = slowStepIters(K.tran_tol, K.tran_alpha, K.tran_max_iters);

    for (int p = g; p < pl.P; p += G16) Pv[p] = params[(int64_t)p * B + b];
    for (int t = g; t < pl.nTerms; t += G16) T[t] = 0.0;
    for (int i = g; i < N; i += G16) {
        const double v = xio[(int64_t)i * B + b];
        xs[i] = v;
        xp[i] = v;
        Rs[i] = 0.0;
    }
    if (g == 0) Gs[pl.nnzG] = 0.0;
    wave_sync();
    terms_const<true>(pl, Pv, T, dt, g, G16);
    if (g == 0) T[pl.termGmin] = K.tran_gmin;
    wave_sync();

    if (mine && stepFirst == 0 && d0 == 0 && wave)                         // t = 0 row (:250)
        for (int pq = g; pq < nProbe; pq += G16) wave[((int64_t)0 * nProbe + pq) * B + b] = xs[probeEq[pq]];

    unsigned st = (status[b] & CSIM_ST_TRAN_NONFINITE);
    if (done) st |= CSIM_ST_SCHED_FALLBACK;
    long long itTotal = 0;
    bool aborted = (st & CSIM_ST_TRAN_NONFINITE) != 0;                     // the reference would have thrown: stay stopped
    long long s = d0 + 1, sLast = d0;
    bool running = mine && !aborted && s <= sEnd;
    while (__any(running)) {
        if (running) sLast = s;
        bool stepKnown = curPiv != nullptr;
        bool stepConverged = false;
        const long long gstep = stepFirst + s;
        const double tNow = (double)(int)gstep * dt;                       // :256
        terms_step_tran(pl, Pv, T, xp, tNow, g, G16);
        wave_sync();
        int it = 0;
        bool active = running;
        for (int iter = 0; iter < K.tran_max_iters; ++iter) {
            if (!__any(active)) break;
            terms_iter_mos(pl, Pv, T, xs, g, G16);
            wave_sync();
            gather_nonzeros(pl, T, Gs, Rs, g);                             // :259-356
            wave_sync();
            double xr[S];
            lu_solve_dispatch(Gs, Rs, rowMap, pl.nnzG, N, LD, K.lu_eps, g, q, active, st, curPiv, xr);   // :359
            if (curPiv) {                                                  // uniform: every group compares, the active ones keep the answer
                const bool known = grp_sequence_known(curPiv, knownAlts, nKnown, N, g, q);
                if (active && stepKnown) stepKnown = known;
            }
            if (active) ++it;
            const bool finite = grp_all_finite<S>(xr, N, g, q);
            double xn[S], dx[S];
#pragma unroll
            for (int sl = 0; sl < S; ++sl) {
                const double xo = (16 * sl + g < N) ? xs[16 * sl + g] : 0.0;
                xn[sl] = xo + K.tran_alpha * (xr[sl] - xo);                // :365
                dx[sl] = xn[sl] - xo;
            }
            const double err = grp_norm<S>(dx, sc, N, g);                  // :366
            if (active && !finite) {                                       // :360-362
                st |= CSIM_ST_TRAN_NONFINITE;
                aborted = true;
                active = false;
            } else if (active) {
#pragma unroll
                for (int sl = 0; sl < S; ++sl)
                    if (16 * sl + g < N) xs[16 * sl + g] = xn[sl];         // :367
                if (err < K.tran_tol) { stepConverged = it <= slowIters; active = false; }          // :369-371
                else if (iter == K.tran_max_iters - 1) st |= CSIM_ST_TRAN_NONCONV;                   // :372-376
            }
            wave_sync();
        }
        if (running) {
            itTotal += it;
            if (stepIters && g == 0) stepIters[(s - 1) * (int64_t)B + b] = it;
        }
        if (running && !aborted)
            for (int i = g; i < N; i += G16) xp[i] = xs[i];                // :381-417
        wave_sync();
        if (running && !aborted && wave && (gstep % outStride) == 0)       // :419
            for (int pq = g; pq < nProbe; pq += G16)
                wave[((gstep / outStride) * nProbe + pq) * (int64_t)B + b] = xs[probeEq[pq]];
        // hybrid stepping: back on a recorded schedule and converging -> hand the instance back
        if (running && (aborted || (stepKnown && stepConverged))) running = false;
        if (running) { ++s; running = s <= sEnd; }
    }
    if (mine) {
        for (int i = g; i < N; i += G16) xio[(int64_t)i * B + b] = xs[i];
        if (g == 0) {
            iters[b] += itTotal;
            status[b] |= st;
            if (done) done[b] = aborted ? (int32_t)nSteps : (int32_t)sLast;
        }
    }
}

// ------------------------------------------------------------------ launchers
// lanes per instance of the general kernels for this circuit: 16 = these kernels, 64 = a wave per instance
// (kernels_general.hip: N > 32, or four instances with the staged plan do not fit the LDS of a workgroup)
int packedLanesFor(const GenPlan& pl)
{
    if (pl.N > 32) return 64;
    return packedLdsBytes(pl) <= kStagedLdsLimit ? 16 : 64;
}

hipError_t launchDcPacked(const GenPlan& pl, const double* dParams, int B, double* dX, int32_t* dIters,
                          uint32_t* dStatus, hipStream_t stream, const uint8_t* dOnly)
{
    if (packedLanesFor(pl) != 16) return hipErrorInvalidValue;
    const size_t lds = packedLdsBytes(pl);
    const dim3 grid((B + IPW - 1) / IPW);
    if (pl.N <= 16) hipLaunchKernelGGL(k_dc_packed<1>, grid, dim3(64), lds, stream, pl, dParams, B, dX, dIters, dStatus, dOnly);
    else hipLaunchKernelGGL(k_dc_packed<2>, grid, dim3(64), lds, stream, pl, dParams, B, dX, dIters, dStatus, dOnly);
    return hipGetLastError();
}

hipError_t launchTranPacked(const GenPlan& pl, const double* dParams, int B, double dt, long long stepFirst,
                            long long nSteps, const int32_t* dProbeEq, int nProbe, int outStride, double* dWave,
                            double* dX, long long* dIters, uint32_t* dStatus, int32_t* dStepIters, const uint8_t* dOnly,
                            hipStream_t stream, int32_t* dDone, int maxSteps, const int32_t* dKnownAlts, int nKnown)
{
    if (packedLanesFor(pl) != 16) return hipErrorInvalidValue;
    const size_t lds = packedLdsBytes(pl);
    const dim3 grid((B + IPW - 1) / IPW);
    if (pl.N <= 16)
        hipLaunchKernelGGL(k_tran_packed<1>, grid, dim3(64), lds, stream, pl, dParams, B, dt, stepFirst, nSteps, dProbeEq,
                           nProbe, outStride, dWave, dX, dIters, dStatus, dStepIters, dOnly, dDone, maxSteps, dKnownAlts, nKnown);
    else
        hipLaunchKernelGGL(k_tran_packed<2>, grid, dim3(64), lds, stream, pl, dParams, B, dt, stepFirst, nSteps, dProbeEq,
                           nProbe, outStride, dWave, dX, dIters, dStatus, dStepIters, dOnly, dDone, maxSteps, dKnownAlts, nKnown);
    return hipGetLastError();
}

} // namespace csim
