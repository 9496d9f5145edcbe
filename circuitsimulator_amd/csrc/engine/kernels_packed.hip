// kernels_packed.hip -- the general (dynamic-pivot, bit-faithful) kernels with FOUR instances per
// wavefront (G = 16 lanes per instance) for circuits with N <= 15 unknowns (tests/buffer.sp: N = 13).
//
// kernels_general.hip gives a whole wavefront to one instance.  Its round-1 profile shows it is
// issue-bound, not latency-bound: ~4 500 vector + ~4 200 scalar instructions per NR iteration at N = 31,
// four waves per SIMD sharing the issue slots, while half (N = 31) or three quarters (N = 13) of every
// instruction's lanes idle.  Packing the instances of a wave shares each instruction between them.
//
// Same algorithm, same operation order per instance as kernels_general.hip / device_common.hpp
// (reference: include/solver.hpp:30-131, src/tanalisis.cpp:238-420, src/dcanalysis.cpp:46-68,95-163,
// 268-307): results are bit-identical to the one-instance-per-wave kernels.  What changes is the
// plumbing:
//   * a value of sub-lane s of a group reaches the group through ds_bpermute (__shfl), not through
//     v_readlane (which broadcasts one lane to the whole wave);
//   * ballots are cut into per-group masks, and the walks over candidate rows / active rows run as many
//     trips as the busiest group needs, the other groups idling under a predicate;
//   * control flow is wave-uniform: the groups of a wave may be at different time steps (hybrid
//     stepping), converge after different numbers of passes, or stop -- each of those is a per-group flag,
//     never a branch around a barrier.
// The planner (pivot log) stays on the one-instance-per-wave kernels.
//
// Measured (MI355X, B = 4096): buffer.sp DC operating points 20.1 ms -> 8.8 ms with G = 16.  The same
// code with G = 32 (two instances per wave, N <= 31) was measured on dbmixer.sp and is NOT used: 6.0e7
// NR-iter*inst/s with ds_bpermute broadcasts and 4.9e7 with pairs of v_readlane + select, against 6.7e7
// for one instance per wave -- at two per wave the vector forms of the walks (per-lane ffs, masks,
// predicates) cost about what the second instance saves, and the doubled LDS footprint halves the waves
// that hide the rest.
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace csim {

#pragma clang fp contract(off)

namespace {

// value of sub-lane `sub` of each group (sub is uniform within a group, may differ between groups)
template <int G> __device__ __forceinline__ double grp_get(double v, int sub, int q) { return __shfl(v, q * G + sub); }
template <int G> __device__ __forceinline__ unsigned grp_mask(bool pred, int q)
{
    return (unsigned)((__ballot(pred) >> (q * G)) & ((G == 32) ? 0xFFFFFFFFull : ((1ull << G) - 1ull)));
}

__device__ __forceinline__ double clampd_p(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ double base_gmin_p(const csim_consts& k, double s)
{
    s = clampd_p(s, 0.0, 1.0);
    return k.gmin_high * (1.0 - s) + k.gmin_low * s;
}

// lu_solve_wave of device_common.hpp for one group of G lanes (sub-lane g, group q); `on` = this group's
// solve counts (flags are only raised for such groups).  Returns the solution component of sub-lane g.
template <int G>
__device__ __forceinline__ double lu_solve_group(double* Gm, int N, int LD, double eps, int g, int q, bool on,
                                                 unsigned& flags, int32_t* curPiv)
{
    double diag = 1.0;
    bool failed = false;
    for (int k = 0; k < N; ++k) {
        double colv = (g < N) ? Gm[g * LD + k] : 0.0;
        const double av = fabs(colv);
        const double akk = grp_get<G>(av, k, q);
        int piv = k;
        double maxAbs = akk;
        // first row attaining the column maximum (solver.hpp:48-56): ascending walk over the non-zero
        // candidates below the diagonal; a NaN diagonal keeps pivot = k
        unsigned cand = grp_mask<G>(g > k && g < N && av > 0.0, q);
        if (!(akk == akk) || failed) cand = 0u;
        while (__any(cand != 0u)) {
            const bool has = cand != 0u;
            const int i = has ? __ffs((int)cand) - 1 : k;
            cand &= cand - 1u;
            const double v = grp_get<G>(av, i, q);
            if (has && v > maxAbs) { maxAbs = v; piv = i; }
        }
        if (!failed && maxAbs < eps) failed = true;                       // :58-61
        const bool live = !failed;
        if (curPiv && g == 0 && live) curPiv[k] = piv;
        const bool sw = live && piv != k;                                 // swap rows k and piv (columns >= k, RHS)
        if (sw && g >= k && g <= N) {
            const double a = Gm[k * LD + g], b = Gm[piv * LD + g];
            Gm[k * LD + g] = b;
            Gm[piv * LD + g] = a;
        }
        const double ck = grp_get<G>(colv, k, q), cp = grp_get<G>(colv, piv, q);
        if (sw && g == k) colv = cp;
        if (sw && g == piv) colv = ck;
        wave_sync();
        const double pivv = grp_get<G>(colv, k, q);
        if (g == k) diag = pivv;
        const double rowv = (g > k && g <= N) ? Gm[k * LD + g] : 0.0;
        const bool active = live && g > k && g < N && colv != 0.0;
        const double fmine = active ? colv / pivv : 0.0;                  // :71, row = sub-lane
        unsigned todo = grp_mask<G>(active, q);
        while (__any(todo != 0u)) {
            const bool has = todo != 0u;
            const int i = has ? __ffs((int)todo) - 1 : 0;
            todo &= todo - 1u;
            const double f = grp_get<G>(fmine, i, q);
            if (has && g > k && g <= N) Gm[i * LD + g] -= f * rowv;       // :74 (+ RHS)
        }
        wave_sync();
    }
    if (failed) {                                                         // :94-97: zero vector
        if (on) flags |= CSIM_ST_LU_TINY_PIVOT;
        return 0.0;
    }
    // back substitution (:116-128): row i descending subtracts U(i,j) x(j) for j ascending
    const double y = (g < N) ? Gm[g * LD + N] : 0.0;
    double xv = 0.0;
    for (int i = N - 1; i >= 0; --i) {
        const double u = (g > i && g < N) ? Gm[i * LD + g] : 0.0;
        const double prod = u * xv;
        unsigned todo = grp_mask<G>(g > i && g < N && prod != 0.0, q);
        double sum = grp_get<G>(y, i, q);
        while (__any(todo != 0u)) {
            const bool has = todo != 0u;
            const int j = has ? __ffs((int)todo) - 1 : 0;
            todo &= todo - 1u;
            const double pj = grp_get<G>(prod, j, q);
            if (has) sum -= pj;
        }
        const double d = grp_get<G>(diag, i, q);
        double xi;
        if (fabs(d) < eps) { xi = 0.0; if (on) flags |= CSIM_ST_LU_ZERO_DIAG; }
        else xi = sum / d;
        if (g == i) xv = xi;
    }
    return xv;
}

template <int G> __device__ __forceinline__ bool grp_all_finite(double v, int N, int g, int q)
{
    return grp_mask<G>(g < N && !isfinite(v), q) == 0u;
}

// every sub-lane returns the same sum of squares in index order (the oracle's norm)
__device__ __forceinline__ double grp_norm(double d, double* sc, int N, int g)
{
    if (g < N) sc[g] = d * d;
    wave_sync();
    double ss = 0.0;
    for (int i = 0; i < N; ++i) ss += sc[i];
    wave_sync();
    return sqrt(ss);
}

template <int G>
__device__ __forceinline__ bool grp_sequence_known(const int32_t* curPiv, const int32_t* alts, int nAlts, int N, int g, int q)
{
    wave_sync();
    bool known = false;
    for (int a = 0; a < nAlts; ++a) {
        bool same = true;
        for (int k = g; k < N; k += G) same = same && (curPiv[k] == alts[a * N + k]);
        known = known || grp_mask<G>(!same, q) == 0u;
    }
    return known;
}

} // namespace

// ------------------------------------------------------------------ DC (K2g packed)
template <int G>
__global__ void __launch_bounds__(64)
k_dc_packed(GenPlan pl, const double* __restrict__ params, int B, double* __restrict__ xout,
            int32_t* __restrict__ iters, uint32_t* __restrict__ status, const uint8_t* __restrict__ only)
{
    extern __shared__ double smp[];
    constexpr int IPW = 64 / G;
    const int lane = threadIdx.x, g = lane % G, q = lane / G;
    const int bRaw = blockIdx.x * IPW + q;
    const bool exists = bRaw < B;
    const int b = exists ? bRaw : B - 1;
    const bool mine = exists && !(only && !only[b]);         // fallback launches touch the flagged instances only
    if (!__any(mine)) return;
    const int N = pl.N, LD = pl.LD;
    const LdsLayout L = ldsLayout(N, LD, pl.nTerms, pl.P);
    double* base = smp + (size_t)q * (L.total + 1);
    double* Gm = base + L.G;
    double* T = base + L.T;
    double* Pv = base + L.P;
    double* xs = base + L.xs;
    double* sc = base + L.sc;
    const csim_consts& K = pl.k;

    for (int p = g; p < pl.P; p += G) Pv[p] = params[(int64_t)p * B + b];
    for (int t = g; t < pl.nTerms; t += G) T[t] = 0.0;
    if (g < N) xs[g] = 0.0;
    wave_sync();
    terms_const<false>(pl, Pv, T, 0.0, g, G);
    wave_sync();

    unsigned st = 0;
    int itTotal = 0;
    if (!pl.hasNonlinear) {
        terms_step_dc(pl, Pv, T, 1.0, g, G);                 // one solve at x = 0, full sources, no gmin (:46-68)
        if (g == 0) T[pl.termGmin] = 0.0;
        wave_sync();
        assemble(pl, T, Gm, g, G);
        const double xr = lu_solve_group<G>(Gm, N, LD, K.lu_eps, g, q, mine, st, nullptr);
        if (g < N) xs[g] = xr;
        itTotal = 1;
    } else {
        for (int step = 1; step <= K.dc_ramp_steps; ++step) {
            const double scale = (double)step / K.dc_ramp_steps;
            double gmin = base_gmin_p(K, scale);
            double prevErr = INFINITY;
            terms_step_dc(pl, Pv, T, scale, g, G);
            wave_sync();
            bool active = mine;                               // this group still iterates in this ramp step
            for (int iter = 0; iter < K.dc_max_iters; ++iter) {
                if (!__any(active)) break;
                terms_iter_mos(pl, Pv, T, xs, g, G);
                if (g == 0) T[pl.termGmin] = gmin;
                wave_sync();
                assemble(pl, T, Gm, g, G);
                const double xr = lu_solve_group<G>(Gm, N, LD, K.lu_eps, g, q, active, st, nullptr);
                if (active) ++itTotal;
                const bool finite = grp_all_finite<G>(xr, N, g, q);
                // ConvController::update, computed by every group, applied by the active ones
                const double alpha = clampd_p(K.dc_alpha, K.dc_alpha_min, K.dc_alpha_max);
                const double xo = (g < N) ? xs[g] : 0.0;
                const double xn = xo + alpha * (xr - xo);
                const double err = grp_norm(xn - xo, sc, N, g);
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
                    if (g < N) xs[g] = xn;
                    gmin = gnext;
                    prevErr = err;
                    if (err < K.dc_tol) active = false;                             // :150
                    else if (iter == K.dc_max_iters - 1) st |= CSIM_ST_DC_NONCONV;  // :153-158
                }
                wave_sync();
            }
        }
    }
    wave_sync();
    if (mine) {
        if (g < N) xout[(int64_t)g * B + b] = xs[g];
        if (g == 0) { iters[b] = itTotal; status[b] = only ? (st | CSIM_ST_SCHED_FALLBACK_DC) : st; }
    }
}

// ------------------------------------------------------------ transient (K1g packed)
template <int G>
__global__ void __launch_bounds__(64)
k_tran_packed(GenPlan pl, const double* __restrict__ params, int B, double dt,
              long long stepFirst, long long nSteps, const int32_t* __restrict__ probeEq, int nProbe, int outStride,
              double* __restrict__ wave, double* __restrict__ xio, long long* __restrict__ iters,
              uint32_t* __restrict__ status, int32_t* __restrict__ stepIters, const uint8_t* __restrict__ only,
              int32_t* __restrict__ done, int maxSteps, const int32_t* __restrict__ knownAlts, int nKnown)
{
    extern __shared__ double smp[];
    constexpr int IPW = 64 / G;
    const int lane = threadIdx.x, g = lane % G, q = lane / G;
    const int bRaw = blockIdx.x * IPW + q;
    const bool exists = bRaw < B;
    const int b = exists ? bRaw : B - 1;
    const long long d0 = done ? (long long)done[b] : 0;
    const bool mine = exists && !(only && !only[b]) && !(done && d0 >= nSteps);
    if (!__any(mine)) return;
    const long long sEnd = done ? (d0 + maxSteps < nSteps ? d0 + maxSteps : nSteps) : nSteps;
    const int N = pl.N, LD = pl.LD;
    const LdsLayout L = ldsLayout(N, LD, pl.nTerms, pl.P);
    double* base = smp + (size_t)q * (L.total + 1);
    double* Gm = base + L.G;
    double* T = base + L.T;
    double* Pv = base + L.P;
    double* xs = base + L.xs;
    double* xp = base + L.xp;
    double* sc = base + L.sc;
    int32_t* curPiv = (done && knownAlts) ? reinterpret_cast<int32_t*>(base + L.piv) : nullptr;
    const csim_consts& K = pl.k;
    const int slowIters = slowStepIters(K.tran_tol, K.tran_alpha, K.tran_max_iters);

    for (int p = g; p < pl.P; p += G) Pv[p] = params[(int64_t)p * B + b];
    for (int t = g; t < pl.nTerms; t += G) T[t] = 0.0;
    if (g < N) {
        const double v = xio[(int64_t)g * B + b];
        xs[g] = v;
        xp[g] = v;
    }
    wave_sync();
    terms_const<true>(pl, Pv, T, dt, g, G);
    if (g == 0) T[pl.termGmin] = K.tran_gmin;
    wave_sync();

    if (mine && stepFirst == 0 && d0 == 0 && wave)                         // t = 0 row (:250)
        for (int pq = g; pq < nProbe; pq += G) wave[((int64_t)0 * nProbe + pq) * B + b] = xs[probeEq[pq]];

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
        terms_step_tran(pl, Pv, T, xp, tNow, g, G);
        wave_sync();
        int it = 0;
        bool active = running;
        for (int iter = 0; iter < K.tran_max_iters; ++iter) {
            if (!__any(active)) break;
            terms_iter_mos(pl, Pv, T, xs, g, G);
            wave_sync();
            assemble(pl, T, Gm, g, G);                                     // :259-356
            const double xr = lu_solve_group<G>(Gm, N, LD, K.lu_eps, g, q, active, st, curPiv);   // :359
            if (curPiv) {                                                  // uniform: every group compares, the active ones keep the answer
                const bool known = grp_sequence_known<G>(curPiv, knownAlts, nKnown, N, g, q);
                if (active && stepKnown) stepKnown = known;
            }
            if (active) ++it;
            const bool finite = grp_all_finite<G>(xr, N, g, q);
            const double xo = (g < N) ? xs[g] : 0.0;
            const double xn = xo + K.tran_alpha * (xr - xo);               // :365
            const double err = grp_norm(xn - xo, sc, N, g);                // :366
            if (active && !finite) {                                       // :360-362
                st |= CSIM_ST_TRAN_NONFINITE;
                aborted = true;
                active = false;
            } else if (active) {
                if (g < N) xs[g] = xn;                                     // :367
                if (err < K.tran_tol) { stepConverged = it <= slowIters; active = false; }          // :369-371
                else if (iter == K.tran_max_iters - 1) st |= CSIM_ST_TRAN_NONCONV;                   // :372-376
            }
            wave_sync();
        }
        if (running) {
            itTotal += it;
            if (stepIters && g == 0) stepIters[(s - 1) * (int64_t)B + b] = it;
        }
        if (running && !aborted) {
            if (g < N) xp[g] = xs[g];                                      // :381-417
        }
        wave_sync();
        if (running && !aborted && wave && (gstep % outStride) == 0)       // :419
            for (int pq = g; pq < nProbe; pq += G)
                wave[((gstep / outStride) * nProbe + pq) * (int64_t)B + b] = xs[probeEq[pq]];
        // hybrid stepping: back on a recorded schedule and converging -> hand the instance back
        if (running && (aborted || (stepKnown && stepConverged))) running = false;
        if (running) { ++s; running = s <= sEnd; }
    }
    if (mine) {
        if (g < N) xio[(int64_t)g * B + b] = xs[g];
        if (g == 0) {
            iters[b] += itTotal;
            status[b] |= st;
            if (done) done[b] = aborted ? (int32_t)nSteps : (int32_t)sLast;
        }
    }
}

// ------------------------------------------------------------------ launchers
int packedLanesFor(int N) { return N <= 15 ? 16 : 64; }

hipError_t launchDcPacked(const GenPlan& pl, const double* dParams, int B, double* dX, int32_t* dIters,
                          uint32_t* dStatus, hipStream_t stream, const uint8_t* dOnly)
{
    const int G = packedLanesFor(pl.N), ipw = 64 / G;
    const LdsLayout L = ldsLayout(pl.N, pl.LD, pl.nTerms, pl.P);
    const size_t lds = sizeof(double) * (size_t)(L.total + 1) * ipw;
    const dim3 grid((B + ipw - 1) / ipw);
    if (G != 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dc_packed<16>, grid, dim3(64), lds, stream, pl, dParams, B, dX, dIters, dStatus, dOnly);
    return hipGetLastError();
}

hipError_t launchTranPacked(const GenPlan& pl, const double* dParams, int B, double dt, long long stepFirst,
                            long long nSteps, const int32_t* dProbeEq, int nProbe, int outStride, double* dWave,
                            double* dX, long long* dIters, uint32_t* dStatus, int32_t* dStepIters, const uint8_t* dOnly,
                            hipStream_t stream, int32_t* dDone, int maxSteps, const int32_t* dKnownAlts, int nKnown)
{
    const int G = packedLanesFor(pl.N), ipw = 64 / G;
    const LdsLayout L = ldsLayout(pl.N, pl.LD, pl.nTerms, pl.P);
    const size_t lds = sizeof(double) * (size_t)(L.total + 1) * ipw;
    const dim3 grid((B + ipw - 1) / ipw);
    if (G != 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tran_packed<16>, grid, dim3(64), lds, stream, pl, dParams, B, dt, stepFirst, nSteps, dProbeEq,
                       nProbe, outStride, dWave, dX, dIters, dStatus, dStepIters, dOnly, dDone, maxSteps, dKnownAlts, nKnown);
    return hipGetLastError();
}

} // namespace csim
