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

// ---- cross-lane plumbing inside one DPP row of 16 lanes (= one instance)
template <int L> __device__ __forceinline__ double row_bcast(double v)          // lane L of the row -> every lane of it
{
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + L, 0xF, 0xF, true);      // v_mov_b64_dpp row_newbcast:L
}
// the same with the lane in a variable that is a constant once the caller's loop is unrolled
__device__ __forceinline__ double row_bcast_at(double v, int L)
{
    switch (L) {
        case 0: return row_bcast<0>(v);   case 1: return row_bcast<1>(v);   case 2: return row_bcast<2>(v);   case 3: return row_bcast<3>(v);
        case 4: return row_bcast<4>(v);   case 5: return row_bcast<5>(v);   case 6: return row_bcast<6>(v);   case 7: return row_bcast<7>(v);
        case 8: return row_bcast<8>(v);   case 9: return row_bcast<9>(v);   case 10: return row_bcast<10>(v); case 11: return row_bcast<11>(v);
        case 12: return row_bcast<12>(v); case 13: return row_bcast<13>(v); case 14: return row_bcast<14>(v); default: return row_bcast<15>(v);
    }
}
// maximum over the 16 lanes of a row, in every lane; a NaN operand is ignored (v_max_f64)
__device__ __forceinline__ double row_shr_keep(double v, int n)                 // lane i <- lane i - n of the row; the first n lanes keep v
{
    const long long b = __double_as_longlong(v);
    long long r;
    switch (n) {
        case 1: r = __builtin_amdgcn_update_dpp(b, b, 0x111, 0xF, 0xF, false); break;
        case 2: r = __builtin_amdgcn_update_dpp(b, b, 0x112, 0xF, 0xF, false); break;
        case 4: r = __builtin_amdgcn_update_dpp(b, b, 0x114, 0xF, 0xF, false); break;
        default: r = __builtin_amdgcn_update_dpp(b, b, 0x118, 0xF, 0xF, false); break;
    }
    return __longlong_as_double(r);
}
__device__ __forceinline__ double row_max16(double v)
{
    v = fmax(v, row_shr_keep(v, 1));
    v = fmax(v, row_shr_keep(v, 2));
    v = fmax(v, row_shr_keep(v, 4));
    v = fmax(v, row_shr_keep(v, 8));
    return row_bcast<15>(v);                                                    // lane 15 holds the maximum of all 16
}

// Solver::luDecompose + solveLinearSystemLU (solver.hpp:30-131) for one group of 16 lanes with the augmented
// matrix in REGISTERS: sub-lane g keeps row g, a[j] = column j, a[N] = right-hand side.  Same operations in the
// same order as the reference -- rows are exchanged for real, so the pivot row of column k always sits in lane k
// and reaches the others with a compile-time DPP broadcast.  What the LDS version (lu_solve_wave,
// device_common.hpp) walks one candidate / one non-zero row at a time, a cross-lane round trip each, is one pass
// here: the pivot is max-reduced over the row of lanes and located with a ballot (the FIRST lane attaining it,
// solver.hpp:48-56), and every row below the pivot applies its own multiplier at once -- including the rows
// whose multiplier is zero, as solver.hpp:70-76 does (the LDS version skips those; equal unless the pivot row
// holds an Inf or NaN).  N is a template parameter: every loop unrolls, every index is a register.
// `on` = this group's solve counts (flags are only raised for such groups).
template <int N>
__device__ __forceinline__ double lu_solve_rows16_n(const double* Gm, int LD, double eps, int g, int q, bool on,
                                                    unsigned& flags, int32_t* curPiv)
{
    double a[N + 1];
#pragma unroll
    for (int j = 0; j <= N; ++j) a[j] = (g < N) ? Gm[g * LD + j] : 0.0;
    bool failed = false;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const double av = fabs(a[k]);
        const double akk = row_bcast_at(av, k);
        int piv = k;
        double maxAbs = akk;
        if (k + 1 < N) {
            // candidates below the diagonal; a NaN entry never wins a "val > maxAbs" (:53) and v_max_f64 drops it
            const double m = row_max16((g > k && g < N) ? av : -1.0);
            if (akk == akk && m > akk) {                                  // a NaN diagonal keeps pivot = k
                const unsigned cand = grp_mask<16>(g > k && g < N && av == m, q);
                maxAbs = m;
                piv = __ffs((int)cand) - 1;
            }
        }
        if (!failed && maxAbs < eps) failed = true;                       // :58-61
        const bool live = !failed;
        if (curPiv && g == 0 && live) curPiv[k] = piv;
        const bool sw = live && piv != k;                                 // :64-67 (columns >= k and the RHS matter)
        if (k + 1 < N && __any(sw)) {
            const int src = q * 16 + (sw ? (g == k ? piv : (g == piv ? k : g)) : g);
#pragma unroll
            for (int j = k; j <= N; ++j) a[j] = __shfl(a[j], src);
        }
        if (k + 1 < N) {
            const double pivv = row_bcast_at(a[k], k);
            double u[N + 1];
#pragma unroll
            for (int j = k + 1; j <= N; ++j) u[j] = row_bcast_at(a[j], k);           // the pivot row, to every lane
            if (live && g > k && g < N) {                                 // :70-76, rows below the pivot
                const double f = a[k] / pivv;                             // :71
#pragma unroll
                for (int j = k + 1; j <= N; ++j) a[j] = a[j] - f * u[j];  // :74 (+ RHS = forward substitution)
            }
        }
    }
    if (failed) {                                                         // :94-97: zero vector
        if (on) flags |= CSIM_ST_LU_TINY_PIVOT;
        return 0.0;
    }
    // back substitution (:116-128): row i (descending) subtracts U(i,j) x(j) for j ascending.  Every lane runs
    // the sum on its own row; lane i's is row i's, and its x(i) is broadcast for the rows above.
    double x[N];
    double xv = 0.0;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double sum = a[N];
#pragma unroll
        for (int j = i + 1; j < N; ++j) sum -= a[j] * x[j];               // :119
        const double d = row_bcast_at(a[i], i);                           // :121 U(i,i)
        const bool tiny = fabs(d) < eps;
        const double xi = tiny ? 0.0 : sum / d;                           // :122-126 (lane i's is x(i))
        if (on && tiny) flags |= CSIM_ST_LU_ZERO_DIAG;
        x[i] = row_bcast_at(xi, i);
        if (g == i) xv = xi;
    }
    return xv;
}

__device__ __forceinline__ double lu_solve_rows16(const double* Gm, int N, int LD, double eps, int g, int q, bool on,
                                                  unsigned& flags, int32_t* curPiv)
{
    switch (N) {        // uniform; one body per size, so that the size is a constant inside
        case 1: return lu_solve_rows16_n<1>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 2: return lu_solve_rows16_n<2>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 3: return lu_solve_rows16_n<3>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 4: return lu_solve_rows16_n<4>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 5: return lu_solve_rows16_n<5>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 6: return lu_solve_rows16_n<6>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 7: return lu_solve_rows16_n<7>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 8: return lu_solve_rows16_n<8>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 9: return lu_solve_rows16_n<9>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 10: return lu_solve_rows16_n<10>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 11: return lu_solve_rows16_n<11>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 12: return lu_solve_rows16_n<12>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 13: return lu_solve_rows16_n<13>(Gm, LD, eps, g, q, on, flags, curPiv);
        case 14: return lu_solve_rows16_n<14>(Gm, LD, eps, g, q, on, flags, curPiv);
        default: return lu_solve_rows16_n<15>(Gm, LD, eps, g, q, on, flags, curPiv);
    }
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
template <int G, bool STAGED>
__global__ void __launch_bounds__(64)
k_dc_packed(GenPlan plArg, const double* __restrict__ params, int B, double* __restrict__ xout,
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
    const int N = plArg.N, LD = plArg.LD;
    const LdsLayout L = ldsLayout(N, LD, plArg.nTerms, plArg.P);
    // one copy of the plan's index arrays for the groups of the wave, behind their private areas
    const GenPlan pl = STAGED ? plan_in_lds(plArg, reinterpret_cast<int32_t*>(smp + (size_t)IPW * (L.total + 1)), lane, 64) : plArg;
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
        const double xr = lu_solve_rows16(Gm, N, LD, K.lu_eps, g, q, mine, st, nullptr);
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
                const double xr = lu_solve_rows16(Gm, N, LD, K.lu_eps, g, q, active, st, nullptr);
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
template <int G, bool STAGED>
__global__ void __launch_bounds__(64)
k_tran_packed(GenPlan plArg, const double* __restrict__ params, int B, double dt,
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
    const int N = plArg.N, LD = plArg.LD;
    const LdsLayout L = ldsLayout(N, LD, plArg.nTerms, plArg.P);
    // one copy of the plan's index arrays for the groups of the wave, behind their private areas
    const GenPlan pl = STAGED ? plan_in_lds(plArg, reinterpret_cast<int32_t*>(smp + (size_t)IPW * (L.total + 1)), lane, 64) : plArg;
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
            const double xr = lu_solve_rows16(Gm, N, LD, K.lu_eps, g, q, active, st, curPiv);   // :359
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
    const size_t lds = sizeof(double) * (size_t)(L.total + 1) * ipw, staged = lds + sizeof(int32_t) * (size_t)planLdsInts(pl);
    const dim3 grid((B + ipw - 1) / ipw);
    if (G != 16) return hipErrorInvalidValue;
    if (staged <= kStagedLdsLimit)
        hipLaunchKernelGGL((k_dc_packed<16, true>), grid, dim3(64), staged, stream, pl, dParams, B, dX, dIters, dStatus, dOnly);
    else
        hipLaunchKernelGGL((k_dc_packed<16, false>), grid, dim3(64), lds, stream, pl, dParams, B, dX, dIters, dStatus, dOnly);
    return hipGetLastError();
}

hipError_t launchTranPacked(const GenPlan& pl, const double* dParams, int B, double dt, long long stepFirst,
                            long long nSteps, const int32_t* dProbeEq, int nProbe, int outStride, double* dWave,
                            double* dX, long long* dIters, uint32_t* dStatus, int32_t* dStepIters, const uint8_t* dOnly,
                            hipStream_t stream, int32_t* dDone, int maxSteps, const int32_t* dKnownAlts, int nKnown)
{
    const int G = packedLanesFor(pl.N), ipw = 64 / G;
    const LdsLayout L = ldsLayout(pl.N, pl.LD, pl.nTerms, pl.P);
    const size_t lds = sizeof(double) * (size_t)(L.total + 1) * ipw, staged = lds + sizeof(int32_t) * (size_t)planLdsInts(pl);
    const dim3 grid((B + ipw - 1) / ipw);
    if (G != 16) return hipErrorInvalidValue;
    if (staged <= kStagedLdsLimit)
        hipLaunchKernelGGL((k_tran_packed<16, true>), grid, dim3(64), staged, stream, pl, dParams, B, dt, stepFirst, nSteps, dProbeEq,
                           nProbe, outStride, dWave, dX, dIters, dStatus, dStepIters, dOnly, dDone, maxSteps, dKnownAlts, nKnown);
    else
        hipLaunchKernelGGL((k_tran_packed<16, false>), grid, dim3(64), lds, stream, pl, dParams, B, dt, stepFirst, nSteps, dProbeEq,
                           nProbe, outStride, dWave, dX, dIters, dStatus, dStepIters, dOnly, dDone, maxSteps, dKnownAlts, nKnown);
    return hipGetLastError();
}

} // namespace csim
