// kernels_general.hip -- wave-per-instance kernels (gfx950): DC operating
// point (K2), backward-Euler transient (K1) and the stand-alone batched LU.
//
// One 64-lane wavefront = one workgroup = one circuit instance; the dense
// augmented system sits in LDS and is solved in place (device_common.hpp).
// These kernels take any circuit with N <= 63 unknowns and make every pivot
// decision at run time, exactly as the reference does; they are the planner
// and the fallback of the circuit-specialised lane-per-instance kernels.
//
// Iteration control restated from the reference:
//   k_dc_general    dcSolveLU dispatch          src/dcanalysis.cpp:242-262
//                   dcSolveDirectLU             src/dcanalysis.cpp:46-68
//                   dcSolveNewtonLU             src/dcanalysis.cpp:95-163
//                   ConvController::update      src/dcanalysis.cpp:268-307
//   k_tran_general  runTransientAnalysisBackwardEuler  src/tanalisis.cpp:238-420
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace csim {

namespace {

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ double base_gmin(const csim_consts& k, double s)
{
    s = clampd(s, 0.0, 1.0);
    return k.gmin_high * (1.0 - s) + k.gmin_low * s;        // dcanalysis.hpp:45-48
}

__device__ __forceinline__ bool wave_all_finite(double v, int N, int lane)
{
    const bool bad = (lane < N) && !isfinite(v);
    return __ballot(bad) == 0ull;
}

} // namespace

// ------------------------------------------------------------------ DC (K2)
__global__ void __launch_bounds__(64)
k_dc_general(GenPlan pl, const double* __restrict__ params, int B,
             double* __restrict__ xout, int32_t* __restrict__ iters, uint32_t* __restrict__ status,
             const uint8_t* __restrict__ only, int32_t* __restrict__ pivLog, int pivInstance)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (only && !only[b]) return;           // fallback / planner launches touch the flagged instances only
    int32_t* myPivLog = (pivLog && b == pivInstance) ? pivLog : nullptr;
    const int N = pl.N, LD = pl.LD;
    const LdsLayout L = ldsLayout(N, LD, pl.nTerms, pl.P);
    double* Gm = sm + L.G;
    double* T = sm + L.T;
    double* Pv = sm + L.P;
    double* xs = sm + L.xs;
    double* sc = sm + L.sc;
    const csim_consts& K = pl.k;

    for (int p = lane; p < pl.P; p += 64) Pv[p] = params[(int64_t)p * B + b];
    for (int t = lane; t < pl.nTerms; t += 64) T[t] = 0.0;
    if (lane < N) xs[lane] = 0.0;
    wave_sync();
    terms_const<false>(pl, Pv, T, 0.0, lane);
    wave_sync();

    unsigned st = 0;
    int itTotal = 0;

    if (!pl.hasNonlinear) {
        // linear circuit: one solve at x = 0, full sources, NO gmin (dcanalysis.cpp:46-68)
        terms_step_dc(pl, Pv, T, 1.0, lane);
        if (lane == 0) T[pl.termGmin] = 0.0;
        wave_sync();
        assemble(pl, T, Gm, lane);
        const double xr = lu_solve_wave(Gm, N, LD, K.lu_eps, lane, st, myPivLog);
        if (lane < N) xs[lane] = xr;
        itTotal = 1;
    } else {
        for (int step = 1; step <= K.dc_ramp_steps; ++step) {
            const double scale = (double)step / K.dc_ramp_steps;        // :113
            double gmin = base_gmin(K, scale);                          // :116
            double prevErr = INFINITY;                                  // :117
            terms_step_dc(pl, Pv, T, scale, lane);
            wave_sync();
            for (int iter = 0; iter < K.dc_max_iters; ++iter) {
                terms_iter_mos(pl, Pv, T, xs, lane);
                if (lane == 0) T[pl.termGmin] = gmin;
                wave_sync();
                assemble(pl, T, Gm, lane);
                const double xr = lu_solve_wave(Gm, N, LD, K.lu_eps, lane, st, myPivLog, nullptr);   // :134
                ++itTotal;
                if (!wave_all_finite(xr, N, lane)) {                    // :135-138
                    gmin = fmin(gmin * K.gmin_nonfinite_mul, K.gmin_nonfinite_cap);
                    st |= CSIM_ST_DC_NONFINITE;
                    continue;
                }
                // ConvController::update
                const double alpha = clampd(K.dc_alpha, K.dc_alpha_min, K.dc_alpha_max);   // :274
                const double xo = (lane < N) ? xs[lane] : 0.0;
                const double xn = xo + alpha * (xr - xo);               // :275
                const double err = norm_in_order(xn - xo, sc, N, lane); // :276
                const double gb = base_gmin(K, scale);
                double gnext = gb;
                if (iter == 0 || !isfinite(prevErr)) gnext = gb;                          // :280-282
                else if (err > prevErr * K.slow_ratio) gnext = fmin(gmin * 2.0, K.gmin_abs_max);   // :285-288
                else if (err < prevErr * K.fast_ratio) gnext = 0.5 * gmin + 0.5 * gb;     // :289-293
                else gnext = 0.7 * gmin + 0.3 * gb;                                       // :296
                if (lane < N) xs[lane] = xn;                            // :145
                wave_sync();
                gmin = gnext;
                prevErr = err;
                if (err < K.dc_tol) break;                              // :150
                if (iter == K.dc_max_iters - 1) st |= CSIM_ST_DC_NONCONV;   // :153-158
            }
        }
    }
    wave_sync();
    if (lane < N) xout[(int64_t)lane * B + b] = xs[lane];
    if (lane == 0) {
        iters[b] = itTotal;
        status[b] = (only && !pivLog) ? (st | CSIM_ST_SCHED_FALLBACK_DC) : st;
    }
}

// ------------------------------------------------------------ transient (K1)
__global__ void __launch_bounds__(64)
k_tran_general(GenPlan pl, const double* __restrict__ params, int B, double dt,
               long long stepFirst, long long nSteps,
               const int32_t* __restrict__ probeEq, int nProbe, int outStride,
               double* __restrict__ wave, double* __restrict__ xio,
               long long* __restrict__ iters, uint32_t* __restrict__ status,
               int32_t* __restrict__ stepIters, const uint8_t* __restrict__ only,
               int32_t* __restrict__ pivLog, int pivInstance, int32_t* __restrict__ done, int maxSteps,
               const int32_t* __restrict__ knownAlts, int nKnown)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (only && !only[b]) return;           // planner launches touch the chosen instance only
    // hybrid stepping: `done` = steps of this launch each instance has completed; this kernel
    // advances an unfinished instance by at most maxSteps and hands it back
    const long long d0 = done ? (long long)done[b] : 0;
    if (done && d0 >= nSteps) return;
    const long long sEnd = done ? (d0 + maxSteps < nSteps ? d0 + maxSteps : nSteps) : nSteps;
    int32_t* myPivLog = (pivLog && b == pivInstance) ? pivLog : nullptr;
    const int N = pl.N, LD = pl.LD;
    const LdsLayout L = ldsLayout(N, LD, pl.nTerms, pl.P);
    double* Gm = sm + L.G;
    double* T = sm + L.T;
    double* Pv = sm + L.P;
    double* xs = sm + L.xs;
    double* xp = sm + L.xp;
    double* sc = sm + L.sc;
    int32_t* curPiv = (done && knownAlts) ? reinterpret_cast<int32_t*>(sm + L.piv) : nullptr;
    const csim_consts& K = pl.k;

    for (int p = lane; p < pl.P; p += 64) Pv[p] = params[(int64_t)p * B + b];
    for (int t = lane; t < pl.nTerms; t += 64) T[t] = 0.0;
    if (lane < N) {
        const double v = xio[(int64_t)lane * B + b];
        xs[lane] = v;
        xp[lane] = v;                       // histories come from the previous state (:139-180)
    }
    wave_sync();
    terms_const<true>(pl, Pv, T, dt, lane);
    if (lane == 0) T[pl.termGmin] = K.tran_gmin;
    wave_sync();

    if (stepFirst == 0 && d0 == 0 && wave)          // t = 0 row (:250)
        for (int q = lane; q < nProbe; q += 64) wave[((int64_t)0 * nProbe + q) * B + b] = xs[probeEq[q]];

    unsigned st = (status[b] & CSIM_ST_TRAN_NONFINITE);
    if (done) st |= CSIM_ST_SCHED_FALLBACK;
    long long itTotal = 0;
    bool aborted = (st & CSIM_ST_TRAN_NONFINITE) != 0;                 // an instance the reference would have thrown on stays stopped

    const int slowIters = slowStepIters(K.tran_tol, K.tran_alpha, K.tran_max_iters);
    long long sLast = d0;
    for (long long s = d0 + 1; s <= sEnd && !aborted; ++s) {
        sLast = s;
        bool stepKnown = curPiv != nullptr;     // every factorisation of this step on a known sequence?
        bool stepConverged = false;             // a slow step (plan.hpp slowStepIters) is not handed back either
        const long long gstep = stepFirst + s;
        const double tNow = (double)(int)gstep * dt;                    // :256
        terms_step_tran(pl, Pv, T, xp, tNow, lane);
        wave_sync();
        int it = 0;
        for (int iter = 0; iter < K.tran_max_iters; ++iter) {
            terms_iter_mos(pl, Pv, T, xs, lane);
            wave_sync();
            assemble(pl, T, Gm, lane);                                  // :259-356
            const double xr = lu_solve_wave(Gm, N, LD, K.lu_eps, lane, st, myPivLog, curPiv);   // :359
            if (stepKnown) stepKnown = sequence_is_known(curPiv, knownAlts, nKnown, N, lane);
            ++it;
            if (!wave_all_finite(xr, N, lane)) {                        // :360-362
                st |= CSIM_ST_TRAN_NONFINITE;
                aborted = true;
                break;
            }
            const double xo = (lane < N) ? xs[lane] : 0.0;
            const double xn = xo + K.tran_alpha * (xr - xo);            // :365
            const double err = norm_in_order(xn - xo, sc, N, lane);     // :366
            if (lane < N) xs[lane] = xn;                                // :367
            wave_sync();
            if (err < K.tran_tol) { stepConverged = it <= slowIters; break; }   // :369-371
            if (iter == K.tran_max_iters - 1) st |= CSIM_ST_TRAN_NONCONV;   // :372-376
        }
        itTotal += it;
        if (stepIters && lane == 0) stepIters[(s - 1) * (int64_t)B + b] = it;
        if (aborted) break;
        if (lane < N) xp[lane] = xs[lane];                              // :381-417
        wave_sync();
        if (wave && (gstep % outStride) == 0)                           // :419
            for (int q = lane; q < nProbe; q += 64)
                wave[((gstep / outStride) * nProbe + q) * (int64_t)B + b] = xs[probeEq[q]];
        if (stepKnown && stepConverged) break;  // back on a recorded schedule and converging: hand the instance back
    }

    if (lane < N) xio[(int64_t)lane * B + b] = xs[lane];
    if (lane == 0) {
        iters[b] += itTotal;
        status[b] |= st;
        if (done) done[b] = aborted ? (int32_t)nSteps : (int32_t)sLast;
    }
}

// ------------------------------------------------------ stand-alone LU solve
// Solver::solveLinearSystemLU for B systems (include/solver.hpp:83-131)
__global__ void __launch_bounds__(64)
k_lu_solve(int n, int LD, int B, const double* __restrict__ A, const double* __restrict__ rhs,
           double* __restrict__ x, uint32_t* __restrict__ flags, double eps)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const double* Ab = A + (int64_t)b * n * n;
    for (int i = lane; i < n * n; i += 64) sm[(i / n) * LD + (i % n)] = Ab[i];
    if (lane < n) sm[lane * LD + n] = rhs[(int64_t)b * n + lane];
    wave_sync();
    unsigned st = 0;
    const double xv = lu_solve_wave(sm, n, LD, eps, lane, st);
    if (lane < n) x[(int64_t)b * n + lane] = xv;
    if (flags && lane == 0) flags[b] = st;
}

// Solver::luDecompose for B matrices (include/solver.hpp:30-80)
__global__ void __launch_bounds__(64)
k_lu_factor(int n, int LD, int B, const double* __restrict__ A, double* __restrict__ LU,
            int32_t* __restrict__ perm, uint32_t* __restrict__ flags, double eps)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const double* Ab = A + (int64_t)b * n * n;
    for (int i = lane; i < n * n; i += 64) sm[(i / n) * LD + (i % n)] = Ab[i];
    wave_sync();
    int pv = 0;
    const bool ok = lu_factor_wave(sm, n, LD, eps, lane, pv);
    wave_sync();
    for (int i = lane; i < n * n; i += 64) LU[(int64_t)b * n * n + i] = sm[(i / n) * LD + (i % n)];
    if (lane < n) perm[(int64_t)b * n + lane] = pv;
    if (flags && lane == 0) flags[b] = ok ? 0u : CSIM_ST_LU_TINY_PIVOT;
}

// ------------------------------------------------------------------ launchers
hipError_t launchLuFactor(int n, int B, const double* dA, double* dLU, int32_t* dPerm, uint32_t* dFlags,
                          double eps, hipStream_t stream)
{
    const int LD = ldFor(n);
    const size_t lds = sizeof(double) * (size_t)n * (size_t)LD;
    hipLaunchKernelGGL(k_lu_factor, dim3(B), dim3(64), lds, stream, n, LD, B, dA, dLU, dPerm, dFlags, eps);
    return hipGetLastError();
}

hipError_t launchDcGeneral(const GenPlan& pl, const double* dParams, int B, double* dX,
                           int32_t* dIters, uint32_t* dStatus, hipStream_t stream, const uint8_t* dOnly,
                           int32_t* dPivLog, int pivInstance)
{
    // small circuits: several instances per wavefront (kernels_packed.hip); the planner keeps the wave per instance
    if (!dPivLog && packedLanesFor(pl) < 64) return launchDcPacked(pl, dParams, B, dX, dIters, dStatus, stream, dOnly);
    const LdsLayout L = ldsLayout(pl.N, pl.LD, pl.nTerms, pl.P);
    const size_t lds = sizeof(double) * (size_t)L.total;
    hipLaunchKernelGGL(k_dc_general, dim3(B), dim3(64), lds, stream, pl, dParams, B, dX, dIters, dStatus, dOnly,
                       dPivLog, pivInstance);
    return hipGetLastError();
}

hipError_t launchTranGeneral(const GenPlan& pl, const double* dParams, int B, double dt,
                             long long stepFirst, long long nSteps, const int32_t* dProbeEq, int nProbe,
                             int outStride, double* dWave, double* dX, long long* dIters,
                             uint32_t* dStatus, int32_t* dStepIters, const uint8_t* dOnly,
                             hipStream_t stream, int32_t* dPivLog, int pivInstance, int32_t* dDone, int maxSteps,
                             const int32_t* dKnownAlts, int nKnown)
{
    if (!dPivLog && packedLanesFor(pl) < 64)
        return launchTranPacked(pl, dParams, B, dt, stepFirst, nSteps, dProbeEq, nProbe, outStride, dWave, dX, dIters, dStatus,
                                dStepIters, dOnly, stream, dDone, maxSteps, dKnownAlts, nKnown);
    const LdsLayout L = ldsLayout(pl.N, pl.LD, pl.nTerms, pl.P);
    const size_t lds = sizeof(double) * (size_t)L.total;
    hipLaunchKernelGGL(k_tran_general, dim3(B), dim3(64), lds, stream, pl, dParams, B, dt, stepFirst, nSteps,
                       dProbeEq, nProbe, outStride, dWave, dX, dIters, dStatus, dStepIters, dOnly, dPivLog, pivInstance, dDone, maxSteps, dKnownAlts, nKnown);
    return hipGetLastError();
}

hipError_t launchLuSolve(int n, int B, const double* dA, const double* dRhs, double* dX,
                         uint32_t* dFlags, double eps, hipStream_t stream)
{
    if (n <= 32) return launchLuSolvePacked(n, B, dA, dRhs, dX, dFlags, eps, stream);     // four systems per wave, in registers
    const int LD = ldFor(n);
    const size_t lds = sizeof(double) * (size_t)n * (size_t)LD;
    hipLaunchKernelGGL(k_lu_solve, dim3(B), dim3(64), lds, stream, n, LD, B, dA, dRhs, dX, dFlags, eps);
    return hipGetLastError();
}

size_t generalLdsBytes(const GenPlan& pl)
{
    return sizeof(double) * (size_t)ldsLayout(pl.N, pl.LD, pl.nTerms, pl.P).total;
}

} // namespace csim
