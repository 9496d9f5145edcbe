// kernels_gs.hip -- the reference's Gauss-Seidel linear solver and the DC operating point
// built on it (gfx950).
//
//   k_gs_solve   Solver::solveLinearSystemGaussSeidel     include/solver.hpp:139-204
//   k_dc_gs      dcSolveGaussSeidel (dispatch)             src/dcanalysis.cpp:254-258
//                dcSolveDirectGS                           src/dcanalysis.cpp:71-92
//                dcSolveNewtonGS                           src/dcanalysis.cpp:166-237
//
// A Gauss-Seidel sweep is sequential in the row index (row i uses the values rows < i just
// produced) and the reference subtracts the products of one row one after the other, so the only
// parallel axis that keeps its results bit for bit is the batch:
//   * k_gs_solve gives one LANE to one system; the matrices are transposed on the device to
//     [n*n][B] so that the 64 lanes of a wave read 64 consecutive doubles;
//   * k_dc_gs gives one wavefront to one circuit instance (stamping is element-parallel as in
//     the LU kernels); its sweeps run on lane 0 over the structural non-zeros of each row.
// Exactness of the short cuts of k_dc_gs: (a) a structural zero contributes "sum -= 0 * x(j)",
// which leaves sum as it is while x(j) is finite; (b) the moment any component turns non-finite
// the dense loops of the reference multiply it into EVERY other row (0 * inf = NaN), so the
// returned vector is non-finite -- and a non-finite vector is only ever tested with allFinite()
// and dropped (dcanalysis.cpp:209-215).  The sweeps therefore stop at the first non-finite
// component and report it.  On the shipped netlists every pass ends that way (their voltage
// sources put zeros on the diagonal, which the solver replaces by 1e-12): the reference's
// dcSolveGaussSeidel returns the zero vector for them, and so does this kernel.
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace csim {

#pragma clang fp contract(off)

// ---------------------------------------------------------------- stand-alone
// A: [n*n][B] (entry (i,j) of system t at (i*n+j)*B + t), b/x0/x/xOld: [n][B].
// Non-finite components are carried through the dense loops exactly as upstream.
__global__ void __launch_bounds__(64)
k_gs_solve(int n, int B, const double* __restrict__ A, const double* __restrict__ rhs,
           const double* __restrict__ x0, int maxIters, double tol, double* __restrict__ x,
           double* __restrict__ xOld, int32_t* __restrict__ sweeps)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const long long SB = B;
    for (int i = 0; i < n; ++i) x[i * SB + t] = x0 ? x0[i * SB + t] : 0.0;         // :145, :154-157
    const double diagEps = 1e-12;                                                    // :160
    int done = 0;
    for (int iter = 0; iter < maxIters; ++iter) {                                    // :162
        ++done;
        for (int i = 0; i < n; ++i) xOld[i * SB + t] = x[i * SB + t];                // :163
        for (int i = 0; i < n; ++i) {
            double diag = A[((long long)i * n + i) * SB + t];                        // :166
            if (fabs(diag) < diagEps) diag = (diag >= 0.0 ? 1.0 : -1.0) * diagEps;   // :169-173
            double sum = rhs[i * SB + t];                                            // :175
            for (int j = 0; j < i; ++j) sum -= A[((long long)i * n + j) * SB + t] * x[j * SB + t];          // :178-180
            for (int j = i + 1; j < n; ++j) sum -= A[((long long)i * n + j) * SB + t] * xOld[j * SB + t];   // :181-183
            x[i * SB + t] = sum / diag;                                              // :185
        }
        double ss = 0.0;                                                             // :188, index order
        for (int i = 0; i < n; ++i) { const double d = x[i * SB + t] - xOld[i * SB + t]; ss += d * d; }
        if (sqrt(ss) < tol) break;                                                   // :189-192
    }
    if (sweeps) sweeps[t] = done;
}

// ------------------------------------------------------------------ DC (GS)
namespace {

__device__ __forceinline__ double clampd_gs(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ double base_gmin_gs(const csim_consts& k, double s)
{
    s = clampd_gs(s, 0.0, 1.0);
    return k.gmin_high * (1.0 - s) + k.gmin_low * s;        // dcanalysis.hpp:45-48
}

// Sweeps on the LDS system [G | I] (row-major, leading dimension LD), warm start in xs, result in
// xr.  rowPtr/rowCol: column indices (ascending, diagonal excluded) of the entries of each row that
// may be non-zero.  Lane 0 works; returns false when a component turned non-finite.
__device__ bool gs_sweeps_lane0(const double* Gm, int N, int LD, const int32_t* rowPtr, const int32_t* rowCol,
                                const double* xs, double* xr, double* xo, int maxIters, double tol)
{
    for (int i = 0; i < N; ++i) xr[i] = xs[i];
    for (int iter = 0; iter < maxIters; ++iter) {
        for (int i = 0; i < N; ++i) xo[i] = xr[i];
        for (int i = 0; i < N; ++i) {
            double diag = Gm[i * LD + i];
            if (fabs(diag) < 1e-12) diag = (diag >= 0.0 ? 1.0 : -1.0) * 1e-12;
            double sum = Gm[i * LD + N];
            for (int c = rowPtr[i]; c < rowPtr[i + 1]; ++c) {
                const int j = rowCol[c];
                sum -= Gm[i * LD + j] * (j < i ? xr[j] : xo[j]);
            }
            const double v = sum / diag;
            xr[i] = v;
            if (!isfinite(v)) return false;
        }
        double ss = 0.0;
        for (int i = 0; i < N; ++i) { const double d = xr[i] - xo[i]; ss += d * d; }
        if (sqrt(ss) < tol) break;
    }
    return true;
}

// The reference's dense loops, every j != i, no early exit (include/solver.hpp:162-193): what dcSolveDirectGS returns
// when the sweeps diverge is whatever they left -- a mix of +-inf and NaN that depends on the order in which the
// non-finite values spread, structural zeros included (0 * inf = NaN).  Lane 0, only after gs_sweeps_lane0 gave up.
__device__ void gs_sweeps_dense_lane0(const double* Gm, int N, int LD, const double* xs, double* xr, double* xo,
                                      int maxIters, double tol)
{
    for (int i = 0; i < N; ++i) xr[i] = xs[i];
    for (int iter = 0; iter < maxIters; ++iter) {
        for (int i = 0; i < N; ++i) xo[i] = xr[i];
        for (int i = 0; i < N; ++i) {
            double diag = Gm[i * LD + i];
            if (fabs(diag) < 1e-12) diag = (diag >= 0.0 ? 1.0 : -1.0) * 1e-12;
            double sum = Gm[i * LD + N];
            for (int j = 0; j < i; ++j) sum -= Gm[i * LD + j] * xr[j];
            for (int j = i + 1; j < N; ++j) sum -= Gm[i * LD + j] * xo[j];
            xr[i] = sum / diag;
        }
        double ss = 0.0;
        for (int i = 0; i < N; ++i) { const double d = xr[i] - xo[i]; ss += d * d; }
        if (sqrt(ss) < tol) break;
    }
}

} // namespace

__global__ void __launch_bounds__(64)
k_dc_gs(GenPlan pl, const int32_t* __restrict__ rowPtr, const int32_t* __restrict__ rowCol,
        const double* __restrict__ params, int B, double* __restrict__ xout,
        int32_t* __restrict__ iters, uint32_t* __restrict__ status)
{
    extern __shared__ double sm[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int N = pl.N, LD = pl.LD;
    const LdsLayout L = ldsLayout(N, LD, pl.nTerms, pl.P);
    double* Gm = sm + L.G;
    double* T = sm + L.T;
    double* Pv = sm + L.P;
    double* xs = sm + L.xs;
    double* xr = sm + L.xp;                 // raw solution of the inner solve
    double* xo = sm + L.total;              // previous sweep (N doubles appended by the launcher)
    int* flag = reinterpret_cast<int*>(xo + N);
    const csim_consts& K = pl.k;
    const int gsSweeps = 2000;              // dcanalysis.cpp:88, :207
    const double gsTol = 1e-10;
    const int maxNewton = 60;               // :176
    const double tol = 1e-9;                // :177

    for (int p = lane; p < pl.P; p += 64) Pv[p] = params[(int64_t)p * B + b];
    for (int t = lane; t < pl.nTerms; t += 64) T[t] = 0.0;
    if (lane < N) xs[lane] = 0.0;
    wave_sync();
    terms_const<false>(pl, Pv, T, 0.0, lane);
    wave_sync();

    unsigned st = 0;
    int itTotal = 0;
    if (!pl.hasNonlinear) {
        // dcSolveDirectGS: one system at x = 0, full sources, no gmin, solved from the zero vector (:71-92)
        terms_step_dc(pl, Pv, T, 1.0, lane);
        if (lane == 0) T[pl.termGmin] = 0.0;
        wave_sync();
        assemble(pl, T, Gm, lane);
        if (lane == 0) {
            // the reference returns whatever the sweeps left, finite or not, and checks nothing (:89-91): when the
            // sparse sweeps meet a non-finite value the solve is redone with the reference's dense loops, whose
            // pattern of +-inf / NaN is then the reference's, component by component
            if (!gs_sweeps_lane0(Gm, N, LD, rowPtr, rowCol, xs, xr, xo, gsSweeps, gsTol))
                gs_sweeps_dense_lane0(Gm, N, LD, xs, xr, xo, gsSweeps, gsTol);
        }
        wave_sync();
        if (lane < N) xs[lane] = xr[lane];
        itTotal = 1;
    } else {
        for (int step = 1; step <= K.dc_ramp_steps; ++step) {                       // :183
            const double scale = (double)step / K.dc_ramp_steps;
            double gmin = base_gmin_gs(K, scale);                                   // :186
            double prevErr = INFINITY;
            const int maxIterThisStep = (step == K.dc_ramp_steps) ? maxNewton * 2 : maxNewton;   // :188-191
            terms_step_dc(pl, Pv, T, scale, lane);
            wave_sync();
            for (int iter = 0; iter < maxIterThisStep; ++iter) {
                terms_iter_mos(pl, Pv, T, xs, lane);
                if (lane == 0) T[pl.termGmin] = gmin;
                wave_sync();
                assemble(pl, T, Gm, lane);                                          // :193-203
                if (lane == 0) *flag = gs_sweeps_lane0(Gm, N, LD, rowPtr, rowCol, xs, xr, xo, gsSweeps, gsTol) ? 1 : 0;   // :206-207
                wave_sync();
                ++itTotal;
                if (!*flag) {                                                       // :209-215
                    gmin = fmin(gmin * 10.0, 1e-2);
                    st |= CSIM_ST_DC_NONFINITE;
                    wave_sync();
                    continue;
                }
                // ConvController::update (:268-307); it ignores alphaCurrent (:274)
                const double alpha = clampd_gs(K.dc_alpha, K.dc_alpha_min, K.dc_alpha_max);
                const double xo_ = (lane < N) ? xs[lane] : 0.0;
                const double xn = xo_ + alpha * ((lane < N ? xr[lane] : 0.0) - xo_);
                const double err = norm_in_order(xn - xo_, Gm, N, lane);            // Gm is dead here: scratch
                const double gb = base_gmin_gs(K, scale);
                double gnext = gb;
                if (iter == 0 || !isfinite(prevErr)) gnext = gb;
                else if (err > prevErr * K.slow_ratio) gnext = fmin(gmin * 2.0, K.gmin_abs_max);
                else if (err < prevErr * K.fast_ratio) gnext = 0.5 * gmin + 0.5 * gb;
                else gnext = 0.7 * gmin + 0.3 * gb;
                if (lane < N) xs[lane] = xn;                                        // :220
                wave_sync();
                gmin = gnext;
                prevErr = err;
                if (err < tol) break;                                               // :225-227
                if (iter == maxNewton - 1) st |= CSIM_ST_DC_NONCONV;                // :228-233
            }
        }
    }
    wave_sync();
    if (lane < N) xout[(int64_t)lane * B + b] = xs[lane];
    if (lane == 0) { iters[b] = itTotal; status[b] = st; }
}

// ------------------------------------------------------------------ launchers
hipError_t launchGsSolve(int n, int B, const double* dAt, const double* dRhs, const double* dX0, int maxIters,
                         double tol, double* dX, double* dXold, int32_t* dSweeps, hipStream_t stream)
{
    hipLaunchKernelGGL(k_gs_solve, dim3((B + 63) / 64), dim3(64), 0, stream, n, B, dAt, dRhs, dX0, maxIters, tol, dX,
                       dXold, dSweeps);
    return hipGetLastError();
}

hipError_t launchDcGs(const GenPlan& pl, const int32_t* dRowPtr, const int32_t* dRowCol, const double* dParams, int B,
                      double* dX, int32_t* dIters, uint32_t* dStatus, hipStream_t stream)
{
    const LdsLayout L = ldsLayout(pl.N, pl.LD, pl.nTerms, pl.P);
    const size_t lds = sizeof(double) * (size_t)(L.total + pl.N + 1);
    hipLaunchKernelGGL(k_dc_gs, dim3(B), dim3(64), lds, stream, pl, dRowPtr, dRowCol, dParams, B, dX, dIters, dStatus);
    return hipGetLastError();
}

} // namespace csim
