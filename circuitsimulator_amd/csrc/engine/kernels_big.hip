// kernels_big.hip -- general kernels for circuits with 64 <= N <= 1024 unknowns (as far as one CU's LDS
// holds the structure: bigSupports())
// (BASELINE config [3]: RC ladder, N = 257).
//
// The LDS-dense layout of kernels_general.hip needs N*(N+2)*8 bytes per instance
// (528 KB at N = 257) and one row per lane; neither holds here.  This variant
// keeps the same algorithm and the same run-time pivoting but stores the system
// differently:
//   * values: one dense row-major N x LD scratch matrix per instance in GLOBAL
//     memory (HBM/L2), all zeros between solves;
//   * structure: an LDS bit matrix (N x (N+1) bits, 9.3 KB at N = 257) of the
//     entries that may be non-zero -- stamped entries plus fill.  Only entries whose
//     bit is set are ever read, updated or cleared, so the traffic is that of the
//     sparse system (a few entries per elimination step), not of the dense one;
//   * row swaps are logical (rowOf[position] in LDS): the pivot rule still walks
//     POSITIONS in ascending order, so "first row attaining the column maximum"
//     (solver.hpp:48-56) is preserved.
// One wavefront = one workgroup = one instance, as in the small kernels.  It is the
// planner, the DC kernel and the fallback for large N; the throughput path is the
// generated lane-per-instance kernel (codegen.cpp).
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace csim {

namespace {

constexpr int BIG_MAX_N = 1024;          // register arrays below have one entry per 64 columns
constexpr int BIG_CHUNKS = (BIG_MAX_N + 1 + 63) / 64;       // column chunks of 64 lanes (incl. RHS)

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ double base_gmin(const csim_consts& k, double s)
{
    s = clampd(s, 0.0, 1.0);
    return k.gmin_high * (1.0 - s) + k.gmin_low * s;
}

struct BigLds {
    double* T; double* Pv; double* xs; double* xp; double* xr;
    uint32_t* bm;       // [N][W32]
    int32_t* rowOf;     // [N] position -> physical row
    int32_t* candRow;   // [64] candidate rows of the current column
    double* candVal;    // [64]
    int W32;
};

__host__ __device__ inline size_t bigLdsBytes(int N, int nTerms, int P)
{
    const int W32 = (N + 1 + 31) / 32;
    size_t b = sizeof(double) * (size_t)(nTerms + P + 3 * N + 64);
    b += sizeof(uint32_t) * (size_t)N * W32;
    b += sizeof(int32_t) * (size_t)(N + 64);
    return b + 64;
}

__device__ __forceinline__ BigLds carve(unsigned char* base, int N, int nTerms, int P)
{
    BigLds L;
    L.W32 = (N + 1 + 31) / 32;
    double* d = reinterpret_cast<double*>(base);
    L.T = d; d += nTerms;
    L.Pv = d; d += P;
    L.xs = d; d += N;
    L.xp = d; d += N;
    L.xr = d; d += N;
    L.candVal = d; d += 64;
    L.bm = reinterpret_cast<uint32_t*>(d);
    L.rowOf = reinterpret_cast<int32_t*>(L.bm + (size_t)N * L.W32);
    L.candRow = L.rowOf + N;
    return L;
}

__device__ __forceinline__ bool bit_of(const BigLds& L, int r, int c) { return (L.bm[r * L.W32 + (c >> 5)] >> (c & 31)) & 1u; }

// build [G | I]: only the structural non-zeros are written; their bits are set
__device__ __forceinline__ void assemble_big(const GenPlan& pl, const BigLds& L, double* __restrict__ Gg, int lane)
{
    const int N = pl.N, LD = pl.LD;
    for (int i = lane; i < N * L.W32; i += 64) L.bm[i] = 0u;
    for (int i = lane; i < N; i += 64) L.rowOf[i] = i;
    wave_sync();
    for (int n = lane; n < pl.nnzG; n += 64) {
        double acc = 0.0;
        for (int c = pl.gPtr[n]; c < pl.gPtr[n + 1]; ++c) {
            const int con = pl.gCon[c];
            const double v = L.T[con >> 1];
            acc = (con & 1) ? acc - v : acc + v;
        }
        const int pos = pl.gPos[n];
        Gg[pos] = acc;
        atomicOr(&L.bm[(pos / LD) * L.W32 + ((pos % LD) >> 5)], 1u << ((pos % LD) & 31));
    }
    for (int n = lane; n < pl.nnzI; n += 64) {
        double acc = 0.0;
        for (int c = pl.iPtr[n]; c < pl.iPtr[n + 1]; ++c) {
            const int con = pl.iCon[c];
            const double v = L.T[con >> 1];
            acc = (con & 1) ? acc - v : acc + v;
        }
        const int r = pl.iRow[n];
        Gg[r * LD + N] = acc;
        atomicOr(&L.bm[r * L.W32 + (N >> 5)], 1u << (N & 31));
    }
    __threadfence_block();
    wave_sync();
}

// put the scratch matrix back to all zeros (only the touched entries)
__device__ __forceinline__ void clear_big(const GenPlan& pl, const BigLds& L, double* __restrict__ Gg, int lane)
{
    const int N = pl.N, LD = pl.LD;
    for (int r = lane; r < N; r += 64) {
        for (int w = 0; w < L.W32; ++w) {
            uint32_t m = L.bm[r * L.W32 + w];
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                Gg[r * LD + w * 32 + b] = 0.0;
            }
        }
    }
    __threadfence_block();
    wave_sync();
}

// Solver::solveLinearSystemLU on the bit-guided storage.  Result in L.xr[0..N).
// Same pivot rule, same elimination order per row, forward substitution fused
// (RHS = column N), back substitution row-wise with ascending j (solver.hpp:116-128).
__device__ __forceinline__ void lu_solve_big(const GenPlan& pl, const BigLds& L, double* __restrict__ Gg, double eps,
                                             int lane, unsigned& flags, int32_t* pivLog)
{
    const int N = pl.N, LD = pl.LD;
    int32_t* const logCur = pivLog ? pivlog_cur(pivLog, N) : nullptr;
    bool failed = false;

    for (int k = 0; k < N && !failed; ++k) {
        // ---- pivot search over positions k..N-1, ascending; candidates = set bits in column k
        int piv = k;
        double maxAbs = 0.0;
        int nCand = 0;
        {
            const int rk = L.rowOf[k];
            const double akk = bit_of(L, rk, k) ? fabs(Gg[rk * LD + k]) : 0.0;     // uniform read
            maxAbs = akk;
            const bool nanDiag = akk != akk;
            for (int base = k; base < N; base += 64) {
                const int i = base + lane;
                int r = -1;
                double a = 0.0;
                if (i < N) {
                    r = L.rowOf[i];
                    if (bit_of(L, r, k)) a = Gg[r * LD + k]; else r = -1;
                }
                const double av = fabs(a);
                // every position >= k with a non-zero entry in column k is a row of this step:
                // one of them becomes the pivot row, the others are eliminated.  The list holds
                // PHYSICAL rows, so the logical swap below does not disturb it.
                const bool inList = r >= 0 && a != 0.0;
                const unsigned long long has = __ballot(inList);
                if (inList) {
                    const int slot = nCand + __popcll(has & ((1ull << lane) - 1ull));
                    if (slot < 64) { L.candRow[slot] = r; L.candVal[slot] = a; }
                }
                nCand += __popcll(has);
                if (!nanDiag) {
                    unsigned long long cand = __ballot(r >= 0 && i > k && av > 0.0);
                    while (cand) {
                        const int l = __ffsll((long long)cand) - 1;
                        cand &= cand - 1;
                        const double v = read_lane(av, l);
                        if (v > maxAbs) { maxAbs = v; piv = base + l; }
                    }
                }
            }
        }
        if (nCand > 64) { flags |= CSIM_ST_LU_TINY_PIVOT; failed = true; break; }   // > 64 rows in one column: not this kernel
        if (maxAbs < eps) { failed = true; break; }
        if (logCur && lane == 0) logCur[k] = piv;
        wave_sync();
        // ---- logical swap of positions k and piv
        const int rowK = L.rowOf[piv];          // physical row that becomes the pivot row
        const int rowP = L.rowOf[k];
        wave_sync();
        if (lane == 0) { L.rowOf[k] = rowK; L.rowOf[piv] = rowP; }
        wave_sync();
        const double pivv = Gg[rowK * LD + k];

        // pivot row entries right of the diagonal (incl. RHS), one register per column chunk
        double u[BIG_CHUNKS];
        bool ub[BIG_CHUNKS];
#pragma unroll
        for (int c = 0; c < BIG_CHUNKS; ++c) {
            const int j = c * 64 + lane;
            ub[c] = (c * 64 <= N) && (j > k && j <= N) && bit_of(L, rowK, j);
            u[c] = ub[c] ? Gg[rowK * LD + j] : 0.0;
        }
        // ---- eliminate every other candidate row (rows are independent of each other, so the
        // order among them does not matter; each row applies its multipliers in ascending k)
        for (int t = 0; t < nCand; ++t) {
            const int r = L.candRow[t];
            if (r == rowK) continue;
            const double f = L.candVal[t] / pivv;                        // solver.hpp:71
#pragma unroll
            for (int c = 0; c < BIG_CHUNKS; ++c) {
                if (c * 64 > N) break;
                const int j = c * 64 + lane;
                if (ub[c]) {
                    const double old = bit_of(L, r, j) ? Gg[r * LD + j] : 0.0;
                    Gg[r * LD + j] = old - f * u[c];                      // :74 (+ RHS)
                }
                const unsigned long long nb = __ballot(ub[c]);
                if (lane == 0 && (uint32_t)nb) L.bm[r * L.W32 + 2 * c] |= (uint32_t)nb;
                if (lane == 32 && (uint32_t)(nb >> 32) && 2 * c + 1 < L.W32) L.bm[r * L.W32 + 2 * c + 1] |= (uint32_t)(nb >> 32);
            }
        }
        __threadfence_block();
        wave_sync();
    }

    if (pivLog) pivlog_commit(pivLog, N, failed, lane);
    if (failed) {
        flags |= CSIM_ST_LU_TINY_PIVOT;
        for (int i = lane; i < N; i += 64) L.xr[i] = 0.0;
        wave_sync();
        return;
    }

    // ---- back substitution, position i descending, columns ascending
    for (int i = N - 1; i >= 0; --i) {
        const int r = L.rowOf[i];
        double sum = Gg[r * LD + N];                                     // y_i  (uniform read)
        if (!bit_of(L, r, N)) sum = 0.0;
        for (int c = 0; c * 64 <= N - 1; ++c) {
            const int j = c * 64 + lane;
            const bool on = (j > i && j < N) && bit_of(L, r, j);
            const double prod = on ? Gg[r * LD + j] * L.xr[j] : 0.0;      // :119
            unsigned long long todo = __ballot(on && prod != 0.0);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                sum -= read_lane(prod, l);
            }
        }
        const double d = Gg[r * LD + i];
        double xi;
        if (fabs(d) < eps) { xi = 0.0; flags |= CSIM_ST_LU_ZERO_DIAG; }
        else xi = sum / d;
        wave_sync();
        if (lane == 0) L.xr[i] = xi;
        wave_sync();
    }
}

__device__ __forceinline__ double norm_big(const BigLds& L, double* scratch, int N, int lane)
{
    // scratch[i] already holds d_i; sum of squares in index order
    wave_sync();
    double ss = 0.0;
    for (int i = 0; i < N; ++i) { const double d = scratch[i]; ss += d * d; }
    wave_sync();
    return sqrt(ss);
}

__device__ __forceinline__ bool all_finite_big(const double* v, int N, int lane)
{
    bool bad = false;
    for (int i = lane; i < N; i += 64) bad = bad || !isfinite(v[i]);
    return __ballot(bad) == 0ull;
}

} // namespace

// ------------------------------------------------------------------ DC
__global__ void __launch_bounds__(64)
k_dc_big(GenPlan pl, const double* __restrict__ params, int B, double* __restrict__ scratch,
         double* __restrict__ xout, int32_t* __restrict__ iters, uint32_t* __restrict__ status,
         const uint8_t* __restrict__ only, int32_t* __restrict__ pivLog, int pivInstance)
{
    extern __shared__ unsigned char smraw[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (only && !only[b]) return;           // fallback / planner launches touch the flagged instances only
    int32_t* myPivLog = (pivLog && b == pivInstance) ? pivLog : nullptr;
    const int N = pl.N;
    const BigLds L = carve(smraw, N, pl.nTerms, pl.P);
    double* Gg = scratch + (size_t)b * N * pl.LD;
    const csim_consts& K = pl.k;

    for (int p = lane; p < pl.P; p += 64) L.Pv[p] = params[(int64_t)p * B + b];
    for (int t = lane; t < pl.nTerms; t += 64) L.T[t] = 0.0;
    for (int i = lane; i < N; i += 64) L.xs[i] = 0.0;
    wave_sync();
    terms_const<false>(pl, L.Pv, L.T, 0.0, lane);
    wave_sync();

    unsigned st = 0;
    int itTotal = 0;
    if (!pl.hasNonlinear) {
        terms_step_dc(pl, L.Pv, L.T, 1.0, lane);
        if (lane == 0) L.T[pl.termGmin] = 0.0;
        wave_sync();
        assemble_big(pl, L, Gg, lane);
        lu_solve_big(pl, L, Gg, K.lu_eps, lane, st, myPivLog);
        clear_big(pl, L, Gg, lane);
        for (int i = lane; i < N; i += 64) L.xs[i] = L.xr[i];
        itTotal = 1;
    } else {
        for (int step = 1; step <= K.dc_ramp_steps; ++step) {
            const double scale = (double)step / K.dc_ramp_steps;
            double gmin = base_gmin(K, scale);
            double prevErr = INFINITY;
            terms_step_dc(pl, L.Pv, L.T, scale, lane);
            wave_sync();
            for (int iter = 0; iter < K.dc_max_iters; ++iter) {
                terms_iter_mos(pl, L.Pv, L.T, L.xs, lane);
                if (lane == 0) L.T[pl.termGmin] = gmin;
                wave_sync();
                assemble_big(pl, L, Gg, lane);
                lu_solve_big(pl, L, Gg, K.lu_eps, lane, st, nullptr);
                clear_big(pl, L, Gg, lane);
                ++itTotal;
                if (!all_finite_big(L.xr, N, lane)) {
                    gmin = fmin(gmin * K.gmin_nonfinite_mul, K.gmin_nonfinite_cap);
                    st |= CSIM_ST_DC_NONFINITE;
                    continue;
                }
                const double alpha = clampd(K.dc_alpha, K.dc_alpha_min, K.dc_alpha_max);
                for (int i = lane; i < N; i += 64) {
                    const double xo = L.xs[i];
                    const double xn = xo + alpha * (L.xr[i] - xo);
                    L.xp[i] = xn - xo;              // xp doubles as the difference buffer in DC
                    L.xr[i] = xn;
                }
                const double err = norm_big(L, L.xp, N, lane);
                const double gb = base_gmin(K, scale);
                double gnext = gb;
                if (iter == 0 || !isfinite(prevErr)) gnext = gb;
                else if (err > prevErr * K.slow_ratio) gnext = fmin(gmin * 2.0, K.gmin_abs_max);
                else if (err < prevErr * K.fast_ratio) gnext = 0.5 * gmin + 0.5 * gb;
                else gnext = 0.7 * gmin + 0.3 * gb;
                for (int i = lane; i < N; i += 64) L.xs[i] = L.xr[i];
                wave_sync();
                gmin = gnext;
                prevErr = err;
                if (err < K.dc_tol) break;
                if (iter == K.dc_max_iters - 1) st |= CSIM_ST_DC_NONCONV;
            }
        }
    }
    wave_sync();
    for (int i = lane; i < N; i += 64) xout[(int64_t)i * B + b] = L.xs[i];
    if (lane == 0) { iters[b] = itTotal; status[b] = (only && !pivLog) ? (st | CSIM_ST_SCHED_FALLBACK_DC) : st; }
}

// ------------------------------------------------------------ transient
__global__ void __launch_bounds__(64)
k_tran_big(GenPlan pl, const double* __restrict__ params, int B, double dt, long long stepFirst, long long nSteps,
           const int32_t* __restrict__ probeEq, int nProbe, int outStride, double* __restrict__ wave,
           double* __restrict__ xio, long long* __restrict__ iters, uint32_t* __restrict__ status,
           int32_t* __restrict__ stepIters, const uint8_t* __restrict__ only, double* __restrict__ scratch,
           const int32_t* __restrict__ slotOf, int32_t* __restrict__ pivLog, int pivInstance,
           int32_t* __restrict__ done, int maxSteps)
{
    extern __shared__ unsigned char smraw[];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (only && !only[b]) return;
    const long long d0 = done ? (long long)done[b] : 0;        // hybrid stepping, see kernels_general.hip
    if (done && d0 >= nSteps) return;
    const long long sEnd = done ? (d0 + maxSteps < nSteps ? d0 + maxSteps : nSteps) : nSteps;
    const int N = pl.N;
    const BigLds L = carve(smraw, N, pl.nTerms, pl.P);
    // scratch matrices are handed out per running instance: slotOf maps instance -> slot
    const int slot = slotOf ? slotOf[b] : b;
    double* Gg = scratch + (size_t)slot * N * pl.LD;
    const csim_consts& K = pl.k;
    int32_t* myPivLog = (pivLog && b == pivInstance) ? pivLog : nullptr;

    for (int p = lane; p < pl.P; p += 64) L.Pv[p] = params[(int64_t)p * B + b];
    for (int t = lane; t < pl.nTerms; t += 64) L.T[t] = 0.0;
    for (int i = lane; i < N; i += 64) { const double v = xio[(int64_t)i * B + b]; L.xs[i] = v; L.xp[i] = v; }
    wave_sync();
    terms_const<true>(pl, L.Pv, L.T, dt, lane);
    if (lane == 0) L.T[pl.termGmin] = K.tran_gmin;
    wave_sync();

    if (stepFirst == 0 && d0 == 0 && wave)
        for (int q = lane; q < nProbe; q += 64) wave[((int64_t)0 * nProbe + q) * B + b] = L.xs[probeEq[q]];

    unsigned st = (status[b] & CSIM_ST_TRAN_NONFINITE);
    if (done) st |= CSIM_ST_SCHED_FALLBACK;
    long long itTotal = 0;
    bool aborted = (st & CSIM_ST_TRAN_NONFINITE) != 0;

    for (long long s = d0 + 1; s <= sEnd && !aborted; ++s) {
        const long long gstep = stepFirst + s;
        const double tNow = (double)(int)gstep * dt;
        terms_step_tran(pl, L.Pv, L.T, L.xp, tNow, lane);
        wave_sync();
        int it = 0;
        for (int iter = 0; iter < K.tran_max_iters; ++iter) {
            terms_iter_mos(pl, L.Pv, L.T, L.xs, lane);
            wave_sync();
            assemble_big(pl, L, Gg, lane);
            lu_solve_big(pl, L, Gg, K.lu_eps, lane, st, myPivLog);
            clear_big(pl, L, Gg, lane);
            ++it;
            if (!all_finite_big(L.xr, N, lane)) { st |= CSIM_ST_TRAN_NONFINITE; aborted = true; break; }
            // damped update; the differences go through T's tail? no: reuse xr as xn and keep d in registers
            double dsave[BIG_CHUNKS];
#pragma unroll
            for (int c = 0; c < BIG_CHUNKS; ++c) {
                const int i = c * 64 + lane;
                dsave[c] = 0.0;
                if (i < N) {
                    const double xo = L.xs[i];
                    const double xn = xo + K.tran_alpha * (L.xr[i] - xo);
                    dsave[c] = xn - xo;
                    L.xs[i] = xn;
                }
            }
            wave_sync();
#pragma unroll
            for (int c = 0; c < BIG_CHUNKS; ++c) { const int i = c * 64 + lane; if (i < N) L.xr[i] = dsave[c]; }
            const double err = norm_big(L, L.xr, N, lane);
            if (err < K.tran_tol) break;
            if (iter == K.tran_max_iters - 1) st |= CSIM_ST_TRAN_NONCONV;
        }
        itTotal += it;
        if (stepIters && lane == 0) stepIters[(s - 1) * (int64_t)B + b] = it;
        if (aborted) break;
        for (int i = lane; i < N; i += 64) L.xp[i] = L.xs[i];
        wave_sync();
        if (wave && (gstep % outStride) == 0)
            for (int q = lane; q < nProbe; q += 64)
                wave[((gstep / outStride) * nProbe + q) * (int64_t)B + b] = L.xs[probeEq[q]];
    }

    for (int i = lane; i < N; i += 64) xio[(int64_t)i * B + b] = L.xs[i];
    if (lane == 0) {
        iters[b] += itTotal;
        status[b] |= st;
        if (done) done[b] = aborted ? (int32_t)nSteps : (int32_t)sEnd;
    }
}

// ------------------------------------------------------------------ launchers
size_t bigScratchBytesPerInstance(const GenPlan& pl) { return sizeof(double) * (size_t)pl.N * (size_t)pl.LD; }
int bigMaxUnknowns() { return BIG_MAX_N; }

// element terms, parameters, three N-vectors and the N x (N+1) bit matrix must fit one CU's LDS (160 KB)
bool bigSupports(int N, int nTerms, int P) { return N <= BIG_MAX_N && bigLdsBytes(N, nTerms, P) <= 160 * 1024; }

hipError_t launchDcBig(const GenPlan& pl, const double* dParams, int B, double* dScratch, double* dX,
                       int32_t* dIters, uint32_t* dStatus, hipStream_t stream, const uint8_t* dOnly, int32_t* dPivLog,
                       int pivInstance)
{
    const size_t lds = bigLdsBytes(pl.N, pl.nTerms, pl.P);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dc_big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dc_big, dim3(B), dim3(64), lds, stream, pl, dParams, B, dScratch, dX, dIters, dStatus, dOnly, dPivLog,
                       pivInstance);
    return hipGetLastError();
}

hipError_t launchTranBig(const GenPlan& pl, const double* dParams, int B, double dt, long long stepFirst,
                         long long nSteps, const int32_t* dProbeEq, int nProbe, int outStride, double* dWave,
                         double* dX, long long* dIters, uint32_t* dStatus, int32_t* dStepIters,
                         const uint8_t* dOnly, double* dScratch, const int32_t* dSlotOf, hipStream_t stream,
                         int32_t* dPivLog, int pivInstance, int32_t* dDone, int maxSteps)
{
    const size_t lds = bigLdsBytes(pl.N, pl.nTerms, pl.P);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tran_big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_tran_big, dim3(B), dim3(64), lds, stream, pl, dParams, B, dt, stepFirst, nSteps, dProbeEq,
                       nProbe, outStride, dWave, dX, dIters, dStatus, dStepIters, dOnly, dScratch, dSlotOf, dPivLog,
                       pivInstance, dDone, maxSteps);
    return hipGetLastError();
}

} // namespace csim
