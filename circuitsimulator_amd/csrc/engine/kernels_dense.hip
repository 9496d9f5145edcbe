// kernels_dense.hip -- Solver::luDecompose / Solver::solveLinearSystemLU for DENSE systems
// with 64 <= n <= 1024 (reference include/solver.hpp:30-131), one workgroup per system.
//
// The small kernels (kernels_general.hip) keep the whole matrix in LDS (n <= 63) and the
// circuit kernels for large N (kernels_big.hip) rely on the circuit's sparsity (at most 64
// rows per elimination column).  A caller of the stand-alone Solver API may pass any dense
// matrix (BASELINE config [3]: n = 257 -> 528 KB), so this variant works in place on the
// matrix in global memory:
//   * pivot rule: first row attaining the column maximum (solver.hpp:48-56) = maximum value,
//     lowest row index, NaN entries never selected, a NaN diagonal never replaced;
//   * rows are swapped physically, like the reference's row(k).swap(row(p)) (:63-66);
//   * multipliers a(i,k) / a(k,k) stored in place (:71), every update separately rounded
//     (device_common.hpp turns FP contraction off);
//   * forward substitution fused: the right-hand side rides along as y (LDS) -- per row the
//     same subtractions in the same ascending-k order as solver.hpp:108-113;
//   * back substitution row-wise, products subtracted in ascending j (:116-128).
// 256 threads: the four waves take rows k+1+w, k+1+w+4, ... of an elimination step, lanes take
// columns.  Throughput is not the point (this is API coverage, not the hot path).
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace csim {

namespace {

template <bool SOLVE>
__global__ void __launch_bounds__(256)
k_lu_dense(int n, double* __restrict__ LUall, const double* __restrict__ rhs, double* __restrict__ xout,
           int32_t* __restrict__ permOut, uint32_t* __restrict__ flags, double eps)
{
    extern __shared__ double dyn[];
    __shared__ double redV[4];
    __shared__ int redI[4];
    __shared__ int sPiv, sFail;
    double* y = dyn;                                        // [n]
    double* xs = dyn + n;                                   // [n]
    int32_t* perm = reinterpret_cast<int32_t*>(dyn + 2 * n);   // [n]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    double* LU = LUall + (int64_t)b * n * n;

    for (int i = tid; i < n; i += 256) {
        perm[i] = i;
        y[i] = SOLVE ? rhs[(int64_t)b * n + i] : 0.0;
        xs[i] = 0.0;
    }
    __syncthreads();

    bool failed = false;
    for (int k = 0; k < n; ++k) {
        // ---- pivot search (solver.hpp:48-56)
        const double akk = fabs(LU[(int64_t)k * n + k]);
        double bv = -1.0;
        int bi = 0x7fffffff;
        for (int i = k + 1 + tid; i < n; i += 256) {
            const double av = fabs(LU[(int64_t)i * n + k]);
            if (av > bv) { bv = av; bi = i; }               // ascending i per thread: first maximum
        }
        for (int m = 32; m >= 1; m >>= 1) {
            const double ov = __shfl_xor(bv, m);
            const int oi = __shfl_xor(bi, m);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { redV[wave] = bv; redI[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double mv = redV[0];
            int mi = redI[0];
            for (int w = 1; w < 4; ++w)
                if (redV[w] > mv || (redV[w] == mv && redI[w] < mi)) { mv = redV[w]; mi = redI[w]; }
            double maxAbs = akk;
            int piv = k;
            if (!(akk != akk) && mv > akk) { maxAbs = mv; piv = mi; }   // a NaN diagonal is never replaced
            sPiv = piv;
            sFail = (maxAbs < eps) ? 1 : 0;                              // :58-61
        }
        __syncthreads();
        if (sFail) { failed = true; break; }
        const int piv = sPiv;
        if (piv != k) {                                                  // :63-66
            for (int j = tid; j < n; j += 256) {
                const double t = LU[(int64_t)k * n + j];
                LU[(int64_t)k * n + j] = LU[(int64_t)piv * n + j];
                LU[(int64_t)piv * n + j] = t;
            }
            if (tid == 0) {
                const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
                const double ty = y[k]; y[k] = y[piv]; y[piv] = ty;
            }
        }
        __threadfence_block();
        __syncthreads();

        // ---- elimination (:68-76) with the right-hand side carried along
        const double pivv = LU[(int64_t)k * n + k];
        const double yk = y[k];
        for (int i = k + 1 + wave; i < n; i += 4) {
            double* row = LU + (int64_t)i * n;
            const double f = row[k] / pivv;                              // :71
            for (int j = k + 1 + lane; j < n; j += 64)
                row[j] = row[j] - f * LU[(int64_t)k * n + j];            // :74
            if (lane == 0) {
                row[k] = f;
                if (SOLVE) y[i] = y[i] - f * yk;                         // :108-113, same order
            }
        }
        __threadfence_block();
        __syncthreads();
    }

    if (failed) {
        if (tid == 0 && flags) flags[b] = CSIM_ST_LU_TINY_PIVOT;
        if (SOLVE) for (int i = tid; i < n; i += 256) xout[(int64_t)b * n + i] = 0.0;   // :94-97
        return;
    }
    if (permOut) for (int i = tid; i < n; i += 256) permOut[(int64_t)b * n + i] = perm[i];

    unsigned st = 0;
    if (SOLVE) {
        // ---- back substitution (:116-128): one wave, products of a row computed lane-parallel and
        // subtracted in ascending j (zero products skipped: x - 0 == x)
        if (wave == 0) {
            for (int i = n - 1; i >= 0; --i) {
                const double* row = LU + (int64_t)i * n;
                double sum = y[i];
                for (int base = i + 1; base < n; base += 64) {
                    const int j = base + lane;
                    const double prod = (j < n) ? row[j] * xs[j] : 0.0;
                    unsigned long long todo = __ballot(j < n && prod != 0.0);
                    while (todo) {
                        const int l = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        sum = sum - read_lane(prod, l);
                    }
                }
                const double d = row[i];
                double xi;
                if (fabs(d) < eps) { xi = 0.0; st |= CSIM_ST_LU_ZERO_DIAG; }        // :122-125
                else xi = sum / d;
                // one wave only: LDS operations of a wave complete in issue order; the fences keep the
                // compiler from moving the store across the reads (wave_sync() is a workgroup barrier
                // here -- four waves -- and must not be used inside this branch)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (lane == 0) xs[i] = xi;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) xout[(int64_t)b * n + i] = xs[i];
    }
    if (tid == 0 && flags) flags[b] = st;
}

size_t denseLds(int n) { return sizeof(double) * 2 * (size_t)n + sizeof(int32_t) * (size_t)n + 16; }

} // namespace

// dWork: [B][n][n] copy of A, overwritten by its factors
hipError_t launchLuSolveDense(int n, int B, double* dWork, const double* dRhs, double* dX, uint32_t* dFlags,
                              double eps, hipStream_t stream)
{
    hipLaunchKernelGGL(k_lu_dense<true>, dim3(B), dim3(256), denseLds(n), stream, n, dWork, dRhs, dX,
                       static_cast<int32_t*>(nullptr), dFlags, eps);
    return hipGetLastError();
}

// dLU: [B][n][n] copy of A on entry, LU on return
hipError_t launchLuFactorDense(int n, int B, double* dLU, int32_t* dPerm, uint32_t* dFlags, double eps,
                               hipStream_t stream)
{
    hipLaunchKernelGGL(k_lu_dense<false>, dim3(B), dim3(256), denseLds(n), stream, n, dLU,
                       static_cast<const double*>(nullptr), static_cast<double*>(nullptr), dPerm, dFlags, eps);
    return hipGetLastError();
}

} // namespace csim
