// kernels_packed_lu.hip -- batched stand-alone Solver::solveLinearSystemLU for n <= 32 on the register LU of
// the packed general kernels (packed_lu.hpp; design in kernels_packed.hip).  Its own translation unit: the 25
// size-specialised bodies compile next to, not after, those of the DC and transient kernels.
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace csim {

#pragma clang fp contract(off)

#include "packed_lu.hpp"

// ------------------------------------------------------ stand-alone LU solve, n <= 32
// Solver::solveLinearSystemLU for B dense systems (include/solver.hpp:83-131), four per wavefront: the same
// register LU, fed with a dense "non-zero list" (entry (r, c) at r n + c)
template <int S>
__global__ void __launch_bounds__(64)
k_lu_solve_packed(int n, int B, const double* __restrict__ A, const double* __restrict__ rhs, double* __restrict__ x,
                  uint32_t* __restrict__ flags, double eps)
{
    extern __shared__ double smp[];
    const int lane = threadIdx.x, g = lane % G16, q = lane / G16;
    const int bRaw = blockIdx.x * IPW + q;
    const bool exists = bRaw < B;
    const int64_t b = exists ? bRaw : B - 1;
    const int LD = n + 1, cells = n * n, per = cells + 1 + ((n + 1) & ~1);
    double* Gs = smp + (size_t)q * per;
    double* Rs = Gs + cells + 1;
    int32_t* rowMap = reinterpret_cast<int32_t*>(smp + (size_t)IPW * per);
    for (int i = lane; i < n * LD; i += 64) rowMap[i] = (i % LD < n) ? (i / LD) * n + i % LD : cells;
    for (int i = g; i < cells; i += G16) Gs[i] = A[b * cells + i];
    for (int i = g; i < n; i += G16) Rs[i] = rhs[b * n + i];
    if (g == 0) Gs[cells] = 0.0;
    wave_sync();
    unsigned st = 0;
    double xr[S];
    lu_solve_dispatch(Gs, Rs, rowMap, cells, n, LD, eps, g, q, exists, st, nullptr, xr);
    if (exists) {
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (16 * s + g < n) x[b * n + 16 * s + g] = xr[s];
        if (flags && g == 0) flags[b] = st;
    }
}

hipError_t launchLuSolvePacked(int n, int B, const double* dA, const double* dRhs, double* dX, uint32_t* dFlags, double eps,
                               hipStream_t stream)
{
    if (n < 1 || n > 32) return hipErrorInvalidValue;
    const size_t per = (size_t)n * n + 1 + ((n + 1) & ~1);
    const size_t lds = sizeof(double) * per * IPW + sizeof(int32_t) * (size_t)n * (n + 1);
    const dim3 grid((B + IPW - 1) / IPW);
    if (n <= 16) hipLaunchKernelGGL(k_lu_solve_packed<1>, grid, dim3(64), lds, stream, n, B, dA, dRhs, dX, dFlags, eps);
    else hipLaunchKernelGGL(k_lu_solve_packed<2>, grid, dim3(64), lds, stream, n, B, dA, dRhs, dX, dFlags, eps);
    return hipGetLastError();
}


} // namespace csim
