// mc.hip -- K3: Monte-Carlo parameter table on the device, plus the small
// layout-transpose kernel used by the host-pointer entry points.
//
// One thread per table entry params[p][b] (b fastest: coalesced stores).  The
// draw is a pure function of (seed, global instance index, slot) -- see
// mc_draw.h -- so a rank can regenerate exactly its shard with no scatter.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "mc_draw.h"

namespace csim {

__global__ void __launch_bounds__(256)
k_mc_params(int P, int B, long long bFirst, uint64_t seed, double sigma,
            const int32_t* __restrict__ kind, const double* __restrict__ nominal,
            const double* __restrict__ mu, const double* __restrict__ cox,
            const double* __restrict__ w, const double* __restrict__ l,
            double* __restrict__ params)
{
    const long long total = (long long)P * B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i / B);
        const int b = (int)(i % B);
        const int kd = kind[p];
        const double z = kd ? csim_mc::draw_z(seed, (uint64_t)(bFirst + b), (uint64_t)p) : 0.0;
        params[i] = csim_mc::perturb(kd, nominal[p], mu[p], cox[p], w[p], l[p], sigma, z);
    }
}

hipError_t launchMcParams(int P, int B, long long bFirst, uint64_t seed, double sigma,
                          const int32_t* dKind, const double* dNominal, const double* dMu,
                          const double* dCox, const double* dW, const double* dL,
                          double* dParams, hipStream_t stream)
{
    const long long total = (long long)P * B;
    if (total == 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mc_params, dim3((unsigned)blocks), dim3(256), 0, stream, P, B, bFirst, seed, sigma,
                       dKind, dNominal, dMu, dCox, dW, dL, dParams);
    return hipGetLastError();
}

// out[c][r] = in[r][c]; 32x32 tiles through LDS (33-word pitch)
__global__ void __launch_bounds__(256)
k_transpose(const double* __restrict__ in, double* __restrict__ out, int rows, int cols)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = in[(int64_t)r * cols + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) out[(int64_t)c * rows + r] = tile[tx][k];
    }
}

hipError_t launchTranspose(const double* dIn, double* dOut, int rows, int cols, hipStream_t stream)
{
    if (rows == 0 || cols == 0) return hipSuccess;
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, stream, dIn, dOut, rows, cols);
    return hipGetLastError();
}

} // namespace csim
