// kernels_verify.hip -- bookkeeping kernels of the near-threshold verification.
//
// A fast generated kernel (FMA contraction, reciprocal pivots) that takes an `err < tol` decision
// (reference src/tanalisis.cpp:369) within its own rounding noise goes on speculatively and leaves, per
// instance, the state at the start of that time step (nearX), the step (nearStep), the pass count it took
// (nearIt) and the passes it counted from that step on (nearItAfter).  The engine then has the FAITHFUL
// generated kernel (the reference's arithmetic) redo exactly that step from the same state:
//   k_near_prep     sets up the faithful kernel's launch: flagged instances start at their step (done =
//                   step - 1, hand-over reason 2 = "one step only"), everything else is already finished;
//   k_near_resolve  compares pass counts.  Equal: the speculation stands.  Different (or the faithful kernel
//                   could not run the step): the instance is rolled back to the start of that step, its
//                   speculative passes are taken off the counter, and it is handed to the ladder with reason 2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace csim {

__global__ void k_near_prep(int B, int N, int nSteps, const int32_t* __restrict__ nearStep,
                            const double* __restrict__ nearX, double* __restrict__ verX, int32_t* __restrict__ verDone,
                            long long* __restrict__ verIters, uint32_t* __restrict__ verStatus,
                            unsigned char* __restrict__ verFallback)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int s = nearStep[b];
    verIters[b] = 0;
    verStatus[b] = 0u;
    if (s > 0) {
        verDone[b] = s - 1;
        verFallback[b] = 2;
        for (int i = 0; i < N; ++i) verX[(size_t)i * B + b] = nearX[(size_t)i * B + b];
    } else {
        verDone[b] = nSteps;
        verFallback[b] = 0;
    }
}

__global__ void k_near_resolve(int B, int N, int32_t* __restrict__ nearStep, const int32_t* __restrict__ nearIt,
                               const long long* __restrict__ nearItAfter, const double* __restrict__ nearX,
                               const int32_t* __restrict__ verDone, const long long* __restrict__ verIters,
                               double* __restrict__ x, int32_t* __restrict__ done, long long* __restrict__ iters,
                               unsigned char* __restrict__ fallback, int32_t* __restrict__ flags, int forceMismatch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int s = nearStep[b];
    if (s <= 0) return;
    nearStep[b] = 0;
    atomicAdd(&flags[2], 1);                                  // statistics: decisions verified
    if (!forceMismatch && verDone[b] == s && verIters[b] == (long long)nearIt[b]) return;     // same pass count: the speculation stands
    for (int i = 0; i < N; ++i) x[(size_t)i * B + b] = nearX[(size_t)i * B + b];
    done[b] = s - 1;
    iters[b] -= nearItAfter[b];
    fallback[b] = 2;
    flags[0] = 1;
    atomicAdd(&flags[3], 1);                                  // statistics: instances rolled back
}

hipError_t launchNearPrep(int B, int N, long long nSteps, const int32_t* dNearStep, const double* dNearX, double* dVerX,
                          int32_t* dVerDone, long long* dVerIters, uint32_t* dVerStatus, unsigned char* dVerFallback,
                          hipStream_t stream)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_near_prep, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, B, N, (int)nSteps, dNearStep,
                       dNearX, dVerX, dVerDone, dVerIters, dVerStatus, dVerFallback);
    return hipGetLastError();
}

hipError_t launchNearResolve(int B, int N, int32_t* dNearStep, const int32_t* dNearIt, const long long* dNearItAfter,
                             const double* dNearX, const int32_t* dVerDone, const long long* dVerIters, double* dX,
                             int32_t* dDone, long long* dIters, unsigned char* dFallback, int32_t* dFlags,
                             bool forceMismatch, hipStream_t stream)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_near_resolve, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, stream, B, N, dNearStep, dNearIt,
                       dNearItAfter, dNearX, dVerDone, dVerIters, dX, dDone, dIters, dFallback, dFlags, forceMismatch ? 1 : 0);
    return hipGetLastError();
}

} // namespace csim
