// kernels.hpp -- host-callable launchers of the HIP kernels (library-private).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"

namespace csim {

// wave-per-instance kernels (kernels_general.hip)
hipError_t launchDcGeneral(const GenPlan& pl, const double* dParams, int B, double* dX,
                           int32_t* dIters, uint32_t* dStatus, hipStream_t stream, const uint8_t* dOnly = nullptr,
                           int32_t* dPivLog = nullptr, int pivInstance = 0);
hipError_t launchTranGeneral(const GenPlan& pl, const double* dParams, int B, double dt,
                             long long stepFirst, long long nSteps, const int32_t* dProbeEq, int nProbe,
                             int outStride, double* dWave, double* dX, long long* dIters,
                             uint32_t* dStatus, int32_t* dStepIters, const uint8_t* dOnly,
                             hipStream_t stream, int32_t* dPivLog = nullptr, int pivInstance = -1,
                             int32_t* dDone = nullptr, int maxSteps = 0,
                             const int32_t* dKnownAlts = nullptr, int nKnown = 0);
hipError_t launchLuSolve(int n, int B, const double* dA, const double* dRhs, double* dX,
                         uint32_t* dFlags, double eps, hipStream_t stream);
hipError_t launchLuFactor(int n, int B, const double* dA, double* dLU, int32_t* dPerm, uint32_t* dFlags,
                          double eps, hipStream_t stream);
size_t generalLdsBytes(const GenPlan& pl);

// the same kernels with 4 instances per wavefront and the matrix in registers for N <= 32 (kernels_packed.hip);
// the launchers above use them whenever no pivot log is asked for
int packedLanesFor(const GenPlan& pl);
hipError_t launchLuSolvePacked(int n, int B, const double* dA, const double* dRhs, double* dX, uint32_t* dFlags, double eps,
                               hipStream_t stream);
hipError_t launchDcPacked(const GenPlan& pl, const double* dParams, int B, double* dX, int32_t* dIters,
                          uint32_t* dStatus, hipStream_t stream, const uint8_t* dOnly);
hipError_t launchTranPacked(const GenPlan& pl, const double* dParams, int B, double dt, long long stepFirst,
                            long long nSteps, const int32_t* dProbeEq, int nProbe, int outStride, double* dWave,
                            double* dX, long long* dIters, uint32_t* dStatus, int32_t* dStepIters, const uint8_t* dOnly,
                            hipStream_t stream, int32_t* dDone, int maxSteps, const int32_t* dKnownAlts, int nKnown);

// dense stand-alone LU for 64 <= n <= 1024, one workgroup per system, in place (kernels_dense.hip)
hipError_t launchLuSolveDense(int n, int B, double* dWork, const double* dRhs, double* dX, uint32_t* dFlags,
                              double eps, hipStream_t stream);
hipError_t launchLuFactorDense(int n, int B, double* dLU, int32_t* dPerm, uint32_t* dFlags, double eps,
                               hipStream_t stream);

// wave-per-instance kernels for 64 <= N <= 320 (kernels_big.hip)
size_t bigScratchBytesPerInstance(const GenPlan& pl);
int bigMaxUnknowns();
bool bigSupports(int N, int nTerms, int P);
hipError_t launchDcBig(const GenPlan& pl, const double* dParams, int B, double* dScratch, double* dX,
                       int32_t* dIters, uint32_t* dStatus, hipStream_t stream, const uint8_t* dOnly = nullptr,
                       int32_t* dPivLog = nullptr, int pivInstance = 0);
hipError_t launchTranBig(const GenPlan& pl, const double* dParams, int B, double dt, long long stepFirst,
                         long long nSteps, const int32_t* dProbeEq, int nProbe, int outStride, double* dWave,
                         double* dX, long long* dIters, uint32_t* dStatus, int32_t* dStepIters,
                         const uint8_t* dOnly, double* dScratch, const int32_t* dSlotOf, hipStream_t stream,
                         int32_t* dPivLog = nullptr, int pivInstance = -1, int32_t* dDone = nullptr, int maxSteps = 0);

// Gauss-Seidel solver and the DC operating point built on it (kernels_gs.hip)
hipError_t launchGsSolve(int n, int B, const double* dAt /*[n*n][B]*/, const double* dRhs /*[n][B]*/,
                         const double* dX0 /*[n][B] or null*/, int maxIters, double tol, double* dX /*[n][B]*/,
                         double* dXold /*[n][B] scratch*/, int32_t* dSweeps, hipStream_t stream);
hipError_t launchDcGs(const GenPlan& pl, const int32_t* dRowPtr, const int32_t* dRowCol, const double* dParams, int B,
                      double* dX, int32_t* dIters, uint32_t* dStatus, hipStream_t stream);

// near-threshold verification of the fast generated kernels (kernels_verify.hip)
hipError_t launchNearPrep(int B, int N, long long nSteps, const int32_t* dNearStep, const double* dNearX, double* dVerX,
                          int32_t* dVerDone, long long* dVerIters, uint32_t* dVerStatus, unsigned char* dVerFallback,
                          hipStream_t stream);
hipError_t launchNearResolve(int B, int N, int32_t* dNearStep, const int32_t* dNearIt, const long long* dNearItAfter,
                             const double* dNearX, const int32_t* dVerDone, const long long* dVerIters, double* dX,
                             int32_t* dDone, long long* dIters, unsigned char* dFallback, int32_t* dFlags,
                             bool forceMismatch, hipStream_t stream);

// Monte-Carlo parameter table (mc.hip)
hipError_t launchMcParams(int P, int B, long long bFirst, uint64_t seed, double sigma,
                          const int32_t* dKind, const double* dNominal, const double* dMu,
                          const double* dCox, const double* dW, const double* dL,
                          double* dParams, hipStream_t stream);

// layout helpers (transpose.hip): [rows][cols] <-> [cols][rows] of doubles
hipError_t launchTranspose(const double* dIn, double* dOut, int rows, int cols, hipStream_t stream);

} // namespace csim
