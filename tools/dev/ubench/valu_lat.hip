// dev microbenchmark (GPU box): issue cost and dependent latency of the FP64 operations the generated
// kernels are made of, one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_lat.hip -o valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define OUTER 8192
template <int MODE> __global__ void k(double* out, long long* cyc, double a, double b)
{
    double x0 = a + threadIdx.x, x1 = a * 2 + threadIdx.x, x2 = a * 3, x3 = a * 4, x4 = a * 5, x5 = a * 6, x6 = a * 7, x7 = a * 8;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int o = 0; o < OUTER; ++o)
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        if (MODE == 0) { x0 = fma(x0, b, a); }                                             // dependent FMA chain
        if (MODE == 1) { x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
                         x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a); }   // 8 independent
        if (MODE == 2) { x0 = __builtin_amdgcn_update_dpp(0.0, x0, 0x150 + 3, 0xF, 0xF, true); x0 = x0 + a; }   // bcast + add chain
        if (MODE == 3) { x0 = __builtin_amdgcn_rcp(x0) + a; }                               // rcp + add chain
        if (MODE == 4) { x0 = __builtin_amdgcn_update_dpp(0.0, x1, 0x150 + 3, 0xF, 0xF, true); x2 = __builtin_amdgcn_update_dpp(0.0, x3, 0x150 + 5, 0xF, 0xF, true);
                         x4 = __builtin_amdgcn_update_dpp(0.0, x5, 0x150 + 7, 0xF, 0xF, true); x6 = __builtin_amdgcn_update_dpp(0.0, x7, 0x150 + 9, 0xF, 0xF, true);
                         x1 += x0; x3 += x2; x5 += x4; x7 += x6; }                         // 4 independent bcast+add
        if (MODE == 5) { x0 = __builtin_amdgcn_rcp(x0); x1 = __builtin_amdgcn_rcp(x1); x2 = __builtin_amdgcn_rcp(x2); x3 = __builtin_amdgcn_rcp(x3); }   // independent rcp
        if (MODE == 6) { x0 = fmax(x0, x1 * b); }                                           // dependent max(mul)
        if (MODE == 7) { x0 = (threadIdx.x == 5) ? x1 : x0; x1 = x1 + a; }                   // select
        if (MODE == 8) { x0 = x0 * b; }                                                      // dependent mul chain
        if (MODE == 9) { x0 = x0 + b; }                                                      // dependent add chain
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* what, int opsPerRep)
{
    double* out; long long* cyc;
    hipMalloc(&out, 2048 * 64 * 8); hipMalloc(&cyc, 2048 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 1024, 2048}) {
        k<MODE><<<blocks, 64>>>(out, cyc, 1.000001, 0.999999);
        hipEventRecord(e0);
        k<MODE><<<blocks, 64>>>(out, cyc, 1.000001, 0.999999);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double nsPerRep = ms * 1e6 / ((double)OUTER * REP);
        printf("%-28s blocks %4d: %7.2f ns per rep (%d ops) = %6.2f cycles at 2.4 GHz per op\n", what, blocks, nsPerRep, opsPerRep, nsPerRep * 2.4 / opsPerRep);
    }
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0>("dependent fma", 1); run<1>("8 independent fma", 8); run<8>("dependent mul", 1); run<9>("dependent add", 1);
    run<2>("bcast(dpp)+add chain", 2); run<4>("4 independent bcast+add", 8);
    run<3>("rcp+add chain", 2); run<5>("4 independent rcp", 4); run<6>("max(mul) chain", 2); run<7>("select + add", 3);
    // wall-clock calibration of the counter: s_memtime ticks at a fixed 100 MHz on this family
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("clockRate %d kHz, wallClockRate? %d\n", p.clockRate, p.clockInstructionRate);
    return 0;
}
