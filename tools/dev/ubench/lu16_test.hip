// dev test (GPU box): the register LU of kernels_packed.hip against a plain host LU on random small systems.
// hipcc --offload-arch=gfx950 -O3 -I include -I circuitsimulator_amd/csrc/engine -I circuitsimulator_amd/csrc/api lu16_test.hip
#include "../../../circuitsimulator_amd/csrc/engine/kernels_packed.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
#include <cstdlib>

namespace csim {
template <int S>
__global__ void k_test(const double* A, int N, int nSys, double* X, unsigned* F, int* P)
{
    __shared__ double gs[4][32 * 32 + 1];
    __shared__ double rs[4][32];
    __shared__ int rowMap[32 * 33];
    __shared__ int pv[4 * 32];
    const int g = threadIdx.x % 16, q = threadIdx.x / 16;
    const int sys = blockIdx.x * 4 + q;
    const int LD = N + 1;
    for (int i = threadIdx.x; i < N * LD; i += 64) rowMap[i] = (i % LD < N) ? (i / LD) * N + i % LD : N * N;
    if (sys < nSys) {
        for (int i = g; i < N * N; i += 16) gs[q][i] = A[(size_t)sys * N * LD + (i / N) * LD + i % N];
        for (int i = g; i < N; i += 16) rs[q][i] = A[(size_t)sys * N * LD + i * LD + N];
    } else {
        for (int i = g; i < N * N; i += 16) gs[q][i] = (i / N == i % N) ? 1.0 : 0.0;
        for (int i = g; i < N; i += 16) rs[q][i] = 0.0;
    }
    if (g == 0) gs[q][N * N] = 0.0;
    __syncthreads();
    unsigned fl = 0;
    double x[S];
    lu_solve_dispatch(gs[q], rs[q], rowMap, N * N, N, LD, 1e-15, g, q, true, fl, pv + q * 32, x);
    __syncthreads();
    if (sys < nSys)
        for (int s = 0; s < S; ++s)
            if (16 * s + g < N) { X[(size_t)sys * N + 16 * s + g] = x[s]; P[(size_t)sys * N + 16 * s + g] = pv[q * 32 + 16 * s + g]; }
    if (sys < nSys && g == 0) F[sys] = fl;
}
}
static bool hostSolve(int n, const double* Ab, double* x, int* piv)
{
    std::vector<double> a(Ab, Ab + n * (n + 1));
    const int LD = n + 1;
    for (int k = 0; k < n; ++k) {
        int p = k; double m = std::fabs(a[k * LD + k]);
        for (int i = k + 1; i < n; ++i) { const double v = std::fabs(a[i * LD + k]); if (v > m) { m = v; p = i; } }
        if (m < 1e-15) return false;
        piv[k] = p;
        if (p != k) for (int j = 0; j <= n; ++j) std::swap(a[k * LD + j], a[p * LD + j]);
        for (int i = k + 1; i < n; ++i) {
            const double f = a[i * LD + k] / a[k * LD + k];
            for (int j = k + 1; j <= n; ++j) { volatile double t = f * a[k * LD + j]; a[i * LD + j] -= t; }
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = a[i * LD + n];
        for (int j = i + 1; j < n; ++j) { volatile double t = a[i * LD + j] * x[j]; s -= t; }
        x[i] = s / a[i * LD + i];
    }
    return true;
}
int main()
{
    int bad = 0;
    for (int N : {1, 2, 5, 13, 15, 16, 17, 18, 25, 31, 32}) {
        const int nSys = 203, LD = N + 1;
        std::vector<double> A((size_t)nSys * N * LD);
        srand(7 + N);
        for (auto& v : A) { const int r = rand() % 10; v = r < 4 ? 0.0 : (r < 6 ? 1.0 : (rand() / (double)RAND_MAX - 0.5)); }
        for (int s = 0; s < nSys; ++s) for (int i = 0; i < N; ++i) if (rand() % 3) A[(size_t)s * N * LD + i * LD + i] += 2.0;
        double *dA, *dX; unsigned* dF; int* dP;
        hipMalloc(&dA, A.size() * 8); hipMalloc(&dX, (size_t)nSys * N * 8); hipMalloc(&dF, nSys * 4); hipMalloc(&dP, (size_t)nSys * N * 4);
        hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
        if (N <= 16) csim::k_test<1><<<(nSys + 3) / 4, 64>>>(dA, N, nSys, dX, dF, dP); else csim::k_test<2><<<(nSys + 3) / 4, 64>>>(dA, N, nSys, dX, dF, dP);
        std::vector<double> X((size_t)nSys * N); std::vector<unsigned> F(nSys); std::vector<int> P((size_t)nSys * N);
        hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(F.data(), dF, nSys * 4, hipMemcpyDeviceToHost);
        hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost);
        int nb = 0, nfail = 0;
        for (int s = 0; s < nSys; ++s) {
            std::vector<double> x(N, 0.0); std::vector<int> pv(N, -1);
            const bool ok = hostSolve(N, &A[(size_t)s * N * LD], x.data(), pv.data());
            if (!ok) { ++nfail; if (!(F[s] & 1u) && !(F[s])) { if (nb++ < 3) printf("N %d sys %d: host fails, device flags %x\n", N, s, F[s]); } continue; }
            bool same = std::memcmp(x.data(), &X[(size_t)s * N], N * 8) == 0;
            bool samep = true; for (int k = 0; k < N; ++k) samep = samep && pv[k] == P[(size_t)s * N + k];
            if (!same || !samep) { if (nb++ < 3) { printf("N %d sys %d differs (pivots %s): host x0 %.17g dev x0 %.17g flags %x\n", N, s, samep ? "same" : "DIFFER", x[0], X[(size_t)s * N], F[s]);
                for (int k = 0; k < N; ++k) printf(" %d/%d", pv[k], P[(size_t)s * N + k]); printf("\n"); } }
        }
        printf("N %2d: %d systems, %d singular on host, %d mismatches\n", N, nSys, nfail, nb);
        bad += nb;
        hipFree(dA); hipFree(dX); hipFree(dF); hipFree(dP);
    }
    return bad ? 1 : 0;
}
