// dev probe (GPU box): the linear kernels' quotient  s / p  formed from p and r = (1.0 / p, correctly rounded by the
// division operator) as  q0 = s * r;  e = fma(-p, q0, s);  q = fma(e, r, q0)  (codegen_linear.cpp emitQuotient),
// against the `/` operator on the device and on the host, bit for bit.
//     hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/dev/ubench/div_markstein.hip -o /tmp/divm && /tmp/divm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const double* s, const double* p, double* qm, double* qd, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double r = 1.0 / p[i];
    const double q0 = s[i] * r;
    const double e = fma(-p[i], q0, s[i]);
    qm[i] = fma(e, r, q0);
    qd[i] = s[i] / p[i];
}
int main()
{
    const int n = 1 << 24;
    std::vector<double> s(n), p(n), qm(n), qd(n);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&st]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (int i = 0; i < n; ++i) {
        // random significands (incl. all-ones / all-zeros patterns now and then), exponents within +-300
        auto mk = [&](int spread) {
            unsigned long long m = rnd() & 0xFFFFFFFFFFFFFull;
            const unsigned sel = rnd() % 16;
            if (sel == 0) m = 0xFFFFFFFFFFFFFull; else if (sel == 1) m = 0; else if (sel == 2) m = 0xFFFFFFFFFFFFEull; else if (sel == 3) m = 1;
            const long long ex = 1023 + (long long)(rnd() % (2 * spread + 1)) - spread;
            unsigned long long bits = ((rnd() & 1ull) << 63) | ((unsigned long long)ex << 52) | m;
            double d; std::memcpy(&d, &bits, 8); return d;
        };
        s[i] = mk(300); p[i] = mk(300);
    }
    double *ds, *dp, *dm, *dd;
    hipMalloc(&ds, n * 8); hipMalloc(&dp, n * 8); hipMalloc(&dm, n * 8); hipMalloc(&dd, n * 8);
    hipMemcpy(ds, s.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(ds, dp, dm, dd, n);
    hipMemcpy(qm.data(), dm, n * 8, hipMemcpyDeviceToHost); hipMemcpy(qd.data(), dd, n * 8, hipMemcpyDeviceToHost);
    long bad = 0, badDev = 0;
    for (int i = 0; i < n; ++i) {
        const double host = s[i] / p[i];
        if (std::memcmp(&qm[i], &host, 8) != 0 && bad++ < 5) std::printf("  s=%a p=%a  quotient %a  host %a\n", s[i], p[i], qm[i], host);
        badDev += std::memcmp(&qd[i], &host, 8) != 0;
    }
    std::printf("%d operand pairs: residual-corrected quotient != host division in %ld; device `/` != host in %ld\n", n, bad, badDev);
    return bad != 0;
}
