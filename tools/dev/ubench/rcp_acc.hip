// dev probe (GPU box): accuracy of v_rcp_f64 and of the refinements built on it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstdlib>
__global__ void k(const double* x, double* r0, double* r2, double* rc, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    double q = fma(fma(-a, r, 1.0), r, r);
    q = fma(fma(-a, q, 1.0), q, q);
    r2[i] = q;
    const double e = fma(-a, r, 1.0);
    rc[i] = fma(fma(e, e, e), r, r);          // one cubic step
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    srand(1);
    for (int i = 0; i < n; ++i) { const double m = 1.0 + rand() / (double)RAND_MAX; x[i] = std::ldexp(m, rand() % 80 - 40) * ((rand() & 1) ? 1 : -1); }
    double *dx, *da, *db, *dc;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, dc, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e2 = 0, ec = 0; long ne2 = 0, nec = 0, ndiff = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        const double exact = 1.0 / x[i];
        e0 = std::fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
        e2 = std::fmax(e2, (double)fabsl(((long double)b[i] - t) / t));
        ec = std::fmax(ec, (double)fabsl(((long double)c[i] - t) / t));
        ne2 += b[i] != exact; nec += c[i] != exact; ndiff += b[i] != c[i];
    }
    // special operands through the cubic-step reciprocal the generated kernels use (rcp_nr / grp_rcp_nr): what a pivot of
    // that kind turns into, against the true quotient 1 / a
    {
        const double sp[12] = {INFINITY, -INFINITY, 0.0, -0.0, 4.9406564584124654e-324, 2.2250738585072014e-308, -2.2250738585072014e-308,
                               1e-310, 1.7976931348623157e308, 8.98846567431158e307, 1e-300, NAN};
        hipMemcpy(dx, sp, sizeof sp, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dx, da, db, dc, 12);
        double out[12];
        hipMemcpy(out, dc, sizeof out, hipMemcpyDeviceToHost);
        for (int i = 0; i < 12; ++i) printf("  rcp_nr(%-24.17g) = %-24.17g   1/a = %.17g\n", sp[i], out[i], 1.0 / sp[i]);
    }
    printf("max rel err: v_rcp_f64 %.3g (2^%.1f), two quadratic steps %.3g, one cubic step %.3g; not correctly rounded: %ld / %ld of %d; quadratic != cubic in %ld\n",
           e0, std::log2(e0), e2, ec, ne2, nec, n, ndiff);
    return 0;
}
