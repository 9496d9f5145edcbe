// mc_draw.h -- counter-based Monte-Carlo draw shared by the device generator
// (mc.hip) and its host mirror (csim_mc_params_host).
//
// The reference has no Monte-Carlo axis; the batch axis and
// its distribution are this build's addition, frozen here:
//   instance 0 is the nominal circuit; for instance b > 0 every R, C, L value
//   and every MOSFET's VT and MU is multiplied by (1 + sigma*z),
//   z ~ N(0,1) clipped to +-3, z = z(seed, b, slot); K is rebuilt as
//   (MU')*COX*(W/L) in the association of the reference's
//   Circuit::addMosfet (src/circuit.cpp:144).
//
// Everything below uses only IEEE-754 basic operations (+ - * / sqrt, frexp)
// in a fixed order with contraction disabled, so host and device produce the
// same bits and any shard can regenerate any instance.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define CSIM_HD __host__ __device__
#else
#define CSIM_HD
#endif

namespace csim_mc {

CSIM_HD inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// uniform in (0,1), 53 bits
CSIM_HD inline double uniform01(uint64_t seed, uint64_t instance, uint64_t slot)
{
    const uint64_t h = mix64(mix64(mix64(seed) ^ instance) ^ slot);
    return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

// natural log from basic operations only: x = m*2^e, m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^23
CSIM_HD inline double det_log(double x)
{
#pragma clang fp contract(off)
    int e = 0;
    double m = frexp(x, &e);            // m in [0.5, 1)
    if (m < 0.70710678118654752440) { m = m * 2.0; e = e - 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double acc = 1.0 / 23.0;
    acc = acc * s2 + 1.0 / 21.0;
    acc = acc * s2 + 1.0 / 19.0;
    acc = acc * s2 + 1.0 / 17.0;
    acc = acc * s2 + 1.0 / 15.0;
    acc = acc * s2 + 1.0 / 13.0;
    acc = acc * s2 + 1.0 / 11.0;
    acc = acc * s2 + 1.0 / 9.0;
    acc = acc * s2 + 1.0 / 7.0;
    acc = acc * s2 + 1.0 / 5.0;
    acc = acc * s2 + 1.0 / 3.0;
    acc = acc * s2 + 1.0;
    return (double)e * 0.69314718055994530942 + 2.0 * s * acc;
}

// inverse standard-normal CDF, P. J. Acklam's rational approximation
// (relative error < 1.2e-9), evaluated in a fixed order
CSIM_HD inline double inv_norm_cdf(double p)
{
#pragma clang fp contract(off)
    const double a0 = -3.969683028665376e+01, a1 = 2.209460984245205e+02, a2 = -2.759285104469687e+02,
                 a3 = 1.383577518672690e+02, a4 = -3.066479806614716e+01, a5 = 2.506628277459239e+00;
    const double b0 = -5.447609879822406e+01, b1 = 1.615858368580409e+02, b2 = -1.556989798598866e+02,
                 b3 = 6.680131188771972e+01, b4 = -1.328068155288572e+01;
    const double c0 = -7.784894002430293e-03, c1 = -3.223964580411365e-01, c2 = -2.400758277161838e+00,
                 c3 = -2.549732539343734e+00, c4 = 4.374664141464968e+00, c5 = 2.938163982698783e+00;
    const double d0 = 7.784695709041462e-03, d1 = 3.224671290700398e-01, d2 = 2.445134137142996e+00,
                 d3 = 3.754408661907416e+00;
    const double plow = 0.02425, phigh = 1.0 - 0.02425;
    if (p < plow) {
        const double q = sqrt(-2.0 * det_log(p));
        return (((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) /
               ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
    }
    if (p <= phigh) {
        const double q = p - 0.5;
        const double r = q * q;
        return (((((a0 * r + a1) * r + a2) * r + a3) * r + a4) * r + a5) * q /
               (((((b0 * r + b1) * r + b2) * r + b3) * r + b4) * r + 1.0);
    }
    const double q = sqrt(-2.0 * det_log(1.0 - p));
    return -(((((c0 * q + c1) * q + c2) * q + c3) * q + c4) * q + c5) /
            ((((d0 * q + d1) * q + d2) * q + d3) * q + 1.0);
}

// clipped standard-normal draw for (seed, instance, slot)
CSIM_HD inline double draw_z(uint64_t seed, uint64_t instance, uint64_t slot)
{
    if (instance == 0) return 0.0;
    double z = inv_norm_cdf(uniform01(seed, instance, slot));
    if (z > 3.0) z = 3.0;
    if (z < -3.0) z = -3.0;
    return z;
}

// perturbed value of one parameter slot
//   kind 0: nominal; 1: nominal*(1+sigma z); 2: ((mu*(1+sigma z))*cox)*(w/l)
CSIM_HD inline double perturb(int kind, double nominal, double mu, double cox, double w, double l,
                              double sigma, double z)
{
#pragma clang fp contract(off)
    if (kind == 1) return nominal * (1.0 + sigma * z);
    if (kind == 2) {
        if (z == 0.0) return nominal;
        const double mu2 = mu * (1.0 + sigma * z);
        return mu2 * cox * (w / l);
    }
    return nominal;
}

} // namespace csim_mc
