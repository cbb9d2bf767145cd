// pvalue.h — tail of compareSketches: Mash distance and p-value (host + device).
//
// distance: CommandDistance.cpp:387-407.  p-value: pValue, CommandDistance.cpp:427-448,
// which calls gsl_cdf_binomial_Q(x-1, r, n) (or Boost's binomial complement CDF) — a
// third-party routine not in the reference tree.  Both equal the regularized incomplete
// beta I_r(x, n-x+1); evaluated here with the Lentz continued fraction in log space so
// values down to 1e-300 keep full relative precision.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MG_PV_HD __host__ __device__ inline
#else
#define MG_PV_HD inline
#endif

namespace mg {

MG_PV_HD double beta_cf(double a, double b, double x)
{
    const double tiny = 1e-300, eps = 1e-16;
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    double c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 100000; m++) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < eps) break;
    }
    return h;
}

// log-gamma: on the host the reentrant form -- plain lgamma() stores the sign in the global
// `signgam`, and threads finishing pairs side by side would pass that cache line around
// (measured: 16 threads no faster than 3); same value either way
MG_PV_HD double ln_gamma(double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return lgamma(v);
#else
    int sign;
    return lgamma_r(v, &sign);
#endif
}

MG_PV_HD double reg_inc_beta(double a, double b, double x)
{
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    const double ln_pre = ln_gamma(a + b) - ln_gamma(a) - ln_gamma(b) + a * log(x) + b * log1p(-x);
    if (x < (a + 1.0) / (a + b + 2.0)) return exp(ln_pre) * beta_cf(a, b, x) / a;
    return 1.0 - exp(ln_pre) * beta_cf(b, a, 1.0 - x) / b;
}

// P[Binomial(n, p) > k]
MG_PV_HD double binomial_q(uint64_t k, double p, uint64_t n)
{
    if (k >= n) return 0.0;
    return reg_inc_beta((double)k + 1.0, (double)n - (double)k, p);
}

MG_PV_HD double p_value(uint64_t x, uint64_t len_ref, uint64_t len_qry, double kmer_space,
                        uint64_t sketch_size)
{
    if (x == 0) return 1.0;
    const double pX = 1.0 / (1.0 + kmer_space / (double)len_ref);
    const double pY = 1.0 / (1.0 + kmer_space / (double)len_qry);
    const double r = pX * pY / (pX + pY - pX * pY);
    return binomial_q(x - 1, r, sketch_size);
}

MG_PV_HD double mash_distance(uint64_t common, uint64_t denom, int kmer_size)
{
    if (common == denom) return 0.0;          // avoid -0
    if (common == 0) return 1.0;              // avoid inf
    const double jaccard = (double)common / (double)denom;
    double d = -log(2 * jaccard / (1. + jaccard)) / kmer_size;
    if (d > 1) d = 1;
    return d;
}

}  // namespace mg
