// pvalue.h — tail of compareSketches: Mash distance and p-value (host + device).
//
// distance: CommandDistance.cpp:387-407.  p-value: pValue, CommandDistance.cpp:427-448, which
// calls gsl_cdf_binomial_Q(x-1, r, n) (or Boost's binomial complement CDF) — a third-party routine
// that is not in the reference tree and whose version the reference does not pin.  Both libraries
// evaluate the regularized incomplete beta in log space (error ~ |ln p| * 2^-53: hundreds of ulp at
// 1e-200, measured for Boost in tests/test_pvalue_exact.py), so "the" reference value is only
// defined up to that.  Here the tail is evaluated EXACTLY instead: P[X >= x] for X ~ Binomial(n, r)
// is a finite sum of positive terms C(n,j) r^j (1-r)^(n-j); with integer x and n every factor is a
// product / quotient of doubles, and the whole sum is carried in double-double arithmetic (106-bit
// mantissa, separate binary exponent), then rounded once.  The result is the correctly rounded value
// of the exact tail (<= 1 ulp, measured against 60-digit sums), on the host and on the device alike:
// only + - * / fma frexp ldexp are used — no exp / log / lgamma, whose device and host versions
// differ in the last bits — so the device finish kernel returns bit-identical p-values.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MG_PV_HD __host__ __device__ inline
#else
#define MG_PV_HD inline
#endif

namespace mg {

// ---- double-double with a separate exponent: value = (h + l) * 2^e, |l| <= ulp(h)/2 ----
struct DD { double h, l; };
struct SD { DD m; int e; };

// error-free transforms; contraction must stay off (a*b+c fused behind our back breaks them)
#if defined(__clang__)
#define MG_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define MG_NO_CONTRACT
#endif

MG_PV_HD DD dd_quick_two_sum(double a, double b)      // |a| >= |b|
{
    MG_NO_CONTRACT
    const double s = a + b;
    return DD{s, b - (s - a)};
}

MG_PV_HD DD dd_two_sum(double a, double b)
{
    MG_NO_CONTRACT
    const double s = a + b, bb = s - a;
    return DD{s, (a - (s - bb)) + (b - bb)};
}

MG_PV_HD DD dd_mul(DD x, DD y)
{
    MG_NO_CONTRACT
    const double p = x.h * y.h;
    double e = fma(x.h, y.h, -p);
    e += x.h * y.l + x.l * y.h;
    return dd_quick_two_sum(p, e);
}

MG_PV_HD DD dd_mul_d(DD x, double b)
{
    MG_NO_CONTRACT
    const double p = x.h * b;
    double e = fma(x.h, b, -p);
    e += x.l * b;
    return dd_quick_two_sum(p, e);
}

MG_PV_HD DD dd_div_d(DD x, double b)
{
    MG_NO_CONTRACT
    const double q1 = x.h / b;
    double r = fma(-q1, b, x.h);                    // exact remainder of the leading quotient
    r += x.l;
    const double q2 = r / b;
    return dd_quick_two_sum(q1, q2);
}

MG_PV_HD DD dd_add(DD x, DD y)
{
    MG_NO_CONTRACT
    DD s = dd_two_sum(x.h, y.h);
    const DD t = dd_two_sum(x.l, y.l);
    s.l += t.h;
    s = dd_quick_two_sum(s.h, s.l);
    s.l += t.l;
    return dd_quick_two_sum(s.h, s.l);
}

MG_PV_HD DD dd_div(DD x, DD y)
{
    MG_NO_CONTRACT
    const double q1 = x.h / y.h;
    DD r = dd_add(x, dd_mul_d(y, -q1));
    const double q2 = r.h / y.h;
    r = dd_add(r, dd_mul_d(y, -q2));
    const double q3 = r.h / y.h;
    DD q = dd_quick_two_sum(q1, q2);
    return dd_add(q, DD{q3, 0.0});
}

MG_PV_HD SD sd_norm(SD v)
{
    if (v.m.h == 0.0) return v;
    int k;
    (void)frexp(v.m.h, &k);
    if (k > 200 || k < -200) {                      // keep the mantissa far from over/underflow
        v.m.h = ldexp(v.m.h, -k);
        v.m.l = ldexp(v.m.l, -k);
        v.e += k;
    }
    return v;
}

MG_PV_HD SD sd_mul(SD a, SD b) { return sd_norm(SD{dd_mul(a.m, b.m), a.e + b.e}); }

// base^n by squaring
MG_PV_HD SD sd_pow(DD base, uint64_t n)
{
    SD r{DD{1.0, 0.0}, 0}, b = sd_norm(SD{base, 0});
    while (n) {
        if (n & 1) r = sd_mul(r, b);
        n >>= 1;
        if (n) b = sd_mul(b, b);
    }
    return r;
}

// C(n, j) p^j q^(n-j), q = 1 - p given exactly as a double-double
MG_PV_HD SD binom_pmf_sd(uint64_t j, uint64_t n, double p, DD q)
{
    const uint64_t m = j < n - j ? j : n - j;       // C(n, j) = C(n, n - j): the shorter product
    SD c{DD{1.0, 0.0}, 0};
    for (uint64_t i = 1; i <= m; i++) {
        c.m = dd_div_d(dd_mul_d(c.m, (double)(n - m + i)), (double)i);
        if (c.m.h > 0x1p400) c = sd_norm(c);         // by magnitude: a factor is below 2^64, whatever n is
    }
    c = sd_norm(c);
    return sd_mul(sd_mul(c, sd_pow(DD{p, 0.0}, j)), sd_pow(q, n - j));
}

// P[Binomial(n, p) > k], correctly rounded (see the header)
MG_PV_HD double binomial_q(uint64_t k, double p, uint64_t n)
{
    if (k >= n) return 0.0;
    if (!(p > 0.0)) return 0.0;
    if (p >= 1.0) return 1.0;
    const uint64_t x = k + 1;                        // P[X >= x]
    const double mean1 = ((double)n + 1.0) * p;
    const DD q = dd_two_sum(1.0, -p);                // 1 - p, exact
    if ((double)x > mean1) {
        // upper tail: terms fall from j = x on.  Certain underflow first: P <= (e n p / x)^x for x > n p.
        {
            const double b = 2.718281828459045 * (double)n * p / (double)x;
            if (b < 1.0) {
                int eb;
                const double mb = frexp(b, &eb);     // b = mb 2^eb, mb in [0.5, 1): log2 b <= eb
                (void)mb;
                if ((double)x * (double)eb < -1100.0) return 0.0;
            }
        }
        const SD t0 = binom_pmf_sd(x, n, p, q);
        if (t0.m.h == 0.0) return 0.0;
        int k0;
        (void)frexp(t0.m.h, &k0);
        if (t0.e + k0 < -1140) return 0.0;           // the sum has < 2^33 terms no larger than this one: still 0
        const DD pq = dd_div(DD{p, 0.0}, q);
        DD term{1.0, 0.0}, sum{1.0, 0.0};
        for (uint64_t j = x; j < n; j++) {
            term = dd_div_d(dd_mul_d(dd_mul(term, pq), (double)(n - j)), (double)(j + 1));
            sum = dd_add(sum, term);
            if (term.h < sum.h * 0x1p-110) break;
        }
        const DD r = dd_mul(t0.m, sum);
        return ldexp(r.h, t0.e) + ldexp(r.l, t0.e);
    }
    // x <= (n + 1) p: the tail is large; subtract the lower sum, whose terms fall from j = x - 1 down
    const SD t0 = binom_pmf_sd(x - 1, n, p, q);
    const DD qp = dd_div(q, DD{p, 0.0});
    DD term{1.0, 0.0}, sum{1.0, 0.0};
    for (uint64_t j = x - 1; j > 0; j--) {
        term = dd_div_d(dd_mul_d(dd_mul(term, qp), (double)j), (double)(n - j + 1));
        sum = dd_add(sum, term);
        if (term.h < sum.h * 0x1p-110) break;
    }
    DD low = dd_mul(t0.m, sum);
    low.h = ldexp(low.h, t0.e);
    low.l = ldexp(low.l, t0.e);
    const DD r = dd_add(DD{1.0, 0.0}, DD{-low.h, -low.l});
    return r.h < 0.0 ? 0.0 : r.h;
}

MG_PV_HD double p_value(uint64_t x, uint64_t len_ref, uint64_t len_qry, double kmer_space,
                        uint64_t sketch_size)
{
    MG_NO_CONTRACT                                   // r must not depend on fused multiply-adds (device vs host)
    if (x == 0) return 1.0;
    const double pX = 1.0 / (1.0 + kmer_space / (double)len_ref);
    const double pY = 1.0 / (1.0 + kmer_space / (double)len_qry);
    const double r = pX * pY / (pX + pY - pX * pY);
    return binomial_q(x - 1, r, sketch_size);
}

// host only: the device takes distances from a table built with this (same libm as the reference)
inline double mash_distance(uint64_t common, uint64_t denom, int kmer_size)
{
    if (common == denom) return 0.0;          // avoid -0
    if (common == 0) return 1.0;              // avoid inf
    const double jaccard = (double)common / (double)denom;
    double d = -log(2 * jaccard / (1. + jaccard)) / kmer_size;
    if (d > 1) d = 1;
    return d;
}

}  // namespace mg
