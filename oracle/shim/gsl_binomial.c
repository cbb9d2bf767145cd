/* oracle/shim/gsl_binomial.c — TEST INFRASTRUCTURE, see gsl/gsl_cdf.h. */
#include "../mash_oracle.h"
#include "gsl/gsl_cdf.h"

double gsl_cdf_binomial_Q(unsigned int k, double p, unsigned int n) { return oracle_binomial_q(k, p, n); }
double gsl_cdf_binomial_P(unsigned int k, double p, unsigned int n) { return 1.0 - oracle_binomial_q(k, p, n); }
