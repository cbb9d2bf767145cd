/* oracle/shim/gsl/gsl_cdf.h — TEST INFRASTRUCTURE: the two GSL entry points the reference calls
 * (CommandDistance.cpp:446, CommandScreen.cpp:613, CommandBounds.cpp:103), supplied by the oracle's
 * regularized-incomplete-beta restatement (oracle/shim/gsl_binomial.c); GSL itself is not installed. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
double gsl_cdf_binomial_Q(unsigned int k, double p, unsigned int n);
double gsl_cdf_binomial_P(unsigned int k, double p, unsigned int n);
#ifdef __cplusplus
}
#endif
