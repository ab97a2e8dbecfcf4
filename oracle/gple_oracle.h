/* gple_oracle.h — C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a plain-C++ CPU restatement of the reference's GPR fit + predict
 * algorithm (reference files cited per function in gple_oracle.cpp).  It exists to CHECK the HIP path and to be
 * timed as the CPU baseline.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * nothing under gaussian_process_liouville_equation_amd/ links, imports or calls it.
 *
 * Parity status: the reference itself cannot be built in this image (Eigen / xtensor / NLopt / spdlog headers are
 * absent) and ships no golden vectors, so this oracle is pinned by (i) an independent numpy + mpmath (50 digit)
 * restatement (oracle/gen_golden.py -> tests/golden/ *.npz) and (ii) internal identities (brute-force leave-one-out,
 * finite differences, quadrature of analytic integrals).  "parity unpinned" by reference-owned fixtures.
 *
 * The interface mirrors include/gple.h one-to-one (prefix oracle_ instead of gple_, no context, host pointers only)
 * so that the parity tests drive both sides with the same code.  It reuses gple.h's structs and enums.
 */
#ifndef GPLE_ORACLE_H
#define GPLE_ORACLE_H

#include "../include/gple.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_real_fit oracle_real_fit;
typedef struct oracle_complex_fit oracle_complex_fit;

int oracle_real_gram(const double theta[4], const double* left, size_t R, const double* right, size_t C,
	int same_features, double* K, double* dK);
int oracle_complex_gram(const double theta[8], const double* left, size_t R, const double* right, size_t C, int same_features,
	double* K, double* Kt, double* dK, double* dKt);
int oracle_cutoff_factor(const double* prediction, int is_complex, const double* variance, size_t M, double* factor);

int oracle_real_fit_create(const double theta[4], const double* X, const double* y, int y_is_complex, size_t N,
	unsigned flags, gple_real_fit_scalars* scalars, oracle_real_fit** out);
int oracle_real_fit_release(oracle_real_fit* fit);
int oracle_real_fit_get(oracle_real_fit* fit, gple_real_array which, double* dst);
int oracle_real_predict(const oracle_real_fit* fit, const double* Xs, size_t M, unsigned flags, const double* labels,
	double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* scalars);

int oracle_complex_fit_create(const double theta[8], const double* X, const double* y, size_t N, unsigned flags,
	gple_complex_fit_scalars* scalars, oracle_complex_fit** out);
int oracle_complex_fit_release(oracle_complex_fit* fit);
int oracle_complex_fit_get(oracle_complex_fit* fit, gple_complex_array which, double* dst);
int oracle_complex_predict(const oracle_complex_fit* fit, const double* Xs, size_t M, unsigned flags,
	const double* labels, double* prediction, double* variance, double* cutoff_prediction,
	gple_predict_scalars* scalars);

int oracle_loose_function(const double* x, size_t n, const double* X, const double* y, size_t N,
	const double* X_extra, const double* y_extra, size_t M_extra, double* value, double* grad);

int oracle_nlml(const double x[4], const double* X, const double* y, size_t N, double* value, double* grad);
int oracle_nlml_predict(const double x[4], const double* X, const double* y, size_t N, const double* Xs, size_t M,
	double* mean);

int oracle_nlml_cross(const double x[5], const double* X, const double* y, size_t N, double* value, double* grad);
int oracle_nlml_cross_predict(const double x[5], const double* X, const double* y, size_t N, const double* Xs, size_t M,
	double* mean);

/* number of OpenMP threads the oracle will use (for the cpu_baseline "cores" field) */
int oracle_num_threads(void);
/* the OpenMP default (all online CPUs) oversubscribes a cgroup-limited box badly: oracle/binding.py sets this to the
 * CPU share actually available */
int oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
