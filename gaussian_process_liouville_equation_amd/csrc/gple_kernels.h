// gple_kernels.h — launchers of the non-GEMM device kernels (gple_kernels.hip, gple_predict.hip).
#pragma once
#include "gple_internal.h"

namespace gple
{
	// One squared-exponential kernel  amp * (exp(-((dx*rl0)^2 + (dp*rl1)^2)/2) + n2 * delta).
	// `l` keeps the lengths for the bit-faithful (division) form used when a Gram matrix is materialised.
	struct SEParam
	{
		double amp, n2;
		double l0, l1;
		double rl0, rl1;
	};
	// Kernel functions between "typed" points.  The real GP has one type; the complex (widely linear) GP is run as a
	// real GP on [Re; Im] whose 2N x 2N covariance has blocks  [kR + n/2, kC; kC, kI + n/2]  (DESIGN.md §complex):
	// p[type_a + type_b] with type 0 = real part, 1 = imaginary part  ->  p[0] = RR, p[1] = RI (correlation), p[2] = II.
	struct SEParamSet
	{
		SEParam p[3];
	};

	// cutoff_factor, kernel.h:301-332: 1 where |pred|^2 >= 4 var, 0 where |pred|^2 <= var, the cubic in between (one definition for every epilogue)
	__device__ __forceinline__ double cutoff_value(double pred_square, double abs_pred, double var)
	{
		if (pred_square >= 4.0 * var) return 1.0;
		if (pred_square <= var) return 0.0;
		const double a = abs_pred / sqrt(var);
		return (3.0 * 2.0 - 2.0 * a - 1.0) * ((a - 1.0) * (a - 1.0)) / 1.0;
	}
	// the factorisation's info word in the fit's scalar block (gple_capi.hip, SDEV_INFO): negative = the one-launch scheme gave up, everything derived
	// from the factor is NaN until the host has repeated it
	__device__ __forceinline__ bool fit_gave_up(const double* s_dev) { return *reinterpret_cast<const int*>(s_dev + 31) < 0; }
	// exp(x) for finite x <= 0 (the argument of a squared-exponential kernel), fp64, < 1 ulp:
	// x = k ln2 + r (Cody-Waite with an FMA), |r| <= ln2/2, degree-12 Taylor/Horner, scaling by v_ldexp_f64 (which
	// handles the gradual underflow for x < -708).  No special-case branches.
	__device__ __forceinline__ double exp_nonpos(double x)
	{
		x = fmax(x, -1100.0); // below that the result is 0 anyway; keeps the int conversion in range
		const double kd = rint(x * 1.4426950408889634074);
		double r = fma(kd, -6.93147180369123816490e-01, x);
		r = fma(kd, -1.90821492927058770002e-10, r);
		double p = 2.08767569878680989792e-09; // 1/12!
		p = fma(p, r, 2.50521083854417187751e-08);
		p = fma(p, r, 2.75573192239858906526e-07);
		p = fma(p, r, 2.75573192239858906526e-06);
		p = fma(p, r, 2.48015873015873015873e-05);
		p = fma(p, r, 1.98412698412698412698e-04);
		p = fma(p, r, 1.38888888888888888889e-03);
		p = fma(p, r, 8.33333333333333333333e-03);
		p = fma(p, r, 4.16666666666666666667e-02);
		p = fma(p, r, 1.66666666666666666667e-01);
		p = fma(p, r, 0.5);
		p = fma(p, r, 1.0);
		p = fma(p, r, 1.0);
		return ldexp(p, static_cast<int>(kd));
	}

	// ---- training side ---------------------------------------------------------------------------------------
	// ys[i] = s * y[i*stride + offset] for i < N (0 for N <= i < Np), s = 10 / max_i |label_i|, written to *s_out.
	// complex_abs != 0: |label| is the complex modulus of the (re,im) pair and both halves are produced:
	// ys[0..Np) = s*re, ys[Np..2Np) = s*im.
	hipError_t launch_prep_labels(hipStream_t s, const double* y, int stride, int complex_abs, int N, int Np, double* ys, double* s_out,
		const double* X, double* Xt, int nscal);
	// K_pad (n_total x n_total, ld) of the typed training set: n_total = Np (real) or 2*Np (complex, split at Np);
	// points Xt (N interleaved); padded rows/cols get the identity.
	// ys != nullptr: K has n_total + 64 rows; row n_total receives ys (length n_total), the 63 rows below it zeros
	hipError_t launch_gram_train(hipStream_t s, const double* Xt, int N, int Np, int n_total, SEParamSet ps, double* K, long ld,
		const double* ys = nullptr);
	// u = T * ys (lower triangular T, n x n): partial sums then reduction. part: (n/256) * n doubles.
	hipError_t launch_trmv_lower(hipStream_t s, const double* T, long ldt, int n, const double* ys, double* part, double* u);
	// v[k] = sum_i T(i,k) u[i],  w[k] = sum_i T(i,k)^2 ; optionally wx[k] = sum_i T(i,k) T(i,k+shift) (complex: diag of Mxy).
	// info != nullptr: a negative *info (the factorisation gave up, gple_chol.hip) turns v, w, wx into NaN
	hipError_t launch_colpass(hipStream_t s, const double* T, long ldt, int n, const double* u, double* v, double* w, int shift,
		double* wx, const int* info = nullptr);
	// raw sums for the real fit: out[0]=sum (v/w)^2, out[1]=sum v, out[2]=sum x v, out[3]=sum p v, out[4]=sum ys v
	// out[0..4]: the five sums of a real fit; qpart != nullptr: also *qout = sum of the nq per-block partials of a quadratic form
	// (launch_quadform_partials), saving the separate reduction launch
	hipError_t launch_real_fit_sums(hipStream_t s, const double* Xt, const double* ys, const double* v, const double* w, int N, double* out,
		const double* qpart = nullptr, int nq = 0, double* qout = nullptr);
	hipError_t launch_quadform_partials(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, int fdim, double* part);
	// out[0] = sum_ij a_i k(x_i,x_j) b_j over i,j < N with the bit-faithful SE kernel p (amp folded in, no noise)
	// fdim >= 0 multiplies every entry by ((x_i,d - x_j,d)/l_d)^2 / l_d: the kernel's derivative over l_d (kernel.cpp:99-160)
	hipError_t launch_quadform(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, int fdim, double* part,
		double* out);

	// ---- complex derivative path (gple_cderiv.hip): everything in the real [Re; Im] embedding ------------------------
	// One block of a derivative matrix: amp * G(l) * (c0 + c1 * ((x_d - x'_d)/l_d)^2 / l_d); active == 0 -> zero block.
	struct DSpec
	{
		double amp, c0, c1;
		double l0, l1;
		int dim, active;
	};
	// blocks indexed by type_a + type_b: 0 = Re-Re, 1 = Re-Im (both off-diagonal blocks), 2 = Im-Im
	struct DSpecSet
	{
		DSpec b[3];
	};
	// dC (n x n, padded with zeros) of the typed training set
	hipError_t launch_typed_deriv_gram(hipStream_t s, const double* Xt, int N, int Np, int n, DSpecSet spec, double* D);
	// Diagonals of M dC M for a dC with ONE zero diagonal block (every sub-kernel parameter of the complex GP, build_dspecs): with the
	// non-zero diagonal block A and the off-diagonal block B of dC, E = A M_a and F = B M_b (Np x n each; M_a = the rows of M on A's side,
	// M_b the others; roff = first row of M_a) give
	//   diag(M dC M)_i          = sum_j M_a(j, i) (E + 2 F)(j, i)                                   i < n
	//   (M dC M)(i, Np + i)      = sum_j M_a(j, i) (E + F)(j, Np + i) + M_a(j, Np + i) F(j, i)        i < Np
	// out_diag (n) and out_off (Np) receive alpha times these: half the flops of the full product dC M.
	hipError_t launch_cderiv_diag(hipStream_t s, const double* M, long ldm, int roff, const double* E, const double* F, long lde, int Np, int n, double alpha,
		double* out_diag, double* out_off);
	// cnt parameters that share roff in one launch: E, F advance by ef_stride per parameter, the outputs by diag_stride / off_stride
	hipError_t launch_cderiv_diag_batch(hipStream_t s, int cnt, const double* M, long ldm, int roff, const double* E, const double* F, long lde, int Np, int n,
		double alpha, double* out_diag, double* out_off, long ef_stride, long diag_stride, long off_stride);
	// out[ip] (ip = 0..7) = TrainingComplexKernel::ErrorDerivatives (complex_kernel.cpp:444-474) from the embedded quantities:
	// w (weights, n), wd (diag M, n), wx (diag of the Re-Im block, Np) and their derivatives dw (8 x n), dwd (8 x n), dwx (8 x Np)
	hipError_t launch_complex_deriv_sums(hipStream_t s, const double* w, const double* wd, const double* wx, const double* dw, const double* dwd,
		const double* dwx, int N, int Np, double* out8);
	// nine sums of one generated kernel X = amp G(l): out[3 * pair + var], pair in {(a,a), (b,b), (a,b)}, var in {1, f_0, f_1}
	hipError_t launch_multi_quadform(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, double* part, double* out9);
	// ya = X a, yb = X b for the generated kernel X (N x N); part: 2 * ceil(N/256) * N doubles
	hipError_t launch_aux_matvec(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, double* part, double* ya,
		double* yb);
	// out[4 * ip + q] = (ga.dw_x, gb.dw_y, ga.dw_y, gb.dw_x)[q] for ip = 0..7 with dw_x = dw[ip*n .. ], dw_y = dw[ip*n + Np ..]
	hipError_t launch_aux_dots(hipStream_t s, const double* ga, const double* gb, const double* dw, int N, int Np, int n, double* out32);
	// PredictiveComplexKernel::ErrorDerivatives (complex_kernel.cpp:648-668) from the 15 accumulator planes of the derivative pass
	hipError_t launch_predict_deriv_finish_complex(hipStream_t s, const double* acc, int m_rows, int m_split, const double* q, int M, double self,
		double s0, const double* s_dev, const double* labels, double* part, double* out8);

	// ---- derivative path (gple_deriv.hip) ------------------------------------------------------------------------
	// D0, D1 (n x n, ld = n, padded with zeros): dK/dl_d of the training Gram, zero diagonal (kernel.cpp:123-157, 190)
	hipError_t launch_deriv_gram(hipStream_t s, const double* Xt, int N, int n, SEParam p, double* D0, double* D1);
	// y = alpha * A x for a full column-major n x n matrix (n multiple of 256); part: (n/256) * n doubles
	hipError_t launch_gemv(hipStream_t s, const double* A, long lda, int n, const double* x, double alpha, double* part, double* y);
	// out[i] = alpha * sum_j A(j,i) B(j,i + shift)   (columns i with i + shift < n; shift = 0: plain column dots)
	hipError_t launch_coldot(hipStream_t s, const double* A, long lda, const double* B, long ldb, int n, int shift, double alpha,
		double* out);
	hipError_t launch_scale(hipStream_t s, const double* x, double alpha, int n, double* y);
	// up to three independent items of the above in ONE launch (small matrices: a derivative fit at the sizes the reference runs is a string of 4-5 us launches)
	hipError_t launch_gemv_batch(hipStream_t s, int n, int cnt, const double* const* A, long lda, const double* const* x, const double* alpha, double* part,
		double* const* y); // n <= 1024; part: cnt * (n / 256) * n doubles
	hipError_t launch_coldot_batch(hipStream_t s, int n, int cnt, const double* const* A, long lda, const double* const* B, long ldb, int shift, const double* alpha,
		double* const* out);
	hipError_t launch_scale_batch(hipStream_t s, int cnt, const double* const* x, const double* alpha, const int* n, double* const* y);
	// raw sums of the real derivative path: out[ip] = error derivative (kernel.cpp:381-400), out[4 + ip] = sum_i dv[ip][i]
	hipError_t launch_real_deriv_sums(hipStream_t s, const double* v, const double* w, const double* dv, const double* dwd, int N, int ld,
		double* out);
	// PredictiveKernel::ErrorDerivatives (kernel.cpp:524-542) from the per-row accumulators of the derivative generation pass
	hipError_t launch_predict_deriv_finish_real(hipStream_t s, const double* acc, int m_rows, const double* q, int M, double self, double sf,
		const double* s_dev, const double* labels, double* part, double* out4);

	// ---- KernelBase / cutoff ----------------------------------------------------------------------------------
	// p.amp = sf*sf, p.n2 = sn*sn; sf and sn are passed too because the derivative formulas use them unsquared.
	hipError_t launch_gram_rect(hipStream_t s, const double* L, int R, const double* Rt, int C, int same, SEParam p, double sf, double sn,
		double* K, double* dK);
	// ComplexKernelBase (gple_cgram.hip): K (R x C), K~ (R x C (re,im) pairs, nullable), dK (8 x R x C, nullable), dK~ (8 x R x C pairs, nullable)
	hipError_t launch_complex_gram(hipStream_t s, const double theta[8], const double* L, int R, const double* Rt, int C, int same, double* K,
		double* Kt, double* dK, double* dKt);
	hipError_t launch_sum(hipStream_t s, const double* part, int n, double* out);
	hipError_t launch_cutoff(hipStream_t s, const double* pred, int is_complex, const double* var, int M, double* factor);

	// ---- NLML path of test/gpr.cpp (gple_nlml.hip); x = (w_d, w_g, a, c, b): ARD weight matrix [[a, 0], [c, b]] --------------
	hipError_t launch_nlml_gram(hipStream_t s, const double* Xt, int N, int n, const double x[5], double* K);
	hipError_t launch_nlml_value(hipStream_t s, const double* T, long ldt, const double* y, const double* b, int N, double* out);
	hipError_t launch_nlml_grad(hipStream_t s, const double* Xt, int N, const double* W, long ldw, const double* b, const double x[5], double* part,
		double* out5);
	// part: nlml_predict_ksplit(M, N) * M doubles of partial sums
	int nlml_predict_ksplit(int M, int N);
	hipError_t launch_nlml_predict(hipStream_t s, const double* Xs, int M, const double* Xt, int N, const double* b, const double x[5], double* part, double* mean);

	// ---- step loop around the GP (gple_evolve.hip): Tully models, MQCLE propagation, Metropolis ---------------------------------
	hipError_t launch_pes(hipStream_t s, const double* x, int M, int model, double* out6);
	// query-list layout of one tick: off[s] = first point of source element s, qoff[e][s] / qlen[e] = rows of target e's list
	void evolve_layout(const int n[3], long qoff[3][3], long qlen[3], int off[3], int new_points = 0);
	hipError_t launch_evolve_prepare(hipStream_t s, const double* r, const int n[3], double mass, double dt, int model, double* r_new,
		unsigned char* coupled, double* const q[3], int new_points = 0);
	hipError_t launch_evolve_combine(hipStream_t s, const double* r_old, const double* r_new, const double* rho_old, const unsigned char* coupled,
		const int n[3], double mass, double dt, int model, const double* const pred[3], double* rho_new, int new_points = 0);
	hipError_t launch_mc_propose(hipStream_t s, const double* r, int n, unsigned step, unsigned long long seed, double d, double* r_prop);
	hipError_t launch_mc_weight(hipStream_t s, const double* pred, int is_complex, int n, double* weight);
	hipError_t launch_mc_accept(hipStream_t s, double* r, const double* r_prop, const double* pred, int is_complex, int n, unsigned step,
		unsigned long long seed, double* weight, unsigned* accepted);

	// ---- predict ------------------------------------------------------------------------------------------------
	struct PredictArgs
	{
		const double* Xs; // test points, M interleaved pairs
		int M;            // number of test points
		int m_rows;       // rows of the (typed) test set: Mh (real) or 2*Mh (complex), Mh = round_up(M, 128)
		int m_split;      // rows >= m_split are imaginary-part rows (== m_rows for the real GP)
		int m_split_few;  // few-points path only: the same boundary in its compact row numbering
		const double* Xt; // training points (N valid, readable up to n_split entries)
		int N;
		int n_total; // Np or 2*Np
		int n_split; // Np
		const double* T;
		long ldt;
		const double* v; // n_total weights (K^-1 y in the typed basis)
		const double* dv; // derivative pass: nparam x n_total derivatives of v (real: 4, complex embedding: 8), or nullptr
		double* dacc;     // derivative pass: accumulator planes of m_rows each
		                  //   real:    7 = [K* v, K* dv_0..3, (dK*/dl_0) v, (dK*/dl_1) v]
		                  //   complex: 15 = [c w, c dw_0..7, dc_1..6 w]
		int complex_deriv;      // 1: use dspec (complex embedding)
		DSpecSet dspec[6];      // derivative blocks of the parameters 1..6 (sub-kernel magnitudes and lengths)
		double* q;       // m_rows: || T k*_m ||^2
		double* mu;      // m_rows: k*_m . v
		SEParamSet ps;
		// What the caller consumes decides how much of the variance contraction has to run (gple_predict.hip, launch_predict_q):
		int mean_only;                   // 1: neither the variance nor the cut-off factor is consumed (PredictiveKernel's Error alone, kernel.cpp:522,
		                                 //    uses the UNCUT mean): no K*, no contraction, q = 0
		double cut_thr;                  // > 0: the variance only decides the cut-off factor (ErrorDerivatives, kernel.cpp:527): a point with
		                                 //    |mu|^2 >= cut_thr = 4 k(x*,x*) >= 4 var has factor 1 whatever q is (kernel.h:301-332) and is not contracted
		double prune_thr;                // > 0: rows with |k*|^2 below it are not contracted (gple_predict.hip, Prune)
		unsigned long long* prune_stats; // device, 4 words: [0] += ceil(live rows / 128), [1] += rows / 128, [2] work-queue counter, [3] live-row count
		// PredictiveKernel's epilogue (kernel.cpp:496-522) for a kernel that can do it itself (predict_fused256_kernel): fin_sdev != nullptr offers it,
		// launch_predict_q says through *finished whether it was taken (q and mu are then not written).  Real GP without labels only.
		const double* fin_sdev; // the fit's scalar block: [0] the label scale, [31] the factorisation's info word
		double fin_self;        // k(x*, x*)
		double *fin_mean, *fin_var, *fin_cut; // M each, nullable
	};
	// Rolling K* scratch of the predict path (HBM): bounded so that M x N never has to exist at once.
	constexpr size_t PREDICT_SCRATCH_BYTES = size_t(4) << 30;
	// doubles of scratch launch_predict_q needs for `a`; *chunk_rows = rows of the typed test set handled per pass
	size_t predict_scratch_doubles(const PredictArgs& a, int* chunk_rows, bool* few_rows);
	// a handful of test points (<= 16 typed rows, no derivatives): triangular mat-vecs with K* generated on the fly
	bool predict_is_few(const PredictArgs& a);
	size_t predict_few_scratch_doubles(const PredictArgs& a);
	hipError_t launch_predict_few(hipStream_t s, const PredictArgs& a, double* scratch);
	// fills a.q and a.mu; per chunk: kstar_gen_kernel then rownorm_kernel (bracketed by the context's chunk timers)
	hipError_t launch_predict_q(Ctx* ctx, hipStream_t s, const PredictArgs& a, double* scratch, int chunk_rows, bool few_rows, bool* finished = nullptr);
	// part of a predict beside the fit it follows (gple_predict.hip; GPLE_PREDICT_OVERLAP=1): applicable? / stream, events, buffers (*xs: room for the caller's
	// points, to be written on ctx->early_stream) / the launch (t_early: the event behind which T's first early_rows rows are final)
	bool predict_overlap_enabled();
	bool predict_overlap_applicable(const Ctx* ctx, const PredictArgs& a, int early_rows);
	hipError_t predict_overlap_prepare(Ctx* ctx, const PredictArgs& a, size_t xs_doubles, double** xs);
	hipError_t launch_predict_overlapped(Ctx* ctx, hipStream_t s, const PredictArgs& a, size_t xs_doubles, int early_rows, hipEvent_t t_early, hipEvent_t points_ready);
	// real finish: var = self - q, cutoff, cut = mu*cf/s ; optional labels -> err_out[0] += sum (mu - s t)^2
	hipError_t launch_predict_finish_real(hipStream_t s, const double* q, const double* mu, int M, double self, const double* s_dev,
		const double* labels, double* mean, double* var, double* cut, double* err_out);
	// complex finish: rows [0,M) real part, rows [m_split, m_split+M) imaginary part
	hipError_t launch_predict_finish_complex(hipStream_t s, const double* q, const double* mu, int M, int m_split, double self,
		const double* s_dev, const double* labels, double* mean, double* var, double* cut, double* err_out);
	// ---- N-level step loop (gple_evolve_n.hip): NumPES = 2 or 3, elements in the packing order (0,0), (1,0), (1,1), (2,0), ... ----
	// adiabatic quantities: out[(NP + 2 NE) i + ...] = E (NP) | F lower-packed (NE) | NAC lower-packed (NE)
	hipError_t launch_pes_n(hipStream_t s, int num_pes, const double* x, int M, int model, double* out);
	// rows of every element's query list: NE branches of every point of every element
	void evolve_layout_n(int num_pes, const int* n, long* qlen);
	hipError_t launch_evolve_prepare_n(hipStream_t s, int num_pes, const double* r, const int* n, double mass, double dt, int model, double* r_new,
		double* const* q, int new_points);
	hipError_t launch_evolve_combine_n(hipStream_t s, int num_pes, const double* r_new, const double* rho_old, const int* n, double mass, double dt, int model,
		const double* const* pred, double* rho_new, int new_points);
} // namespace gple
