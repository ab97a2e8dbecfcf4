/* gple.h — C-ABI of the MI355X-native GPR fit + predict hot path.
 *
 * Drop-in boundary for kaigu1997/gaussian_process_liouville_equation (reference paths are relative to
 * /root/reference/gaussian_process_liouville_equation/ unless they start with test/).
 * The reference has NO FFI of its own: its boundary is the C++ classes of kernel.h / complex_kernel.h /
 * predict.h / opt.h.  Every entry point below names the reference constructor / getter it replaces; the
 * header-only C++ adapters in gaussian_process_liouville_equation_amd/host/ re-create those classes on top
 * of this ABI.
 *
 * Conventions
 *   - all arithmetic is fp64; complex numbers are interleaved (re, im) pairs of doubles;
 *   - phase-space points are 2 x N column-major == interleaved [x0,p0,x1,p1,...]   (stdafx.h:153);
 *   - matrices returned by the *_get calls are column-major (Eigen default)          (stdafx.h:133);
 *   - every function returns GPLE_OK (0) or a GPLE_ERR_* code; numerical breakdown is NOT an error:
 *     like the reference (opt.cpp:420-431, LDLT::info() never checked) non-finite values are returned
 *     in the outputs and `info` is set;
 *   - array arguments are host pointers unless GPLE_IO_DEVICE is set in `flags`, in which case every
 *     array argument of that call (inputs and outputs) is a device pointer on the context's device and
 *     the call is asynchronous on the context's stream except for the scalar result block;
 *   - predict calls are thread-safe on a shared const fit handle (evolve.cpp:392-420 calls the reference
 *     predictors from TBB workers); fit calls on one context must not run concurrently.
 */
#ifndef GPLE_H
#define GPLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPLE_OK 0
#define GPLE_ERR_BAD_ARG 1
#define GPLE_ERR_HIP 2
#define GPLE_ERR_ALLOC 3
#define GPLE_ERR_COLLECTIVE 5 /* RCCL could not be resolved or a collective failed (gple_ctx_last_error has the text) */
#define GPLE_ERR_STATE 4 /* e.g. derivative output requested from a fit built without GPLE_CALC_DERIVATIVE; call on a destroyed context */
/* The factorisation of a fit runs as ONE launch per block of panels whose workgroups hand tiles to each other through flags; every wait is bounded
 * (about a second), and a wave that gives up leaves the launch unfinished.  That never yields a wrong number: on the device everything derived
 * from the unfinished factor becomes NaN, and the first call on the fit that drains the stream (*_fit_create with a scalars struct,
 * *_fit_get_scalars, *_fit_get, a predict with host pointers or labels, the objective and NLML entry points) notices, repeats the
 * factorisation with one launch per panel (no waits between workgroups) and returns the correct result — the caller sees nothing but the delay.
 * GPLE_ERR_TIMEOUT is returned only (a) by such a draining call when OTHER calls had been enqueued on the fit before it (device pointers, no labels:
 * those never drain the stream) — their outputs are NaN, the fit is valid from here on, gple_ctx_last_error says how many to repeat; or
 * (b) when the repetition failed as well (it cannot give up: it has no waits). */
#define GPLE_ERR_TIMEOUT 6

/* The three bools of the Training*Kernel constructors (kernel.h:128-134, complex_kernel.h:167-173). */
#define GPLE_CALC_ERROR 0x1u
#define GPLE_CALC_AVERAGE 0x2u
#define GPLE_CALC_DERIVATIVE 0x4u
/* Array arguments of this call are device pointers. */
#define GPLE_IO_DEVICE 0x100u
/* *_predict: contract every test row.  By default a predict first sums K*^2 per test row (the generation's arithmetic without its
 * HBM write), then generates K* and contracts only the rows with |k*|^2 >= 2^-56 sf^2 sn^2 k(x*,x*) (complex: half that): for the
 * others k(x*,x*) - k* K^-1 k*^T rounds to k(x*,x*) whatever the contraction returns — grid points far from every training
 * point, most of a phase-space grid.  The outputs are bit-identical either way.  Pass the flag for measurements that want the
 * full contraction timed, and for test points known to lie on the data, where nothing can be skipped and the extra pass costs
 * 2-13 % (gple_predict_batch, the step loop and the objective evaluations do that themselves). */
#define GPLE_PREDICT_FULL 0x200u

#define GPLE_REAL_NPARAM 4    /* KernelBase::NumTotalParameters        kernel.h:33          */
#define GPLE_COMPLEX_NPARAM 8 /* ComplexKernelBase::NumTotalParameters complex_kernel.h:22  */

typedef struct gple_ctx gple_ctx;
typedef struct gple_real_fit gple_real_fit;       /* TrainingKernel        kernel.h:111-280        */
typedef struct gple_complex_fit gple_complex_fit; /* TrainingComplexKernel complex_kernel.h:150-318 */

/* Scalar getters of TrainingKernel. Fields whose flag was not requested are NaN. */
typedef struct gple_real_fit_scalars {
	double rescale_factor;           /* get_rescale_factor            kernel.h:146, kernel.cpp:279 */
	double magnitude;                /* get_magnitude                 kernel.h:167-179             */
	double error;                    /* get_error (LOOCV)             kernel.cpp:285               */
	double population;               /* get_population                kernel.cpp:286-297           */
	double first_order_average[2];   /* get_1st_order_average         kernel.cpp:298-312           */
	double purity;                   /* get_purity                    kernel.cpp:325-335           */
	double error_derivative[4];      /* get_error_derivative          kernel.cpp:381-400           */
	double population_derivative[4]; /* get_population_derivative     kernel.cpp:401-435           */
	double purity_derivative[4];     /* get_purity_derivative         kernel.cpp:436-477           */
	int info;                        /* 0: factorisation fine; k>0: non-positive pivot met at column k (NaN propagates, as in the reference).  Never -1 in a
	                                    struct a call returned with GPLE_OK: a give-up of the one-launch factorisation is recovered from first (GPLE_ERR_TIMEOUT above) */
} gple_real_fit_scalars;

/* Scalar getters of TrainingComplexKernel. */
typedef struct gple_complex_fit_scalars {
	double rescale_factor;       /* complex_kernel.cpp:262       */
	double magnitude;            /* complex_kernel.h:192-204     */
	double error;                /* complex_kernel.cpp:270-286   */
	double purity;               /* complex_kernel.cpp:357-377   */
	double error_derivative[8];  /* complex_kernel.cpp:444-474   */
	double purity_derivative[8]; /* complex_kernel.cpp:475-590   */
	int info;                    /* as in gple_real_fit_scalars  */
} gple_complex_fit_scalars;

/* Scalar getters of PredictiveKernel / PredictiveComplexKernel (only the first 4 entries of
 * error_derivative are meaningful for the real kernel). NaN when no labels / no derivative were asked. */
typedef struct gple_predict_scalars {
	double error;               /* kernel.cpp:522, complex_kernel.cpp:646         */
	double error_derivative[8]; /* kernel.cpp:524-542, complex_kernel.cpp:648-668 */
} gple_predict_scalars;

/* Arrays that the reference exposes through matrix/vector getters. */
typedef enum gple_real_array {
	GPLE_R_KERNEL = 0,       /* KernelBase::get_kernel,  N*N                kernel.h:82   */
	GPLE_R_INVERSE = 1,      /* get_inverse,             N*N                kernel.h:153  */
	GPLE_R_INVLBL = 2,       /* get_inverse_times_label, N                  kernel.h:160  */
	GPLE_R_INVLBL_DERIV = 3, /* get_inverse_times_label_derivative, 4*N     kernel.h:215  */
	GPLE_R_LABEL = 4,        /* rescaled label, N                           kernel.cpp:280 */
	GPLE_R_INVERSE_DIAG = 5  /* diagonal of get_inverse, N                                */
} gple_real_array;

typedef enum gple_complex_array {
	GPLE_C_KERNEL = 0,  /* get_kernel, N*N real                                 complex_kernel.h:90  */
	GPLE_C_PSEUDO = 1,  /* get_pseudo_kernel, N*N complex                       complex_kernel.h:97  */
	GPLE_C_UPPER_LEFT = 2,  /* get_upper_left_block_of_augmented_inverse, N*N complex   :208 */
	GPLE_C_LOWER_LEFT = 3,  /* get_lower_left_block_of_augmented_inverse, N*N complex   :215 */
	GPLE_C_INVLBL = 4,      /* get_upper_part_of_augmented_inverse_times_label, N complex :222 */
	GPLE_C_INVLBL_DERIV = 5,/* ..._derivative, 8*N complex                               :245 */
	GPLE_C_LABEL = 6        /* rescaled label, N complex                    complex_kernel.cpp:263 */
} gple_complex_array;

/* ---- context ------------------------------------------------------------------------------------- */
/* device: HIP device ordinal; stream: a hipStream_t to run on, or NULL for a stream owned by the context. */
int gple_ctx_create(int device, void* stream, gple_ctx** out);
/* Lifetime rule: every fit / objective handle holds a reference on the context it was created from.  gple_ctx_destroy()
 * drains the stream, CLOSES the context (every entry point that takes `ctx` returns GPLE_ERR_STATE from then on) and drops
 * the creator's reference; device buffers, the stream and the context itself are freed when the last handle created from it
 * is released.  Handle-only calls (*_fit_get_scalars, *_fit_get, *_fit_release, gple_objective_release) therefore stay valid
 * after the destroy and return a status instead of touching freed memory.  A second destroy of a context that handles
 * still keep alive returns GPLE_ERR_STATE; a caller-provided stream must outlive the last handle. */
int gple_ctx_destroy(gple_ctx* ctx);
int gple_ctx_synchronize(gple_ctx* ctx);
/* The context keeps every device buffer it ever needed in a grow-only pool (no hipMalloc on the steady-state path; a single
 * large predict can leave up to 4 GiB of K* scratch behind). This synchronises the stream and frees the buffers that no live
 * fit owns; *bytes_freed (nullable) receives the amount. */
int gple_ctx_trim(gple_ctx* ctx, size_t* bytes_freed);
const char* gple_status_string(int status);
/* Last HIP error text seen by this context (empty string when none). */
const char* gple_ctx_last_error(const gple_ctx* ctx);

/* ---- tracing --------------------------------------------------------------------------------------- */
/* The reference only logs wall-clock seconds per output (output.cpp:246-248). With timing enabled the library
 * brackets its phases with HIP events on the context's stream.  Timing never forces a synchronisation of its own: the
 * intervals are collected at the library's next stream synchronisation (gple_ctx_synchronize, gple_ctx_get_timing, a
 * scalar getter, a call with host outputs). */
typedef enum gple_timer {
	GPLE_TIMER_FIT = 0,            /* whole *_fit_create call (device side)                 */
	GPLE_TIMER_PREDICT = 1,        /* whole *_predict call (device side)                    */
	GPLE_TIMER_PREDICT_KERNEL = 2, /* the MFMA row-norm kernel of *_predict alone; count = its launches */
	GPLE_TIMER_DERIV_GEMM = 3      /* the dK * K^-1 MFMA GEMM of a GPLE_CALC_DERIVATIVE fit (kernel.cpp:354); count = its launches */
} gple_timer;
int gple_ctx_enable_timing(gple_ctx* ctx, int on);
/* Synchronises the stream, then: last = milliseconds of the most recent interval; total / count = accumulated since
 * enable (any may be NULL). */
int gple_ctx_get_timing(gple_ctx* ctx, gple_timer which, double* last_ms, double* total_ms, long* count);
/* Synchronises the stream, then: test rows the pruned predicts contracted / saw, in units of 128 rows (ceil of the live rows of
 * each predict), since the context was created or since the last call with reset != 0 (see GPLE_PREDICT_FULL). */
int gple_ctx_get_prune_stats(gple_ctx* ctx, unsigned long long* contracted_blocks, unsigned long long* seen_blocks, int reset);

/* ---- KernelBase (kernel.h:29-106, kernel.cpp:8-242) ---------------------------------------------- */
/* K = sf^2 (G + sn^2 delta) for theta = (sf, lx, lp, sn), left 2 x R, right 2 x C, K is R x C column-major.
 * same_features != 0 reproduces the `LeftFeature.data() == RightFeature.data()` branch (identity delta,
 * symmetric derivative with zero diagonal); dK (nullable) receives the 4 derivative matrices, 4*R*C. */
int gple_real_gram(gple_ctx* ctx, const double theta[4], const double* left, size_t R, const double* right, size_t C,
	int same_features, unsigned flags, double* K, double* dK);

/* ---- ComplexKernelBase (complex_kernel.h:14-145, complex_kernel.cpp:20-200) ------------------------ */
/* theta = (s, sR, lRx, lRp, sI, lIx, lIp, sn).  K = s^2 (K_R + K_I + sn^2 delta) is R x C real; Kt (nullable) = s^2 (K_R - K_I +
 * 2i K_C), R x C (re,im) pairs; dK (nullable) receives the 8 derivative matrices of K (8*R*C doubles, get_derivative()), dKt
 * (nullable) the 8 of Kt (16*R*C doubles, get_pseudo_derivative()).  same_features as in gple_real_gram. */
int gple_complex_gram(gple_ctx* ctx, const double theta[8], const double* left, size_t R, const double* right, size_t C,
	int same_features, unsigned flags, double* K, double* Kt, double* dK, double* dKt);

/* cutoff_factor<T> (kernel.h:301-332). prediction has M (real) or 2M (complex) doubles. */
int gple_cutoff_factor(gple_ctx* ctx, const double* prediction, int is_complex, const double* variance, size_t M,
	unsigned flags, double* factor);

/* ---- TrainingKernel (kernel.cpp:244-479) ---------------------------------------------------------- */
/* theta = (sf, lx, lp, sn); X = 2N; y: N labels; y_is_complex != 0 means y holds N (re,im) pairs of which
 * the real part is used (ElementTrainingSet = tuple<PhasePoints, VectorXcd>, kernel.h:14, kernel.cpp:280).
 * scalars == NULL defers the scalar members: with device-resident inputs (GPLE_IO_DEVICE) the call then only enqueues
 * work and returns; gple_real_fit_get_scalars() drains the stream when a getter is first needed. */
int gple_real_fit_create(gple_ctx* ctx, const double theta[4], const double* X, const double* y, int y_is_complex,
	size_t N, unsigned flags, gple_real_fit_scalars* scalars, gple_real_fit** out);
/* The scalar getters of TrainingKernel (kernel.h:181-243: rescale factor, magnitude, error, population, <r>, purity and
 * their derivatives); synchronises the context's stream on first use after a deferred create. */
int gple_real_fit_get_scalars(gple_real_fit* fit, gple_real_fit_scalars* scalars);
int gple_real_fit_retain(gple_real_fit* fit);
int gple_real_fit_release(gple_real_fit* fit);
size_t gple_real_fit_size(const gple_real_fit* fit);
/* Copies one array into dst (host pointer, or device pointer with GPLE_IO_DEVICE). Matrices that the hot
 * path does not need (the explicit inverse) are materialised on first request. */
int gple_real_fit_get(gple_real_fit* fit, gple_real_array which, unsigned flags, double* dst);

/* ---- PredictiveKernel (kernel.cpp:481-544) -------------------------------------------------------- */
/* Xs = 2M test points. labels (nullable, M) switches on `error`; GPLE_CALC_DERIVATIVE + labels switches on
 * error_derivative (needs a fit created with GPLE_CALC_DERIVATIVE). prediction (nullable) receives the
 * rescaled mean `Prediction`; variance / cutoff_prediction (nullable) as get_variance / get_cutoff_prediction. */
int gple_real_predict(gple_ctx* ctx, const gple_real_fit* fit, const double* Xs, size_t M, unsigned flags,
	const double* labels, double* prediction, double* variance, double* cutoff_prediction,
	gple_predict_scalars* scalars);

/* ---- TrainingComplexKernel / PredictiveComplexKernel (complex_kernel.cpp:221-670) ------------------ */
/* theta = (s, sR, lRx, lRp, sI, lIx, lIp, sn); y = N (re,im) pairs. */
int gple_complex_fit_create(gple_ctx* ctx, const double theta[8], const double* X, const double* y, size_t N,
	unsigned flags, gple_complex_fit_scalars* scalars, gple_complex_fit** out);
/* complex_kernel.h:214-265; deferred like gple_real_fit_get_scalars when the create call passed scalars == NULL. */
int gple_complex_fit_get_scalars(gple_complex_fit* fit, gple_complex_fit_scalars* scalars);
int gple_complex_fit_retain(gple_complex_fit* fit);
int gple_complex_fit_release(gple_complex_fit* fit);
size_t gple_complex_fit_size(const gple_complex_fit* fit);
int gple_complex_fit_get(gple_complex_fit* fit, gple_complex_array which, unsigned flags, double* dst);
/* labels: M (re,im) pairs or NULL; prediction / cutoff_prediction: M (re,im) pairs; variance: M. */
int gple_complex_predict(gple_ctx* ctx, const gple_complex_fit* fit, const double* Xs, size_t M, unsigned flags,
	const double* labels, double* prediction, double* variance, double* cutoff_prediction,
	gple_predict_scalars* scalars);

/* ---- loose_function (opt.cpp:441-482) -------------------------------------------------------------- */
/* Objective of the NLopt drivers: LOOCV error of the training set + squared error on the extra set, and
 * (grad != NULL) its gradient; n = 4 -> real kernel, n = 8 -> complex kernel. Labels are (re,im) pairs in
 * both cases (the real kernel takes the real part, opt.cpp:451). make_normal (opt.cpp:420-431) is applied. */
int gple_loose_function(gple_ctx* ctx, const double* x, size_t n, const double* X, const double* y, size_t N,
	const double* X_extra, const double* y_extra, size_t M_extra, double* value, double* grad);

/* The same objective with its data resident: ElementTrainingParameters (opt.cpp:16), i.e. the training set and the extra set
 * an NLopt optimiser carries in its `void* params` across hundreds of evaluations, are uploaded once; an evaluation then moves
 * the parameter vector in and the value (+ gradient) out.  X: 2N, y: N (re,im) pairs, extra set likewise (M_extra may be 0). */
typedef struct gple_objective gple_objective;
int gple_objective_create(gple_ctx* ctx, const double* X, const double* y, size_t N, const double* X_extra, const double* y_extra,
	size_t M_extra, gple_objective** out);
/* loose_function(x, grad, params): n = 4 (real element) or 8 (complex element); grad may be NULL. */
int gple_objective_eval(gple_objective* objective, const double* x, size_t n, double* value, double* grad);
/* One rank's share of an evaluation, for a gradient split over the GPUs of a node (configs[3]: the opt.cpp loop on 4 GPUs, where the complex
 * element alone is the whole step): every rank holds the objective (same data) and fits (replicated, like the grid-sharded predict); rank
 * `part` of `nparts` forms the N^3 derivative products of the parameters ip with ip % nparts == part (the cheap first and last parameters on
 * part 0), predicts its contiguous share of the extra points, and returns its partial value and gradient.  The sum over the parts (one
 * all-reduce of n + 1 doubles) is gple_objective_eval's value and gradient up to the rounding of that sum; make_normal (opt.cpp:420-431:
 * NaN / Inf -> DBL_MAX) is applied by the caller AFTER the sum.  nparts == 1 is gple_objective_eval.  One evaluation at a time per context
 * (the parameter split is a state of the context for the duration of the call). */
int gple_objective_eval_part(gple_objective* objective, const double* x, size_t n, int part, int nparts, double* value, double* grad);
int gple_objective_release(gple_objective* objective);

/* ---- grid-sharded predict for C++ callers (output.cpp:181-233 over several GPUs) ---------------------- */
/* One process per GPU; every rank holds the same fit (replicated: DESIGN.md §7) and calls this with the WHOLE grid Xs (2M).
 * The rank predicts its share of the points — the 128-point blocks rank, rank + world, ... (block-cyclic, so that the live
 * blocks of a mostly empty phase-space grid spread over the ranks, see GPLE_PREDICT_FULL) — and the shares are all-gathered
 * with ncclAllGather on the context's stream (RCCL over xGMI), so that prediction / variance / cutoff_prediction (each nullable, full length M resp.
 * 2M for the complex kernel) are complete on every rank when the call returns (host outputs) or when the stream reaches that
 * point (GPLE_IO_DEVICE).  nccl_comm is the caller's ncclComm_t; the RCCL entry points are resolved at first use from the
 * process (the caller links librccl) or from librccl.so.1.  world == 1 with nccl_comm == NULL is the plain predict. */
/* For callers that shard by hand (contiguous slices, what bench.py / parallel.py do): slice [lo, hi) of M points for rank, and
 * the padded slice length `per` every rank allocates. */
int gple_shard_bounds(size_t M, int rank, int world, size_t* lo, size_t* hi, size_t* per);
/* Callers with another transport (MPI, host staging) plug their own all-gather: same signature and semantics as
 * ncclAllGather(sendbuff, recvbuff, sendcount, datatype = 8 (double), comm, hipStream_t), device buffers, 0 = success.
 * NULL restores RCCL.  Process-wide; set it before the first sharded call of any thread. */
int gple_set_allgather_function(void* fn);
int gple_real_predict_sharded(gple_ctx* ctx, const gple_real_fit* fit, const double* Xs, size_t M, unsigned flags, int rank, int world,
	void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction);
int gple_complex_predict_sharded(gple_ctx* ctx, const gple_complex_fit* fit, const double* Xs, size_t M, unsigned flags, int rank,
	int world, void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction);
/* Weighted deal, for plans that give the ranks unequal shares of an element's grid (the elements of TrainingKernels,
 * predict.cpp:290-360, differ 8x in cost, and 3 + 3 elements do not divide 8 GPUs): out of every cycle of S = sum(weights) consecutive
 * 128-point blocks rank r predicts the weights[r] blocks that follow those of the ranks before it.  weights[rank] == 0: the rank takes no
 * part in this element (fit may be NULL) but still enters the all-gather — every rank of the communicator calls this once per element, in
 * the same order — and receives the full result like everyone else.  weights == NULL is the plain deal of gple_*_predict_sharded. */
int gple_real_predict_dealt(gple_ctx* ctx, const gple_real_fit* fit, const double* Xs, size_t M, unsigned flags, int rank, int world,
	const int* weights, void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction);
int gple_complex_predict_dealt(gple_ctx* ctx, const gple_complex_fit* fit, const double* Xs, size_t M, unsigned flags, int rank, int world,
	const int* weights, void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction);
/* Host-only view of a deal (no device call): the number of points of rank's share, the padded share length every rank allocates, and
 * (indices != NULL, n_local entries) the grid indices of the share in the order the rank predicts them. */
int gple_deal_share(size_t M, int rank, int world, const int* weights, size_t* n_local, size_t* per, size_t* indices);

/* ---- batched point-predict ------------------------------------------------------------------ */
/* The reference evaluates its DistributionFunction (stdafx.h:155) one phase-space point at a time: main.cpp:75-101 constructs a
 * Predictive*Kernel per call, evolve.cpp:298 asks for 8 points per sample, mc.cpp:158-172 for one per Metropolis step.  This entry
 * takes all requests of a tick at once: request r wants the cut-off prediction (get_cutoff_prediction().value(), main.cpp:83,94)
 * of element element_of_request[r] at point points[2r .. 2r+1].  elements[e] names the fit of density-matrix element e: exactly one
 * of real / cplx is set, or neither for an element without a kernel (result 0, main.cpp:86-88).  One predict per element that
 * has requests (gather -> predict -> scatter); out receives n_req (re,im) pairs (imaginary part 0 for real elements).
 * Thread-safe; host pointers only. */
typedef struct gple_element
{
	const gple_real_fit* real;
	const gple_complex_fit* cplx;
} gple_element;
int gple_predict_batch(gple_ctx* ctx, const gple_element* elements, size_t n_elements, const double* points, const int* element_of_request,
	size_t n_req, double* out);

/* ---- the searches of Optimization (opt.cpp:333-355, 517-587, 730-800, 940-1015) ------------------------- */
/* Own implementations behind NLopt's C callback ABIs (NLopt is un-vendored and absent): the reference's objective and
 * constraint callbacks plug in unchanged.  options == NULL: the reference's settings (opt.cpp:344-346). */
typedef double (*gple_objective_fn)(unsigned n, const double* x, double* grad, void* data);                                /* nlopt_func  */
typedef void (*gple_constraint_fn)(unsigned m, double* result, unsigned n, const double* x, double* grad, void* data); /* nlopt_mfunc */
typedef struct gple_opt_options
{
	double xtol_rel, ftol_rel, xtol_abs, ftol_abs; /* 1e-5, 1e-5, 1e-15, 1e-15 */
	double initial_step;                           /* 0.5 (derivative-free search only) */
	int max_eval;                                  /* 0: 400 per free dimension (Nelder-Mead) / 2000 (augmented Lagrangian) */
} gple_opt_options;
/* LN_NELDERMEAD stand-in inside the box [lb, ub] (NULL = unbounded; lb[i] == ub[i] fixes x_i).  x: start point in, minimiser out. */
int gple_minimize_neldermead(gple_objective_fn f, void* data, unsigned n, const double* lb, const double* ub, const gple_opt_options* options,
	double* x, double* fmin, int* n_eval);
/* The same search on the resident objective (loose_function, opt.cpp:441-482; n = 4 or 8) with the simplex vertices evaluated
 * concurrently: objectives[k] are handles on the SAME data created on different contexts (one HIP stream each). */
int gple_objective_minimize_neldermead(gple_objective* const* objectives, size_t n_objectives, size_t n, const double* lb, const double* ub,
	const gple_opt_options* options, double* x, double* fmin, int* n_eval);
/* GN_DIRECT_L stand-in — the global tier (opt.h:54, opt.cpp:336, 1344-1365): Gablonsky & Kelley's locally-biased DIRECT inside the finite
 * box [lb, ub] (lb[i] == ub[i] fixes x_i; x on entry only supplies the fixed coordinates).  options->max_eval == 0: the reference's
 * MaximumEvaluations = 100000 (opt.cpp:339); the tolerances stop the search the way NLopt's do (xtol on the rectangles an iteration
 * divides, ftol on an iteration that improves the minimum). */
int gple_minimize_direct_l(gple_objective_fn f, void* data, unsigned n, const double* lb, const double* ub, const gple_opt_options* options,
	double* x, double* fmin, int* n_eval);
/* The same search on the resident objective with all new rectangle centres of an iteration evaluated concurrently (objectives[k]: handles
 * on the SAME data on different contexts).  is_log (nullable, n entries): coordinates that are logarithms of the parameter they stand for —
 * loose_function_global_wrapper, opt.cpp:489-497; lb, ub and x are in those coordinates. */
int gple_objective_minimize_direct_l(gple_objective* const* objectives, size_t n_objectives, size_t n, const double* lb, const double* ub,
	const unsigned char* is_log, const gple_opt_options* options, double* x, double* fmin, int* n_eval);
/* AUGLAG_EQ stand-in: minimise f subject to h(x) = 0 (m equality constraints, row-major m x n gradient) inside the box. */
int gple_minimize_auglag_eq(gple_objective_fn f, void* fdata, gple_constraint_fn h, void* hdata, unsigned m, unsigned n, const double* lb,
	const double* ub, const gple_opt_options* options, double* x, double* fmin, int* n_eval);

/* ---- the step loop around the GP (NumPES = 2, Dim = 1 as the reference instantiates it) ---------------- */
typedef enum gple_pes_model /* pes.h:27-41; the reference's default TestModel is DAC */
{
	GPLE_PES_SAC = 0, /* Tully I:   simple avoided crossing            */
	GPLE_PES_DAC = 1, /* Tully II:  dual avoided crossing               */
	GPLE_PES_ECR = 2  /* Tully III: extended coupling with reflection  */
} gple_pes_model;
/* adiabatic_potential / adiabatic_force / adiabatic_coupling (pes.cpp:98-155) at M positions:
 * out[6 i + {0..5}] = E0, E1, F(0,0), F(1,0), F(1,1), NAC(0,1). */
int gple_pes_adiabatic(gple_ctx* ctx, int model, const double* x, size_t M, unsigned flags, double* out);

/* One tick of evolve() (evolve.cpp:377-423): the points of the three density-matrix elements (0,0), (1,0), (1,1) are propagated
 * (two half steps forward) and their densities rebuilt by the 3-branch back-propagation of non_adiabatic_evolve_predict
 * (evolve.cpp:184-372), with `distribution` = the cut-off prediction of elements[e] (main.cpp:75-101; 0 for an element without
 * a fit).  density[e].r (2 n_e, interleaved) and density[e].rho (n_e (re,im) pairs) are updated in place; host pointers, or
 * device pointers with GPLE_IO_DEVICE.  All 8 * (n_0 + n_1 + n_2) back-propagated points are predicted in three batches. */
typedef struct gple_points
{
	double* r;
	double* rho;
	size_t n;
} gple_points;
/* flags: GPLE_IO_DEVICE, and GPLE_EVOLVE_NEW_POINTS = new_point_predict (evolve.cpp:425-443) for every point instead of a tick:
 * the points stay where they are, rho (input ignored) receives the density the three-branch back-propagation predicts there
 * from the fits alone — no exact density enters — or 0 where the point does not couple (what is_very_small, evolve.cpp:445-478,
 * and new_element_point_selection, mc.cpp:405-537, evaluate for an element that has no points yet). */
#define GPLE_EVOLVE_NEW_POINTS 0x400u
int gple_evolve(gple_ctx* ctx, const gple_element elements[3], int pes_model, double mass, double dt, gple_points density[3],
	unsigned flags);

/* The same for N-level systems — SURVEY.md §8f N3 "extend non_adiabatic_evolve_predict beyond NumPES == 2" (the reference asserts there,
 * evolve.cpp:367-371); num_pes = 2 or 3, elements and density hold NE = num_pes (num_pes + 1) / 2 entries in the reference's packing order
 * (0,0), (1,0), (1,1), (2,0), (2,1), (2,2): real fits on the diagonal, complex ones off it.  The back-propagation is the N-level form of the
 * operator splitting the two-level code implements: NE momentum branches (the eigen-pairs of the off-diagonal force matrix)
 * x NE source elements of predicted densities per point, one batch per element; at num_pes = 2 it reproduces gple_evolve to rounding.
 * pes_model 0-2: pes.cpp's diabatic_potential as it compiles for num_pes levels (Tully's two surfaces; at three levels plus the uncoupled
 * third diabat at V = 0 that the unfilled matrix entries give); 3 (num_pes = 3 only): a three-state avoided-crossing model of this library,
 * V00 = A tanh(B x), V11 = 0, V22 = -A tanh(B x), V01 = V12 = C sech(D x) with A = 0.02, B = 0.8, C = 0.005, D = 0.5 — the reference has no
 * genuinely three-level model.  Adiabatic states: ascending energy, last non-zero component of every eigenvector positive. */
int gple_evolve_n(gple_ctx* ctx, int num_pes, const gple_element* elements, int pes_model, double mass, double dt, gple_points* density,
	unsigned flags);
/* adiabatic_potential / adiabatic_force / adiabatic_coupling (pes.cpp:98-155) for num_pes levels at M positions:
 * out[(num_pes + 2 NE) i + ...] = E (num_pes, ascending) | F lower-packed (NE) | NAC lower-packed (NE; NAC(j, k) = F(j, k) / (E_j - E_k), j > k). */
int gple_pes_adiabatic_n(gple_ctx* ctx, int num_pes, int model, const double* x, size_t M, unsigned flags, double* out);

/* generate_markov_chain (mc.cpp:118-165) for n walkers at once on the fitted distribution |cut-off prediction| of `element`:
 * num_steps Metropolis steps with uniform displacements in [-max_displacement, max_displacement) per dimension; r (2n) holds
 * the start points and receives the last points, accept_ratio (nullable, n) the accepted fraction per walker.  Random numbers:
 * Philox4x32-10, counter (walker, step, block, 0), key = seed (the reference's generator is seeded from the clock and shared
 * between threads, mc.cpp:17: no stream of it is reproducible).  Host pointers. */
int gple_markov_chain(gple_ctx* ctx, const gple_element* element, size_t num_steps, double max_displacement, unsigned long long seed,
	double* r, size_t n, double* accept_ratio);
/* The same with every chain recorded (the WholeChain of mc.cpp:118-165, what autocorrelation_optimize_steps mc.cpp:167-285 looks at):
 * chain[(step * n + walker) * 2 + d], step = 0 (start point) .. num_steps. */
int gple_markov_chain_trace(gple_ctx* ctx, const gple_element* element, size_t num_steps, double max_displacement,
	unsigned long long seed, double* r, size_t n, double* accept_ratio, double* chain);

/* ---- negative_log_marginal_likelihood / predict (test/gpr.cpp:499-532, 654-706) -------------------- */
/* Kernel = w_d^2 * Diag + w_g^2 * GaussianARD(weights), x = (w_d, w_g, a_x, a_p) with `a` the diagonal ARD
 * weights = inverse lengths (NOCROSS build, test/gpr.cpp:99,323-326). value = y^T K^-1 y / 2 + sum log L_ii;
 * grad (nullable, 4) = tr[(K^-1 - b b^T) dK] / 2 with the reference's dK (test/gpr.cpp:408-468). */
int gple_nlml(gple_ctx* ctx, const double x[4], const double* X, const double* y, size_t N, double* value,
	double* grad);
/* Mean-only prediction k(x*, X) K^-1 y with the noise kernel excluded off the training set
 * (test/gpr.cpp:384-388, 692-700). */
int gple_nlml_predict(gple_ctx* ctx, const double x[4], const double* X, const double* y, size_t N, const double* Xs,
	size_t M, unsigned flags, double* mean);

/* The default (cross-term) build of test/gpr.cpp (:99-103, 313-321, 436-452): the ARD kernel carries the lower-triangular weight
 * matrix W = [[a, 0], [c, b]], k = w_g^2 exp(-|W^T (x - x')|^2 / 2) = w_g^2 exp(-(x - x')^T W W^T (x - x') / 2);
 * x = (w_d, w_g, a, c, b) in the reference's hyper-parameter order ("rowwise parameters", :313-321).  grad (nullable, 5): the same
 * trace formula with dK/da, dK/dc, dK/db scaled as at :436-452.  Shogun's matrix-weight kernel is restated from the formula
 * comment (:356-367): parity unpinned (Shogun). */
int gple_nlml_cross(gple_ctx* ctx, const double x[5], const double* X, const double* y, size_t N, double* value, double* grad);
int gple_nlml_cross_predict(gple_ctx* ctx, const double x[5], const double* X, const double* y, size_t N, const double* Xs,
	size_t M, unsigned flags, double* mean);

#ifdef __cplusplus
}
#endif
#endif /* GPLE_H */
