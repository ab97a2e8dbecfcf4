// gple_capi.hip — C-ABI entry points of include/gple.h (host orchestration of the HIP kernels).
//
// Data layout in HBM (all fp64):
//   training points  Xt   : 2 x Np interleaved, Np = round_up(N, 256), zero padded
//   typed labels     ys   : n_total   (real GP: Np;  complex GP: [s Re y ; s Im y], 2 Np)
//   inverse factor   T    : n_total x n_total column-major, T = chol(K_pad)^-1 (lower), upper part zero
//   weights          v    : n_total   (K^-1 ys),  diag(K^-1) w : n_total,  complex: diag of the off-diagonal block wx : Np
//   explicit inverse W    : n_total x n_total, built only for get_inverse()/derivatives (W = T^T T)
// A fit handle owns Xt, ys, T, v, w (+ lazily W and the derivative vectors); the Cholesky work matrix, the merge-tree
// workspace and all predict scratch belong to the context's buffer pool and are reused across calls.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>

#include "gple_debug.h"
#include "gple_kernels.h"

using namespace gple;

namespace gple
{
	int record_hip_error(Ctx* ctx, hipError_t e, const char* what, int line)
	{
		if (ctx)
		{
			std::lock_guard<std::mutex> lk(ctx->mu);
			ctx->last_error = std::string(hipGetErrorString(e)) + " in " + what + " (gple_capi.hip:" + std::to_string(line) + ")";
		}
		(void)hipGetLastError();
		return e == hipErrorOutOfMemory ? GPLE_ERR_ALLOC : GPLE_ERR_HIP;
	}
	static bool take_event(Ctx* c, hipEvent_t* e)
	{
		if (!c->ev_free.empty())
		{
			*e = c->ev_free.back();
			c->ev_free.pop_back();
			return true;
		}
		return hipEventCreate(e) == hipSuccess;
	}
	void timer_start(Ctx* c, int which)
	{
		if (!c->timing || c->pending.size() >= 8192) return; // nobody collects: stop recording rather than grow
		Ctx::TimedSpan sp{which, nullptr, nullptr};
		if (!take_event(c, &sp.e0)) return;
		if (!take_event(c, &sp.e1))
		{
			c->ev_free.push_back(sp.e0);
			return;
		}
		(void)hipEventRecord(sp.e0, c->stream);
		c->open_span[which] = sp;
	}
	void timer_stop(Ctx* c, int which)
	{
		Ctx::TimedSpan& sp = c->open_span[which];
		if (!sp.e0) return;
		(void)hipEventRecord(sp.e1, c->stream);
		c->pending.push_back(sp);
		sp = Ctx::TimedSpan{};
	}
	void chunk_timer_start(Ctx* c) { timer_start(c, GPLE_TIMER_PREDICT_KERNEL); }
	void chunk_timer_stop(Ctx* c) { timer_stop(c, GPLE_TIMER_PREDICT_KERNEL); }
	void timer_collect(Ctx* c)
	{
		for (const Ctx::TimedSpan& sp : c->pending)
		{
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, sp.e0, sp.e1) == hipSuccess)
			{
				c->t_last[sp.which] = ms;
				c->t_total[sp.which] += ms;
				c->t_count[sp.which] += 1;
			}
			c->ev_free.push_back(sp.e0);
			c->ev_free.push_back(sp.e1);
		}
		c->pending.clear();
	}
} // namespace gple

// ---- context with a grow-only buffer pool ----------------------------------------------------------------
struct gple_ctx: gple::Ctx
{
	struct PoolEntry
	{
		void* p;
		size_t bytes;
		bool used;
	};
	std::vector<PoolEntry> pool;
	std::mutex pool_mu;
	std::mutex call_mu; // serialises fit / predict calls that share the pooled scratch
	// Lifetime: the creator holds one reference, every live fit / objective one more.  gple_ctx_destroy() closes the context
	// (entry points that take it return GPLE_ERR_STATE from then on) and drops the creator's reference; the device buffers,
	// the stream and the struct itself go when the last handle created from it is released.
	std::atomic<int> refs{1};
	std::atomic<bool> closed{false};

	double* acquire(size_t bytes, hipError_t* err)
	{
		std::lock_guard<std::mutex> lk(pool_mu);
		*err = hipSuccess;
		if (bytes == 0) bytes = 8;
		PoolEntry* best = nullptr;
		for (PoolEntry& e : pool)
			if (!e.used && e.bytes >= bytes && e.bytes <= 2 * bytes + 4096 && (!best || e.bytes < best->bytes)) best = &e;
		if (best)
		{
			best->used = true;
			return static_cast<double*>(best->p);
		}
		void* p = nullptr;
		*err = hipMalloc(&p, bytes);
		if (*err != hipSuccess) return nullptr;
		pool.push_back({p, bytes, true});
		return static_cast<double*>(p);
	}
	void give_back(void* p)
	{
		if (!p) return;
		std::lock_guard<std::mutex> lk(pool_mu);
		for (PoolEntry& e : pool)
			if (e.p == p) e.used = false;
	}
};

static void ctx_retain(gple_ctx* c) { c->refs.fetch_add(1); }
// drops one reference; the last one tears the context down
static void ctx_drop(gple_ctx* ctx)
{
	if (ctx->refs.fetch_sub(1) != 1) return;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	for (auto& e : ctx->pool) (void)hipFree(e.p);
	if (ctx->host_scalars) (void)hipHostFree(ctx->host_scalars);
	if (ctx->prune_stats) (void)hipFree(ctx->prune_stats);
	if (ctx->dag_flags) (void)hipFree(ctx->dag_flags);
	timer_collect(ctx);
	for (hipEvent_t e : ctx->ev_free) (void)hipEventDestroy(e);
	if (ctx->side_stream)
	{
		(void)hipStreamSynchronize(ctx->side_stream);
		(void)hipStreamDestroy(ctx->side_stream);
		for (hipEvent_t ev : ctx->side_forks) (void)hipEventDestroy(ev);
		if (ctx->side_join) (void)hipEventDestroy(ctx->side_join);
	}
	if (ctx->early_stream)
	{
		(void)hipStreamSynchronize(ctx->early_stream);
		(void)hipStreamDestroy(ctx->early_stream);
		for (hipEvent_t ev : {ctx->early_points, ctx->early_done, ctx->early_free})
			if (ev) (void)hipEventDestroy(ev);
		if (ctx->early_buf) (void)hipFree(ctx->early_buf);
	}
	if (ctx->fit_early_own) (void)hipEventDestroy(ctx->fit_early_own);
	if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}
#define GPLE_OPEN(ctx)                                  \
	do                                                  \
	{                                                   \
		if ((ctx)->closed.load()) return GPLE_ERR_STATE; \
	} while (0)

namespace
{
	// per-fit device scalar block (doubles) and offsets into the context's pinned host block
	constexpr int SDEV_N = 512;        // [0] s, [1..15] base sums, [16..28] real derivative sums, [31] info, [32..39] complex error derivative,
	                                   // [64..108] complex purity quadratic forms (5 kernels x 9), [128..287] aux dots (5 x 8 x 4)
	constexpr int HS_PRED_ERR = 512, HS_PRED_DERIV = 520, HS_NLML = 540;
	constexpr int SDEV_INFO = 31; // the factorisation's info word (an int in the double's slot; the finish kernels read it: gple_kernels.hip, fit_gave_up)
	// a handful of test points with host pointers (the reference's one-point predicts): inputs and outputs go through the pinned
	// block itself (device-visible), not through four hipMemcpyAsync of pageable memory
	constexpr int HS_FEW_XS = 600, HS_FEW_LAB = 640, HS_FEW_MEAN = 680, HS_FEW_VAR = 720, HS_FEW_CUT = 740;
	constexpr size_t FEW_HOST_POINTS = 16;

	// pooled buffer with scope lifetime
	struct Scratch
	{
		gple_ctx* ctx;
		double* p = nullptr;
		explicit Scratch(gple_ctx* c): ctx(c) {}
		Scratch(const Scratch&) = delete;
		~Scratch() { ctx->give_back(p); }
		hipError_t get(size_t doubles)
		{
			hipError_t e;
			p = ctx->acquire(doubles * sizeof(double), &e);
			return e;
		}
	};

	double nan_() { return std::numeric_limits<double>::quiet_NaN(); }

	SEParam make_se(double amp, double n2, double l0, double l1) { return SEParam{amp, n2, l0, l1, 1.0 / l0, 1.0 / l1}; }

	// kernel.h:285-294
	SEParam purity_aux(double mag, double l0, double l1)
	{
		const double m = mag * mag * std::sqrt(l0 * l1);
		return make_se(m * m, 0.0, std::sqrt(2.0) * l0, std::sqrt(2.0) * l1);
	}

	// copies `n` doubles host->device or device->device depending on the IO flag
	hipError_t copy_in(hipStream_t s, double* dst, const double* src, size_t n, bool dev)
	{
		if (n == 0) return hipSuccess;
		return hipMemcpyAsync(dst, src, n * sizeof(double), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
	}
	hipError_t copy_out(hipStream_t s, double* dst, const double* src, size_t n, bool dev)
	{
		if (n == 0 || dst == nullptr) return hipSuccess;
		return hipMemcpyAsync(dst, src, n * sizeof(double), dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s);
	}
} // namespace

// ---- fit handles ---------------------------------------------------------------------------------------------
struct FitCommon
{
	std::atomic<int> refs{1};
	gple_ctx* ctx = nullptr;
	int N = 0, Np = 0, n_total = 0;
	unsigned flags = 0;
	bool is_complex = false;
	double* Xt = nullptr;   // 2*Np
	double* ys = nullptr;   // n_total
	double* T = nullptr;    // n_total^2
	double* v = nullptr;    // n_total
	double* w = nullptr;    // n_total (diag of K^-1)
	double* wx = nullptr;   // Np (complex only)
	double* W = nullptr;    // n_total^2, lazy
	double* dv = nullptr;   // derivatives of v over the parameters, [nparam][n_total] (GPLE_CALC_DERIVATIVE fits only)
	double sf = 1.0;        // real kernel: magnitude (needed unsquared by the derivative formulas)
	double s0 = 1.0;        // complex kernel: global magnitude
	DSpecSet dspec[6];      // complex kernel: derivative blocks of the parameters 1..6 (zero-initialised = inactive)
	double* sdev = nullptr; // [0] rescale factor, [1..] raw sums, [31] info (as int)
	double s_host = 0.0;
	bool sc_ready = false; // host scalars computed (deferred when the caller passed no scalars struct)
	// The one-launch factorisation may give up waiting (info = -1, gple_chol.hip): the device then turns everything derived from T into NaN, and
	// the first host synchronisation on this fit (validate_fit: the scalar getters, every *_fit_get, a predict that drains the stream) repeats
	// the factorisation with a launch per panel.  Calls that consumed the fit before that — enqueued, never synchronised — have produced NaN:
	mutable std::atomic<bool> validated{false}; // the host has seen info >= 0 (or has recovered)
	mutable std::atomic<int> stale_uses{0};     // predicts enqueued on the not yet validated fit
	int early_rows = 0; // T(0 .. early_rows, .) is final behind the context's fit_early_event (0: only behind the whole fit)
	unsigned deriv_mask = 0xFFu; // which parameters' N^3 products a derivative fit forms (bit ip; gple_objective_eval_part splits them over ranks)
	SEParamSet ps{};
	double self = 0.0; // k(x*, x*)
	FitCommon() { std::memset(dspec, 0, sizeof(dspec)); }
	std::mutex lazy_mu;
	// one-point predicts from several host threads on this fit (the reference calls its DistributionFunction from TBB workers,
	// evolve.cpp:392-420, mc.cpp:214-246): requests that arrive while a predict is in flight are served together by the next one
	struct PointRequest
	{
		const double* x;
		double *mean, *var, *cut;
		int status;
		bool done;
	};
	mutable std::mutex point_mu;
	mutable std::condition_variable point_cv;
	mutable std::vector<PointRequest*> point_pending;
	mutable bool point_leader = false;

	~FitCommon()
	{
		if (!ctx) return;
		for (double* p : {Xt, ys, T, v, w, wx, W, dv, sdev}) ctx->give_back(p);
		ctx_drop(ctx); // the reference fit_common() took
	}
};
struct gple_real_fit: FitCommon
{
	double theta[4];
	gple_real_fit_scalars sc;
};
struct gple_complex_fit: FitCommon
{
	double theta[8];
	gple_complex_fit_scalars sc;
};

namespace
{
	// K^-1 (explicit) on first request
	int ensure_inverse(FitCommon* f)
	{
		std::lock_guard<std::mutex> lk(f->lazy_mu);
		if (f->W) return GPLE_OK;
		gple_ctx* ctx = f->ctx;
		hipError_t e;
		double* W = ctx->acquire(static_cast<size_t>(f->n_total) * f->n_total * sizeof(double), &e);
		GPLE_HIP(ctx, e);
		e = lauum_full(ctx->stream, f->T, f->n_total, W, f->n_total, f->n_total);
		if (e != hipSuccess)
		{
			ctx->give_back(W);
			GPLE_HIP(ctx, e);
		}
		f->W = W;
		return GPLE_OK;
	}

	int fit_factor(gple_ctx* ctx, FitCommon* f);
	// shared front half of both fits: upload, label scaling, Gram, Cholesky, inverse factor, weights
	int fit_common(gple_ctx* ctx, FitCommon* f, const double* X, const double* y, int y_stride, size_t N, unsigned flags)
	{
		hipStream_t st = ctx->stream;
		const bool dev = flags & GPLE_IO_DEVICE;
		f->ctx = ctx;
		ctx_retain(ctx); // a fit keeps its context alive (released in ~FitCommon)
		f->N = static_cast<int>(N);
		f->Np = static_cast<int>(round_up(N, NPAD));
		f->n_total = f->is_complex ? 2 * f->Np : f->Np;
		f->flags = flags;
		const int Np = f->Np, nt = f->n_total;
		hipError_t e;
		f->Xt = ctx->acquire(2 * static_cast<size_t>(Np) * 8, &e);
		GPLE_HIP(ctx, e);
		f->ys = ctx->acquire(static_cast<size_t>(nt) * 8, &e);
		GPLE_HIP(ctx, e);
		f->T = ctx->acquire(static_cast<size_t>(nt) * nt * 8, &e);
		GPLE_HIP(ctx, e);
		f->v = ctx->acquire(static_cast<size_t>(nt) * 8, &e);
		GPLE_HIP(ctx, e);
		f->w = ctx->acquire(static_cast<size_t>(nt) * 8, &e);
		GPLE_HIP(ctx, e);
		f->sdev = ctx->acquire(SDEV_N * 8, &e);
		GPLE_HIP(ctx, e);
		if (f->is_complex)
		{
			f->wx = ctx->acquire(static_cast<size_t>(Np) * 8, &e);
			GPLE_HIP(ctx, e);
		}
		Scratch ytmp(ctx);
		timer_start(ctx, GPLE_TIMER_FIT);
		const size_t ylen = N * static_cast<size_t>(y_stride);
		GPLE_HIP(ctx, ytmp.get(ylen));

		// device inputs are read in place; host inputs are staged by two copies.  One launch then pads the points, clears the
		// scalar block and rescales the labels (every launch costs 2.9 us on the GPU timeline)
		const double *Xin = X, *yin = y;
		Scratch xtmp(ctx);
		if (!dev)
		{
			GPLE_HIP(ctx, xtmp.get(2 * N));
			GPLE_HIP(ctx, copy_in(st, xtmp.p, X, 2 * N, false));
			GPLE_HIP(ctx, copy_in(st, ytmp.p, y, ylen, false));
			Xin = xtmp.p, yin = ytmp.p;
		}
		GPLE_HIP(ctx, launch_prep_labels(st, yin, y_stride, f->is_complex ? 1 : 0, f->N, Np, f->ys, f->sdev, Xin, f->Xt, SDEV_N));
		return fit_factor(ctx, f);
	}

	// Gram, Cholesky, inverse factor, weights of a fit whose points and scaled labels are in place (fit_common; recover_fit repeats it under the
	// launch-per-panel scheme after a give-up).  The scheme is the calling thread's (CholSchemeScope) or the context's (debug knob).
	int fit_factor(gple_ctx* ctx, FitCommon* f)
	{
		hipStream_t st = ctx->stream;
		const int Np = f->Np, nt = f->n_total;
		CholSchemeScope scheme(ctx->chol_scheme);
		Scratch Lbuf(ctx), work(ctx), u(ctx);
		GPLE_HIP(ctx, Lbuf.get(static_cast<size_t>(nt + CHOL_NB) * nt));
		GPLE_HIP(ctx, work.get(chol_inverse_work_doubles(nt)));
		GPLE_HIP(ctx, u.get(nt));
		// T is lower block-triangular and every consumer keeps to the blocks on and below the diagonal (the diagonal 64-blocks leave the
		// panel step complete, zeros above the diagonal included), so the blocks above are never written and never read: no 8 n^2-byte
		// memset in front of the factorisation (35 us at n = 4096).  GPLE_POISON_T=1 fills T with NaN bit patterns first — the GPU
		// test session runs that way (tests/conftest.py), so a reader of an unwritten block cannot go unnoticed.
		static const bool poison_t = [] {
			const char* e = getenv("GPLE_POISON_T");
			return e != nullptr && atoi(e) != 0;
		}();
		if (poison_t) GPLE_HIP(ctx, hipMemsetAsync(f->T, 0xFF, static_cast<size_t>(nt) * nt * 8, st));
		if (poison_t) GPLE_HIP(ctx, hipMemsetAsync(Lbuf.p, 0xFF, static_cast<size_t>(nt + CHOL_NB) * nt * 8, st)); // likewise the working matrix above its diagonal blocks
		// the scaled labels ride below the matrix as one more block row: its factor is u = L^-1 ys (= T ys, without the two launches)
		const long ldl = nt + CHOL_NB;
		GPLE_HIP(ctx, launch_gram_train(st, f->Xt, f->N, Np, nt, f->ps, Lbuf.p, ldl, f->ys));
		int* info_dev = reinterpret_cast<int*>(f->sdev + 31);
		GPLE_HIP(ctx, chol_inverse_factor(ctx, st, Lbuf.p, ldl, nt, f->T, nt, info_dev, work.p, u.p));
		f->early_rows = ctx->fit_early_rows; // (0 under the launch-per-panel scheme, for one-block matrices, and where the first row block's inverse is a merge tree)
		GPLE_HIP(ctx, launch_colpass(st, f->T, nt, nt, u.p, f->v, f->w, Np, f->wx, info_dev)); // (a give-up of the factorisation: NaN)
		return GPLE_OK;
	}

	// GPLE_DERIV_BATCH=0: a launch per product in the derivative fits of small matrices too (A/B, and the bits check of the batched path)
	static bool deriv_batching()
	{
		static const bool on = [] {
			const char* e = getenv("GPLE_DERIV_BATCH");
			return e == nullptr || atoi(e) != 0;
		}();
		return on;
	}
	// TrainingKernel derivative members (kernel.cpp:337-477) for the real kernel; raw sums land in sdev[16..28]
	int real_fit_derivatives(gple_ctx* ctx, FitCommon* f, double sf, double l0, double l1, double sn, unsigned flags)
	{
		hipStream_t st = ctx->stream;
		const int nt = f->n_total;
		const size_t n2 = static_cast<size_t>(nt) * nt;
		GPLE_TRY(ensure_inverse(f));
		hipError_t e;
		f->dv = ctx->acquire(4 * static_cast<size_t>(nt) * 8, &e);
		GPLE_HIP(ctx, e);
		Scratch D(ctx), C(ctx), part(ctx), tvec(ctx), dwd(ctx), qpart(ctx);
		GPLE_HIP(ctx, D.get(2 * n2));
		GPLE_HIP(ctx, C.get(n2));
		GPLE_HIP(ctx, part.get(static_cast<size_t>(nt / 256) * nt));
		GPLE_HIP(ctx, tvec.get(nt));
		GPLE_HIP(ctx, dwd.get(4 * static_cast<size_t>(nt)));
		double* dv = f->dv;
		GPLE_HIP(ctx, launch_deriv_gram(st, f->Xt, f->N, nt, f->ps.p[0], D.p, D.p + n2));
		if (nt <= 1024 && deriv_batching())
		{
			// Small matrices: the same products, batched into a third of the launches (17 -> 8; a derivative fit at N = 256 is 0.1 ms of 4-5 us kernels).
			// Items of a batch are independent; every item's arithmetic is that of the launch-per-item path below (same bits).
			const double cn = -2.0 * (sf * sf) * sn;
			Scratch tv2(ctx), C2(ctx), part3(ctx);
			GPLE_HIP(ctx, tv2.get(2 * static_cast<size_t>(nt)));
			GPLE_HIP(ctx, part3.get(3 * static_cast<size_t>(nt / 256) * nt));
			{
				const double* x[2] = {f->v, f->w};
				const double al[2] = {-2.0 / sf, -2.0 / sf};
				const int nn[2] = {nt, nt};
				double* y[2] = {dv, dwd.p};
				GPLE_HIP(ctx, launch_scale_batch(st, 2, x, al, nn, y)); // magnitude: dW = -2 W / sf (:349)
			}
			{
				const double* A[3] = {D.p, D.p + n2, f->W};
				const double* x[3] = {f->v, f->v, f->v};
				const double al[3] = {1.0, 1.0, cn};
				double* y[3] = {tv2.p, tv2.p + nt, dv + 3 * static_cast<size_t>(nt)};
				GPLE_HIP(ctx, launch_gemv_batch(st, nt, 3, A, nt, x, al, part3.p, y)); // dK_d v; noise: (dW) y = cn W v (:358)
			}
			{
				const double* A[2] = {f->W, f->W};
				const double* x[2] = {tv2.p, tv2.p + nt};
				const double al[2] = {-1.0, -1.0};
				double* y[2] = {dv + static_cast<size_t>(nt), dv + 2 * static_cast<size_t>(nt)};
				GPLE_HIP(ctx, launch_gemv_batch(st, nt, 2, A, nt, x, al, part3.p, y)); // (dW) y = -W (dK v) (:354)
			}
			int act[2], nact = 0;
			for (int d = 0; d < 2; ++d)
			{
				if (f->deriv_mask >> (1 + d) & 1u) act[nact++] = d;
				else GPLE_HIP(ctx, hipMemsetAsync(dwd.p + static_cast<size_t>(1 + d) * nt, 0, static_cast<size_t>(nt) * sizeof(double), st));
			}
			const double* cA[3];
			const double* cB[3];
			double cal[3];
			double* cy[3];
			int nc = 0;
			if (nact)
			{
				GPLE_HIP(ctx, C2.get(nact * n2));
				GemmDesc g{};
				g.A = D.p + act[0] * n2, g.lda = nt, g.strideA = nact == 2 ? static_cast<long>(n2) : 0, g.B = f->W, g.ldb = nt, g.strideB = 0, g.C = C2.p, g.ldc = nt,
				g.strideC = static_cast<long>(n2);
				g.M = nt, g.N = nt, g.K = nt, g.batch = nact, g.alpha = 1.0, g.beta = 0.0, g.krange = K_FULL, g.lower_only = 0;
				g.a_kmajor = false, g.b_kmajor = false, g.c_trans = false;
				timer_start(ctx, GPLE_TIMER_DERIV_GEMM);
				GPLE_HIP(ctx, launch_gemm(st, g, gemm_pick_tile(nt, nt, 1, false))); // (the tile of ONE product: the items must not change kernel, i.e. bits, with the batch)
				timer_stop(ctx, GPLE_TIMER_DERIV_GEMM);
				for (int a = 0; a < nact; ++a) cA[nc] = f->W, cB[nc] = C2.p + a * n2, cal[nc] = -1.0, cy[nc] = dwd.p + static_cast<size_t>(1 + act[a]) * nt, ++nc;
			}
			cA[nc] = f->W, cB[nc] = f->W, cal[nc] = cn, cy[nc] = dwd.p + 3 * static_cast<size_t>(nt), ++nc;
			GPLE_HIP(ctx, launch_coldot_batch(st, nt, nc, cA, nt, cB, nt, 0, cal, cy)); // diag(dW)_i = -sum_j W(j,i) (dK W)(j,i); noise: cn sum_j W(j,i)^2
		}
		else
		{
		// magnitude: dW = -2 W / sf (:349)
		GPLE_HIP(ctx, launch_scale(st, f->v, -2.0 / sf, nt, dv));
		GPLE_HIP(ctx, launch_scale(st, f->w, -2.0 / sf, nt, dwd.p));
		// lengths: dW = -W dK W (:354): (dW) y = -W (dK v), diag(dW)_i = -sum_j W(j,i) (dK W)(j,i)
		for (int d = 0; d < 2; ++d)
		{
			const double* Dd = D.p + d * n2;
			GPLE_HIP(ctx, launch_gemv(st, Dd, nt, nt, f->v, 1.0, part.p, tvec.p));
			GPLE_HIP(ctx, launch_gemv(st, f->W, nt, nt, tvec.p, -1.0, part.p, dv + static_cast<size_t>(1 + d) * nt));
			if (!(f->deriv_mask >> (1 + d) & 1u)) // another rank forms this parameter's diag(W dK W) (gple_objective_eval_part); (dW) y above is needed by every rank's predict
			{
				GPLE_HIP(ctx, hipMemsetAsync(dwd.p + static_cast<size_t>(1 + d) * nt, 0, static_cast<size_t>(nt) * sizeof(double), st));
				continue;
			}
			GemmDesc g{};
			g.A = Dd, g.lda = nt, g.B = f->W, g.ldb = nt, g.C = C.p, g.ldc = nt;
			g.M = nt, g.N = nt, g.K = nt, g.batch = 1, g.alpha = 1.0, g.beta = 0.0, g.krange = K_FULL, g.lower_only = 0;
			g.a_kmajor = false, g.b_kmajor = false, g.c_trans = false;
			timer_start(ctx, GPLE_TIMER_DERIV_GEMM);
			GPLE_HIP(ctx, launch_gemm(st, g, gemm_pick_tile(nt, nt, 1, false)));
			timer_stop(ctx, GPLE_TIMER_DERIV_GEMM);
			GPLE_HIP(ctx, launch_coldot(st, f->W, nt, C.p, nt, nt, 0, -1.0, dwd.p + static_cast<size_t>(1 + d) * nt));
		}
		// noise: dW = -2 sf^2 sn W W (:358)
		const double cn = -2.0 * (sf * sf) * sn;
		GPLE_HIP(ctx, launch_gemv(st, f->W, nt, nt, f->v, cn, part.p, dv + 3 * static_cast<size_t>(nt)));
		GPLE_HIP(ctx, launch_coldot(st, f->W, nt, f->W, nt, nt, 0, cn, dwd.p + 3 * static_cast<size_t>(nt)));
		}
		GPLE_HIP(ctx, launch_real_deriv_sums(st, f->v, f->w, dv, dwd.p, f->N, nt, f->sdev + 16));
		if (flags & GPLE_CALC_AVERAGE)
		{
			const size_t g = (static_cast<size_t>(f->N) + 63) / 64;
			GPLE_HIP(ctx, qpart.get(g * g));
			const SEParam k1 = purity_aux(sf, l0, l1);
			for (int d = 0; d < 2; ++d) // v^T (dK1/dl'_d) v and dv_d^T K1 v
			{
				GPLE_HIP(ctx, launch_quadform(st, f->Xt, f->N, k1, f->v, f->v, d, qpart.p, f->sdev + 24 + d));
				GPLE_HIP(ctx, launch_quadform(st, f->Xt, f->N, k1, dv + static_cast<size_t>(1 + d) * nt, f->v, -1, qpart.p, f->sdev + 26 + d));
			}
			GPLE_HIP(ctx, launch_quadform(st, f->Xt, f->N, k1, dv + 3 * static_cast<size_t>(nt), f->v, -1, qpart.p, f->sdev + 28));
		}
		return GPLE_OK;
	}

	struct ComplexAux
	{
		double sC, lC[2];
		SEParam k[5]; // purity auxiliary kernels R', I', C', RC, IC (complex_kernel.cpp:287-356)
		double lRC[2], lIC[2];
	};
	ComplexAux complex_aux(const double* th)
	{
		const double sR = th[1], lR0 = th[2], lR1 = th[3], sI = th[4], lI0 = th[5], lI1 = th[6];
		ComplexAux a;
		const double ss0 = lR0 * lR0 + lI0 * lI0, ss1 = lR1 * lR1 + lI1 * lI1;
		a.sC = std::sqrt(sR * sI * ((2.0 * lR0 * lI0 / ss0) * (2.0 * lR1 * lI1 / ss1))); // complex_kernel.cpp:144-157
		a.lC[0] = std::sqrt(ss0 / 2.0), a.lC[1] = std::sqrt(ss1 / 2.0);
		auto mixed = [](double m1, double a0, double a1, double mb, double b0, double b1) { // complex_kernel.cpp:206-219
			const double prod = (0.5 * (1.0 / (a0 * a0) + 1.0 / (b0 * b0))) * (0.5 * (1.0 / (a1 * a1) + 1.0 / (b1 * b1)));
			const double m = m1 * mb / std::sqrt(std::sqrt(prod));
			return make_se(m * m, 0.0, std::sqrt(a0 * a0 + b0 * b0), std::sqrt(a1 * a1 + b1 * b1));
		};
		a.k[0] = purity_aux(sR, lR0, lR1);
		a.k[1] = purity_aux(sI, lI0, lI1);
		a.k[2] = purity_aux(a.sC, a.lC[0], a.lC[1]);
		a.k[3] = mixed(sR, lR0, lR1, a.sC, a.lC[0], a.lC[1]);
		a.k[4] = mixed(sI, lI0, lI1, a.sC, a.lC[0], a.lC[1]);
		a.lRC[0] = a.k[3].l0, a.lRC[1] = a.k[3].l1, a.lIC[0] = a.k[4].l0, a.lIC[1] = a.k[4].l1;
		return a;
	}
	// derivative blocks dC_p (p = 1..6) of the reference's dK / dK~ (complex_kernel.cpp:20-132), without the s^2 factor (sic)
	void build_dspecs(const double* th, DSpecSet* out)
	{
		const ComplexAux a = complex_aux(th);
		const double sR = th[1], sI = th[4];
		const double lR[2] = {th[2], th[3]}, lI[2] = {th[5], th[6]};
		const double aR = sR * sR, aI = sI * sI, aC = a.sC * a.sC;
		auto spec = [](double amp, double c0, double c1, const double* l, int dim) { return DSpec{amp, c0, c1, l[0], l[1], dim, 1}; };
		std::memset(out, 0, 6 * sizeof(DSpecSet));
		out[0].b[0] = spec(aR, 2.0 / sR, 0.0, lR, 0); // d/d sR: dC_xx = 2 KR / sR, dC_xy = KC / sR
		out[0].b[1] = spec(aC, 1.0 / sR, 0.0, a.lC, 0);
		out[3].b[2] = spec(aI, 2.0 / sI, 0.0, lI, 0); // d/d sI: dC_yy = 2 KI / sI, dC_xy = KC / sI
		out[3].b[1] = spec(aC, 1.0 / sI, 0.0, a.lC, 0);
		for (int d = 0; d < 2; ++d)
		{
			out[1 + d].b[0] = spec(aR, 0.0, 1.0, lR, d); // d/d lR_d: dC_xx = DR_d
			out[1 + d].b[1] = spec(aC, 1.0 / lR[d] - lR[d] / (a.lC[d] * a.lC[d]), 0.5 * lR[d] / a.lC[d], a.lC, d);
			out[4 + d].b[2] = spec(aI, 0.0, 1.0, lI, d); // d/d lI_d: dC_yy = DI_d
			out[4 + d].b[1] = spec(aC, 1.0 / lI[d] - lI[d] / (a.lC[d] * a.lC[d]), 0.5 * lI[d] / a.lC[d], a.lC, d);
		}
	}

	// TrainingComplexKernel derivative members (complex_kernel.cpp:379-590); raw sums land in sdev[32..39], [64..108], [128..287]
	int complex_fit_derivatives(gple_ctx* ctx, FitCommon* f, const double* th, unsigned flags)
	{
		hipStream_t st = ctx->stream;
		const int nt = f->n_total, Np = f->Np;
		const size_t n2 = static_cast<size_t>(nt) * nt;
		const double s0 = th[0], sn = th[7];
		GPLE_TRY(ensure_inverse(f));
		hipError_t e;
		f->dv = ctx->acquire(8 * static_cast<size_t>(nt) * 8, &e);
		GPLE_HIP(ctx, e);
		Scratch D(ctx), C(ctx), part(ctx), tvec(ctx), dwd(ctx), dwx(ctx);
		GPLE_HIP(ctx, D.get(n2));
		GPLE_HIP(ctx, C.get(n2));
		GPLE_HIP(ctx, part.get(static_cast<size_t>(nt / 256) * nt));
		GPLE_HIP(ctx, tvec.get(nt));
		GPLE_HIP(ctx, dwd.get(8 * static_cast<size_t>(nt)));
		GPLE_HIP(ctx, dwx.get(8 * static_cast<size_t>(Np)));
		double* dw = f->dv;
		if (nt <= 1024 && deriv_batching() && (f->deriv_mask & 0x7eu) == 0x7eu)
		{
			// Small matrices, every parameter formed here: the launch-per-product sequence below with its independent products batched (55 -> 25 launches;
			// a complex derivative fit at N = 256 is 0.3 ms of 4-5 us kernels).  Every item keeps its arithmetic (same bits).
			Scratch D6(ctx), tv6(ctx), part3(ctx), EF(ctx);
			const size_t ef = 2 * static_cast<size_t>(Np) * nt; // [E | F] of one parameter
			GPLE_HIP(ctx, D6.get(6 * n2));
			GPLE_HIP(ctx, tv6.get(6 * static_cast<size_t>(nt)));
			GPLE_HIP(ctx, part3.get(3 * static_cast<size_t>(nt / 256) * nt));
			GPLE_HIP(ctx, EF.get(6 * ef));
			{
				const double* x[3] = {f->v, f->w, f->wx};
				const double al[3] = {-2.0 / s0, -2.0 / s0, -2.0 / s0};
				const int nn[3] = {nt, nt, Np};
				double* y[3] = {dw, dwd.p, dwx.p};
				GPLE_HIP(ctx, launch_scale_batch(st, 3, x, al, nn, y)); // global magnitude: dM = -2 M / s
			}
			GPLE_HIP(ctx, launch_gemv(st, f->W, nt, nt, f->v, -sn, part.p, dw + 7 * static_cast<size_t>(nt))); // noise: dM = -sn M M
			GPLE_HIP(ctx, launch_coldot(st, f->W, nt, f->W, nt, nt, 0, -sn, dwd.p + 7 * static_cast<size_t>(nt)));
			GPLE_HIP(ctx, launch_coldot(st, f->W, nt, f->W, nt, nt, Np, -sn, dwx.p + 7 * static_cast<size_t>(Np)));
			for (int ip = 1; ip <= 6; ++ip) GPLE_HIP(ctx, launch_typed_deriv_gram(st, f->Xt, f->N, Np, nt, f->dspec[ip - 1], D6.p + (ip - 1) * n2));
			for (int grp = 0; grp < 2; ++grp) // parameters 1..3 (R kernel), 4..6 (I kernel)
			{
				const double *A[3], *x[3];
				double al[3];
				double* y[3];
				for (int z = 0; z < 3; ++z) A[z] = D6.p + (3 * grp + z) * n2, x[z] = f->v, al[z] = 1.0, y[z] = tv6.p + static_cast<size_t>(3 * grp + z) * nt;
				GPLE_HIP(ctx, launch_gemv_batch(st, nt, 3, A, nt, x, al, part3.p, y));
				for (int z = 0; z < 3; ++z) A[z] = f->W, x[z] = tv6.p + static_cast<size_t>(3 * grp + z) * nt, al[z] = -1.0, y[z] = dw + static_cast<size_t>(1 + 3 * grp + z) * nt;
				GPLE_HIP(ctx, launch_gemv_batch(st, nt, 3, A, nt, x, al, part3.p, y));
				// E = A M_a, F = B M_b of the three parameters (the comment at the launch-per-product path below)
				const long ao = grp == 0 ? 0 : Np, bo = grp == 0 ? Np : 0;
				double* E = EF.p + 3 * grp * ef;
				GemmDesc g{};
				g.lda = nt, g.strideA = static_cast<long>(n2), g.B = f->W + ao * static_cast<long>(nt), g.ldb = nt, g.strideB = 0, g.C = E, g.ldc = Np, g.strideC = static_cast<long>(ef);
				g.A = D6.p + 3 * grp * n2 + ao + ao * static_cast<long>(nt);
				g.M = Np, g.N = nt, g.K = Np, g.batch = 3, g.alpha = 1.0, g.beta = 0.0, g.krange = K_FULL, g.lower_only = 0;
				g.a_kmajor = false, g.b_kmajor = false, g.c_trans = false;
				timer_start(ctx, GPLE_TIMER_DERIV_GEMM);
				GPLE_HIP(ctx, launch_gemm(st, g, gemm_pick_tile(Np, nt, 1, false))); // (the tile of ONE product: an item's kernel must not depend on the batch)
				g.A = D6.p + 3 * grp * n2 + ao + bo * static_cast<long>(nt);
				g.B = f->W + bo * static_cast<long>(nt), g.C = E + static_cast<size_t>(Np) * nt;
				GPLE_HIP(ctx, launch_gemm(st, g, gemm_pick_tile(Np, nt, 1, false)));
				timer_stop(ctx, GPLE_TIMER_DERIV_GEMM);
				GPLE_HIP(ctx, launch_cderiv_diag_batch(st, 3, f->W, nt, static_cast<int>(ao), E, E + static_cast<size_t>(Np) * nt, Np, Np, nt, -1.0,
					dwd.p + static_cast<size_t>(1 + 3 * grp) * nt, dwx.p + static_cast<size_t>(1 + 3 * grp) * Np, static_cast<long>(ef), nt, Np));
			}
		}
		else
		{
		// global magnitude: dC = 2 C / s  ->  dM = -2 M / s
		GPLE_HIP(ctx, launch_scale(st, f->v, -2.0 / s0, nt, dw));
		GPLE_HIP(ctx, launch_scale(st, f->w, -2.0 / s0, nt, dwd.p));
		GPLE_HIP(ctx, launch_scale(st, f->wx, -2.0 / s0, Np, dwx.p));
		// noise: dK = 2 sn I, dK~ = 0 (complex_kernel.cpp:49-56, 129)  ->  dC = sn I  ->  dM = -sn M M
		GPLE_HIP(ctx, launch_gemv(st, f->W, nt, nt, f->v, -sn, part.p, dw + 7 * static_cast<size_t>(nt)));
		GPLE_HIP(ctx, launch_coldot(st, f->W, nt, f->W, nt, nt, 0, -sn, dwd.p + 7 * static_cast<size_t>(nt)));
		GPLE_HIP(ctx, launch_coldot(st, f->W, nt, f->W, nt, nt, Np, -sn, dwx.p + 7 * static_cast<size_t>(Np)));
		for (int ip = 1; ip <= 6; ++ip)
		{
			GPLE_HIP(ctx, launch_typed_deriv_gram(st, f->Xt, f->N, Np, nt, f->dspec[ip - 1], D.p));
			GPLE_HIP(ctx, launch_gemv(st, D.p, nt, nt, f->v, 1.0, part.p, tvec.p));
			GPLE_HIP(ctx, launch_gemv(st, f->W, nt, nt, tvec.p, -1.0, part.p, dw + static_cast<size_t>(ip) * nt));
			if (!(f->deriv_mask >> ip & 1u)) // another rank forms this parameter's diagonals (gple_objective_eval_part)
			{
				GPLE_HIP(ctx, hipMemsetAsync(dwd.p + static_cast<size_t>(ip) * nt, 0, static_cast<size_t>(nt) * sizeof(double), st));
				GPLE_HIP(ctx, hipMemsetAsync(dwx.p + static_cast<size_t>(ip) * Np, 0, static_cast<size_t>(Np) * sizeof(double), st));
				continue;
			}
			// Only diag(M dC M) and the diagonal of its Re-Im block are consumed, and every dC of a sub-kernel parameter has one zero
			// diagonal block (build_dspecs: C_yy does not depend on the R kernel's parameters, C_xx not on the I kernel's).  With A the
			// non-zero diagonal block, B the off-diagonal one and M_a / M_b the matching row halves of M,
			//   m_i^T dC m_i' = M_a,i^T (A M_a + B M_b)_i' + M_a,i'^T (B M_b)_i :
			// two products of Np x n x Np (E = A M_a, F = B M_b) instead of one of n x n x n — half the flops of the form before
			// (six n^3 GEMMs were 94 of the 153 ms of a complex objective evaluation at N = 4096).  M symmetric: M(k, c) is read as M(c, k).
			const bool a_is_xx = ip <= 3;                                // ip 1..3: parameters of the R kernel (C_yy' = 0); 4..6: of the I kernel (C_xx' = 0)
			const long ao = a_is_xx ? 0 : Np, bo = a_is_xx ? Np : 0;      // row / column offset of the A side and of the B side
			double *E = C.p, *F = C.p + static_cast<size_t>(Np) * nt;     // Np x n each
			GemmDesc g{};
			g.lda = nt, g.B = f->W + ao * static_cast<long>(nt), g.ldb = nt, g.C = E, g.ldc = Np;
			g.A = D.p + ao + ao * static_cast<long>(nt);                  // A: the non-zero diagonal block of dC
			g.M = Np, g.N = nt, g.K = Np, g.batch = 1, g.alpha = 1.0, g.beta = 0.0, g.krange = K_FULL, g.lower_only = 0;
			g.a_kmajor = false, g.b_kmajor = false, g.c_trans = false;
			timer_start(ctx, GPLE_TIMER_DERIV_GEMM);
			GPLE_HIP(ctx, launch_gemm(st, g, gemm_pick_tile(Np, nt, 1, false)));
			g.A = D.p + ao + bo * static_cast<long>(nt);                  // B: rows on the A side, columns on the other
			g.B = f->W + bo * static_cast<long>(nt), g.C = F;
			GPLE_HIP(ctx, launch_gemm(st, g, gemm_pick_tile(Np, nt, 1, false)));
			timer_stop(ctx, GPLE_TIMER_DERIV_GEMM);
			GPLE_HIP(ctx, launch_cderiv_diag(st, f->W, nt, static_cast<int>(ao), E, F, Np, Np, nt, -1.0, dwd.p + static_cast<size_t>(ip) * nt,
				dwx.p + static_cast<size_t>(ip) * Np));
		}
		}
		GPLE_HIP(ctx, launch_complex_deriv_sums(st, f->v, f->w, f->wx, dw, dwd.p, dwx.p, f->N, Np, f->sdev + 32));
		if (flags & GPLE_CALC_AVERAGE)
		{
			const ComplexAux a = complex_aux(th);
			const size_t g64 = (static_cast<size_t>(f->N) + 63) / 64, g256 = (static_cast<size_t>(f->N) + 255) / 256;
			Scratch qpart(ctx), mpart(ctx), ga(ctx), gb(ctx);
			GPLE_HIP(ctx, qpart.get(9 * g64 * g64));
			GPLE_HIP(ctx, mpart.get(2 * g256 * g256 * 256));
			GPLE_HIP(ctx, ga.get(g256 * 256));
			GPLE_HIP(ctx, gb.get(g256 * 256));
			const double *wr = f->v, *wi = f->v + Np;
			for (int x = 0; x < 5; ++x)
			{
				GPLE_HIP(ctx, launch_multi_quadform(st, f->Xt, f->N, a.k[x], wr, wi, qpart.p, f->sdev + 64 + 9 * x));
				GPLE_HIP(ctx, launch_aux_matvec(st, f->Xt, f->N, a.k[x], wr, wi, mpart.p, ga.p, gb.p));
				GPLE_HIP(ctx, launch_aux_dots(st, ga.p, gb.p, dw, f->N, Np, nt, f->sdev + 128 + 32 * x));
			}
		}
		return GPLE_OK;
	}

	// complex_kernel.cpp:475-590 from the raw sums (weights w = 2 v, so every quadratic form carries 1/4)
	void complex_purity_derivative(const double* th, const double* h, double s, double* out8)
	{
		const ComplexAux a = complex_aux(th);
		const double sR = th[1], sI = th[4];
		const double lR[2] = {th[2], th[3]}, lI[2] = {th[5], th[6]};
		const double GlobalFactor = (2.0 * M_PI) * 2.0 * M_PI; // :497 — no magnitude^4 here (reference quirk, :584 vs :370)
		auto Q = [&](int x, int pair, int var) { return h[64 + 9 * x + 3 * pair + var]; };
		auto Dd = [&](int x, int ip, int q) { return h[128 + 32 * x + 4 * ip + q]; };
		for (int ip = 0; ip < 8; ++ip)
		{
			double c0[5] = {0, 0, 0, 0, 0}, c1[5] = {0, 0, 0, 0, 0};
			int d = 0;
			if (ip == 1)
				c0[0] = 4.0 / sR, c0[2] = 2.0 / sR, c0[3] = 3.0 / sR, c0[4] = 1.0 / sR; // :525-529
			else if (ip == 2 || ip == 3)
			{
				d = ip - 2;
				const double l = lR[d], roc2 = l / (a.lC[d] * a.lC[d]);
				c0[0] = 1.0 / l, c1[0] = std::sqrt(2.0);                                                                 // :534
				c0[2] = 2.0 / l - 3.0 * roc2 / 2.0, c1[2] = 1.0 / std::sqrt(2.0) * (l / a.lC[d]);                          // :536-537
				c0[3] = (2.0 / l - roc2 / 2.0) - 1.5 * (l / a.lRC[d]) / a.lRC[d], c1[3] = 1.5 * (l / a.lRC[d]);             // :538-539
				c0[4] = (1.0 / l - roc2 / 2.0) - (l / a.lIC[d]) / 2.0 / a.lIC[d], c1[4] = (l / a.lIC[d]) / 2.0;             // :540-541
			}
			else if (ip == 4)
				c0[1] = 4.0 / sI, c0[2] = 2.0 / sI, c0[3] = 1.0 / sI, c0[4] = 3.0 / sI; // :548-552
			else if (ip == 5 || ip == 6)
			{
				d = ip - 5;
				const double l = lI[d], ioc2 = l / (a.lC[d] * a.lC[d]);
				c0[1] = 1.0 / l, c1[1] = std::sqrt(2.0);                                                                 // :558
				c0[2] = 2.0 / l - 3.0 * ioc2 / 2.0, c1[2] = 1.0 / std::sqrt(2.0) * (l / a.lC[d]);                          // :559-560
				c0[3] = (1.0 / l - ioc2 / 2.0) - (l / a.lRC[d]) / 2.0 / a.lRC[d], c1[3] = (l / a.lRC[d]) / 2.0;             // :561-562
				c0[4] = (2.0 / l - ioc2 / 2.0) - 1.5 * (l / a.lIC[d]) / a.lIC[d], c1[4] = 1.5 * (l / a.lIC[d]);             // :563-564
			}
			auto dX = [&](int x, int pair) { return c0[x] * Q(x, pair, 0) + c1[x] * Q(x, pair, 1 + d); };
			const double k1c[3] = {1.0, 1.0, 2.0};
			double T1 = 0.0, T2 = 0.0;
			for (int x = 0; x < 3; ++x)
			{
				T1 += k1c[x] * (Dd(x, ip, 0) + Dd(x, ip, 1)); // Re(v^H K1 dv)
				T2 += k1c[x] * (dX(x, 0) + dX(x, 1));         // Re(v^H K1' v)
			}
			const double T3 = (Dd(0, ip, 0) - Dd(1, ip, 0)) - (Dd(0, ip, 1) - Dd(1, ip, 1)) + 2.0 * (Dd(3, ip, 2) + Dd(4, ip, 2))
				+ 2.0 * (Dd(3, ip, 3) + Dd(4, ip, 3));                                                      // Re(v^T K2 dv)
			const double T4 = (dX(0, 0) - dX(1, 0)) - (dX(0, 1) - dX(1, 1)) + 2.0 * 2.0 * (dX(3, 2) + dX(4, 2)); // Re(v^T K2' v)
			out8[ip] = 0.25 * (2.0 * T1 + T2 + 2.0 * T3 + T4) * GlobalFactor / (s * s); // :580-584
		}
	}

	template <typename S>
	void fill_nan_scalars(S* sc)
	{
		double* d = reinterpret_cast<double*>(sc);
		for (size_t i = 0; i < (sizeof(S) - sizeof(int)) / sizeof(double); ++i) d[i] = nan_();
		sc->info = 0;
	}

	// complex raw sums: out[0] = LOOCV error, out[1] = Re(conj(ys).v) (both in the reference's complex convention)
	__global__ void __launch_bounds__(1024) complex_fit_sums_kernel(const double* __restrict__ ys, const double* __restrict__ wv,
		const double* __restrict__ wd, const double* __restrict__ wx, int N, int Np, double* __restrict__ out)
	{
		__shared__ double red[16];
		double s0 = 0.0, s1 = 0.0;
		for (int i = threadIdx.x; i < N; i += 1024)
		{
			// P_ii = (Mxx + Myy)/4, Q_ii = ((Mxx - Myy) - 2i Mxy)/4, v_i = (wx + i wy)/2   
			const double mxx = wd[i], myy = wd[Np + i], mxy = wx[i];
			const double p = 0.25 * (mxx + myy), qr = 0.25 * (mxx - myy), qi = -0.5 * mxy;
			const double vr = 0.5 * wv[i], vi = 0.5 * wv[Np + i];
			// numerator P v - conj(Q v)   (complex_kernel.cpp:281)
			const double qvr = qr * vr - qi * vi, qvi = qr * vi + qi * vr;
			const double nr = p * vr - qvr, ni = p * vi + qvi;
			const double den = p * p - (qr * qr + qi * qi);
			const double dr = nr / den, di = ni / den;
			s0 += dr * dr + di * di;
			s1 += ys[i] * vr + ys[Np + i] * vi;
		}
		for (int q = 0; q < 2; ++q)
		{
			double x = q == 0 ? s0 : s1;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
			__syncthreads();
			if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
			__syncthreads();
			if (threadIdx.x == 0)
			{
				double tot = 0.0;
				for (int i = 0; i < 16; ++i) tot += red[i];
				out[q] = tot;
			}
		}
	}

	// P / Q blocks of the augmented inverse from the real 2Np x 2Np inverse M (column-major, ld = 2 Np):
	//   P = ((Mxx + Myy) + i (Mxy^T - Mxy)) / 4,   Q = ((Mxx - Myy) - i (Mxy + Mxy^T)) / 4
	__global__ void __launch_bounds__(256) complex_blocks_kernel(const double* __restrict__ Mi, long ld, int N, int Np, int which,
		double* __restrict__ out)
	{
		const int i = blockIdx.x * 64 + (threadIdx.x & 63);
		const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
		if (i >= N || j >= N) return;
		const double mxx = Mi[i + j * ld], myy = Mi[(Np + i) + (Np + j) * ld];
		const double mxy = Mi[i + (Np + j) * ld], mxyT = Mi[j + (Np + i) * ld];
		double re, im;
		if (which == 0) re = 0.25 * (mxx + myy), im = 0.25 * (mxyT - mxy);
		else re = 0.25 * (mxx - myy), im = -0.25 * (mxy + mxyT);
		out[2 * (i + static_cast<long>(j) * N)] = re;
		out[2 * (i + static_cast<long>(j) * N) + 1] = im;
	}
	// K (real) and pseudo-kernel (complex) of the complex training set, regenerated for the getters
	__global__ void __launch_bounds__(256) complex_kernels_kernel(const double* __restrict__ C, long ld, int N, int Np, int which,
		double* __restrict__ out)
	{
		const int i = blockIdx.x * 64 + (threadIdx.x & 63);
		const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
		if (i >= N || j >= N) return;
		const double cxx = C[i + j * ld], cyy = C[(Np + i) + (Np + j) * ld], cxy = C[i + (Np + j) * ld];
		if (which == 0) out[i + static_cast<long>(j) * N] = cxx + cyy; // K = Cxx + Cyy
		else
		{
			out[2 * (i + static_cast<long>(j) * N)] = cxx - cyy; // Re Kt
			out[2 * (i + static_cast<long>(j) * N) + 1] = 2.0 * cxy; // Im Kt
		}
	}
	__global__ void __launch_bounds__(256) halve_pair_kernel(const double* __restrict__ a, int N, int Np, double* __restrict__ out)
	{
		const int i = blockIdx.x * 256 + threadIdx.x;
		if (i < N) out[2 * i] = 0.5 * a[i], out[2 * i + 1] = 0.5 * a[Np + i];
	}
	__global__ void __launch_bounds__(256) pair_kernel(const double* __restrict__ a, int N, int Np, double* __restrict__ out)
	{
		const int i = blockIdx.x * 256 + threadIdx.x;
		if (i < N) out[2 * i] = a[i], out[2 * i + 1] = a[Np + i];
	}
} // namespace

extern "C"
{
	const char* gple_status_string(int status)
	{
		switch (status)
		{
		case GPLE_OK: return "ok";
		case GPLE_ERR_BAD_ARG: return "bad argument";
		case GPLE_ERR_TIMEOUT: return "the factorisation gave up waiting; results enqueued before the synchronisation that noticed are NaN";
		case GPLE_ERR_HIP: return "HIP runtime error";
		case GPLE_ERR_ALLOC: return "device allocation failed";
		case GPLE_ERR_STATE: return "requested output was not computed by this fit (flags), or the context was destroyed";
		case GPLE_ERR_COLLECTIVE: return "RCCL collective failed or librccl could not be resolved";
		default: return "unknown status";
		}
	}

	int gple_ctx_create(int device, void* stream, gple_ctx** out)
	{
		if (!out) return GPLE_ERR_BAD_ARG;
		*out = nullptr;
		int count = 0;
		if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return GPLE_ERR_HIP;
		if (hipSetDevice(device) != hipSuccess) return GPLE_ERR_HIP;
		gple_ctx* c = new (std::nothrow) gple_ctx;
		if (!c) return GPLE_ERR_ALLOC;
		c->device = device;
		if (stream) c->stream = static_cast<hipStream_t>(stream);
		else
		{
			if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
			{
				delete c;
				return GPLE_ERR_HIP;
			}
			c->owns_stream = true;
		}
		if (hipHostMalloc(reinterpret_cast<void**>(&c->host_scalars), 2 * SDEV_N * sizeof(double)) != hipSuccess)
		{
			if (c->owns_stream) (void)hipStreamDestroy(c->stream);
			delete c;
			return GPLE_ERR_ALLOC;
		}
		*out = c;
		return GPLE_OK;
	}
	int gple_ctx_destroy(gple_ctx* ctx)
	{
		if (!ctx) return GPLE_OK;
		if (ctx->closed.exchange(true)) return GPLE_ERR_STATE; // destroyed twice while handles keep it alive
		{
			std::lock_guard<std::mutex> lk(ctx->call_mu); // a call in flight on another thread finishes first
			(void)hipSetDevice(ctx->device);
			(void)hipStreamSynchronize(ctx->stream);
		}
		ctx_drop(ctx);
		return GPLE_OK;
	}
	int gple_ctx_trim(gple_ctx* ctx, size_t* bytes_freed)
	{
		if (!ctx) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		GPLE_HIP(ctx, hipStreamSynchronize(ctx->stream)); // nothing in flight may still be using an idle buffer
		timer_collect(ctx);
		size_t freed = 0;
		{
			std::lock_guard<std::mutex> pl(ctx->pool_mu);
			std::vector<gple_ctx::PoolEntry> kept;
			for (const gple_ctx::PoolEntry& e : ctx->pool)
				if (e.used) kept.push_back(e);
				else
				{
					(void)hipFree(e.p);
					freed += e.bytes;
				}
			ctx->pool.swap(kept);
		}
		if (bytes_freed) *bytes_freed = freed;
		return GPLE_OK;
	}
	int gple_ctx_synchronize(gple_ctx* ctx)
	{
		if (!ctx) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipStreamSynchronize(ctx->stream));
		timer_collect(ctx);
		return GPLE_OK;
	}
	const char* gple_ctx_last_error(const gple_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

	int gple_ctx_enable_timing(gple_ctx* ctx, int on)
	{
		if (!ctx) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		GPLE_HIP(ctx, hipStreamSynchronize(ctx->stream));
		timer_collect(ctx); // intervals still in flight belong to the old accumulators
		ctx->timing = on != 0;
		for (int w = 0; w < Ctx::NTIMERS; ++w) ctx->t_last[w] = ctx->t_total[w] = 0.0, ctx->t_count[w] = 0;
		return GPLE_OK;
	}
	int gple_ctx_get_prune_stats(gple_ctx* ctx, unsigned long long* contracted_blocks, unsigned long long* seen_blocks, int reset)
	{
		if (!ctx) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		unsigned long long h[2] = {0, 0};
		if (ctx->prune_stats)
		{
			GPLE_HIP(ctx, hipMemcpyAsync(h, ctx->prune_stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
			if (reset) GPLE_HIP(ctx, hipMemsetAsync(ctx->prune_stats, 0, sizeof(h), ctx->stream));
			GPLE_HIP(ctx, hipStreamSynchronize(ctx->stream));
		}
		if (contracted_blocks) *contracted_blocks = h[0];
		if (seen_blocks) *seen_blocks = h[1];
		return GPLE_OK;
	}
	int gple_ctx_get_timing(gple_ctx* ctx, gple_timer which, double* last_ms, double* total_ms, long* count)
	{
		if (!ctx || static_cast<int>(which) < 0 || static_cast<int>(which) >= Ctx::NTIMERS) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		GPLE_HIP(ctx, hipStreamSynchronize(ctx->stream));
		timer_collect(ctx);
		if (last_ms) *last_ms = ctx->t_last[which];
		if (total_ms) *total_ms = ctx->t_total[which];
		if (count) *count = ctx->t_count[which];
		return GPLE_OK;
	}

	// ---- KernelBase --------------------------------------------------------------------------------------------
	int gple_real_gram(gple_ctx* ctx, const double theta[4], const double* left, size_t R, const double* right, size_t C,
		int same_features, unsigned flags, double* K, double* dK)
	{
		if (!ctx || !theta || !K || (R && !left) || (C && !right)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (R == 0 || C == 0) return GPLE_OK;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const bool dev = flags & GPLE_IO_DEVICE;
		const SEParam p = make_se(theta[0] * theta[0], theta[3] * theta[3], theta[1], theta[2]);
		if (dev)
		{
			GPLE_HIP(ctx, launch_gram_rect(st, left, (int)R, right, (int)C, same_features, p, theta[0], theta[3], K, dK));
			return GPLE_OK;
		}
		Scratch l(ctx), r(ctx), k(ctx), dk(ctx);
		GPLE_HIP(ctx, l.get(2 * R));
		GPLE_HIP(ctx, r.get(2 * C));
		GPLE_HIP(ctx, k.get(R * C));
		if (dK) GPLE_HIP(ctx, dk.get(4 * R * C));
		GPLE_HIP(ctx, copy_in(st, l.p, left, 2 * R, false));
		GPLE_HIP(ctx, copy_in(st, r.p, right, 2 * C, false));
		GPLE_HIP(ctx, launch_gram_rect(st, l.p, (int)R, r.p, (int)C, same_features, p, theta[0], theta[3], k.p, dK ? dk.p : nullptr));
		GPLE_HIP(ctx, copy_out(st, K, k.p, R * C, false));
		if (dK) GPLE_HIP(ctx, copy_out(st, dK, dk.p, 4 * R * C, false));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		return GPLE_OK;
	}

	// ---- ComplexKernelBase ---------------------------------------------------------------------------------------
	int gple_complex_gram(gple_ctx* ctx, const double theta[8], const double* left, size_t R, const double* right, size_t C, int same_features,
		unsigned flags, double* K, double* Kt, double* dK, double* dKt)
	{
		if (!ctx || !theta || !K || (R && !left) || (C && !right)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (R == 0 || C == 0) return GPLE_OK;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		if (flags & GPLE_IO_DEVICE)
		{
			GPLE_HIP(ctx, launch_complex_gram(st, theta, left, (int)R, right, (int)C, same_features, K, Kt, dK, dKt));
			return GPLE_OK;
		}
		const size_t rc = R * C;
		Scratch l(ctx), r(ctx), k(ctx), kt(ctx), dk(ctx), dkt(ctx);
		GPLE_HIP(ctx, l.get(2 * R));
		GPLE_HIP(ctx, r.get(2 * C));
		GPLE_HIP(ctx, k.get(rc));
		if (Kt) GPLE_HIP(ctx, kt.get(2 * rc));
		if (dK) GPLE_HIP(ctx, dk.get(8 * rc));
		if (dKt) GPLE_HIP(ctx, dkt.get(16 * rc));
		GPLE_HIP(ctx, copy_in(st, l.p, left, 2 * R, false));
		GPLE_HIP(ctx, copy_in(st, r.p, right, 2 * C, false));
		GPLE_HIP(ctx, launch_complex_gram(st, theta, l.p, (int)R, r.p, (int)C, same_features, k.p, Kt ? kt.p : nullptr, dK ? dk.p : nullptr,
						  dKt ? dkt.p : nullptr));
		GPLE_HIP(ctx, copy_out(st, K, k.p, rc, false));
		if (Kt) GPLE_HIP(ctx, copy_out(st, Kt, kt.p, 2 * rc, false));
		if (dK) GPLE_HIP(ctx, copy_out(st, dK, dk.p, 8 * rc, false));
		if (dKt) GPLE_HIP(ctx, copy_out(st, dKt, dkt.p, 16 * rc, false));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		return GPLE_OK;
	}

	int gple_cutoff_factor(gple_ctx* ctx, const double* prediction, int is_complex, const double* variance, size_t M, unsigned flags,
		double* factor)
	{
		if (!ctx || (M && (!prediction || !variance || !factor))) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (M == 0) return GPLE_OK;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		if (flags & GPLE_IO_DEVICE)
		{
			GPLE_HIP(ctx, launch_cutoff(st, prediction, is_complex, variance, (int)M, factor));
			return GPLE_OK;
		}
		const size_t pl = is_complex ? 2 * M : M;
		Scratch p(ctx), v(ctx), f(ctx);
		GPLE_HIP(ctx, p.get(pl));
		GPLE_HIP(ctx, v.get(M));
		GPLE_HIP(ctx, f.get(M));
		GPLE_HIP(ctx, copy_in(st, p.p, prediction, pl, false));
		GPLE_HIP(ctx, copy_in(st, v.p, variance, M, false));
		GPLE_HIP(ctx, launch_cutoff(st, p.p, is_complex, v.p, (int)M, f.p));
		GPLE_HIP(ctx, copy_out(st, factor, f.p, M, false));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		return GPLE_OK;
	}

	// ---- TrainingKernel ----------------------------------------------------------------------------------------
	// Host side of TrainingKernel's scalar members: drains the stream, reads the raw device sums back and applies the
	// closed-form factors.  Called with ctx->call_mu held, either from gple_real_fit_create (scalars requested) or later
	// from gple_real_fit_get_scalars.
	// what follows the factorisation of a real fit: the raw sums of the scalar members (+ the derivative members); enqueue only
	static int real_fit_post(gple_ctx* ctx, gple_real_fit* f)
	{
		hipStream_t st = ctx->stream;
		const unsigned flags = f->flags;
		const double sf = f->theta[0], l0 = f->theta[1], l1 = f->theta[2], sn = f->theta[3];
		const size_t N = f->N;
		Scratch part(ctx);
		const size_t g = (N + 63) / 64;
		const bool avg = (flags & GPLE_CALC_AVERAGE) != 0;
		if (avg) // purity: v^T K1 v; its per-block partials are summed by the same launch that forms the other sums
		{
			GPLE_HIP(ctx, part.get(g * g));
			GPLE_HIP(ctx, launch_quadform_partials(st, f->Xt, f->N, purity_aux(sf, l0, l1), f->v, f->v, -1, part.p));
		}
		GPLE_HIP(ctx, launch_real_fit_sums(st, f->Xt, f->ys, f->v, f->w, f->N, f->sdev + 1, avg ? part.p : nullptr, static_cast<int>(g * g), f->sdev + 6));
		if (flags & GPLE_CALC_DERIVATIVE) GPLE_TRY(real_fit_derivatives(ctx, f, sf, l0, l1, sn, flags));
		return GPLE_OK;
	}
	static int complex_fit_post(gple_ctx* ctx, gple_complex_fit* f);

	// A give-up of the one-launch factorisation (info = -1; the stream is drained, call_mu held): the SAME fit again with one launch per panel
	// — the scheme of rounds 2-3, no workgroup of which waits for another — from the Gram on: points, scaled labels and parameters are the
	// handle's, the factorisation is in place so the Gram is regenerated (28 us at n = 4096), and the layout is the one a process started
	// with GPLE_CHOL_SCHEME=step uses, so the recovered fit has that process's bits (tests/test_gpu_chol_diag.py).  Products of the bad
	// factor that were built lazily (W, the derivative vectors) are dropped and rebuilt.
	static int recover_fit(gple_ctx* ctx, FitCommon* f)
	{
		hipStream_t st = ctx->stream;
		ctx->dag_recoveries += 1;
		for (double** p : {&f->W, &f->dv})
		{
			ctx->give_back(*p);
			*p = nullptr;
		}
		GPLE_HIP(ctx, hipMemsetAsync(f->sdev + SDEV_INFO, 0, sizeof(double), st));
		int status;
		{
			CholSchemeScope step(0);
			const int keep = ctx->chol_scheme;
			ctx->chol_scheme = -1; // (the scope above decides; a context forced to the one-launch scheme by the debug knob recovers like any other)
			status = fit_factor(ctx, f);
			ctx->chol_scheme = keep;
		}
		if (status == GPLE_OK)
			status = f->is_complex ? complex_fit_post(ctx, static_cast<gple_complex_fit*>(f)) : real_fit_post(ctx, static_cast<gple_real_fit*>(f));
		return status;
	}
	// reads the fit's scalar block back into the pinned host block (drains the stream), looks at the factorisation's info word and recovers
	// from a give-up.  GPLE_ERR_TIMEOUT: the repeated factorisation gave up as well (cannot happen — it has no waits)
	static int validate_fit(gple_ctx* ctx, const FitCommon* fc)
	{
		FitCommon* f = const_cast<FitCommon*>(fc);
		hipStream_t st = ctx->stream;
		int info_i = 0;
		for (int attempt = 0;; ++attempt)
		{
			GPLE_HIP(ctx, hipMemcpyAsync(ctx->host_scalars, f->sdev, SDEV_N * 8, hipMemcpyDeviceToHost, st));
			GPLE_HIP(ctx, hipStreamSynchronize(st));
			std::memcpy(&info_i, ctx->host_scalars + SDEV_INFO, sizeof(int));
			if (info_i >= 0) break;
			ctx->dag_giveups += 1;
			if (attempt == 1)
			{
				std::lock_guard<std::mutex> lk(ctx->mu);
				ctx->last_error = "the factorisation gave up waiting (info = -1) and so did its repetition with one launch per panel";
				return GPLE_ERR_TIMEOUT;
			}
			GPLE_TRY(recover_fit(ctx, f));
		}
		f->validated.store(true);
		return GPLE_OK;
	}
	// GPLE_ERR_TIMEOUT for the caller of a synchronising entry point when work enqueued on the fit BEFORE it was validated has consumed a factor that
	// was then found unfinished: the fit itself is good now, those earlier results are NaN
	static int report_stale_uses(gple_ctx* ctx, const FitCommon* f, bool recovered)
	{
		const int stale = f->stale_uses.exchange(0);
		if (!recovered || stale == 0) return GPLE_OK;
		std::lock_guard<std::mutex> lk(ctx->mu);
		ctx->last_error = "the one-launch factorisation gave up waiting; the fit has been repeated with one launch per panel and is valid now, but "
			+ std::to_string(stale) + " call(s) enqueued on it before this synchronisation produced NaN: repeat them";
		return GPLE_ERR_TIMEOUT;
	}

	static int real_fit_finalize(gple_ctx* ctx, gple_real_fit* f)
	{
		hipStream_t st = ctx->stream;
		(void)st;
		const long recoveries_before = ctx->dag_recoveries;
		GPLE_TRY(validate_fit(ctx, f)); // the scalar block is in the pinned host block now, of a factorisation that completed
		timer_collect(ctx);
		const unsigned flags = f->flags;
		const size_t N = f->N;
		const double sf = f->theta[0], l0 = f->theta[1], l1 = f->theta[2];
		const double* h = ctx->host_scalars;
		gple_real_fit_scalars& sc = f->sc;
		fill_nan_scalars(&sc);
		int info_i;
		std::memcpy(&info_i, h + 31, sizeof(int));
		sc.info = info_i;
		const double s = h[0];
		f->s_host = s;
		sc.rescale_factor = s;
		{
			const double within = h[5] / static_cast<double>(N); // kernel.h:169
			sc.magnitude = within < 0 ? std::sqrt(-within) : std::sqrt(within);
		}
		if (flags & GPLE_CALC_ERROR) sc.error = h[1];
		if (flags & GPLE_CALC_AVERAGE)
		{
			const double GlobalFactor = 2.0 * M_PI; // power<Dim>(2 pi), kernel.cpp:291
			const double lprod = l0 * l1;
			sc.population = GlobalFactor * (sf * sf) * lprod * h[2] / s;                 // :293
			sc.first_order_average[0] = GlobalFactor * (sf * sf) * lprod * h[3] / s;     // :308
			sc.first_order_average[1] = GlobalFactor * (sf * sf) * lprod * h[4] / s;
			const double PurityGlobal = (2.0 * M_PI) * M_PI; // PurityFactor * pi^Dim, :330
			sc.purity = PurityGlobal * h[6] / (s * s);        // :331
		}
		if (flags & GPLE_CALC_DERIVATIVE)
		{
			if (flags & GPLE_CALC_ERROR)
				for (int ip = 0; ip < 4; ++ip) sc.error_derivative[ip] = h[16 + ip]; // kernel.cpp:381-400
			if (flags & GPLE_CALC_AVERAGE)
			{
				// kernel.cpp:401-435
				const double ThisTimeFactor = (2.0 * M_PI) * (sf * sf) * (l0 * l1);
				const double ls[2] = {l0, l1};
				sc.population_derivative[0] = 0.0;
				for (int d = 0; d < 2; ++d) sc.population_derivative[1 + d] = ThisTimeFactor * (h[2] / ls[d] + h[21 + d]);
				sc.population_derivative[3] = ThisTimeFactor * h[23];
				for (double& d : sc.population_derivative) d /= s;
				// kernel.cpp:436-477
				const double PurityGlobal = (2.0 * M_PI) * M_PI;
				sc.purity_derivative[0] = 0.0;
				for (int d = 0; d < 2; ++d)
					sc.purity_derivative[1 + d] = ((h[6] / ls[d] + std::sqrt(2.0) * h[24 + d]) + 2.0 * h[26 + d]) * PurityGlobal;
				sc.purity_derivative[3] = 2.0 * PurityGlobal * h[28];
				for (double& d : sc.purity_derivative) d /= s * s;
			}
		}
		f->sc_ready = true;
		return report_stale_uses(ctx, f, ctx->dag_recoveries != recoveries_before);
	}

	// deriv_mask: which parameters' N^3 products a derivative fit forms (bit ip) — all of them, except for gple_objective_eval_part
	static int real_fit_create_masked(gple_ctx* ctx, const double theta[4], const double* X, const double* y, int y_is_complex, size_t N,
		unsigned flags, unsigned deriv_mask, gple_real_fit_scalars* scalars, gple_real_fit** out)
	{
		if (!ctx || !theta || !X || !y || !out || N == 0 || N > (1u << 20)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		*out = nullptr;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		gple_real_fit* f = new (std::nothrow) gple_real_fit;
		if (!f) return GPLE_ERR_ALLOC;
		std::memcpy(f->theta, theta, sizeof(f->theta));
		const double sf = theta[0], l0 = theta[1], l1 = theta[2], sn = theta[3];
		(void)l0, (void)l1;
		f->ps.p[0] = f->ps.p[1] = f->ps.p[2] = make_se(sf * sf, sn * sn, theta[1], theta[2]);
		f->self = (sf * sf) * (1.0 + (sn * sn) * 1.0); // KernelBase(params, col, col).get_kernel().value(), kernel.cpp:512
		f->sf = sf;
		hipStream_t st = ctx->stream;
		f->deriv_mask = deriv_mask;
		int status = fit_common(ctx, f, X, y, y_is_complex ? 2 : 1, N, flags);
		if (status == GPLE_OK)
		{
			status = real_fit_post(ctx, f);
			timer_stop(ctx, GPLE_TIMER_FIT);
		}
		if (status != GPLE_OK)
		{
			(void)hipStreamSynchronize(st);
			delete f;
			return status;
		}
		if (scalars) // no struct to fill: everything stays enqueued, gple_real_fit_get_scalars() drains the stream later
		{
			status = real_fit_finalize(ctx, f); // (nothing was enqueued on the fit before this: never GPLE_ERR_TIMEOUT for stale uses)
			if (status != GPLE_OK)
			{
				delete f;
				return status;
			}
			*scalars = f->sc;
		}
		*out = f;
		return GPLE_OK;
	}
	int gple_real_fit_create(gple_ctx* ctx, const double theta[4], const double* X, const double* y, int y_is_complex, size_t N,
		unsigned flags, gple_real_fit_scalars* scalars, gple_real_fit** out)
	{
		return real_fit_create_masked(ctx, theta, X, y, y_is_complex, N, flags, 0xFFu, scalars, out);
	}
	int gple_real_fit_get_scalars(gple_real_fit* fit, gple_real_fit_scalars* out)
	{
		if (!fit || !out) return GPLE_ERR_BAD_ARG;
		gple_ctx* ctx = fit->ctx;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		if (!fit->sc_ready) GPLE_TRY(real_fit_finalize(ctx, fit));
		*out = fit->sc;
		return GPLE_OK;
	}
	int gple_real_fit_retain(gple_real_fit* fit)
	{
		if (!fit) return GPLE_ERR_BAD_ARG;
		fit->refs.fetch_add(1);
		return GPLE_OK;
	}
	int gple_real_fit_release(gple_real_fit* fit)
	{
		if (!fit) return GPLE_OK;
		if (fit->refs.fetch_sub(1) == 1)
		{
			(void)hipSetDevice(fit->ctx->device);
			(void)hipStreamSynchronize(fit->ctx->stream); // work that still reads the fit's buffers
			delete fit;                                   // may drop the last reference on a destroyed context
		}
		return GPLE_OK;
	}
	size_t gple_real_fit_size(const gple_real_fit* fit) { return fit ? fit->N : 0; }

	int gple_real_fit_get(gple_real_fit* f, gple_real_array which, unsigned flags, double* dst)
	{
		if (!f || !dst) return GPLE_ERR_BAD_ARG;
		gple_ctx* ctx = f->ctx;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const bool dev = flags & GPLE_IO_DEVICE;
		const size_t N = f->N;
		const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
		if (!f->validated.load()) GPLE_TRY(validate_fit(ctx, f)); // (a getter drains the stream anyway)
		switch (which)
		{
		case GPLE_R_KERNEL:
		{
			Scratch k(ctx);
			GPLE_HIP(ctx, k.get(N * N));
			GPLE_HIP(ctx, launch_gram_rect(st, f->Xt, (int)N, f->Xt, (int)N, 1, f->ps.p[0], f->theta[0], f->theta[3], k.p, nullptr));
			GPLE_HIP(ctx, hipMemcpyAsync(dst, k.p, N * N * 8, kind, st));
			GPLE_HIP(ctx, hipStreamSynchronize(st));
			break;
		}
		case GPLE_R_INVERSE:
			GPLE_TRY(ensure_inverse(f));
			GPLE_HIP(ctx, hipMemcpy2DAsync(dst, N * 8, f->W, static_cast<size_t>(f->n_total) * 8, N * 8, N, kind, st));
			break;
		case GPLE_R_INVLBL: GPLE_HIP(ctx, hipMemcpyAsync(dst, f->v, N * 8, kind, st)); break;
		case GPLE_R_LABEL: GPLE_HIP(ctx, hipMemcpyAsync(dst, f->ys, N * 8, kind, st)); break;
		case GPLE_R_INVERSE_DIAG: GPLE_HIP(ctx, hipMemcpyAsync(dst, f->w, N * 8, kind, st)); break;
		case GPLE_R_INVLBL_DERIV:
			if (!f->dv) return GPLE_ERR_STATE;
			GPLE_HIP(ctx, hipMemcpy2DAsync(dst, N * 8, f->dv, static_cast<size_t>(f->n_total) * 8, N * 8, 4, kind, st));
			break;
		default: return GPLE_ERR_BAD_ARG;
		}
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		return GPLE_OK;
	}

	// ---- PredictiveKernel --------------------------------------------------------------------------------------
	// internal flag of predict_common (never part of the ABI's flag space): enqueue everything, including the D2H copies of
	// the error scalars, but leave the synchronisation and the scalar read-out to the caller
	constexpr unsigned PREDICT_NO_SYNC = 0x10000u;
	// internal: the test points lie in the context's early buffer, written on its early stream (predict_sharded with GPLE_PREDICT_OVERLAP=1)
	constexpr unsigned PREDICT_XS_EARLY = 0x20000u;
	static void predict_scalars_from_host(gple_ctx* ctx, bool has_labels, bool want_deriv, bool cplx, gple_predict_scalars* scalars)
	{
		if (!scalars) return;
		if (has_labels) scalars->error = ctx->host_scalars[HS_PRED_ERR];
		if (want_deriv)
			for (int ip = 0; ip < (cplx ? 8 : 4); ++ip) scalars->error_derivative[ip] = ctx->host_scalars[HS_PRED_DERIV + ip];
	}
	static bool predict_pruning()
	{
		static const bool on = [] {
			const char* e = getenv("GPLE_PREDICT_PRUNE");
			return !e || atoi(e) != 0;
		}();
		return on;
	}
	static int predict_pass(gple_ctx* ctx, const FitCommon* f, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* scalars, bool* repeat);
	static int predict_common(gple_ctx* ctx, const FitCommon* f, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* scalars)
	{
		if (scalars)
		{
			scalars->error = nan_();
			for (double& d : scalars->error_derivative) d = nan_();
		}
		if (M == 0) return GPLE_OK;
		if ((flags & GPLE_CALC_DERIVATIVE) && labels && !f->dv) return GPLE_ERR_STATE; // needs a fit built with GPLE_CALC_DERIVATIVE
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		// A predict that drains the stream on a fit the host has not looked at yet looks at it (validate_fit): if the one-launch factorisation had
		// given up, the fit has been repeated with a launch per panel by then and this predict — whose first pass produced NaN — runs again.
		bool repeat = false;
		int status = predict_pass(ctx, f, Xs, M, flags, labels, prediction, variance, cutoff_prediction, scalars, &repeat);
		if (status == GPLE_OK && repeat) status = predict_pass(ctx, f, Xs, M, flags, labels, prediction, variance, cutoff_prediction, scalars, &repeat);
		return status;
	}
	static int predict_pass(gple_ctx* ctx, const FitCommon* f, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* scalars, bool* repeat)
	{
		*repeat = false;
		const bool unvalidated = !f->validated.load();
		const bool want_deriv = (flags & GPLE_CALC_DERIVATIVE) && labels;
		hipStream_t st = ctx->stream;
		const bool dev = flags & GPLE_IO_DEVICE;
		const bool cplx = f->is_complex;
		const int Mi = static_cast<int>(M);
		const int Mh = static_cast<int>(round_up(M, 128));
		const int m_rows = cplx ? 2 * Mh : Mh;
		const size_t ow = cplx ? 2 : 1; // doubles per prediction entry
		Scratch xs(ctx), q(ctx), mu(ctx), lab(ctx), o_mean(ctx), o_var(ctx), o_cut(ctx), epart(ctx), dacc(ctx), dpart(ctx);
		timer_start(ctx, GPLE_TIMER_PREDICT);
		const double* xs_dev = Xs;
		const bool small_host = !dev && M <= FEW_HOST_POINTS && !(flags & PREDICT_NO_SYNC); // (a deferred call must not leave its inputs in the shared block)
		if (small_host)
		{
			std::memcpy(ctx->host_scalars + HS_FEW_XS, Xs, 2 * M * sizeof(double));
			xs_dev = ctx->host_scalars + HS_FEW_XS;
		}
		else if (!dev)
		{
			GPLE_HIP(ctx, xs.get(2 * M));
			GPLE_HIP(ctx, copy_in(st, xs.p, Xs, 2 * M, false));
			xs_dev = xs.p;
		}
		GPLE_HIP(ctx, q.get(m_rows));
		GPLE_HIP(ctx, mu.get(m_rows));
		PredictArgs a{};
		a.Xs = xs_dev, a.M = Mi, a.m_rows = m_rows, a.m_split = cplx ? Mh : m_rows;
		a.Xt = f->Xt, a.N = f->N, a.n_total = f->n_total, a.n_split = cplx ? f->Np : f->n_total;
		a.T = f->T, a.ldt = f->n_total, a.v = f->v, a.q = q.p, a.mu = mu.p, a.ps = f->ps;
		if (!(flags & GPLE_PREDICT_FULL) && predict_pruning())
		{
			// |k*|^2 below this cannot move the variance (gple_predict.hip, Prune): lambda_min(K) >= amp n2, k(x*,x*) = self.
			// Complex GP in its [Re; Im] embedding: the covariance is a valid (positive semi-definite) cross-covariance of two
			// squared-exponential processes plus s^2 sn^2 / 2 on the whole diagonal (p[0].amp p[0].n2 = p[2].amp p[2].n2), and the
			// variance subtracts the contractions of TWO typed rows per point: half the budget each.
			a.prune_thr = std::ldexp(f->self * f->ps.p[0].amp * f->ps.p[0].n2, cplx ? -57 : -56);
			if (!ctx->prune_stats)
			{
				GPLE_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->prune_stats), 4 * sizeof(unsigned long long))); // [2]: the work-queue counter
				GPLE_HIP(ctx, hipMemsetAsync(ctx->prune_stats, 0, 4 * sizeof(unsigned long long), st));
			}
			a.prune_stats = ctx->prune_stats;
		}
		if (want_deriv)
		{
			GPLE_HIP(ctx, dacc.get((cplx ? 15 : 7) * static_cast<size_t>(m_rows)));
			a.dv = f->dv, a.dacc = dacc.p;
			a.complex_deriv = cplx ? 1 : 0;
			if (cplx) std::memcpy(a.dspec, f->dspec, sizeof(a.dspec));
		}
		// How much of the variance contraction the caller's outputs need (GPLE_PREDICT_SKIP=0: always all of it, for A/B runs).  The objective
		// of opt.cpp:441-482 asks for Error (+ ErrorDerivatives) only: Error uses the UNCUT mean (kernel.cpp:522) — no contraction at all, every
		// value-only evaluation of the derivative-free searches; ErrorDerivatives use the cut one (:527), where the variance only decides the
		// cut-off factor — and a point with |mu|^2 >= 4 k(x*,x*) >= 4 var has factor 1 (kernel.h:301-332) whatever q is.
		static const bool skip_ok = [] {
			const char* e = getenv("GPLE_PREDICT_SKIP");
			return e == nullptr || atoi(e) != 0;
		}();
		if (skip_ok && !variance && !cutoff_prediction)
		{
			if (!want_deriv) a.mean_only = 1;
			else
			{
				a.cut_thr = 4.0 * f->self;
				if (!ctx->prune_stats)
				{
					GPLE_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->prune_stats), 4 * sizeof(unsigned long long)));
					GPLE_HIP(ctx, hipMemsetAsync(ctx->prune_stats, 0, 4 * sizeof(unsigned long long), st));
				}
				a.prune_stats = ctx->prune_stats;
			}
		}
		int chunk_rows = 0;
		bool few_rows = false, late_contraction = false, finished = false;
		Scratch kstar(ctx);
		if (predict_is_few(a)) // one-point predicts of the reference's callers: no padding, no K*, no GEMM
		{
			GPLE_HIP(ctx, kstar.get(predict_few_scratch_doubles(a)));
			GPLE_HIP(ctx, launch_predict_few(st, a, kstar.p));
		}
		else
		{
			GPLE_HIP(ctx, kstar.get(predict_scratch_doubles(a, &chunk_rows, &few_rows)));
			late_contraction = true;
		}
		const double* lab_dev = labels;
		if (labels && small_host)
		{
			std::memcpy(ctx->host_scalars + HS_FEW_LAB, labels, ow * M * sizeof(double));
			lab_dev = ctx->host_scalars + HS_FEW_LAB;
		}
		else if (labels && !dev)
		{
			GPLE_HIP(ctx, lab.get(ow * M));
			GPLE_HIP(ctx, copy_in(st, lab.p, labels, ow * M, false));
			lab_dev = lab.p;
		}
		double *d_mean = prediction, *d_var = variance, *d_cut = cutoff_prediction;
		if (small_host)
		{
			if (prediction) d_mean = ctx->host_scalars + HS_FEW_MEAN;
			if (variance) d_var = ctx->host_scalars + HS_FEW_VAR;
			if (cutoff_prediction) d_cut = ctx->host_scalars + HS_FEW_CUT;
		}
		else if (!dev)
		{
			if (prediction)
			{
				GPLE_HIP(ctx, o_mean.get(ow * M));
				d_mean = o_mean.p;
			}
			if (variance)
			{
				GPLE_HIP(ctx, o_var.get(M));
				d_var = o_var.p;
			}
			if (cutoff_prediction)
			{
				GPLE_HIP(ctx, o_cut.get(ow * M));
				d_cut = o_cut.p;
			}
		}
		if (late_contraction)
		{
			// Beside the fit (GPLE_PREDICT_OVERLAP=1): the fit is still in flight, the first rows of its T are final behind an event it left, the points are
			// device memory that the early stream may read (the caller's, complete before the fit was enqueued — the extra contract of the switch —
			// or the shard predict_sharded put into the early buffer)
			const bool early_xs = flags & PREDICT_XS_EARLY;
			if ((dev || early_xs) && unvalidated && !labels && !want_deriv && ctx->fit_early_event && predict_overlap_applicable(ctx, a, f->early_rows))
			{
				double* xs_early = nullptr;
				GPLE_HIP(ctx, predict_overlap_prepare(ctx, a, early_xs ? 2 * M : 0, &xs_early));
				GPLE_HIP(ctx, launch_predict_overlapped(ctx, st, a, early_xs ? 2 * M : 0, f->early_rows, ctx->fit_early_event, early_xs ? ctx->early_points : nullptr));
			}
			else
			{
			if (early_xs) GPLE_HIP(ctx, hipStreamWaitEvent(st, ctx->early_points, 0)); // (a second pass behind a recovered fit: the points still lie in the early buffer)
			// (the output buffers are known by now: a kernel that holds a row's mean and q together may write the outputs itself — predict_fused256_kernel)
			if (!cplx && !labels && !want_deriv) a.fin_sdev = f->sdev, a.fin_self = f->self, a.fin_mean = d_mean, a.fin_var = d_var, a.fin_cut = d_cut;
			GPLE_HIP(ctx, launch_predict_q(ctx, st, a, kstar.p, chunk_rows, few_rows, &finished));
			if (early_xs)
			{
				GPLE_HIP(ctx, hipEventRecord(ctx->early_free, st));
				ctx->early_free_pending = true;
			}
			}
		}
		const int nblk = (Mi + 255) / 256;
		double* err_part = nullptr;
		if (labels)
		{
			GPLE_HIP(ctx, epart.get(nblk + 1));
			err_part = epart.p;
		}
		if (finished) {}
		else if (cplx)
			GPLE_HIP(ctx, launch_predict_finish_complex(st, q.p, mu.p, Mi, Mh, f->self, f->sdev, lab_dev, d_mean, d_var, d_cut, err_part));
		else
			GPLE_HIP(ctx, launch_predict_finish_real(st, q.p, mu.p, Mi, f->self, f->sdev, lab_dev, d_mean, d_var, d_cut, err_part));
		if (labels)
		{
			GPLE_HIP(ctx, launch_sum(st, err_part, nblk, err_part + nblk));
			GPLE_HIP(ctx, hipMemcpyAsync(ctx->host_scalars + HS_PRED_ERR, err_part + nblk, 8, hipMemcpyDeviceToHost, st));
		}
		if (want_deriv)
		{
			GPLE_HIP(ctx, dpart.get(8 * static_cast<size_t>(nblk) + 8));
			if (cplx)
				GPLE_HIP(ctx, launch_predict_deriv_finish_complex(st, dacc.p, m_rows, Mh, q.p, Mi, f->self, f->s0, f->sdev, lab_dev, dpart.p,
								  dpart.p + 8 * nblk));
			else
				GPLE_HIP(ctx, launch_predict_deriv_finish_real(st, dacc.p, m_rows, q.p, Mi, f->self, f->sf, f->sdev, lab_dev, dpart.p, dpart.p + 8 * nblk));
			GPLE_HIP(ctx, hipMemcpyAsync(ctx->host_scalars + HS_PRED_DERIV, dpart.p + 8 * nblk, 8 * 8, hipMemcpyDeviceToHost, st));
		}
		if (!dev && !small_host)
		{
			GPLE_HIP(ctx, copy_out(st, prediction, d_mean, ow * M, false));
			GPLE_HIP(ctx, copy_out(st, variance, d_var, M, false));
			GPLE_HIP(ctx, copy_out(st, cutoff_prediction, d_cut, ow * M, false));
		}
		timer_stop(ctx, GPLE_TIMER_PREDICT);
		// host outputs (and the error scalar) need the stream drained; device-pointer calls without labels stay asynchronous
		// (pooled scratch is only ever reused by later work on this same stream, which the stream orders)
		if ((flags & PREDICT_NO_SYNC) || !(!dev || labels))
		{
			// enqueued only (internal NO_SYNC: the caller drains the stream and reads the scalars itself; device pointers without labels): if the
			// factorisation turns out to have given up, these outputs are NaN and the fit's next synchronising call says so (report_stale_uses)
			if (unvalidated) f->stale_uses.fetch_add(1);
			if (flags & PREDICT_NO_SYNC) return GPLE_OK;
		}
		else
		{
			GPLE_HIP(ctx, hipStreamSynchronize(st));
			timer_collect(ctx);
			if (unvalidated)
			{
				const long before = ctx->dag_recoveries;
				GPLE_TRY(validate_fit(ctx, f));
				if (ctx->dag_recoveries != before)
				{
					*repeat = true; // the fit is good now; what this pass computed is NaN
					return GPLE_OK;
				}
			}
		}
		if (small_host) // the kernels wrote into the pinned block
		{
			if (prediction) std::memcpy(prediction, ctx->host_scalars + HS_FEW_MEAN, ow * M * sizeof(double));
			if (variance) std::memcpy(variance, ctx->host_scalars + HS_FEW_VAR, M * sizeof(double));
			if (cutoff_prediction) std::memcpy(cutoff_prediction, ctx->host_scalars + HS_FEW_CUT, ow * M * sizeof(double));
		}
		predict_scalars_from_host(ctx, labels != nullptr, want_deriv, cplx, scalars);
		return GPLE_OK;
	}

	// One-point predict with host pointers and no labels — what main.cpp:83,94 issues per call, from as many threads as TBB has.
	// The calls on one context are serialised anyway (one stream); instead of queueing behind each other, the requests that
	// pile up while a predict is in flight ride together on the next one (up to 16 typed rows: the few-points path costs the
	// same for 16 points as for one).  Every caller still returns with its own result; a single-threaded caller sees a batch of one.
	static int predict_point_combined(gple_ctx* ctx, const FitCommon* f, const double* x, double* prediction, double* variance,
		double* cutoff_prediction)
	{
		const size_t ow = f->is_complex ? 2 : 1, cap = FEW_HOST_POINTS / ow;
		FitCommon::PointRequest r{x, prediction, variance, cutoff_prediction, GPLE_OK, false};
		std::unique_lock<std::mutex> lk(f->point_mu);
		f->point_pending.push_back(&r);
		for (;;)
		{
			if (r.done) return r.status;
			if (f->point_leader)
			{
				f->point_cv.wait(lk);
				continue;
			}
			f->point_leader = true; // serve the oldest requests (this thread's own is among the pending ones; FIFO gets to it)
			const size_t nb = std::min(cap, f->point_pending.size());
			FitCommon::PointRequest* batch[FEW_HOST_POINTS];
			double xs[2 * FEW_HOST_POINTS], mean[2 * FEW_HOST_POINTS], var[FEW_HOST_POINTS], cut[2 * FEW_HOST_POINTS];
			for (size_t i = 0; i < nb; ++i)
			{
				batch[i] = f->point_pending[i];
				xs[2 * i] = batch[i]->x[0], xs[2 * i + 1] = batch[i]->x[1];
			}
			f->point_pending.erase(f->point_pending.begin(), f->point_pending.begin() + static_cast<long>(nb));
			lk.unlock();
			const int st = predict_common(ctx, f, xs, nb, 0u, nullptr, mean, var, cut, nullptr);
			lk.lock();
			for (size_t i = 0; i < nb; ++i)
			{
				FitCommon::PointRequest* q = batch[i];
				if (st == GPLE_OK)
				{
					if (q->mean) std::memcpy(q->mean, mean + ow * i, ow * sizeof(double));
					if (q->var) *q->var = var[i];
					if (q->cut) std::memcpy(q->cut, cut + ow * i, ow * sizeof(double));
				}
				q->status = st;
				q->done = true;
			}
			f->point_leader = false;
			f->point_cv.notify_all();
		}
	}
	static bool point_combining()
	{
		static const bool on = [] {
			const char* e = getenv("GPLE_POINT_COMBINE");
			return !e || atoi(e) != 0;
		}();
		return on;
	}

	int gple_real_predict(gple_ctx* ctx, const gple_real_fit* fit, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* scalars)
	{
		if (!ctx || !fit || (M && !Xs)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (M == 1 && !labels && !(flags & GPLE_IO_DEVICE) && point_combining())
		{
			if (scalars)
			{
				scalars->error = nan_();
				for (double& d : scalars->error_derivative) d = nan_();
			}
			return predict_point_combined(ctx, fit, Xs, prediction, variance, cutoff_prediction);
		}
		return predict_common(ctx, fit, Xs, M, flags, labels, prediction, variance, cutoff_prediction, scalars);
	}

	// Host side of TrainingComplexKernel's scalar members (see real_fit_finalize).
	static int complex_fit_finalize(gple_ctx* ctx, gple_complex_fit* f)
	{
		const long recoveries_before = ctx->dag_recoveries;
		GPLE_TRY(validate_fit(ctx, f)); // the scalar block is in the pinned host block now, of a factorisation that completed
		timer_collect(ctx);
		const unsigned flags = f->flags;
		const size_t N = f->N;
		const double* theta = f->theta;
		const double s0 = theta[0];
		double purity_sums[6];
		const double* h = ctx->host_scalars;
		for (int q = 0; q < 6; ++q) purity_sums[q] = h[8 + q];
		gple_complex_fit_scalars& sc = f->sc;
		fill_nan_scalars(&sc);
		int info_i;
		std::memcpy(&info_i, h + 31, sizeof(int));
		sc.info = info_i;
		const double s = h[0];
		f->s_host = s;
		sc.rescale_factor = s;
		{
			const double within = h[2] / static_cast<double>(N); // complex_kernel.h:194
			sc.magnitude = within < 0 ? std::sqrt(-within) : std::sqrt(within);
		}
		if (flags & GPLE_CALC_ERROR) sc.error = h[1];
		if (flags & GPLE_CALC_AVERAGE)
		{
			const double GlobalFactor = (2.0 * M_PI) * 2.0 * M_PI; // PurityFactor * 2 pi^Dim, complex_kernel.cpp:369
			const double ThisTimeFactor = GlobalFactor * ((s0 * s0) * (s0 * s0));
			// weights w = 2 v  ->  every quadratic form carries 1/4
			const double qf = 0.25 * (2.0 * purity_sums[0] + 2.0 * purity_sums[1] + 2.0 * (purity_sums[2] + purity_sums[3])
				+ 4.0 * (purity_sums[4] + purity_sums[5]));
			sc.purity = ThisTimeFactor * qf / (s * s); // :373
		}
		if (flags & GPLE_CALC_DERIVATIVE)
		{
			if (flags & GPLE_CALC_ERROR)
				for (int ip = 0; ip < 8; ++ip) sc.error_derivative[ip] = h[32 + ip];
			if (flags & GPLE_CALC_AVERAGE) complex_purity_derivative(theta, h, s, sc.purity_derivative);
		}
		f->sc_ready = true;
		return report_stale_uses(ctx, f, ctx->dag_recoveries != recoveries_before);
	}
	// what follows the factorisation of a complex fit: raw sums, purity quadratic forms (+ the derivative members); enqueue only
	static int complex_fit_post(gple_ctx* ctx, gple_complex_fit* f)
	{
		hipStream_t st = ctx->stream;
		const unsigned flags = f->flags;
		const double* theta = f->theta;
		const size_t N = f->N;
		const double sR = theta[1], lR0 = theta[2], lR1 = theta[3], sI = theta[4], lI0 = theta[5], lI1 = theta[6];
		hipLaunchKernelGGL(complex_fit_sums_kernel, dim3(1), dim3(1024), 0, st, f->ys, f->v, f->w, f->wx, f->N, f->Np, f->sdev + 1);
		GPLE_HIP(ctx, hipGetLastError());
		if (flags & GPLE_CALC_AVERAGE)
		{
			// purity quadratic forms in the [Re; Im] weights w = 2 v (complex_kernel.cpp:287-377):
			// Re(v^H K1 v) + Re(v^T K2 v) = 2 vr'KR'vr + 2 vi'KI'vi + 2 (vr'KC'vr + vi'KC'vi) + 4 vr'(KRC + KIC)vi
			const ComplexAux a = complex_aux(theta);
			Scratch part(ctx);
			const size_t g = (N + 63) / 64;
			GPLE_HIP(ctx, part.get(g * g));
			const SEParam aR = purity_aux(sR, lR0, lR1), aI = purity_aux(sI, lI0, lI1), aC = purity_aux(a.sC, a.lC[0], a.lC[1]);
			const double *wr = f->v, *wi = f->v + f->Np;
			const SEParam ks[6] = {aR, aI, aC, aC, a.k[3], a.k[4]};
			const double* as[6] = {wr, wi, wr, wi, wr, wr};
			const double* bs[6] = {wr, wi, wr, wi, wi, wi};
			for (int q = 0; q < 6; ++q) GPLE_HIP(ctx, launch_quadform(st, f->Xt, f->N, ks[q], as[q], bs[q], -1, part.p, f->sdev + 8 + q));
		}
		if (flags & GPLE_CALC_DERIVATIVE) GPLE_TRY(complex_fit_derivatives(ctx, f, theta, flags));
		return GPLE_OK;
	}

	// ---- TrainingComplexKernel ------------------------------------------------------------------------------
	static int complex_fit_create_masked(gple_ctx* ctx, const double theta[8], const double* X, const double* y, size_t N, unsigned flags,
		unsigned deriv_mask, gple_complex_fit_scalars* scalars, gple_complex_fit** out)
	{
		if (!ctx || !theta || !X || !y || !out || N == 0 || N > (1u << 19)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		*out = nullptr;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		gple_complex_fit* f = new (std::nothrow) gple_complex_fit;
		if (!f) return GPLE_ERR_ALLOC;
		f->is_complex = true;
		std::memcpy(f->theta, theta, sizeof(f->theta));
		const double s0 = theta[0], sR = theta[1], lR0 = theta[2], lR1 = theta[3], sI = theta[4], lI0 = theta[5], lI1 = theta[6],
					 sn = theta[7];
		// correlation kernel parameters, complex_kernel.cpp:144-157
		const double ss0 = lR0 * lR0 + lI0 * lI0, ss1 = lR1 * lR1 + lI1 * lI1;
		const double sC = std::sqrt(sR * sI * ((2.0 * lR0 * lI0 / ss0) * (2.0 * lR1 * lI1 / ss1)));
		const double lC0 = std::sqrt(ss0 / 2.0), lC1 = std::sqrt(ss1 / 2.0);
		const double m2 = s0 * s0;
		// blocks of the real covariance of [Re; Im]: Cxx = s^2 (kR + sn^2/2 d), Cyy = s^2 (kI + sn^2/2 d), Cxy = s^2 kC
		f->ps.p[0] = make_se(m2 * (sR * sR), (sn * sn) / (2.0 * sR * sR), lR0, lR1);
		f->ps.p[1] = make_se(m2 * (sC * sC), 0.0, lC0, lC1);
		f->ps.p[2] = make_se(m2 * (sI * sI), (sn * sn) / (2.0 * sI * sI), lI0, lI1);
		f->self = m2 * (sR * sR * (1.0 + 0.0) + sI * sI * (1.0 + 0.0) + sn * sn * 1.0); // complex_kernel.cpp:632
		f->s0 = s0;
		build_dspecs(theta, f->dspec);
		hipStream_t st = ctx->stream;
		f->deriv_mask = deriv_mask;
		int status = fit_common(ctx, f, X, y, 2, N, flags);
		if (status == GPLE_OK)
		{
			status = complex_fit_post(ctx, f);
			timer_stop(ctx, GPLE_TIMER_FIT);
		}
		if (status != GPLE_OK)
		{
			(void)hipStreamSynchronize(st);
			delete f;
			return status;
		}
		if (scalars) // no struct to fill: everything stays enqueued, gple_complex_fit_get_scalars() drains the stream later
		{
			status = complex_fit_finalize(ctx, f);
			if (status != GPLE_OK)
			{
				delete f;
				return status;
			}
			*scalars = f->sc;
		}
		*out = f;
		return GPLE_OK;
	}
	int gple_complex_fit_create(gple_ctx* ctx, const double theta[8], const double* X, const double* y, size_t N, unsigned flags,
		gple_complex_fit_scalars* scalars, gple_complex_fit** out)
	{
		return complex_fit_create_masked(ctx, theta, X, y, N, flags, 0xFFu, scalars, out);
	}
	int gple_complex_fit_get_scalars(gple_complex_fit* fit, gple_complex_fit_scalars* out)
	{
		if (!fit || !out) return GPLE_ERR_BAD_ARG;
		gple_ctx* ctx = fit->ctx;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		if (!fit->sc_ready) GPLE_TRY(complex_fit_finalize(ctx, fit));
		*out = fit->sc;
		return GPLE_OK;
	}
	int gple_complex_fit_retain(gple_complex_fit* fit)
	{
		if (!fit) return GPLE_ERR_BAD_ARG;
		fit->refs.fetch_add(1);
		return GPLE_OK;
	}
	int gple_complex_fit_release(gple_complex_fit* fit)
	{
		if (!fit) return GPLE_OK;
		if (fit->refs.fetch_sub(1) == 1)
		{
			(void)hipSetDevice(fit->ctx->device);
			(void)hipStreamSynchronize(fit->ctx->stream); // work that still reads the fit's buffers
			delete fit;                                   // may drop the last reference on a destroyed context
		}
		return GPLE_OK;
	}
	size_t gple_complex_fit_size(const gple_complex_fit* fit) { return fit ? fit->N : 0; }

	int gple_complex_fit_get(gple_complex_fit* f, gple_complex_array which, unsigned flags, double* dst)
	{
		if (!f || !dst) return GPLE_ERR_BAD_ARG;
		gple_ctx* ctx = f->ctx;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const bool dev = flags & GPLE_IO_DEVICE;
		const size_t N = f->N;
		const int Np = f->Np, nt = f->n_total;
		const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
		Scratch tmp(ctx), big(ctx);
		const dim3 grid2((N + 63) / 64, (N + 3) / 4), blk(256);
		if (!f->validated.load()) GPLE_TRY(validate_fit(ctx, f)); // (a getter drains the stream anyway)
		switch (which)
		{
		case GPLE_C_KERNEL:
		case GPLE_C_PSEUDO:
		{
			const size_t len = which == GPLE_C_KERNEL ? N * N : 2 * N * N;
			GPLE_HIP(ctx, big.get(static_cast<size_t>(nt) * nt));
			GPLE_HIP(ctx, tmp.get(len));
			GPLE_HIP(ctx, launch_gram_train(st, f->Xt, f->N, Np, nt, f->ps, big.p, nt));
			hipLaunchKernelGGL(complex_kernels_kernel, grid2, blk, 0, st, big.p, nt, (int)N, Np, which == GPLE_C_KERNEL ? 0 : 1, tmp.p);
			GPLE_HIP(ctx, hipGetLastError());
			GPLE_HIP(ctx, hipMemcpyAsync(dst, tmp.p, len * 8, kind, st));
			break;
		}
		case GPLE_C_UPPER_LEFT:
		case GPLE_C_LOWER_LEFT:
			GPLE_TRY(ensure_inverse(f));
			GPLE_HIP(ctx, tmp.get(2 * N * N));
			hipLaunchKernelGGL(complex_blocks_kernel, grid2, blk, 0, st, f->W, nt, (int)N, Np, which == GPLE_C_UPPER_LEFT ? 0 : 1, tmp.p);
			GPLE_HIP(ctx, hipGetLastError());
			GPLE_HIP(ctx, hipMemcpyAsync(dst, tmp.p, 2 * N * N * 8, kind, st));
			break;
		case GPLE_C_INVLBL:
			GPLE_HIP(ctx, tmp.get(2 * N));
			hipLaunchKernelGGL(halve_pair_kernel, dim3((N + 255) / 256), blk, 0, st, f->v, (int)N, Np, tmp.p);
			GPLE_HIP(ctx, hipGetLastError());
			GPLE_HIP(ctx, hipMemcpyAsync(dst, tmp.p, 2 * N * 8, kind, st));
			break;
		case GPLE_C_LABEL:
			GPLE_HIP(ctx, tmp.get(2 * N));
			hipLaunchKernelGGL(pair_kernel, dim3((N + 255) / 256), blk, 0, st, f->ys, (int)N, Np, tmp.p);
			GPLE_HIP(ctx, hipGetLastError());
			GPLE_HIP(ctx, hipMemcpyAsync(dst, tmp.p, 2 * N * 8, kind, st));
			break;
		case GPLE_C_INVLBL_DERIV:
			if (!f->dv) return GPLE_ERR_STATE;
			GPLE_HIP(ctx, tmp.get(16 * N));
			for (int ip = 0; ip < 8; ++ip)
			{
				hipLaunchKernelGGL(halve_pair_kernel, dim3((N + 255) / 256), blk, 0, st, f->dv + static_cast<size_t>(ip) * nt, (int)N, Np, tmp.p + 2 * ip * N);
				GPLE_HIP(ctx, hipGetLastError());
			}
			GPLE_HIP(ctx, hipMemcpyAsync(dst, tmp.p, 16 * N * 8, kind, st));
			break;
		default: return GPLE_ERR_BAD_ARG;
		}
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		return GPLE_OK;
	}

	int gple_complex_predict(gple_ctx* ctx, const gple_complex_fit* fit, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* scalars)
	{
		if (!ctx || !fit || (M && !Xs)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (M == 1 && !labels && !(flags & GPLE_IO_DEVICE) && point_combining())
		{
			if (scalars)
			{
				scalars->error = nan_();
				for (double& d : scalars->error_derivative) d = nan_();
			}
			return predict_point_combined(ctx, fit, Xs, prediction, variance, cutoff_prediction);
		}
		return predict_common(ctx, fit, Xs, M, flags, labels, prediction, variance, cutoff_prediction, scalars);
	}

	// ---- grid-sharded predict: slice -> predict -> ncclAllGather -> unpack ------------------------------------------------------
	int gple_shard_bounds(size_t M, int rank, int world, size_t* lo, size_t* hi, size_t* per)
	{
		if (world <= 0 || rank < 0 || rank >= world) return GPLE_ERR_BAD_ARG;
		const size_t p = M ? (M + static_cast<size_t>(world) - 1) / static_cast<size_t>(world) : 0; // parallel.shard_bounds
		const size_t l = std::min(M, static_cast<size_t>(rank) * p), h = std::min(M, l + p);
		if (lo) *lo = l;
		if (hi) *hi = h;
		if (per) *per = p;
		return GPLE_OK;
	}
	namespace
	{
		// RCCL's C entry point, resolved lazily: from the process image when the caller links librccl (their ncclComm_t then
		// belongs to that very library), else from librccl.so.1.  No RCCL header or link dependency in this library.
		using allgather_fn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);
		std::atomic<allgather_fn> allgather_override{nullptr};
		allgather_fn resolve_allgather()
		{
			if (allgather_fn o = allgather_override.load()) return o;
			static allgather_fn fn = [] {
				void* sym = nullptr;
				if (const char* named = getenv("GPLE_RCCL_LIBRARY")) // the caller names the library its ncclComm_t comes from: nothing else is tried
				{
					if (void* h = dlopen(named, RTLD_NOW | RTLD_GLOBAL)) sym = dlsym(h, "ncclAllGather");
					return reinterpret_cast<allgather_fn>(sym);
				}
				sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
				if (!sym)
					for (const char* name : {"librccl.so.1", "librccl.so"})
						if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))
							if ((sym = dlsym(h, "ncclAllGather"))) break;
				return reinterpret_cast<allgather_fn>(sym);
			}();
			return fn;
		}
		// Block-cyclic deal of the test points.  Contiguous slices would balance the full contraction just as well, but with far-row
		// pruning (the default) the live blocks of a phase-space grid sit in one corner of it and a contiguous slice holds anything
		// between all and none of them.  Plain deal: 128-point block b belongs to rank b % world (local block b / world).  Weighted deal
		// (plans that give the ranks unequal shares of an element): out of every cycle of S = sum(w) consecutive blocks rank
		// r takes the w[r] blocks cum[r] .. cum[r] + w[r] - 1; the plain deal is w = 1 for everyone.
		constexpr size_t SHARD_BLOCK = 128;
		constexpr int DEAL_MAX_WORLD = 64;
		struct Deal
		{
			int world, S;
			int cum[DEAL_MAX_WORLD + 1];
			__host__ __device__ int weight(int r) const { return cum[r + 1] - cum[r]; }
			__host__ __device__ int owner(size_t b) const
			{
				const int p = static_cast<int>(b % S);
				int r = 0;
				while (cum[r + 1] <= p) ++r;
				return r;
			}
			// block `lb` of rank r's share -> block of the grid
			__host__ __device__ size_t global_block(int r, size_t lb) const { return (lb / weight(r)) * S + cum[r] + lb % weight(r); }
			// block b of the grid (owned by r) -> block of r's share
			__host__ __device__ size_t local_block(int r, size_t b) const { return (b / S) * weight(r) + (b % S - cum[r]); }
			size_t blocks_of(int r, size_t nblocks) const
			{
				const size_t rem = nblocks % S, w = static_cast<size_t>(weight(r)), c = static_cast<size_t>(cum[r]);
				return (nblocks / S) * w + (rem > c ? std::min(rem - c, w) : 0);
			}
		};
		bool make_deal(int world, const int* weights, Deal& d)
		{
			if (world < 1 || world > DEAL_MAX_WORLD) return false;
			d.world = world, d.cum[0] = 0;
			for (int r = 0; r < world; ++r)
			{
				const int w = weights ? weights[r] : 1;
				if (w < 0 || w > (1 << 20)) return false;
				d.cum[r + 1] = d.cum[r] + w;
			}
			d.S = d.cum[world];
			return d.S > 0;
		}
		__global__ void __launch_bounds__(256) shard_points_kernel(const double* __restrict__ Xs, size_t M, int rank, Deal deal, size_t n_local,
			double* __restrict__ out)
		{
			const size_t j = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
			if (j >= n_local) return;
			const size_t i = deal.global_block(rank, j / SHARD_BLOCK) * SHARD_BLOCK + j % SHARD_BLOCK;
			out[2 * j] = Xs[2 * i], out[2 * j + 1] = Xs[2 * i + 1];
		}
		// gathered[r][...] (world blocks of (2 ow + 1) * per doubles: mean | var | cut of rank r's points) -> full-length outputs
		__global__ void __launch_bounds__(256) unshard_kernel(const double* __restrict__ g, size_t per, int ow, size_t M, Deal deal, double* __restrict__ mean,
			double* __restrict__ var, double* __restrict__ cut)
		{
			const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
			if (i >= M) return;
			const size_t b = i / SHARD_BLOCK;
			const int r = deal.owner(b);
			const size_t q = deal.local_block(r, b) * SHARD_BLOCK + i % SHARD_BLOCK;
			const double* __restrict__ blk = g + r * (2 * ow + 1) * per;
			for (int k = 0; k < ow; ++k)
			{
				if (mean) mean[ow * i + k] = blk[ow * q + k];
				if (cut) cut[ow * i + k] = blk[(ow + 1) * per + ow * q + k];
			}
			if (var) var[i] = blk[ow * per + q];
		}
		// rank's share under a deal: its number of points and the padded share length every rank allocates
		void deal_counts(const Deal& d, size_t M, int rank, size_t& n_local, size_t& per)
		{
			const size_t nblocks = (M + SHARD_BLOCK - 1) / SHARD_BLOCK;
			size_t most = 0;
			for (int r = 0; r < d.world; ++r) most = std::max(most, d.blocks_of(r, nblocks));
			per = most * SHARD_BLOCK;
			n_local = d.blocks_of(rank, nblocks) * SHARD_BLOCK;
			if (nblocks && d.owner(nblocks - 1) == rank) n_local -= nblocks * SHARD_BLOCK - M; // owner of the short block: it is the last of its share
		}
	} // namespace
	// Rehearsal transport for ONE rank of a world that is not there (bench.py --emulate-rank r/P on a one-GPU box): ncclAllGather's signature;
	// `comm` is not a communicator but the number 1 + rank + 256 * world.  This rank's block lands in its slot, the other ranks' slots are
	// zero-filled (roughly the HBM writes a real gather makes; the fabric's share of the time is what the rehearsal cannot show).
	int gple_debug_solo_allgather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, void* hip_stream)
	{
		hipStream_t stream = static_cast<hipStream_t>(hip_stream);
		const size_t code = reinterpret_cast<size_t>(comm);
		if (datatype != 8 || code < 257) return 4;
		const size_t world = (code - 1) / 256, rank = (code - 1) % 256;
		if (rank >= world) return 4;
		char* dst = static_cast<char*>(recvbuff);
		const size_t blk = sendcount * sizeof(double);
		if (rank > 0 && hipMemsetAsync(dst, 0, rank * blk, stream) != hipSuccess) return 1;
		if (rank + 1 < world && hipMemsetAsync(dst + (rank + 1) * blk, 0, (world - rank - 1) * blk, stream) != hipSuccess) return 1;
		return hipMemcpyAsync(dst + rank * blk, sendbuff, blk, hipMemcpyDeviceToDevice, stream) == hipSuccess ? 0 : 1;
	}
	int gple_set_allgather_function(void* fn)
	{
		allgather_override.store(reinterpret_cast<allgather_fn>(fn));
		return GPLE_OK;
	}
	static int predict_sharded(gple_ctx* ctx, const FitCommon* f, bool is_complex, const double* Xs, size_t M, unsigned flags, int rank, int world, const int* weights,
		void* comm, double* prediction, double* variance, double* cutoff_prediction)
	{
		if (world < 1 || rank < 0 || rank >= world) return GPLE_ERR_BAD_ARG;
		if (world > 1 && !comm) return GPLE_ERR_BAD_ARG;
		Deal deal;
		if (!make_deal(world, weights, deal)) return GPLE_ERR_BAD_ARG;
		if (!f && deal.weight(rank) > 0) return GPLE_ERR_BAD_ARG; // only a rank without a share may come without the fit
		if (M == 0) return GPLE_OK;
		const bool dev = flags & GPLE_IO_DEVICE;
		if (!comm) return predict_common(ctx, f, Xs, M, flags & (GPLE_IO_DEVICE | GPLE_PREDICT_FULL), nullptr, prediction, variance, cutoff_prediction, nullptr);
		const allgather_fn allgather = resolve_allgather();
		if (!allgather)
		{
			std::lock_guard<std::mutex> lk(ctx->mu);
			const char* de = dlerror(); // one call: dlerror() clears the state it returns
			ctx->last_error = std::string("ncclAllGather not found: ") + (de ? de : "librccl is not loadable");
			return GPLE_ERR_COLLECTIVE;
		}
		// this rank's points: its blocks of every cycle (the last block of the grid may be short)
		size_t n_local = 0, per = 0;
		deal_counts(deal, M, rank, n_local, per);
		const size_t ow = is_complex ? 2 : 1, blk = (2 * ow + 1) * per;
		hipStream_t st = ctx->stream;
		// the points go through device buffers whatever the caller's pointers are: the collective runs on device memory
		Scratch local(ctx), gathered(ctx), xs_all(ctx), xs(ctx), om(ctx), ov(ctx), oc(ctx);
		double* xs_shard = nullptr; // this rank's points: pooled (xs) or in the early buffer
		unsigned xs_flag = 0;
		{
			std::lock_guard<std::mutex> lk(ctx->call_mu);
			GPLE_HIP(ctx, hipSetDevice(ctx->device));
			GPLE_HIP(ctx, local.get(blk));
			GPLE_HIP(ctx, gathered.get(blk * world));
			GPLE_HIP(ctx, hipMemsetAsync(local.p, 0, blk * 8, st)); // the padded tail of the ranks with fewer points
			const double* all_dev = Xs;
			if (!dev)
			{
				GPLE_HIP(ctx, xs_all.get(2 * M));
				GPLE_HIP(ctx, copy_in(st, xs_all.p, Xs, 2 * M, false));
				all_dev = xs_all.p;
			}
			if (n_local)
			{
				// Beside the fit (GPLE_PREDICT_OVERLAP=1, device pointers): this rank's points are dealt out on the early stream into the early buffer —
				// the main stream is busy with the fit, and a pooled buffer may still be that fit's workspace
				bool early = false;
				if (dev && f && !f->validated.load() && f->early_rows > 0 && ctx->fit_early_event && predict_overlap_enabled())
				{
					const int Mh = static_cast<int>(round_up(n_local, 128));
					PredictArgs pa{};
					pa.M = static_cast<int>(n_local), pa.m_rows = is_complex ? 2 * Mh : Mh, pa.n_total = f->n_total;
					if (predict_overlap_applicable(ctx, pa, f->early_rows))
					{
						GPLE_HIP(ctx, predict_overlap_prepare(ctx, pa, 2 * n_local, &xs_shard));
						hipLaunchKernelGGL(shard_points_kernel, dim3(static_cast<unsigned>((n_local + 255) / 256)), dim3(256), 0, ctx->early_stream, all_dev, M, rank, deal, n_local,
							xs_shard);
						GPLE_HIP(ctx, hipGetLastError());
						GPLE_HIP(ctx, hipEventRecord(ctx->early_points, ctx->early_stream));
						early = true, xs_flag = PREDICT_XS_EARLY;
					}
				}
				if (!early)
				{
					GPLE_HIP(ctx, xs.get(2 * n_local));
					xs_shard = xs.p;
					hipLaunchKernelGGL(shard_points_kernel, dim3(static_cast<unsigned>((n_local + 255) / 256)), dim3(256), 0, st, all_dev, M, rank, deal, n_local, xs.p);
					GPLE_HIP(ctx, hipGetLastError());
				}
			}
		}
		// A rank whose own predict fails still enters the collective (with whatever its buffer holds: the other ranks get their
		// result, this one reports its error afterwards) -- returning here would leave every other rank waiting in ncclAllGather.
		// Only a failure to allocate the collective's own buffers above returns early; the caller then has to abort the communicator.
		int rc_local = GPLE_OK;
		std::string err_local;
		if (n_local)
		{
			rc_local = predict_common(ctx, f, xs_shard, n_local, GPLE_IO_DEVICE | (flags & GPLE_PREDICT_FULL) | xs_flag, nullptr, local.p, local.p + ow * per,
				local.p + (ow + 1) * per, nullptr);
			if (rc_local != GPLE_OK)
			{
				std::lock_guard<std::mutex> l2(ctx->mu);
				err_local = ctx->last_error;
			}
		}
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		const int rc = allgather(local.p, gathered.p, blk, /* ncclDouble */ 8, comm, st);
		if (rc != 0)
		{
			std::lock_guard<std::mutex> l2(ctx->mu);
			ctx->last_error = "ncclAllGather returned " + std::to_string(rc);
			return GPLE_ERR_COLLECTIVE;
		}
		if (rc_local != GPLE_OK)
		{
			GPLE_HIP(ctx, hipStreamSynchronize(st)); // the scratch buffers go back to the pool when this returns
			std::lock_guard<std::mutex> l2(ctx->mu);
			ctx->last_error = err_local;
			return rc_local;
		}
		double *d_mean = prediction, *d_var = variance, *d_cut = cutoff_prediction;
		if (!dev)
		{
			if (prediction)
			{
				GPLE_HIP(ctx, om.get(ow * M));
				d_mean = om.p;
			}
			if (variance)
			{
				GPLE_HIP(ctx, ov.get(M));
				d_var = ov.p;
			}
			if (cutoff_prediction)
			{
				GPLE_HIP(ctx, oc.get(ow * M));
				d_cut = oc.p;
			}
		}
		hipLaunchKernelGGL(unshard_kernel, dim3(static_cast<unsigned>((M + 255) / 256)), dim3(256), 0, st, gathered.p, per, static_cast<int>(ow), M, deal, d_mean, d_var,
			d_cut);
		GPLE_HIP(ctx, hipGetLastError());
		if (!dev)
		{
			GPLE_HIP(ctx, copy_out(st, prediction, d_mean, ow * M, false));
			GPLE_HIP(ctx, copy_out(st, variance, d_var, M, false));
			GPLE_HIP(ctx, copy_out(st, cutoff_prediction, d_cut, ow * M, false));
			GPLE_HIP(ctx, hipStreamSynchronize(st));
			timer_collect(ctx);
		}
		return GPLE_OK;
	}
	int gple_real_predict_sharded(gple_ctx* ctx, const gple_real_fit* fit, const double* Xs, size_t M, unsigned flags, int rank, int world,
		void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction)
	{
		if (!ctx || !fit || (M && !Xs)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		return predict_sharded(ctx, fit, false, Xs, M, flags, rank, world, nullptr, nccl_comm, prediction, variance, cutoff_prediction);
	}
	int gple_complex_predict_sharded(gple_ctx* ctx, const gple_complex_fit* fit, const double* Xs, size_t M, unsigned flags, int rank, int world,
		void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction)
	{
		if (!ctx || !fit || (M && !Xs)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		return predict_sharded(ctx, fit, true, Xs, M, flags, rank, world, nullptr, nccl_comm, prediction, variance, cutoff_prediction);
	}
	int gple_real_predict_dealt(gple_ctx* ctx, const gple_real_fit* fit, const double* Xs, size_t M, unsigned flags, int rank, int world,
		const int* weights, void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction)
	{
		if (!ctx || (M && !Xs)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		return predict_sharded(ctx, fit, false, Xs, M, flags, rank, world, weights, nccl_comm, prediction, variance, cutoff_prediction);
	}
	int gple_complex_predict_dealt(gple_ctx* ctx, const gple_complex_fit* fit, const double* Xs, size_t M, unsigned flags, int rank, int world,
		const int* weights, void* nccl_comm, double* prediction, double* variance, double* cutoff_prediction)
	{
		if (!ctx || (M && !Xs)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		return predict_sharded(ctx, fit, true, Xs, M, flags, rank, world, weights, nccl_comm, prediction, variance, cutoff_prediction);
	}
	int gple_deal_share(size_t M, int rank, int world, const int* weights, size_t* n_local, size_t* per, size_t* indices)
	{
		Deal deal;
		if (rank < 0 || rank >= world || !make_deal(world, weights, deal)) return GPLE_ERR_BAD_ARG;
		size_t nl = 0, p = 0;
		deal_counts(deal, M, rank, nl, p);
		if (n_local) *n_local = nl;
		if (per) *per = p;
		if (indices)
			for (size_t j = 0; j < nl; ++j) indices[j] = deal.global_block(rank, j / SHARD_BLOCK) * SHARD_BLOCK + j % SHARD_BLOCK;
		return GPLE_OK;
	}

	// ---- batched point-predict (N1): gather -> one predict per element -> scatter ------------------------------------
	int gple_predict_batch(gple_ctx* ctx, const gple_element* elements, size_t n_elements, const double* points, const int* element_of_request,
		size_t n_req, double* out)
	{
		if (!ctx || (n_elements && !elements) || (n_req && (!points || !element_of_request || !out))) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::vector<std::vector<size_t>> by_element(n_elements);
		for (size_t r = 0; r < n_req; ++r)
		{
			const int e = element_of_request[r];
			if (e < 0 || static_cast<size_t>(e) >= n_elements) return GPLE_ERR_BAD_ARG;
			by_element[e].push_back(r);
		}
		std::vector<double> pts, cut;
		for (size_t e = 0; e < n_elements; ++e)
		{
			const std::vector<size_t>& req = by_element[e];
			if (req.empty()) continue;
			const gple_element& el = elements[e];
			if (el.real && el.cplx) return GPLE_ERR_BAD_ARG;
			if (!el.real && !el.cplx) // element without a kernel: 0 (main.cpp:86-88, 97-99)
			{
				for (size_t r : req) out[2 * r] = out[2 * r + 1] = 0.0;
				continue;
			}
			const size_t m = req.size();
			pts.resize(2 * m);
			for (size_t q = 0; q < m; ++q) pts[2 * q] = points[2 * req[q]], pts[2 * q + 1] = points[2 * req[q] + 1];
			if (el.real)
			{
				cut.resize(m);
				GPLE_TRY(predict_common(ctx, el.real, pts.data(), m, GPLE_PREDICT_FULL, nullptr, nullptr, nullptr, cut.data(), nullptr));
				for (size_t q = 0; q < m; ++q) out[2 * req[q]] = cut[q], out[2 * req[q] + 1] = 0.0;
			}
			else
			{
				cut.resize(2 * m);
				GPLE_TRY(predict_common(ctx, el.cplx, pts.data(), m, GPLE_PREDICT_FULL, nullptr, nullptr, nullptr, cut.data(), nullptr));
				for (size_t q = 0; q < m; ++q) out[2 * req[q]] = cut[2 * q], out[2 * req[q] + 1] = cut[2 * q + 1];
			}
		}
		return GPLE_OK;
	}

	// ---- step loop (N3) ----------------------------------------------------------------------------------------------------
	int gple_pes_adiabatic(gple_ctx* ctx, int model, const double* x, size_t M, unsigned flags, double* out)
	{
		if (!ctx || model < 0 || model > 2 || (M && (!x || !out))) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (M == 0) return GPLE_OK;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		if (flags & GPLE_IO_DEVICE)
		{
			GPLE_HIP(ctx, launch_pes(st, x, (int)M, model, out));
			return GPLE_OK;
		}
		Scratch xd(ctx), od(ctx);
		GPLE_HIP(ctx, xd.get(M));
		GPLE_HIP(ctx, od.get(6 * M));
		GPLE_HIP(ctx, copy_in(st, xd.p, x, M, false));
		GPLE_HIP(ctx, launch_pes(st, xd.p, (int)M, model, od.p));
		GPLE_HIP(ctx, copy_out(st, out, od.p, 6 * M, false));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		return GPLE_OK;
	}

	// cut-off prediction of `el` at m device points -> out (m doubles, or m (re,im) pairs); out == nullptr on return means "0"
	static int predict_element_cutoff(gple_ctx* ctx, const gple_element& el, const double* pts_dev, size_t m, double* out_dev)
	{
		if (el.real && el.cplx) return GPLE_ERR_BAD_ARG;
		if (m == 0 || (!el.real && !el.cplx)) return GPLE_OK;
		const FitCommon* f = el.real ? static_cast<const FitCommon*>(el.real) : static_cast<const FitCommon*>(el.cplx);
		// (the points of a tick sit on or next to the sampled density: nothing to prune, and the row statistics would cost a second
		// generation pass)
		return predict_common(ctx, f, pts_dev, m, GPLE_IO_DEVICE | GPLE_PREDICT_FULL, nullptr, nullptr, nullptr, out_dev, nullptr);
	}

	int gple_evolve(gple_ctx* ctx, const gple_element elements[3], int pes_model, double mass, double dt, gple_points density[3], unsigned flags)
	{
		if (!ctx || !elements || !density || pes_model < 0 || pes_model > 2 || !(mass > 0.0)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		int n[3];
		size_t total = 0;
		for (int e = 0; e < 3; ++e)
		{
			if (density[e].n > (1u << 28) || (density[e].n && (!density[e].r || !density[e].rho))) return GPLE_ERR_BAD_ARG;
			n[e] = static_cast<int>(density[e].n);
			total += density[e].n;
		}
		if (total == 0) return GPLE_OK;
		const bool dev = flags & GPLE_IO_DEVICE;
		long qoff[3][3], qlen[3];
		int off[3];
		const int new_points = (flags & GPLE_EVOLVE_NEW_POINTS) ? 1 : 0;
		evolve_layout(n, qoff, qlen, off, new_points);
		hipStream_t st = ctx->stream;
		Scratch r_old(ctx), rho_old(ctx), r_new(ctx), rho_new(ctx), cpl(ctx), q0(ctx), q1(ctx), q2(ctx), p0(ctx), p1(ctx), p2(ctx);
		Scratch* q[3] = {&q0, &q1, &q2};
		Scratch* pr[3] = {&p0, &p1, &p2};
		{
			std::lock_guard<std::mutex> lk(ctx->call_mu);
			GPLE_HIP(ctx, hipSetDevice(ctx->device));
			GPLE_HIP(ctx, r_old.get(2 * total));
			GPLE_HIP(ctx, rho_old.get(2 * total));
			GPLE_HIP(ctx, r_new.get(2 * total));
			GPLE_HIP(ctx, rho_new.get(2 * total));
			GPLE_HIP(ctx, cpl.get(total / 8 + 1));
			for (int e = 0; e < 3; ++e)
			{
				GPLE_HIP(ctx, q[e]->get(2 * static_cast<size_t>(qlen[e]) + 2));
				GPLE_HIP(ctx, pr[e]->get(2 * static_cast<size_t>(qlen[e]) + 2));
				// the points of the three elements back to back
				GPLE_HIP(ctx, copy_in(st, r_old.p + 2 * off[e], density[e].r, 2 * density[e].n, dev));
				GPLE_HIP(ctx, copy_in(st, rho_old.p + 2 * off[e], density[e].rho, 2 * density[e].n, dev));
			}
			double* const qp[3] = {q0.p, q1.p, q2.p};
			GPLE_HIP(ctx, launch_evolve_prepare(st, r_old.p, n, mass, dt, pes_model, r_new.p, reinterpret_cast<unsigned char*>(cpl.p), qp, new_points));
		}
		// one batched predict per density-matrix element over everything that was back-propagated into it
		const double* pred[3] = {nullptr, nullptr, nullptr};
		for (int e = 0; e < 3; ++e)
		{
			if (qlen[e] == 0 || (!elements[e].real && !elements[e].cplx)) continue;
			if ((e == 1) != (elements[e].cplx != nullptr)) return GPLE_ERR_BAD_ARG; // (1,0) is the complex element, the diagonal ones are real
			GPLE_TRY(predict_element_cutoff(ctx, elements[e], q[e]->p, static_cast<size_t>(qlen[e]), pr[e]->p));
			pred[e] = pr[e]->p;
		}
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		GPLE_HIP(ctx, launch_evolve_combine(st, r_old.p, r_new.p, rho_old.p, reinterpret_cast<const unsigned char*>(cpl.p), n, mass, dt, pes_model, pred, rho_new.p, new_points));
		for (int e = 0; e < 3; ++e)
		{
			GPLE_HIP(ctx, copy_out(st, density[e].r, r_new.p + 2 * off[e], 2 * density[e].n, dev));
			GPLE_HIP(ctx, copy_out(st, density[e].rho, rho_new.p + 2 * off[e], 2 * density[e].n, dev));
		}
		if (!dev) GPLE_HIP(ctx, hipStreamSynchronize(st));
		return GPLE_OK;
	}

	/* N-level form of gple_evolve / gple_pes_adiabatic (gple_evolve_n.hip): num_pes = 2 or 3, NE = num_pes (num_pes + 1) / 2 elements in the
	 * packing order (0,0), (1,0), (1,1), (2,0), (2,1), (2,2) */
	int gple_pes_adiabatic_n(gple_ctx* ctx, int num_pes, int model, const double* x, size_t M, unsigned flags, double* out)
	{
		if (!ctx || (num_pes != 2 && num_pes != 3) || model < 0 || model > (num_pes == 3 ? 3 : 2) || (M && (!x || !out)) || M > (1u << 28)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (M == 0) return GPLE_OK;
		const bool dev = flags & GPLE_IO_DEVICE;
		const size_t width = static_cast<size_t>(num_pes) + 2 * static_cast<size_t>(num_pes * (num_pes + 1) / 2);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		Scratch xd(ctx), od(ctx);
		const double* xin = x;
		double* o = out;
		if (!dev)
		{
			GPLE_HIP(ctx, xd.get(M));
			GPLE_HIP(ctx, od.get(width * M));
			GPLE_HIP(ctx, copy_in(st, xd.p, x, M, false));
			xin = xd.p, o = od.p;
		}
		GPLE_HIP(ctx, launch_pes_n(st, num_pes, xin, static_cast<int>(M), model, o));
		if (!dev)
		{
			GPLE_HIP(ctx, copy_out(st, out, od.p, width * M, false));
			GPLE_HIP(ctx, hipStreamSynchronize(st));
		}
		return GPLE_OK;
	}

	int gple_evolve_n(gple_ctx* ctx, int num_pes, const gple_element* elements, int pes_model, double mass, double dt, gple_points* density, unsigned flags)
	{
		if (!ctx || !elements || !density || (num_pes != 2 && num_pes != 3) || pes_model < 0 || pes_model > (num_pes == 3 ? 3 : 2) || !(mass > 0.0))
			return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		const int NE = num_pes * (num_pes + 1) / 2;
		int n[6] = {0, 0, 0, 0, 0, 0}, off[6];
		bool diagonal[6];
		size_t total = 0;
		for (int i = 0, e = 0; i < num_pes; ++i)
			for (int j = 0; j <= i; ++j, ++e) diagonal[e] = i == j;
		for (int e = 0; e < NE; ++e)
		{
			if (density[e].n > (1u << 26) || (density[e].n && (!density[e].r || !density[e].rho))) return GPLE_ERR_BAD_ARG;
			if (elements[e].real && elements[e].cplx) return GPLE_ERR_BAD_ARG;
			if ((elements[e].real && !diagonal[e]) || (elements[e].cplx && diagonal[e])) return GPLE_ERR_BAD_ARG; // real GPs on the diagonal, complex ones off it
			n[e] = static_cast<int>(density[e].n), off[e] = static_cast<int>(total);
			total += density[e].n;
		}
		if (total == 0) return GPLE_OK;
		const bool dev = flags & GPLE_IO_DEVICE;
		const int new_points = (flags & GPLE_EVOLVE_NEW_POINTS) ? 1 : 0;
		long qlen[6];
		evolve_layout_n(num_pes, n, qlen);
		hipStream_t st = ctx->stream;
		Scratch r_old(ctx), rho_old(ctx), r_new(ctx), rho_new(ctx);
		std::vector<std::unique_ptr<Scratch>> q, pr;
		double* qp[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
		{
			std::lock_guard<std::mutex> lk(ctx->call_mu);
			GPLE_HIP(ctx, hipSetDevice(ctx->device));
			GPLE_HIP(ctx, r_old.get(2 * total));
			GPLE_HIP(ctx, rho_old.get(2 * total));
			GPLE_HIP(ctx, r_new.get(2 * total));
			GPLE_HIP(ctx, rho_new.get(2 * total));
			for (int e = 0; e < NE; ++e)
			{
				q.emplace_back(new Scratch(ctx)), pr.emplace_back(new Scratch(ctx));
				GPLE_HIP(ctx, q[e]->get(2 * static_cast<size_t>(qlen[e]) + 2));
				GPLE_HIP(ctx, pr[e]->get(2 * static_cast<size_t>(qlen[e]) + 2));
				qp[e] = q[e]->p;
				GPLE_HIP(ctx, copy_in(st, r_old.p + 2 * off[e], density[e].r, 2 * density[e].n, dev));
				GPLE_HIP(ctx, copy_in(st, rho_old.p + 2 * off[e], density[e].rho, 2 * density[e].n, dev));
			}
			GPLE_HIP(ctx, launch_evolve_prepare_n(st, num_pes, r_old.p, n, mass, dt, pes_model, r_new.p, qp, new_points));
		}
		// one batched predict per density-matrix element over everything that was back-propagated into it
		const double* pred[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
		for (int e = 0; e < NE; ++e)
		{
			if (qlen[e] == 0 || (!elements[e].real && !elements[e].cplx)) continue;
			GPLE_TRY(predict_element_cutoff(ctx, elements[e], q[e]->p, static_cast<size_t>(qlen[e]), pr[e]->p));
			pred[e] = pr[e]->p;
		}
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		GPLE_HIP(ctx, launch_evolve_combine_n(st, num_pes, r_new.p, rho_old.p, n, mass, dt, pes_model, pred, rho_new.p, new_points));
		for (int e = 0; e < NE; ++e)
		{
			GPLE_HIP(ctx, copy_out(st, density[e].r, r_new.p + 2 * off[e], 2 * density[e].n, dev));
			GPLE_HIP(ctx, copy_out(st, density[e].rho, rho_new.p + 2 * off[e], 2 * density[e].n, dev));
		}
		GPLE_HIP(ctx, hipStreamSynchronize(st)); // the scratch lists go back to the pool when this returns
		return GPLE_OK;
	}

	static int markov_chain_impl(gple_ctx* ctx, const gple_element* element, size_t num_steps, double max_displacement, unsigned long long seed, double* r,
		size_t n, double* accept_ratio, double* chain)
	{
		if (!ctx || !element || (n && !r) || n > (1u << 28) || (element->real && element->cplx)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (n == 0) return GPLE_OK;
		hipStream_t st = ctx->stream;
		const int ni = static_cast<int>(n), cplx = element->cplx ? 1 : 0;
		const bool has_fit = element->real || element->cplx;
		Scratch rd(ctx), rp(ctx), pred(ctx), weight(ctx), acc(ctx), trace(ctx);
		{
			std::lock_guard<std::mutex> lk(ctx->call_mu);
			GPLE_HIP(ctx, hipSetDevice(ctx->device));
			if (chain) GPLE_HIP(ctx, trace.get((num_steps + 1) * 2 * n)); // the whole chains, [step][walker][2]
			GPLE_HIP(ctx, rd.get(2 * n));
			GPLE_HIP(ctx, rp.get(2 * n));
			GPLE_HIP(ctx, pred.get(2 * n));
			GPLE_HIP(ctx, weight.get(n));
			GPLE_HIP(ctx, acc.get(n / 2 + 1));
			GPLE_HIP(ctx, copy_in(st, rd.p, r, 2 * n, false));
			GPLE_HIP(ctx, hipMemsetAsync(acc.p, 0, (n / 2 + 1) * 8, st));
			if (chain) GPLE_HIP(ctx, hipMemcpyAsync(trace.p, rd.p, 2 * n * 8, hipMemcpyDeviceToDevice, st));
		}
		GPLE_TRY(predict_element_cutoff(ctx, *element, rd.p, n, pred.p));
		{
			std::lock_guard<std::mutex> lk(ctx->call_mu);
			GPLE_HIP(ctx, launch_mc_weight(st, has_fit ? pred.p : nullptr, cplx, ni, weight.p)); // mc.cpp:131
		}
		for (size_t step = 0; step < num_steps; ++step)
		{
			{
				std::lock_guard<std::mutex> lk(ctx->call_mu);
				GPLE_HIP(ctx, launch_mc_propose(st, rd.p, ni, static_cast<unsigned>(step), seed, max_displacement, rp.p));
			}
			GPLE_TRY(predict_element_cutoff(ctx, *element, rp.p, n, pred.p));
			std::lock_guard<std::mutex> lk(ctx->call_mu);
			GPLE_HIP(ctx, launch_mc_accept(st, rd.p, rp.p, has_fit ? pred.p : nullptr, cplx, ni, static_cast<unsigned>(step), seed, weight.p,
							  reinterpret_cast<unsigned*>(acc.p)));
			if (chain) GPLE_HIP(ctx, hipMemcpyAsync(trace.p + (step + 1) * 2 * n, rd.p, 2 * n * 8, hipMemcpyDeviceToDevice, st));
		}
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, copy_out(st, r, rd.p, 2 * n, false));
		if (chain) GPLE_HIP(ctx, copy_out(st, chain, trace.p, (num_steps + 1) * 2 * n, false));
		std::vector<unsigned> counts(accept_ratio ? n : 0);
		if (accept_ratio) GPLE_HIP(ctx, hipMemcpyAsync(counts.data(), acc.p, n * sizeof(unsigned), hipMemcpyDeviceToHost, st));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		if (accept_ratio)
			for (size_t i = 0; i < n; ++i) accept_ratio[i] = num_steps ? static_cast<double>(counts[i]) / static_cast<double>(num_steps) : 0.0;
		return GPLE_OK;
	}

	int gple_markov_chain(gple_ctx* ctx, const gple_element* element, size_t num_steps, double max_displacement, unsigned long long seed, double* r,
		size_t n, double* accept_ratio)
	{
		return markov_chain_impl(ctx, element, num_steps, max_displacement, seed, r, n, accept_ratio, nullptr);
	}
	int gple_markov_chain_trace(gple_ctx* ctx, const gple_element* element, size_t num_steps, double max_displacement, unsigned long long seed,
		double* r, size_t n, double* accept_ratio, double* chain)
	{
		if (!chain) return GPLE_ERR_BAD_ARG;
		return markov_chain_impl(ctx, element, num_steps, max_displacement, seed, r, n, accept_ratio, chain);
	}

	// loose_function (opt.cpp:441-482).  io = 0: host pointers; io = GPLE_IO_DEVICE: everything but x / value / grad is resident
	// (lab = real parts of y_extra, the label vector of the real kernel's PredictiveKernel, opt.cpp:451)
	static int loose_eval(gple_ctx* ctx, const double* x, size_t n, const double* X, const double* y, size_t N, const double* X_extra,
		const double* y_extra, const double* lab, size_t M_extra, unsigned io, double* value, double* grad, int part = 0, int nparts = 1)
	{
		// part / nparts > 1 (gple_objective_eval_part): this call forms the N^3 products of the parameters ip with ip % nparts == part (the cheap
		// first and last parameters belong to part 0) and predicts the rows [lo, hi) of the extra set; the LOOCV error counts on part 0; the
		// sum over the parts is the whole objective and gradient, make_normal is the caller's after that sum
		unsigned owned = 0xFFu; // travels with the fit (FitCommon::deriv_mask), not through the context: other threads' fits on this context are not touched
		if (nparts > 1)
		{
			const size_t per = (M_extra + nparts - 1) / nparts, lo = std::min(M_extra, per * part), hi = std::min(M_extra, lo + per);
			X_extra += 2 * lo, y_extra += 2 * lo, lab += lo, M_extra = hi - lo;
			unsigned mask = 0;
			for (size_t ip = 0; ip < n; ++ip)
				if ((ip == 0 || ip == n - 1) ? part == 0 : static_cast<int>(ip % nparts) == part) mask |= 1u << ip;
			owned = mask;
		}
		const unsigned flags = GPLE_CALC_ERROR | (grad ? GPLE_CALC_DERIVATIVE : 0u);
		gple_predict_scalars ps;
		double result = 0.0;
		// The fit's scalars are deferred and the predict leaves its synchronisation to us: the whole evaluation is enqueued in one
		// go and the stream is drained once (a synchronisation between fit and predict would idle the GPU for the host's turn)
		const bool want_deriv = (flags & GPLE_CALC_DERIVATIVE) && M_extra;
		if (n == 4)
		{
			gple_real_fit_scalars sc;
			gple_real_fit* fit = nullptr;
			GPLE_TRY(real_fit_create_masked(ctx, x, X, y, 1, N, flags | io, owned, nullptr, &fit));
			const unsigned pflags = (flags & GPLE_CALC_DERIVATIVE) | io | PREDICT_NO_SYNC | GPLE_PREDICT_FULL;
			int st = gple_real_predict(ctx, fit, X_extra, M_extra, pflags, lab, nullptr, nullptr, nullptr, &ps);
			if (st == GPLE_OK) st = gple_real_fit_get_scalars(fit, &sc); // drains the stream
			if (st == GPLE_ERR_TIMEOUT && fit->validated.load()) // the factorisation had given up and was repeated: the predict above saw NaN — once more, on the good fit
			{
				st = gple_real_predict(ctx, fit, X_extra, M_extra, pflags, lab, nullptr, nullptr, nullptr, &ps);
				if (st == GPLE_OK) st = gple_ctx_synchronize(ctx); // (the scalars are cached by now: the getter would not drain the stream)
				if (st == GPLE_OK) st = gple_real_fit_get_scalars(fit, &sc);
			}
			gple_real_fit_release(fit);
			GPLE_TRY(st);
			if (M_extra) predict_scalars_from_host(ctx, true, want_deriv, false, &ps);
			result = (part == 0 ? sc.error : 0.0) + (M_extra ? ps.error : 0.0);
			if (grad)
				for (int i = 0; i < 4; ++i) grad[i] = ((owned >> i & 1u) ? sc.error_derivative[i] : 0.0) + (M_extra ? ps.error_derivative[i] : 0.0);
		}
		else
		{
			gple_complex_fit_scalars sc;
			gple_complex_fit* fit = nullptr;
			GPLE_TRY(complex_fit_create_masked(ctx, x, X, y, N, flags | io, owned, nullptr, &fit));
			const unsigned pflags = (flags & GPLE_CALC_DERIVATIVE) | io | PREDICT_NO_SYNC | GPLE_PREDICT_FULL;
			int st = gple_complex_predict(ctx, fit, X_extra, M_extra, pflags, y_extra, nullptr, nullptr, nullptr, &ps);
			if (st == GPLE_OK) st = gple_complex_fit_get_scalars(fit, &sc);
			if (st == GPLE_ERR_TIMEOUT && fit->validated.load())
			{
				st = gple_complex_predict(ctx, fit, X_extra, M_extra, pflags, y_extra, nullptr, nullptr, nullptr, &ps);
				if (st == GPLE_OK) st = gple_ctx_synchronize(ctx);
				if (st == GPLE_OK) st = gple_complex_fit_get_scalars(fit, &sc);
			}
			gple_complex_fit_release(fit);
			GPLE_TRY(st);
			if (M_extra) predict_scalars_from_host(ctx, true, want_deriv, true, &ps);
			result = (part == 0 ? sc.error : 0.0) + (M_extra ? ps.error : 0.0);
			if (grad)
				for (int i = 0; i < 8; ++i) grad[i] = ((owned >> i & 1u) ? sc.error_derivative[i] : 0.0) + (M_extra ? ps.error_derivative[i] : 0.0);
		}
		// make_normal, opt.cpp:420-431
		auto make_normal = [](double& d) {
			if (std::isnan(d) || std::isinf(d)) d = std::numeric_limits<double>::max();
		};
		if (nparts == 1)
		{
			make_normal(result);
			if (grad)
				for (size_t i = 0; i < n; ++i) make_normal(grad[i]);
		}
		*value = result;
		return GPLE_OK;
	}
	int gple_loose_function(gple_ctx* ctx, const double* x, size_t n, const double* X, const double* y, size_t N, const double* X_extra,
		const double* y_extra, size_t M_extra, double* value, double* grad)
	{
		if (!ctx || !x || !X || !y || !value || (n != 4 && n != 8) || (M_extra && (!X_extra || !y_extra))) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::vector<double> lab(M_extra);
		for (size_t i = 0; i < M_extra; ++i) lab[i] = y_extra[2 * i];
		return loose_eval(ctx, x, n, X, y, N, X_extra, y_extra, lab.data(), M_extra, 0u, value, grad);
	}

	// ---- the objective with its data resident (ElementTrainingParameters of opt.cpp:16) ------------------------------------
	struct gple_objective
	{
		gple_ctx* ctx = nullptr;
		size_t N = 0, M = 0;
		double *X = nullptr, *y = nullptr, *Xe = nullptr, *ye = nullptr, *lab = nullptr;
	};
	int gple_objective_create(gple_ctx* ctx, const double* X, const double* y, size_t N, const double* X_extra, const double* y_extra,
		size_t M_extra, gple_objective** out)
	{
		if (!ctx || !X || !y || !out || N == 0 || (M_extra && (!X_extra || !y_extra))) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		*out = nullptr;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		gple_objective* o = new (std::nothrow) gple_objective;
		if (!o) return GPLE_ERR_ALLOC;
		o->ctx = ctx, o->N = N, o->M = M_extra;
		ctx_retain(ctx); // dropped in gple_objective_release
		hipStream_t st = ctx->stream;
		hipError_t e = hipSuccess;
		auto up = [&](double*& dst, const double* src, size_t n) {
			if (e != hipSuccess || n == 0) return;
			dst = ctx->acquire(n * 8, &e);
			if (e == hipSuccess) e = hipMemcpyAsync(dst, src, n * 8, hipMemcpyHostToDevice, st);
		};
		std::vector<double> lab(M_extra);
		for (size_t i = 0; i < M_extra; ++i) lab[i] = y_extra[2 * i];
		up(o->X, X, 2 * N), up(o->y, y, 2 * N), up(o->Xe, X_extra, 2 * M_extra), up(o->ye, y_extra, 2 * M_extra), up(o->lab, lab.data(), M_extra);
		if (e == hipSuccess) e = hipStreamSynchronize(st); // the host arrays may go away once this returns
		if (e != hipSuccess)
		{
			for (double* p : {o->X, o->y, o->Xe, o->ye, o->lab}) ctx->give_back(p);
			delete o;
			const int status = record_hip_error(ctx, e, "objective upload", __LINE__);
			ctx_drop(ctx);
			return status;
		}
		*out = o;
		return GPLE_OK;
	}
	int gple_objective_eval(gple_objective* o, const double* x, size_t n, double* value, double* grad)
	{
		if (!o || !x || !value || (n != 4 && n != 8)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(o->ctx);
		return loose_eval(o->ctx, x, n, o->X, o->y, o->N, o->Xe, o->ye, o->lab, o->M, GPLE_IO_DEVICE, value, grad);
	}
	int gple_objective_eval_part(gple_objective* o, const double* x, size_t n, int part, int nparts, double* value, double* grad)
	{
		if (!o || !x || !value || (n != 4 && n != 8) || nparts < 1 || part < 0 || part >= nparts) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(o->ctx);
		return loose_eval(o->ctx, x, n, o->X, o->y, o->N, o->Xe, o->ye, o->lab, o->M, GPLE_IO_DEVICE, value, grad, part, nparts);
	}
	int gple_objective_release(gple_objective* o)
	{
		if (!o) return GPLE_OK;
		gple_ctx* ctx = o->ctx;
		(void)hipSetDevice(ctx->device);
		(void)hipStreamSynchronize(ctx->stream);
		for (double* p : {o->X, o->y, o->Xe, o->ye, o->lab}) ctx->give_back(p);
		delete o;
		ctx_drop(ctx);
		return GPLE_OK;
	}

	// ---- negative_log_marginal_likelihood / predict_phase (test/gpr.cpp:499-532, 654-706) -------------------------------
	// shared: Gram, Cholesky, inverse factor, b = K^-1 y (labels are NOT rescaled on this path).  Enqueue only; `info` (device, one double's
	// slot) receives the factorisation's info word and a negative one turns b into NaN (colpass_kernel), so that nothing derived from an
	// unfinished factor looks like a number; the callers read the word back with their results and repeat the call with one launch per panel.
	static int nlml_solve(gple_ctx* ctx, const double x[5], const double* X, const double* y, size_t N, Scratch& Xt, Scratch& yd, Scratch& T,
		Scratch& bvec, Scratch& info, int* n_out)
	{
		hipStream_t st = ctx->stream;
		const int n = static_cast<int>(round_up(N, NPAD));
		*n_out = n;
		Scratch L(ctx), work(ctx), part(ctx), u(ctx), w(ctx);
		GPLE_HIP(ctx, Xt.get(2 * static_cast<size_t>(n)));
		GPLE_HIP(ctx, yd.get(n));
		GPLE_HIP(ctx, T.get(static_cast<size_t>(n) * n));
		GPLE_HIP(ctx, bvec.get(n));
		GPLE_HIP(ctx, L.get(static_cast<size_t>(n) * n));
		GPLE_HIP(ctx, work.get(chol_inverse_work_doubles(n)));
		GPLE_HIP(ctx, part.get(static_cast<size_t>(n / 256) * n));
		GPLE_HIP(ctx, u.get(n));
		GPLE_HIP(ctx, w.get(n));
		GPLE_HIP(ctx, info.get(1));
		GPLE_HIP(ctx, hipMemsetAsync(Xt.p, 0, 2 * static_cast<size_t>(n) * 8, st));
		GPLE_HIP(ctx, hipMemsetAsync(yd.p, 0, static_cast<size_t>(n) * 8, st));
		GPLE_HIP(ctx, hipMemsetAsync(info.p, 0, 8, st));
		// (unlike a fit's, this T is cleared: trmv_lower below walks whole 256-column chunks of a row, the blocks above the diagonal 64-blocks
		// included — a fit gets u = L^-1 y from the label row of its factorisation instead; 30 us at n = 4096)
		GPLE_HIP(ctx, hipMemsetAsync(T.p, 0, static_cast<size_t>(n) * n * 8, st));
		GPLE_HIP(ctx, copy_in(st, Xt.p, X, 2 * N, false));
		GPLE_HIP(ctx, copy_in(st, yd.p, y, N, false));
		GPLE_HIP(ctx, launch_nlml_gram(st, Xt.p, static_cast<int>(N), n, x, L.p));
		timer_start(ctx, GPLE_TIMER_FIT); // the factorisation + inverse factor: what the NLML workloads of bench.py price against the fp64 MFMA peak
		GPLE_HIP(ctx, chol_inverse_factor(ctx, st, L.p, n, n, T.p, n, reinterpret_cast<int*>(info.p), work.p));
		timer_stop(ctx, GPLE_TIMER_FIT);
		GPLE_HIP(ctx, launch_trmv_lower(st, T.p, n, n, yd.p, part.p, u.p));
		GPLE_HIP(ctx, launch_colpass(st, T.p, n, n, u.p, bvec.p, w.p, 0, nullptr, reinterpret_cast<int*>(info.p)));
		return GPLE_OK;
	}
	// after the caller's synchronisation: did the factorisation of this attempt give up?  (the word was copied to host_scalars[HS_NLML + 8])
	static int nlml_gave_up(gple_ctx* ctx, int attempt, bool* again)
	{
		int info_i;
		std::memcpy(&info_i, ctx->host_scalars + HS_NLML + 8, sizeof(int));
		*again = false;
		if (info_i >= 0) return GPLE_OK;
		ctx->dag_giveups += 1;
		if (attempt == 1)
		{
			std::lock_guard<std::mutex> lk(ctx->mu);
			ctx->last_error = "the factorisation gave up waiting (info = -1) and so did its repetition with one launch per panel";
			return GPLE_ERR_TIMEOUT;
		}
		ctx->dag_recoveries += 1;
		*again = true;
		return GPLE_OK;
	}

	// n = 4: (w_d, w_g, a_x, a_p), the NOCROSS build; n = 5: (w_d, w_g, a, c, b), the default build's lower-triangular weight matrix
	static void nlml_params(const double* x, size_t n, double x5[5])
	{
		x5[0] = x[0], x5[1] = x[1], x5[2] = x[2];
		x5[3] = n == 5 ? x[3] : 0.0;
		x5[4] = n == 5 ? x[4] : x[3];
	}
	static int nlml_impl(gple_ctx* ctx, const double* x, size_t n, const double* X, const double* y, size_t N, double* value, double* grad)
	{
		if (!ctx || !x || !X || !y || !value || N == 0) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		double x5[5];
		nlml_params(x, n, x5);
		for (int attempt = 0;; ++attempt)
		{
			CholSchemeScope scheme(attempt == 0 ? ctx->chol_scheme : 0); // second attempt: one launch per panel (a give-up of the one-launch scheme)
			Scratch Xt(ctx), yd(ctx), T(ctx), b(ctx), out(ctx), W(ctx), part(ctx), info(ctx);
			int np = 0;
			GPLE_TRY(nlml_solve(ctx, x5, X, y, N, Xt, yd, T, b, info, &np));
			GPLE_HIP(ctx, out.get(8));
			GPLE_HIP(ctx, launch_nlml_value(st, T.p, np, yd.p, b.p, static_cast<int>(N), out.p));
			if (grad)
			{
				const size_t g = (N + 63) / 64;
				GPLE_HIP(ctx, W.get(static_cast<size_t>(np) * np));
				GPLE_HIP(ctx, part.get(5 * g * g));
				GPLE_HIP(ctx, lauum_full(st, T.p, np, W.p, np, np));
				GPLE_HIP(ctx, launch_nlml_grad(st, Xt.p, static_cast<int>(N), W.p, np, b.p, x5, part.p, out.p + 1));
			}
			GPLE_HIP(ctx, hipMemcpyAsync(ctx->host_scalars + HS_NLML, out.p, 6 * 8, hipMemcpyDeviceToHost, st));
			GPLE_HIP(ctx, hipMemcpyAsync(ctx->host_scalars + HS_NLML + 8, info.p, 8, hipMemcpyDeviceToHost, st));
			GPLE_HIP(ctx, hipStreamSynchronize(st));
			timer_collect(ctx);
			bool again;
			GPLE_TRY(nlml_gave_up(ctx, attempt, &again));
			if (!again) break;
		}
		*value = ctx->host_scalars[HS_NLML];
		if (grad)
		{
			const double* g5 = ctx->host_scalars + HS_NLML + 1; // (w_d, w_g, a, c, b)
			if (n == 5)
				for (int i = 0; i < 5; ++i) grad[i] = g5[i];
			else
				grad[0] = g5[0], grad[1] = g5[1], grad[2] = g5[2], grad[3] = g5[4];
		}
		return GPLE_OK;
	}
	static int nlml_predict_impl(gple_ctx* ctx, const double* x, size_t n, const double* X, const double* y, size_t N, const double* Xs, size_t M,
		unsigned flags, double* mean)
	{
		if (!ctx || !x || !X || !y || N == 0 || (M && (!Xs || !mean))) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		if (M == 0) return GPLE_OK;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const bool dev = flags & GPLE_IO_DEVICE; // applies to Xs / mean only; the training set is small and host-side
		double x5[5];
		nlml_params(x, n, x5);
		for (int attempt = 0;; ++attempt)
		{
			CholSchemeScope scheme(attempt == 0 ? ctx->chol_scheme : 0);
			Scratch Xt(ctx), yd(ctx), T(ctx), b(ctx), xs(ctx), o(ctx), part(ctx), info(ctx);
			int np = 0;
			GPLE_TRY(nlml_solve(ctx, x5, X, y, N, Xt, yd, T, b, info, &np));
			const double* xs_dev = Xs;
			double* o_dev = mean;
			if (!dev)
			{
				GPLE_HIP(ctx, xs.get(2 * M));
				GPLE_HIP(ctx, o.get(M));
				GPLE_HIP(ctx, copy_in(st, xs.p, Xs, 2 * M, false));
				xs_dev = xs.p, o_dev = o.p;
			}
			GPLE_HIP(ctx, part.get(static_cast<size_t>(nlml_predict_ksplit(static_cast<int>(M), static_cast<int>(N))) * M));
			timer_start(ctx, GPLE_TIMER_PREDICT);
			GPLE_HIP(ctx, launch_nlml_predict(st, xs_dev, static_cast<int>(M), Xt.p, static_cast<int>(N), b.p, x5, part.p, o_dev));
			timer_stop(ctx, GPLE_TIMER_PREDICT);
			if (!dev) GPLE_HIP(ctx, copy_out(st, mean, o.p, M, false));
			GPLE_HIP(ctx, hipMemcpyAsync(ctx->host_scalars + HS_NLML + 8, info.p, 8, hipMemcpyDeviceToHost, st));
			GPLE_HIP(ctx, hipStreamSynchronize(st));
			timer_collect(ctx);
			bool again;
			GPLE_TRY(nlml_gave_up(ctx, attempt, &again));
			if (!again) break;
		}
		return GPLE_OK;
	}

	int gple_nlml(gple_ctx* ctx, const double x[4], const double* X, const double* y, size_t N, double* value, double* grad)
	{
		return nlml_impl(ctx, x, 4, X, y, N, value, grad);
	}
	int gple_nlml_predict(gple_ctx* ctx, const double x[4], const double* X, const double* y, size_t N, const double* Xs, size_t M, unsigned flags,
		double* mean)
	{
		return nlml_predict_impl(ctx, x, 4, X, y, N, Xs, M, flags, mean);
	}
	int gple_nlml_cross(gple_ctx* ctx, const double x[5], const double* X, const double* y, size_t N, double* value, double* grad)
	{
		return nlml_impl(ctx, x, 5, X, y, N, value, grad);
	}
	int gple_nlml_cross_predict(gple_ctx* ctx, const double x[5], const double* X, const double* y, size_t N, const double* Xs, size_t M,
		unsigned flags, double* mean)
	{
		return nlml_predict_impl(ctx, x, 5, X, y, N, Xs, M, flags, mean);
	}

	/* gple_debug.h (not part of include/gple.h): instrumented launch of the diagonal-block kernel for probes/diag_probe.py.
	 * A: 64 x 64 column-major SPD block (host); T out: inv(chol(A)) (host); stamps: 16 cycle-counter values (host). */
	int gple_debug_potrf_diag(gple_ctx* ctx, const double* A, double* T, long long* stamps, int reps, float* ms_per_launch)
	{
		if (!ctx || !A || !T || !stamps) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		Scratch a(ctx), t(ctx), aux(ctx);
		GPLE_HIP(ctx, a.get(64 * 64));
		GPLE_HIP(ctx, t.get(64 * 64));
		GPLE_HIP(ctx, aux.get(32));
		GPLE_HIP(ctx, hipMemsetAsync(aux.p, 0, 32 * 8, st));
		GPLE_HIP(ctx, copy_in(st, a.p, A, 64 * 64, false));
		hipEvent_t e0, e1;
		GPLE_HIP(ctx, hipEventCreate(&e0));
		GPLE_HIP(ctx, hipEventCreate(&e1));
		GPLE_HIP(ctx, debug_potrf_diag(st, a.p, t.p, reinterpret_cast<int*>(aux.p), reinterpret_cast<long long*>(aux.p + 8)));
		GPLE_HIP(ctx, hipEventRecord(e0, st));
		for (int i = 0; i < reps; ++i) GPLE_HIP(ctx, debug_potrf_diag(st, a.p, t.p, reinterpret_cast<int*>(aux.p), reinterpret_cast<long long*>(aux.p + 8)));
		GPLE_HIP(ctx, hipEventRecord(e1, st));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		float ms = 0.f;
		GPLE_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
		if (ms_per_launch) *ms_per_launch = reps > 0 ? ms / reps : 0.f;
		(void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
		GPLE_HIP(ctx, hipMemcpy(T, t.p, 64 * 64 * 8, hipMemcpyDeviceToHost));
		GPLE_HIP(ctx, hipMemcpy(stamps, aux.p + 8, 16 * 8, hipMemcpyDeviceToHost));
		return GPLE_OK;
	}

	/* gple_debug.h: the layout the factorisation of an n-column matrix will use (host logic only, no device call): outer block bounds
	 * (0 … n), fork points of the block-row inverse, workspace doubles.  Arrays of `cap` ints; counts come back in nb / nf. */
	// test knobs of the factorisation on this context; a negative argument leaves that knob as it is.  scheme: 0 = a launch per panel, 1 = one
	// launch per outer block, 2 = back to GPLE_CHOL_SCHEME; poll_limit: polls before a waiting wave of the one-launch scheme gives up (0: default);
	// dag_blocks: workgroups of its launches (0: one per CU).  giveups / recoveries (nullable): the context's counters so far.
	int gple_debug_chol_knobs(gple_ctx* ctx, int scheme, int poll_limit, int dag_blocks, long* giveups, long* recoveries)
	{
		if (!ctx) return GPLE_ERR_BAD_ARG;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		if (scheme >= 0) ctx->chol_scheme = scheme >= 2 ? -1 : scheme;
		if (poll_limit >= 0) ctx->dag_poll_limit = poll_limit;
		if (dag_blocks >= 0) ctx->dag_blocks = dag_blocks;
		if (giveups) *giveups = ctx->dag_giveups;
		if (recoveries) *recoveries = ctx->dag_recoveries;
		return GPLE_OK;
	}
	const char* gple_debug_last_contraction_kernel(gple_ctx* ctx) { return ctx ? ctx->last_contraction : ""; }
	long gple_debug_overlapped_predicts(gple_ctx* ctx) { return ctx ? ctx->overlapped_predicts : 0; }
	int gple_debug_predict_knobs(gple_ctx* ctx, int rownorm_pipe, int fused_small)
	{
		if (!ctx) return GPLE_ERR_BAD_ARG;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		if (rownorm_pipe >= 0) ctx->rownorm_pipe = rownorm_pipe >= 2 ? -1 : rownorm_pipe;
		if (fused_small >= 0) ctx->fused_small = fused_small >= 2 ? -1 : fused_small;
		return GPLE_OK;
	}
	int gple_debug_chol_layout(int n, int cap, int* bounds, int* nb, int* forks, int* nf, unsigned long long* work_doubles)
	{
		if (n <= 0 || n % 64 || !bounds || !nb || !forks || !nf || !work_doubles) return GPLE_ERR_BAD_ARG;
		std::vector<int> b, f;
		size_t w = 0;
		chol_layout(n, b, f, w);
		if (static_cast<int>(b.size()) > cap || static_cast<int>(f.size()) > cap) return GPLE_ERR_BAD_ARG;
		std::copy(b.begin(), b.end(), bounds), std::copy(f.begin(), f.end(), forks);
		*nb = static_cast<int>(b.size()), *nf = static_cast<int>(f.size()), *work_doubles = w;
		return GPLE_OK;
	}

	/* gple_debug.h: the outcome of the side stream's queue probing (gple_chol.hip, pick_side_stream) */
	int gple_debug_side_stream(gple_ctx* ctx, int* attempts, int* overlaps)
	{
		if (!ctx || !attempts || !overlaps) return GPLE_ERR_BAD_ARG;
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		*attempts = ctx->side_attempts, *overlaps = ctx->side_overlaps ? 1 : 0;
		return GPLE_OK;
	}

	/* gple_debug.h: instrumented launches of the one-launch panel step (potrf_step_kernel) at block column 1 of an n x n matrix,
	 * n = 128 + below (below = 0 | 64), column-major on the host, overwritten with what the first launch leaves; T (n x n) likewise. */
	int gple_debug_potrf_step(gple_ctx* ctx, double* A, double* T, int pend, int below, long long* stamps, int reps, float* ms_per_launch)
	{
		if (!ctx || !A || !T || !stamps || (below != 0 && below != 64)) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const int n = 128 + below;
		Scratch a(ctx), t(ctx), aux(ctx);
		GPLE_HIP(ctx, a.get(n * n));
		GPLE_HIP(ctx, t.get(n * n));
		GPLE_HIP(ctx, aux.get(40));
		GPLE_HIP(ctx, hipMemsetAsync(aux.p, 0, 40 * 8, st));
		GPLE_HIP(ctx, hipMemsetAsync(t.p, 0, sizeof(double) * n * n, st));
		GPLE_HIP(ctx, copy_in(st, a.p, A, n * n, false));
		hipEvent_t e0, e1;
		GPLE_HIP(ctx, hipEventCreate(&e0));
		GPLE_HIP(ctx, hipEventCreate(&e1));
		GPLE_HIP(ctx, debug_potrf_step(st, a.p, n, t.p, n, reinterpret_cast<int*>(aux.p), reinterpret_cast<long long*>(aux.p + 8), pend, below));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		GPLE_HIP(ctx, hipMemcpy(A, a.p, sizeof(double) * n * n, hipMemcpyDeviceToHost));
		GPLE_HIP(ctx, hipMemcpy(T, t.p, sizeof(double) * n * n, hipMemcpyDeviceToHost));
		GPLE_HIP(ctx, hipEventRecord(e0, st));
		for (int i = 0; i < reps; ++i)
			GPLE_HIP(ctx, debug_potrf_step(st, a.p, n, t.p, n, reinterpret_cast<int*>(aux.p), reinterpret_cast<long long*>(aux.p + 8), pend, below));
		GPLE_HIP(ctx, hipEventRecord(e1, st));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		float ms = 0.f;
		GPLE_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
		if (ms_per_launch) *ms_per_launch = reps > 0 ? ms / reps : 0.f;
		(void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
		GPLE_HIP(ctx, hipMemcpy(stamps, aux.p + 8, 24 * 8, hipMemcpyDeviceToHost)); // of the last (warm) launch
		{
			int info = 0; // as the first launch left it (atomicCAS from 0): reported in the last stamp slot
			GPLE_HIP(ctx, hipMemcpy(&info, aux.p, sizeof(int), hipMemcpyDeviceToHost));
			stamps[23] = info;
		}
		return GPLE_OK;
	}

	/* gple_debug.h (not part of include/gple.h): one product of the fp64 MFMA GEMM family on host operands, for tests/test_gpu_gemm.py.
	 * C(m,n) = alpha sum_k A(m,k) B(n,k) + beta C(m,n); layouts and k-ranges as GemmDesc (csrc/gple_internal.h);
	 * tile = 32 | 64 | 128 | 0 (the library's own choice for this shape). */
	int gple_debug_gemm(gple_ctx* ctx, const double* A, long lda, int a_kmajor, const double* B, long ldb, int b_kmajor, double* C, long ldc,
		int c_trans, int M, int N, int K, double alpha, double beta, int krange, int lower_only, int tile)
	{
		if (!ctx || !A || !B || !C || M <= 0 || N <= 0 || K <= 0) return GPLE_ERR_BAD_ARG;
		GPLE_OPEN(ctx);
		std::lock_guard<std::mutex> lk(ctx->call_mu);
		GPLE_HIP(ctx, hipSetDevice(ctx->device));
		hipStream_t st = ctx->stream;
		const size_t na = static_cast<size_t>(lda) * (a_kmajor ? M : K), nb = static_cast<size_t>(ldb) * (b_kmajor ? N : K),
					 nc = static_cast<size_t>(ldc) * (c_trans ? M : N);
		Scratch a(ctx), b(ctx), c(ctx);
		GPLE_HIP(ctx, a.get(na));
		GPLE_HIP(ctx, b.get(nb));
		GPLE_HIP(ctx, c.get(nc));
		GPLE_HIP(ctx, copy_in(st, a.p, A, na, false));
		GPLE_HIP(ctx, copy_in(st, b.p, B, nb, false));
		GPLE_HIP(ctx, copy_in(st, c.p, C, nc, false));
		GemmDesc g{};
		g.A = a.p, g.lda = lda, g.B = b.p, g.ldb = ldb, g.C = c.p, g.ldc = ldc;
		g.M = M, g.N = N, g.K = K, g.batch = 1, g.alpha = alpha, g.beta = beta, g.krange = krange, g.lower_only = lower_only;
		g.a_kmajor = a_kmajor != 0, g.b_kmajor = b_kmajor != 0, g.c_trans = c_trans != 0;
		GPLE_HIP(ctx, launch_gemm(st, g, tile ? tile : gemm_pick_tile(M, N, 1, krange != K_FULL || lower_only)));
		GPLE_HIP(ctx, hipStreamSynchronize(st));
		GPLE_HIP(ctx, hipMemcpy(C, c.p, nc * 8, hipMemcpyDeviceToHost));
		return GPLE_OK;
	}
}
