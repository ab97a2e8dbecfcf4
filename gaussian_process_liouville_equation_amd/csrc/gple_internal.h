// gple_internal.h — shared declarations of the HIP implementation behind include/gple.h (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gple.h"

namespace gple
{
	// ---- geometry ------------------------------------------------------------------------------------------
	// Every N x N matrix of a fit lives in HBM column-major with the training size padded to a multiple of NPAD
	// (identity on the padded diagonal, zero elsewhere), so that no dense kernel needs edge handling:
	//   K_pad = [K 0; 0 I]  =>  chol, inverse factor and inverse are the padded versions of the true ones.
	constexpr int NPAD = 256;   // = N-tile of the predict kernel; multiple of every other tile size
	constexpr int CHOL_NB = 64; // panel width of the blocked Cholesky / size of the LDS-resident diagonal block

	inline size_t round_up(size_t n, size_t m) { return (n + m - 1) / m * m; }

	// ---- error plumbing ------------------------------------------------------------------------------------
	struct Ctx;
	int record_hip_error(Ctx* ctx, hipError_t e, const char* what, int line);
#define GPLE_HIP(ctx, expr)                                                        \
	do                                                                             \
	{                                                                              \
		hipError_t gple_e_ = (expr);                                               \
		if (gple_e_ != hipSuccess) return gple::record_hip_error((ctx), gple_e_, #expr, __LINE__); \
	} while (0)
#define GPLE_TRY(expr)                        \
	do                                        \
	{                                         \
		int gple_s_ = (expr);                 \
		if (gple_s_ != GPLE_OK) return gple_s_; \
	} while (0)

	// ---- device buffer -------------------------------------------------------------------------------------
	struct DevBuf
	{
		void* p = nullptr;
		size_t bytes = 0;
		DevBuf() = default;
		DevBuf(const DevBuf&) = delete;
		DevBuf& operator=(const DevBuf&) = delete;
		~DevBuf() { release(); }
		hipError_t alloc(size_t nbytes)
		{
			release();
			if (nbytes == 0) return hipSuccess;
			hipError_t e = hipMalloc(&p, nbytes);
			if (e == hipSuccess) bytes = nbytes;
			else p = nullptr;
			return e;
		}
		hipError_t ensure(size_t nbytes) { return nbytes <= bytes ? hipSuccess : alloc(nbytes); }
		void release()
		{
			if (p) (void)hipFree(p);
			p = nullptr;
			bytes = 0;
		}
		double* d() const { return static_cast<double*>(p); }
	};

	// ---- context -------------------------------------------------------------------------------------------
	struct Ctx
	{
		int device = 0;
		hipStream_t stream = nullptr;
		bool owns_stream = false;
		std::string last_error;
		std::mutex mu; // guards last_error + scratch (predict calls may come from several host threads)
		// second stream + two events for the part of a fit that does not sit on the factorisation's critical path (created on
		// first use by chol_inverse_factor; the main stream waits for the side work before anything reads its results)
		hipStream_t side_stream = nullptr;
		int side_attempts = 0;       // candidate streams tried until one ran beside the main stream (pick_side_stream)
		bool side_overlaps = false;  // the chosen one did
		hipEvent_t side_join = nullptr;
		// flags of the factorisation's one-launch scheme (gple_chol.hip, potrf_dag_kernel): never cleared, every factorisation takes a new epoch
		int* dag_flags = nullptr;
		size_t dag_flags_ints = 0;
		int dag_epoch = 0;
		// debug knobs of the factorisation (gple_debug_chol_knobs; the give-up test): scheme of this context's fits (-1: GPLE_CHOL_SCHEME, 0: a launch
		// per panel, 1: one launch per outer block), polls before a wave of the one-launch scheme gives up, workgroups of its launches (0: defaults)
		int chol_scheme = -1, dag_poll_limit = 0, dag_blocks = 0;
		// Part of a predict beside the fit it follows (gple_predict.hip, launch_predict_overlapped; GPLE_PREDICT_OVERLAP=1): the fit leaves behind the event
		// behind which the first rows of T are final (chol_inverse_factor); the early work runs on a stream and in buffers of its own (the pooled scratch is
		// ordered by the main stream: a pooled buffer may still be in use by the fit that is running)
		hipEvent_t fit_early_event = nullptr; // of the LAST fit enqueued on this context (a fork event of the factorisation, or fit_early_own)
		hipEvent_t fit_early_own = nullptr;   // recorded on the side stream behind the first row block's merge tree
		int fit_early_rows = 0;               // T(0 .. rows, .) is final behind it (0: nothing is)
		hipStream_t early_stream = nullptr;
		hipEvent_t early_points = nullptr, early_done = nullptr, early_free = nullptr; // test points in place / early work done (early stream); buffers free again (main stream)
		bool early_free_pending = false;
		long overlapped_predicts = 0; // predicts that ran their early part beside a fit (gple_debug_overlapped_predicts)
		void* early_buf = nullptr;
		size_t early_bytes = 0;
		const char* last_contraction = ""; // the kernel the last large predict's contraction ran on (gple_debug_last_contraction_kernel)
		int fused_small = -1;  // gple_debug_predict_knobs: -1 = GPLE_PREDICT_FUSED_SMALL's choice, 0 = separate kernels, 1 = predict_fused256_kernel where it applies
		int rownorm_pipe = -1; // gple_debug_predict_knobs: -1 = GPLE_ROWNORM_PIPE's choice, 0 = rownorm2_kernel, 1 = rownormp_kernel (same bits)
		long dag_giveups = 0, dag_recoveries = 0; // give-ups seen by the host / factorisations repeated with a launch per panel because of one
		std::vector<hipEvent_t> side_forks;
		// pinned host block for scalar results
		double* host_scalars = nullptr;
		// device counters of the predict path's far-row pruning: [0] blocks contracted, [1] blocks seen (lazily allocated)
		unsigned long long* prune_stats = nullptr;
		// tracing (gple_ctx_enable_timing): every timed interval takes an event pair from a free list and joins `pending`
		// until the next stream synchronisation collects it, so that timing never forces a synchronisation of its own
		struct TimedSpan
		{
			int which;
			hipEvent_t e0, e1;
		};
		bool timing = false;
		std::vector<hipEvent_t> ev_free;
		std::vector<TimedSpan> pending;
		static constexpr int NTIMERS = 4; // gple_timer
		TimedSpan open_span[NTIMERS] = {};
		double t_last[NTIMERS] = {0, 0, 0, 0}, t_total[NTIMERS] = {0, 0, 0, 0};
		long t_count[NTIMERS] = {0, 0, 0, 0};
	};
	// one interval per rownorm_kernel launch (GPLE_TIMER_PREDICT_KERNEL)
	void chunk_timer_start(Ctx* c);
	void chunk_timer_stop(Ctx* c);
	// record the start / stop event of timer `which` (no-ops unless timing is on); timer_collect() after a stream sync
	void timer_start(Ctx* c, int which);
	void timer_stop(Ctx* c, int which);
	void timer_collect(Ctx* c);

	// ---- fp64 MFMA GEMM family (gple_gemm.hip) -------------------------------------------------------------
	// C(m,n) (+)= alpha * sum_k A(m,k) * B(n,k)    [all sizes multiples of the tile]
	// Operand layouts: *_kmajor == false: element (r,k) at r + k*ld (r contiguous);  true: at k + r*ld.
	// c_trans == false: C(m,n) at m + n*ldc;  true: at n + m*ldc.
	enum KRange
	{
		K_FULL = 0,
		K_GE_N = 1,     // B(n,k) != 0 only for k >= n       -> start at the tile's first n
		K_LE_M = 2,     // A(m,k) != 0 only for k <= m       -> stop after the tile's last m
		K_GE_MAX_MN = 3 // A(m,k)!=0 for k>=m and B(n,k)!=0 for k>=n (T^T T)
	};
	struct GemmDesc
	{
		const double* A;
		long lda;
		long strideA; // per batch item
		const double* B;
		long ldb;
		long strideB;
		double* C;
		long ldc;
		long strideC;
		int M, N, K;
		int batch;
		double alpha, beta;
		int krange;     // KRange
		int lower_only; // only tiles with m0 + BM > n0 are computed (symmetric / triangular results)
		bool a_kmajor, b_kmajor, c_trans;
	};
	// tile: 64 (64x64 per workgroup), 128 (128x128 per workgroup) or 32 (32x32 per workgroup, the four waves split the k-range:
	// for latency-bound problems of a few dozen 64-tiles; falls back to 64 for operand layouts it is not instantiated for)
	hipError_t launch_gemm(hipStream_t s, const GemmDesc& d, int tile);
	// tile size for an m x n (x batch) result: 128 once there are enough 128-tiles to give every CU two workgroups
	int gemm_pick_tile(long m, long n, long batch, bool triangular);

	// ---- dense factorisation drivers (gple_chol.hip) -------------------------------------------------------
	// Lower Cholesky of the n x n (n multiple of CHOL_NB) column-major matrix A: the blocks strictly below the block
	// diagonal are overwritten by the factor; the CHOL_NB diagonal blocks of A keep their (updated, unfactored) contents — every
	// workgroup of a panel step reads its diagonal block, none may overwrite it — and the upper part is not referenced.  The
	// diagonal blocks of the factor are never stored: on return T (n x n, ldt) holds inv(L_jj) in every diagonal block (full
	// 64 x 64 blocks, zeros above the diagonal), i.e. the diagonal blocks of T = L^-1 (potrf_diag_kernel).  info (device int): 0 or 1 + index of the
	// first non-positive pivot.
	// the scheme of the factorisations this host thread issues while the object lives (scheme < 0: unchanged)
	struct CholSchemeScope
	{
		explicit CholSchemeScope(int scheme);
		~CholSchemeScope();
		CholSchemeScope(const CholSchemeScope&) = delete;
		int saved;
	};
	hipError_t potrf_lower(hipStream_t s, double* A, long lda, int n, double* T, long ldt, int* info, double* uvec = nullptr, Ctx* ctx = nullptr,
		double* tt = nullptr, double* pa = nullptr); // pa: 64 n doubles of scratch (without: a launch per panel); tt: n * n doubles — the launch also completes T = L^-1
	hipError_t debug_potrf_diag(hipStream_t s, const double* A, double* T, int* info, long long* stamps);
	void chol_layout(int n, std::vector<int>& bounds, std::vector<int>& forks, size_t& work_doubles);
	hipError_t debug_potrf_step(hipStream_t s, double* A, long lda, double* T, long ldt, int* info, long long* stamps, int pend, int below);
	// Completes T = L^-1 (lower) given its diagonal blocks; work: at least n*n/4 doubles.
	hipError_t trtri_lower_from_diag(hipStream_t s, const double* L, long ldl, double* T, long ldt, int n, double* work, int origin = 0); // origin: global column of L(0,0) (128-tile rule)
	// L = chol(A) and T = L^-1 in one go (what a fit needs): potrf_lower + trtri_lower_from_diag, with the inverse of the
	// leading half (its merge tree and the first GEMM of the last merge, more than half of the tree's work) running on the
	// context's side stream while the main stream factors the trailing half.  work: chol_inverse_work_doubles(n) doubles.
	size_t chol_inverse_work_doubles(int n);
	// uvec != nullptr: A has n + CHOL_NB rows (lda >= n + CHOL_NB), row n holds a right-hand side y and the rows below it zeros;
	// the extra block row is factored along and uvec receives u = L^-1 y (what a fit otherwise computes as T y in two more launches).
	hipError_t chol_inverse_factor(Ctx* ctx, hipStream_t s, double* A, long lda, int n, double* T, long ldt, int* info, double* work,
		double* uvec = nullptr);
	// W = T^T T (full symmetric n x n).
	hipError_t lauum_full(hipStream_t s, const double* T, long ldt, double* W, long ldw, int n);
} // namespace gple
