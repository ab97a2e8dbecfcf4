/* gple_debug.h — diagnostic entry points of libgple_hip.so that are NOT part of the drop-in C-ABI (include/gple.h): they run one
 * internal kernel family on host operands so that tests and probes can check it in isolation.
 *   tests/test_gpu_chol_diag.py, probes/diag_probe.py -> gple_debug_potrf_diag
 *   tests/test_gpu_chol_diag.py, probes/step_probe.py -> gple_debug_potrf_step
 *   tests/test_host_logic.py                          -> gple_debug_chol_layout (no device call)
 *   tests/test_gpu_gemm.py                            -> gple_debug_gemm
 *   tests/test_gpu_chol_diag.py                       -> gple_debug_side_stream, gple_debug_chol_knobs (the give-up test)
 *   tests/test_gpu_parity.py                          -> gple_debug_predict_knobs */
#ifndef GPLE_DEBUG_H
#define GPLE_DEBUG_H
#include "../../include/gple.h"
#ifdef __cplusplus
extern "C"
{
#endif
	/* One instrumented launch sequence of the diagonal-block kernel of the Cholesky panel step (gple_chol.hip, potrf_diag_kernel).
	 * A: 64 x 64 column-major SPD block; T out: inv(chol(A)); stamps: 16 shader-clock stamps of the kernel's stages;
	 * ms_per_launch: average over `reps` back-to-back launches. */
	int gple_debug_potrf_diag(gple_ctx* ctx, const double* A, double* T, long long* stamps, int reps, float* ms_per_launch);
	/* Layout of the factorisation of an n-column matrix (host logic only): outer block bounds, fork points of the block-row inverse, workspace. */
	int gple_debug_chol_layout(int n, int cap, int* bounds, int* nb, int* forks, int* nf, unsigned long long* work_doubles);
	/* The one-launch panel step (potrf_step_kernel) at block column 1 of an n x n matrix, n = 128 + below (below = 0 | 64); A and T
	 * (n x n, column-major) come back as the first launch leaves them; stamps: 24 slots — shader-clock stamps of workgroup 0, the last slot = `info` (0, or 1 + the first column whose pivot was not positive). */
	int gple_debug_potrf_step(gple_ctx* ctx, double* A, double* T, int pend, int below, long long* stamps, int reps, float* ms_per_launch);
	/* C(m,n) = alpha sum_k A(m,k) B(n,k) + beta C(m,n) by the fp64 MFMA GEMM family (gple_gemm.hip); layouts and k-ranges as GemmDesc
	 * in gple_internal.h; tile = 32 | 64 | 128 | 0 (the library's own choice). */
	int gple_debug_gemm(gple_ctx* ctx, const double* A, long lda, int a_kmajor, const double* B, long ldb, int b_kmajor, double* C, long ldc,
		int c_trans, int M, int N, int K, double alpha, double beta, int krange, int lower_only, int tile);
	/* The side stream the context's fits run their block-row inverse on (created by the first fit with n >= 1024): how many candidate streams
	 * were tried until one ran beside the main stream, and whether the chosen one did (0: none did, or no fit has needed one yet). */
	int gple_debug_side_stream(gple_ctx* ctx, int* attempts, int* overlaps);
	/* A transport for gple_set_allgather_function() that stands in for a world that is not there: rank r of P on a one-GPU box (bench.py
	 * --emulate-rank r/P).  Pass (void*)(1 + r + 256 * P) as the communicator: this rank's block is copied into its slot of the gathered buffer,
	 * the other slots are zero-filled.  What the rank computes, enqueues and unpacks is what a real rank does; the fabric is missing. */
	int gple_debug_solo_allgather(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, void* hip_stream);
	/* Test knobs of the factorisation on ONE context; a negative argument leaves its knob alone.  scheme: 0 = a launch per panel (what
	 * GPLE_CHOL_SCHEME=step selects process-wide), 1 = one launch per outer block, 2 = back to the environment's; poll_limit: polls before a waiting
	 * wave of the one-launch scheme gives up (0 = the default, 2^23); dag_blocks: workgroups of its launches (0 = one per CU).  giveups / recoveries
	 * (nullable): how often the host has seen info = -1 on this context / repeated a factorisation with a launch per panel because of it. */
	int gple_debug_chol_knobs(gple_ctx* ctx, int scheme, int poll_limit, int dag_blocks, long* giveups, long* recoveries);
	/* Test knobs of the predict path on ONE context; 0 / 1 set a knob, 2 hands it back to the environment, a negative argument leaves it alone.
	 * rownorm_pipe: the contraction kernel of large predicts — 0 = rownorm2_kernel (a barrier-to-barrier k-step), 1 = rownormp_kernel (the k-steps
	 * of a unit as one pipeline); environment: GPLE_ROWNORM_PIPE.  fused_small: real fits with N <= 256 — 0 = K* generation, contraction and sums
	 * as separate kernels, 1 = predict_fused256_kernel; environment: GPLE_PREDICT_FUSED_SMALL.  Either pair must agree bit for bit
	 * (tests/test_gpu_parity.py). */
	int gple_debug_predict_knobs(gple_ctx* ctx, int rownorm_pipe, int fused_small);
	/* Name of the kernel the last predict of this context ran its variance contraction on ("" before the first one; bench.py's roofline label). */
	const char* gple_debug_last_contraction_kernel(gple_ctx* ctx);
	/* How many predicts of this context ran their early part beside the fit they followed (GPLE_PREDICT_OVERLAP=1; tests/test_gpu_overlap.py). */
	long gple_debug_overlapped_predicts(gple_ctx* ctx);
#ifdef __cplusplus
}
#endif
#endif
