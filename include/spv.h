/*
 * spv.h -- C-ABI of libspv_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * Spectre-ViT encoder training step.
 *
 * This is the drop-in boundary below the reference's nn.Module surface.  The
 * reference has no native layer at all: every entry point here replaces a chain of stock ATen ops
 * issued by a reference module, cited per function as  <reference file>:<lines>.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the duration of the call (nothing is kept,
 *     nothing is allocated); tensors are dense row-major unless a leading dimension is given;
 *   - `dtype` arguments: SPV_F32 (0) or SPV_BF16 (1); statistics, parameters, gradients of
 *     parameters and all accumulation are fp32 regardless;
 *   - `stream` is the hipStream_t to launch on (the caller's current stream); calls never
 *     synchronise and are hipGraph-capturable;
 *   - return value: 0 = ok, non-zero = error (bad shape / alignment / launch failure); the
 *     message is available from spv_last_error() (thread local).
 */
#ifndef SPV_H
#define SPV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPV_F32 0
#define SPV_BF16 1

#define SPV_ABI_VERSION 1

int spv_version(void);
/* Dispatch census (test aid; no reference counterpart): how many calls each kernel family has served in this process, so a
 * parity test can assert that the shapes it ran really took the fused / strip kernels the benchmark times. */
enum {
    SPV_PATH_GEMM_STRIP = 0,     /* gemm_nt_strip_kernel<*, false> */
    SPV_PATH_GEMM_STRIP_ACC = 1, /* gemm_nt_strip_kernel<*, true>  */
    SPV_PATH_GEMM_TN = 2,        /* weight-gradient kernel (spv_gemm_tn) */
    SPV_PATH_TAIL_LC = 3,        /* lane-contiguous SpectreLinear tail kernels */
    SPV_PATH_TAIL_UP = 4,        /* spv_spectre_tail_bwd_up */
    SPV_PATH_TAIL_LN = 5,        /* spv_spectre_tail_ln_fwd / _bwd */
    SPV_PATH_FNET_MFMA = 6,      /* fnet_mfma_kernel (bf16, dim 512) */
    SPV_PATH_GATHER_LDS = 7,     /* LDS-staged MHPermutMix gather */
    SPV_PATH_GEMM_TN_WIDE = 9,   /* gemm_tn_wide_kernel (256 x 128 tile; M % 256 == 0, N % 128 == 0) */
    SPV_PATH_GEMM_TN_DMA = 8,    /* gemm_tn_dma_kernel (LDS-DMA ring; M, N % 128 == 0, K % 64 == 0) */
    SPV_PATH_GEMM_TN_BATCH = 10, /* gemm_tn_batch_kernel: up to eight weight gradients in one launch (spv_gemm_tn_batch) */
    SPV_PATH_GEMM_STRIP_POOL = 11, /* gemm_nt_strip_kernel<*, 2 / 3>: data gradient + pooled-broadcast term (spv_gemm_nt_pool_bwd) */
    SPV_PATH_GEMM_ROWS = 13,     /* gemm_nt_rows_kernel: few-rows NT GEMM, one 32 x 32 tile per workgroup, no split-K (the CLS-only last layer) */
    SPV_PATH_PERMUT_ROW0 = 12,   /* spv_permut_row0_fwd / _bwd: MHPermutMix at token row 0 (the CLS-only last layer) */
    SPV_PATH_COUNT = 16
};
long long spv_path_count(int which);
const char* spv_last_error(void);

/* ---- precision plumbing -------------------------------------------------------------------
 * torch.autocast's weight cast (spectre_vit/repl/train.py:219) as one kernel; `transpose` also
 * writes the [cols][rows] copy used by the data-gradient GEMM. */
int spv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* dst[c][r] = src[map(r)][c]; dst leading dimension ld >= rows (pad zero filled).  rows_per_group > 0 reads the
 * grouped source row (r / rows_per_group) * group_stride + row_offset + r % rows_per_group (e.g. the token rows
 * of every image without its CLS row). */
int spv_cast_transpose(const void* src, int src_dtype, void* dst, int dst_dtype, int rows, int cols, int ld,
                       int rows_per_group, int group_stride, int row_offset, void* stream);
/* spv_weight_shadows: both operand layouts of an fp32 nn.Linear weight [rows, cols] (layers.py:85-86) in one pass: the plain
 * copy in the compute dtype (plain may be NULL: fp32 reads the parameter itself) and the transposed copy [cols, ld >= rows]
 * for the data-gradient GEMM.  Rebuilt every training step: optimizers update parameters in place. */
int spv_weight_shadows(const float* w, void* plain, void* transposed, int rows, int cols, int ld, int dtype, void* stream);
/* The same for every weight of a model in one launch: `table` = device array of {const float* src; void* plain; void* transposed;
 * int rows, cols, ld, pad;}; workgroup b serves the 32 x 64 tile (tile_x[b], tile_y[b]) of tensor tile_tensor[b]. */
int spv_weight_shadows_multi(const void* table, const int* tile_tensor, const int* tile_x, const int* tile_y, int ntiles, int dtype,
                             void* stream);

/* ---- dense contraction: C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (+ C) ------------------------
 * nn.Linear inside SpectreLinear (spectre_vit/models/spectre/layers.py:85-86,100), its data and
 * weight gradients, and the patch projection (spectre.py:148).  MFMA: v_mfma_f32_32x32x16_bf16
 * (bf16 in) or v_mfma_f32_32x32x2_f32 (exact fp32 in); fp32 accumulate.
 * K, lda, ldb must be multiples of 16 bytes / sizeof(in element); A,B 16-byte aligned.
 * splits > 1: split-K over `splits` slices, fp32 partial slabs in `workspace`
 * (>= splits*M*N*4 bytes), reduced by a second kernel (deterministic). */
int spv_gemm_nt(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda,
                int ldb, int ldc, int in_dtype, int out_dtype, int accumulate, int splits,
                void* workspace, void* stream);
/* Same contraction, output row m written to row (m / rows_per_group) * group_stride + row_offset + m % rows_per_group
 * and bias2d[m % rows_per_group][N] added: the patch projection writing straight into the token tensor
 * (position embedding + bias as bias2d, every image's CLS row skipped) -- spectre.py:148-152. */
int spv_gemm_nt_grouped_rows(const void* A, const void* B, const float* bias, const float* bias2d, void* C, int M,
                             int N, int K, int lda, int ldb, int ldc, int in_dtype, int out_dtype,
                             int rows_per_group, int group_stride, int row_offset, void* stream);
/* The same with the dropout that follows the projection (spectre.py:156) applied in the epilogue: the mask of spv_dropout /
 * spv_embed_bwd for the same seed over the flat index of the (contiguous, ldc == N) output. */
int spv_gemm_nt_grouped_rows_drop(const void* A, const void* B, const float* bias, const float* bias2d, void* C, int M, int N, int K,
                                  int lda, int ldb, int ldc, int in_dtype, int out_dtype, int rows_per_group, int group_stride,
                                  int row_offset, float p_drop, uint64_t seed, void* stream);

/* Data gradient of a SpectreLinear whose skip pools exact windows (in = pool_window * out, the MHPermutMix linear,
 * layers.py:66,93): C[M,N] = A[M,K] . B[N,K]^T + dout[M, N/pool_window][.., n / pool_window] / pool_window -- the
 * transposed average pooling is added in the epilogue instead of going through a rows x N buffer. */
int spv_gemm_nt_pool_bwd(const void* A, const void* B, void* C, const void* dout, int pool_window, int M, int N, int K,
                         int lda, int ldb, int ldc, int in_dtype, int out_dtype, int dout_dtype, void* stream);

/* Multi-GPU runs: keep n CUs (0..128) free of the one-workgroup-per-CU layer GEMM (it needs a whole CU, so a CU that holds a resident
 * RCCL channel would push its workgroup into a second dispatch round).  spectre_vit.dp.GradReducer sets 16 when world_size > 1; the
 * reference has no counterpart (single device, train.py:41). */
int spv_set_reserved_cus(int n);
/* TN contraction C[M,N] = sum_k A[k][m] B[k][n], A [K,lda>=M], B [K,ldb>=N] row-major bf16: the weight gradient
 * dW = dh^T . x (backward of layers.py:86) straight from the row-major activations (transposing LDS reads,
 * ds_read_b64_tr_b16); M, N, lda, ldb multiples of 8; split-K as above. */
int spv_gemm_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int out_dtype,
                int accumulate, int splits, void* workspace, void* stream);
/* The same, with up to sixteen folds of row kernels' per-workgroup partial column sums riding in the split-K reduce launch as extra
 * workgroups: out[p][c] = sum_w partials[w][p][c], p < nsum <= 5 (dgamma, dbeta, dbias[, dgamma2, dbeta2]), c < n, fixed order.
 * The row kernel (spv_spectre_tail_bwd*, spv_spectre_tail_ln_bwd, spv_fnet_ln_bwd) is then called with its parameter-gradient
 * pointers NULL: it writes the partials and launches no fold (parts = spv_tail_bwd_parts(rows) for the tails, batch for the FNet
 * kernel).  With splits == 1 the folds run as one launch of their own.  Why: a 5-us launch between two large kernels costs the
 * step ~20 us (eight per-layer fold launches removed: 157 us of 2.26 ms). */
typedef struct spv_fold_job {
    const float* partials;
    float* out[5];
    int parts, nsum, n;
} spv_fold_job;
int spv_tail_bwd_parts(int rows);
/* Folds with no reduce to ride in: up to any number of jobs in one launch per sixteen (the end-of-backward flush of folds that were
 * held back for a later spv_gemm_tn_fold which never came; sixteen per launch). */
int spv_fold_multi(const spv_fold_job* folds, int nfolds, void* stream);
int spv_gemm_tn_fold(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int out_dtype,
                     int accumulate, int splits, void* workspace, const spv_fold_job* folds, int nfolds, void* stream);
/* Up to 8 weight gradients with the same K (C_i [m_i, n_i] fp32 = A_i[K, lda_i]^T . B_i[K, ldb_i], bf16 operands) in ONE launch
 * + ONE split-K reduce that also carries up to 16 fold jobs.  workspace: splits * sum(m_i * n_i) floats.  For gradients that
 * nobody reads before the backward pass is over (spectre_vit/hip_ops.py holds them back; reference: the nn.Linear weight
 * gradients autograd computes node by node, layers.py:85-101). */
typedef struct spv_tn_problem {
    const void* a;
    const void* b;
    void* c;
    int m, n, lda, ldb, ldc;
    int k;   /* rows THIS problem reduces over: 0 or K = the call's K (split-K as asked); 1..K-1 = a short problem (the CLS-only last
              * layer's 512-row gradients beside the 33 280-row ones): its first k rows, unsplit, stored by the GEMM launch itself */
} spv_tn_problem;
int spv_gemm_tn_batch(const spv_tn_problem* probs, int nprob, int K, int splits, void* workspace, const spv_fold_job* folds, int nfolds,
                      void* stream);
/* The same call as two: part = 1 launches the batched GEMM only, part = 2 the split-K reduce + folds only (same arguments both
 * times).  For a measuring caller that brackets each launch with its own pair of HIP events (bench.py's roofline pass). */
int spv_gemm_tn_batch_part(const spv_tn_problem* probs, int nprob, int K, int splits, void* workspace, const spv_fold_job* folds,
                           int nfolds, int part, void* stream);

/* ---- SpectreLinear tail: out = dropout(GELU_erf(LayerNorm(h)) + adaptive_avg_pool(x)) ---------
 * spectre_vit/models/spectre/layers.py:85-101 (LN eps 1e-5, nn.GELU exact, AdaptiveAvgPool1d over the
 * channel axis; identity skip when k_in == n) + the nn.Dropout that follows it in
 * spectre.py:70-73.  h: [rows,n] pre-norm linear output, x: [rows,k_in] the layer input.
 * Saves mean/rstd [rows] fp32 for the backward.  p_drop == 0 disables dropout. */
int spv_spectre_tail_fwd(const void* h, const void* x, const float* gamma, const float* beta, void* out,
                         float* mean, float* rstd, int rows, int n, int k_in, int dtype, int out_dtype,
                         float p_drop, uint64_t seed, void* stream);

/* Backward of the tail.  dout [rows,n] (dout_dtype) -> dh [rows,n] (dtype), dx_pool [rows,k_in] (dtype; the
 * transposed pooling of the masked dout, to which the caller accumulates dh.W), and the fp32
 * column sums dgamma/dbeta/dbias [n].  `partials` is fp32 scratch of spv_rowop_partial_floats(n)
 * floats.  `dx_add` (nullable, [rows,k_in], dtype): a residual-stream gradient folded into dx_pool
 * (x1 feeds both norm2's residual and linear1, spectre.py:67,70-73), instead of a separate add pass. */
int spv_spectre_tail_bwd(const void* dout, const void* h, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, void* dh, void* dx_pool, float* dgamma,
                         float* dbeta, float* dbias, float* partials, int rows, int n, int k_in, int dtype,
                         int dout_dtype, float p_drop, uint64_t seed, const void* dx_add, void* stream);
int64_t spv_rowop_partial_floats(int n);
/* spv_spectre_tail_bwd for the 512 -> 768 layer (linear1 of the encoder layer, layers.py:85-101 under spectre.py:70-73) that ALSO
 * takes the skip gradient of the layer above (linear3, 768 -> 512) at its source: dout_eff = dout + AdaptiveAvgPool1d^T(mask_up * up_src),
 * up_src [rows, 512] = the gradient that entered linear3's tail (`ds` of spv_spectre_tail_ln_bwd), up_p_drop / up_seed = linear3's
 * dropout.  linear3's backward is then called with dx_pool = NULL and its data-gradient GEMM stores instead of accumulating.
 * spv_tail_up_supported: 1 for (n, k_in) = (768, 512). */
int spv_tail_up_supported(int n, int k_in, int dtype);
int spv_spectre_tail_bwd_up(const void* dout, const void* h, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, void* dh, void* dx_pool, float* dgamma, float* dbeta, float* dbias, float* partials,
                            int rows, int n, int k_in, int dtype, int dout_dtype, float p_drop, uint64_t seed, const void* dx_add,
                            const void* up_src, float up_p_drop, uint64_t up_seed, void* stream);
/* Second half of the encoder layer's elementwise work as ONE kernel each way:
 *   f3 = SpectreLinear3-tail(h3, f1) (as spv_spectre_tail_fwd), x2 = LayerNorm2(x1 + f3)   (spectre.py:67, 70-73)
 * spv_tail_ln_supported: shapes covered (512 outputs from 768 inputs, either dtype); otherwise compose
 * spv_spectre_tail_* with spv_add_layernorm_* (mode 1).
 *   fwd: out = f3 (kept: the backward re-forms x1 + f3 from it), out2 = x2, mean/rstd of the tail's LayerNorm,
 *        mean2/rstd2 of LayerNorm-2; res = x1.
 *   bwd: ds = LayerNorm2-backward(dout2) (written: the residual branch needs it) is used at once as the tail's incoming
 *        gradient; dh, dx_pool (NULL: not formed, see spv_spectre_tail_bwd_up), dgamma, dbeta, dbias as spv_spectre_tail_bwd; dgamma2,
 *        dbeta2 of LayerNorm-2;
 *        partials: spv_tail_ln_partial_floats(n) floats. */
int spv_tail_ln_supported(int n, int k_in, int dtype);
int64_t spv_tail_ln_partial_floats(int n);
int spv_spectre_tail_ln_fwd(const void* h, const void* x, const float* gamma, const float* beta, void* out, float* mean, float* rstd,
                            const void* res, const float* gamma2, const float* beta2, void* out2, float* mean2, float* rstd2, int rows,
                            int n, int k_in, int dtype, float p_drop, uint64_t seed, void* stream);
int spv_spectre_tail_ln_bwd(const void* dout2, const void* f3, const void* res, const float* mean2, const float* rstd2,
                            const float* gamma2, void* ds, float* dgamma2, float* dbeta2, const void* h, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, void* dh, void* dx_pool, float* dgamma,
                            float* dbeta, float* dbias, float* partials, int rows, int n, int k_in, int dtype, float p_drop,
                            uint64_t seed, void* stream);

/* ---- residual + LayerNorm ------------------------------------------------------------------------
 * mode 0: out = LN(a) + b      norm1(mix(x)) + x   spectre_vit/models/spectre/spectre.py:66
 * mode 1: out = LN(a + b)      norm2(x + ff(x))    spectre.py:67
 * backward returns d(LN input) in `din` and dgamma/dbeta; the residual gradient of mode 0's `b`
 * is dout itself. */
/* One-level Haar DWT along the embedding axis + LayerNorm-1 + residual as one row kernel each way (bf16, dim 512 or 1024):
 * out = LN(haar(x)) * gamma + beta + x -- SpectreEncoderLayer's norm1(mix(x)) + x (spectre.py:66) with the 'dwt_embed' mixer of
 * BASELINE config 3 (repl/dwt_experiments.py:56).  Same values as spv_haar_dwt followed by spv_add_layernorm_fwd(mode 0), the bf16
 * rounding of the band tensor between them included; the backward recomputes haar(x) from x, nothing of the mixer is stored.
 * dgamma == NULL: the caller folds `partials` (parts = spv_tail_bwd_parts(rows), nsum = 2, n = dim). */
int spv_haar_ln_supported(int dim, int dtype);
int spv_haar_ln_fwd(const void* x, const float* gamma, const float* beta, void* out, float* mean, float* rstd, int rows, int dim,
                    int dtype, void* stream);
int spv_haar_ln_bwd(const void* dout, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                    float* dgamma, float* dbeta, float* partials, int rows, int dim, int dtype, void* stream);
int spv_add_layernorm_fwd(const void* a, const void* b, const float* gamma, const float* beta, void* out,
                          float* mean, float* rstd, int rows, int n, int mode, int dtype, void* stream);
int spv_add_layernorm_bwd(const void* dout, const void* a, const void* b, const float* mean,
                          const float* rstd, const float* gamma, void* din, float* dgamma, float* dbeta,
                          float* partials, int rows, int n, int mode, int dtype, void* stream);

/* ---- MHPermutMix gather: g[b,f] = x[b, perm[f]] * sign[f], f in [0, heads*d) ------------------------
 * spectre_vit/models/spectre/layers.py:68-72 (advanced-index gather, sign multiply, raw reshape to
 * (B, tokens, embed*heads) -- the reshape is free: g is written flat).  `idx` is the packed table
 * from spv_permut_pack: bit31 = sign (1 => -1), bits 0..30 = source index.
 * backward: dx[b,i] = sum_h sign[h,inv_h(i)] * dg[b,h,inv_h(i)] (each perms[h] is a permutation:
 * no atomics, head by head in LDS). */
/* idx: spv_permut_table_words(heads, d) uint32 words, opaque to the caller: the wide tables uint32 [2][heads][d] -- [0] forward
 * (perm | sign), [1] inverse (inverse perm | sign) -- followed, when d <= 65 536 and d % 8 == 0, by their compact form (uint16
 * indices + one sign bit per element) which the bf16 kernels read instead: every workgroup walks the whole table, so its
 * width is L2 traffic; and, when a bf16 row of d elements does not fit the LDS (Spectre-ViT-Base at 224 / 16), by two sets
 * of scatter lists (backward and forward): per (head, quarter of the output row) the sources whose target lies in that quarter, in
 * ascending order. */
int64_t spv_permut_table_words(int heads, int d);
/* 1 when spv_permut_gather_fwd emits `pooled` for this window on rows longer than the LDS (any window that divides d and a quarter of it) */
int spv_permut_pool_supported(int heads, int d, int pool_window, int dtype);
int spv_permut_pack(const int64_t* perms, const float* signs, uint32_t* idx, int heads, int d, void* stream);
/* pooled (nullable, [batch, heads*d / pool_window]): the average of every pool_window consecutive gathered elements
 * (what the SpectreLinear skip needs), produced on the fly so the tail kernel does not re-read g. */
int spv_permut_gather_fwd(const void* x, const uint32_t* idx, void* g, void* pooled, int pool_window, int batch,
                          int heads, int d, int dtype, void* stream);
int spv_permut_gather_bwd(const void* dg, const uint32_t* idx, void* dx, int batch, int heads, int d,
                          int dtype, void* stream);
/* Token row 0 of the gathered matrix only (the last layer of a stack whose consumer reads the CLS row): g0[b][c] = +-x[b][idx[c]]
 * for the first n = heads * embed entries of the forward table (reference layers.py:71-72: row t of the raw view is the chunk
 * [t n, (t + 1) n) of the flattened (heads, d) gather), x0[b] = the sample's own row 0.  Backward: dx[b] = dx0[b] in row 0, zero
 * elsewhere, + the n scattered values.  idx: the table of spv_permut_pack (its forward part comes first). */
int spv_permut_row0_fwd(const void* x, const uint32_t* idx, void* g0, void* x0, int batch, int d, int n, int embed, int dtype,
                        void* stream);
int spv_permut_row0_bwd(const void* dg0, const void* dx0, const uint32_t* idx, void* dx, int batch, int d, int n, int embed,
                        int dtype, void* stream);


/* ---- FNet token mixer y = Re(fft2(x)) over (tokens, dim), un-normalised ------------------------------
 * spectre_vit/models/spectre_branch/spectre_branch.py:79, repl/orthogonal_permut.py:23-28 ('fft_bare',
 * spectre.py:31).  The operator is symmetric, so the same call is its own backward.
 * `twiddle`: token-axis cos/sin table of spv_fnet_twiddle_floats(tokens) floats filled once by
 * spv_fnet_make_twiddle (the caller caches it).  `workspace`: fp32 scratch of
 * spv_fnet_workspace_floats(batch,tokens,dim) floats (0 on the LDS fast path: dim a power of two,
 * tokens <= 79 and (tokens+3)*dim*4 <= 160 KiB).  `add_in` (nullable, same shape/dtype as y) is added to the
 * output: in the backward, the residual-stream gradient that bypasses the mixer (spectre.py:66). */
int spv_fnet_mix(const void* x, void* y, const void* add_in, const float* twiddle, int batch, int tokens, int dim,
                 int dtype, float* workspace, void* stream);
int64_t spv_fnet_workspace_floats(int batch, int tokens, int dim);
/* First half of the encoder layer as ONE kernel each way: x1 = LayerNorm1(Re(fft2(x))) + x
 * (spectre_vit/models/spectre/spectre.py:66 with the FFT mixer).  spv_fnet_ln_supported tells whether the fused kernels
 * cover a shape (bf16, dim 512, 2 <= tokens <= 65); otherwise callers compose spv_fnet_mix with spv_add_layernorm_*.
 *   fwd: prenorm = Re(fft2(x)) (kept for the backward), out = LN(prenorm) * gamma + beta + x, mean / rstd [batch*tokens].
 *   bwd: dx = Re(fft2(LN1-backward(dout))) + dout; dgamma / dbeta [dim]; partials: batch * 2 * dim floats of scratch. */
int spv_fnet_ln_supported(int tokens, int dim, int dtype);
int spv_fnet_ln_fwd(const void* x, void* prenorm, void* out, const float* gamma, const float* beta, float* mean, float* rstd,
                    const float* twiddle, int batch, int tokens, int dim, int dtype, void* stream);
int spv_fnet_ln_bwd(const void* dout, const void* prenorm, const float* mean, const float* rstd, const float* gamma, void* dx,
                    float* dgamma, float* dbeta, float* partials, const float* twiddle, int batch, int tokens, int dim, int dtype,
                    void* stream);

/* Row 0 only of the same node, x1[:, 0, :] = LayerNorm1(Re(fft2(x))[0, :]) + x[:, 0, :] -- the last layer of a stack whose
 * consumer reads the CLS row (reference spectre.py:66 + 198).  Token frequency 0 is the sum over tokens: one pass over the sample
 * and ONE dim-point FFT.  out [batch, dim] (dtype), m0 [batch, dim] fp32 (the pre-norm row, kept for the backward), mean / rstd
 * [batch].  Backward: g1 [batch, dim] -> dx [batch, tokens, dim] (every row the same spectrum, row 0 + g1), partials
 * [batch][2][dim] = each sample's dgamma / dbeta contribution (summed by a fold job: spv_fold_multi / a riding fold). */
int spv_fnet_cls_supported(int tokens, int dim, int dtype);
int spv_fnet_cls_fwd(const void* x, const float* gamma, const float* beta, void* out, float* m0, float* mean, float* rstd, int batch,
                     int tokens, int dim, int dtype, void* stream);
int spv_fnet_cls_bwd(const void* g1, const float* m0, const float* mean, const float* rstd, const float* gamma, void* dx,
                     float* partials, int batch, int tokens, int dim, int dtype, void* stream);
int64_t spv_fnet_twiddle_floats(int tokens);
int spv_fnet_make_twiddle(float* twiddle, int tokens, void* stream);

/* FFT module: y[..,k] = sum_d x[..,d] cos(2 pi k d / D), k in [0, D/2]  (rfft(x).real)
 * spectre_vit/modules/spectre.py:9-14.  transpose=1 computes the adjoint (the backward). */
int spv_rfft_real(const void* x, void* y, int rows, int dim, int transpose, int dtype, void* stream);

/* ---- Haar DWT mixers ('dwt_embed' along dim, 'dwt_token' along tokens) ---------------------------
 * named in spectre_vit/models/spectre/spectre.py:33-34; only call site repl/dwt_experiments.py:56.
 * Pairs a=(x0+x1)/sqrt2, d=(x0-x1)/sqrt2 (PyWavelets' 'haar'), output [a_J | d_J | ... | d_1] (pywt.wavedec's order).  `inverse`
 * bit 0: apply the adjoint (the backward; = the inverse when the map is orthonormal).  bit 1 selects what happens to the unpaired
 * last element of an odd length: 0 = it passes through into the approximation band (orthonormal), 1 = pywt's mode="zero" of the
 * reference's call (dwt_experiments.py:56): paired with a zero, a_last = x_last / sqrt2, the duplicate d_last is not stored. */
int spv_haar_dwt(const void* x, void* y, int batch, int tokens, int dim, int axis, int levels, int inverse,
                 int dtype, void* scratch, void* stream);

/* ---- dropout seeds under HIP-graph replay ----------------------------------------------------------------
 * The dropout masks of spectre.py:70-73 / vit.py:30-38 come from a counter hash of (seed, element); seeds are by-value kernel
 * arguments, frozen when a graph is captured.  spv_set_seed_device_ptr(word) makes every dropout kernel add the 64-bit device
 * word to its seed (NULL restores eager behaviour); spv_seed_advance(word, stream) steps it -- captured once per training step, so
 * every replay draws fresh masks.  No reference counterpart (the reference is eager PyTorch). */
int spv_set_seed_device_ptr(const void* seed_word);
int spv_seed_advance(void* seed_word, void* stream);

/* ---- AdamW over many tensors in one launch ------------------------------------------------------------
 * The optimizer step the script drives: torch.optim.AdamW(lr, betas, weight_decay), spectre_vit/repl/train.py:199-201,237
 * (decoupled weight decay, bias correction; amsgrad / maximize off).  `table`: device array of {float* p; const float* g;
 * float* m; float* v;} per tensor; `sizes[t]` its element count; workgroup c updates elements [chunk_off[c], chunk_off[c] + 2048)
 * of tensor chunk_tensor[c].  one_minus_beta1/2: 1 - beta rounded from double by the caller (torch forms them in double: 1 - 0.999
 * taken in fp32 is off by 1.3e-5).  bias_correction1/2 = 1 - beta^step computed by the caller, or -- step_dev != NULL, for HIP-graph
 * capture -- taken from the device-side step count *step_dev (already advanced for this step). */
int spv_adamw_multi(const void* table, const int* chunk_tensor, const int* chunk_off, const int* sizes, int nchunks, float lr,
                    float beta1, float beta2, float one_minus_beta1, float one_minus_beta2, float eps, float weight_decay,
                    float bias_correction1, float bias_correction2, const float* step_dev, void* stream);

/* ---- Walsh-Hadamard butterflies along the last axis -----------------------------------
 * spectre_vit/models/spectre/hadamar.py: fwht :12-32 / hadamard_transform :83-112 (mode 0, natural order; scale = n^-1/2
 * when normalised), fwht_fast :58-80 (mode 1: every stage interleaves sum / difference, un-normalised) and its transpose
 * (mode 2: the backward of mode 1; mode 0 is its own transpose).  LearnableHadamard.forward :127-141 is ONE call: rows of
 * n_in values zero-padded to n (a power of two), `repeat` = num_blocks passes, the first n_out values kept, `residual`
 * (nullable, [rows, n_out]) added.  x [rows, n_in], y [rows, n_out]; n <= 16384. */
int spv_fwht(const void* x, void* y, const void* residual, int rows, int n_in, int n, int n_out, int mode, int repeat,
             float scale, int dtype, void* stream);

/* ---- patch embedding ---------------------------------------------------------------------------
 * tokens[b,0,:] = cls + pos[0]; tokens[b,1+n,:] = W_full . patch(b,n) + bias + pos[1+n]
 * where patch(b,n) is the (c,p,q)-ordered P x P pixel block.  With W_full = conv weight this is
 * PatchEmbedding (spectre_vit/modules/patch_embeddings.py:28-43); with
 * W_full = (proj.weight * freq_h (x) freq_w) . R, R = Re(rfft2 ortho) it is SpectralPatchEmbed
 * (spectre_vit/models/spectre/spectre.py:124-156).  The contraction itself is spv_gemm_nt_grouped_rows;
 * these entry points are the plumbing around it:
 *   spv_patchify        pixel blocks -> [B*Np, ld>=K] rows (transposed = 0), the [K, ld>=B*Np] transpose (1), or token rows
 *                       [B][Np+1][ld>=K] whose first row per image -- the CLS slot -- is zero (2: the token GEMM and the TN
 *                       weight-gradient GEMM read that layout as it lies)
 *   spv_embed_posbias   bias2d[t][e] = pos[1+t][e] + bias[e]; with cls != NULL one more row in front: cls[e] + pos[0][e]
 *                       (out then has patches + 1 rows and the GEMM over the zero CLS rows writes the CLS tokens)
 *   spv_embed_cls_rows  tokens[b,0,:] = cls + pos[0]
 *   spv_spectral_fold / _bwd   W_full from (proj.weight, freq_h, freq_w) and the gradients back
 *                       (scratch: embed*chans*patch*(patch/2+1) floats)
 *   spv_dropout         y = x * mask(seed, index) / (1-p): nn.Dropout (spectre.py:154) with a counter-based
 *                       mask that the backward regenerates (same call on the gradient). */
int spv_patchify(const float* img, void* out, int batch, int chans, int height, int width, int patch, int ld,
                 int transposed, int out_dtype, void* stream);
/* spv_patchify_u8: the same rows from the data loader's uint8 HWC batch [B,H,W,C], normalised on the fly as
 * ToTensor + Normalize(mean, std) do on the host (spectre_vit/repl/train.py:102-112):
 * (pixel / 255 - mean[c]) * inv_std[c]. */
int spv_patchify_u8(const unsigned char* img_hwc, const float* mean, const float* inv_std, void* out, int batch, int chans,
                    int height, int width, int patch, int ld, int transposed, int out_dtype, void* stream);
int spv_embed_posbias(const float* pos, const float* bias, const float* cls, float* out, int patches, int embed, void* stream);
int spv_embed_cls_rows(const float* cls, const float* pos, void* tokens, int batch, int tokens_per_image, int embed,
                       int dtype, void* stream);
int spv_spectral_fold(const float* proj_w, const float* freq_h, const float* freq_w, float* w_full, int embed,
                      int chans, int patch, void* stream);
/* spv_spectral_fold that also writes the bf16 copy the token GEMM reads in a bf16 step (one launch instead of fold + cast) */
int spv_spectral_fold_bf16(const float* proj_w, const float* freq_h, const float* freq_w, float* w_full, void* w_full_bf16, int embed,
                           int chans, int patch, void* stream);
int spv_spectral_fold_bwd(const float* dw_full, const float* proj_w, const float* freq_h, const float* freq_w,
                          float* dproj_w, float* dfreq_h, float* dfreq_w, float* scratch, int embed, int chans,
                          int patch, void* stream);
int spv_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);
/* Backward of the token tensor in one pass + one fold: dtok = dropout_mask(g [+ gcls on the CLS rows]) (the mask of the forward's
 * spv_dropout with the same seed over the same flat index; dtok may be NULL when it would equal g), and the sums over the batch that
 * dpos [tokens][embed], dbias[e] = sum_{t >= 1} dpos[t][e] and dcls[e] = dpos[0][e] are (gradients of position_embeddings, proj.bias
 * and cls_token, spectre.py:133-135,150-155).  partials: spv_embed_bwd_groups(batch) * tokens * embed floats. */
int spv_embed_bwd_groups(int batch);
int spv_embed_bwd(const void* g, const void* gcls, void* dtok, float* partials, float* dpos, float* dbias, float* dcls, int batch,
                  int tokens, int embed, float p_drop, uint64_t seed, int dtype, void* stream);

/* ---- softmax attention core for the baseline ViT ------------------------------------------------
 * nn.MultiheadAttention inside the stock nn.TransformerEncoderLayer, spectre_vit/models/vit/vit.py:30-38:
 * ctx = dropout(softmax(Q K^T / sqrt(head_dim))) V per (sequence, head).
 * qkv [seqs, len, 3*heads*head_dim] (q|k|v), ctx/dctx [seqs, len, heads*head_dim], probs and dscores
 * [seqs, heads, len, len] (probs is saved for the backward, dscores is scratch of the same size).
 * len <= 1024, head_dim <= 128.  Which tensor axis is `len` is the caller's business: the reference feeds (B,N,E) with
 * batch_first=False, so len = B. */
int spv_attention_fwd(const void* qkv, void* ctx, void* probs, int seqs, int len, int heads, int head_dim, int dtype,
                      float p_drop, uint64_t seed, void* stream);
int spv_attention_bwd(const void* dctx, const void* qkv, const void* probs, void* dscores, void* dqkv, int seqs, int len,
                      int heads, int head_dim, int dtype, float p_drop, uint64_t seed, void* stream);

/* ---- generic helpers used by the module mirror ------------------------------------------------------
 * GELU (TransformerEncoderLayer MLP, vit.py:30-36), column sums (bias / position-embedding gradients;
 * partials: >= min(rows,512)*n floats), a*x + b*y. */
int spv_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int spv_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream);
int spv_colsum(const void* x, float* out, float* partials, int rows, int n, int dtype, void* stream);
int spv_axpby(const void* x, const void* y, void* out, float a, float b, int64_t n, int dtype, void* stream);

/* ---- the classifier end of the step: SpectreLinear over a few rows, and mean cross-entropy -------------------------
 * The class head  mlp_head = SpectreLinear(embed, classes)  (spectre_vit/models/spectre/spectre.py:190-192, applied to
 * (x + src)[:, 0] :199-202) and  nn.CrossEntropyLoss()  (spectre_vit/repl/train.py:196,226).  For rows <= 4096,
 * n <= 128, k <= 1024, n <= k (spv_small_sl_supported) the whole SpectreLinear -- x = xa[row*lda] (+ xb[row*ldb]),
 * h = x W^T + b over the fp32 master weights, LayerNorm, exact-erf GELU, adaptive-average-pooled skip -- is ONE
 * launch; its backward two (rows: dh, dx, partial column sums; weights: dW and the fold of dgamma/dbeta/dbias).
 * out, h, xs (the summed input, fp32 [rows,k]), mean, rstd are written by the forward and read by the backward;
 * partials: spv_small_sl_partial_floats(rows, n) floats; dtype = type of xa / xb / dx. */
int spv_small_sl_supported(int rows, int n, int k);
int64_t spv_small_sl_partial_floats(int rows, int n);
int spv_small_sl_fwd(const void* xa, int64_t lda, const void* xb, int64_t ldb, const float* W, const float* bias,
                     const float* gamma, const float* beta, float* out, float* h, float* xs, float* mean, float* rstd,
                     int rows, int n, int k, int dtype, void* stream);
int spv_small_sl_bwd(const float* dout, const float* h, const float* xs, const float* mean, const float* rstd,
                     const float* W, const float* gamma, const float* beta, float* dh, void* dx, float* dW,
                     float* dgamma, float* dbeta, float* dbias, float* partials, int rows, int n, int k, int dx_dtype,
                     void* stream);
/* loss = mean_r(logsumexp(logits[r]) - logits[r][labels[r]]) (fp32 logits [rows, classes], int64 labels; a label outside
 * [0, classes) makes the loss NaN).  lse [rows] is kept for the backward: dlogits = (softmax - onehot) * grad_out[0] / rows.
 * workspace: spv_cross_entropy_workspace_floats() floats, ZEROED ONCE by the caller (it holds an arrival counter that
 * every launch re-arms); the sum over rows is taken in a fixed order. */
int64_t spv_cross_entropy_workspace_floats(void);
int spv_cross_entropy_fwd(const float* logits, const int64_t* labels, float* lse, float* loss, float* workspace, int rows,
                          int classes, void* stream);
int spv_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* lse, const float* grad_out,
                          float* dlogits, int rows, int classes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPV_H */
