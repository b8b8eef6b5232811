/* cxrmate_hip.h -- C ABI of libcxrmate_hip.so: hand-written gfx950 (MI355X) kernels for the CXRMate hot path.
 *
 * The reference (aehrc/cxrmate) has no native boundary of its own: its model layer is a Python object API
 * (modules/transformers/{single,multi,longitudinal}_model/modelling_*.py, tools/rewards/cxrbert.py) and all arithmetic runs
 * inside third-party `transformers` / torch ops. This header is the FFI a maintainer would bind (ctypes stub in INTEGRATION.md)
 * to run that arithmetic on MI355X; each entry point names the reference / third-party code whose math it replaces.
 *   TF5:cvt  = transformers/models/cvt/modeling_cvt.py    (5.15.0)      TF5:bert = transformers/models/bert/modeling_bert.py
 *   TF5:gen  = transformers/generation/utils.py                          REF:     = file under /root/reference
 *
 * Conventions: every pointer is a DEVICE pointer owned by the caller (no allocation, no ownership transfer, no host sync);
 * activations are bf16 (raw uint16 bits) row-major "token-major" [rows, channels] with explicit leading dimensions in ELEMENTS;
 * parameters arrive as bf16 shadows (GEMM operands) or fp32 (norm scales, biases); accumulation is fp32; `stream` is a
 * hipStream_t; return 0 on success, <0 on error (-1 bad argument, -2 launch failure). Re-entrant per stream.
 */
#ifndef CXRMATE_HIP_H
#define CXRMATE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* ---- dense contractions ------------------------------------------------------------------------------------------------
 * C[M,N] = epi(alpha * A[M,K] . W[N,K]^T): every nn.Linear of CvT (TF5:cvt:171-173,224,274-293), the projection head
 * (REF:modules/transformers/single_model/modelling_single.py:25-40), BERT q/k/v/o/FFN/LM head (TF5:bert:164-203,289-351,466-496),
 * the patch-embedding convs after im2col (TF5:cvt:77-90) and all their backward products (via cxr_transpose_bf16).
 * epi: +bias[N] (fp32) -> act (0 none | 1 GELU(erf), pre-activation optionally stored to aux | 2 multiply by GELU'(aux)) -> +residual[M,N]
 * -> store bf16 or fp32 (optionally accumulating). Requires K%32==0, N%4==0, lda/ldw%8==0.
 * Train mode, between act and residual: drop_p > 0 = nn.Dropout of the dense output (TF5:bert:298,464) with the keep hash of cxr_dropout_mask
 * (row m = sequence m / drop_rows_per_b at position drop_t0 + m % drop_rows_per_b); row_scale [M / rs_rows] = CvtDropPath factor per image
 * (TF5:cvt:372), applied after the residual when rs_after != 0 (TF5:cvt:382-383 scales the whole layer output). */
int cxr_gemm_nt_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias, const void* residual, long ldr,
                     void* aux, long ldaux, int M, int N, int K, float alpha, int act, int out_f32, int accumulate, float drop_p,
                     const unsigned int* drop_seed, unsigned int drop_site, int drop_rows_per_b, int drop_t0, const float* row_scale,
                     int rs_rows, int rs_after, hipStream_t stream);
/* up to three cxr_gemm_nt_bf16 problems with equal N and K in ONE launch (the query / key / value projections of a CvT or BERT layer and their
 * input-gradient products: the small key / value GEMMs ride in the tail of the query GEMM instead of paying a launch and half-empty CUs each).
 * `d` is a HOST array; fields = the arguments of cxr_gemm_nt_bf16 in order. */
typedef struct cxr_gemm_nt_desc {
    const void* A; long lda; const void* W; long ldw; void* C; long ldc; const float* bias; const void* residual; long ldr; void* aux; long ldaux;
    int M, N, K; float alpha; int act, out_f32, accumulate; float drop_p; const unsigned int* drop_seed; unsigned int drop_site;
    int drop_rows_per_b, drop_t0; const float* row_scale; int rs_rows, rs_after;
} cxr_gemm_nt_desc;
int cxr_gemm_nt_group_bf16(const cxr_gemm_nt_desc* d, int n, hipStream_t stream);
/* nn.Conv2d(Cin, N, ksz, stride, pad) of the CvT stage embeddings (TF5 models/cvt/modeling_cvt.py:77-90, stages 2 / 3: 3 x 3, stride 2, padding 1) as an
 * IMPLICIT GEMM on a token-major activation x [Bn, Hin*Win, Cin] (bf16, batch / row strides in elements): the im2col matrix is the GEMM's A operand but
 * never exists in memory -- the kernel's LDS-DMA staging gathers each 64-channel piece of each tap from x itself (zeros outside the image).
 * W [N, ksz*ksz*Cin] in (ky, kx, c) order, C [Bn*Ho*Wo, N] bf16 = conv + bias. Requires Cin % 64 == 0. */
int cxr_gemm_nt_conv_bf16(const void* x, long x_bs, long x_rs, int Bn, int Hin, int Win, int Cin, int ksz, int stride, int pad,
                          const void* W, long ldw, void* C, long ldc, const float* bias, int N, hipStream_t stream);
/* weight gradient: C[I,J] += alpha * sum_r P[r,I] Q[r,J]  (dW += dY^T X), dbias[I] += colsum(P); token dimension split across
 * workgroups. Deterministic accumulation: one split = the workgroup owns its tile; several splits = partial tiles into `ws` (fp32, the caller's
 * scratch: 8 M floats cover every shape of this model; must not be shared by launches on different streams) + a reduce launch in split order.
 * ws NULL / too small (or CXR_TN_ATOMICS=1): fp32 atomics into C. Requires I%8==0, J%8==0. Nothing else may accumulate into C concurrently. */
int cxr_gemm_tn_bf16(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, float* dbias, int R, int I, int J, float alpha,
                     float* ws, long ws_floats, hipStream_t stream);
/* The same product with the sum over the token splits DEFERRED: the launch leaves its partial tiles in `ws` (which must then stay untouched and
 * belong to this product alone until the reduce) and fills *pending (HOST struct); cxr_gemm_tn_reduce_batch adds any number of pending sums to
 * their C / dbias in ONE launch per 40 of them, in the fixed split order of cxr_gemm_tn_bf16 (bit-identical results). pending->splits == 0: nothing
 * is pending (one split, or no usable scratch: the launch accumulated by itself). A training step has ~170 of these sums; as single launches they
 * cost 3.7 ms of weight-gradient-stream time. Replaces what torch does for the reference's Linear / Conv2d weight gradients inside loss.backward()
 * (reference modules/lightning_modules/single.py:449-475 training_step -> Lightning's backward). */
typedef struct cxr_tn_pending {
    const float* ws; const float* wsb; float* C; float* dbias; long ldc; int I, J, Ip, Jp, splits, reserved;
} cxr_tn_pending;
int cxr_gemm_tn_partial_bf16(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, float* dbias, int R, int I, int J, float alpha,
                             float* ws, long ws_floats, cxr_tn_pending* pending, hipStream_t stream);
int cxr_gemm_tn_reduce_batch(const cxr_tn_pending* pending, int n, hipStream_t stream);
/* launch plan of cxr_gemm_tn_bf16 for a shape: token splits and the scratch floats its deterministic path needs (0 with one split) */
int cxr_gemm_tn_plan(int R, int I, int J, int* splits, long* ws_floats);
int cxr_gemm_set_regstage(int on);   /* debug: 1 = stage operands through registers instead of LDS-DMA */
/* on != 0 (default): cxr_gemm_nt_bf16 may pick the persistent one-workgroup-per-CU kernels (csrc/gemm_ws.hip, csrc/gemm_pk.hip). The training
 * step turns it off while its weight-gradient stream runs kernels beside the main stream (their workgroups cannot share a CU with a 144 KB one). */
int cxr_gemm_set_exclusive(int on);
/* cxr_gemm_nt_bf16 sends tall problems (M >= min_rows, K%64==0, N%8==0, 16-byte aligned rows) to the persistent 256-row-tile kernel of
 * csrc/gemm_pk.hip (same results bit for bit). Tuning / A-B aid: enabled 0|1, bn 0 (automatic) | 128 | 256 | 1000 + i = tile configuration, wgs =
 * workgroups of a launch (0 = fill the chip); a negative argument keeps the current value. */
int cxr_gemm_pk_config(int enabled, int bn, int min_rows, int wgs);
int cxr_gemm_pk_stamps(void* out, long bytes);   /* timing experiments: copies the in-kernel s_memtime stamps of the last stamped launch to HOST memory */
/* K = 384, N % 384 == 0 problems (the channel-width products of CvT stage 3) go to the W-stationary kernel of csrc/gemm_ws.hip (weights resident in
 * registers, A row blocks streamed through LDS; same results bit for bit). Tuning / A-B aid: enabled 0|1, bm 0 (automatic) | 32 | 64 rows per block,
 * wgs = workgroups of a launch, dbg = timing-experiment bits; a negative argument keeps the current value. */
int cxr_gemm_ws_config(int enabled, int bm, int min_rows, int wgs, int dbg);
/* round 6: row-strip kernel for M x 384 x K and M x 192 x K products (csrc/gemm_strip.hip: one strip of 16 mt rows x ALL columns per workgroup).
   Tuning / A-B aid: enabled 0 | 1, mt 0 (automatic) | 2 | 4 | 6 | 10 (N = 384) | 8 | 12 | 16 (N = 192), min_rows (rows from which it is used; -2: the
   shipped thresholds), stages 0 (automatic) | 2 | 3 | 4 (stages of 32-deep steps); negative = keep */
int cxr_gemm_strip_config(int enabled, int mt, int min_rows, int stages);
int cxr_gemm_ws_stamps(void* out, long bytes);   /* timing experiments: in-kernel s_memtime stamps of the last stamped launch -> HOST memory */
int cxr_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int R, int C, hipStream_t stream);
int cxr_transpose_batched_bf16(const long* table, int n, long total_tiles, hipStream_t stream);   /* n transposes in one launch; table (device)
                                   int64 [n][8] = {in, out, ld_in, ld_out, R, C, first_tile, tiles_x}, 64x64 tiles numbered row-major per matrix */
int cxr_colsum_bf16(const void* in, long ld, float* out, int R, int C, hipStream_t stream);   /* out[c] += sum_r in[r][c] (bias grads) */

/* ---- attention -----------------------------------------------------------------------------------------------------------
 * O = softmax(scale * Q K^T + mask) V, head_dim 64, flash-style. CvT: einsum/softmax/einsum with scale = embed_dim^-0.5
 * (TF5:cvt:152,205-209, quirk Q1); BERT eager_attention_forward (TF5:bert:111-136) with causal and/or key-padding mask
 * (kpm[B,Tk] bytes, 1 = attend). Element (b,t,h,d) of X lives at X + b*x_bs + t*x_rs + h*64 + d. LSE[B,H,Tq] (natural log) optional.
 * drop_p > 0: train-mode dropout on the attention probabilities (TF5:bert:131): the keep decision of (b*H+h, drop_t0 + query, key) is the
 * counter-based hash of cxr_dropout_mask with `drop_site`; *drop_seed is read on the device. */
int cxr_attn_fwd_bf16(const void* Q, const void* K, const void* V, void* O, float* LSE, const void* kpm, long q_bs, long q_rs, long k_bs,
                      long k_rs, long v_bs, long v_rs, long o_bs, long o_rs, long kpm_bs, int B, int H, int Tq, int Tk, float scale, int causal,
                      int causal_shift, float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t0, hipStream_t stream);
/* forward with the context written as e4m3 (value * inv_scale, saturating; strides in bytes) for a consumer that is an e4m3 GEMM: no bf16 output,
 * no separate quantisation pass (frozen encoder, BASELINE.json configs[4]) */
int cxr_attn_fwd_q8_bf16(const void* Q, const void* K, const void* V, void* O8, long o8_bs, long o8_rs, float inv_scale, const void* kpm, long q_bs,
                         long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long kpm_bs, int B, int H, int Tq, int Tk, float scale, int causal,
                         int causal_shift, hipStream_t stream);
/* backward: dQ,dK,dV (contiguous [B,T,H*64]) from dO with P recomputed from (Q,K,LSE); delta[B,H,Tq] = rowsum(dO*O) is scratch */
int cxr_attn_bwd_bf16(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* delta, void* dQ,
                      void* dK, void* dV, const void* kpm, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs,
                      long o_rs, long kpm_bs, int B, int H, int Tq, int Tk, float scale, int causal, int causal_shift, float drop_p,
                      const unsigned int* drop_seed, unsigned int drop_site, int drop_t0, long dkv_bs, long dkv_rs, long dq_bs, long dq_rs, hipStream_t stream);
                      /* dkv_rs != 0 / dq_rs != 0: dK, dV / dQ are written with these batch / row strides (elements) instead of contiguously */
/* kernel generation behind the two entry points above (A/B measurements, parity tests): forward 1 = 32 query rows per wave, two barriers per tile
 * (rounds 1-2); 2 = 64 rows per wave, double-buffered tiles, one barrier (round 3, default). Backward likewise. Any other value keeps the setting. */
int cxr_attn_config(int fwd_version, int bwd_version);

/* ---- train-mode dropout / DropPath (TF5:bert:106,298,464; TF5:cvt:297-316) ---------------------------------------------------
 * out = resid + f * y with f = keep(seed, site, b, t, col)/(1-p) per element (b = row / rows_per_b, t = t0 + row % rows_per_b), or
 * f = row_scale[row / rows_per_b] (DropPath: 0 or 1/keep_prob per image). Backward applies the same call to the incoming gradient
 * (resid = NULL). cxr_dropout_mask materialises the keep bytes / factors of a site (parity tests feed them to the CPU oracle). */
int cxr_dropout_add_bf16(const void* y, long ldy, const void* resid, long ldr, void* out, long ldo, long R, int C, float p,
                         const unsigned int* seed, unsigned int site, int rows_per_b, int t0, const float* row_scale, hipStream_t stream);
int cxr_dropout_mask(unsigned char* mask, float* factor, long R, int C, float p, const unsigned int* seed, unsigned int site, int rows_per_b,
                     int t0, hipStream_t stream);
int cxr_dropout_site_factors(float* factor, int nsites, int Bn, float p, const unsigned int* seed, unsigned int site0, hipStream_t stream);
                                   /* factor[s][b] = DropPath factor of site site0 + s for image b (= cxr_dropout_mask(R = Bn, C = 1) per site), one launch */

/* ---- LoRA branch with dropout on its input (peft Linear in train mode; REF:modelling_longitudinal.py:163-170: r = 8, alpha = 32,
 * lora_dropout = 0.1 on self-attention query / key):  y = base(x) + s * B(A(dropout(x))). Eval mode merges the branch into the weight; in
 * train mode the three rank-8 contractions run here. W is addressed as W[r*w_rs + k*w_cs] (A: w_rs = K, w_cs = 1; B [N,8]: w_rs = 1, w_cs = 8);
 * t is fp32 [M][8]; masks are the counter-based hash of cxr_dropout_mask (site, row / rows_per_b, tpos0 + row % rows_per_b, column).
 *   down : t[m,r] = scale * sum_k f(m,k) x[m,k] W(r,k)   (two problems per launch: query and key; optional LayerNorm of the raw row x)
 *   up   : y[m,n] += f(m,n) * sum_r t[m,r] W(r,n)         (forward with W = B, p = 0; backward dx with W = A and the forward mask)
 *   outer: G[k*g_ks + r*g_rs] += scale * sum_m f(m,k) a[m,k] t[m,r]   (dB from (dy, t); dA from (x with the forward mask, dt)) */
int cxr_lora_down_bf16(const void* x, long ldx, long M, int K, const void* W0, long w0_rs, long w0_cs, float* t0, float p0, unsigned int site0,
                       const void* W1, long w1_rs, long w1_cs, float* t1, float p1, unsigned int site1, const unsigned int* seed, int rows_per_b,
                       int tpos0, const float* ln_gamma, const float* ln_beta, float ln_eps, float scale, hipStream_t stream);
int cxr_lora_up_add_bf16(void* y, long ldy, long M, int N, const float* t, const void* W, long w_rs, long w_cs, float p, const unsigned int* seed,
                         unsigned int site, int rows_per_b, int tpos0, hipStream_t stream);
int cxr_lora_outer_bf16(const void* a, long lda, long M, int K, const float* t, float* G, long g_ks, long g_rs, float scale, float p,
                        const unsigned int* seed, unsigned int site, int rows_per_b, int tpos0, hipStream_t stream);
/* The same three contractions for the teacher-forced pass (many rows), several problems per launch -- a BERT layer's query and key adapters
 * together (REF:modelling_longitudinal.py:163-170; TF5:bert:195-260 BertSelfAttention):
 *   down_multi  (<= 2 problems, each with its own input rows: forward x / x, backward dq / dk): matrix cores, 16 rows per workgroup, K % 32 == 0,
 *               K <= 1024, ldx % 8 == 0; no LayerNorm option
 *   up_add_multi(<= 2 problems; problems that name the SAME y are applied one after the other by the same thread -- the backward adds both
 *               adapters' terms into dx in one read-modify-write; N % 8 == 0, N <= 2048)
 *   outer_multi (<= 4 problems: dB and dA of both adapters; K % 8 == 0, K <= 2048; fp32 atomics into G as cxr_lora_outer_bf16) */
typedef struct cxr_lora_down_desc { const void* x; long ldx; const void* W; long w_rs, w_cs; float* t; float p; unsigned int site; } cxr_lora_down_desc;
typedef struct cxr_lora_up_desc { void* y; long ldy; const float* t; const void* W; long w_rs, w_cs; float p; unsigned int site; } cxr_lora_up_desc;
typedef struct cxr_lora_outer_desc { const void* a; long lda; const float* t; float* G; long g_ks, g_rs; float p; unsigned int site; } cxr_lora_outer_desc;
int cxr_lora_down_multi_bf16(const cxr_lora_down_desc* probs, int nprob, long M, int K, const unsigned int* seed, int rows_per_b, int tpos0,
                             float scale, hipStream_t stream);
int cxr_lora_up_add_multi_bf16(const cxr_lora_up_desc* probs, int nprob, long M, int N, const unsigned int* seed, int rows_per_b, int tpos0,
                               hipStream_t stream);
int cxr_lora_outer_multi_bf16(const cxr_lora_outer_desc* probs, int nprob, long M, int K, float scale, const unsigned int* seed, int rows_per_b,
                              int tpos0, hipStream_t stream);

/* ---- LayerNorm (TF5:cvt:79,363-364; REF:modelling_single.py:29; TF5:bert:103,292,350,478) ------------------------------- */
int cxr_layernorm_fwd_bf16(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, float* stats, long rows, int C,
                           float eps, hipStream_t stream);
/* LayerNorm whose consumer is an e4m3 GEMM: y8[rows, C] (row stride ldy8 bytes) = e4m3(LN(x) * inv_scale), saturating; y (bf16) may be NULL */
int cxr_layernorm_q8_bf16(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, void* y8, long ldy8, float inv_scale,
                          float* stats, long rows, int C, float eps, hipStream_t stream);           /* stats[rows][2] = (mean, rstd), optional */
int cxr_layernorm_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* gamma, const float* stats, const void* add,
                           long ldadd, void* dx, long lddx, float* dgamma, float* dbeta, float* workspace, long rows, int C, void* dx2,
                           long lddx2, float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_rows_per_b, int drop_t0,
                           const float* row_scale, hipStream_t stream);
/* workspace != NULL and dgamma == NULL: the per-workgroup (dgamma, dbeta) partial rows are left in `workspace` and this second half adds them
 * into dgamma / dbeta later (parameter gradients only feed the optimiser: the host puts it on the weight-gradient side stream) */
int cxr_layernorm_bwd_reduce(const float* workspace, long rows, int C, float* dgamma, float* dbeta, hipStream_t stream);
                           /* workspace: fp32 [cxr_layernorm_bwd_grid(rows,C)][2][C] partial sums. dx2 (optional) = f * dx: the gradient of a
                              dropout / DropPath branch fed by this LayerNorm's input (forward mask re-applied: element hash, or
                              row_scale[row / drop_rows_per_b]) */
int cxr_layernorm_bwd_grid(long rows, int C);

/* ---- CvT convolutional pieces ---------------------------------------------------------------------------------------------
 * patch-embedding Conv2d (TF5:cvt:77-90) = im2col + cxr_gemm_nt_bf16; depthwise 3x3 + BatchNorm2d projections of q/k/v
 * (TF5:cvt:93-110,157-169) on token-major activations with the class token passed through (TF5:cvt:195-198). */
int cxr_im2col_nchw_f32(const float* px, void* col, int Bn, int Cin, int H, int W, int KS, int stride, int pad, int Ho, int Wo, int Kpad,
                        hipStream_t stream);
/* The whole stage-1 patch embedding of CvT in one launch (TF5 models/cvt/modeling_cvt.py:77-90: Conv2d(3, 64, 7, stride 4, padding 2) -> flatten ->
 * LayerNorm(64)): fp32 NCHW pixels px [Bn, 3, H, W] -> y [Bn*(H/4)*(W/4), 64] bf16 = LayerNorm(conv + bias) as a direct convolution on the matrix
 * cores (no im2col matrix). e_out (may be NULL) receives conv + bias (bf16: the LayerNorm backward's input), stats (may be NULL; needs e_out) the
 * per-token (mean, rstd). Wpk = the weights packed by cxr_patch_embed_pack_f32 from the nn.Conv2d layout [64, 3, 7, 7] (fp32). Requires H % 4 == 0,
 * W % 4 == 0, W <= 384 (the reference's 384 x 384 input), 16-byte aligned px. */
int cxr_patch_embed_pack_f32(const float* w, void* out, hipStream_t stream);
int cxr_patch_embed_s1_f32(const float* px, const void* Wpk, const float* bias, const float* gamma, const float* beta, float eps,
                           void* e_out, void* y_out, float* stats, int Bn, int H, int W, hipStream_t stream);
int cxr_im2col_tok_bf16(const void* x, long x_bs, long x_rs, void* col, int Bn, int Cin, int H, int W, int stride, int pad, int Ho, int Wo,
                        hipStream_t stream);
int cxr_col2im_tok_bf16(const void* dcol, void* dx, long dx_bs, long dx_rs, int Bn, int Cin, int H, int W, int stride, int pad, int Ho, int Wo,
                        hipStream_t stream);
int cxr_bn_fold(const float* w, const float* g, const float* b, const float* mean, const float* var, float eps, float* wf, float* sh, int C,
                hipStream_t stream);
int cxr_bn_fold_bwd(const float* w, const float* g, const float* mean, const float* var, float eps, const float* G, const float* S, float* dw,
                    float* dg, float* db, int C, hipStream_t stream);
int cxr_dwconv_bn_fwd_bf16(const void* x, long x_bs, long x_rs, const float* wf0, const float* sh0, const float* wf1, const float* sh1, void* y0,
                           void* y1, long y_bs, long y_rs, int Bn, int C, int H, int W, int stride, int tok0, hipStream_t stream);
int cxr_dwconv_bn_bwd_dx_bf16(const void* dy0, const float* wf0, long bs0, long rs0, int stride0, const void* dy1, const float* wf1, long bs1,
                              long rs1, int stride1, const void* dy2, const float* wf2, long bs2, long rs2, int stride2, int nproj, void* dx,
                              long dx_bs, long dx_rs, int Bn, int C, int H, int W, int tok0, hipStream_t stream);
int cxr_dwconv_bn_bwd_w_bf16(const void* x, long x_bs, long x_rs, const void* dy, long dy_bs, long dy_rs, float* GS, float* ws, int Bn, int C,
                             int H, int W, int stride, int tok0, hipStream_t stream);   /* GS [10][C] = tap sums G[9][C] then S[C], overwritten;
                                                                                           ws = scratch, cxr_dwconv_ws_floats(C) fp32 */
int cxr_dwconv_ws_floats(int C);

/* train-mode BatchNorm2d (batch statistics; the reference trains under model.train(), and the "frozen" SCST encoder stays in train mode:
 * modules/lightning_modules/longitudinal/scst/gt_prompt.py:34-36). stats fp32 [nproj][2][C] (sum, sum of
 * squares of the raw depthwise conv outputs); finalize turns them into batch mean/rstd, moves the running statistics in place and emits the
 * folded taps for cxr_dwconv_bn_fwd_bf16. Backward: (sum dy, sum dy*c) from the same statistics kernel -> coef -> dy rewritten in place as the
 * gradient of the raw conv output -> the ordinary dx / tap-sum kernels with the RAW taps wr [9][C]. */
int cxr_dwconv_stats_bf16(const void* x, long x_bs, long x_rs, const float* w0, const float* w1, const void* dy0, const void* dy1, long dy_bs,
                          long dy_rs, float* stats, float* ws, int Bn, int C, int H, int W, int stride, int tok0, hipStream_t stream);
                          /* dy0 == NULL: (sum c, sum c^2) of the raw conv outputs c; else (sum dy, sum dy*c) for the backward */
/* statistics pass + per-channel epilogue in ONE call (two launches) for one (wt1 == NULL) or two projections; wt_i = raw taps [9][C], w_i = the
 * conv parameter [C,9]; see cxr_dwconv_stats_bf16 / cxr_bn_train_finalize / cxr_bn_train_bwd_coef for the pieces */
int cxr_dwconv_bn_train_fwd_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int stride, int tok0, float eps,
                                       float momentum, float* ws, const float* wt0, const float* w0, const float* g0, const float* b0, float* rm0,
                                       float* rv0, float* mean0, float* rstd0, float* wf0, float* sh0, const float* wt1, const float* w1,
                                       const float* g1, const float* b1, float* rm1, float* rv1, float* mean1, float* rstd1, float* wf1, float* sh1,
                                       hipStream_t stream);
int cxr_dwconv_bn_train_bwd_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int stride, int tok0, float* ws,
                                       const float* wt0, const void* dy0, const float* g0, const float* mean0, const float* rstd0, float* dg0, float* db0,
                                       float* coef0, const float* wt1, const void* dy1, const float* g1, const float* mean1, const float* rstd1,
                                       float* dg1, float* db1, float* coef1, long dy_bs, long dy_rs, hipStream_t stream);
int cxr_bn_train_finalize(const float* stats, long count, const float* w, const float* g, const float* b, float eps, float momentum,
                          float* run_mean, float* run_var, float* mean_out, float* rstd_out, float* wf, float* sh, int C, hipStream_t stream);
int cxr_bn_train_bwd_coef(const float* g, const float* mean, const float* rstd, const float* SD, long count, float* dg, float* db, float* coef,
                          int C, hipStream_t stream);
int cxr_dwconv_bn_train_dc_bf16(const void* x, long x_bs, long x_rs, const float* wr, const float* coef, void* dy, long dy_bs, long dy_rs,
                                int Bn, int C, int H, int W, int stride, int tok0, hipStream_t stream);
int cxr_tap_grad_accum(const float* G, float* dw, int C, hipStream_t stream);

/* ---- fused query / key / value convolutional projections (TF5:cvt:93-130: depthwise 3x3, pad 1, stride 1 | 2, + BatchNorm2d) ----------------
 * The three projections of a CvT layer read the same layer-normed activation x [Bn, tok0 + H*W, C] (C % 64 == 0). These entry points stage
 * each (image, row band, 64-channel slice) of x once in LDS and serve all taps of all `nproj` <= 3 projections from there; they supersede the
 * per-projection cxr_dwconv_* calls above on the training path (same arithmetic, 4-10x less L1 traffic, 8 launches per layer instead of 23).
 * `projs` is a HOST array of descriptors; all pointers inside are device pointers. Fields read per call:
 *   apply            : stride, taps (BatchNorm-folded [9][C]), shift [C], y/y_bs/y_rs (output [Bn, tok0 + Ho*Wo, C]; class rows copied through)
 *   bn_train_stats   : stride, taps (raw [9][C]), w [C,9], gamma, beta, run_mean / run_var (moved in place by `momentum`), outputs mean, rstd [C],
 *                      taps_out [9][C], shift_out [C] (the fold apply takes)          -- nn.BatchNorm2d under model.train()
 *   bn_train_bwd_stats: stride, taps (raw), y = dL/d(BN output), gamma, mean, rstd; dgamma / dbeta are ACCUMULATED, coef [3][C] written
 *   dc_taps          : stride, taps (raw), y (rewritten in place as dc = a*dy + kb + kc*c when coef != NULL; untouched when coef == NULL =
 *                      eval-mode fold), GS [10][C] (9 tap sums sum dc*x_t, then sum dc; may be NULL) and/or dw [C,9] (+= tap sums)
 *   dx               : stride, taps (raw in train mode, folded in eval mode), y = dc -> dx [Bn, tok0 + H*W, C] (class row = sum of class rows)
 * ws = fp32 scratch of cxr_dwproj_ws_floats(Bn, C, H, W) elements (partial rows of the per-channel reductions; no atomics, deterministic). */
typedef struct cxr_dwproj {
    int stride;
    const float* taps; const float* shift;
    void* y; long y_bs, y_rs;
    const float* w; const float* gamma; const float* beta; float* run_mean; float* run_var;
    float* mean; float* rstd; float* taps_out; float* shift_out;
    float* dgamma; float* dbeta; float* coef;
    float* GS; float* dw;
    const void* yf; long yf_bs, yf_rs;   /* bn_train_bwd_stats only, optional (with beta): the projection's FORWARD output [Bn, tok0 + Ho*Wo, C] bf16 -- the raw
                                            convolution output is then recovered from it (c = mean + (yf - beta) / (gamma * rstd)) instead of recomputed,
                                            per 64-channel slice wherever |gamma * rstd| >= 1e-3 */
} cxr_dwproj;
int cxr_dwproj_ws_floats(int Bn, int C, int H, int W);      /* returns the element count (> 0) or a negative error */
int cxr_dwproj_apply_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                          hipStream_t stream);
/* cxr_dwproj_apply_bf16 for consumers that are e4m3 GEMMs (frozen encoder, BASELINE.json configs[4]): projs[q].y is an e4m3 matrix (y_bs / y_rs in
 * bytes) that receives output * inv_scale[q], saturating -- no separate quantisation pass */
int cxr_dwproj_apply_q8(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                        const float* inv_scale, hipStream_t stream);
int cxr_dwproj_bn_train_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, float eps, float momentum,
                                   const cxr_dwproj* projs, int nproj, float* ws, hipStream_t stream);
int cxr_dwproj_bn_train_bwd_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                                       float* ws, hipStream_t stream);
int cxr_dwproj_dc_taps_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj, float* ws,
                            hipStream_t stream);
int cxr_dwproj_dx_bf16(void* dx, long dx_bs, long dx_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                       hipStream_t stream);
int cxr_dwproj_taps_layout(const long* table, int n, int total_blocks, hipStream_t stream);   /* raw taps [C,9] -> [9,C] for n projections in one launch;
                                   table (device) int64 [n][4] = {src, dst, C, first_block}, one workgroup per 256 destination elements */

/* ---- BERT embeddings (TF5:bert:70-108) ------------------------------------------------------------------------------------ */
int cxr_bert_embed_fwd(const long* ids, const long* tt, const long* pid, const void* word, const void* type, const void* posw,
                       const float* gamma, const float* beta, float eps, void* sum_out, void* out, float* stats, long R, int T, int pos_offset,
                       int C, float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int out_dal, hipStream_t stream);
                       /* out_dal != 0 (R <= 64): `out` in the decode activation layout of cxr_dec_gemm_bf16 */
                       /* drop_p > 0: embeddings dropout (TF5:bert:106) on the LayerNorm output, keyed (row / T, pos_offset + row % T, column) */
int cxr_bert_embed_bwd(const void* dsum, const long* ids, const long* tt, const long* pid, float* dword, float* dtype, float* dpos, long R,
                       int T, int pos_offset, long padding_idx, int C, hipStream_t stream);

/* ---- token indexing, bit-exact (REF:modules/transformers/longitudinal_model/modelling_longitudinal.py:297-364,274-277;
 *      REF:modules/transformers/multi_model/modelling_multi.py:80) ---------------------------------------------------------- */
int cxr_token_type_ids(const long* ids, long ld, int B, int T, const long* special, const long* sections, int nspecial, long* out, long ldo,
                       int past, hipStream_t stream);
int cxr_mask_position_ids(const long* ids, long ld, int B, int T, long mask_token_id, void* mask, long ldm, long* pos, long ldp,
                          hipStream_t stream);
int cxr_image_mask(const float* px, long img_stride, int BN, int tokens, void* out, hipStream_t stream);

/* ---- losses and token selection (REF:modules/lightning_modules/single.py:467-469;
 *      REF:modules/lightning_modules/longitudinal/scst/gt_prompt.py:211-246; TF5:gen:2894-2937; TopKLogitsWarper) ------------ */
int cxr_softmax_ce(const void* logits, long ld, const long* labels, long ignore_index, const float* thr, const float* row_w, float* row_loss,
                   void* dlogits, long lddl, long R, int V, int logits_bf16, hipStream_t stream);
                   /* logits fp32, or bf16 when logits_bf16 != 0 (the LM head's output dtype under the reference's bf16 autocast; softmax in fp32) */
int cxr_ce_weights(const long* labels, long R, long ignore_index, int mode, const float* reward, int T, float* row_w, hipStream_t stream);
int cxr_ce_reduce(const float* row_loss, const float* row_w, long R, float* loss, hipStream_t stream);
int cxr_topk_threshold(const float* logits, long ld, long R, int V, int k, float top_p, float temperature, float* thr, hipStream_t stream);
                       /* thr[r] = value below which TopKLogitsWarper(k) then TopPLogitsWarper(top_p) at `temperature` remove entries (top_p = 1: top-k only;
                          k <= 0: top-p only, over the whole vocabulary) */
int cxr_select_token(const float* logits, long ld, long R, int V, int mode, float temperature, int top_k, float top_p, const float* u, long* next,
                     long next_stride, int* unfinished, long eos, long pad, float* margin, int n_sample, hipStream_t stream);
                     /* next[r * next_stride] (e.g. a column of the running id buffer); mode 1 with n_sample < R: rows [0, n_sample) are sampled,
                        the rest take the argmax in the same launch (negative n_sample = all rows) */
int cxr_log_softmax_rows(float* x, long ld, long R, int V, const float* add_row, hipStream_t stream);

/* ---- fp8 (OCP e4m3fn) linear layers of the frozen encoder (BASELINE.json configs[4]; caller REF:modules/lightning_modules/longitudinal/scst/
 * gen_prompt.py:174-259, encoder under no_grad). C[M,N] = epi(scale * A8[M,K] . W8[N,K]^T): one scale per tensor (scale = s_A * s_W), fp32
 * accumulation on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales; epi = + bias, GELU (act 1), + residual (bf16); output bf16 (C) and / or
 * e4m3 (C8 = value * c8_inv_scale, saturating at +-448: the next fp8 layer's input). K % 64 == 0, lda / ldw % 16 == 0. ------------------------- */
int cxr_gemm_nt_fp8(const void* A8, long lda, const void* W8, long ldw, void* C, long ldc, void* C8, long ldc8, float c8_inv_scale, int M, int N,
                    int K, float scale, const float* bias, const void* residual, long ldr, int act, const float* row_scale, int rs_rows, int rs_after,
                    hipStream_t stream);   /* row_scale [M / rs_rows]: per-image DropPath factor on the branch (rs_after 0) or on branch + residual (1) */
int cxr_quantize_fp8(const void* x, long ldx, void* out, long ldo, int M, int K, float inv_scale, hipStream_t stream);   /* bf16 -> e4m3(x * inv_scale) */

/* ---- autoregressive decode helpers (TF5:gen:3388-3485 beam continuation search + cache reorder) ------------------------- */
int cxr_gather_batch_bf16(const void* in, long in_bs, long in_rs, void* out, long out_bs, long out_rs, const long* idx, int B, int rows, int C,
                          hipStream_t stream);
/* decode-step (one new token per row) kernels: weight-streaming GEMM for M <= 64 rows (K % 128 == 0) and single-query attention over
 * the KV cache / the cross-attention K,V (TF5:bert:164-203,230-279 with a cache, q length 1) */
int cxr_gemm_skinny_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias, const void* residual, long ldr,
                         int M, int N, int K, int act, int out_f32, const float* lnA_gamma, const float* lnA_beta, float lnA_eps,
                         float* lnA_stats, const float* lnR_stats, const float* lnR_gamma, const float* lnR_beta, float drop_p,
                         const unsigned int* drop_seed, unsigned int drop_site, int drop_t, hipStream_t stream);
                         /* lnA_gamma != NULL (K == 768): A is a RAW pre-LayerNorm sum, normalised on the fly (TF5:bert:292,350,478); (mean, rstd)
                            per row are published to lnA_stats [M][2]. lnR_stats != NULL: the residual is LayerNorm(residual) with those published
                            statistics. drop_p > 0: dropout of the dense output before the residual, rows = sequences at absolute position
                            drop_t (same hash as cxr_dropout_add_bf16) */
int cxr_gemm_skinny3_bf16(const void* A, long lda, const void* W0, const float* b0, void* C0, long ldc0, const void* W1, const float* b1,
                          void* C1, long ldc1, const void* W2, const float* b2, void* C2, long ldc2, long ldw, int M, int N, int K,
                          const float* lnA_gamma, const float* lnA_beta, float lnA_eps, float* lnA_stats, const float* lr_t0, const void* lr_B0,
                          const float* lr_t1, const void* lr_B1, const void* lr_A0, const void* lr_A1, float lr_p, const unsigned int* lr_seed,
                          unsigned int lr_site0, unsigned int lr_site1, int lr_tpos, float lr_scale, hipStream_t stream);
                          /* q / k / v projections of one decode step in a single launch; lr_t_i fp32 [M][8] / lr_B_i bf16 [N][8]: optional
                             rank-8 LoRA term of problem i (t from cxr_lora_down_bf16) 
                             lr_A_i bf16 [8][K] given (K == 768): the down-projection t_i = lr_scale * dropout(LN(A)) . lr_A_i^T is computed
                             inside the kernel (input dropout lr_p keyed by (lr_seed, lr_site_i, row, lr_tpos) as in cxr_lora_down_bf16); lr_t_i is
                             then not read */
int cxr_attn_decode_bf16(const void* Q, const void* K, const void* V, void* O, const void* kpm, long q_bs, long k_bs, long k_rs, long v_bs,
                         long v_rs, long o_bs, long kpm_bs, int B, int H, int Tk, float scale, int kv_share, float* ws, long kv_hs, float drop_p,
                         const unsigned int* drop_seed, unsigned int drop_site, int drop_t, int wg_keys, int o_dal, int kpm_bits,
                         hipStream_t stream);
                         /* kv_share = 2: K, V, kpm have B/2 rows and query rows b, b + B/2 read row b (sample + greedy halves of one SCST step
                            share the cross-attention K/V); ws (optional, B*H*8*66 fp32): lets the launch split long key ranges over workgroups
                            (flash-decoding) + a merge kernel; wg_keys = keys per workgroup pass: 0 (auto), 256, 288, 576 or 1152 (288 / 576 / 1152 tile the encoder's 576
                            tokens per image); negative = that many keys per pass but never split (one looping workgroup per row and
                            head); kpm_bits != 0: kpm holds bit words from cxr_pack_mask_bits (row stride kpm_bs in BYTES) instead of bytes; o_dal != 0: O is
                            written in the decode activation layout of cxr_dec_gemm_bf16 (B <= 64); kv_hs = head stride of K/V in elements (64 for [B,T,H*64], T*64 for head-major [B,H,T,64]); drop_t = absolute position of the query */
/* Decode-step linear layers (TF5:bert:164-203,289-351,466-496 at query length 1) in the form csrc/decode_gemm.hip explains:
 *  - weights re-laid out once per weight version into MFMA-fragment order by cxr_dec_pack_weight_bf16, optionally with the LayerNorm that feeds
 *    the layer FOLDED in: LN(x) . W^T + b = rstd * (x . W'^T - mean * colsum) + b', W' = W diag(gamma); bc fp32 [N][2] = (b', colsum);
 *  - activations between the kernels of one decode step in the "decode activation layout" (DAL): element (m, k) of an [Mpad, K] matrix,
 *    Mpad = 16 * ceil(M / 16) (48 -> 64), at ((k/32)*(Mpad/16) + m/16)*512 + ((k%32)/8*16 + m%16)*8 + k%8; cxr_dec_to_dal_bf16 /
 *    cxr_dec_from_dal_bf16 convert from / to row-major;
 *  - row statistics travel as per-16-column-tile partials: a launch with out_stats != NULL publishes (sum, M2 about the tile mean) per row for
 *    each 16-column tile of problem 0's (rounded) output, fp32 [N0/16][M][2]; a consumer names them as `stats` (+ stats_tiles = N/16) and
 *    combines them itself (Chan's update: deterministic). `stats` serves EITHER the A operand (problems with fold != 0) OR the residual
 *    (rgb != NULL: residual = LN(residual), rgb fp32 [N][2] = (gamma, beta) of that LayerNorm).
 * Up to three problems (same A) per launch; epilogue per problem: LN-fold -> +bias -> LoRA term -> act (1 = GELU) -> dropout (drop_p, rows =
 * sequences at absolute position drop_t) -> + residual (problem 0; DAL when ldr == 0) -> store (c_dal: bf16 in DAL; else row-major bf16 / fp32
 * with ldc). LoRA (train mode, REF:modelling_longitudinal.py:162-171): lr_Ap = cxr_dec_pack_lora_bf16 of A [8][K] with the feeding LayerNorm's
 * gamma / beta (NULL for a problem without fold), lr_B bf16 [N][8]: C += lr_scale * dropout_{lr_p}(LN(x)) . A^T . B^T with the mask of
 * (lr_seed, lr_site, row, lr_t). Requires M <= 64, K = 768 or 3072 (the instantiated reductions: BERT-base hidden / intermediate size),
 * N even (Wp holds 16 * ceil(N / 16) rows, zero padded; out_stats needs N % 16 == 0). mt_hint: 0 auto, 1 = one 16-row tile per workgroup,
 * 2 = all rows in one workgroup. nc_hint: 0, or 16-column tiles per workgroup (1, 4; default 4 for vocabulary-sized N). */
typedef struct cxr_dec_gemm_prob {
    const void* Wp; const float* bc; void* C; long ldc; int N, c_dal, fold, no_bias;
    const void* lr_Ap; const void* lr_B; unsigned int lr_site;
} cxr_dec_gemm_prob;
typedef struct cxr_dec_gemm_desc {
    const void* A; int M, K, nprob, act, out_f32, nc_hint, mt_hint;
    cxr_dec_gemm_prob p[3];
    const float* stats; int stats_tiles; float eps;
    const void* residual; long ldr; const float* rgb;
    float* out_stats;
    float drop_p; const unsigned int* drop_seed; unsigned int drop_site; int drop_t;
    float lr_p; const unsigned int* lr_seed; float lr_scale; int lr_t;
} cxr_dec_gemm_desc;
int cxr_dec_gemm_bf16(const cxr_dec_gemm_desc* d, hipStream_t stream);      /* d is a HOST struct (device pointers inside) */
int cxr_dec_pack_weight_bf16(const void* W, long ldw, const float* gamma, const float* beta, const float* bias, void* Wp, float* bc, int N, int K,
                             hipStream_t stream);      /* W bf16 [N,K] row-major; gamma / beta NULL: no fold (colsum still written) */
int cxr_dec_pack_lora_bf16(const void* A, const float* gamma, const float* beta, void* out, int K, hipStream_t stream);   /* A bf16 [8][K] -> out bf16 [16*K] */
int cxr_dec_to_dal_bf16(const void* x, long ldx, int M, int K, void* out, float* stats, hipStream_t stream);   /* out and/or the out_stats partials of x */
int cxr_dec_from_dal_bf16(const void* x, int M, int K, void* out, long ldo, hipStream_t stream);
/* cross-attention of a cached decode step on the matrix cores. Kp / Vp: fragment-ordered copies of the studies' cross K / V (cxr_pack_cross_kv_bf16,
 * once per decode: K, V [Bkv, Tk, H*64] with batch / row strides kv_bs / kv_rs -> Bkv*Tk*H*64 elements each, 1 KB per MFMA fragment);
 * kpm_bits uint32 [Bkv][mb_words] (cxr_pack_mask_bits) or NULL; kv_share = B / Bkv <= 4 query rows per K/V stream (rows b + g*Bkv); Tk % 32 == 0,
 * Tk <= 9216; output as cxr_attn_decode_bf16 (same dropout hash; probabilities enter P.V as bf16) */
int cxr_pack_cross_kv_bf16(const void* K, const void* V, long kv_bs, long kv_rs, void* Kp, void* Vp, int Bkv, int H, int Tk, hipStream_t stream);
int cxr_attn_cross_mfma_bf16(const void* Q, const void* Kp, const void* Vp, void* O, const unsigned int* kpm_bits, long q_bs, long o_bs, long mb_words,
                             int B, int H, int Tk, float scale, int kv_share, float drop_p, const unsigned int* drop_seed, unsigned int drop_site,
                             int drop_t, int o_dal, float* ws, hipStream_t stream);
/* cxr_attn_cross_mfma_bf16 with the cross-attention QUERY projection inside the kernel: xA = the raw hidden rows [x_M, 768] in the decode activation
 * layout (x_mtl 16-row tiles), xstats = the producer's partial row statistics fp32 [x_tiles][x_M][2] (as cxr_dec_gemm_bf16 publishes them), qWp / qbc =
 * the query Linear packed by cxr_dec_pack_weight_bf16 with the LayerNorm folded in. H * 64 == 768, kv_share <= 4, Tk <= 1920. One launch instead of
 * cxr_dec_gemm_bf16 (query) + cxr_attn_cross_mfma_bf16 per layer and token-step. */
int cxr_attn_cross_mfma_q_bf16(const void* xA, int x_mtl, int x_M, const float* xstats, int x_tiles, float x_eps, const void* qWp, const float* qbc,
                               const void* Kp, const void* Vp, void* O, const unsigned int* kpm_bits, long o_bs, long mb_words, int B, int H, int Tk,
                               float scale, int kv_share, float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t, int o_dal,
                               hipStream_t stream);   /* ws (B*H*8*66 floats): needed for Tk > 1152 (key range split + merge) */
int cxr_pack_mask_bits(const void* kpm, long kpm_bs, int B, int T, unsigned int* out, int words, hipStream_t stream);
                       /* key-padding bytes [B,T] (1 = attend) -> uint32 [B][words], bit k%32 of word k/32 */
int cxr_topk_rows(const float* x, long ld, long R, int n, int K, float* vals, long* inds, hipStream_t stream);
/* One beam-search step of all studies in one launch (TF5:gen:3208-3560 `_beam_search`, do_sample=False, early_stopping=False, one EOS id):
 * log-softmax + running scores, the 2*beams best continuations, running / finished split, merge into the finished set, improvement test,
 * cache-reorder indices. Rows are beam-major (row = beam * B + study). logits fp32 [beams*B, V] raw; running / sequences int64 [beams,B,L];
 * run_scores / beam_scores fp32 [B,beams]; finished u8 [B,beams]; unsat / allhit int32 [2,B] (double-buffered by the parity of `cur`; initial
 * state: unsat = 1, allhit = 0 in both halves); beam_idx int64 [beams*B] out. div = (cur + 1 - prompt_len) ** length_penalty. Once no study
 * can improve (or all candidates hit max_length) later launches leave the state unchanged and emit the identity reorder. Two kernels: a scan
 * over (vocabulary chunk, row) workgroups -> ws (beams*B * ceil(V/4096) * (2 + 4*beams) floats), then one workgroup per study. */
int cxr_beam_step(const float* logits, long ld, long* running, long* sequences, float* run_scores, float* beam_scores, unsigned char* finished,
                  int* unsat, int* allhit, long* beam_idx, float* ws, int B, int nb, int V, int L, int cur, int max_length, long eos, float div,
                  hipStream_t stream);
/* out[i][r, :rows, :] = in[i][idx[r], :rows, :] for n <= 16 tensors [B, *, C] of one geometry (host arrays of device pointers): the KV-cache
 * reorder of a beam step for all layers in one launch (TF5:gen:3468-3478) */
int cxr_gather_batch_multi_bf16(const void* const* in, void* const* out, int n, long bs, long rs, const long* idx, int B, int rows, int C,
                                hipStream_t stream);

/* ---- input pipeline tail (next-row f3): ToTensor + Normalize + pad_sequence of the reference collate (REF:modules/lightning_modules/
 * single.py:248-262, multi.py:155-164). src: packed uint8 HWC images of all studies; first_image int64 [B+1] (prefix sums of images per
 * study); dst fp32 [B, Nmax, 3, H, W], absent images = 0.0 (quirk Q3). */
int cxr_pixels_u8_to_f32(const void* src, const long* first_image, float* dst, int B, int Nmax, int H, int W, float mean0, float mean1,
                         float mean2, float std0, float std1, float std2, hipStream_t stream);

/* ---- CheXbert labeller heads (next-row f4; REF:tools/chexbert.py:74-81): out[r][s] = argmax over columns [offsets[s], offsets[s+1]) of
 * the stacked head logits (lowest index wins ties, like torch.argmax); offsets int32 [nseg+1], out int64 [R][nseg]. */
int cxr_segment_argmax_f32(const float* x, long ld, const int* offsets, int nseg, long* out, long R, hipStream_t stream);

/* ---- cached-decode step inputs in one launch (REF:modelling_longitudinal.py:251-295 prepare_inputs_for_generation and its single / multi
 * copies): from ids[r, strip:cur] -> new_id = last token, tt = token_ids_to_token_type_ids_past (:340-364; rows < half_rows use special0, the
 * rest special1), mask = ids != mask_token_id and pos = relu(cumsum(mask)-1)[-1] (:274-277; mask == NULL skips both); tt / pos are also written
 * to column `cur` of the optional per-row histories. */
int cxr_decode_step_inputs(const long* ids, long ld, int rows, int strip, int cur, const long* special0, int n0, const long* special1, int n1,
                           const long* sections, int half_rows, long mask_token_id, long* new_id, long* tt, long* pos, void* mask, long ldm,
                           long* tt_hist, long* pos_hist, long ldh, hipStream_t stream);

/* cxr_decode_step_inputs fused with the BERT embeddings of the new token (cxr_bert_embed_fwd for one position per row: word[new_id] + type[tt] +
 * position[pos, or the absolute position when mask == NULL] -> LayerNorm -> dropout keyed by (row, absolute position)): `out` bf16 [rows, 768],
 * in the decode activation layout when out_dal != 0. One launch per token instead of two; cur - strip <= 512, at most 4 separators per set. */
int cxr_decode_step_embed(const long* ids, long ld, int rows, int strip, int cur, const long* special0, int n0, const long* special1, int n1,
                          const long* sections, int half_rows, long mask_token_id, long* new_id, long* tt, long* pos, void* mask, long ldm,
                          long* tt_hist, long* pos_hist, long ldh, const void* word, const void* type, const void* posw, const float* gamma,
                          const float* beta, float eps, void* out, int out_dal, float drop_p, const unsigned int* drop_seed,
                          unsigned int drop_site, hipStream_t stream);

/* ---- reward (REF:tools/rewards/cxrbert.py:66-71 torch.nn.functional.cosine_similarity of the CLS projections) ---------------- */
int cxr_cosine_rows_f32(const float* a, long lda, const float* b, long ldb, float* out, long R, int C, float eps, hipStream_t stream);

/* hipError_t of the most recent failed launch (return code -2); cxr_last_hip_error_string() gives its text */
int cxr_last_hip_error(void);

/* ---- optimiser and plumbing (REF:modules/lightning_modules/single.py:426-431 torch.optim.AdamW defaults) ------------------ */
int cxr_adamw_step(float* p, const float* g, float* m, float* v, void* p16, long n, float lr, float b1, float b2, float eps, float wd, int step,
                   const int* step_ptr, float gscale, hipStream_t stream);   /* step==0: bias corrections from the device counter *step_ptr */
int cxr_increment_i32(int* p, hipStream_t stream);
int cxr_cast_f32_to_bf16(const float* in, void* out, long n, hipStream_t stream);
int cxr_cast_bf16_to_f32(const void* in, float* out, long n, hipStream_t stream);
int cxr_gelu_bwd_bf16(const void* dy, const void* u, void* dx, long n, hipStream_t stream);   /* dx = dy * GELU'(u), contiguous */
int cxr_add_bf16(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long rows, int C, hipStream_t stream);
int cxr_copy_rows_bf16(const void* in, long in_bs, long in_rs, void* out, long out_bs, long out_rs, int B, int rows, int C, hipStream_t stream);
int cxr_bcast_row_f32_bf16(const float* row, void* out, long out_bs, int B, int C, hipStream_t stream);
int cxr_sum_row0_bf16_f32(const void* in, long in_bs, float* out, int B, int C, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
