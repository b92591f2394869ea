/*
 * otter_hip.h -- C ABI of libotter_hip.so: the MI355X (gfx950) kernels of the Otter vision->language fusion hot path.
 *
 * The reference (Luodian/Otter) is 100% Python and has NO FFI / operator registry for this path:
 * every entry point below replaces a run of ATen calls inside one of the reference's nn.Module.forward bodies
 * (and their autograd backward).  The file:line each one replaces is given per function, relative to
 * /root/reference/src/otter_ai/models/.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: device pointers + sizes; no torch / hip types in the signatures (streams travel as void*).
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates persistent memory;
 *     scratch is passed in (`*_workspace_bytes` tells how much).
 *   - asynchronous on the given stream (hipStream_t cast to void*; NULL = default stream); thread-safe for
 *     distinct streams; no host synchronisation inside.
 *   - return 0 on success, a negative otter_status on failure (never throws); otter_last_error() gives a
 *     thread-local message.
 *   - matrices are row-major; `ld*` / `*_stride` are ROW strides in ELEMENTS.
 *   - dtypes: OTTER_F32 / OTTER_BF16 (bf16 = upper 16 bits of an IEEE float, round-to-nearest-even on store).
 *     All reductions / accumulations are fp32 whatever the storage type.
 */
#ifndef OTTER_HIP_H
#define OTTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTTER_ABI_VERSION 3   /* bump on ANY change of an exported signature or struct layout (otter_amd/_capi.py reads this line) */

typedef enum { OTTER_F32 = 0, OTTER_BF16 = 1 } otter_dtype;

typedef enum {
    OTTER_OK = 0,
    OTTER_ERR_ARG = -1,      /* bad shape / dtype / null pointer */
    OTTER_ERR_UNSUPPORTED = -2,
    OTTER_ERR_LAUNCH = -3,   /* hipGetLastError() after the launch */
    OTTER_ERR_WORKSPACE = -4
} otter_status;

int otter_abi_version(void);
const char* otter_last_error(void);
/* number of CUs etc. are not needed by callers; this just proves the library sees a gfx950 device. */
int otter_device_check(void);

/* ---------------------------------------------------------------------------------------------------------
 * LayerNorm / RMSNorm.  Replaces nn.LayerNorm at otter/modeling_otter.py:136-137,144,211,253,365 (perceiver +
 * gated cross-attention), LPLayerNorm at mpt/norm.py:16-45 (beta == NULL: MPT-7B has no_bias) and LlamaRMSNorm at
 * /root/reference/xformers_model/llama.py:95-112.
 *
 * Row map: output row of input row r is  (r / grp_rows) * grp_stride + row_off + (r % grp_rows)  when
 * grp_rows > 0, else r.  It lets the perceiver write norm_media(x) and norm_latents(latents) straight into
 * the [x ; latents] buffer that `to_kv` consumes (modeling_otter.py:166) without a torch.cat copy.
 * y2 (optional, may be NULL) receives the same rows again, un-mapped (dense [rows, D]).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t grp_rows;   /* 0 = identity map */
    int64_t grp_stride; /* rows */
    int64_t row_off;    /* rows */
} otter_rowmap;

int otter_layernorm_fwd(const void* x, int x_dtype, const void* gamma, const void* beta, int w_dtype, void* y,
                        int y_dtype, otter_rowmap y_map, void* y2, float* mean, float* rstd, int64_t rows, int64_t D,
                        float eps, void* stream);

/* Fused residual add + LayerNorm: xsum = x + delta (stored with x's dtype), y = LN(xsum).  Replaces the
 * `x = x + resid_attn_dropout(b); m = norm_2(x)` pair of MPTBlock.forward (mpt/blocks.py:83-84): one pass over the
 * residual stream instead of two.  Backward = otter_layernorm_bwd with dres = d(xsum). */
int otter_add_layernorm_fwd(const void* x, int x_dtype, const void* delta, int delta_dtype, void* xsum, const void* gamma,
                            const void* beta, int w_dtype, void* y, int y_dtype, float* mean, float* rstd, int64_t rows,
                            int64_t D, float eps, void* stream);

/* dx = LN'(dy) (+ dres if given, same dtype as dx).  dy is read through dy_map (same convention as the forward's
 * y_map).  dx_bf16 (optional) receives a bf16 copy of dx in the same pass (the gradient of the bf16 branch output that was
 * added to the fp32 residual stream).  dgamma/dbeta (fp32, [D]) are OVERWRITTEN (or accumulated when accumulate != 0).
 * ws: see otter_layernorm_bwd_workspace_bytes.  gamma may be NULL (treated as ones); dbeta may be NULL. */
int64_t otter_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D);
int otter_layernorm_bwd(const void* dy, int dy_dtype, otter_rowmap dy_map, const void* x, int x_dtype,
                        const void* gamma, int w_dtype, const float* mean, const float* rstd, const void* dres,
                        void* dx, int dx_dtype, void* dx_bf16, float* dgamma, float* dbeta, int accumulate, void* ws,
                        int64_t rows, int64_t D, void* stream);

/* out[c] (+)= sum_r src[map(r)][c]  (fp32 [D]).  Gradient of a broadcast embedding row: frame_embs /
 * media_time_embs (modeling_otter.py:224-229).  ws: otter_layernorm_bwd_workspace_bytes(rows, D). */
int otter_colsum(const void* src, int src_dtype, otter_rowmap src_map, float* out, int accumulate, void* ws, int64_t rows,
                 int64_t D, void* stream);

/* y = w * cast_to_xdtype(x * rsqrt(mean(x^2) + eps))   (HF LlamaRMSNorm rounding order) */
int otter_rmsnorm_fwd(const void* x, int x_dtype, const void* w, int w_dtype, void* y, float* rstd, int64_t rows,
                      int64_t D, float eps, void* stream);
int otter_rmsnorm_bwd(const void* dy, const void* x, int x_dtype, const void* w, int w_dtype, const float* rstd,
                      void* dx, float* dw, int accumulate, void* ws, int64_t rows, int64_t D, void* stream);

/* LLaMA host (config C4) forms of the two above.  otter_add_rmsnorm_fwd: optional fused residual add (xsum = x + delta, both
 * NULL for a plain norm) and an output dtype of its own (y = bf16 of the fp32 result: what autocast feeds the next Linear,
 * rounded once).  Replaces `hidden = residual + hidden; hidden = post_attention_layernorm(hidden)` of the decoder layer
 * (/root/reference/xformers_model/llama.py:311-318) in one pass over the residual stream.  otter_rmsnorm_bwd_ex: dy in its
 * own dtype, optional dres added to dx, optional bf16 copy of dx, dw optional (NULL for the frozen decoder). */
int otter_add_rmsnorm_fwd(const void* x, int x_dtype, const void* delta, int delta_dtype, void* xsum, const void* w, int w_dtype,
                          void* y, int y_dtype, float* rstd, int64_t rows, int64_t D, float eps, void* stream);
int otter_rmsnorm_bwd_ex(const void* dy, int dy_dtype, const void* x, int x_dtype, const void* w, int w_dtype, const float* rstd,
                         const void* dres, void* dx, int dx_dtype, void* dx_bf16, float* dw, int accumulate, void* ws, int64_t rows,
                         int64_t D, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue( A[M,K] . B[N,K]^T )  -- both operands K-contiguous ("NT": exactly nn.Linear).
 * Replaces every bias-free nn.Linear on the path (modeling_otter.py:139-141,145,147,255-257,366,368) with the
 * element-wise tail that follows it fused into the epilogue:
 *   OTTER_EPI_STORE     C = s * acc                      (s = tanh(*gate) if gate else 1; accumulate: C += ...)
 *                       with an fp32 C and partial != NULL also partial[block] = sum over the tile of C^2 (the values as stored): the
 *                       clip_grad_norm_ reduction of a weight gradient (pipeline/train/instruction_following.py:246-247) taken in the
 *                       producing launch instead of by a second sweep over the tensor; summed by otter_clip_coef like any other partial.
 *   OTTER_EPI_GELU      C = gelu_erf(acc); C2 = acc       (Linear -> nn.GELU, :366-367 / :145-146; C2 optional)
 *                       with aux_is_gelu_input == 3 (round 6c): C2 = gelu_erf'(acc) -- the DERIVATIVE is stashed for a backward that needs
 *                       nothing else of the pre-activation (a frozen MLP: mpt/blocks.py:9-20), see OTTER_EPI_GATE_BWD
 *   OTTER_EPI_SCALE_RES C = acc * s + R                  (x = attn(...) * attn_gate.tanh() + x, :380-393;
 *                                                          gate == NULL gives the perceiver's plain residual :180,184)
 *   OTTER_EPI_GATE_BWD  C = s * acc * f'(aux);  partial[block] = sum(acc * f(aux))
 *                       f = identity (aux_is_gelu_input == 0), gelu_erf (1) or the squared ReLU relu(a)^2 of the Persimmon MLP
 *                       (2; /root/reference/src/otter_ai/models/fuyu/modeling_persimmon.py:180-194); 3: aux already holds f'(a)
 *                       (stashed by the forward launch): C = s * acc * aux, partial must be NULL.
 *                       This is the dgrad GEMM of a gated branch: it yields d(branch input) and, through the
 *                       deterministic two-stage reduction otter_reduce_partials, the gradient of the scalar gate.
 * A/B are both ab_dtype.  bf16 operands run on MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate); f32 operands
 * run on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) -- the parity mode.
 * ------------------------------------------------------------------------------------------------------- */
typedef enum { OTTER_EPI_STORE = 0, OTTER_EPI_GELU = 1, OTTER_EPI_SCALE_RES = 2, OTTER_EPI_GATE_BWD = 3 } otter_epilogue;

typedef struct {
    int kind;            /* otter_epilogue */
    int accumulate;      /* STORE only, C must be f32 */
    const float* gate;   /* device scalar (pre-tanh) or NULL */
    const void* R;       /* SCALE_RES residual [M,N] */
    int64_t ldr;
    int r_dtype;
    void* C2;            /* GELU: pre-activation copy (same dtype as C) or NULL */
    int64_t ldc2;
    const void* aux;     /* GATE_BWD */
    int64_t ldaux;
    int aux_dtype;
    int aux_is_gelu_input; /* 0 identity, 1 erf GELU, 2 squared ReLU, 3 stashed derivative (GELU launch: C2 = GELU'; GATE_BWD launch: aux = f') */
    float* partial;      /* GATE_BWD, STORE with an f32 C: [otter_gemm_num_partials(M,N)] floats, or NULL */
    int grid_mode;       /* otter_grid_mode of THIS launch (ABI 2): how the large-grid kernel maps tiles to workgroups */
} otter_epilogue_args;

/* Per-call grid shape of the large-grid GEMM.  A persistent grid (one workgroup per CU walking its tile list) assumes the launch owns
 * the chip; while a collective's kernels hold CUs (a DP reducer is live) one workgroup per tile degrades in proportion to the CUs
 * taken instead of doubling the launch.  The caller that knows -- otter_amd.train.TrainStep -- says so per
 * launch; OTTER_GRID_DEFAULT defers to the process-wide otter_gemm_set_persistent switch (tools / A-B runs only). */
typedef enum { OTTER_GRID_DEFAULT = 0, OTTER_GRID_PERSISTENT = 1, OTTER_GRID_PER_TILE = 2 } otter_grid_mode;

int64_t otter_gemm_num_partials(int64_t M, int64_t N, int ab_dtype);
int otter_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                  int64_t K, int ab_dtype, int c_dtype, const otter_epilogue_args* epi, void* stream);
/* C[M,N] = epilogue(op(A) . op(B)^T) with K-MAJOR operands (round 3): with a_kmajor != 0, A is handed over as [K rows][M columns]
 * row-major (row stride lda), i.e. the reduction index is the ROW index; likewise B as [K][N] with b_kmajor.  These are the
 * layouts the backward products of an nn.Linear have their operands in -- dW[out,in] = dy^T x (both K-major: K = token rows;
 * autograd's `grad_output.t().mm(input)`, the backward of every nn.Linear at otter/modeling_otter.py:140-147,254-256,366-368) and
 * dx = dy W with W as stored [out,in] (B K-major) -- so no operand is transposed in HBM first.  Same epilogues, dtypes and stream
 * semantics as otter_gemm_nt (which is this call with both flags 0).  Only shapes for which otter_gemm_kmajor_supported returns 1
 * are accepted (bf16, K % 128 == 0 -- any K >= 128 when BOTH operands are K-major --, >= 192 output tiles of 256 x 256, M / N / leading
 * dimensions multiples of 8, operands < 4 GB);
 * the host transposes (otter_transpose) and calls otter_gemm_nt otherwise. */
int otter_gemm(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C, int64_t ldc, int64_t M,
               int64_t N, int64_t K, int ab_dtype, int c_dtype, const otter_epilogue_args* epi, void* stream);
int otter_gemm_kmajor_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int a_kmajor, int b_kmajor, int ab_dtype);
/* Threading: every compute entry point of this header is asynchronous on the stream it is given and may be called concurrently from
 * different host threads on DISTINCT streams (autograd's backward thread and the training thread do).  The otter_*_set_* switches below
 * (variant, debug, CU budget, grid shape) and the otter_prof_* hooks are process-wide settings for tools, benchmarks and the training
 * step's set-up phase: change them from one thread, between steps, never while another thread is launching.  The one-time
 * hipFuncSetAttribute calls behind the first launch of each kernel are idempotent (a benign race). */
/* selects the bf16 kernel schedule: 0 = auto, 1 = 128x128 register-staged, 2 = 256x256 register-staged,
 * 3 = 256x256 direct-to-LDS (global_load_lds).  Process-wide; for A/B measurements. */
int otter_gemm_set_variant(int variant);
/* 1 if `variant` is compiled into this library.  The product build carries 0-3, 13, 25, 26 (the kernels pick_cfg can choose);
 * the earlier kernel generations (4-12, 14-23, 27-29) live in the tools-only experimental build
 * (`python -m otter_amd.build --experimental` -> lib/libotter_hip_experimental.so, -DOTTER_EXPERIMENTAL). */
int otter_gemm_variant_available(int variant);
/* Caps the persistent GEMM grids at `cus` workgroups (0 = every CU of the device, the default; values below 8 are raised to 8).
 * Returns the grid size now in effect.  Used by the data-parallel step: with `cus` a little below the device's CU count, RCCL's
 * all-reduce kernels get CUs of their own and run concurrently with the backward GEMMs (nothing to match in the reference:
 * torch DDP relies on the GPU scheduler).  Process-wide, like the other otter_gemm_set_* switches: set it from the thread that
 * launches the GEMMs, between steps. */
int otter_gemm_set_cu_budget(int cus);
/* on = 0: the large-grid kernel (variant 26) is launched with ONE workgroup per output tile instead of one persistent workgroup per CU.
 * A persistent grid assumes it owns every CU: when a concurrent kernel (RCCL's all-reduce in the data-parallel step) holds a few CUs, the
 * workgroups that could not start run their whole tile list AFTER the others finish and the launch takes twice as long; with one
 * workgroup per tile the loss is proportional to the CUs taken (measured).  Costs ~8 us of fixed time per launch
 * when nothing else runs.  Default 1.  Process-wide, set between steps. */
int otter_gemm_set_persistent(int on);
/* Diagnostics and A/B switches of the GEMM kernels, one word.  Never set by the product path.
 *   bits 0-1  roofline ablations (results are WRONG): bit0 = no global loads inside the K loop, bit1 = no MFMAs
 *   bit 4 / 6 masked (edge-tile) tail on every tile / tile-phase timeline stamps (otter_gemm_read_timeline); bit 8: 4-wide fused tail
 *   bits 9-13 tile order of the large-grid kernel (super-tile shape, walk direction; bit 13 = one workgroup per tile)
 *   bits 14-15 (round 6) cross-tile ring of the large-grid kernel: 0 = process default (on; env OTTER_GEMM_XT), 1 = off, 2 = on
 *   bits 16-22 + bit 23 (round 6): with bit 23 set, bits 16-22 replace the K-tile order of a tile (default 3 = rotated by the tile's N panel;
 *             0 = plain: the order the bit-for-bit comparisons against other kernels use; env OTTER_GEMM_KORDER sets the process default) */
int otter_gemm_set_debug(int flags);

/* out[0] (op) = scale(gate) * sum(partial[0..n))   with scale = (1 - tanh(*gate)^2) when gate != NULL.
 * accumulate != 0 adds into out[0].  Deterministic (single block, fixed order). */
int otter_reduce_partials(const float* partial, int64_t n, const float* gate, float* out, int accumulate, void* stream);

/* dst[c][r] = src[r][c]  (+ optional same-layout cast copy dst_same[r][c]).  Used for the weight shadows
 * (fp32 master -> bf16 W and bf16 W^T) and for the activation transposes the wgrad GEMMs need. */
int otter_transpose(const void* src, int64_t ld_src, int src_dtype, void* dst_t, int64_t ld_dst_t, void* dst_same,
                    int64_t ld_dst_same, int dst_dtype, int64_t rows, int64_t cols, void* stream);
int otter_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * text_time: cumulative count of <image> tokens per text position, plus the attend_previous=False rewrite.
 * Replaces modeling_otter.py:298-311.  media_locations: uint8/bool [B,T]; text_time: int32 [B,T].
 * ------------------------------------------------------------------------------------------------------- */
int otter_text_time(const uint8_t* media_locations, int32_t* text_time, int64_t B, int64_t T, int attend_previous,
                    void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Attention core (head_dim = 64 only -- the reference hard-codes dim_head=64 for both users):
 *   perceiver latent cross-attention  modeling_otter.py:168-179   (mask_mode NONE, Tq = 64 latents, M = n1+64)
 *   masked cross-attention            modeling_otter.py:290-333   (mask_mode EQ / GE with text_time)
 * q [B,Tq,H*64] (row stride q_stride), k/v [B,M,H*64] (row stride kv_stride; k and v usually point into one
 * to_kv output buffer), o like q.  sim = scale * q.k ; masked entries := -FLT_MAX ; softmax ; rows with
 * text_time == 0 are zeroed when mask_mode == EQ (only_attend_immediate_media) ; rows whose every key is masked
 * come out uniform (1/M) exactly as the reference's masked_fill(-finfo.max) + amax does.
 * media index of key j is j / n_per_media.  lse [B,H,Tq] fp32 is saved for the backward.
 * ------------------------------------------------------------------------------------------------------- */
typedef enum { OTTER_MASK_NONE = 0, OTTER_MASK_EQ = 1, OTTER_MASK_GE = 2 } otter_mask_mode;

int otter_attn_fwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, void* o,
                   int64_t o_stride, float* lse, const int32_t* text_time, int64_t B, int64_t H, int64_t Tq, int64_t M,
                   int64_t n_per_media, int mask_mode, float scale, int dtype, void* stream);

/* tuning / A-B hook: 0 = MFMA kernels for bf16 inputs (attn_mfma.hip), 1 = fp32 VALU kernels for every dtype */
int otter_attn_set_variant(int variant);

int64_t otter_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Tq, int64_t M);
/* dq like q; dk/dv like k/v (stride dkv_stride).  ws from otter_attn_bwd_workspace_bytes. */
int otter_attn_bwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, const void* o,
                   const void* d_o, int64_t o_stride, const float* lse, const int32_t* text_time, void* dq,
                   int64_t dq_stride, void* dk, void* dv, int64_t dkv_stride, void* ws, int64_t B, int64_t H,
                   int64_t Tq, int64_t M, int64_t n_per_media, int mask_mode, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Flash attention of the frozen decoder host, head_dim 128 -- and, round 3, head_dim 64 (Persimmon / Fuyu-8B,
 * /root/reference/src/otter_ai/models/fuyu/modeling_persimmon.py:310 flash_attn_func(causal=True)) --, bf16, MFMA:
 *   scaled_multihead_dot_product_attention   /root/reference/src/otter_ai/models/mpt/attention.py:22-84
 *   ALiBi bias (key-position form)           .../mpt/attention.py:447-464      bias[h,j] = slope[h] * (j - (Sk-1))
 *   key-padding mask                         .../mpt/modeling_mpt.py:135-144
 *   causal mask                              .../mpt/attention.py:64-72        key j visible to query i iff j <= i + Sk - Sq
 * out = softmax(scale * q k^T + bias + masks) v without materialising the scores; the bias is evaluated in fp32 in
 * the kernel.  q/k/v/o (and the gradients) are [B, S, H, 128] views given by element strides, so the three slices
 * of a fused Wqkv output and of its gradient buffer are addressed in place (no chunk/cat copies).
 * lse [B,H,Sq] fp32 (natural log) is written by the forward and read by the backward; delta is backward workspace of
 * 2*B*H*Sq floats (row dot products, then a log2-domain copy of lse).  A query row with no visible key yields 0 (lse = -inf) and zero gradients; the reference's
 * masked_fill(finfo.min) would yield the uniform average there -- such rows only exist for left-padded prompts,
 * which the host routes to its SDPA path.
 * head_dim 64: one workgroup takes two adjacent heads (H must be even); the second head of a pair lies head_stride elements after
 * the first inside the same token row (64 <= head_stride <= seq_stride - 64), so compact [B,S,H,64] tensors and the v slots of
 * Persimmon's per-head interleaved [B,S,H,3,64] projection buffer (head_stride 192) are both read / written in place.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct otter_flash_view {
    int64_t batch_stride, seq_stride, head_stride; /* in elements; multiples of 8 */
} otter_flash_view;

typedef struct otter_flash_desc {
    const void* q; const void* k; const void* v; otter_flash_view qv, kv, vv;
    void* o; otter_flash_view ov;
    float* lse;
    const float* alibi_slopes;   /* [H] fp32 or NULL */
    const uint8_t* key_valid;    /* [B, Sk] (0 = padded key) or NULL */
    int B, H, Sq, Sk, head_dim, causal;
    float scale;
    /* backward only */
    const void* dout; otter_flash_view dov;
    float* delta;
    void* dq; void* dk; void* dv; otter_flash_view dqv, dkv, dvv;
} otter_flash_desc;

int otter_flash_attn_fwd(const otter_flash_desc* d, void* stream);
/* tuning / A-B hook: 0 = default (LDS-DMA tiles, longest-first block order under the causal mask, delta inside the dQ kernel, one persistent
 * dK/dV workgroup per (batch, head) when those fill the chip evenly), 1 = register-staged tiles (v1), 2 = LDS-DMA tiles on the plain 3-D grid,
 * 3 / 5 = 2 / 0 with dK+dV at two workgroups per CU, 4 = 0 with the separate delta launch, 6 = 0 with forward version 3 (16 queries per
 * wave), 7 = 0 with one dK/dV workgroup per key block, 8 = 0 with the persistent dK/dV form on every launch that can take it
 * (head_dim 64: 1 is refused, 3 / 5 = 2 / 0) */
int otter_flash_set_variant(int variant);
int otter_flash_attn_bwd(const otter_flash_desc* d, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * RoPE, half-split (non-interleaved) layout, optional partial rotary.  Replaces
 * /root/reference/xformers_model/llama.py:158-166 (config C4) and flash_attn apply_rotary_emb at
 * fuyu/modeling_persimmon.py:303-304 (config C5).  x [B,S,H,d]; cos/sin fp32 [S, rot_dim]; in place allowed.
 * inverse != 0 applies the transpose rotation (= the backward).
 * ------------------------------------------------------------------------------------------------------- */
int otter_rope(const void* x, void* y, const float* cos_t, const float* sin_t, int64_t B, int64_t S, int64_t H,
               int64_t d, int64_t rot_dim, int inverse, int dtype, void* stream);

/* Same rotation, bf16, full rotary, 16-byte vectors, on STRIDED tokens: `tokens` = B*S tokens whose H rotated heads are
 * contiguous (head stride d) and which are x_token_stride / y_token_stride elements apart -- q and k inside the fused
 * [B,S,3,H,d] projection buffer are H' = 2H heads with token stride 3*H*d.  In place allowed (y == x, equal strides). */
int otter_rope_strided(const void* x, void* y, const float* cos_t, const float* sin_t, int64_t tokens, int64_t S, int64_t H,
                       int64_t d, int inverse, int64_t x_token_stride, int64_t y_token_stride, void* stream);

/* quick-GELU of the frozen CLIP tower's MLP (/root/reference/xformers_model/clip.py:84-95: x * sigmoid(1.702 x)); bf16 or
 * f32, n elements (multiple of 8), in place allowed. */
int otter_quick_gelu(const void* x, void* y, int64_t n, int dtype, void* stream);

/* Exact-erf GELU of the frozen MPT decoder's MLP (/root/reference/src/otter_ai/models/mpt/blocks.py:37-49: act = nn.GELU(approximate=
 * 'none')) and its backward dx = dy * (Phi(x) + x phi(x)); bf16 or f32, n elements (multiple of 8), y == x / dx == dy allowed. */
int otter_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream);
int otter_gelu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, void* stream);

/* SwiGLU of the LLaMA MLP (/root/reference/xformers_model/llama.py:216-223: down(act(gate(x)) * up(x)), act = SiLU) on the
 * [rows, 2*I] bf16 output of the concatenated gate|up projection: h[rows, I] = silu(g) * u;  backward writes
 * dgate_up[rows, 2*I] = (dh * u * silu'(g) | dh * silu(g)). */
int otter_swiglu_fwd(const void* gate_up, void* h, int64_t rows, int64_t I, void* stream);
int otter_swiglu_bwd(const void* gate_up, const void* dh, void* dgate_up, int64_t rows, int64_t I, void* stream);

/* x[i] += y[i] for a broadcast row block:  x [groups, rows, D] += emb [rows_e .. broadcast]; used for
 * frame_embs (modeling_otter.py:224-226).  x viewed as [outer, F, inner, D]; emb [F, D] fp32 master. */
int otter_add_frame_embs(void* x, int x_dtype, const float* emb, int64_t outer, int64_t F, int64_t inner, int64_t D,
                         void* stream);

/* dst[r] += src[map(r)] for r in [0, rows): the perceiver backward sums the two gradient paths of
 * norm_latents(latents) -- through to_q and through the latent rows of the [x ; latents] to_kv input
 * (modeling_otter.py:165-167) -- without materialising a gathered copy. */
int otter_add_rows(void* dst, const void* src, otter_rowmap src_map, int64_t rows, int64_t D, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Single-query attention over a KV cache: the cached decode step of generate() (/root/reference/src/otter_ai/models/otter/
 * modeling_otter.py:999-1042 -> mpt/attention.py:22-84 with past_key_value + ALiBi :447-464; LLaMA llama.py:169-213).
 * q, o [B, H, 128] (bf16); K and V are addressed by explicit batch / head / key / dim strides (elements), so the reference's
 * MPT cache layout k [B,H,d,S], v [B,H,S,d] and the [B,H,S,d] layout of the LLaMA host are both read in place.
 * alibi_slopes [H] fp32 or NULL (bias slope * (j - (Sk-1))); key_valid [B, Sk] (0 = padded key) or NULL.  Sk <= 16384.
 * ------------------------------------------------------------------------------------------------------- */
int otter_decode_attn(const void* q, int64_t q_batch_stride, int64_t q_head_stride, const void* k, int64_t k_batch_stride,
                      int64_t k_head_stride, int64_t k_key_stride, int64_t k_dim_stride, const void* v, int64_t v_batch_stride,
                      int64_t v_head_stride, int64_t v_key_stride, int64_t v_dim_stride, void* o, int64_t o_batch_stride,
                      int64_t o_head_stride, const float* alibi_slopes, const uint8_t* key_valid, int64_t B, int64_t H, int64_t Sk,
                      int64_t head_dim, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * OtterHD / Fuyu-8B path (config C5): row-wise kernels of the Persimmon decoder and the patch scatter.
 *   otter_qk_norm_rope_fwd  /root/reference/src/otter_ai/models/fuyu/modeling_persimmon.py:262-304: the per-head interleaved
 *       projection output qkv [tokens, H, 3, 64] (bf16) is read in place; q and k get LayerNorm over the 64-wide head
 *       (gamma / beta fp32 [64]) and the partial rotary embedding on their first `rot` dims (cos / sin fp32 [S, rot], position
 *       = token % S); q', k' (and v unless v_out is NULL: the head_dim-64 attention kernels read v in place from qkv) are written
 *       as bf16 [tokens, H, out_width]: out_width 64 = compact heads, 128 = columns 64..127 zero (head-dim padding for the
 *       128-wide flash kernels, round 2's layout).  stats [tokens, H, 2, 2] fp32 = (mean, rstd) of q and k, for the backward.
 *   otter_qk_norm_rope_bwd  dq / dk / dv [tokens, H, in_width] (columns 0..63 used) -> dqkv [tokens, H, 3, 64]; dv NULL: the v slots
 *       of dqkv are left alone (the attention backward wrote them in place); partial
 *       [otter_qk_norm_rope_bwd_blocks(tokens, H), 4, 64] fp32 = per-block sums of (dgamma_q, dbeta_q, dgamma_k, dbeta_k),
 *       to be summed over the first axis by the caller (deterministic).
 *   otter_sqrelu_fwd/_bwd   relu(x)^2 (:180-194), bf16, n % 8 == 0.
 *   otter_scatter_rows      fuyu/modeling_fuyu.py:44-77: out[b,s,:] = idx[b,s] < 0 ? word[b,s,:] : patch[b, idx[b,s], :].
 * ------------------------------------------------------------------------------------------------------- */
int otter_qk_norm_rope_fwd(const void* qkv, const float* gamma_q, const float* beta_q, const float* gamma_k, const float* beta_k,
                           const float* cos_t, const float* sin_t, void* q_out, void* k_out, void* v_out, float* stats, int64_t tokens,
                           int64_t S, int64_t H, int64_t rot, float eps, int64_t out_width, void* stream);
int64_t otter_qk_norm_rope_bwd_blocks(int64_t tokens, int64_t H);
int otter_qk_norm_rope_bwd(const void* dq, const void* dk, const void* dv, const void* qkv, const float* stats, const float* gamma_q,
                           const float* gamma_k, const float* cos_t, const float* sin_t, void* dqkv, float* partial, int64_t tokens,
                           int64_t S, int64_t H, int64_t rot, int64_t in_width, void* stream);
int otter_sqrelu_fwd(const void* x, void* y, int64_t n, void* stream);
int otter_sqrelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);
int otter_scatter_rows(const void* word, int word_dtype, const void* patch, int patch_dtype, const int64_t* idx, void* out, int64_t B,
                       int64_t S, int64_t P, int64_t D, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Token cross-entropy of the decoder host on bf16 logits: F.cross_entropy(logits.view(-1, V), labels) with
 * ignore_index = -100 and mean reduction     /root/reference/src/otter_ai/models/mpt/modeling_mpt.py:428-435
 * (the caller rolls the labels).  fwd: lse[r], nll[r] (0 for ignored rows) from one read of the logits; the mean
 * is sum(nll) / max(n_valid, 1).  bwd: dlogits (bf16) = (softmax - onehot) * (*dloss) / max(*n_valid, 1), zero rows for
 * ignored labels; dloss and n_valid are device scalars (no host synchronisation).  Row strides: multiples of 4 elements (16-byte
 * accesses when they are multiples of 8 and the bases 16-byte aligned, 8-byte accesses otherwise -- LLaMA's 32004-wide vocabulary).
 * ------------------------------------------------------------------------------------------------------- */
int otter_cross_entropy_fwd(const void* logits, int64_t ld, const int64_t* labels, float* lse, float* nll, int64_t rows, int64_t V,
                            void* stream);
int otter_cross_entropy_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* dloss,
                            const float* n_valid, void* dlogits, int64_t ldd, int64_t rows, int64_t V, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Optimizer step of the recipe: torch.nn.utils.clip_grad_norm_(params, max_norm) followed by
 * torch.optim.AdamW.step()          /root/reference/pipeline/train/instruction_following.py:246-251
 * as two sweeps: otter_grad_sumsq (+ otter_clip_coef -> {norm, coefficient} on the device) and otter_adamw_step, which
 * multiplies the gradients by *grad_scale on the fly (the scaled gradients are NOT written back), follows torch's
 * fused AdamW arithmetic (decoupled weight decay, lerp first moment, bias corrections passed in) and optionally
 * refreshes a bf16 copy of each parameter.  torch.optim.AdamW keeps one `step` per parameter and one lr per param
 * group: a table row may carry its own lr / bias corrections (bias_correction1 != 0), which then override the
 * launch-wide arguments for that tensor.  `tensors` is a DEVICE array; block i of the launch owns elements
 * [blk_chunk[i] * otter_adamw_chunk(), ...) of tensors[blk_tensor[i]].
 * ------------------------------------------------------------------------------------------------------- */
typedef struct otter_adamw_tensor {
    float* p; const float* g; float* m; float* v;
    uint16_t* shadow;      /* bf16 copy of p, or NULL */
    int64_t numel;
    float weight_decay;
    float lr;                      /* used when bias_correction1 != 0 (per-group learning rate) */
    float bias_correction1;        /* 1 - beta1^step of THIS tensor; 0 = use the launch-wide lr / bias corrections */
    float bias_correction2_sqrt;   /* sqrt(1 - beta2^step) of this tensor */
} otter_adamw_tensor;              /* 64 bytes */

int otter_adamw_chunk(void);
int otter_grad_sumsq(const otter_adamw_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk, int64_t nblocks,
                     float* partials, void* stream);
int otter_clip_coef(const float* partials, int64_t n, float max_norm, float* out2, void* stream);
int otter_adamw_step(const otter_adamw_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk, int64_t nblocks, float lr,
                     float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, const float* grad_scale,
                     void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Profiling hook used by bench.py for the `roofline` object: when enabled, every launch of the bf16 GEMM whose
 * (M,N,K) equals the armed shape is bracketed by hipEvents on the launch stream; otter_prof_collect waits for
 * them and returns count and total milliseconds.
 * ------------------------------------------------------------------------------------------------------- */
int otter_prof_arm_gemm(int64_t M, int64_t N, int64_t K, int max_events);
/* Diagnostics: with otter_gemm_set_debug bit 64 set, the one-wave-per-SIMD bf16 kernels (variants 18-20) record shader-clock
 * timestamps at their tile-phase boundaries for two blocks; this copies the first n of the 512 uint64 slots
 * ([block 2][wave 4][tile 8][mark 8]: 0 tile start, 1 prologue done, 2 K loop done, 3 tail done, 4 tile end; the large-grid kernel also
 * stamps the 100 MHz wall clock at tile start / end in marks 5 / 6: the shader clock of that very launch, and the launch's shape and operand
 * layout in mark 7: M << 42 | N << 21 | K, bits 63 / 62 = A / B K-major) to the host. */
int otter_gemm_read_timeline(unsigned long long* out, int n);
int otter_prof_disarm(void);
int otter_prof_collect(int* count, double* total_ms);
/* same, with the launches that had a K-major operand (otter_gemm) counted separately as well (they are included in count / total_ms) */
int otter_prof_collect_split(int* count, double* total_ms, int* count_kmajor, double* kmajor_ms);
/* Diagnostics (bench.py OTTER_BENCH_OCCUPY_CUS): n workgroups that each pin one CU's whole LDS and sleep until *flag
 * (device memory) becomes non-zero or max_ticks of the 100 MHz wall clock pass -- stands in for a communication kernel's hold on CUs. */
int otter_debug_occupy_cus(int n_workgroups, int* flag, unsigned long long max_ticks, void* stream);

/* Machine calibration for bench.py's `roofline` object (round 5; nothing in the reference to replace -- SURVEY.md 8d asks the builder to
 * "confirm with a microbenchmark and report measured peaks"): n_workgroups x 4 waves issue `iters` x 64 back-to-back
 * v_mfma_f32_16x16x32_bf16 (the product GEMM's instruction; 1 048 576 FLOP per wave and iteration) on the caller's operand bits
 * (`operands`: 1 MiB, read once).  out: uint64[2 * n_workgroups (+ slack: allocate 2 * n_workgroups + 256 * n_workgroups)]:
 * [2b] = shader-clock cycles (s_memtime), [2b + 1] = 100 MHz wall-clock ticks (s_memrealtime) of workgroup b's loop. */
int otter_probe_mfma(const void* operands, void* out, int iters, int n_workgroups, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTTER_HIP_H */
