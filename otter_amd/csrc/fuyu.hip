// fuyu.hip -- element-wise / row-wise kernels of the OtterHD (Fuyu-8B / Persimmon) path, config C5.
// All HBM-bound; 16-byte vectors; bf16 storage, fp32 arithmetic.
//   otter_qk_norm_rope_fwd/_bwd  q/k LayerNorm over head_dim 64 (fuyu/modeling_persimmon.py:285-287, flash-attn's fused_layer_norm),
//                                partial rotary on the first `rot` dims (:290-304), read IN PLACE from the per-head interleaved
//                                [tokens, H, 3, 64] projection buffer (_split_heads :262-275), written as compact [tokens, H, 64]
//                                heads for csrc/flash.hip's head-pair kernels (v is then read in place: no copy; round 3), or --
//                                round 2's layout, kept for odd head counts and A/B runs -- as [tokens, H, 128] with the upper 64
//                                columns zero (head-dim padding for the 128-wide kernels: exact).
//   otter_sqrelu_fwd/_bwd        relu(x)^2 of the MLP (:180-194, fused_mlp_func "sqrelu")
//   otter_scatter_rows           patch embeddings into the word-embedding sequence (fuyu/modeling_fuyu.py:44-77)
#include "common.h"

namespace {

constexpr int HD = 64, PD = 128;

// lane l8 of an 8-lane group owns elements 8*l8 .. 8*l8+7 of a 64-wide head vector
__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// rotary on the lane's 8 elements: partner lane holds the other half of each pair (half = rot/2 elements = half/8 lanes away)
__device__ __forceinline__ void rope8(float (&x)[8], const float* __restrict__ cs, const float* __restrict__ sn, int s, int rot, int l8,
                                      bool inverse) {
    const int half = rot >> 1, hl = half >> 3;          // lanes per half
    const bool in_rot = 8 * l8 < rot, first = l8 < hl;
    const int partner = (threadIdx.x & ~7) | (first ? l8 + hl : l8 - hl);
    float xp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xp[i] = __shfl(x[i], in_rot ? (partner & 63) : (threadIdx.x & 63), 64);
    if (!in_rot) return;
    const float* c = cs + (int64_t)s * rot + 8 * l8;
    const float* sp = sn + (int64_t)s * rot + 8 * l8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // forward: y1 = x1 c - x2 s, y2 = x2 c + x1 s ; inverse (transpose): x1 = y1 c + y2 s, x2 = y2 c - y1 s
        const float sg = (first != inverse) ? -1.f : 1.f;
        x[i] = x[i] * c[i] + sg * xp[i] * sp[i];
    }
}

// NSEL = 3: q, k and v vectors (v copied); 2: q and k only (the attention kernels read v in place from the projection buffer).
// PDW = 128: outputs zero-padded to 128 columns; 64: compact [tokens, H, 64]
template <int NSEL, int PDW>
__global__ __launch_bounds__(256) void qk_norm_rope_fwd_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ gq, const float* __restrict__ bq,
                                                             const float* __restrict__ gk, const float* __restrict__ bk,
                                                             const float* __restrict__ cs, const float* __restrict__ sn, bf16_t* __restrict__ qo,
                                                             bf16_t* __restrict__ ko, bf16_t* __restrict__ vo, float* __restrict__ stats, int64_t nvec,
                                                             int64_t S, int H, int rot, float eps) {
    // one 8-lane group per (token, head, q|k|v) vector
    const int64_t vec = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int l8 = threadIdx.x & 7;
    const bool live = vec < nvec;
    const int64_t vc = live ? vec : nvec - 1;
    const int sel = (int)(vc % NSEL);
    const int64_t th = vc / NSEL;                    // token * H + head
    const int64_t tok = th / H;
    float x[8];
    Vec8<bf16_t>::load(qkv + (th * 3 + sel) * HD + 8 * l8, x);
    bf16_t* out = (sel == 0 ? qo : sel == 1 ? ko : vo) + th * PDW;
    if (sel < 2) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sum += x[i];
        const float mean = group8_sum(sum) * (1.0f / HD);
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = x[i] - mean; var += d * d; }
        const float rstd = rsqrtf(group8_sum(var) * (1.0f / HD) + eps);
        const float* g = (sel == 0 ? gq : gk) + 8 * l8;
        const float* b = (sel == 0 ? bq : bk) + 8 * l8;
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (x[i] - mean) * rstd * g[i] + b[i];
        if (live && l8 == 0) { stats[(th * 2 + sel) * 2] = mean; stats[(th * 2 + sel) * 2 + 1] = rstd; }
        rope8(x, cs, sn, (int)(tok % S), rot, l8, false);
    } else {
        float dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        rope8(dummy, cs, sn, 0, 0, l8, false);      // keep the shuffles convergent (rot = 0: nothing rotates)
    }
    if (live) {
        Vec8<bf16_t>::store(out + 8 * l8, x);
        if (PDW == PD) *reinterpret_cast<uint4*>(out + HD + 8 * l8) = make_uint4(0, 0, 0, 0);
    }
}

template <int NSEL, int PDW>   // NSEL = 2: the attention backward has already written the v slots of dqkv
__global__ __launch_bounds__(256) void qk_norm_rope_bwd_kernel(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk, const bf16_t* __restrict__ dv,
                                                             const bf16_t* __restrict__ qkv, const float* __restrict__ stats,
                                                             const float* __restrict__ gq, const float* __restrict__ gk, const float* __restrict__ cs,
                                                             const float* __restrict__ sn, bf16_t* __restrict__ dqkv, float* __restrict__ partial,
                                                             int64_t nvec, int64_t S, int H, int rot, int64_t vec_per_block) {
    // block b owns vectors [b * vec_per_block, ...): 32 at a time; per-thread partial sums of dgamma / dbeta for q and k,
    // reduced over the 32 groups through LDS and written to partial[b][4][64] (dgq, dbq, dgk, dbk): deterministic
    __shared__ float red[4][32][64 + 1];
    const int l8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
    float agq[8] = {0}, abq[8] = {0}, agk[8] = {0}, abk[8] = {0};
    const int64_t v0 = (int64_t)blockIdx.x * vec_per_block;
    for (int64_t it = 0; it < vec_per_block; it += 32) {
        const int64_t vec = v0 + it + grp;
        const bool live = vec < nvec && it + grp < vec_per_block;
        const int64_t vc = live ? vec : (nvec - 1);
        const int sel = (int)(vc % NSEL);
        const int64_t th = vc / NSEL, tok = th / H;
        float dy[8];
        Vec8<bf16_t>::load((sel == 0 ? dq : sel == 1 ? dk : dv) + th * PDW + 8 * l8, dy);
        if (sel < 2) {
            rope8(dy, cs, sn, (int)(tok % S), rot, l8, true);
            float x[8];
            Vec8<bf16_t>::load(qkv + (th * 3 + sel) * HD + 8 * l8, x);
            const float mean = stats[(th * 2 + sel) * 2], rstd = stats[(th * 2 + sel) * 2 + 1];
            const float* g = (sel == 0 ? gq : gk) + 8 * l8;
            float s1 = 0.f, s2 = 0.f, gg[8], xh[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[i] = (x[i] - mean) * rstd;
                gg[i] = dy[i] * g[i];
                s1 += gg[i];
                s2 += gg[i] * xh[i];
            }
            s1 = group8_sum(s1) * (1.0f / HD);
            s2 = group8_sum(s2) * (1.0f / HD);
            if (live) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (sel == 0) { agq[i] += dy[i] * xh[i]; abq[i] += dy[i]; }
                    else { agk[i] += dy[i] * xh[i]; abk[i] += dy[i]; }
                    dy[i] = rstd * (gg[i] - s1 - xh[i] * s2);
                }
            }
        } else {
            float dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            rope8(dummy, cs, sn, 0, 0, l8, true);
            (void)group8_sum(0.f);
            (void)group8_sum(0.f);
        }
        if (live) Vec8<bf16_t>::store(dqkv + (th * 3 + sel) * HD + 8 * l8, dy);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[0][grp][8 * l8 + i] = agq[i]; red[1][grp][8 * l8 + i] = abq[i];
        red[2][grp][8 * l8 + i] = agk[i]; red[3][grp][8 * l8 + i] = abk[i];
    }
    __syncthreads();
    const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
    float t = 0.f;
#pragma unroll 8
    for (int g = 0; g < 32; ++g) t += red[which][g][col];
    partial[((int64_t)blockIdx.x * 4 + which) * 64 + col] = t;
}

__global__ void sqrelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t nch) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nch) return;
    float v[8];
    Vec8<bf16_t>::load(x + 8 * t, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float r = fmaxf(v[i], 0.f); v[i] = r * r; }
    Vec8<bf16_t>::store(y + 8 * t, v);
}
__global__ void sqrelu_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int64_t nch) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nch) return;
    float v[8], d[8];
    Vec8<bf16_t>::load(x + 8 * t, v);
    Vec8<bf16_t>::load(dy + 8 * t, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = 2.0f * fmaxf(v[i], 0.f) * d[i];
    Vec8<bf16_t>::store(dx + 8 * t, d);
}

// out[b, s, :] = idx[b, s] < 0 ? word[b, s, :] : patch[b, idx[b, s], :]     (one pass: the clone and the scatter)
// An index >= P is never dereferenced: the row is filled with NaN (the reference's advanced indexing raises for it; the host
// wrapper validates the range before the launch, this is the in-kernel backstop -- same policy as loss.hip's label >= V).
template <typename TW, typename TP>
__global__ void scatter_rows_kernel(const TW* __restrict__ word, const TP* __restrict__ patch, const int64_t* __restrict__ idx, TW* __restrict__ out,
                                    int64_t S, int64_t P, int64_t D, int64_t nch) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nch) return;
    const int64_t per_row = D / 8, row = t / per_row, c = t % per_row;
    const int64_t b = row / S, j = idx[row];
    float v[8];
    if (j < 0) Vec8<TW>::load(word + row * D + 8 * c, v);
    else if (j < P) Vec8<TP>::load(patch + (b * P + j) * D + 8 * c, v);
    else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_nanf("");
    }
    Vec8<TW>::store(out + row * D + 8 * c, v);
}

}  // namespace

extern "C" {

int otter_qk_norm_rope_fwd(const void* qkv, const float* gamma_q, const float* beta_q, const float* gamma_k, const float* beta_k,
                           const float* cos_t, const float* sin_t, void* q_out, void* k_out, void* v_out, float* stats, int64_t tokens, int64_t S,
                           int64_t H, int64_t rot, float eps, int64_t out_width, void* stream) {
    OTTER_REQUIRE(qkv && gamma_q && beta_q && gamma_k && beta_k && cos_t && sin_t && q_out && k_out && stats, "qk_norm_rope_fwd: null pointer");
    OTTER_REQUIRE(tokens > 0 && S > 0 && H > 0 && rot > 0 && rot <= HD && rot % 16 == 0, "qk_norm_rope_fwd: rot=%ld must be a multiple of 16 in (0, 64]", (long)rot);
    OTTER_REQUIRE(out_width == HD || out_width == PD, "qk_norm_rope_fwd: out_width %ld (64 = compact heads, 128 = zero-padded)", (long)out_width);
    const int64_t nvec = tokens * H * (v_out ? 3 : 2);
    const dim3 grid((unsigned)cdiv64(nvec, 32)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define OTTER_QKN_FWD(NSEL_, PDW_)                                                                                                                  \
    hipLaunchKernelGGL((qk_norm_rope_fwd_kernel<NSEL_, PDW_>), grid, block, 0, st, (const bf16_t*)qkv, gamma_q, beta_q, gamma_k, beta_k, cos_t, sin_t, \
                       (bf16_t*)q_out, (bf16_t*)k_out, (bf16_t*)v_out, stats, nvec, S, (int)H, (int)rot, eps)
    if (v_out && out_width == PD) OTTER_QKN_FWD(3, PD);
    else if (v_out) OTTER_QKN_FWD(3, HD);
    else if (out_width == PD) OTTER_QKN_FWD(2, PD);
    else OTTER_QKN_FWD(2, HD);
#undef OTTER_QKN_FWD
    OTTER_CHECK_LAUNCH("qk_norm_rope_fwd");
    return OTTER_OK;
}

int64_t otter_qk_norm_rope_bwd_blocks(int64_t tokens, int64_t H) {
    const int64_t nvec = tokens * H * 3;
    // 16 iterations of 32 vectors per block, at most 4096 blocks (C5 shape: 256 -> 180 us per launch against 64 iterations / 2048 blocks, which
    // left the 256 CUs with under three blocks each; 8 / 8192 and 4 / 16384 are slower again: the partial sums grow)
    int64_t nb = cdiv64(nvec, 32 * 16);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return nb;
}

int otter_qk_norm_rope_bwd(const void* dq, const void* dk, const void* dv, const void* qkv, const float* stats, const float* gamma_q,
                           const float* gamma_k, const float* cos_t, const float* sin_t, void* dqkv, float* partial, int64_t tokens, int64_t S,
                           int64_t H, int64_t rot, int64_t in_width, void* stream) {
    OTTER_REQUIRE(dq && dk && qkv && stats && gamma_q && gamma_k && cos_t && sin_t && dqkv && partial, "qk_norm_rope_bwd: null pointer");
    OTTER_REQUIRE(tokens > 0 && S > 0 && H > 0 && rot > 0 && rot <= HD && rot % 16 == 0, "qk_norm_rope_bwd: bad rot");
    OTTER_REQUIRE(in_width == HD || in_width == PD, "qk_norm_rope_bwd: in_width %ld (64 = compact heads, 128 = zero-padded)", (long)in_width);
    const int64_t nvec = tokens * H * (dv ? 3 : 2);
    const int64_t nb = otter_qk_norm_rope_bwd_blocks(tokens, H);
    const int64_t per = cdiv64(cdiv64(nvec, nb), 32) * 32;
    const dim3 grid((unsigned)nb), block(256);
    hipStream_t st = (hipStream_t)stream;
#define OTTER_QKN_BWD(NSEL_, PDW_)                                                                                                                  \
    hipLaunchKernelGGL((qk_norm_rope_bwd_kernel<NSEL_, PDW_>), grid, block, 0, st, (const bf16_t*)dq, (const bf16_t*)dk, (const bf16_t*)dv,           \
                       (const bf16_t*)qkv, stats, gamma_q, gamma_k, cos_t, sin_t, (bf16_t*)dqkv, partial, nvec, S, (int)H, (int)rot, per)
    if (dv && in_width == PD) OTTER_QKN_BWD(3, PD);
    else if (dv) OTTER_QKN_BWD(3, HD);
    else if (in_width == PD) OTTER_QKN_BWD(2, PD);
    else OTTER_QKN_BWD(2, HD);
#undef OTTER_QKN_BWD
    OTTER_CHECK_LAUNCH("qk_norm_rope_bwd");
    return OTTER_OK;
}

int otter_sqrelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    OTTER_REQUIRE(x && y && n > 0 && n % 8 == 0, "sqrelu_fwd: n %% 8");
    hipLaunchKernelGGL(sqrelu_fwd_kernel, dim3((unsigned)cdiv64(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n / 8);
    OTTER_CHECK_LAUNCH("sqrelu_fwd");
    return OTTER_OK;
}

int otter_sqrelu_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
    OTTER_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, "sqrelu_bwd: n %% 8");
    hipLaunchKernelGGL(sqrelu_bwd_kernel, dim3((unsigned)cdiv64(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy,
                       (bf16_t*)dx, n / 8);
    OTTER_CHECK_LAUNCH("sqrelu_bwd");
    return OTTER_OK;
}

int otter_scatter_rows(const void* word, int word_dtype, const void* patch, int patch_dtype, const int64_t* idx, void* out, int64_t B, int64_t S,
                       int64_t P, int64_t D, void* stream) {
    OTTER_REQUIRE(word && patch && idx && out && B > 0 && S > 0 && P > 0 && D % 8 == 0, "scatter_rows: bad args (D %% 8)");
    const int64_t n = B * S * (D / 8);
    dim3 grid((unsigned)cdiv64(n, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (word_dtype == OTTER_F32 && patch_dtype == OTTER_F32)
        hipLaunchKernelGGL((scatter_rows_kernel<float, float>), grid, block, 0, st, (const float*)word, (const float*)patch, idx, (float*)out, S, P, D, n);
    else if (word_dtype == OTTER_F32)
        hipLaunchKernelGGL((scatter_rows_kernel<float, bf16_t>), grid, block, 0, st, (const float*)word, (const bf16_t*)patch, idx, (float*)out, S, P, D, n);
    else if (patch_dtype == OTTER_F32)
        hipLaunchKernelGGL((scatter_rows_kernel<bf16_t, float>), grid, block, 0, st, (const bf16_t*)word, (const float*)patch, idx, (bf16_t*)out, S, P, D, n);
    else
        hipLaunchKernelGGL((scatter_rows_kernel<bf16_t, bf16_t>), grid, block, 0, st, (const bf16_t*)word, (const bf16_t*)patch, idx, (bf16_t*)out, S, P, D, n);
    OTTER_CHECK_LAUNCH("scatter_rows");
    return OTTER_OK;
}

}  // extern "C"
