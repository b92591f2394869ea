// attn.hip -- attention cores of the fusion path (head_dim 64) + the media text_time scan.
//
//   masked cross-attention  (otter/modeling_otter.py:290-333): many query rows (B*T text tokens), tiny key set
//                           (T_img*64 latents) -> split-Q: 256 query rows per block share one LDS image of a key chunk.
//   perceiver attention     (otter/modeling_otter.py:168-179): 64 latent queries, long key set (F*256+64 features)
//                           -> split-K: the 4 waves of a block walk disjoint key chunks and merge (m, l, acc) in LDS.
//
// Both are HBM/latency-bound (QK^T+PV is 0.07 of the 4.5 GF/sample the cross-attention module costs), so
// v1 keeps the arithmetic in fp32 on the VALU with one thread per (query row, head): q and the running output live
// in registers, K/V rows are read from LDS as wave-uniform (broadcast, conflict-free) 16-B reads, softmax is the
// online form in fp32.  Mask semantics are restated exactly: masked score := -FLT_MAX (so a fully masked row is
// UNIFORM over all keys), rows with text_time == 0 are zeroed in EQ mode, masked entries get zero gradient.
#include <math.h>

#include "common.h"

// MFMA implementation of the same entry points for bf16 (attn_mfma.hip)
namespace otter_xattn {
bool eligible(int dtype, int mask_mode, int64_t n_per_media, int64_t q_stride, int64_t kv_stride, const void* q, const void* k, const void* v);
int fwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, void* o, int64_t o_stride, float* lse,
        const int32_t* tt, int64_t B, int64_t H, int64_t Tq, int64_t M, int64_t npm, int mode, float scale, hipStream_t st);
int bwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, const void* o, const void* d_o, int64_t o_stride,
        const float* lse, const int32_t* tt, void* dq, int64_t dq_stride, float* delta, float* part_k, float* part_v, int64_t B, int64_t H,
        int64_t Tq, int64_t M, int64_t npm, int mode, float scale, hipStream_t st);
}  // namespace otter_xattn

namespace {

constexpr int HD = 64;  // head dim
int g_attn_variant = 0;  // 0 = MFMA kernels for bf16 when eligible, 1 = fp32 VALU kernels always

template <typename T>
__device__ __forceinline__ void load_row64(const T* p, float (&v)[HD]) {
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
        float t[8];
        Vec8<T>::load(p + c * 8, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c * 8 + i] = t[i];
    }
}
template <typename T>
__device__ __forceinline__ void store_row64(T* p, const float (&v)[HD]) {
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = v[c * 8 + i];
        Vec8<T>::store(p + c * 8, t);
    }
}

__device__ __forceinline__ float dot_lds(const float* __restrict__ row, const float (&q)[HD]) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
        const float4 k4 = *reinterpret_cast<const float4*>(row + c * 4);
        s = fmaf(q[c * 4 + 0], k4.x, s);
        s = fmaf(q[c * 4 + 1], k4.y, s);
        s = fmaf(q[c * 4 + 2], k4.z, s);
        s = fmaf(q[c * 4 + 3], k4.w, s);
    }
    return s;
}

struct RowMode {
    bool zero;     // EQ mode, text_time == 0  -> output 0, no gradients
    bool uniform;  // every key masked         -> softmax is uniform 1/M, no gradient to q/k
    int tt;
};
__device__ __forceinline__ RowMode row_mode(int mask_mode, const int32_t* text_time, int64_t idx, int t_img) {
    RowMode r;
    r.zero = false;
    r.uniform = false;
    r.tt = 0;
    if (mask_mode == OTTER_MASK_NONE) return r;
    r.tt = text_time[idx];
    if (mask_mode == OTTER_MASK_EQ) {
        r.zero = r.tt == 0;
        r.uniform = !r.zero && (r.tt < 1 || r.tt > t_img);
    } else {
        r.uniform = r.tt < 1;
    }
    return r;
}
__device__ __forceinline__ bool key_allowed(int mask_mode, int tt, int media_time) {
    return mask_mode == OTTER_MASK_NONE || (mask_mode == OTTER_MASK_EQ ? tt == media_time : tt >= media_time);
}

// stage `nkeys` K and V rows (head h) starting at key j0 into LDS as fp32 [key][64]; `nthr` threads cooperate
template <typename T>
__device__ __forceinline__ void stage_kv(const T* __restrict__ kb, const T* __restrict__ vb, int64_t kv_stride, int j0, int nkeys,
                                         int M, float* __restrict__ Ks, float* __restrict__ Vs, int t, int nthr) {
    // nkeys*8 vec8 chunks per tensor
    for (int c = t; c < nkeys * 8; c += nthr) {
        const int key = c >> 3, part = c & 7;
        float kv8[8], vv8[8];
        if (j0 + key < M) {
            Vec8<T>::load(kb + (int64_t)(j0 + key) * kv_stride + part * 8, kv8);
            Vec8<T>::load(vb + (int64_t)(j0 + key) * kv_stride + part * 8, vv8);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) kv8[i] = vv8[i] = 0.f;
        }
        Vec8<float>::store(Ks + key * HD + part * 8, kv8);
        Vec8<float>::store(Vs + key * HD + part * 8, vv8);
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------
constexpr int KC_Q = 64;  // keys per chunk, split-Q
constexpr int KC_K = 32;  // keys per chunk per wave, split-K

template <typename T, bool SPLITK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, int64_t q_stride, const T* __restrict__ k,
                                                       const T* __restrict__ v, int64_t kv_stride, T* __restrict__ o,
                                                       int64_t o_stride, float* __restrict__ lse,
                                                       const int32_t* __restrict__ text_time, int H, int Tq, int M,
                                                       int n_per_media, int mask_mode, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int i = SPLITK ? blockIdx.x * 64 + lane : blockIdx.x * 256 + tid;
    const bool active = i < Tq;
    const int t_img = M / n_per_media;
    const T* kb = k + (int64_t)b * M * kv_stride + h * HD;
    const T* vb = v + (int64_t)b * M * kv_stride + h * HD;

    float qr[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { qr[d] = 0.f; acc[d] = 0.f; }
    RowMode rm;
    rm.zero = true; rm.uniform = false; rm.tt = 0;
    if (active) {
        load_row64<T>(q + ((int64_t)b * Tq + i) * q_stride + h * HD, qr);
        rm = row_mode(mask_mode, text_time, (int64_t)b * Tq + i, t_img);
    }
    float m = -INFINITY, l = 0.f;

    const int KC = SPLITK ? KC_K : KC_Q;
    const int nchunks = (M + KC - 1) / KC;
    const int iters = SPLITK ? (nchunks + 3) / 4 : nchunks;
    float* Ks = SPLITK ? smem + wave * (2 * KC_K * HD) : smem;
    float* Vs = Ks + KC * HD;
    for (int it = 0; it < iters; ++it) {
        const int chunk = SPLITK ? it * 4 + wave : it;
        const int j0 = chunk * KC;
        __syncthreads();  // previous chunk fully consumed
        if (SPLITK) {
            if (chunk < nchunks) stage_kv<T>(kb, vb, kv_stride, j0, KC, M, Ks, Vs, lane, 64);
        } else {
            stage_kv<T>(kb, vb, kv_stride, j0, KC, M, Ks, Vs, tid, 256);
        }
        __syncthreads();
        if (active && !rm.zero && chunk < nchunks) {
            const int jn = (M - j0) < KC ? (M - j0) : KC;
            for (int jj = 0; jj < jn; ++jj) {
                float s;
                if (rm.uniform) {
                    s = 0.f;
                } else {
                    if (!key_allowed(mask_mode, rm.tt, (j0 + jj) / n_per_media + 1)) continue;
                    s = scale * dot_lds(Ks + jj * HD, qr);
                }
                if (s > m) {
                    const float alpha = expf(m - s);
                    l *= alpha;
#pragma unroll
                    for (int d = 0; d < HD; ++d) acc[d] *= alpha;
                    m = s;
                }
                const float p = expf(s - m);
                l += p;
                const float* vr = Vs + jj * HD;
#pragma unroll
                for (int c = 0; c < HD / 4; ++c) {
                    const float4 v4 = *reinterpret_cast<const float4*>(vr + c * 4);
                    acc[c * 4 + 0] = fmaf(p, v4.x, acc[c * 4 + 0]);
                    acc[c * 4 + 1] = fmaf(p, v4.y, acc[c * 4 + 1]);
                    acc[c * 4 + 2] = fmaf(p, v4.z, acc[c * 4 + 2]);
                    acc[c * 4 + 3] = fmaf(p, v4.w, acc[c * 4 + 3]);
                }
            }
        }
    }

    if (!SPLITK) {
        if (!active) return;
        float outv[HD];
        const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) outv[d] = acc[d] * inv;
        store_row64<T>(o + ((int64_t)b * Tq + i) * o_stride + h * HD, outv);
        if (lse) lse[((int64_t)b * H + h) * Tq + i] = l > 0.f ? m + logf(l) : INFINITY;
        return;
    }
    // split-K merge: LDS [4 waves][64 rows][65] acc + [4][64] m + [4][64] l
    __syncthreads();
    float* Am = smem;                 // 4*64*65
    float* Mm = smem + 4 * 64 * 65;   // 4*64
    float* Lm = Mm + 4 * 64;          // 4*64
#pragma unroll
    for (int d = 0; d < HD; ++d) Am[(wave * 64 + lane) * 65 + d] = acc[d];
    Mm[wave * 64 + lane] = m;
    Lm[wave * 64 + lane] = l;
    __syncthreads();
    float mstar = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mstar = fmaxf(mstar, Mm[w * 64 + lane]);
    float f[4], ltot = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float mw = Mm[w * 64 + lane];
        f[w] = (mw == -INFINITY) ? 0.f : expf(mw - mstar);
        ltot += Lm[w * 64 + lane] * f[w];
    }
    if (!active) return;
    const float inv = ltot > 0.f ? 1.0f / ltot : 0.f;
    // this wave writes dims [16*wave, 16*wave+16) of row `lane`
    float outv[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += Am[(w * 64 + lane) * 65 + wave * 16 + d] * f[w];
        outv[d] = t * inv;
    }
    T* op = o + ((int64_t)b * Tq + i) * o_stride + h * HD + wave * 16;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float t8[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) t8[d] = outv[c * 8 + d];
        Vec8<T>::store(op + c * 8, t8);
    }
    if (lse && wave == 0) lse[((int64_t)b * H + h) * Tq + i] = ltot > 0.f ? mstar + logf(ltot) : INFINITY;
}

// ------------------------------------------------------------------------------------------------------------
// backward, query side: dq and delta = sum(dO * O)
// ------------------------------------------------------------------------------------------------------------
template <typename T, bool SPLITK>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const T* __restrict__ q, int64_t q_stride, const T* __restrict__ k,
                                                          const T* __restrict__ v, int64_t kv_stride, const T* __restrict__ o,
                                                          const T* __restrict__ d_o, int64_t o_stride,
                                                          const float* __restrict__ lse, const int32_t* __restrict__ text_time,
                                                          T* __restrict__ dq, int64_t dq_stride, float* __restrict__ delta,
                                                          int H, int Tq, int M, int n_per_media, int mask_mode, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int i = SPLITK ? blockIdx.x * 64 + lane : blockIdx.x * 256 + tid;
    const bool active = i < Tq;
    const int t_img = M / n_per_media;
    const T* kb = k + (int64_t)b * M * kv_stride + h * HD;
    const T* vb = v + (int64_t)b * M * kv_stride + h * HD;
    float qr[HD], dor[HD], dqr[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { qr[d] = 0.f; dor[d] = 0.f; dqr[d] = 0.f; }
    RowMode rm;
    rm.zero = true; rm.uniform = false; rm.tt = 0;
    float L = INFINITY, dl = 0.f;
    if (active) {
        load_row64<T>(q + ((int64_t)b * Tq + i) * q_stride + h * HD, qr);
        load_row64<T>(d_o + ((int64_t)b * Tq + i) * o_stride + h * HD, dor);
        rm = row_mode(mask_mode, text_time, (int64_t)b * Tq + i, t_img);
        L = lse[((int64_t)b * H + h) * Tq + i];
        {
            float orow[HD];
            load_row64<T>(o + ((int64_t)b * Tq + i) * o_stride + h * HD, orow);
#pragma unroll
            for (int d = 0; d < HD; ++d) dl = fmaf(dor[d], orow[d], dl);
        }
        if (!SPLITK || wave == 0) delta[((int64_t)b * H + h) * Tq + i] = dl;
    }
    const bool work = active && !rm.zero && !rm.uniform;
    const int KC = SPLITK ? KC_K : KC_Q;
    const int nchunks = (M + KC - 1) / KC;
    const int iters = SPLITK ? (nchunks + 3) / 4 : nchunks;
    float* Ks = SPLITK ? smem + wave * (2 * KC_K * HD) : smem;
    float* Vs = Ks + KC * HD;
    for (int it = 0; it < iters; ++it) {
        const int chunk = SPLITK ? it * 4 + wave : it;
        const int j0 = chunk * KC;
        __syncthreads();
        if (SPLITK) {
            if (chunk < nchunks) stage_kv<T>(kb, vb, kv_stride, j0, KC, M, Ks, Vs, lane, 64);
        } else {
            stage_kv<T>(kb, vb, kv_stride, j0, KC, M, Ks, Vs, tid, 256);
        }
        __syncthreads();
        if (work && chunk < nchunks) {
            const int jn = (M - j0) < KC ? (M - j0) : KC;
            for (int jj = 0; jj < jn; ++jj) {
                if (!key_allowed(mask_mode, rm.tt, (j0 + jj) / n_per_media + 1)) continue;
                const float* kr = Ks + jj * HD;
                const float s = scale * dot_lds(kr, qr);
                const float p = expf(s - L);
                const float dp = dot_lds(Vs + jj * HD, dor);
                const float ds = p * (dp - dl) * scale;
#pragma unroll
                for (int c = 0; c < HD / 4; ++c) {
                    const float4 k4 = *reinterpret_cast<const float4*>(kr + c * 4);
                    dqr[c * 4 + 0] = fmaf(ds, k4.x, dqr[c * 4 + 0]);
                    dqr[c * 4 + 1] = fmaf(ds, k4.y, dqr[c * 4 + 1]);
                    dqr[c * 4 + 2] = fmaf(ds, k4.z, dqr[c * 4 + 2]);
                    dqr[c * 4 + 3] = fmaf(ds, k4.w, dqr[c * 4 + 3]);
                }
            }
        }
    }
    if (!SPLITK) {
        if (active) store_row64<T>(dq + ((int64_t)b * Tq + i) * dq_stride + h * HD, dqr);
        return;
    }
    __syncthreads();
    float* Am = smem;  // [4][64][65]
#pragma unroll
    for (int d = 0; d < HD; ++d) Am[(wave * 64 + lane) * 65 + d] = dqr[d];
    __syncthreads();
    if (!active) return;
    T* dp_ = dq + ((int64_t)b * Tq + i) * dq_stride + h * HD + wave * 16;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float t8[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += Am[(w * 64 + lane) * 65 + wave * 16 + c * 8 + d];
            t8[d] = t;
        }
        Vec8<T>::store(dp_ + c * 8, t8);
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward, key side: partial dK / dV per query chunk z (64 query rows), thread = (key, half of the head dim)
// ------------------------------------------------------------------------------------------------------------
constexpr int QC = 64;

template <typename T>
__global__ __launch_bounds__(128) void attn_bwd_dkv_kernel(const T* __restrict__ q, int64_t q_stride, const T* __restrict__ k,
                                                           const T* __restrict__ v, int64_t kv_stride, const T* __restrict__ d_o,
                                                           int64_t o_stride, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, const int32_t* __restrict__ text_time,
                                                           float* __restrict__ part_k, float* __restrict__ part_v, int B, int H,
                                                           int Tq, int M, int n_per_media, int mask_mode, float scale) {
    __shared__ __attribute__((aligned(16))) float Qs[QC * HD];
    __shared__ __attribute__((aligned(16))) float Ds[QC * HD];
    __shared__ float Ls[QC], Dl[QC];
    __shared__ int Tt[QC];
    const int tid = threadIdx.x;
    const int jl = tid & 63, half = tid >> 6;
    const int nkt = (M + 63) / 64;
    const int kt = blockIdx.x % nkt, z = blockIdx.x / nkt;
    const int h = blockIdx.y, b = blockIdx.z;
    const int j = kt * 64 + jl;
    const bool kact = j < M;
    const int t_img = M / n_per_media;
    float kr[HD], vr[HD], dk[32], dv[32];
#pragma unroll
    for (int d = 0; d < HD; ++d) { kr[d] = 0.f; vr[d] = 0.f; }
#pragma unroll
    for (int d = 0; d < 32; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    if (kact) {
        load_row64<T>(k + ((int64_t)b * M + j) * kv_stride + h * HD, kr);
        load_row64<T>(v + ((int64_t)b * M + j) * kv_stride + h * HD, vr);
    }
    const int media_time = j / n_per_media + 1;
    const int i0 = z * QC;
    // stage the query chunk
    for (int c = tid; c < QC * 8; c += 128) {
        const int r = c >> 3, part = c & 7;
        float a8[8], b8[8];
        if (i0 + r < Tq) {
            Vec8<T>::load(q + ((int64_t)b * Tq + i0 + r) * q_stride + h * HD + part * 8, a8);
            Vec8<T>::load(d_o + ((int64_t)b * Tq + i0 + r) * o_stride + h * HD + part * 8, b8);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) a8[t] = b8[t] = 0.f;
        }
        Vec8<float>::store(Qs + r * HD + part * 8, a8);
        Vec8<float>::store(Ds + r * HD + part * 8, b8);
    }
    if (tid < QC) {
        const int i = i0 + tid;
        if (i < Tq) {
            Ls[tid] = lse[((int64_t)b * H + h) * Tq + i];
            Dl[tid] = delta[((int64_t)b * H + h) * Tq + i];
            Tt[tid] = mask_mode == OTTER_MASK_NONE ? 0 : text_time[(int64_t)b * Tq + i];
        } else {
            Ls[tid] = INFINITY;  // p = exp(s - inf) = 0: padded rows contribute nothing
            Dl[tid] = 0.f;
            Tt[tid] = -1;
        }
    }
    __syncthreads();
    if (kact) {
        for (int r = 0; r < QC; ++r) {
            const float L = Ls[r];
            if (L == INFINITY) continue;  // zeroed / padded row (wave-uniform branch)
            const int tt = Tt[r];
            bool uniform = false;
            if (mask_mode == OTTER_MASK_EQ) uniform = (tt < 1 || tt > t_img);
            else if (mask_mode == OTTER_MASK_GE) uniform = tt < 1;
            const float* qrow = Qs + r * HD;
            const float* drow = Ds + r * HD;
            float p, ds = 0.f;
            if (uniform) {
                p = expf(-L);  // = 1/M
            } else {
                if (!key_allowed(mask_mode, tt, media_time)) continue;
                const float s = scale * dot_lds(qrow, kr);
                p = expf(s - L);
                const float dp = dot_lds(drow, vr);
                ds = p * (dp - Dl[r]) * scale;
            }
            const float* qh = qrow + half * 32;
            const float* dh = drow + half * 32;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 d4 = *reinterpret_cast<const float4*>(dh + c * 4);
                const float4 q4 = *reinterpret_cast<const float4*>(qh + c * 4);
                dv[c * 4 + 0] = fmaf(p, d4.x, dv[c * 4 + 0]);
                dv[c * 4 + 1] = fmaf(p, d4.y, dv[c * 4 + 1]);
                dv[c * 4 + 2] = fmaf(p, d4.z, dv[c * 4 + 2]);
                dv[c * 4 + 3] = fmaf(p, d4.w, dv[c * 4 + 3]);
                dk[c * 4 + 0] = fmaf(ds, q4.x, dk[c * 4 + 0]);
                dk[c * 4 + 1] = fmaf(ds, q4.y, dk[c * 4 + 1]);
                dk[c * 4 + 2] = fmaf(ds, q4.z, dk[c * 4 + 2]);
                dk[c * 4 + 3] = fmaf(ds, q4.w, dk[c * 4 + 3]);
            }
        }
        const int64_t off = (((int64_t)z * B + b) * M + j) * ((int64_t)H * HD) + h * HD + half * 32;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t8[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) t8[d] = dk[c * 8 + d];
            Vec8<float>::store(part_k + off + c * 8, t8);
#pragma unroll
            for (int d = 0; d < 8; ++d) t8[d] = dv[c * 8 + d];
            Vec8<float>::store(part_v + off + c * 8, t8);
        }
    }
}

// sum the Z partial slabs and write dk / dv (dtype T, row stride dkv_stride)
template <typename T>
__global__ void attn_bwd_dkv_reduce_kernel(const float* __restrict__ part_k, const float* __restrict__ part_v, T* __restrict__ dk,
                                           T* __restrict__ dv, int64_t dkv_stride, int64_t rows /* B*M */, int HDtot, int Z) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // vec8 chunk index
    const int per_row = HDtot / 8;
    if (c >= rows * per_row) return;
    const int64_t row = c / per_row;
    const int col = (int)(c % per_row) * 8;
    float sk[8], sv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sk[i] = sv[i] = 0.f;
    for (int z = 0; z < Z; ++z) {
        float a[8], b[8];
        Vec8<float>::load(part_k + ((int64_t)z * rows + row) * HDtot + col, a);
        Vec8<float>::load(part_v + ((int64_t)z * rows + row) * HDtot + col, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) { sk[i] += a[i]; sv[i] += b[i]; }
    }
    Vec8<T>::store(dk + row * dkv_stride + col, sk);
    Vec8<T>::store(dv + row * dkv_stride + col, sv);
}

// ------------------------------------------------------------------------------------------------------------
// text_time: inclusive scan of the <image> indicator, one wave per batch row
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void text_time_kernel(const uint8_t* __restrict__ ml, int32_t* __restrict__ tt, int T,
                                                       int attend_previous) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const uint8_t* row = ml + (int64_t)b * T;
    int32_t* out = tt + (int64_t)b * T;
    int carry = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        int x = (t < T && row[t]) ? 1 : 0;
        // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (t < T) out[t] = carry + x;
        carry += __shfl(x, 63, 64);
    }
    if (!attend_previous) {
        // text_time[~media_locations] += 1 ; text_time[text_time > count] = 0   (modeling_otter.py:301-311)
        const int count = carry;
        for (int t = lane; t < T; t += 64) {
            int vtt = out[t];
            if (!row[t]) vtt += 1;
            if (vtt > count) vtt = 0;
            out[t] = vtt;
        }
    }
}

template <typename T>
int launch_fwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, void* o, int64_t o_stride,
               float* lse, const int32_t* tt, int64_t B, int64_t H, int64_t Tq, int64_t M, int64_t n_per_media, int mask_mode,
               float scale, hipStream_t st) {
    if (g_attn_variant == 0 && otter_xattn::eligible(sizeof(T) == 2 ? OTTER_BF16 : OTTER_F32, mask_mode, n_per_media, q_stride, kv_stride, q, k, v) &&
        o_stride % 8 == 0 && (((uintptr_t)o & 15) == 0))
        return otter_xattn::fwd(q, q_stride, k, v, kv_stride, o, o_stride, lse, tt, B, H, Tq, M, n_per_media, mask_mode, scale, st);
    const bool splitk = (Tq <= 64 && M > 256) || (B * H * cdiv64(Tq, 256) < 64 && M >= 256);
    if (splitk) {
        const int smem = (4 * 64 * 65 + 8 * 64) * 4;  // merge image (>= 4 * 2*KC_K*HD staging floats)
        static bool once = false;
        if (!once) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            once = true;
        }
        dim3 grid((unsigned)cdiv64(Tq, 64), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL((attn_fwd_kernel<T, true>), grid, dim3(256), smem, st, (const T*)q, q_stride, (const T*)k, (const T*)v,
                           kv_stride, (T*)o, o_stride, lse, tt, (int)H, (int)Tq, (int)M, (int)n_per_media, mask_mode, scale);
    } else {
        const int smem = 2 * KC_Q * HD * 4;
        dim3 grid((unsigned)cdiv64(Tq, 256), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL((attn_fwd_kernel<T, false>), grid, dim3(256), smem, st, (const T*)q, q_stride, (const T*)k,
                           (const T*)v, kv_stride, (T*)o, o_stride, lse, tt, (int)H, (int)Tq, (int)M, (int)n_per_media, mask_mode,
                           scale);
    }
    OTTER_CHECK_LAUNCH("attn_fwd");
    return OTTER_OK;
}

template <typename T>
int launch_bwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, const void* o, const void* d_o,
               int64_t o_stride, const float* lse, const int32_t* tt, void* dq, int64_t dq_stride, void* dk, void* dv,
               int64_t dkv_stride, void* ws, int64_t B, int64_t H, int64_t Tq, int64_t M, int64_t n_per_media, int mask_mode,
               float scale, hipStream_t st) {
    float* delta = (float*)ws;
    const int64_t Z = cdiv64(Tq, QC);
    const int64_t slab = Z * B * M * H * HD;
    float* part_k = delta + ((B * H * Tq + 63) / 64) * 64;
    float* part_v = part_k + slab;
    if (g_attn_variant == 0 && otter_xattn::eligible(sizeof(T) == 2 ? OTTER_BF16 : OTTER_F32, mask_mode, n_per_media, q_stride, kv_stride, q, k, v) &&
        o_stride % 8 == 0 && dq_stride % 8 == 0 && ((((uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq) & 15) == 0)) {
        int rc = otter_xattn::bwd(q, q_stride, k, v, kv_stride, o, d_o, o_stride, lse, tt, dq, dq_stride, delta, part_k, part_v, B, H, Tq, M,
                                  n_per_media, mask_mode, scale, st);
        if (rc) return rc;
        const int64_t chunks = B * M * (H * HD / 8);
        hipLaunchKernelGGL((attn_bwd_dkv_reduce_kernel<T>), dim3((unsigned)cdiv64(chunks, 256)), dim3(256), 0, st, part_k, part_v,
                           (T*)dk, (T*)dv, dkv_stride, B * M, (int)(H * HD), (int)Z);
        OTTER_CHECK_LAUNCH("attn_bwd_dkv_reduce");
        return OTTER_OK;
    }
    const bool splitk = (Tq <= 64 && M > 256) || (B * H * cdiv64(Tq, 256) < 64 && M >= 256);
    if (splitk) {
        const int smem = (4 * 64 * 65 + 8 * 64) * 4;
        static bool once = false;
        if (!once) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            once = true;
        }
        dim3 grid((unsigned)cdiv64(Tq, 64), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, true>), grid, dim3(256), smem, st, (const T*)q, q_stride, (const T*)k,
                           (const T*)v, kv_stride, (const T*)o, (const T*)d_o, o_stride, lse, tt, (T*)dq, dq_stride, delta, (int)H,
                           (int)Tq, (int)M, (int)n_per_media, mask_mode, scale);
    } else {
        const int smem = 2 * KC_Q * HD * 4;
        dim3 grid((unsigned)cdiv64(Tq, 256), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, false>), grid, dim3(256), smem, st, (const T*)q, q_stride, (const T*)k,
                           (const T*)v, kv_stride, (const T*)o, (const T*)d_o, o_stride, lse, tt, (T*)dq, dq_stride, delta, (int)H,
                           (int)Tq, (int)M, (int)n_per_media, mask_mode, scale);
    }
    OTTER_CHECK_LAUNCH("attn_bwd_dq");
    {
        dim3 grid((unsigned)(cdiv64(M, 64) * Z), (unsigned)H, (unsigned)B);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T>), grid, dim3(128), 0, st, (const T*)q, q_stride, (const T*)k, (const T*)v,
                           kv_stride, (const T*)d_o, o_stride, lse, delta, tt, part_k, part_v, (int)B, (int)H, (int)Tq, (int)M,
                           (int)n_per_media, mask_mode, scale);
        OTTER_CHECK_LAUNCH("attn_bwd_dkv");
        const int64_t chunks = B * M * (H * HD / 8);
        hipLaunchKernelGGL((attn_bwd_dkv_reduce_kernel<T>), dim3((unsigned)cdiv64(chunks, 256)), dim3(256), 0, st, part_k, part_v,
                           (T*)dk, (T*)dv, dkv_stride, B * M, (int)(H * HD), (int)Z);
        OTTER_CHECK_LAUNCH("attn_bwd_dkv_reduce");
    }
    return OTTER_OK;
}

int check_common(const void* q, const void* k, const void* v, int64_t q_stride, int64_t kv_stride, int64_t B, int64_t H,
                 int64_t Tq, int64_t M, int64_t n_per_media, int mask_mode, const int32_t* tt) {
    OTTER_REQUIRE(q && k && v, "attn: null pointer");
    OTTER_REQUIRE(B > 0 && H > 0 && Tq > 0 && M > 0, "attn: empty shape");
    OTTER_REQUIRE(q_stride % 8 == 0 && kv_stride % 8 == 0, "attn: strides must be multiples of 8 elements");
    OTTER_REQUIRE(n_per_media > 0 && M % n_per_media == 0, "attn: M=%ld not a multiple of n_per_media=%ld", (long)M,
                  (long)n_per_media);
    OTTER_REQUIRE(mask_mode == OTTER_MASK_NONE || tt, "attn: text_time required for mask_mode %d", mask_mode);
    OTTER_REQUIRE(B <= 65535 && H <= 65535, "attn: B/H exceed grid limits");
    return OTTER_OK;
}

}  // namespace

extern "C" {

int otter_attn_set_variant(int v) {
    OTTER_REQUIRE(v == 0 || v == 1, "attn variant %d (0 = MFMA kernels for bf16, 1 = fp32 VALU kernels)", v);
    g_attn_variant = v;
    return OTTER_OK;
}

int otter_text_time(const uint8_t* media_locations, int32_t* text_time, int64_t B, int64_t T, int attend_previous, void* stream) {
    OTTER_REQUIRE(media_locations && text_time && B > 0 && T > 0, "text_time: bad args");
    hipLaunchKernelGGL(text_time_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, media_locations, text_time, (int)T,
                       attend_previous);
    OTTER_CHECK_LAUNCH("text_time");
    return OTTER_OK;
}

int otter_attn_fwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, void* o, int64_t o_stride,
                   float* lse, const int32_t* text_time, int64_t B, int64_t H, int64_t Tq, int64_t M, int64_t n_per_media,
                   int mask_mode, float scale, int dtype, void* stream) {
    int rc = check_common(q, k, v, q_stride, kv_stride, B, H, Tq, M, n_per_media, mask_mode, text_time);
    if (rc) return rc;
    OTTER_REQUIRE(o && o_stride % 8 == 0, "attn_fwd: bad output");
    if (dtype == OTTER_BF16)
        return launch_fwd<bf16_t>(q, q_stride, k, v, kv_stride, o, o_stride, lse, text_time, B, H, Tq, M, n_per_media, mask_mode,
                                  scale, (hipStream_t)stream);
    return launch_fwd<float>(q, q_stride, k, v, kv_stride, o, o_stride, lse, text_time, B, H, Tq, M, n_per_media, mask_mode, scale,
                             (hipStream_t)stream);
}

int64_t otter_attn_bwd_workspace_bytes(int64_t B, int64_t H, int64_t Tq, int64_t M) {
    const int64_t Z = cdiv64(Tq, QC);
    return (((B * H * Tq + 63) / 64) * 64 + 2 * Z * B * M * H * HD) * 4;
}

int otter_attn_bwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, const void* o, const void* d_o,
                   int64_t o_stride, const float* lse, const int32_t* text_time, void* dq, int64_t dq_stride, void* dk, void* dv,
                   int64_t dkv_stride, void* ws, int64_t B, int64_t H, int64_t Tq, int64_t M, int64_t n_per_media, int mask_mode,
                   float scale, int dtype, void* stream) {
    int rc = check_common(q, k, v, q_stride, kv_stride, B, H, Tq, M, n_per_media, mask_mode, text_time);
    if (rc) return rc;
    OTTER_REQUIRE(o && d_o && lse && dq && dk && dv, "attn_bwd: null pointer");
    OTTER_REQUIRE(o_stride % 8 == 0 && dq_stride % 8 == 0 && dkv_stride % 8 == 0, "attn_bwd: strides must be multiples of 8");
    if (!ws) OTTER_FAIL(OTTER_ERR_WORKSPACE, "attn_bwd: workspace required");
    if (dtype == OTTER_BF16)
        return launch_bwd<bf16_t>(q, q_stride, k, v, kv_stride, o, d_o, o_stride, lse, text_time, dq, dq_stride, dk, dv, dkv_stride,
                                  ws, B, H, Tq, M, n_per_media, mask_mode, scale, (hipStream_t)stream);
    return launch_bwd<float>(q, q_stride, k, v, kv_stride, o, d_o, o_stride, lse, text_time, dq, dq_stride, dk, dv, dkv_stride, ws, B,
                             H, Tq, M, n_per_media, mask_mode, scale, (hipStream_t)stream);
}

}  // extern "C"
