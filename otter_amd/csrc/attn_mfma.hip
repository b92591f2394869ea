// attn_mfma.hip -- MFMA version of the fusion path's attention cores (head_dim 64, bf16 storage, fp32 math):
//   masked cross-attention  otter/modeling_otter.py:290-333   (mask EQ / GE on text_time, zeroed rows)
//   perceiver attention     otter/modeling_otter.py:168-179   (no mask)
// Same entry points and workspace as the fp32 VALU kernels of attn.hip (which remain the fp32 / parity-mode path and the
// fallback for n_per_media not a multiple of 32); same orientation and register tricks as flash.hip (S^T = K Q^T for the
// forward / dQ so the softmax statistics are lane-local, S = Q K^T for dK / dV, second products fed from the accumulator
// registers, their other operand read with ds_read_b64_tr_b16).
// Mask semantics, restated exactly:  a masked score is the FINITE fill value (the reference's -finfo.max): a row whose
// keys are all masked therefore comes out UNIFORM over all M keys (its lse carries the fill, so the backward recovers
// p = 1/M), rows with text_time == 0 are zeroed in EQ mode and get no gradient, masked entries never pass a gradient to
// q / k (masked_fill blocks it) but DO feed dV through their probability.
#include "common.h"

namespace otter_xattn {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

constexpr int HD = 64;
constexpr int LDK = HD + 8;    // 144-B rows: conflict-free ds_read_b128 of row fragments
constexpr int LDT = HD + 32;   // 192-B rows: conflict-free ds_read_b64_tr_b16 (4 rows -> 4 distinct 64-B quarters)
constexpr int QC = 64;         // query rows per dK/dV partial slab (must equal attn.hip's QC)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float FILL = -1.0e30f;  // log2-domain stand-in for masked_fill(-finfo.max): finite, underflows against any real score

struct XArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; int64_t q_stride, kv_stride;
    bf16_t* o; int64_t o_stride;
    float* lse; const int32_t* tt;
    int B, H, Tq, M, npm, mode;
    float scale;
    const bf16_t* d_o; float* delta;
    bf16_t* dq; int64_t dq_stride;
    float* part_k; float* part_v;
};

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* p) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 8 * LDT));
    const s16x8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ int tr_lane_off(int lane) {
    const int g = lane >> 4, i = lane & 15;
    return (4 * (g >> 1) + (i >> 2)) * LDT + 16 * (g & 1) + 4 * (i & 3);
}
__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& x, int r0) {
    bf16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[r0 + j];
    return r;
}
__device__ __forceinline__ bool allowed_at(int mode, int tt, int media_time) {
    return mode == OTTER_MASK_NONE || (mode == OTTER_MASK_EQ ? tt == media_time : tt >= media_time);
}
// stage `rows` x 64 bf16 rows (row r at src + (r0 + r) * stride, zero past `limit`) into one or two LDS images
template <int NT>
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ src, int64_t stride, int r0, int rows, int limit, bf16_t* a, int lda,
                                        bf16_t* b, int ldb, int tid) {
    for (int ch = tid; ch < rows * 8; ch += NT) {
        const int row = ch >> 3, c8 = ch & 7;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (r0 + row < limit) x = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + row) * stride + c8 * 8);
        *reinterpret_cast<uint4*>(a + row * lda + c8 * 8) = x;
        if (b) *reinterpret_cast<uint4*>(b + row * ldb + c8 * 8) = x;
    }
}
// C registers of a [d, row] product -> row-major [row][64] bf16 (lane = row)
__device__ __forceinline__ void store_dt(bf16_t* rowp, const f32x16_t (&acc)[2], float mul, int h2) {
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack2bf(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul);
            w.y = pack2bf(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(rowp + 32 * db + 8 * g + 4 * h2) = w;
        }
}

// ---------------------------------------------------------------- forward: grid (ceil(Tq/128), H, B), 4 waves x 32 queries
__global__ __launch_bounds__(256) void xattn_fwd_kernel(XArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * 128;
    const int qi = q0 + wave * 32 + ql, qc = qi < a.Tq ? qi : a.Tq - 1;
    const int tt = a.mode != OTTER_MASK_NONE ? a.tt[(int64_t)b * a.Tq + qc] : 0;
    const bool zero_row = a.mode == OTTER_MASK_EQ && tt == 0;
    const bf16_t* qp = a.q + ((int64_t)b * a.Tq + qc) * a.q_stride + hd * HD;
    bf16x8_t qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * c + 8 * h2);
    const bf16_t* kb = a.k + (int64_t)b * a.M * a.kv_stride + hd * HD;
    const bf16_t* vb = a.v + (int64_t)b * a.M * a.kv_stride + hd * HD;
    const float sc2 = a.scale * LOG2E;
    const int troff = tr_lane_off(lane);
    float m = -INFINITY, lsum = 0.f;
    f32x16_t o[2] = {zero16(), zero16()};
    const int nkt = (a.M + 63) >> 6;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 64;
        __syncthreads();
        stage64<256>(kb, a.kv_stride, k0, 64, a.M, Ks, LDK, nullptr, 0, tid);
        stage64<256>(vb, a.kv_stride, k0, 64, a.M, Vt, LDT, nullptr, 0, tid);
        __syncthreads();
        f32x16_t s[2];
        float mx = -INFINITY;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            s[kbk] = zero16();
            const bf16_t* rowp = Ks + (32 * kbk + ql) * LDK + 8 * h2;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(rowp + 16 * c), qf[c], s[kbk], 0, 0, 0);
            // one media per 32-key block (n_per_media % 32 == 0, checked by the launcher)
            const bool ok_blk = allowed_at(a.mode, tt, (k0 + 32 * kbk) / a.npm + 1);
            const int lim = a.M - 1 - k0 - 32 * kbk - 4 * h2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cidx = (r & 3) + 8 * (r >> 2);
                float x = ok_blk ? s[kbk][r] * sc2 : FILL;
                x = cidx <= lim ? x : -INFINITY;
                s[kbk][r] = x;
                mx = fmaxf(mx, x);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m, mx);
        const float muse = mnew == -INFINITY ? 0.f : mnew;
        const float alpha = __builtin_amdgcn_exp2f(m - muse);
        m = mnew;
        lsum *= alpha;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kbk][r] - muse);
                s[kbk][r] = p;
                lsum += p;
            }
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8_t pf = pack8(s[kbk], 8 * c);
                const bf16_t* tp = Vt + troff + (32 * kbk + 16 * c) * LDT;
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tp + 32 * db), pf, o[db], 0, 0, 0);
            }
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = (lsum > 0.f && !zero_row) ? 1.0f / lsum : 0.f;
    if (qi < a.Tq) {
        store_dt(a.o + ((int64_t)b * a.Tq + qi) * a.o_stride + hd * HD, o, inv, h2);
        if (a.lse && h2 == 0) a.lse[((int64_t)b * a.H + hd) * a.Tq + qi] = lsum > 0.f ? m * LN2 + logf(lsum) : INFINITY;
    }
}

// ---------------------------------------------------------------- delta[b,h,i] = sum_d dO . O  (8 lanes per row)
__global__ __launch_bounds__(256) void xattn_delta_kernel(XArgs a) {
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int c = threadIdx.x & 7;
    const int64_t nrows = (int64_t)a.B * a.H * a.Tq;
    float acc = 0.f;
    if (row < nrows) {
        const int qi = (int)(row % a.Tq);
        const int hd = (int)((row / a.Tq) % a.H);
        const int b = (int)(row / ((int64_t)a.Tq * a.H));
        float x[8], y[8];
        Vec8<bf16_t>::load(a.d_o + ((int64_t)b * a.Tq + qi) * a.o_stride + hd * HD + 8 * c, x);
        Vec8<bf16_t>::load(a.o + ((int64_t)b * a.Tq + qi) * a.o_stride + hd * HD + 8 * c, y);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += x[i] * y[i];
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row < nrows && c == 0) a.delta[row] = acc;
}

// ---------------------------------------------------------------- dQ: same decomposition as the forward
__global__ __launch_bounds__(256) void xattn_bwd_dq_kernel(XArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Kt[64 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * 128;
    const int qi = q0 + wave * 32 + ql, qc = qi < a.Tq ? qi : a.Tq - 1;
    const int tt = a.mode != OTTER_MASK_NONE ? a.tt[(int64_t)b * a.Tq + qc] : 0;
    const bool live = qi < a.Tq && !(a.mode == OTTER_MASK_EQ && tt == 0);
    const bf16_t* qp = a.q + ((int64_t)b * a.Tq + qc) * a.q_stride + hd * HD;
    const bf16_t* dop = a.d_o + ((int64_t)b * a.Tq + qc) * a.o_stride + hd * HD;
    bf16x8_t qf[4], dof[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qf[c] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * c + 8 * h2);
        dof[c] = *reinterpret_cast<const bf16x8_t*>(dop + 16 * c + 8 * h2);
    }
    const int64_t srow = ((int64_t)b * a.H + hd) * a.Tq + qc;
    const float lse2 = a.lse[srow] * LOG2E, dl = a.delta[srow];   // lse = +inf (empty row) -> p = 0
    const bf16_t* kb = a.k + (int64_t)b * a.M * a.kv_stride + hd * HD;
    const bf16_t* vb = a.v + (int64_t)b * a.M * a.kv_stride + hd * HD;
    const float sc2 = a.scale * LOG2E;
    const int troff = tr_lane_off(lane);
    f32x16_t dq[2] = {zero16(), zero16()};
    const int nkt = (a.M + 63) >> 6;
    for (int kt = 0; kt < nkt; ++kt) {
        const int k0 = kt * 64;
        __syncthreads();
        stage64<256>(kb, a.kv_stride, k0, 64, a.M, Ks, LDK, Kt, LDT, tid);
        stage64<256>(vb, a.kv_stride, k0, 64, a.M, Vs, LDK, nullptr, 0, tid);
        __syncthreads();
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            f32x16_t s = zero16(), dp = zero16();
            const bf16_t* krow = Ks + (32 * kbk + ql) * LDK + 8 * h2;
            const bf16_t* vrow = Vs + (32 * kbk + ql) * LDK + 8 * h2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(krow + 16 * c), qf[c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(vrow + 16 * c), dof[c], dp, 0, 0, 0);
            }
            const bool ok_blk = live && allowed_at(a.mode, tt, (k0 + 32 * kbk) / a.npm + 1);  // gradient only through allowed scores
            const int lim = a.M - 1 - k0 - 32 * kbk - 4 * h2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cidx = (r & 3) + 8 * (r >> 2);
                const float p = (ok_blk && cidx <= lim) ? __builtin_amdgcn_exp2f(s[r] * sc2 - lse2) : 0.f;
                s[r] = p * (dp[r] - dl);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8_t dsf = pack8(s, 8 * c);
                const bf16_t* tp = Kt + troff + (32 * kbk + 16 * c) * LDT;
#pragma unroll
                for (int db = 0; db < 2; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tp + 32 * db), dsf, dq[db], 0, 0, 0);
            }
        }
    }
    if (qi < a.Tq) store_dt(a.dq + ((int64_t)b * a.Tq + qi) * a.dq_stride + hd * HD, dq, a.scale, h2);
}

// ---------------------------------------------------------------- dK, dV partial slabs: grid (nkt * Z, H, B), 2 waves x 32 keys,
// the QC = 64 queries of slab z in two tiles of 32; fp32 partials [Z][B*M][H*64] summed by attn.hip's reduce kernel
__global__ __launch_bounds__(128) void xattn_bwd_dkv_kernel(XArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Qs[32 * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t dOs[32 * LDK];
    __shared__ __attribute__((aligned(16))) bf16_t Qt[32 * LDT];
    __shared__ __attribute__((aligned(16))) bf16_t dOt[32 * LDT];
    __shared__ __attribute__((aligned(16))) float lse_s[32];
    __shared__ __attribute__((aligned(16))) float dl_s[32];
    __shared__ __attribute__((aligned(16))) int tt_s[32];
    __shared__ __attribute__((aligned(16))) float pu_s[32];  // 1/M for a row whose keys are ALL masked (uniform softmax), else -1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    const int nkt = (a.M + 63) >> 6;
    const int kt = blockIdx.x % nkt, z = blockIdx.x / nkt;
    const int hd = blockIdx.y, b = blockIdx.z;
    const int kj = kt * 64 + wave * 32 + ql, kc = kj < a.M ? kj : a.M - 1;
    const bf16_t* kp = a.k + ((int64_t)b * a.M + kc) * a.kv_stride + hd * HD;
    const bf16_t* vp = a.v + ((int64_t)b * a.M + kc) * a.kv_stride + hd * HD;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        kf[c] = *reinterpret_cast<const bf16x8_t*>(kp + 16 * c + 8 * h2);
        vf[c] = *reinterpret_cast<const bf16x8_t*>(vp + 16 * c + 8 * h2);
    }
    const bool kin = kj < a.M;
    const int mt = kc / a.npm + 1;
    const float sc2 = a.scale * LOG2E;
    const bf16_t* qb = a.q + (int64_t)b * a.Tq * a.q_stride + hd * HD;
    const bf16_t* dob = a.d_o + (int64_t)b * a.Tq * a.o_stride + hd * HD;
    const float* lseb = a.lse + ((int64_t)b * a.H + hd) * a.Tq;
    const float* dlb = a.delta + ((int64_t)b * a.H + hd) * a.Tq;
    const int troff = tr_lane_off(lane);
    f32x16_t dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    for (int sub = 0; sub < QC / 32; ++sub) {
        const int i0 = z * QC + sub * 32;
        if (i0 >= a.Tq) break;
        __syncthreads();
        stage64<128>(qb, a.q_stride, i0, 32, a.Tq, Qs, LDK, Qt, LDT, tid);
        stage64<128>(dob, a.o_stride, i0, 32, a.Tq, dOs, LDK, dOt, LDT, tid);
        if (tid < 32) {
            const int i = i0 + tid;
            const bool in = i < a.Tq;
            const int t = (in && a.mode != OTTER_MASK_NONE) ? a.tt[(int64_t)b * a.Tq + i] : 0;
            const bool dead = !in || (a.mode == OTTER_MASK_EQ && t == 0);
            lse_s[tid] = dead ? INFINITY : lseb[i] * LOG2E;   // exp2(x - inf) = 0: dead rows contribute nothing
            dl_s[tid] = in ? dlb[i] : 0.f;
            tt_s[tid] = t;
            const int t_img = a.M / a.npm;
            const bool uni = !dead && (a.mode == OTTER_MASK_EQ ? (t < 1 || t > t_img) : (a.mode == OTTER_MASK_GE ? t < 1 : false));
            pu_s[tid] = uni ? 1.0f / (float)a.M : -1.0f;
        }
        __syncthreads();
        f32x16_t s = zero16(), dp = zero16();
        {
            const bf16_t* qrow = Qs + ql * LDK + 8 * h2;
            const bf16_t* drow = dOs + ql * LDK + 8 * h2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(qrow + 16 * c), kf[c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(drow + 16 * c), vf[c], dp, 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 8 * g + 4 * h2);
            const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 8 * g + 4 * h2);
            const int4 t4 = *reinterpret_cast<const int4*>(tt_s + 8 * g + 4 * h2);
            const float4 u4 = *reinterpret_cast<const float4*>(pu_s + 8 * g + 4 * h2);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w};
            const int tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const bool ok = allowed_at(a.mode, tv[e], mt);
                // allowed score: softmax probability from the saved lse; masked score: 0, or exactly 1/M in a uniform row
                float p = ok ? __builtin_amdgcn_exp2f(s[r] * sc2 - lv[e]) : (uv[e] > 0.f ? uv[e] : 0.f);
                p = kin ? p : 0.f;
                s[r] = p;                                   // feeds dV (uniform rows included)
                dp[r] = ok ? p * (dp[r] - dvv[e]) : 0.f;    // feeds dK: masked scores pass no gradient
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bf16x8_t pf = pack8(s, 8 * c), dsf = pack8(dp, 8 * c);
            const bf16_t* tq = Qt + troff + (16 * c) * LDT;
            const bf16_t* td = dOt + troff + (16 * c) * LDT;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(td + 32 * db), pf, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tq + 32 * db), dsf, dk[db], 0, 0, 0);
            }
        }
    }
    if (kin) {
        const int64_t rows = (int64_t)a.B * a.M;
        float* pk = a.part_k + (((int64_t)z * rows + (int64_t)b * a.M + kj) * a.H + hd) * HD;
        float* pv = a.part_v + (((int64_t)z * rows + (int64_t)b * a.M + kj) * a.H + hd) * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 32 * db + 8 * g + 4 * h2;
                *reinterpret_cast<float4*>(pk + d) = make_float4(dk[db][4 * g] * a.scale, dk[db][4 * g + 1] * a.scale, dk[db][4 * g + 2] * a.scale,
                                                                 dk[db][4 * g + 3] * a.scale);
                *reinterpret_cast<float4*>(pv + d) = make_float4(dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]);
            }
    }
}

XArgs make_args(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, const int32_t* tt, int64_t B, int64_t H,
                int64_t Tq, int64_t M, int64_t npm, int mode, float scale) {
    XArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.q_stride = q_stride; a.kv_stride = kv_stride;
    a.tt = tt; a.B = (int)B; a.H = (int)H; a.Tq = (int)Tq; a.M = (int)M; a.npm = (int)npm; a.mode = mode; a.scale = scale;
    return a;
}

}  // namespace

// bf16 + (no mask, or media boundaries on 32-key block boundaries) + 16-byte aligned rows
bool eligible(int dtype, int mask_mode, int64_t n_per_media, int64_t q_stride, int64_t kv_stride, const void* q, const void* k, const void* v) {
    if (dtype != OTTER_BF16) return false;
    if (mask_mode != OTTER_MASK_NONE && n_per_media % 32 != 0) return false;
    return ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0) && q_stride % 8 == 0 && kv_stride % 8 == 0;
}

int fwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, void* o, int64_t o_stride, float* lse,
        const int32_t* tt, int64_t B, int64_t H, int64_t Tq, int64_t M, int64_t npm, int mode, float scale, hipStream_t st) {
    XArgs a = make_args(q, q_stride, k, v, kv_stride, tt, B, H, Tq, M, mode == OTTER_MASK_NONE ? M : npm, mode, scale);
    a.o = (bf16_t*)o; a.o_stride = o_stride; a.lse = lse;
    hipLaunchKernelGGL(xattn_fwd_kernel, dim3((unsigned)cdiv64(Tq, 128), (unsigned)H, (unsigned)B), dim3(256), 0, st, a);
    OTTER_CHECK_LAUNCH("attn_fwd(mfma)");
    return OTTER_OK;
}

int bwd(const void* q, int64_t q_stride, const void* k, const void* v, int64_t kv_stride, const void* o, const void* d_o, int64_t o_stride,
        const float* lse, const int32_t* tt, void* dq, int64_t dq_stride, float* delta, float* part_k, float* part_v, int64_t B, int64_t H,
        int64_t Tq, int64_t M, int64_t npm, int mode, float scale, hipStream_t st) {
    XArgs a = make_args(q, q_stride, k, v, kv_stride, tt, B, H, Tq, M, mode == OTTER_MASK_NONE ? M : npm, mode, scale);
    a.o = (bf16_t*)const_cast<void*>(o); a.o_stride = o_stride; a.lse = const_cast<float*>(lse);
    a.d_o = (const bf16_t*)d_o; a.delta = delta; a.dq = (bf16_t*)dq; a.dq_stride = dq_stride; a.part_k = part_k; a.part_v = part_v;
    const int64_t nrows = B * H * Tq;
    hipLaunchKernelGGL(xattn_delta_kernel, dim3((unsigned)cdiv64(nrows, 32)), dim3(256), 0, st, a);
    OTTER_CHECK_LAUNCH("attn_delta(mfma)");
    hipLaunchKernelGGL(xattn_bwd_dq_kernel, dim3((unsigned)cdiv64(Tq, 128), (unsigned)H, (unsigned)B), dim3(256), 0, st, a);
    OTTER_CHECK_LAUNCH("attn_bwd_dq(mfma)");
    const int64_t Z = cdiv64(Tq, QC);
    hipLaunchKernelGGL(xattn_bwd_dkv_kernel, dim3((unsigned)(cdiv64(M, 64) * Z), (unsigned)H, (unsigned)B), dim3(128), 0, st, a);
    OTTER_CHECK_LAUNCH("attn_bwd_dkv(mfma)");
    return OTTER_OK;
}

}  // namespace otter_xattn
