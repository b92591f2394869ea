// decode.hip -- single-query attention over a KV cache for generate() (modeling_otter.py:999-1042 -> the decoder hosts'
// cached step: mpt/attention.py:22-84 with past_key_value, ALiBi :447-464; LLaMA llama.py:169-213).  HBM-bound: K and V of
// one (batch, head) are read exactly once.  One workgroup per (head, batch), head_dim 128, bf16 cache, fp32 arithmetic.
// K and V are addressed through explicit (key, dim) strides, so BOTH cache layouts are read in place:
//   MPT (the reference's): k [B,H,d,S] (key stride 1, dim stride S_alloc), v [B,H,S,d]
//   LLaMA:                 k, v [B,H,S,d] (key stride d, dim stride 1)
// Phase 1: lane = key: score = scale * q.k (+ ALiBi slope * (j - (Sk-1)), padding mask) for every key, scores parked in LDS,
// block max / sum.  Phase 2: thread = (dim, key slice): o[d] = sum_j p[j] v[j][d], slices combined through LDS.
#include "common.h"

namespace {

constexpr int HD = 128, NT = 256;

struct DecArgs {
    const bf16_t* q; int64_t q_bs, q_hs;                  // q [B, H, 128]: batch / head strides
    const bf16_t* k; int64_t k_bs, k_hs, k_ss, k_ds;      // K: batch, head, key, dim strides (elements)
    const bf16_t* v; int64_t v_bs, v_hs, v_ss, v_ds;
    bf16_t* o; int64_t o_bs, o_hs;
    const float* slopes; const uint8_t* kvalid;           // [H] or null; [B, Sk] or null
    int B, H, Sk; float scale;
};

__global__ __launch_bounds__(NT) void decode_attn_kernel(DecArgs a) {
    extern __shared__ float sc[];                         // Sk scores, then reused as the [2][128] slice buffer
    __shared__ float qs[HD];
    __shared__ float red[NT / 64];
    const int hd = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < HD) qs[tid] = bf2f(a.q[b * a.q_bs + hd * a.q_hs + tid]) * a.scale;
    __syncthreads();
    const bf16_t* kb = a.k + b * a.k_bs + hd * a.k_hs;
    const bf16_t* vb = a.v + b * a.v_bs + hd * a.v_hs;
    const float slope = a.slopes ? a.slopes[hd] : 0.f;
    const uint8_t* kv = a.kvalid ? a.kvalid + (int64_t)b * a.Sk : nullptr;
    float mx = -INFINITY;
    for (int j = tid; j < a.Sk; j += NT) {
        float s = 0.f;
        const bf16_t* kp = kb + (int64_t)j * a.k_ss;
        if (a.k_ds == 1) {
#pragma unroll 4
            for (int d = 0; d < HD; d += 8) {
                float x[8];
                Vec8<bf16_t>::load(kp + d, x);
#pragma unroll
                for (int i = 0; i < 8; ++i) s = fmaf(qs[d + i], x[i], s);
            }
        } else {
#pragma unroll 8
            for (int d = 0; d < HD; ++d) s = fmaf(qs[d], bf2f(kp[(int64_t)d * a.k_ds]), s);
        }
        s += slope * (float)(j - (a.Sk - 1));
        if (kv && kv[j] == 0) s = -INFINITY;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < a.Sk; j += NT) {
        const float p = mx == -INFINITY ? 0.f : __expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    // phase 2: thread (d = tid & 127, half = tid >> 7) walks the keys j = half, half + 2, ...
    const int d = tid & (HD - 1), half = tid >> 7;
    float acc = 0.f;
    for (int j = half; j < a.Sk; j += 2) acc = fmaf(sc[j], bf2f(vb[(int64_t)j * a.v_ss + (int64_t)d * a.v_ds]), acc);
    __syncthreads();                                      // everyone is done reading sc[] as scores
    float* part = sc;                                     // [2][128]
    part[half * HD + d] = acc;
    __syncthreads();
    if (tid < HD) {
        const float t = part[tid] + part[HD + tid];
        a.o[b * a.o_bs + hd * a.o_hs + tid] = f2bf(sum > 0.f ? t / sum : 0.f);
    }
}

}  // namespace

extern "C" {

int otter_decode_attn(const void* q, int64_t q_batch_stride, int64_t q_head_stride, const void* k, int64_t k_batch_stride, int64_t k_head_stride,
                      int64_t k_key_stride, int64_t k_dim_stride, const void* v, int64_t v_batch_stride, int64_t v_head_stride, int64_t v_key_stride,
                      int64_t v_dim_stride, void* o, int64_t o_batch_stride, int64_t o_head_stride, const float* alibi_slopes,
                      const uint8_t* key_valid, int64_t B, int64_t H, int64_t Sk, int64_t head_dim, float scale, void* stream) {
    OTTER_REQUIRE(q && k && v && o && B > 0 && H > 0 && Sk > 0, "decode_attn: bad args");
    OTTER_REQUIRE(head_dim == HD, "decode_attn: head_dim %ld (128 only)", (long)head_dim);
    OTTER_REQUIRE(Sk <= 16384, "decode_attn: Sk=%ld exceeds the LDS score buffer (16384)", (long)Sk);
    OTTER_REQUIRE(k_dim_stride != 1 || (k_key_stride % 8 == 0 && (((uintptr_t)k) & 15) == 0 && k_head_stride % 8 == 0 && k_batch_stride % 8 == 0),
                  "decode_attn: a dim-contiguous K needs 16-byte aligned rows");
    DecArgs a;
    a.q = (const bf16_t*)q; a.q_bs = q_batch_stride; a.q_hs = q_head_stride;
    a.k = (const bf16_t*)k; a.k_bs = k_batch_stride; a.k_hs = k_head_stride; a.k_ss = k_key_stride; a.k_ds = k_dim_stride;
    a.v = (const bf16_t*)v; a.v_bs = v_batch_stride; a.v_hs = v_head_stride; a.v_ss = v_key_stride; a.v_ds = v_dim_stride;
    a.o = (bf16_t*)o; a.o_bs = o_batch_stride; a.o_hs = o_head_stride;
    a.slopes = alibi_slopes; a.kvalid = key_valid; a.B = (int)B; a.H = (int)H; a.Sk = (int)Sk; a.scale = scale;
    const size_t smem = sizeof(float) * (size_t)(Sk > 2 * HD ? Sk : 2 * HD);
    if (smem > 32768) {
        // above the default dynamic-LDS allowance the launch needs the opt-in (64 KiB of scores + the kernel's 528 B of static LDS at
        // Sk = 16384, of the CU's 160 KiB).  Set once per process; thread-safe (a benign repeated call).
        static bool raised = false;
        if (!raised) {
            hipError_t e = hipFuncSetAttribute((const void*)decode_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
            OTTER_REQUIRE(e == hipSuccess, "decode_attn: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
            raised = true;
        }
    }
    hipLaunchKernelGGL(decode_attn_kernel, dim3((unsigned)H, (unsigned)B), dim3(NT), smem, (hipStream_t)stream, a);
    OTTER_CHECK_LAUNCH("decode_attn");
    return OTTER_OK;
}

}  // extern "C"
