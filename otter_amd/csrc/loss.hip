// loss.hip -- token cross-entropy of the decoder host on bf16 logits (mpt/modeling_mpt.py:428-435: labels rolled by -1,
// F.cross_entropy(logits.view(-1, V), labels) with ignore_index -100, mean over the valid rows).
// torch runs this as: bf16 -> fp32 copy of the [4096, 50432] logits, log_softmax, nll, and three more fp32 passes in the
// backward.  Here: one read of the bf16 logits for (lse, nll) and, in the backward, one read + one bf16 write for
// dlogits = (softmax - onehot) * dloss / n_valid -- rows with label -100 are skipped (zero gradient row); a label >= V
// gives a NaN loss (never an out-of-bounds read).
// fp32 arithmetic on the bf16 values, i.e. what `logits.float()` feeds torch.
#include "common.h"

namespace {

constexpr int NT = 256;

// VEC = 8: 16-byte accesses (row stride % 8 == 0: MPT's 50432-wide vocabulary); VEC = 4: 8-byte accesses for row strides that are only a
// multiple of 4 elements (LLaMA's 32004: every other row of the contiguous [rows, V] logits is 8- but not 16-byte aligned)
template <int VEC> struct VecN;
template <> struct VecN<8> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) { Vec8<bf16_t>::load(p, v); }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) { Vec8<bf16_t>::store(p, v); }
    static __device__ __forceinline__ void zero(bf16_t* p) { *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0); }
};
template <> struct VecN<4> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
        const uint2 r = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        uint2 r;
        r.x = pack2bf(v[0], v[1]);
        r.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = r;
    }
    static __device__ __forceinline__ void zero(bf16_t* p) { *reinterpret_cast<uint2*>(p) = make_uint2(0, 0); }
};

__device__ __forceinline__ void block_combine(float& m, float& s, float* red /* [2 * NT/64] */) {
    // wave-level (max, sum-of-exp) merge, then across the 4 waves through LDS
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
        m = mn;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * w] = m; red[2 * w + 1] = s; }
    __syncthreads();
    float M = -INFINITY, S = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) {
        const float m2 = red[2 * i], s2 = red[2 * i + 1];
        const float mn = fmaxf(M, m2);
        S = (M == -INFINITY ? 0.f : S * __expf(M - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
        M = mn;
    }
    m = M;
    s = S;
}

template <int VEC>
__global__ __launch_bounds__(NT) void ce_fwd_kernel(const bf16_t* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                    float* __restrict__ lse, float* __restrict__ nll, int V) {
    __shared__ float red[2 * NT / 64];
    const int64_t row = blockIdx.x;
    const int64_t lab = labels[row];
    if (lab < 0) {  // ignore_index (any negative label): no loss, no gradient
        if (threadIdx.x == 0) { lse[row] = 0.f; nll[row] = 0.f; }
        return;
    }
    if (lab >= V) {  // out-of-range label (tokenizer / vocabulary-padding mismatch): F.cross_entropy device-asserts here; we
                     // never read x[lab] and poison the loss instead (NaN), the backward zero-fills the row
        if (threadIdx.x == 0) { lse[row] = 0.f; nll[row] = __builtin_nanf(""); }
        return;
    }
    const bf16_t* x = logits + row * ld;
    float m = -INFINITY, s = 0.f;
    const int nch = V / VEC;
    for (int c = threadIdx.x; c < nch; c += NT) {
        float v[VEC];
        VecN<VEC>::load(x + VEC * c, v);
        float mx = v[0];
#pragma unroll
        for (int i = 1; i < VEC; ++i) mx = fmaxf(mx, v[i]);
        const float mn = fmaxf(m, mx);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc += __expf(v[i] - mn);
        s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + acc;
        m = mn;
    }
    for (int i = nch * VEC + threadIdx.x; i < V; i += NT) {  // tail when V % VEC != 0
        const float v = bf2f(x[i]);
        const float mn = fmaxf(m, v);
        s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + __expf(v - mn);
        m = mn;
    }
    block_combine(m, s, red);
    if (threadIdx.x == 0) {
        const float l = m + logf(s);
        lse[row] = l;
        nll[row] = l - bf2f(x[lab]);
    }
}

// dlogits[row, :] = (exp(x - lse) - onehot(label)) * (*dloss) / max(*n_valid, 1)   (bf16), zero rows for ignored labels
template <int VEC>
__global__ __launch_bounds__(NT) void ce_bwd_kernel(const bf16_t* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                    const float* __restrict__ lse, const float* __restrict__ dloss,
                                                    const float* __restrict__ n_valid, bf16_t* __restrict__ dlogits, int64_t ldd, int V) {
    const int64_t row = blockIdx.x;
    const int64_t lab = labels[row];
    const bf16_t* x = logits + row * ld;
    bf16_t* d = dlogits + row * ldd;
    const int nch = V / VEC;
    if (lab < 0 || lab >= V) {
        for (int c = threadIdx.x; c < nch; c += NT) VecN<VEC>::zero(d + VEC * c);
        for (int i = nch * VEC + threadIdx.x; i < V; i += NT) d[i] = 0;
        return;
    }
    const float nv = *n_valid;
    const float sc = *dloss / (nv > 1.f ? nv : 1.f);
    const float l = lse[row];
    for (int c = threadIdx.x; c < nch; c += NT) {
        float v[VEC];
        VecN<VEC>::load(x + VEC * c, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float p = __expf(v[i] - l);
            if ((int64_t)(VEC * c + i) == lab) p -= 1.0f;
            v[i] = p * sc;
        }
        VecN<VEC>::store(d + VEC * c, v);
    }
    for (int i = nch * VEC + threadIdx.x; i < V; i += NT) {
        float p = __expf(bf2f(x[i]) - l);
        if ((int64_t)i == lab) p -= 1.0f;
        d[i] = f2bf(p * sc);
    }
}

}  // namespace

extern "C" {

int otter_cross_entropy_fwd(const void* logits, int64_t ld, const int64_t* labels, float* lse, float* nll, int64_t rows, int64_t V,
                            void* stream) {
    OTTER_REQUIRE(logits && labels && lse && nll && rows > 0 && V > 0, "cross_entropy_fwd: bad args");
    OTTER_REQUIRE(ld % 4 == 0 && (((uintptr_t)logits) & 7) == 0, "cross_entropy_fwd: row stride must be a multiple of 4, base 8-byte aligned");
    if (ld % 8 == 0 && (((uintptr_t)logits) & 15) == 0)
        hipLaunchKernelGGL(ce_fwd_kernel<8>, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, labels, lse, nll, (int)V);
    else
        hipLaunchKernelGGL(ce_fwd_kernel<4>, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, labels, lse, nll, (int)V);
    OTTER_CHECK_LAUNCH("cross_entropy_fwd");
    return OTTER_OK;
}

int otter_cross_entropy_bwd(const void* logits, int64_t ld, const int64_t* labels, const float* lse, const float* dloss,
                            const float* n_valid, void* dlogits, int64_t ldd, int64_t rows, int64_t V, void* stream) {
    OTTER_REQUIRE(logits && labels && lse && dloss && n_valid && dlogits && rows > 0 && V > 0, "cross_entropy_bwd: bad args");
    OTTER_REQUIRE(ld % 4 == 0 && ldd % 4 == 0 && ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 7) == 0,
                  "cross_entropy_bwd: row strides must be multiples of 4, bases 8-byte aligned");
    if (ld % 8 == 0 && ldd % 8 == 0 && ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) == 0)
        hipLaunchKernelGGL(ce_bwd_kernel<8>, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, labels, lse, dloss,
                           n_valid, (bf16_t*)dlogits, ldd, (int)V);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<4>, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, labels, lse, dloss,
                           n_valid, (bf16_t*)dlogits, ldd, (int)V);
    OTTER_CHECK_LAUNCH("cross_entropy_bwd");
    return OTTER_OK;
}

}  // extern "C"
