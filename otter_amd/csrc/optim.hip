// optim.hip -- multi-tensor gradient-norm clipping + AdamW for the trainable parameters of the recipe
// (pipeline/train/instruction_following.py:246-251: clip_grad_norm_(1.0) then AdamW.step()).
//
// torch does this in four sweeps over the 1.39 B fp32 parameters' state: per-tensor norms, an in-place scale of every
// gradient, and the fused AdamW update.  Here: one read of the gradients (sum of squares -> clip coefficient, left on the
// device) and one pass that reads g, p, m, v, applies the coefficient on the fly and writes p, m, v -- the scaled
// gradients are never written back (nothing reads them before the next zero_grad).  Arithmetic follows torch's fused
// AdamW kernel (ATen/native/cuda/fused_adam_utils.cuh, ADAMW mode, no amsgrad, no maximize) operation by operation.
// All tensors of a step are addressed through one device-resident table; a block owns one CHUNK of one tensor.
#include "common.h"

namespace {

constexpr int CHUNK = 8192;  // elements per block
constexpr int NT = 256;

__device__ __forceinline__ bool vec_ok(const otter_adamw_tensor& t) {
    return ((((uintptr_t)t.p) | ((uintptr_t)t.g) | ((uintptr_t)t.m) | ((uintptr_t)t.v)) & 15) == 0;
}

__global__ __launch_bounds__(NT) void sumsq_kernel(const otter_adamw_tensor* __restrict__ tensors, const int32_t* __restrict__ blk_tensor,
                                                   const int32_t* __restrict__ blk_chunk, float* __restrict__ partials) {
    const otter_adamw_tensor t = tensors[blk_tensor[blockIdx.x]];
    const int64_t beg = (int64_t)blk_chunk[blockIdx.x] * CHUNK;
    const int64_t end = beg + CHUNK < t.numel ? beg + CHUNK : t.numel;
    float acc = 0.f;
    if ((((uintptr_t)t.g) & 15) == 0) {
        const int64_t nv = (end - beg) >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(t.g + beg);
        for (int64_t i = threadIdx.x; i < nv; i += NT) {
            const float4 x = g4[i];
            acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        }
        for (int64_t i = beg + (nv << 2) + threadIdx.x; i < end; i += NT) acc += t.g[i] * t.g[i];
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += NT) acc += t.g[i] * t.g[i];
    }
    acc = wave_sum(acc);
    __shared__ float red[NT / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(1024) void clip_coef_kernel(const float* __restrict__ partials, int64_t n, float max_norm, float* __restrict__ out) {
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += (double)partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        const float norm = (float)sqrt(t);
        out[0] = norm;
        const float c = max_norm / (norm + 1e-6f);
        out[1] = c < 1.0f ? c : 1.0f;
    }
}

struct Hyper { float lr, beta1, beta2, eps, bc1, bc2_sqrt; };

// per-tensor overrides (torch keeps one `step` per parameter and one lr per param group): a table row with
// bias_correction1 != 0 carries its own lr and bias corrections, otherwise the launch-wide values apply
__device__ __forceinline__ Hyper hyper_of(const otter_adamw_tensor& t, Hyper h) {
    if (t.bias_correction1 != 0.f) { h.lr = t.lr; h.bc1 = t.bias_correction1; h.bc2_sqrt = t.bias_correction2_sqrt; }
    return h;
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float wd, const Hyper& h) {
    p -= h.lr * wd * p;
    m = m + (1.0f - h.beta1) * (g - m);                         // lerp(exp_avg, grad, 1 - beta1)
    v = h.beta2 * v + (1.0f - h.beta2) * g * g;
    const float step_size = h.lr / h.bc1;
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(NT) void adamw_kernel(const otter_adamw_tensor* __restrict__ tensors, const int32_t* __restrict__ blk_tensor,
                                                   const int32_t* __restrict__ blk_chunk, Hyper h0, const float* __restrict__ grad_scale) {
    const otter_adamw_tensor t = tensors[blk_tensor[blockIdx.x]];
    const Hyper h = hyper_of(t, h0);
    const int64_t beg = (int64_t)blk_chunk[blockIdx.x] * CHUNK;
    const int64_t end = beg + CHUNK < t.numel ? beg + CHUNK : t.numel;
    const float gs = grad_scale ? *grad_scale : 1.0f;
    int64_t tail = beg;
    if (vec_ok(t)) {
        const int64_t nv = (end - beg) >> 2;
        float4* p4 = reinterpret_cast<float4*>(t.p + beg);
        const float4* g4 = reinterpret_cast<const float4*>(t.g + beg);
        float4* m4 = reinterpret_cast<float4*>(t.m + beg);
        float4* v4 = reinterpret_cast<float4*>(t.v + beg);
        for (int64_t i = threadIdx.x; i < nv; i += NT) {
            float4 p = p4[i], m = m4[i], v = v4[i];
            const float4 g = g4[i];
            adamw_one(p.x, g.x * gs, m.x, v.x, t.weight_decay, h);
            adamw_one(p.y, g.y * gs, m.y, v.y, t.weight_decay, h);
            adamw_one(p.z, g.z * gs, m.z, v.z, t.weight_decay, h);
            adamw_one(p.w, g.w * gs, m.w, v.w, t.weight_decay, h);
            p4[i] = p; m4[i] = m; v4[i] = v;
            if (t.shadow) {
                uint2 w;
                w.x = pack2bf(p.x, p.y);
                w.y = pack2bf(p.z, p.w);
                *reinterpret_cast<uint2*>(t.shadow + beg + 4 * i) = w;
            }
        }
        tail = beg + (nv << 2);
    }
    for (int64_t i = tail + threadIdx.x; i < end; i += NT) {
        float p = t.p[i], m = t.m[i], v = t.v[i];
        adamw_one(p, t.g[i] * gs, m, v, t.weight_decay, h);
        t.p[i] = p; t.m[i] = m; t.v[i] = v;
        if (t.shadow) t.shadow[i] = f2bf(p);
    }
}

// Variant of the update kernel that streams (OTTER_ADAMW_VARIANT, read once; tools/adamw_bench.py): NT = non-temporal loads and stores
// (every byte of g, p, m, v is touched once per step: 39 GB against a 256 MB Infinity Cache), U = float4 groups per thread in flight.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <bool NTMP, int U>
__global__ __launch_bounds__(NT) void adamw_stream_kernel(const otter_adamw_tensor* __restrict__ tensors, const int32_t* __restrict__ blk_tensor,
                                                          const int32_t* __restrict__ blk_chunk, Hyper h0, const float* __restrict__ grad_scale) {
    const otter_adamw_tensor t = tensors[blk_tensor[blockIdx.x]];
    const Hyper h = hyper_of(t, h0);
    const int64_t beg = (int64_t)blk_chunk[blockIdx.x] * CHUNK;
    const int64_t end = beg + CHUNK < t.numel ? beg + CHUNK : t.numel;
    const float gs = grad_scale ? *grad_scale : 1.0f;
    int64_t tail = beg;
    if (vec_ok(t)) {
        const int64_t nv = (end - beg) >> 2;
        f32x4_t* p4 = reinterpret_cast<f32x4_t*>(t.p + beg);
        const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(t.g + beg);
        f32x4_t* m4 = reinterpret_cast<f32x4_t*>(t.m + beg);
        f32x4_t* v4 = reinterpret_cast<f32x4_t*>(t.v + beg);
        for (int64_t i0 = threadIdx.x; i0 < nv; i0 += NT * U) {
            f32x4_t p[U], m[U], v[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * NT;
                if (i < nv) {
                    if (NTMP) { p[u] = __builtin_nontemporal_load(p4 + i); m[u] = __builtin_nontemporal_load(m4 + i); v[u] = __builtin_nontemporal_load(v4 + i); g[u] = __builtin_nontemporal_load(g4 + i); }
                    else { p[u] = p4[i]; m[u] = m4[i]; v[u] = v4[i]; g[u] = g4[i]; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * NT;
                if (i < nv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pe = p[u][e], me = m[u][e], ve = v[u][e];
                        adamw_one(pe, g[u][e] * gs, me, ve, t.weight_decay, h);
                        p[u][e] = pe; m[u][e] = me; v[u][e] = ve;
                    }
                    if (NTMP) { __builtin_nontemporal_store(p[u], p4 + i); __builtin_nontemporal_store(m[u], m4 + i); __builtin_nontemporal_store(v[u], v4 + i); }
                    else { p4[i] = p[u]; m4[i] = m[u]; v4[i] = v[u]; }
                    if (t.shadow) {
                        uint2 w;
                        w.x = pack2bf(p[u][0], p[u][1]);
                        w.y = pack2bf(p[u][2], p[u][3]);
                        *reinterpret_cast<uint2*>(t.shadow + beg + 4 * i) = w;   // read by the next forward's GEMMs: stays a normal store
                    }
                }
            }
        }
        tail = beg + (nv << 2);
    }
    for (int64_t i = tail + threadIdx.x; i < end; i += NT) {
        float p = t.p[i], m = t.m[i], v = t.v[i];
        adamw_one(p, t.g[i] * gs, m, v, t.weight_decay, h);
        t.p[i] = p; t.m[i] = m; t.v[i] = v;
        if (t.shadow) t.shadow[i] = f2bf(p);
    }
}

int adamw_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OTTER_ADAMW_VARIANT"); v = e ? atoi(e) : 1; }  // default 1: non-temporal, one group in flight (7.53 -> 7.25 ms on 1.34 B parameters)
    return v;
}

}  // namespace

extern "C" {

int otter_adamw_chunk(void) { return CHUNK; }

int otter_grad_sumsq(const otter_adamw_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk, int64_t nblocks,
                     float* partials, void* stream) {
    OTTER_REQUIRE(tensors && blk_tensor && blk_chunk && partials && nblocks > 0, "grad_sumsq: bad args");
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nblocks), dim3(NT), 0, (hipStream_t)stream, tensors, blk_tensor, blk_chunk, partials);
    OTTER_CHECK_LAUNCH("grad_sumsq");
    return OTTER_OK;
}

int otter_clip_coef(const float* partials, int64_t n, float max_norm, float* out2, void* stream) {
    OTTER_REQUIRE(partials && out2 && n > 0 && max_norm > 0.f, "clip_coef: bad args");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, n, max_norm, out2);
    OTTER_CHECK_LAUNCH("clip_coef");
    return OTTER_OK;
}

int otter_adamw_step(const otter_adamw_tensor* tensors, const int32_t* blk_tensor, const int32_t* blk_chunk, int64_t nblocks, float lr,
                     float beta1, float beta2, float eps, float bias_correction1, float bias_correction2, const float* grad_scale,
                     void* stream) {
    OTTER_REQUIRE(tensors && blk_tensor && blk_chunk && nblocks > 0, "adamw_step: bad args");
    OTTER_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, "adamw_step: bias corrections must be positive (step >= 1)");
    Hyper h{lr, beta1, beta2, eps, bias_correction1, sqrtf(bias_correction2)};
    const dim3 grid((unsigned)nblocks), block(NT);
    switch (adamw_variant()) {
        case 1: hipLaunchKernelGGL((adamw_stream_kernel<true, 1>), grid, block, 0, (hipStream_t)stream, tensors, blk_tensor, blk_chunk, h, grad_scale); break;
        case 2: hipLaunchKernelGGL((adamw_stream_kernel<true, 2>), grid, block, 0, (hipStream_t)stream, tensors, blk_tensor, blk_chunk, h, grad_scale); break;
        case 3: hipLaunchKernelGGL((adamw_stream_kernel<false, 2>), grid, block, 0, (hipStream_t)stream, tensors, blk_tensor, blk_chunk, h, grad_scale); break;
        case 4: hipLaunchKernelGGL((adamw_stream_kernel<true, 4>), grid, block, 0, (hipStream_t)stream, tensors, blk_tensor, blk_chunk, h, grad_scale); break;
        default: hipLaunchKernelGGL(adamw_kernel, grid, block, 0, (hipStream_t)stream, tensors, blk_tensor, blk_chunk, h, grad_scale); break;
    }
    OTTER_CHECK_LAUNCH("adamw_step");
    return OTTER_OK;
}

}  // extern "C"
