// common.h -- shared device/host helpers for libotter_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/otter_hip.h"

#define OTTER_WAVE 64

// ---- error plumbing (thread-local message, negative status codes; nothing throws across the C ABI) ----
extern thread_local char g_otter_err[512];
#define OTTER_FAIL(code, ...)                                   \
    do {                                                        \
        snprintf(g_otter_err, sizeof(g_otter_err), __VA_ARGS__); \
        return (code);                                          \
    } while (0)
#define OTTER_REQUIRE(cond, ...)                         \
    do {                                                 \
        if (!(cond)) OTTER_FAIL(OTTER_ERR_ARG, __VA_ARGS__); \
    } while (0)
#define OTTER_CHECK_LAUNCH(name)                                                              \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) OTTER_FAIL(OTTER_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- bf16 <-> f32 (bf16 stored as uint16_t: upper half of an IEEE float, RNE on store) ----
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// gfx950 converts in hardware: v_cvt_pk_bf16_f32 (two values per instruction, round-to-nearest-even, NaN stays NaN).  The
// software form this replaces (add 0x7fff + lsb, NaN branch) was ~10 VALU + an exec-mask branch PER ELEMENT: thousands of
// cycles in every output tile of the GEMM tails, where one wave per SIMD has nothing to hide them behind.
typedef __attribute__((ext_vector_type(2))) float otter_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 otter_bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const otter_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, otter_bf16x2_t));
}

// scalar typed access through a runtime dtype tag (used only on cold / tail paths)
__device__ __forceinline__ float ld_as_f32(const void* p, int64_t i, int dtype) {
    return dtype == OTTER_BF16 ? bf2f(((const bf16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void st_from_f32(void* p, int64_t i, int dtype, float v) {
    if (dtype == OTTER_BF16) ((bf16_t*)p)[i] = f2bf(v);
    else ((float*)p)[i] = v;
}

// 8 consecutive elements -> 8 floats.  bf16: one 16-B load; f32: two 16-B loads.  p must be 16-B aligned (bf16)
// or 16-B aligned (f32) at element index i (i multiple of 8 and row strides multiples of 8 guarantee it).
template <typename T>
struct Vec8;
template <>
struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        uint4 r = *reinterpret_cast<const uint4*>(p);
        uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        uint4 r;
        r.x = pack2bf(v[0], v[1]);
        r.y = pack2bf(v[2], v[3]);
        r.z = pack2bf(v[4], v[5]);
        r.w = pack2bf(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = r;
    }
};
template <>
struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        float4 a = *reinterpret_cast<const float4*>(p);
        float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

// ---- wave64 reductions (butterfly over all 64 lanes; every lane gets the result) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Phi(u) = 0.5 (1 + erf(u / sqrt 2)) and phi(u) = exp(-u^2/2) / sqrt(2 pi) from ONE exponential:
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-roundoff class), 1 v_rcp_f32 + 5 fma + 1 v_exp_f32 instead of
// libm's branchy erff (~40 VALU) plus a separate expf.  Used by the GEMM epilogues (GELU forward, gate/GELU backward).
__device__ __forceinline__ void gelu_cdf_pdf(float u, float& cdf, float& pdf) {
    const float x = fabsf(u) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));  // v_rcp_f32 (1 ulp); __frcp_rn expands to the 11-instruction IEEE division
    const float e = __expf(-x * x);  // = exp(-u^2 / 2)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_abs = p * t * e;             // 1 - erf(|x|)
    const float half = 0.5f * erfc_abs;
    cdf = u >= 0.f ? 1.0f - half : half;
    pdf = 0.39894228040143267794f * e;
}

// Phi(u) for the bf16-output tails of the GEMM (round 6), two elements per instruction: Phi(u) - 1/2 is odd, = u * Q(u^2) with a degree-8
// minimax Q on |u| <= 4.2 (tools/gen/gelu_poly.py), u clamped to +-4.2 where the fit is scaled to reach exactly +-1/2 -- so the result
// saturates at 0 / 1 instead of drifting.  |error| <= 1.5e-5 absolute in fp32 Horner form (bf16's half-ulp at 1 is 2e-3); 2 v_med3 + 11
// packed fp32 instructions per PAIR against 15 scalar ones + rcp + exp per ELEMENT for gelu_cdf_pdf: with one wave per SIMD every
// instruction of the tail is an issue slot (~4 cycles), and the GELU tail was 256 elements x ~17 slots per lane = two thirds of its 28 k cycles.
typedef __attribute__((ext_vector_type(2))) float otter_f2;
// NP independent pairs at once, the Horner levels in the OUTER loop: consecutive instructions belong to different pairs, so no level waits for
// the one before it (written pair by pair, hipcc kept each pair's eleven-deep dependent chain together -- s_nop between every two v_pk_fma --
// and the tail took as long as with the scalar form).  Returns Phi(u[p]); uc = the clamped arguments.
template <int NP>
__device__ __forceinline__ void gelu_cdf_fast(const otter_f2 (&u)[NP], otter_f2 (&cdf)[NP]) {
    otter_f2 uc[NP], s[NP], q[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        uc[p].x = __builtin_amdgcn_fmed3f(u[p].x, -4.2f, 4.2f);
        uc[p].y = __builtin_amdgcn_fmed3f(u[p].y, -4.2f, 4.2f);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = uc[p] * uc[p];
#pragma unroll
    for (int p = 0; p < NP; ++p) q[p] = s[p] * 5.998143648e-11f + -5.633387978e-09f;
    constexpr float C[7] = {2.343703045e-07f, -5.760839940e-06f, 9.457554552e-05f, -1.114161452e-03f, 9.830248542e-03f, -6.636116654e-02f, 3.989122212e-01f};
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int p = 0; p < NP; ++p) q[p] = q[p] * s[p] + C[k];
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) cdf[p] = uc[p] * q[p] + 0.5f;
}
// GELU(v) of NE = 2 NP values
template <int NE>
__device__ __forceinline__ void gelu_fast(const float (&v)[NE], float (&o)[NE]) {
    otter_f2 u[NE / 2], cdf[NE / 2];
#pragma unroll
    for (int p = 0; p < NE / 2; ++p) u[p] = otter_f2{v[2 * p], v[2 * p + 1]};
    gelu_cdf_fast<NE / 2>(u, cdf);
#pragma unroll
    for (int p = 0; p < NE / 2; ++p) {
        const otter_f2 r = u[p] * cdf[p];
        o[2 * p] = r.x; o[2 * p + 1] = r.y;
    }
}
// GELU(v) AND GELU'(v) of NE = 2 NP values (round 6c, the derivative-stash form of the frozen MLP: the forward tail writes h = GELU(u) and
// g = GELU'(u) = Phi(u) + u phi(u) instead of u; the backward tail then only multiplies): the polynomial Phi + one v_exp_f32 per element
template <int NE>
__device__ __forceinline__ void gelu_and_grad_fast(const float (&v)[NE], float (&o)[NE], float (&gp)[NE]) {
    constexpr int NP = NE / 2;
    otter_f2 u[NP], cdf[NP], e[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) u[p] = otter_f2{v[2 * p], v[2 * p + 1]};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const otter_f2 arg = (u[p] * u[p]) * -0.72134752044448170368f;   // exp(-u^2 / 2) = exp2(-u^2 / (2 ln 2))
        e[p].x = __builtin_amdgcn_exp2f(arg.x);
        e[p].y = __builtin_amdgcn_exp2f(arg.y);
    }
    gelu_cdf_fast<NP>(u, cdf);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const otter_f2 r = u[p] * cdf[p];
        const otter_f2 g = (u[p] * e[p]) * 0.39894228040143267794f + cdf[p];
        o[2 * p] = r.x; o[2 * p + 1] = r.y;
        gp[2 * p] = g.x; gp[2 * p + 1] = g.y;
    }
}
// gate / GELU backward of NE values: o = s v GELU'(a) with GELU'(a) = Phi(a) + a phi(a) (the polynomial Phi + ONE v_exp_f32 per element for
// the density), returns sum v GELU(a)
template <int NE>
__device__ __forceinline__ float gelu_bwd_fast(float sc, const float (&v)[NE], const float (&a)[NE], float (&o)[NE]) {
    constexpr int NP = NE / 2;
    otter_f2 a2[NP], v2[NP], cdf[NP], e[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { a2[p] = otter_f2{a[2 * p], a[2 * p + 1]}; v2[p] = otter_f2{v[2 * p], v[2 * p + 1]}; }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const otter_f2 arg = (a2[p] * a2[p]) * -0.72134752044448170368f;   // exp(-a^2 / 2) = exp2(-a^2 / (2 ln 2))
        e[p].x = __builtin_amdgcn_exp2f(arg.x);
        e[p].y = __builtin_amdgcn_exp2f(arg.y);
    }
    gelu_cdf_fast<NP>(a2, cdf);
    otter_f2 part2 = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const otter_f2 grad = (a2[p] * e[p]) * 0.39894228040143267794f + cdf[p];
        part2 = v2[p] * (a2[p] * cdf[p]) + part2;
        const otter_f2 r = (v2[p] * sc) * grad;
        o[2 * p] = r.x; o[2 * p + 1] = r.y;
    }
    return part2.x + part2.y;
}

// exact-erf GELU and its derivative (nn.GELU() default, approximate='none')
__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float u) {
    const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * u * u);
    return cdf + u * pdf;
}

__host__ __device__ __forceinline__ int64_t map_row(int64_t r, otter_rowmap m) {
    return m.grp_rows > 0 ? (r / m.grp_rows) * m.grp_stride + m.row_off + (r % m.grp_rows) : r;
}

static inline size_t dtype_size(int dt) { return dt == OTTER_BF16 ? 2 : 4; }
