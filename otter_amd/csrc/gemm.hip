// gemm.hip -- C[M,N] = epilogue(A[M,K] . B[N,K]^T) on gfx950 MFMA, plus the small helpers around it.
//
// bf16 kernel (the MFMA-bound kernel of the path: the gated cross-attention FFN is 96.8 % of a block's FLOPs):
//   * block tile BM x BN x 64, WM x WN waves; each wave owns a (BM/WM) x (BN/WN) sub-tile as MI x NI blocks of
//     32x32 accumulated by v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
//   * operands are fed SWAPPED (a-operand = B rows, b-operand = A rows) so that a lane's 16 accumulator registers
//     hold, for ONE output row m, four runs of 4 consecutive n: the epilogue then reads/writes 8-byte (bf16) or
//     16-byte (f32) vectors of C / residual / aux instead of 2-byte scalars.
//   * LDS image per operand tile: [rows][64] bf16 = 128 B rows, 16-B slot index XOR-swizzled with (row>>1)&7, so a
//     ds_read_b128 lane group (16 lanes = 16 distinct rows, one k-slot) touches 16 distinct 16-B slots of the two
//     256-B bank rows -> conflict-free (MI355X guide, LDS section).
//   * double-buffered LDS, ONE barrier per K-tile: global loads for tile t+1 are issued before the MFMAs of tile t
//     and written to the other buffer after them (register staging), or -- variant 3 -- go straight to LDS with
//     global_load_lds_dwordx4 (the swizzle then lives on the per-lane SOURCE address; LDS destination is lane-linear).
//   * XCD-aware block order: consecutive (swizzled) ids sweep M inside one N panel, and each XCD gets a contiguous
//     chunk of that order, so a weight panel is fetched into one XCD's L2 once.
// f32 kernel (parity mode, exact f32): v_mfma_f32_32x32x2_f32, 64x64x32 tile, padded LDS rows.
#include "common.h"
#include <type_traits>
typedef float f32x4_t __attribute__((ext_vector_type(4)));

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

thread_local char g_otter_err[512] = {0};

// Diagnostics (otter_gemm_set_debug bit 64 + otter_gemm_read_timeline): tile-phase timestamps (s_memtime, shader cycles) of
// the one-wave-per-SIMD kernels, blocks 0 and 131, every wave, the first 8 tiles of the block:
// [block sel 2][wave 4][tile 8][mark 8] -- marks: 0 tile start, 1 prologue done, 2 K loop done, 3 tail done, 4 tile end; variant 26 also
// stamps the 100 MHz wall clock (s_memrealtime) at tile start (mark 5) and tile end (mark 6): cycles / ticks = the shader clock of that launch,
// and the launch's shape + operand layout (mark 7: M in 20 bits, N, K in 21 bits each, bits 63 / 62 = A / B K-major).
__device__ unsigned long long g_gemm_timeline[2 * 4 * 8 * 8];

// tools-only ablation builds of variant 26's K loop (-DOTTER_T4_ABL=mask; build.build_gemm_define): 1 = no LDS-DMA inside the K loop, 2 = no fragment
// reads, 4 = no workgroup barriers, 8 = no waits for the LDS-DMA, 16 = a piece's scalar instructions without its buffer_load.  Timing only -- the results are wrong by construction; the product build has mask 0.
#ifndef OTTER_T4_ABL
#define OTTER_T4_ABL 0
#endif
// cross-tile K-contiguous instantiations of variant 26: M0 written once per four LDS-DMA pieces (1 default; 0 = one s_mov per piece: the A/B build)
#ifndef OTTER_T4_M0GROUP
#define OTTER_T4_M0GROUP 1
#endif
// ... and the scalar source offsets of a K-tile computed on three free slots of the schedule (0 default; 1 = in one burst in front of the K-tile: the A/B build)
// ... and that M0 written one MFMA slot before the group's first piece (1 default; 0 = in the piece's own slot, with an s_nop: the A/B build)
#ifndef OTTER_T4_M0EARLY
#define OTTER_T4_M0EARLY 1
#endif
#ifndef OTTER_T4_SETK_BURST
#define OTTER_T4_SETK_BURST 0
#endif
namespace {

struct GemmArgs {
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* C; int64_t ldc; int cdt;
    int64_t M, N, K;
    int kind, accumulate;
    const float* gate;
    const void* R; int64_t ldr; int rdt;
    void* C2; int64_t ldc2;
    const void* aux; int64_t ldaux; int auxdt; int aux_gelu;
    float* partial;
    int gm, gn;
    int dbg;  // diagnostics only (otter_gemm_set_debug): bit0 = skip K-loop global loads, bit1 = skip MFMAs
    int order;  // tile order override (diagnostics, see tile_of_block)
    int wide;  // bf16 output and every tensor the fused tail touches allows 8-element accesses (N, ldc, ldc2, ldr, ldaux % 8 == 0)
    int ta, tb;  // operand A / B is K-major ([K rows][M or N columns]); variant 26 only (otter_gemm)
    int grid_mode;  // host side only: 0 = process default (otter_gemm_set_persistent), 1 = persistent grid, 2 = one workgroup per tile
    int korder;  // cross-tile form of variant 26: K-tile order of a tile (bits 0-1 rotation by XCD / tile, bits 2-3 in-group permutation; see xt_tile);
                 // bits 4-6: start phase step (units of 1024 cycles) between the four workgroup phases
};

__device__ __forceinline__ void load4(const void* p, int64_t idx, int dt, float (&v)[4]) {
    if (dt == OTTER_BF16) {
        const uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)p + idx);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    } else {
        const float4 r = *reinterpret_cast<const float4*>((const float*)p + idx);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
}
// Outputs are written once and not re-read by this kernel: non-temporal stores keep them from displacing the operand
// panels in the XCD's L2 (and from being written back in the middle of the next tile's K loop).
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
#ifndef OTTER_DIAG
#define OTTER_DIAG 0
#endif
// diag bit 128: plain (cacheable, write-back) stores in the fused tail instead of non-temporal ones (timing experiment)
template <typename V>
__device__ __forceinline__ void out_store(V v, V* p) {
    if constexpr ((OTTER_DIAG & 128) != 0) *p = v;
    else __builtin_nontemporal_store(v, p);
}
__device__ __forceinline__ void store4(void* p, int64_t idx, int dt, const float (&v)[4]) {
    if (dt == OTTER_BF16) {
        u32x2_t r;
        r.x = pack2bf(v[0], v[1]);
        r.y = pack2bf(v[2], v[3]);
        out_store(r, reinterpret_cast<u32x2_t*>((bf16_t*)p + idx));
    } else {
        f32x4_t r = {v[0], v[1], v[2], v[3]};
        out_store(r, reinterpret_cast<f32x4_t*>((float*)p + idx));
    }
}

// GELU of NE accumulator values / the GELU-backward of the gate-backward tail (o = s v GELU'(a), returns sum v GELU(a)).  bf16 results take
// the packed polynomial Phi of common.h (gelu_cdf_fast2: |error| <= 1.5e-5, far inside bf16 rounding), fp32 results the A&S 7.1.26 form
// (1.5e-7): every kernel and every tail flavour of this file goes through these two, so one launch shape gives one result.
template <int NE>
__device__ __forceinline__ void gelu_apply(bool fast, const float (&v)[NE], float (&o)[NE]) {
    if (fast) {
        gelu_fast<NE>(v, o);
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            float cdf, pdf;
            gelu_cdf_pdf(v[i], cdf, pdf);
            o[i] = v[i] * cdf;
        }
    }
}
// GELU forward whose second output is the DERIVATIVE (otter_epilogue_args::aux_is_gelu_input == 3 on an OTTER_EPI_GELU launch)
template <int NE>
__device__ __forceinline__ void gelu_apply_stash(bool fast, const float (&v)[NE], float (&o)[NE], float (&gp)[NE]) {
    if (fast) {
        gelu_and_grad_fast<NE>(v, o, gp);
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            float cdf, pdf;
            gelu_cdf_pdf(v[i], cdf, pdf);
            o[i] = v[i] * cdf;
            gp[i] = cdf + v[i] * pdf;
        }
    }
}
template <int NE>
__device__ __forceinline__ float gelu_bwd_apply(bool fast, float s, const float (&v)[NE], const float (&a)[NE], float (&o)[NE]) {
    float part = 0.f;
    if (fast) {
        part = gelu_bwd_fast<NE>(s, v, a, o);
    } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) {   // gelu and gelu' share the erf: one transcendental chain instead of two
            float cdf, pdf;
            gelu_cdf_pdf(a[i], cdf, pdf);
            part += v[i] * (a[i] * cdf);
            o[i] = s * v[i] * (cdf + a[i] * pdf);
        }
    }
    return part;
}

// Plain-store launches with an fp32 C (weight gradients) can also hand back sum(C^2) per tile -- the clip_grad_norm_ reduction of that tensor,
// taken while the values are in registers instead of by a second 4-byte-per-element sweep over it (otter_epilogue_args::partial, round 6b).
template <int NE>
__device__ __forceinline__ float sumsq_of(const float (&o)[NE]) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) t = fmaf(o[i], o[i], t);
    return t;
}

// epilogue on 4 consecutive output columns of row m; returns this thread's contribution to the gate partial (plain store, fp32 C: to sum(C^2))
template <int EPI>
__device__ __forceinline__ float epilogue4(const GemmArgs& g, float s, int64_t m, int64_t n, float (&v)[4]) {
    float part = 0.f;
    float o[4];
    switch (EPI) {
        case OTTER_EPI_STORE: {
            if (g.accumulate) {
                float c[4];
                load4(g.C, m * g.ldc + n, g.cdt, c);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = c[i] + s * v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = s * v[i];
            }
            store4(g.C, m * g.ldc + n, g.cdt, o);
            if (g.cdt == OTTER_F32) part = sumsq_of<4>(o);   // (written out only when the launch asked for partials: block_partial)
            break;
        }
        case OTTER_EPI_GELU: {
            if (g.C2 && g.aux_gelu == 3) {   // derivative stash: C2 = GELU'(acc)
                float gp[4];
                gelu_apply_stash<4>(g.cdt == OTTER_BF16, v, o, gp);
                store4(g.C2, m * g.ldc2 + n, g.cdt, gp);
            } else {
                if (g.C2) store4(g.C2, m * g.ldc2 + n, g.cdt, v);
                gelu_apply<4>(g.cdt == OTTER_BF16, v, o);   // bf16 results: the packed polynomial, as in the full-tile tails (tail_apply)
            }
            store4(g.C, m * g.ldc + n, g.cdt, o);
            break;
        }
        case OTTER_EPI_SCALE_RES: {
            float r[4];
            load4(g.R, m * g.ldr + n, g.rdt, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = v[i] * s + r[i];
            store4(g.C, m * g.ldc + n, g.cdt, o);
            break;
        }
        default: {  // OTTER_EPI_GATE_BWD
            float a[4];
            load4(g.aux, m * g.ldaux + n, g.auxdt, a);
            if (g.aux_gelu == 3) {   // aux IS the stashed derivative f'(a): no activation arithmetic, no gate partial
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = s * v[i] * a[i];
            } else if (g.aux_gelu == 2) {   // squared ReLU (Persimmon MLP): f(a) = relu(a)^2, f'(a) = 2 relu(a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = fmaxf(a[i], 0.f);
                    part += v[i] * (r * r);
                    o[i] = s * v[i] * (2.0f * r);
                }
            } else if (g.aux_gelu) {   // tested outside the element loop (a per-element scalar branch serialises the chains)
                part += gelu_bwd_apply<4>(g.cdt == OTTER_BF16, s, v, a, o);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    part += v[i] * a[i];
                    o[i] = s * v[i];
                }
            }
            store4(g.C, m * g.ldc + n, g.cdt, o);
            break;
        }
    }
    return part;
}

// ---- 8 consecutive output columns per lane: 16-byte bf16 (2 x 16-byte f32) accesses, half as many store instructions ----
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ void load8w(const void* p, int64_t idx, int dt, float (&v)[8]) {
    if (dt == OTTER_BF16) {
        const uint4 r = *reinterpret_cast<const uint4*>((const bf16_t*)p + idx);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    } else {
        const float4 a = *reinterpret_cast<const float4*>((const float*)p + idx), b = *reinterpret_cast<const float4*>((const float*)p + idx + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
__device__ __forceinline__ void store8w(void* p, int64_t idx, int dt, const float (&v)[8]) {
    if (dt == OTTER_BF16) {
        u32x4_t r;
        r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]); r.z = pack2bf(v[4], v[5]); r.w = pack2bf(v[6], v[7]);
        out_store(r, reinterpret_cast<u32x4_t*>((bf16_t*)p + idx));
    } else {
        f32x4_t a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        out_store(a, reinterpret_cast<f32x4_t*>((float*)p + idx));
        out_store(b, reinterpret_cast<f32x4_t*>((float*)p + idx + 4));
    }
}
template <int EPI>
__device__ __forceinline__ float epilogue8(const GemmArgs& g, float s, int64_t m, int64_t n, float (&v)[8]) {
    float part = 0.f;
    float o[8];
    switch (EPI) {
        case OTTER_EPI_STORE: {
            if (g.accumulate) {
                float c[8];
                load8w(g.C, m * g.ldc + n, g.cdt, c);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = c[i] + s * v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = s * v[i];
            }
            if (g.cdt == OTTER_F32) part = sumsq_of<8>(o);
            break;
        }
        case OTTER_EPI_GELU: {
            if (g.C2 && g.aux_gelu == 3) {
                float gp[8];
                gelu_apply_stash<8>(g.cdt == OTTER_BF16, v, o, gp);
                store8w(g.C2, m * g.ldc2 + n, g.cdt, gp);
            } else {
                if (g.C2) store8w(g.C2, m * g.ldc2 + n, g.cdt, v);
                gelu_apply<8>(g.cdt == OTTER_BF16, v, o);   // bf16 results: the packed polynomial, as in the full-tile tails (tail_apply)
            }
            break;
        }
        case OTTER_EPI_SCALE_RES: {
            float r[8];
            load8w(g.R, m * g.ldr + n, g.rdt, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = v[i] * s + r[i];
            break;
        }
        default: {  // OTTER_EPI_GATE_BWD
            float a[8];
            load8w(g.aux, m * g.ldaux + n, g.auxdt, a);
            if (g.aux_gelu == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = s * v[i] * a[i];
            } else if (g.aux_gelu == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float r = fmaxf(a[i], 0.f);
                    part += v[i] * (r * r);
                    o[i] = s * v[i] * (2.0f * r);
                }
            } else if (g.aux_gelu) {
                part += gelu_bwd_apply<8>(g.cdt == OTTER_BF16, s, v, a, o);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    part += v[i] * a[i];
                    o[i] = s * v[i];
                }
            }
            break;
        }
    }
    store8w(g.C, m * g.ldc + n, g.cdt, o);
    return part;
}

// Epilogue: a wave parks one 32-row stripe of its sub-tile (two 32x32 accumulator blocks side by side = 32 rows x 64
// columns) in wave-private LDS as fp32, then reads it back row-major: lane l handles row (l>>4) + 4*it, columns
// 4*(l&15)..+3, so 16 lanes cover one full 128-B (bf16) / 256-B (f32) run of an output row and every global access of
// the fused tail (C, C2, residual, aux) is a whole cache line.  The epilogue kind is a template parameter (one copy of
// the tail per kernel) and the row loop is a run-time loop, so the (mi) loop that indexes the accumulator registers
// statically stays small enough to be fully unrolled -- a single runtime-switched body made the unroller give up and
// demoted the accumulators to scratch; an out-of-line body took its arguments through flat (generic) pointers.
constexpr int EPI_LD = 68;  // floats per parked row (64 + 4 pad; rows stay 16-B aligned)

// Compile-time diagnostics for the ablation builds of tools/gemm_ablate.py (`python -m otter_amd.build --diag N` writes
// lib/libotter_hip_diagN.so; results are WRONG by construction).  Phased kernel only: 1 = no DMA, 2 = no MFMA,
// 4 = epilogue without its global loads/stores, 8 = no ds_read of fragments, 16 = no epilogue at all (accumulators are
// only summed so that the MFMAs stay live), 32 = half the A-fragment reads, 64 = half the DMA pieces.  Compile-time on purpose: a runtime flag inside the K-loop lambdas cost the
// product kernels 3-4x.
#ifndef OTTER_DIAG
#define OTTER_DIAG 0
#endif

template <int EPI>
__device__ __forceinline__ float epilogue_stripe(const GemmArgs& g, float s, const float* __restrict__ blk, int64_t m_base,
                                                 int64_t n_base, int lane) {
    float part = 0.f;
    if (g.wide && !((OTTER_DIAG & 4) || (g.dbg & 16))) {
        // lane l: row (l>>3) + 8*it, columns 8*(l&7)..+7 -- 8 lanes cover the 128-B (bf16) run of an output row with one
        // 16-byte access each: half the store / load instructions and loop trips of the 4-wide form below
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int r = (lane >> 3) + 8 * it;
            const int c = (lane & 7) * 8;
            const int64_t m = m_base + r, n = n_base + c;
            if (m < g.M && n < g.N) {
                const float4 t0 = *reinterpret_cast<const float4*>(blk + r * EPI_LD + c);
                const float4 t1 = *reinterpret_cast<const float4*>(blk + r * EPI_LD + c + 4);
                float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                part += epilogue8<EPI>(g, s, m, n, v);
            }
        }
        return part;
    }
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int r = (lane >> 4) + 4 * it;
        const int c = (lane & 15) * 4;
        const int64_t m = m_base + r, n = n_base + c;
        if (m < g.M && n < g.N) {
            const float4 t = *reinterpret_cast<const float4*>(blk + r * EPI_LD + c);
            float v[4] = {t.x, t.y, t.z, t.w};
            if ((OTTER_DIAG & 4) || (g.dbg & 16)) {  // diagnostics: everything but the global traffic of the tail
                part += v[0] + v[1] + v[2] + v[3];
            } else {
                part += epilogue4<EPI>(g, s, m, n, v);
            }
        }
    }
    return part;
}

// park one accumulator block at column offset col0 of the stripe: lane holds row m = lane&31, columns
// col0 + 8*grp + 4*(lane>>5) + 0..3
__device__ __forceinline__ void park_block(float* __restrict__ blk, const f32x16_t& a, int lane, int col0) {
    float* row = blk + (lane & 31) * EPI_LD + col0 + 4 * (lane >> 5);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp)
        *reinterpret_cast<float4*>(row + 8 * grp) = make_float4(a[4 * grp + 0], a[4 * grp + 1], a[4 * grp + 2], a[4 * grp + 3]);
}

// deterministic block reduction of the per-thread partial into partial[blockIdx.x]
template <int NWAVES, int EPI>
__device__ __forceinline__ void block_partial(const GemmArgs& g, float part, float* red /* LDS, >= NWAVES floats */, int slot) {
    if ((EPI != OTTER_EPI_GATE_BWD && EPI != OTTER_EPI_STORE) || g.partial == nullptr) return;
    part = wave_sum(part);
    __syncthreads();  // everyone is done with the LDS that `red` aliases
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) t += red[w];
        g.partial[slot] = t;
    }
}

// ---- fast tail of the one-wave-per-SIMD kernels (variants 17-20), FULL tiles only ----
// With 8 waves per CU the stripe loop above hides its own latencies (another wave always has something to issue); with one
// wave per SIMD every ds_write -> ds_read -> global access chain is exposed, and the run-time switches of the generic tail
// (output dtype, bounds, wide/narrow) sit inside its loops: the K sweep of tools/gemm_ksweep.py priced the fixed part of a
// variant-18 launch at 84 us against 46 us for the 8-wave kernel and 36 us for hipBLASLt.  Here the tile is known to be
// in bounds, the output dtype is a template parameter, the per-stripe loop is fully unrolled (all LDS reads of a stripe are
// issued before the first one is consumed) and consecutive stripes alternate between TWO parking buffers, so the next
// stripe is parked while the current one's global accesses are in flight.  Same arithmetic, same access shapes (whole
// cache lines per row run) as epilogue_stripe.
// The fused tail in two phases, so that the global INPUT of a stripe (residual R, gate-backward aux, the C being accumulated
// into) can be requested one stripe ahead of the arithmetic that consumes it: with one wave per SIMD an un-prefetched
// load -> use chain exposes a full memory latency eight times per tile (gate-backward: 626 us per launch in situ against
// 553 for the 8-wave kernel, whose two waves per SIMD cover each other).  NE = elements per lane per access (8 / 4).
template <int EPI>
__device__ __forceinline__ bool tail_input(const GemmArgs& g, const void*& p, int64_t& ld, int& dt) {
    if constexpr (EPI == OTTER_EPI_STORE) { p = g.C; ld = g.ldc; dt = g.cdt; return g.accumulate != 0; }
    else if constexpr (EPI == OTTER_EPI_SCALE_RES) { p = g.R; ld = g.ldr; dt = g.rdt; return true; }
    else if constexpr (EPI == OTTER_EPI_GATE_BWD) { p = g.aux; ld = g.ldaux; dt = g.auxdt; return true; }
    else { p = nullptr; ld = 0; dt = OTTER_F32; return false; }
}
// raw (unconverted) input of NE elements: the conversion to f32 stays with the consumer so that no use sits next to the load
// INBF16: the input dtype as a TEMPLATE parameter -- as a run-time test inside each load it made the compiler emit a branch and
// an `s_waitcnt vmcnt(0)` between consecutive loads (the "prefetch" ran one memory latency per access)
template <int NE, bool INBF16>
__device__ __forceinline__ void load_raw(const void* p, int64_t idx, uint4 (&raw)[2]) {
    if constexpr (INBF16) {
        if constexpr (NE == 8) raw[0] = *reinterpret_cast<const uint4*>((const bf16_t*)p + idx);
        else { const uint2 t = *reinterpret_cast<const uint2*>((const bf16_t*)p + idx); raw[0].x = t.x; raw[0].y = t.y; }
    } else {
        raw[0] = *reinterpret_cast<const uint4*>((const float*)p + idx);
        if constexpr (NE == 8) raw[1] = *reinterpret_cast<const uint4*>((const float*)p + idx + 4);
    }
}
template <int NE, bool INBF16>
__device__ __forceinline__ void cvt_raw(const uint4 (&raw)[2], float (&a)[NE]) {
    if constexpr (INBF16) {
        const uint32_t w[4] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w};
#pragma unroll
        for (int i = 0; i < NE / 2; ++i) { a[2 * i] = __uint_as_float(w[i] << 16); a[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    } else {
        a[0] = __uint_as_float(raw[0].x); a[1] = __uint_as_float(raw[0].y); a[2] = __uint_as_float(raw[0].z); a[3] = __uint_as_float(raw[0].w);
        if constexpr (NE == 8) {
            a[4] = __uint_as_float(raw[1].x); a[5] = __uint_as_float(raw[1].y); a[6] = __uint_as_float(raw[1].z); a[7] = __uint_as_float(raw[1].w);
        }
    }
}
// arithmetic + stores of NE consecutive output columns given the accumulator values v and the (already loaded) input a:
// exactly epilogue4 / epilogue8 with their load taken out
template <int EPI, int NE>
__device__ __forceinline__ float tail_apply(const GemmArgs& g, float s, int64_t m, int64_t n, const float (&v)[NE], const float (&a)[NE], bool has_in) {
    float part = 0.f;
    float o[NE];
    if constexpr (EPI == OTTER_EPI_STORE) {
        if (has_in) {
#pragma unroll
            for (int i = 0; i < NE; ++i) o[i] = a[i] + s * v[i];
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) o[i] = s * v[i];
        }
        if (g.cdt == OTTER_F32) part = sumsq_of<NE>(o);   // (g.cdt is a compile-time constant in the full-tile tails: bf16 stores carry nothing extra)
    } else if constexpr (EPI == OTTER_EPI_GELU) {
        if (g.C2 && g.aux_gelu == 3) {   // derivative stash (round 6c): C2 = GELU'(acc) instead of acc
            float gp[NE];
            gelu_apply_stash<NE>(g.cdt == OTTER_BF16, v, o, gp);
            if constexpr (NE == 8) store8w(g.C2, m * g.ldc2 + n, g.cdt, gp);
            else store4(g.C2, m * g.ldc2 + n, g.cdt, gp);
        } else {
            if (g.C2) {
                if constexpr (NE == 8) store8w(g.C2, m * g.ldc2 + n, g.cdt, v);
                else store4(g.C2, m * g.ldc2 + n, g.cdt, v);
            }
            gelu_apply<NE>(g.cdt == OTTER_BF16, v, o);   // (the dtype is a compile-time constant in the full-tile tails)
        }
    } else if constexpr (EPI == OTTER_EPI_SCALE_RES) {
#pragma unroll
        for (int i = 0; i < NE; ++i) o[i] = v[i] * s + a[i];
    } else {
        // the aux kind is tested ONCE, outside the element loop: inside it the compiler kept a scalar branch between
        // elements, which serialised eight independent rcp / exp / fma chains (gate-backward tail: 88 k cycles per tile)
        if (g.aux_gelu == 3) {   // aux is the stashed derivative (round 6c)
#pragma unroll
            for (int i = 0; i < NE; ++i) o[i] = s * v[i] * a[i];
        } else if (g.aux_gelu == 2) {   // squared ReLU: f'(a) = 2 relu(a)
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const float r = fmaxf(a[i], 0.f);
                part += v[i] * (r * r);
                o[i] = s * v[i] * (2.0f * r);
            }
        } else if (g.aux_gelu) {
            part += gelu_bwd_apply<NE>(g.cdt == OTTER_BF16, s, v, a, o);
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                part += v[i] * a[i];
                o[i] = s * v[i];
            }
        }
    }
    if constexpr (NE == 8) store8w(g.C, m * g.ldc + n, g.cdt, o);
    else store4(g.C, m * g.ldc + n, g.cdt, o);
    return part;
}

template <bool CBF16>
struct TailShape {  // bf16 output: 4 accesses of 8 columns per lane and stripe; f32 output: 8 accesses of 4 columns
    static constexpr int NE = CBF16 ? 8 : 4, NIT = CBF16 ? 4 : 8;
    static __device__ __forceinline__ int row(int lane, int it) { return CBF16 ? (lane >> 3) + 8 * it : (lane >> 4) + 4 * it; }
    static __device__ __forceinline__ int col(int lane) { return CBF16 ? (lane & 7) * 8 : (lane & 15) * 4; }
};

template <int EPI, bool CBF16, bool INBF16>
__device__ __forceinline__ void tail_stripe_load(const void* p, int64_t ld, int64_t m_base, int64_t n_base, int lane,
                                                 uint4 (&raw)[TailShape<CBF16>::NIT][2]) {
    using T = TailShape<CBF16>;
#pragma unroll
    for (int it = 0; it < T::NIT; ++it) load_raw<T::NE, INBF16>(p, (m_base + T::row(lane, it)) * ld + n_base + T::col(lane), raw[it]);
}

template <int EPI, bool CBF16, bool INBF16>
__device__ __forceinline__ float tail_stripe_full(GemmArgs g, float s, const float* __restrict__ blk, int64_t m_base, int64_t n_base, int lane,
                                                  const uint4 (&raw)[TailShape<CBF16>::NIT][2], bool has_in) {
    using T = TailShape<CBF16>;
    float part = 0.f;
    g.cdt = CBF16 ? OTTER_BF16 : OTTER_F32;  // compile-time constant from here on: the dtype switches of the stores fold away
    float v[T::NIT][T::NE];
#pragma unroll
    for (int it = 0; it < T::NIT; ++it) {
        const float* src = blk + T::row(lane, it) * EPI_LD + T::col(lane);
        const float4 t0 = *reinterpret_cast<const float4*>(src);
        v[it][0] = t0.x; v[it][1] = t0.y; v[it][2] = t0.z; v[it][3] = t0.w;
        if constexpr (CBF16) {
            const float4 t1 = *reinterpret_cast<const float4*>(src + 4);
            v[it][4] = t1.x; v[it][5] = t1.y; v[it][6] = t1.z; v[it][7] = t1.w;
        }
    }
#pragma unroll
    for (int it = 0; it < T::NIT; ++it) {
        float a[T::NE];
        if (has_in) cvt_raw<T::NE, INBF16>(raw[it], a);
        else {
#pragma unroll
            for (int i = 0; i < T::NE; ++i) a[i] = 0.f;
        }
        part += tail_apply<EPI, T::NE>(g, s, m_base + T::row(lane, it), n_base + T::col(lane), v[it], a, has_in);
    }
    return part;
}

// whole 128x128 wave tile = 8 stripes of 32 rows x 64 columns, FOUR parking buffers per wave: half of the wave tile is parked
// in one go (32 ds_write_b128 with static accumulator indices -- a dynamically indexed accumulator array is demoted to
// scratch, and so is one selected by a switch inside a loop: tried, 512 B of scratch per lane with accesses in the K loop),
// then a RUN-TIME loop walks the four parked stripes with ONE copy of the stripe arithmetic (unrolling it eight times made
// the GELU / gate-backward instantiations 22k-39k lines with 956 scratch accesses: 1150 us per launch in situ).  The
// buffers are wave-private and a wave's LDS operations execute in order, so no barrier is needed between the phases.
constexpr int TAIL_STRIPES = 4;
constexpr int TAIL_LDS_BYTES = 4 * TAIL_STRIPES * 32 * EPI_LD * 4;  // 4 waves: 139264 B

// stripe st (0..7) of a 128x128 wave tile = rows 32*(st>>1).., columns 64*(st&1)..; the two accumulator layouts park it differently
__device__ __forceinline__ void park_stripe(float* __restrict__ blk, const f32x16_t (&acc)[4][4], int st, int lane) {
    park_block(blk, acc[st >> 1][2 * (st & 1)], lane, 0);
    park_block(blk, acc[st >> 1][2 * (st & 1) + 1], lane, 32);
}
// 16x16 blocks (v_mfma_f32_16x16x32 with the operands swapped): lane holds row m = lane&15 of the block, columns 4*(lane>>4) + 0..3
__device__ __forceinline__ void park_stripe(float* __restrict__ blk, const f32x4_t (&acc)[8][8], int st, int lane) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4_t& v = acc[2 * (st >> 1) + a][4 * (st & 1) + b];
            *reinterpret_cast<float4*>(blk + (16 * a + (lane & 15)) * EPI_LD + 16 * b + 4 * (lane >> 4)) = make_float4(v[0], v[1], v[2], v[3]);
        }
}

// Accumulators held in EXPLICIT AGPRs (K-major instantiations of variant 26: block (mi, ni) = a[4 (8 mi + ni) .. + 3], written by asm MFMAs):
// parked straight out of the AGPR half with ds_write_b128 (DS instructions take AGPR data on gfx90a+), same LDS image as the overload
// above -- the values never become C++ values, so hipcc has nothing to copy, split or spill.  The caller has waited out the MFMA -> read
// hazard.  LDS operations of a wave complete in order: the tail's compiler-visible reads of `blk` that follow see these writes.
struct AgprAcc {};
#define PARK_AGPR_BLOCK(A_, B_, MI_, NI_)                                                                                             \
    asm volatile("ds_write_b128 %0, a[%c1:%c2] offset:%c3" : : "v"(addr), "n"(((MI_) * 8 + (NI_)) * 4), "n"(((MI_) * 8 + (NI_)) * 4 + 3), \
                 "n"(((A_) * 16 * EPI_LD + (B_) * 16) * 4) : "memory")
#define PARK_AGPR_STRIPE(ST_)                                                                                                          \
    do {                                                                                                                              \
        PARK_AGPR_BLOCK(0, 0, 2 * ((ST_) >> 1), 4 * ((ST_) & 1)); PARK_AGPR_BLOCK(0, 1, 2 * ((ST_) >> 1), 4 * ((ST_) & 1) + 1);         \
        PARK_AGPR_BLOCK(0, 2, 2 * ((ST_) >> 1), 4 * ((ST_) & 1) + 2); PARK_AGPR_BLOCK(0, 3, 2 * ((ST_) >> 1), 4 * ((ST_) & 1) + 3);     \
        PARK_AGPR_BLOCK(1, 0, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1)); PARK_AGPR_BLOCK(1, 1, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1) + 1); \
        PARK_AGPR_BLOCK(1, 2, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1) + 2); PARK_AGPR_BLOCK(1, 3, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1) + 3); \
    } while (0)
__device__ __forceinline__ void park_stripe(float* __restrict__ blk, const AgprAcc&, int st, int lane) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(blk + (lane & 15) * EPI_LD + 4 * (lane >> 4));
    switch (st) {   // st is a constant after unrolling: one case survives
        case 0: PARK_AGPR_STRIPE(0); break;
        case 1: PARK_AGPR_STRIPE(1); break;
        case 2: PARK_AGPR_STRIPE(2); break;
        case 3: PARK_AGPR_STRIPE(3); break;
        case 4: PARK_AGPR_STRIPE(4); break;
        case 5: PARK_AGPR_STRIPE(5); break;
        case 6: PARK_AGPR_STRIPE(6); break;
        default: PARK_AGPR_STRIPE(7); break;
    }
}
#undef PARK_AGPR_STRIPE
#undef PARK_AGPR_BLOCK

// HASIN (round 5): whether the tail reads a global input (accumulate / residual / aux) is a TEMPLATE parameter.  As a run-time flag the
// compiler had to keep the waits of the input prefetch on every path: `s_waitcnt vmcnt(7)` in front of each store of a tail that had
// issued no load at all -- and on gfx9 vmcnt counts stores, so a plain-store tail never had more than 7 stores per wave in flight
// (7 KB per ~1.5 us of write latency = the "9.3 bytes per cycle and CU" that bounded the bf16 store tail).
template <int EPI, bool CBF16, bool INBF16, int NS, bool HASIN, typename ACC>
__device__ __forceinline__ float tail_wave_full_t(const GemmArgs& g, float s, const ACC& acc, float* __restrict__ blk4 /* NS stripes */,
                                                  int64_t m_wave, int64_t n_wave, int lane, const void* ip, int64_t ild) {
    using T = TailShape<CBF16>;
    constexpr bool has_in = HASIN;
    static_assert(NS % 2 == 0, "the stripe loop handles two stripes per iteration (ping-pong input buffers)");
    float part = 0.f;
    // Two input buffers used alternately, NO register copies between them (round 5): the former `cur = nxt` hand-over at the end of every
    // stripe made the compiler wait for the just-issued loads -- s_waitcnt vmcnt(0) / vmcnt(1) once per stripe, i.e. a full drain of every
    // outstanding load AND store (vmcnt counts stores on gfx9): the residual / gate-backward tails paid one store round trip per stripe
    // (34-60 k cycles per tile).  Now a buffer is requested TWO stripes before it is consumed -- also on the loop's
    // entry edge, which is what the compiler's wait-count merge at the loop header takes -- so the first wait of an iteration leaves 15
    // younger operations in flight instead of draining the previous stripe's stores.
    uint4 bufA[T::NIT][2], bufB[T::NIT][2];
    if (has_in) {   // stripes 0 and 1 are requested up front: a buffer is always TWO stripes ahead of its use (see the wait counts above)
        tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave, n_wave, lane, bufA);
        tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave, n_wave + 64, lane, bufB);
    }
#pragma unroll
    for (int half = 0; half < 8 / NS; ++half) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int st = half * NS + q;
            park_stripe(blk4 + q * (32 * EPI_LD), acc, st, lane);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int q = 0; q < NS; q += 2) {
            const int st = half * NS + q;
            part += tail_stripe_full<EPI, CBF16, INBF16>(g, s, blk4 + q * (32 * EPI_LD), m_wave + (st >> 1) * 32, n_wave + (st & 1) * 64, lane, bufA,
                                                         has_in);
            if (has_in && st + 2 < 8)
                tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave + ((st + 2) >> 1) * 32, n_wave + ((st + 2) & 1) * 64, lane, bufA);
            part += tail_stripe_full<EPI, CBF16, INBF16>(g, s, blk4 + (q + 1) * (32 * EPI_LD), m_wave + ((st + 1) >> 1) * 32, n_wave + ((st + 1) & 1) * 64,
                                                         lane, bufB, has_in);
            if (has_in && st + 3 < 8)
                tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave + ((st + 3) >> 1) * 32, n_wave + ((st + 3) & 1) * 64, lane, bufB);
        }
        __builtin_amdgcn_wave_barrier();
    }
    return part;
}

template <int EPI, bool CBF16, int NS = TAIL_STRIPES, typename ACC>
__device__ __forceinline__ float tail_wave_full(const GemmArgs& g, float s, const ACC& acc, float* __restrict__ blk4, int64_t m_wave,
                                                int64_t n_wave, int lane) {
    const void* ip; int64_t ild; int idt;
    const bool has_in = tail_input<EPI>(g, ip, ild, idt);
    if constexpr (EPI == OTTER_EPI_GELU) {       // no global input at all: one instantiation
        return tail_wave_full_t<EPI, CBF16, true, NS, false>(g, s, acc, blk4, m_wave, n_wave, lane, ip, ild);
    } else if constexpr (EPI == OTTER_EPI_STORE) {   // input only when accumulating into C
        if (!has_in) return tail_wave_full_t<EPI, CBF16, true, NS, false>(g, s, acc, blk4, m_wave, n_wave, lane, ip, ild);
        if (idt == OTTER_BF16) return tail_wave_full_t<EPI, CBF16, true, NS, true>(g, s, acc, blk4, m_wave, n_wave, lane, ip, ild);
        return tail_wave_full_t<EPI, CBF16, false, NS, true>(g, s, acc, blk4, m_wave, n_wave, lane, ip, ild);
    } else {                                       // residual / aux kinds always read their input
        if (idt == OTTER_BF16) return tail_wave_full_t<EPI, CBF16, true, NS, true>(g, s, acc, blk4, m_wave, n_wave, lane, ip, ild);
        return tail_wave_full_t<EPI, CBF16, false, NS, true>(g, s, acc, blk4, m_wave, n_wave, lane, ip, ild);
    }
}

// ---- tail of the cross-tile form (round 6): ONE 8 KB parking stripe per wave, above the ring ----
// The ring (2 x 64 KB) is busy during this tail -- the next tile's first two K-tiles are landing in it -- so a wave parks in the 32 KB of LDS
// above it: one stripe of 32 rows x 64 fp32 = 8192 B per wave, unpadded (256-byte rows = one full sweep of the 64 banks), 16-byte chunk c
// of row r stored at chunk c ^ (r & 15): the eight lanes of a ds_write_b128 group (8 rows, one chunk) and the sixteen of a ds_read_b128 group hit
// distinct bank quads.  Single-buffered: stripe s + 1 is parked right behind the read-back of stripe s (a wave's LDS operations execute in order,
// so the reads see the old stripe) and lands under the arithmetic and the global accesses of stripe s.
constexpr int XPARK_BYTES = 32 * 256;   // per wave
__device__ __forceinline__ void xpark_stripe(unsigned base /* LDS byte address of the wave's stripe */, int st, int lane) {
    const unsigned r = (unsigned)lane & 15u, gq = (unsigned)lane >> 4;
    const unsigned row = base + r * 256u;
    const unsigned a0 = row + (((0u + gq) ^ r) << 4), a1 = row + (((4u + gq) ^ r) << 4), a2 = row + (((8u + gq) ^ r) << 4), a3 = row + (((12u + gq) ^ r) << 4);
#define XPARK_BLOCK(ADDR_, A_, MI_, NI_)                                                                                               \
    asm volatile("ds_write_b128 %0, a[%c1:%c2] offset:%c3" : : "v"(ADDR_), "n"(((MI_) * 8 + (NI_)) * 4), "n"(((MI_) * 8 + (NI_)) * 4 + 3), \
                 "n"((A_) * 4096) : "memory")
#define XPARK_STRIPE(ST_)                                                                                                              \
    do {                                                                                                                              \
        XPARK_BLOCK(a0, 0, 2 * ((ST_) >> 1), 4 * ((ST_) & 1)); XPARK_BLOCK(a1, 0, 2 * ((ST_) >> 1), 4 * ((ST_) & 1) + 1);                \
        XPARK_BLOCK(a2, 0, 2 * ((ST_) >> 1), 4 * ((ST_) & 1) + 2); XPARK_BLOCK(a3, 0, 2 * ((ST_) >> 1), 4 * ((ST_) & 1) + 3);            \
        XPARK_BLOCK(a0, 1, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1)); XPARK_BLOCK(a1, 1, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1) + 1);        \
        XPARK_BLOCK(a2, 1, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1) + 2); XPARK_BLOCK(a3, 1, 2 * ((ST_) >> 1) + 1, 4 * ((ST_) & 1) + 3);    \
    } while (0)
    switch (st) {   // run-time stripe index (a scalar branch tree): the AGPR block numbers are immediates
        case 0: XPARK_STRIPE(0); break;
        case 1: XPARK_STRIPE(1); break;
        case 2: XPARK_STRIPE(2); break;
        case 3: XPARK_STRIPE(3); break;
        case 4: XPARK_STRIPE(4); break;
        case 5: XPARK_STRIPE(5); break;
        case 6: XPARK_STRIPE(6); break;
        default: XPARK_STRIPE(7); break;
    }
#undef XPARK_STRIPE
#undef XPARK_BLOCK
}
// read the parked stripe back row-major (the lane -> (row, columns) map of TailShape), through the chunk swizzle
template <bool CBF16>
__device__ __forceinline__ void xpark_read(const char* __restrict__ stripe, int lane, float (&v)[TailShape<CBF16>::NIT][TailShape<CBF16>::NE]) {
    using T = TailShape<CBF16>;
#pragma unroll
    for (int it = 0; it < T::NIT; ++it) {
        const int row = T::row(lane, it), c0 = T::col(lane) >> 2;
        const char* src = stripe + row * 256 + ((c0 ^ (row & 15)) << 4);
        const float4 t0 = *reinterpret_cast<const float4*>(src);
        v[it][0] = t0.x; v[it][1] = t0.y; v[it][2] = t0.z; v[it][3] = t0.w;
        if constexpr (CBF16) {   // c0 is even: chunk c0 + 1 lies at the address with bit 4 flipped
            const float4 t1 = *reinterpret_cast<const float4*>(stripe + row * 256 + (((c0 ^ (row & 15)) << 4) ^ 16));
            v[it][4] = t1.x; v[it][5] = t1.y; v[it][6] = t1.z; v[it][7] = t1.w;
        }
    }
}
template <int EPI, bool CBF16, bool INBF16>
__device__ __forceinline__ float xtail_stripe_apply(GemmArgs g, float s, const float (&v)[TailShape<CBF16>::NIT][TailShape<CBF16>::NE], int64_t m_base,
                                                    int64_t n_base, int lane, const uint4 (&raw)[TailShape<CBF16>::NIT][2], bool has_in) {
    using T = TailShape<CBF16>;
    float part = 0.f;
    g.cdt = CBF16 ? OTTER_BF16 : OTTER_F32;
#pragma unroll
    for (int it = 0; it < T::NIT; ++it) {
        float a[T::NE];
        if (has_in) cvt_raw<T::NE, INBF16>(raw[it], a);
        else {
#pragma unroll
            for (int i = 0; i < T::NE; ++i) a[i] = 0.f;
        }
        part += tail_apply<EPI, T::NE>(g, s, m_base + T::row(lane, it), n_base + T::col(lane), v[it], a, has_in);
    }
    return part;
}
template <int EPI, bool CBF16, bool INBF16, bool HASIN>
__device__ __forceinline__ float xtail_wave_full_t(const GemmArgs& g, float s, char* __restrict__ stripe, int64_t m_wave, int64_t n_wave, int lane,
                                                   const void* ip, int64_t ild) {
    using T = TailShape<CBF16>;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)stripe;
    float part = 0.f;
    uint4 bufA[T::NIT][2], bufB[T::NIT][2];   // input stripes, requested two stripes ahead of their use (see tail_wave_full_t)
    if (HASIN) {
        tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave, n_wave, lane, bufA);
        tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave, n_wave + 64, lane, bufB);
    }
    xpark_stripe(base, 0, lane);
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {   // stripes 2 q (bufA) and 2 q + 1 (bufB)
        float v[T::NIT][T::NE];
        __builtin_amdgcn_wave_barrier();
        xpark_read<CBF16>(stripe, lane, v);
        xpark_stripe(base, 2 * q + 1, lane);
        part += xtail_stripe_apply<EPI, CBF16, INBF16>(g, s, v, m_wave + q * 32, n_wave, lane, bufA, HASIN);
        if (HASIN && q < 3) tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave + (q + 1) * 32, n_wave, lane, bufA);
        __builtin_amdgcn_wave_barrier();
        xpark_read<CBF16>(stripe, lane, v);
        if (q < 3) xpark_stripe(base, 2 * q + 2, lane);
        part += xtail_stripe_apply<EPI, CBF16, INBF16>(g, s, v, m_wave + q * 32, n_wave + 64, lane, bufB, HASIN);
        if (HASIN && q < 3) tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave + (q + 1) * 32, n_wave + 64, lane, bufB);
    }
    return part;
}
template <int EPI, bool CBF16>
__device__ __forceinline__ float xtail_wave_full(const GemmArgs& g, float s, char* __restrict__ stripe, int64_t m_wave, int64_t n_wave, int lane) {
    const void* ip; int64_t ild; int idt;
    const bool has_in = tail_input<EPI>(g, ip, ild, idt);
    if constexpr (EPI == OTTER_EPI_GELU) {
        return xtail_wave_full_t<EPI, CBF16, true, false>(g, s, stripe, m_wave, n_wave, lane, ip, ild);
    } else if constexpr (EPI == OTTER_EPI_STORE) {
        if (!has_in) return xtail_wave_full_t<EPI, CBF16, true, false>(g, s, stripe, m_wave, n_wave, lane, ip, ild);
        if (idt == OTTER_BF16) return xtail_wave_full_t<EPI, CBF16, true, true>(g, s, stripe, m_wave, n_wave, lane, ip, ild);
        return xtail_wave_full_t<EPI, CBF16, false, true>(g, s, stripe, m_wave, n_wave, lane, ip, ild);
    } else {
        if (idt == OTTER_BF16) return xtail_wave_full_t<EPI, CBF16, true, true>(g, s, stripe, m_wave, n_wave, lane, ip, ild);
        return xtail_wave_full_t<EPI, CBF16, false, true>(g, s, stripe, m_wave, n_wave, lane, ip, ild);
    }
}

__device__ __forceinline__ void tile_of_block(const GemmArgs& g, int bid, int& tile_m, int& tile_n) {
    // bijective XCD-chunked order (blocks are dispatched round-robin over the 8 XCDs: block b -> XCD b % 8)
    const int nwg = g.gm * g.gn;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // super-tile shape: 2^lm (M) x 2^(5-lm) (N) tiles, default 8 x 4; g.order (otter_gemm_set_debug bits 9-12, diagnostics) overrides
    // it: bits 0-2 = lm + 1 (0 = default), bit 3 = walk the super-tiles N-major instead of M-major
    const int lm = (g.order & 7) ? (g.order & 7) - 1 : 3;
    const int smt = 1 << lm, snt = 32 >> lm;
    if ((g.gm & (smt - 1)) == 0 && (g.gn & (snt - 1)) == 0) {
        // super-tiles of 8 (M) x 4 (N) tiles: the ~32 blocks an XCD has resident at any time form a compact patch, so its
        // L2 holds 8 A-slabs + 4 B-slabs per K-step (384 KB) instead of 16 + 2 (576 KB) for the plain M-sweep
        const int st = swz >> 5, w = swz & 31, gms = g.gm >> lm, gns = g.gn / snt;
        if (g.order & 8) {
            tile_n = (st % gns) * snt + (w >> lm);
            tile_m = (st / gns) * smt + (w & (smt - 1));
        } else {
            tile_m = (st % gms) * smt + (w & (smt - 1));
            tile_n = (st / gms) * snt + (w >> lm);
        }
    } else {
        tile_m = swz % g.gm;
        tile_n = swz / g.gm;
    }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool GLDS, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_kernel(GemmArgs g) {
    static_assert(BN / WN == 64, "the epilogue stripe is 64 columns wide");
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int A_CH = BM * 8 / NT, B_CH = BN * 8 / NT;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const bf16_t* __restrict__ A = (const bf16_t*)g.A;
    const bf16_t* __restrict__ B = (const bf16_t*)g.B;
    const int64_t K = g.K;
    const int nk = (int)((K + 63) >> 6);
    const float s = g.gate ? tanhf(*g.gate) : 1.0f;
    // Persistent over output tiles: a block keeps its CU and walks virtual block ids vb = blockIdx.x + i*gridDim.x
    // (gridDim.x is a multiple of 8, so vb stays on this block's XCD chunk).  The epilogue stores of tile i drain while
    // the K loop of tile i+1 is already running -- with one block per tile a CU sat idle until its 128 KB of C had been
    // acknowledged (the ablation showed the stores costing 4x their stand-alone time).
    const int ntiles = g.gm * g.gn;
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    int tile_m, tile_n;
    tile_of_block(g, vb, tile_m, tile_n);
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;

    // per-thread chunk coordinates (constant over the K loop)
    const bf16_t* pa[A_CH];
    const bf16_t* pb[B_CH];
    int la[A_CH], lb[B_CH], ka[A_CH], kb[B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = i * NT + tid, row = c >> 3, phys = c & 7;
        const int slot = (GLDS && !(g.dbg & 64)) ? (phys ^ ((row >> 1) & 7)) : phys;        // logical k-slot this lane fetches (dbg 64: linear, wrong results)
        const int dst = GLDS ? phys : (phys ^ ((row >> 1) & 7));        // physical slot it lands in
        int64_t gr = m0 + row;
        if (gr > g.M - 1) gr = g.M - 1;
        pa[i] = A + gr * g.lda + slot * 8;
        ka[i] = slot * 8;
        la[i] = row * 128 + dst * 16;
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int c = i * NT + tid, row = c >> 3, phys = c & 7;
        const int slot = (GLDS && !(g.dbg & 64)) ? (phys ^ ((row >> 1) & 7)) : phys;
        const int dst = GLDS ? phys : (phys ^ ((row >> 1) & 7));
        int64_t gr = n0 + row;
        if (gr > g.N - 1) gr = g.N - 1;
        pb[i] = B + gr * g.ldb + slot * 8;
        kb[i] = slot * 8;
        lb[i] = BM * 128 + row * 128 + dst * 16;
    }

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // (A per-tile rotation of the K walk -- meant to spread the power-of-two operand strides over L2/HBM channels -- was
    // measured 10 % SLOWER in interleaved A/B rounds: tiles that share an A or B panel hit the same lines at the same time
    // when they walk k in lock step, and that temporal L2 sharing is worth more than the channel spread.  Not used.)
    auto krot = [&](int t) { return t; };

    auto compute = [&](int buf) {
        const char* At = smem + buf * TILE_BYTES;
        const char* Bt = At + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = 2 * ks + (lane >> 5);
            bf16x8_t fa[NI], fb[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int row = wn * TN + ni * 32 + (lane & 31);
                fa[ni] = *reinterpret_cast<const bf16x8_t*>(Bt + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int row = wm * TM + mi * 32 + (lane & 31);
                fb[mi] = *reinterpret_cast<const bf16x8_t*>(At + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ni], fb[mi], acc[mi][ni], 0, 0, 0);
        }
    };

    if constexpr (GLDS) {
        // direct global -> LDS (wave-uniform LDS base + lane*16); requires K % 64 == 0 (checked on the host)
        auto stage = [&](int buf, int kt) {
            const int64_t koff = (int64_t)kt * 64;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int wbase = buf * TILE_BYTES + (i * NT + wave * 64) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[i] + koff),
                                                 (__attribute__((address_space(3))) void*)(smem + wbase), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                const int wbase = buf * TILE_BYTES + BM * 128 + (i * NT + wave * 64) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb[i] + koff),
                                                 (__attribute__((address_space(3))) void*)(smem + wbase), 16, 0, 0);
            }
        };
        stage(0, krot(0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nk_run = (g.dbg & 8) ? 0 : nk;
        for (int t = 0; t < nk_run; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk && !(g.dbg & 1)) stage(cur ^ 1, krot(t + 1));
            if (!(g.dbg & 2)) compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        uint4 ra[A_CH], rb[B_CH];
        auto gload = [&](int kt) {
            const int64_t koff = (int64_t)kt * 64;
#pragma unroll
            for (int i = 0; i < A_CH; ++i)
                ra[i] = (koff + ka[i] < K) ? *reinterpret_cast<const uint4*>(pa[i] + koff) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < B_CH; ++i)
                rb[i] = (koff + kb[i] < K) ? *reinterpret_cast<const uint4*>(pb[i] + koff) : make_uint4(0, 0, 0, 0);
        };
        auto lstore = [&](int buf) {
            char* base = smem + buf * TILE_BYTES;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) *reinterpret_cast<uint4*>(base + la[i]) = ra[i];
#pragma unroll
            for (int i = 0; i < B_CH; ++i) *reinterpret_cast<uint4*>(base + lb[i]) = rb[i];
        };
        gload(krot(0));
        lstore(0);
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk && !(g.dbg & 1)) gload(krot(t + 1));   // in flight under the MFMAs below
            if (!(g.dbg & 2)) compute(cur);
            if (t + 1 < nk) lstore(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: accumulators -> wave-private LDS block -> row-major read-back + fused tail ----
    // (the last K-loop barrier has already retired every read of the operand tiles this aliases)
    float part = 0.f;
    float* blk = reinterpret_cast<float*>(smem) + wave * (32 * EPI_LD);
    if (!(g.dbg & 4)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) park_block(blk, acc[mi][ni], lane, ni * 32);
        // the stripe is wave-private and a wave's LDS instructions execute in program order: no workgroup barrier needed,
        // the 8 waves run their epilogues independently (wave_barrier only pins the compiler's instruction order)
        __builtin_amdgcn_wave_barrier();
        part += epilogue_stripe<EPI>(g, s, blk, m0 + wm * TM + mi * 32, n0 + wn * TN, lane);
        __builtin_amdgcn_wave_barrier();
    }
    }  // !(dbg & 4)
    if ((g.dbg & 16) && part == 12345.678f) reinterpret_cast<float*>(g.C)[0] = part;
    block_partial<WM * WN, EPI>(g, part, reinterpret_cast<float*>(smem), vb);
    __syncthreads();  // every wave is done with the LDS it parked in before the next tile's staging overwrites it
    }  // persistent tile loop
}

#ifdef OTTER_EXPERIMENTAL  // tools-only: variant 8: wave-specialised kernel (8 MFMA waves + 4 loader waves)
#include "experimental/gemm_ws_variant8.inc"
#endif

// ------------------------------------------------------------------------------------------------------------
// bf16 phased kernel (variant 6): the 256x256x64 / 8-wave / LDS-DMA kernel above with the K-tile split into four
// quadrant phases and the two wave rows STAGGERED by one barrier (guide section 5, "8-phase" idea):
//   wave (wr, wc) owns rows wr*128.. x cols wc*64..; phase (mh, nh) = its 64x32 quadrant x the whole BK = 8 MFMAs.
//   per phase:  LOAD part (ds_read the quadrant's new fragments [+ issue the next tile's DMA in phase 0])
//               | s_barrier | lgkmcnt(0) | setprio(1) 8x MFMA setprio(0) | s_barrier
//   waves 4-7 (wr = 1) execute one extra s_barrier up front, so at every moment one wave row sits in its MFMA cluster
//   while the other sits in its LOAD part.  A SIMD hosts wave s (wr 0) and wave s+4 (wr 1): its matrix pipe is fed by
//   the two alternately and LDS reads / DMA issue of one always run under the MFMAs of the other.
// Hazards (barrier instance k of row 0 pairs with instance k+1 of row 1):
//   RAW  DMA(t+1) -> ds_read: every wave waits vmcnt(0) in the LOAD part of phase 3, i.e. before a barrier that both its
//        own row and (one instance later) the other row pass before their first read of tile t+1.
//   WAR  ds_read(t-1) -> DMA(t+1) into the same buffer: phase 3 reads nothing new (the nh=0 B fragments are kept in
//        registers) and phase 2's reads are retired by lgkmcnt(0) before its MFMAs, so when any wave issues the DMA in
//        phase 0 of tile t every read of tile t-1 has completed, on both rows.
// ------------------------------------------------------------------------------------------------------------
template <int EPI, int SCH, bool BUF>
__global__ __launch_bounds__(512) void gemm_bf16_ph_kernel(GemmArgs g) {
    constexpr bool CNT = SCH >= 1;
    constexpr int BM = 256, BN = 256, NT = 512;
    constexpr int TILE_BYTES = (BM + BN) * 128;
    constexpr int CH = 4;  // 16-B chunks per thread per operand per K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const bf16_t* __restrict__ A = (const bf16_t*)g.A;
    const bf16_t* __restrict__ B = (const bf16_t*)g.B;
    const int nk = (int)(g.K >> 6);
    const float sgate = g.gate ? tanhf(*g.gate) : 1.0f;
    // BUF: buffer addressing for the DMA (SGPR resource descriptor + 32-bit per-lane offset + scalar K offset): no
    // per-piece 64-bit VALU address arithmetic, and the hardware range check clamps reads past the end of the operand
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
    if constexpr (BUF) {
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)(uint32_t)((g.M - 1) * g.lda * 2 + g.K * 2), 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)(uint32_t)((g.N - 1) * g.ldb * 2 + g.K * 2), 0x00020000);
    }
    const int ntiles = g.gm * g.gn;
#define RAW_BARRIER()                       \
    do {                                    \
        __builtin_amdgcn_sched_barrier(0);  \
        __builtin_amdgcn_s_barrier();       \
        __builtin_amdgcn_sched_barrier(0);  \
    } while (0)
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        int tile_m, tile_n;
        tile_of_block(g, vb, tile_m, tile_n);
        const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
        // global side: 32-bit per-lane byte offsets from a wave-uniform base (SGPR base + VGPR offset addressing: half the
        // address registers of 64-bit pointers and no per-tile 64-bit VALU adds; the host guarantees the operands span < 4 GB)
        uint32_t oa[CH], ob[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = i * NT + tid, row = c >> 3, phys = c & 7;
            const int slot = phys ^ ((row >> 1) & 7);
            int64_t ga = m0 + row; if (ga > g.M - 1) ga = g.M - 1;
            int64_t gb = n0 + row; if (gb > g.N - 1) gb = g.N - 1;
            oa[i] = (uint32_t)((ga * g.lda + slot * 8) * 2);
            ob[i] = (uint32_t)((gb * g.ldb + slot * 8) * 2);
        }
        auto stage = [&](int buf, int kt) {
            if constexpr ((OTTER_DIAG & 1) != 0) return;
            const char* abase = reinterpret_cast<const char*>(A) + (size_t)kt * 128;
            const char* bbase = reinterpret_cast<const char*>(B) + (size_t)kt * 128;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int wbase = buf * TILE_BYTES + (i * NT + wave * 64) * 16;
                if constexpr (BUF)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(smem + wbase), 16,
                                                             (int)oa[i], kt * 128, 0, 0);
                else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + oa[i]),
                                                 (__attribute__((address_space(3))) void*)(smem + wbase), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int wbase = buf * TILE_BYTES + BM * 128 + (i * NT + wave * 64) * 16;
                if constexpr (BUF)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(smem + wbase), 16,
                                                             (int)ob[i], kt * 128, 0, 0);
                else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bbase + ob[i]),
                                                 (__attribute__((address_space(3))) void*)(smem + wbase), 16, 0, 0);
            }
        };
        // CNT schedule: the K-tile is fed as four half-tiles of 128 rows -- A_h(mh) = rows {wr*128 + mh*64 + 0..63 : wr},
        // B_h(nh) = rows {wc*64 + nh*32 + 0..31 : wc} -- i.e. exactly the rows ONE quadrant phase reads.  Two DMA pieces per
        // thread per half-tile; they are issued one or two per phase and retired with COUNTED vmcnt (never drained in
        // steady state): A_h0,B_h0(t+1) in phase 0, B_h1(t+1) in phase 1, A_h1(t+1) in phase 2 -> every half-tile has three
        // phases to land and the wait in phase q-1 for the rows phase q reads leaves the 2-3 newest half-tiles in flight.
        uint32_t oah[2][2], obh[2][2];
        if constexpr (CNT) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = i * NT + tid, rl = c >> 3, phys = c & 7;
                    const int ra = (rl >> 6) * 128 + h * 64 + (rl & 63);
                    const int rb = (rl >> 5) * 64 + h * 32 + (rl & 31);
                    int64_t ga = m0 + ra; if (ga > g.M - 1) ga = g.M - 1;
                    int64_t gb = n0 + rb; if (gb > g.N - 1) gb = g.N - 1;
                    oah[h][i] = (uint32_t)((ga * g.lda + (phys ^ ((ra >> 1) & 7)) * 8) * 2);
                    obh[h][i] = (uint32_t)((gb * g.ldb + (phys ^ ((rb >> 1) & 7)) * 8) * 2);
                }
        }
        auto stage_ah = [&](int buf, int kt, int h, int i0 = 0, int i1 = 2) {
            if constexpr ((OTTER_DIAG & 1) != 0) return;
            const char* abase = reinterpret_cast<const char*>(A) + (size_t)kt * 128;
#pragma unroll
            for (int i = i0; i < ((OTTER_DIAG & 64) ? (i1 < 1 ? i1 : 1) : i1); ++i) {  // diag 64: half the DMA pieces
                const int wbase = buf * TILE_BYTES + (i * 128 + h * 64 + wave * 8) * 128;
                if constexpr (BUF)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(smem + wbase), 16,
                                                             (int)oah[h][i], kt * 128, 0, 0);
                else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + oah[h][i]),
                                                 (__attribute__((address_space(3))) void*)(smem + wbase), 16, 0, 0);
            }
        };
        auto stage_bh = [&](int buf, int kt, int h, int i0 = 0, int i1 = 2) {
            if constexpr ((OTTER_DIAG & 1) != 0) return;
            const char* bbase = reinterpret_cast<const char*>(B) + (size_t)kt * 128;
#pragma unroll
            for (int i = i0; i < ((OTTER_DIAG & 64) ? (i1 < 1 ? i1 : 1) : i1); ++i) {
                const int wbase = buf * TILE_BYTES + BM * 128 + ((2 * i + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8) * 128;
                if constexpr (BUF)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(smem + wbase), 16,
                                                             (int)obh[h][i], kt * 128, 0, 0);
                else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bbase + obh[h][i]),
                                                 (__attribute__((address_space(3))) void*)(smem + wbase), 16, 0, 0);
            }
        };
        // LDS side: the swizzle term (row>>1)&7 only depends on lane&31 (every row offset used below is a multiple of 32),
        // so ONE base per k-step and operand (8 VGPRs) + compile-time immediates reach every fragment of the wave.
        const int swz = ((lane & 31) >> 1) & 7;
        int la[4], lb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int slot = (2 * ks + (lane >> 5)) ^ swz;
            la[ks] = (wr * 128 + (lane & 31)) * 128 + (slot << 4);
            lb[ks] = BM * 128 + (wc * 64 + (lane & 31)) * 128 + (slot << 4);
        }
        auto ld_a = [&](int buf, int mh, bf16x8_t (&fa)[2][4]) {
            if constexpr ((OTTER_DIAG & 8) != 0) return;
            const char* base = smem + buf * TILE_BYTES + mh * (64 * 128);
#pragma unroll
            for (int mi2 = 0; mi2 < ((OTTER_DIAG & 32) ? 1 : 2); ++mi2)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    fa[mi2][ks] = *reinterpret_cast<const bf16x8_t*>(base + la[ks] + mi2 * (32 * 128));
            if constexpr ((OTTER_DIAG & 32) != 0) {  // half the A fragment reads
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[1][ks] = fa[0][ks];
            }
        };
        auto ld_b = [&](int buf, int nh, bf16x8_t (&fb)[4]) {
            if constexpr ((OTTER_DIAG & 8) != 0) return;
            const char* base = smem + buf * TILE_BYTES + nh * (32 * 128);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const bf16x8_t*>(base + lb[ks]);
        };
        f32x16_t acc[4][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        // one quadrant: 8 MFMAs, operands swapped (a = B rows, b = A rows)
#define QUAD(MH, NH, FA, FB)                                                                                            \
    do {                                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < ((OTTER_DIAG & 2) ? 0 : 4); ++ks) {                                    \
            acc[(MH)*2 + 0][NH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[ks], FA[0][ks], acc[(MH)*2 + 0][NH], 0, 0, 0); \
            acc[(MH)*2 + 1][NH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[ks], FA[1][ks], acc[(MH)*2 + 1][NH], 0, 0, 0); \
        }                                                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
    } while (0)

        if constexpr (CNT) {
            stage_ah(0, 0, 0); stage_bh(0, 0, 0); stage_bh(0, 0, 1); stage_ah(0, 0, 1);
        } else {
            stage(0, 0);
        }
        if constexpr (SCH == 2 || SCH == 4) {
            if (nk > 1) { stage_ah(1, 1, 0); stage_bh(1, 1, 0); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (SCH == 6) {
            if (nk > 1) {
                stage_ah(1, 1, 0); stage_bh(1, 1, 0); stage_bh(1, 1, 1); stage_ah(1, 1, 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (wr == 1) RAW_BARRIER();  // stagger the second wave row by one barrier
        bf16x8_t fa[2][4], fb[4];
        [[maybe_unused]] bf16x8_t fb1[4];
        if constexpr ((OTTER_DIAG & 8) != 0) {  // fragments without LDS reads: lane-dependent, non-zero
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                fb[ks] = __builtin_bit_cast(bf16x8_t, uint4{0x3f803f80u + (unsigned)lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u});
                fa[0][ks] = fa[1][ks] = fb[ks];
                fb1[ks] = fb[ks];
            }
        }
        if constexpr (SCH == 6) {
            // Deep-queue schedule: every half-tile slot is refilled ONE phase after its only read, with the data of tile
            // t+2 (same buffer), instead of one K-tile later with tile t+1's -- 10-12 DMA pieces per wave in flight
            // instead of 4-6 (the L2->LDS stream is latency x concurrency bound: DESIGN.md 4.1).
            //   reads  : phase 0: A_h0,B_h0   1: B_h1   2: A_h1   3: -          (B fragments stay in fb / fb1)
            //   refill : phase 1: A_h0(t+2)   2: B_h0(t+2)   3: B_h1(t+2), A_h1(t+2)   (ds_read / DMA = 12/0 4/2 8/2 0/4)
            // WAR: fragment reads are retired (lgkmcnt(0)) BEFORE the first barrier of their phase; the other row passes
            // the matching barrier before its next LOAD part, so a refill issued one phase later can never overtake a read.
            // RAW: counted vmcnt in the phase before the read, before that phase's first barrier (as in the other schedules).
            for (int t = 0; t < nk; ++t) {
                const int cur = t & 1;
                const int e = nk - 1 - t;  // K-tiles after this one
                // ---- phase 0: quadrant (0,0) ----
                ld_a(cur, 0, fa);
                ld_b(cur, 0, fb);
                if (e >= 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 0, fa, fb);
                RAW_BARRIER();
                // ---- phase 1: quadrant (0,1) ----
                ld_b(cur, 1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                if (e >= 2) { stage_ah(cur, t + 2, 0); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
                else if (e == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 1, fa, fb1);
                RAW_BARRIER();
                // ---- phase 2: quadrant (1,1) ----
                ld_a(cur, 1, fa);
                __builtin_amdgcn_sched_barrier(0);
                if (e >= 2) stage_bh(cur, t + 2, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(1, 1, fa, fb1);
                RAW_BARRIER();
                // ---- phase 3: quadrant (1,0) ----
                if (e >= 2) { stage_bh(cur, t + 2, 1); stage_ah(cur, t + 2, 1); asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
                else if (e == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                RAW_BARRIER();
                QUAD(1, 0, fa, fb);
                RAW_BARRIER();
            }
        } else if constexpr (SCH == 5) {
            // Two phases per K-tile (16 MFMAs per cluster, 4 workgroup barriers per K-tile instead of 8):
            //   phase A: reads A0,B0,B1 (16) + 6 DMA pieces [A_h0,B_h0,B_h1](t+1) | quadrants (0,0),(0,1)
            //   phase B: reads A1 (8)       + 2 DMA pieces  A_h1(t+1)            | quadrants (1,1),(1,0)
            // The fragment reads are retired (lgkmcnt(0)) BEFORE the first barrier of their phase, so once a row has
            // passed that barrier none of its reads of the buffer is pending: the other row may refill it right away.
            for (int t = 0; t < nk; ++t) {
                const int cur = t & 1;
                const bool more = t + 1 < nk;
                ld_a(cur, 0, fa);
                ld_b(cur, 0, fb);
                ld_b(cur, 1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { stage_ah(cur ^ 1, t + 1, 0); stage_bh(cur ^ 1, t + 1, 0); stage_bh(cur ^ 1, t + 1, 1); }
                // A_h1(t) (read in phase B); newer: the 6 pieces just issued
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 0, fa, fb);
                QUAD(0, 1, fa, fb1);
                RAW_BARRIER();
                ld_a(cur, 1, fa);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { stage_ah(cur ^ 1, t + 1, 1); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(1, 1, fa, fb1);
                QUAD(1, 0, fa, fb);
                RAW_BARRIER();
            }
        } else if constexpr (SCH == 3) {
            // LOAD-part-only balanced schedule: fb/fb1 resident (phase 3 reads nothing), two DMA pieces per phase:
            //   phase 0: A_h0(t+1), 1: B_h0(t+1), 2: B_h1(t+1), 3: A_h1(t+1); reads/pieces = 12/2, 4/2, 8/2, 0/2.
            for (int t = 0; t < nk; ++t) {
                const int cur = t & 1;
                const bool more = t + 1 < nk;
                ld_a(cur, 0, fa);
                ld_b(cur, 0, fb);
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_ah(cur ^ 1, t + 1, 0);
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 0, fa, fb);
                RAW_BARRIER();
                ld_b(cur, 1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_bh(cur ^ 1, t + 1, 0);
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 1, fa, fb1);
                RAW_BARRIER();
                ld_a(cur, 1, fa);
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_bh(cur ^ 1, t + 1, 1);
                RAW_BARRIER();
                QUAD(1, 1, fa, fb1);
                RAW_BARRIER();
                if (more) { stage_ah(cur ^ 1, t + 1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                RAW_BARRIER();
                QUAD(1, 0, fa, fb);
                RAW_BARRIER();
            }
        } else if constexpr (SCH == 2 || SCH == 4) {
            // Rebalanced counted schedule.  Both nh fragments of B stay in registers (fb, fb1), so phase 3 reads nothing
            // from LDS (24 instead of 28 ds_read_b128 per wave per K-tile, and the WAR argument in the header holds
            // literally); that read-free phase issues the 4 DMA pieces of A_h0,B_h0 of tile t+2 into the buffer tile t
            // vacated one phase earlier, phase 0 (12 reads) issues none, phases 1 / 2 issue B_h1 / A_h1 of tile t+1:
            //   ds_read / DMA pieces per phase = 12/0, 4/2, 8/2, 0/4; every half-tile still has 4 phases to land.
            // In each LOAD part the ds_reads go first and the DMA pieces last, so a wave that queues behind its row's
            // other waves at the texture-address unit does so with its LDS reads already in flight.
            for (int t = 0; t < nk; ++t) {
                const int cur = t & 1;
                const bool more = t + 1 < nk, more2 = t + 2 < nk;
                // ---- phase 0: quadrant (0,0) ----
                ld_a(cur, 0, fa);
                ld_b(cur, 0, fb);
                // B_h1(t) (read in phase 1) must have landed; newer: A_h1(t), [A_h0,B_h0](t+1)
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 0, fa, fb);
                RAW_BARRIER();
                // ---- phase 1: quadrant (0,1) ----
                ld_b(cur, 1, fb1);
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_bh(cur ^ 1, t + 1, 1);
                // A_h1(t) (phase 2); newer: [A_h0,B_h0](t+1), B_h1(t+1)
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 1, fa, fb1);
                RAW_BARRIER();
                // ---- phase 2: quadrant (1,1) ----
                ld_a(cur, 1, fa);
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_ah(cur ^ 1, t + 1, 1);
                RAW_BARRIER();
                QUAD(1, 1, fa, fb1);
                RAW_BARRIER();
                // ---- phase 3: quadrant (1,0), no LDS reads; first half of tile t+2 into the buffer of tile t ----
                // (every read of tile t was retired by the lgkmcnt(0) of phase 2's cluster, one barrier ago on this row,
                //  and the other row's issue point is one barrier later still)
                RAW_BARRIER();
                if constexpr (SCH == 2) {
                    if (more2) { stage_ah(cur, t + 2, 0); stage_bh(cur, t + 2, 0); }
                    QUAD(1, 0, fa, fb);
                } else {  // SCH 4: one DMA piece in the shadow of each k-step's first MFMA
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int ks = 0; ks < ((OTTER_DIAG & 2) ? 0 : 4); ++ks) {
                        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], fa[0][ks], acc[2][0], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (more2) { if (ks < 2) stage_ah(cur, t + 2, 0, ks, ks + 1); else stage_bh(cur, t + 2, 0, ks - 2, ks - 1); }
                        __builtin_amdgcn_sched_barrier(0);
                        acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks], fa[1][ks], acc[3][0], 0, 0, 0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
                // [A_h0,B_h0](t+1) (next phase 0); newer: B_h1(t+1), A_h1(t+1), [A_h0,B_h0](t+2)
                if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                RAW_BARRIER();
            }
        } else if constexpr (CNT) {
            for (int t = 0; t < nk; ++t) {
                const int cur = t & 1;
                const bool more = t + 1 < nk;
                // ---- phase 0: quadrant (0,0) ----
                if (more) { stage_ah(cur ^ 1, t + 1, 0); stage_bh(cur ^ 1, t + 1, 0); }
                ld_a(cur, 0, fa);
                ld_b(cur, 0, fb);
                // rows read in phase 1 (B_h1 of this tile) must have landed before the barrier after the next cluster
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                RAW_BARRIER();
                QUAD(0, 0, fa, fb);
                RAW_BARRIER();
                // ---- phase 1: quadrant (0,1) ----
                if (more) stage_bh(cur ^ 1, t + 1, 1);
                ld_b(cur, 1, fb);
                if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A_h1 of this tile (phase 2)
                RAW_BARRIER();
                QUAD(0, 1, fa, fb);
                RAW_BARRIER();
                // ---- phase 2: quadrant (1,1) ----
                if (more) stage_ah(cur ^ 1, t + 1, 1);
                ld_a(cur, 1, fa);
                RAW_BARRIER();
                QUAD(1, 1, fa, fb);
                RAW_BARRIER();
                // ---- phase 3: quadrant (1,0) ----
                ld_b(cur, 0, fb);
                if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // A_h0, B_h0 of the next tile (its phase 0)
                RAW_BARRIER();
                QUAD(1, 0, fa, fb);
                RAW_BARRIER();
            }
        } else
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            // ---- phase 0: quadrant (0,0) ----
            ld_a(cur, 0, fa);
            ld_b(cur, 0, fb);
            RAW_BARRIER();
            QUAD(0, 0, fa, fb);
            // the next tile's DMA is issued AFTER this cluster: by now the other wave row has retired its last reads of
            // the buffer being refilled (its phase-3 reads complete before its own phase-3 MFMAs, one barrier ago)
            if (t + 1 < nk) stage(cur ^ 1, t + 1);
            RAW_BARRIER();
            // ---- phase 1: quadrant (0,1) ----
            ld_b(cur, 1, fb);
            RAW_BARRIER();
            QUAD(0, 1, fa, fb);
            RAW_BARRIER();
            // ---- phase 2: quadrant (1,1) ----
            ld_a(cur, 1, fa);
            RAW_BARRIER();
            QUAD(1, 1, fa, fb);
            RAW_BARRIER();
            // ---- phase 3: quadrant (1,0); retire this wave's share of the next tile's DMA before the barrier ----
            ld_b(cur, 0, fb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RAW_BARRIER();
            QUAD(1, 0, fa, fb);
            RAW_BARRIER();
        }
        if (wr == 0) RAW_BARRIER();  // undo the stagger
        __syncthreads();
#undef QUAD

        // ---- epilogue (same as the 8-wave kernel: 64-column wave-private stripes) ----
        float part = 0.f;
        float* blk = reinterpret_cast<float*>(smem) + wave * (32 * EPI_LD);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            if constexpr ((OTTER_DIAG & 16) != 0) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part += acc[mi][ni][r];
                continue;
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) park_block(blk, acc[mi][ni], lane, ni * 32);
            __builtin_amdgcn_wave_barrier();
            part += epilogue_stripe<EPI>(g, sgate, blk, m0 + wr * 128 + mi * 32, n0 + wc * 64, lane);
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr ((OTTER_DIAG & 20) != 0) {  // keep `part` (hence the accumulators) observable
            if (part == 12345.678f) reinterpret_cast<float*>(g.C)[threadIdx.x] = part;
        }
        block_partial<8, EPI>(g, part, reinterpret_cast<float*>(smem), vb);
        __syncthreads();
    }
#undef RAW_BARRIER
}

#ifdef OTTER_EXPERIMENTAL  // tools-only: variants 4/5/12 (multi-stage ring), 17, 18-23 (one wave per SIMD, register-resident K-tile on 32x32x16): the round-1/2 generations that led to variant 26
#include "experimental/gemm_generations_r1_r2.inc"
#endif

// ------------------------------------------------------------------------------------------------------------
// bf16 register-resident K-tile kernel on v_mfma_f32_16x16x32_bf16 (variant 26): variant 18's pipeline, LDS image and DMA
// unchanged, the matrix instruction swapped.  Why: tools/probe/mfma_power.hip -- back-to-back MFMAs, one wave per SIMD, 256
// accumulator registers, N(0, 0.05) bf16 operands -- sustains 1.81 PF with 32x32x16 (1.72 GHz under the power cap) and
// 2.07 PF with 16x16x32 (1.97 GHz); with zero operands both run at 2.47 PF (2.35 GHz).  The 16x16x32 shape reads and writes
// its accumulator half as often per flop (k = 32 per pass over a 4-register block against k = 16 over a 16-register block),
// and on real data the GEMM is clock-bound by power, not issue-bound: hipBLASLt's gfx950 kernels use MI16x16 as well.
// Wave tile 128x128 = 8 x 8 blocks of 16x16 (f32x4 each), K-tile = 2 k-steps x 64 MFMAs of 16 cycles; fragment of block i,
// k-step ks: row i*16 + (lane&15), 16-byte slot (4*ks + (lane>>4)) ^ ((row>>1)&7) -- 16 lanes of a quarter-wave hit 8 slots x 2
// row parities = 16 distinct bank groups with the same swizzle as the 32-row fragments.
// Schedule generated by tools/gen/gemm_t4_schedule.py.  Requires K % 128 == 0 and operands spanning < 4 GB.
// ------------------------------------------------------------------------------------------------------------
// K-major operands (TA / TB; round 3): an operand may be handed over as [K rows][M or N columns] row-major -- the layout every
// BACKWARD product of a Linear has one or both operands in (dW = dy^T x: both; dx = dy W with W as stored: B) -- instead of the
// K-contiguous rows of the forward.  Rounds 1-2 produced the K-contiguous form with a transpose kernel per operand (2.7 ms per
// step) and kept a transposed bf16 shadow of every trainable weight (1.2 ms per step to rebuild).  Here the K-tile of such an
// operand lands in LDS as it lies in HBM -- [64 k][256 m], 512-byte rows, each DMA instruction two whole rows = eight whole cache
// lines -- and the MFMA fragment (lane (r, g): row m = 16 i + r, k = 32 ks + 8 g .. + 7) is gathered by two
// ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 m] block, lane r supplying row (r >> 2), columns 4 (r & 3) .. + 3, and
// receiving column r of it.  Same registers, same MFMA operand order, same LDS bytes as the K-contiguous form; the schedule
// slot of a fragment read holds two LDS instructions and one v_xad_u32.  Bank conflicts: one pass of the instruction serves
// lanes 0-31 = 8 k-rows x 32 bytes, all at the same column window of 512-byte rows, i.e. the same 8 banks; the 32-byte column
// blocks are therefore XOR-swizzled within each 256-byte half row by f(k) = (k & 3) | ((k >> 3) & 1) << 2 (the 8 rows of a
// pass get 8 distinct blocks), applied on the DMA SOURCE offset as everywhere else.
// LDS-DMA issued from inline asm (K-major instantiations).  hipcc's wait-count pass cannot disambiguate the transpose-read intrinsic from a
// pending builtin LDS-DMA and puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 that follows a DMA issue -- i.e. it drains
// the two-K-tile prefetch twice per K-tile (measured: SQ_WAIT_ANY 23 M -> 126 M wave-cycles per launch, 382 -> 562 us).  Issued from asm
// the DMA is invisible to that pass; completion is counted by the schedule's own s_waitcnt vmcnt(N) + s_barrier (as in csrc/flash.hip).
__device__ __forceinline__ u32x4_t gemm_rsrc4(const void* p, uint32_t bytes) {
    const uint64_t pa = (uint64_t)p;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void gemm_dma16_asm(u32x4_t r, unsigned lds, uint32_t voff, uint32_t soff) {
    // M0 is clobbered, not restored: nothing else in a K-major instantiation uses it (every LDS-DMA of the kernel is this statement), and
    // with one wave per SIMD every instruction beside an MFMA is an issue slot -- the save / restore / settle form of csrc/flash.hip
    // (6 instructions, > 16 cycles) stalled the matrix pipe at each of the 16 pieces of a K-tile (measured: 449 vs 379 us).
#if (OTTER_T4_ABL & 16)   // ablation build: the scalar part of a piece only
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
#endif
    // (m0 cannot be named as a clobber -- it is a reserved register to hipcc, which never keeps a value in it across statements: every
    //  compiler-generated use is preceded by its own s_mov)
}

// Grouped form (round 6d, cross-tile K-contiguous instantiations): a wave's pieces of one operand are 1 KB apart in LDS, so M0 is written once per
// FOUR pieces and pieces 1-3 of a group ride on the instruction's immediate offset (1024 q) -- which the hardware adds to the LDS address AND to the
// global address: the per-lane source offset of piece q is stored 1024 q low.  tools/probe/dma_issue.hip: an `s_mov m0 + s_nop` pair beside
// back-to-back MFMAs costs ~12 cycles of matrix pipe; 12 of the 16 pairs of a K-tile go (profiles/r06d_dma_issue_probe.txt: +6 % MFMA rate on zeros,
// +3-4 % on random operands).  Nothing else in these kernels touches M0 (checked in the object code: 128 s_mov m0 = 128 LDS-DMA instructions).
#ifndef OTTER_T4_DMA_AUX   // cache-policy bits of the operand loads (A/B builds: " nt", " sc1", " sc0 sc1"); default: none
#define OTTER_T4_DMA_AUX ""
#endif
template <int Q>
__device__ __forceinline__ void gemm_dma16_asm_q(u32x4_t r, unsigned lds_group, uint32_t voff_low, uint32_t soff) {
    if constexpr (Q == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" OTTER_T4_DMA_AUX " lds" : : "s"(lds_group), "v"(voff_low), "s"(r), "s"(soff) : "memory");
    else if constexpr (Q == 4) asm volatile("buffer_load_dwordx4 %1, %2, %3 offen" OTTER_T4_DMA_AUX " lds" : : "s"(lds_group), "v"(voff_low), "s"(r), "s"(soff) : "memory");   // M0 written a slot earlier
    else if constexpr (Q == 1) asm volatile("buffer_load_dwordx4 %1, %2, %3 offen offset:1024" OTTER_T4_DMA_AUX " lds" : : "s"(lds_group), "v"(voff_low), "s"(r), "s"(soff) : "memory");
    else if constexpr (Q == 2) asm volatile("buffer_load_dwordx4 %1, %2, %3 offen offset:2048" OTTER_T4_DMA_AUX " lds" : : "s"(lds_group), "v"(voff_low), "s"(r), "s"(soff) : "memory");
    else asm volatile("buffer_load_dwordx4 %1, %2, %3 offen offset:3072" OTTER_T4_DMA_AUX " lds" : : "s"(lds_group), "v"(voff_low), "s"(r), "s"(soff) : "memory");
}

// Address of a transpose read: k-row, half row, swizzled block and the lane's 8 bytes occupy DISJOINT bit fields of the LDS
// offset (bits 9-13 | 8 | 5-7 | 3-4; the +4 rows of the second read, the k-step, the operand region and the buffer are higher or
// free bits), so  row*512 + half*256 + ((i ^ f) * 32) + 8 (r & 3)  ==  lane_const ^ (i * 32): ONE v_xor_b32 with a literal per
// fragment, everything else rides in the instruction's immediate offset (lane_const is an ABSOLUTE LDS address: no symbol add).  The
// 28 possible values per lane are loop invariant and hipcc would hoist them out of the K loop (28 more live registers in a kernel
// whose file is full): the lane constants are therefore passed through an empty asm once per K-tile (LDF macro), which makes them
// opaque per iteration at no instruction.
template <int I>
__device__ __forceinline__ bf16x8_t ldf_tr(unsigned lane_const /* absolute LDS byte address */, int koff) {
    typedef short s16x4_t_ __attribute__((ext_vector_type(4)));
    typedef short s16x8_t_ __attribute__((ext_vector_type(8)));
    const unsigned a = lane_const ^ (unsigned)(I * 32);
    const s16x4_t_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t_*)(uintptr_t)(a + (unsigned)koff));
    const s16x4_t_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t_*)(uintptr_t)(a + (unsigned)koff + 2048u));
    const s16x8_t_ r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
}
// run-time block index (the prologue's unrolled loop)
__device__ __forceinline__ bf16x8_t ldf_tr_rt(unsigned lane_const, int koff, int i) {
    typedef short s16x4_t_ __attribute__((ext_vector_type(4)));
    typedef short s16x8_t_ __attribute__((ext_vector_type(8)));
    const unsigned a = lane_const ^ (unsigned)(i * 32);
    const s16x4_t_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t_*)(uintptr_t)(a + (unsigned)koff));
    const s16x4_t_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t_*)(uintptr_t)(a + (unsigned)koff + 2048u));
    const s16x8_t_ r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
}

// K-major instantiations keep the 64 accumulator blocks in EXPLICIT AGPRs: block (mi, ni) = a[4 (8 mi + ni) .. + 3].  Each asm MFMA names
// its registers and lists them as clobbers (AC_mi_ni), so hipcc accounts for the whole AGPR half and keeps nothing of its own there across
// a K-tile; it never sees a value it could copy or split (with an "+a" operand it did: v_accvgpr_mov around the MFMAs of the peeled last
// K-tiles, no wait states, elements 0 and 3 of every block wrong).
#define AC_0_0 "a0", "a1", "a2", "a3"
#define AC_0_1 "a4", "a5", "a6", "a7"
#define AC_0_2 "a8", "a9", "a10", "a11"
#define AC_0_3 "a12", "a13", "a14", "a15"
#define AC_0_4 "a16", "a17", "a18", "a19"
#define AC_0_5 "a20", "a21", "a22", "a23"
#define AC_0_6 "a24", "a25", "a26", "a27"
#define AC_0_7 "a28", "a29", "a30", "a31"
#define AC_1_0 "a32", "a33", "a34", "a35"
#define AC_1_1 "a36", "a37", "a38", "a39"
#define AC_1_2 "a40", "a41", "a42", "a43"
#define AC_1_3 "a44", "a45", "a46", "a47"
#define AC_1_4 "a48", "a49", "a50", "a51"
#define AC_1_5 "a52", "a53", "a54", "a55"
#define AC_1_6 "a56", "a57", "a58", "a59"
#define AC_1_7 "a60", "a61", "a62", "a63"
#define AC_2_0 "a64", "a65", "a66", "a67"
#define AC_2_1 "a68", "a69", "a70", "a71"
#define AC_2_2 "a72", "a73", "a74", "a75"
#define AC_2_3 "a76", "a77", "a78", "a79"
#define AC_2_4 "a80", "a81", "a82", "a83"
#define AC_2_5 "a84", "a85", "a86", "a87"
#define AC_2_6 "a88", "a89", "a90", "a91"
#define AC_2_7 "a92", "a93", "a94", "a95"
#define AC_3_0 "a96", "a97", "a98", "a99"
#define AC_3_1 "a100", "a101", "a102", "a103"
#define AC_3_2 "a104", "a105", "a106", "a107"
#define AC_3_3 "a108", "a109", "a110", "a111"
#define AC_3_4 "a112", "a113", "a114", "a115"
#define AC_3_5 "a116", "a117", "a118", "a119"
#define AC_3_6 "a120", "a121", "a122", "a123"
#define AC_3_7 "a124", "a125", "a126", "a127"
#define AC_4_0 "a128", "a129", "a130", "a131"
#define AC_4_1 "a132", "a133", "a134", "a135"
#define AC_4_2 "a136", "a137", "a138", "a139"
#define AC_4_3 "a140", "a141", "a142", "a143"
#define AC_4_4 "a144", "a145", "a146", "a147"
#define AC_4_5 "a148", "a149", "a150", "a151"
#define AC_4_6 "a152", "a153", "a154", "a155"
#define AC_4_7 "a156", "a157", "a158", "a159"
#define AC_5_0 "a160", "a161", "a162", "a163"
#define AC_5_1 "a164", "a165", "a166", "a167"
#define AC_5_2 "a168", "a169", "a170", "a171"
#define AC_5_3 "a172", "a173", "a174", "a175"
#define AC_5_4 "a176", "a177", "a178", "a179"
#define AC_5_5 "a180", "a181", "a182", "a183"
#define AC_5_6 "a184", "a185", "a186", "a187"
#define AC_5_7 "a188", "a189", "a190", "a191"
#define AC_6_0 "a192", "a193", "a194", "a195"
#define AC_6_1 "a196", "a197", "a198", "a199"
#define AC_6_2 "a200", "a201", "a202", "a203"
#define AC_6_3 "a204", "a205", "a206", "a207"
#define AC_6_4 "a208", "a209", "a210", "a211"
#define AC_6_5 "a212", "a213", "a214", "a215"
#define AC_6_6 "a216", "a217", "a218", "a219"
#define AC_6_7 "a220", "a221", "a222", "a223"
#define AC_7_0 "a224", "a225", "a226", "a227"
#define AC_7_1 "a228", "a229", "a230", "a231"
#define AC_7_2 "a232", "a233", "a234", "a235"
#define AC_7_3 "a236", "a237", "a238", "a239"
#define AC_7_4 "a240", "a241", "a242", "a243"
#define AC_7_5 "a244", "a245", "a246", "a247"
#define AC_7_6 "a248", "a249", "a250", "a251"
#define AC_7_7 "a252", "a253", "a254", "a255"
#define AC_ROW(MI) AC_##MI##_0, AC_##MI##_1, AC_##MI##_2, AC_##MI##_3, AC_##MI##_4, AC_##MI##_5, AC_##MI##_6, AC_##MI##_7
// zero one row of blocks (32 registers)
#define ACC_ZERO_ROW(MI)                                                                                                   \
    asm volatile(".irp r," ACC_IRP_##MI "\n\tv_accvgpr_write_b32 a\\r, 0\n\t.endr" ::: AC_ROW(MI))
#define ACC_IRP_0 "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31"
#define ACC_IRP_1 "32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63"
#define ACC_IRP_2 "64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95"
#define ACC_IRP_3 "96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127"
#define ACC_IRP_4 "128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159"
#define ACC_IRP_5 "160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191"
#define ACC_IRP_6 "192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,209,210,211,212,213,214,215,216,217,218,219,220,221,222,223"
#define ACC_IRP_7 "224,225,226,227,228,229,230,231,232,233,234,235,236,237,238,239,240,241,242,243,244,245,246,247,248,249,250,251,252,253,254,255"
#ifndef OTTER_KMDBG
#define OTTER_KMDBG 0   // debug builds only (python -m otter_amd.build --define OTTER_KMDBG=n sfx): 1 = builtin MFMA, 2 = builtin LDS-DMA
#endif
// XT (round 6, variant 31): the DMA ring keeps running ACROSS the tiles of a persistent workgroup -- see the block comment at T4_XT_* below.
template <int EPI, int SCH, bool TA = false, bool TB = false, bool XT = false>
__global__ __launch_bounds__(256) void gemm_bf16_t4_kernel(GemmArgs g) {
    constexpr int BM = 256, BN = 256, NT = 256;
    constexpr int TILE = (BM + BN) * 128;  // 64 KB: [256 A rows ; 256 B rows] x 128 B  (K-major operand: [64 k][256 m] x 2 B, same 32 KB)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* __restrict__ A = (const bf16_t*)g.A;
    const bf16_t* __restrict__ B = (const bf16_t*)g.B;
    // K-tiles (nk even, nk >= 2).  K-contiguous operands: the host guarantees K % 128 == 0.  BOTH operands K-major (the weight-gradient
    // form): any K -- the K-tiles past the last row lie outside both descriptors and read as zeros, so the reduction is simply rounded up
    const int nk = (TA && TB) ? (int)(((g.K + 127) >> 7) << 1) : (int)(g.K >> 6);
    const float sgate = g.gate ? tanhf(*g.gate) : 1.0f;
    // K-major operand: K rows of lda elements, M valid columns; rows past K are beyond num_records and read as zeros
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(A), 0, (int)(uint32_t)(TA ? ((g.K - 1) * g.lda * 2 + g.M * 2) : ((g.M - 1) * g.lda * 2 + g.K * 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_t*>(B), 0, (int)(uint32_t)(TB ? ((g.K - 1) * g.ldb * 2 + g.N * 2) : ((g.N - 1) * g.ldb * 2 + g.K * 2)), 0x00020000);
    const u32x4_t rs4_a = gemm_rsrc4(A, (uint32_t)(TA ? ((g.K - 1) * g.lda * 2 + g.M * 2) : ((g.M - 1) * g.lda * 2 + g.K * 2)));
    const u32x4_t rs4_b = gemm_rsrc4(B, (uint32_t)(TB ? ((g.K - 1) * g.ldb * 2 + g.N * 2) : ((g.N - 1) * g.ldb * 2 + g.K * 2)));
    const unsigned smem_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    // scalar byte step of the DMA source: 64 k = 128 bytes along a row per K-tile (K-contiguous operand), or 16 rows (K-major: a
    // K-tile is four such steps)
    const uint32_t ksa = TA ? (uint32_t)(g.lda * 32) : 128u, ksb = TB ? (uint32_t)(g.ldb * 32) : 128u;
    const int swz = (lane & 15) >> 1;
    int ra[2], rb[2], ra_hi[2], rb_hi[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int slot = (4 * ks + (lane >> 4)) ^ swz;
        ra[ks] = (wm * 128 + (lane & 15)) * 128 + (slot << 4);
        rb[ks] = BM * 128 + (wn * 128 + (lane & 15)) * 128 + (slot << 4);
        ra_hi[ks] = ra[ks] + TILE;
        rb_hi[ks] = rb[ks] + TILE;
        asm volatile("" : "+v"(ra_hi[ks]), "+v"(rb_hi[ks]));
    }
    // transpose-read lane constants (K-major operands): k-row (8 g + (r >> 2)) of the k-step | the wave's 256-byte half row | the
    // swizzle f of that k-row in the block field | the lane's 8 bytes of its 32-byte block (+ operand region and buffer)
    constexpr bool ra_T = TA, rb_T = TB;
    const int tr_lane = (((lane >> 4) * 8 + ((lane & 15) >> 2)) * 512) | (((((lane & 15) >> 2) & 3) | (((lane >> 4) & 1) << 2)) * 32) | ((lane & 3) * 8);
    const unsigned smem_abs = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem;
    if ((TA || TB) && (smem_abs & 255u)) __builtin_trap();   // the xor addressing needs the dynamic LDS region on a bank-row boundary (it starts at 0)
    unsigned ra_tb[2], rb_tb[2];
    ra_tb[0] = smem_abs + (unsigned)(tr_lane | (wm * 256));
    rb_tb[0] = smem_abs + (unsigned)((BM * 128) | tr_lane | (wn * 256));
    ra_tb[1] = ra_tb[0] + TILE;
    rb_tb[1] = rb_tb[0] + TILE;
    asm volatile("" : "+v"(ra_tb[0]), "+v"(rb_tb[0]), "+v"(ra_tb[1]), "+v"(rb_tb[1]));
#define TMARK(K_)                                                                                                         \
    do {                                                                                                                  \
        if ((g.dbg & 64) && (blockIdx.x == 0 || blockIdx.x == 131) && lane == 0 && tcount < 8)                            \
        {                                                                                                                 \
            g_gemm_timeline[(((blockIdx.x ? 1 : 0) * 4 + wave) * 8 + tcount) * 8 + (K_)] = __builtin_amdgcn_s_memtime();  \
            if ((K_) == 0 || (K_) == 4)   /* marks 5 / 6: the 100 MHz wall clock at tile start / end -> shader clock of THIS launch */ \
                g_gemm_timeline[(((blockIdx.x ? 1 : 0) * 4 + wave) * 8 + tcount) * 8 + ((K_) == 0 ? 5 : 6)] = wall_clock64();  \
            if ((K_) == 0)   /* mark 7: which launch this is -- M (20 bits), N, K (21 bits each) + the operand layout (bits 63 / 62 = A / B K-major) */ \
                g_gemm_timeline[(((blockIdx.x ? 1 : 0) * 4 + wave) * 8 + tcount) * 8 + 7] =                                     \
                    ((unsigned long long)(TA ? 1 : 0) << 63) | ((unsigned long long)(TB ? 1 : 0) << 62) |                        \
                    (((unsigned long long)g.M & 0xfffffull) << 42) | (((unsigned long long)g.N & 0x1fffffull) << 21) | ((unsigned long long)g.K & 0x1fffffull); \
        }                                                                                                                 \
    } while (0)
    int tcount = 0;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define LDF(dst, base_lo, base_hi, BUFV, KS, I)                                                                                     \
    do {                                                                                                                          \
        if constexpr ((OTTER_T4_ABL & 2) != 0) { /* ablation build: no fragment reads inside the K loop */ }                         \
        else if constexpr (base_lo##_T) {                                                                                              \
            if constexpr ((I) == 0) asm volatile("" : "+v"(base_lo##_tb[(BUFV) & 1]));   /* opaque per use of block 0: no hoisting */   \
            dst = ldf_tr<(I)>(base_lo##_tb[(BUFV) & 1], (KS) * 16384);                                                              \
        }                                                                                                                         \
        else dst = *reinterpret_cast<const bf16x8_t*>(smem + (((BUFV) & 1) ? base_hi[KS] : base_lo[KS]) + (I) * 2048);              \
    } while (0)

    // split form (KTILE_X0): LDFA = address + first 64-bit transpose read (or the whole 128-bit read of a K-contiguous operand),
    // LDFB = the second transpose read one MFMA slot later (nothing for a K-contiguous operand)
    typedef short s16x4_t_ __attribute__((ext_vector_type(4)));
    typedef short s16x8_t_ __attribute__((ext_vector_type(8)));
    unsigned tr_addr = 0;
    s16x4_t_ tr_lo = {0, 0, 0, 0};
#define LDFA(dst, base_lo, base_hi, BUFV, KS, I)                                                                                    \
    do {                                                                                                                          \
        if constexpr (base_lo##_T) {                                                                                              \
            if constexpr ((I) == 0) asm volatile("" : "+v"(base_lo##_tb[(BUFV) & 1]));                                             \
            tr_addr = (base_lo##_tb[(BUFV) & 1] ^ (unsigned)((I) * 32)) + (unsigned)((KS) * 16384);                                  \
            tr_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t_*)(uintptr_t)tr_addr);      \
        } else dst = *reinterpret_cast<const bf16x8_t*>(smem + (((BUFV) & 1) ? base_hi[KS] : base_lo[KS]) + (I) * 2048);            \
    } while (0)
#define LDFB(dst, base_lo, base_hi, BUFV, KS, I)                                                                                    \
    do {                                                                                                                          \
        if constexpr (base_lo##_T) {                                                                                              \
            const s16x4_t_ hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t_*)(uintptr_t)(tr_addr + 2048u)); \
            const s16x8_t_ r_ = __builtin_shufflevector(tr_lo, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                                        \
            dst = __builtin_bit_cast(bf16x8_t, r_);                                                                                \
        }                                                                                                                         \
    } while (0)

    const int ntiles = g.gm * g.gn;
    uint32_t oa[8], ob[8];
    // XT: scalar byte offsets (tile base + K position) of the K-tile the NEXT dma() calls fetch, set by xt_setk(); the per-lane offsets
    // oa / ob are then tile-invariant (rows past M / N lie outside the descriptor and read as zeros: no clamp)
    uint32_t sa_k = 0, sb_k = 0;
    // piece p (0..7 = A, 8..15 = B) of K-tile kt into buffer BUFV
    // plain form: K-tiles walked from `krot_plain` on (g.korder bits 0-1, a function of the tile's index: see xt_tile) -- kt is the logical index
    int krot_plain = 0;
    int kx_t = 0;                 // K-major schedule: wrapped index and scalar source offsets of the K-tile in flight (T4_SETKX)
    uint32_t kx_a = 0, kx_b = 0;
    static_assert(!XT || (!TA && !TB), "the cross-tile form is built for K-contiguous operands only (its K-tile offsets come from the T4_SETK slots of KTILE_T0)");
    constexpr bool M0G = XT && !TA && !TB && (OTTER_T4_M0GROUP != 0);   // M0 written once per four pieces (gemm_dma16_asm_q)
    // (m0_set: the schedule wrote this group's M0 one slot earlier through dma_m0() -- KTILE_T0's in-loop pieces; the prologue's calls write it themselves)
    auto dma = [&](int bufv, int kt, int p, bool m0_set = false) {
        if constexpr ((OTTER_T4_ABL & 1) != 0) { if (kt >= 2) return; }   // ablation build: no LDS-DMA inside the K loop (timing only, wrong results)
        const int wbase = (bufv & 1) * TILE + (p >> 3) * (BM * 128) + ((p & 7) * NT + wave * 64) * 16;
        if constexpr (!XT) {
            kt += krot_plain;
            if (kt >= nk) kt -= nk;
        }
        if constexpr (XT && M0G) {   // grouped M0 (see gemm_dma16_asm_q): wave w fills the 1 KB chunks 8 w .. 8 w + 7 of each operand's 32 KB, piece by piece
            const unsigned grp = smem_lds + (unsigned)((bufv & 1) * TILE + (p >> 3) * (BM * 128) + (wave * 8 + (p & 4)) * 1024);
            const uint32_t vo = p < 8 ? oa[p & 7] : ob[p & 7];
            const uint32_t so = p < 8 ? sa_k : sb_k;
            switch (p & 3) {
                case 0:
                    if (m0_set && (OTTER_T4_M0EARLY != 0)) { if (p < 8) gemm_dma16_asm_q<4>(rs4_a, grp, vo, so); else gemm_dma16_asm_q<4>(rs4_b, grp, vo, so); }
                    else { if (p < 8) gemm_dma16_asm_q<0>(rs4_a, grp, vo, so); else gemm_dma16_asm_q<0>(rs4_b, grp, vo, so); }
                    break;
                case 1: if (p < 8) gemm_dma16_asm_q<1>(rs4_a, grp, vo, so); else gemm_dma16_asm_q<1>(rs4_b, grp, vo, so); break;
                case 2: if (p < 8) gemm_dma16_asm_q<2>(rs4_a, grp, vo, so); else gemm_dma16_asm_q<2>(rs4_b, grp, vo, so); break;
                default: if (p < 8) gemm_dma16_asm_q<3>(rs4_a, grp, vo, so); else gemm_dma16_asm_q<3>(rs4_b, grp, vo, so); break;
            }
            return;
        }
        if constexpr (XT) {   // asm issue for every instantiation: invisible to hipcc's wait-count pass, counted by the schedule's own s_waitcnt
            const unsigned dst = smem_lds + (unsigned)wbase;
            if (p < 8) gemm_dma16_asm(rs4_a, dst, oa[TA ? (p & 1) : (p & 7)], TA ? sa_k + (uint32_t)((p & 7) >> 1) * ksa : sa_k);
            else gemm_dma16_asm(rs4_b, dst, ob[TB ? (p & 1) : (p & 7)], TB ? sb_k + (uint32_t)((p & 7) >> 1) * ksb : sb_k);
            return;
        }
        // K-major operand: piece i = k-rows 8 i + (tid >> 5); pieces i and i + 2 differ by 16 rows = a UNIFORM byte offset, and the
        // swizzle of a row depends on i only through its parity -- two per-lane offsets (even / odd pieces) instead of eight, the rest
        // rides in the scalar offset (the register file of this kernel is full: 256 accumulators + 128 fragment registers)
        if constexpr ((TA || TB) && !(OTTER_KMDBG & 2)) {   // asm issue for BOTH operands of a K-major instantiation (see gemm_dma16_asm)
            const unsigned dst = smem_lds + (unsigned)wbase;
            if (m0_set) {   // (KTILE_X0 since round 6d: dma_m0x() wrote this piece's M0 one slot earlier and T4_SETKX the K-tile's scalar offsets; the
                            //  prologue's calls do both themselves)
                if (p < 8) {
                    if constexpr (TA) gemm_dma16_asm_q<4>(rs4_a, dst, oa[p & 1], kx_a + (uint32_t)((p & 7) >> 1) * ksa);
                    else gemm_dma16_asm_q<4>(rs4_a, dst, oa[p & 7], kx_a);
                } else {
                    if constexpr (TB) gemm_dma16_asm_q<4>(rs4_b, dst, ob[p & 1], kx_b + (uint32_t)((p & 7) >> 1) * ksb);
                    else gemm_dma16_asm_q<4>(rs4_b, dst, ob[p & 7], kx_b);
                }
            } else
            if (p < 8) {
                if constexpr (TA) gemm_dma16_asm(rs4_a, dst, oa[p & 1], (uint32_t)(4 * kt + ((p & 7) >> 1)) * ksa);   // (OTTER_KMDBG: debug builds)
                else gemm_dma16_asm(rs4_a, dst, oa[p & 7], (uint32_t)kt * ksa);
            } else {
                if constexpr (TB) gemm_dma16_asm(rs4_b, dst, ob[p & 1], (uint32_t)(4 * kt + ((p & 7) >> 1)) * ksb);
                else gemm_dma16_asm(rs4_b, dst, ob[p & 7], (uint32_t)kt * ksb);
            }
        } else if (p < 8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(smem + wbase), 16, (int)oa[TA ? (p & 1) : (p & 7)],
                                                     (int)((TA ? (uint32_t)(4 * kt + ((p & 7) >> 1)) : (uint32_t)kt) * ksa), 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(smem + wbase), 16, (int)ob[TB ? (p & 1) : (p & 7)],
                                                     (int)((TB ? (uint32_t)(4 * kt + ((p & 7) >> 1)) : (uint32_t)kt) * ksb), 0, 0);
        }
    };
    // the M0 of the group of four pieces that starts with piece p (p = 0, 4, 8, 12), written by the schedule one MFMA slot before that piece: the
    // s_mov's latency then hides under the MFMA instead of standing in front of the buffer_load (tools/probe/dma_issue.hip: -40 cycles per K-tile)
    auto dma_m0x = [&](int bufv, int p) {   // K-major schedule (KTILE_X0): the M0 of piece p, one MFMA slot before the piece
        if constexpr ((TA || TB) && !(OTTER_KMDBG & 2)) {
            const unsigned dst = smem_lds + (unsigned)((bufv & 1) * TILE + (p >> 3) * (BM * 128) + ((p & 7) * NT + wave * 64) * 16);
            asm volatile("s_mov_b32 m0, %0" : : "s"(dst) : "memory");
        }
    };
    auto dma_m0 = [&](int bufv, int p) {
        if constexpr ((OTTER_T4_ABL & 1) != 0) return;
        if constexpr (XT && M0G && (OTTER_T4_M0EARLY != 0)) {
            const unsigned grp = smem_lds + (unsigned)((bufv & 1) * TILE + (p >> 3) * (BM * 128) + (wave * 8 + (p & 4)) * 1024);
            asm volatile("s_mov_b32 m0, %0" : : "s"(grp) : "memory");
        }
    };
    // ---- T4_XT_* : the cross-tile form (XT, round 6) --------------------------------------------------------------------------------
    // Variant 26 pays, per output tile, a prologue (two K-tiles requested from a standing start: 5-7 k cycles on cold operands) and the
    // drain of the tail's stores (the `__syncthreads()` that frees the parking buffers waits for vmcnt(0): loads AND stores) -- four times per
    // launch and CU at the FFN shapes, which is the whole distance to hipBLASLt's kernel of the same tile and pipeline there (its
    // workgroups end behind their last store and the next one's prologue runs beside the drain).  Here the ring simply keeps running:
    //   * the K-tile after a tile's last one is the NEXT tile's first: K-tile 0 of tile i+1 is requested in iteration nk-2 of tile i
    //     (into buffer 0, exactly where the steady state would put K-tile nk), K-tile 1 right behind tile i's tail (into buffer 1);
    //   * the tail parks its stripes in buffer 1 + the spare LDS above the ring (two stripes per wave instead of four) and is never
    //     waited for: its stores retire under the next tile's first K-tiles (iteration 0's `vmcnt(14)` is the first wait that counts them);
    //   * per-lane DMA offsets are tile-invariant (the tile's base rides in the scalar offset), so a tile switch is a few SALU ops;
    //   * every DMA is issued from asm (as in the K-major instantiations): hipcc's wait-count pass never sees an LDS-DMA in flight and
    //     puts no vmcnt(0) in front of the tail's LDS traffic; completion is counted by hand (vmcnt + raw s_barrier).
    // K order (g.korder, diagnostics + A/B): tile (m, n) may walk its K-tiles rotated by `krot` and permuted inside aligned groups
    // (kgm + 1 K-tiles, offset kc), see xt_tile(): workgroups that share an operand panel then request DIFFERENT K-tiles at the same
    // instant, so that a panel's first touch (an HBM miss) is paid by one of its sharers per K-tile instead of by all of them at once.
    struct XtTile { uint32_t tba, tbb; int krot, kgm, kc; int64_t m0, n0; };
    auto xt_tile = [&](int vbx) {
        XtTile r;
        int tm, tn;
        tile_of_block(g, vbx, tm, tn);
        r.m0 = (int64_t)tm * BM; r.n0 = (int64_t)tn * BN;
        r.tba = TA ? (uint32_t)r.m0 * 2u : (uint32_t)r.m0 * (uint32_t)g.lda * 2u;
        r.tbb = TB ? (uint32_t)r.n0 * 2u : (uint32_t)r.n0 * (uint32_t)g.ldb * 2u;
        // (xcd: the XCD the tile runs on when the grid is a multiple of 8 workgroups -- always, but for odd CU budgets; taken from the TILE's
        //  index so that the K order, hence the fp32 summation order, does not depend on the grid mode)
        const int rotm = g.korder & 3, perm = (g.korder >> 2) & 3, xcd = vbx & 7;
        int rot = 0;
        if (rotm == 1) rot = (int)((0x13023120u >> (4 * xcd)) & 3u) * (nk >> 2);   // XCDs that share an A / B panel: halves / quarters apart
        else if (rotm == 2) rot = (xcd * nk) >> 3;
        else if (rotm == 3) rot = ((tn >> 2) & 3) * (nk >> 2);   // by the tile's N panel (4 tiles wide) only: rows of C do not influence their own K order
        int gmk = perm == 1 ? 3 : (perm == 2 ? 7 : (perm == 3 ? 1 : 0));
        if (nk & gmk) gmk = 0;
        r.krot = __builtin_amdgcn_readfirstlane(rot);
        r.kgm = __builtin_amdgcn_readfirstlane(gmk);
        r.kc = __builtin_amdgcn_readfirstlane((perm == 3 ? (tm + tn) : (tm + 2 * tn)) & gmk);
        return r;
    };
    // scalar offsets of logical K-tile kt of tile T for the dma() calls that follow
    auto xt_setk = [&](const XtTile& T, int kt) {
        int x = kt + T.krot;
        if (x >= nk) x -= nk;
        const int kp = (x & ~T.kgm) | ((x + T.kc) & T.kgm);
        sa_k = __builtin_amdgcn_readfirstlane(T.tba + (uint32_t)kp * (TA ? 4u * ksa : ksa));
        sb_k = __builtin_amdgcn_readfirstlane(T.tbb + (uint32_t)kp * (TB ? 4u * ksb : ksb));
    };
    // round 6d: xt_setk() in three pieces, issued by the K-tile schedule itself on free slots (T4_SETK(n) in gemm_t4_ktile.inc) for the K-tile its DMA pieces
    // are about to fetch: xs_* = the tile (xs_tile) and the logical K-tile index the K loop announced before the macro
    uint32_t xs_tba = 0, xs_tbb = 0;
    int xs_rot = 0, xs_kgm = 0, xs_kc = 0, xs_kt = 0, xs_x = 0, xs_kp = 0;
    auto xs_tile = [&](const XtTile& T) { xs_tba = T.tba; xs_tbb = T.tbb; xs_rot = T.krot; xs_kgm = T.kgm; xs_kc = T.kc; };
    constexpr int XT_PARK = 2 * TILE;       // the wave's parking stripe: XPARK_BYTES each, above the ring (the whole 160 KB of LDS are in use)
    XtTile xt_cur = {}, xt_nxt = {};
    bool xt_have = false;                   // K-tiles 0 and 1 of the tile about to start were requested by the previous tile's K loop
    int xt_slack = 0;                       // tail operations of the previous tile that iteration 0's wait for K-tile 1 may leave in flight
    if constexpr (XT) {
        // start phase (g.korder bits 4-6 = d, diagnostics + A/B): workgroup b waits ((b >> 3) & 3) * d * 1024 cycles before its first tile.  All
        // workgroups of a launch run tiles of equal length, so without it every CU reaches its tail in the same microseconds and the chip's
        // whole output of that round (256 x 128-256 KB) hits the fabric at once: the tails are write-bandwidth bound (bf16 store 10 k cycles,
        // GELU with its two outputs 27 k, profiles/r06_xt_timeline*.txt), while the K loops in between leave the write path idle
        {
            const int dph = (g.korder >> 4) & 7, ph = (int)((blockIdx.x >> 3) & 3);
            for (int i = 0; i < dph * ph; ++i) __builtin_amdgcn_s_sleep(16);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // (grouped M0: piece i of a wave is the 1 KB chunk 8 wave + i of the operand's LDS image -- rows 64 wave + 8 i .. + 7 -- and its source offset
            //  is stored 1024 (i & 3) low: the instruction's immediate offset puts it back.  Rows >= 8 lie >= 8 row pitches >= 4 KB up: no underflow.)
            const int c = M0G ? (wave * 8 + i) * 64 + lane : i * NT + tid, row = c >> 3, phys = c & 7;
            const int slot = phys ^ ((row >> 1) & 7);
            const int krow = c >> 5, c16 = c & 31;
            const int mlog = ((((c16 >> 1) ^ ((krow & 3) | (((krow >> 3) & 1) << 2))) << 1) | (c16 & 1)) * 8;
            const uint32_t low = M0G ? (uint32_t)(i & 3) * 1024u : 0u;
            if constexpr (TA) { if (i < 2) oa[i] = ((uint32_t)krow * (uint32_t)g.lda + (uint32_t)mlog) * 2u; }
            else oa[i] = ((uint32_t)row * (uint32_t)g.lda + (uint32_t)(slot * 8)) * 2u - low;
            if constexpr (TB) { if (i < 2) ob[i] = ((uint32_t)krow * (uint32_t)g.ldb + (uint32_t)mlog) * 2u; }
            else ob[i] = ((uint32_t)row * (uint32_t)g.ldb + (uint32_t)(slot * 8)) * 2u - low;
        }
    }
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        int tile_m = 0, tile_n = 0;
        if constexpr (XT) {
            // (round 6d) the tile the previous iteration prefetched for IS this one: its XtTile (tile mapping = three integer divisions, K rotation) is reused
            if (xt_have) xt_cur = xt_nxt;
            else xt_cur = xt_tile(vb);
        } else tile_of_block(g, vb, tile_m, tile_n);
        const int64_t m0 = XT ? xt_cur.m0 : (int64_t)tile_m * BM, n0 = XT ? xt_cur.n0 : (int64_t)tile_n * BN;
        TMARK(0);
        const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N) && !(g.dbg & 16) && (g.cdt == OTTER_F32 || g.wide);
        // XT: this tile's K loop requests the next tile's first two K-tiles when there is one and when its own tail leaves the ring alone
        // (the full-tile tail parks above the ring; the masked tail of an edge tile uses the ring's LDS, as in variant 26)
        bool xt_pre = false;
        if constexpr (!XT) {
            const int rotm = g.korder & 3, xcd = vb & 7;
            krot_plain = __builtin_amdgcn_readfirstlane(rotm == 1 ? (int)((0x13023120u >> (4 * xcd)) & 3u) * (nk >> 2)
                                                        : (rotm == 2 ? (xcd * nk) >> 3 : (rotm == 3 ? ((tile_n >> 2) & 3) * (nk >> 2) : 0)));
        }
        if constexpr (XT) {
            xt_pre = full && (vb + (int)gridDim.x < ntiles);
            xt_nxt = xt_pre ? xt_tile(vb + (int)gridDim.x) : xt_cur;
        }
#pragma unroll
        for (int i = 0; i < 8 && !XT; ++i) {
            const int c = i * NT + tid, row = c >> 3, phys = c & 7;
            const int slot = phys ^ ((row >> 1) & 7);
            int64_t ga = m0 + row; if (ga > g.M - 1) ga = g.M - 1;
            int64_t gb = n0 + row; if (gb > g.N - 1) gb = g.N - 1;
            // K-major operand: 16-byte chunk c of the [64 k][256 m] tile = k-row c >> 5, physical 16-byte slot c & 31; the 32-byte
            // block index is un-swizzled by f(k-row) to find the source columns (columns past M / N only feed output rows that are
            // never stored; rows past K are outside the descriptor and read as zeros)
            const int krow = c >> 5, c16 = c & 31;
            const int mlog = ((((c16 >> 1) ^ ((krow & 3) | (((krow >> 3) & 1) << 2))) << 1) | (c16 & 1)) * 8;
            if constexpr (TA) { if (i < 2) oa[i] = ((uint32_t)krow * (uint32_t)g.lda + (uint32_t)m0 + (uint32_t)mlog) * 2u; }
            else oa[i] = (uint32_t)((ga * g.lda + slot * 8) * 2);
            if constexpr (TB) { if (i < 2) ob[i] = ((uint32_t)krow * (uint32_t)g.ldb + (uint32_t)n0 + (uint32_t)mlog) * 2u; }
            else ob[i] = (uint32_t)((gb * g.ldb + slot * 8) * 2);
        }
        // ---- prologue: K-tiles 0 and 1 in flight, 0 readable; the 256 accumulator registers are zeroed while they land ----
        if constexpr (XT) {
            if (!xt_have) {   // first tile of the workgroup, or behind an edge tile: a standing start, as in variant 26
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // (an edge tile's masked tail has read its parking buffers out of the ring)
                xt_setk(xt_cur, 0);
#pragma unroll
                for (int p = 0; p < 16; ++p) dma(0, 0, p);
                xt_setk(xt_cur, 1);
#pragma unroll
                for (int p = 0; p < 16; ++p) dma(1, 1, p);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
            // K-tile 0 is in buffer 0 as far as this wave's pieces go (waited for above, or behind the previous K loop): the barrier makes it so
            // across waves.  K-tile 1 is in flight, and so may be the previous tile's stores (xt_slack)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
#pragma unroll
            for (int p = 0; p < 16; ++p) dma(0, 0, p);
#pragma unroll
            for (int p = 0; p < 16; ++p) dma(1, 1, p);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t acc[8][8];
        // accumulators in explicit AGPRs (asm MFMAs): the K-major instantiations and EVERY cross-tile instantiation (with the builtin MFMA
        // hipcc re-homes accumulator blocks around the restructured tile loop: hundreds of v_accvgpr_mov at the K loop's entry, spills)
        constexpr bool XACC = (TA || TB || XT) && !(OTTER_KMDBG & 1);
        if constexpr (XACC && XT) {
            // nothing: the first K-tile's k-step 0 runs with C = 0 (MMAZ in KTILE_T0F)
        } else if constexpr (XACC) {
            ACC_ZERO_ROW(0); ACC_ZERO_ROW(1); ACC_ZERO_ROW(2); ACC_ZERO_ROW(3); ACC_ZERO_ROW(4); ACC_ZERO_ROW(5); ACC_ZERO_ROW(6); ACC_ZERO_ROW(7);
        } else {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!XT) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        bf16x8_t fm[2][8], fn[2][8];  // [k-step][16-row block]: fm = A rows (b-operand), fn = B rows (a-operand)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (TA) fm[0][i] = ldf_tr_rt(ra_tb[0], 0, i);
            else fm[0][i] = *reinterpret_cast<const bf16x8_t*>(smem + ra[0] + i * 2048);
            if constexpr (TB) fn[0][i] = ldf_tr_rt(rb_tb[0], 0, i);
            else fn[0][i] = *reinterpret_cast<const bf16x8_t*>(smem + rb[0] + i * 2048);
        }
        if constexpr ((OTTER_T4_ABL & 2) != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { fm[1][i] = fm[0][i]; fn[1][i] = fn[0][i]; }
        }
        TMARK(1);
// K-major instantiations: asm MFMAs on explicit AGPR blocks (see AC_mi_ni above).  With the builtin, hipcc's register allocator gives up
// on these variants (two 64-bit transpose reads per fragment instead of one 128-bit read): accumulators end up in VGPRs, every MFMA
// result is copied out of a[0:3] and ~1 400 v_accvgpr_* / 360 s_nop land in the K loop (2x slower, measured).  MFMAs on one block are 64
// slots apart; the wait states before the tail's reads are spelled out after the loop.
#define MMA(KS, MI, NI)                                                                                                             \
    do {                                                                                                                          \
        if constexpr (XACC)                                                                                                        \
            asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]"                                                  \
                         : : "v"(fn[KS][NI]), "v"(fm[KS][MI]), "n"(((MI) * 8 + (NI)) * 4), "n"(((MI) * 8 + (NI)) * 4 + 3) : AC_##MI##_##NI); \
        else acc[MI][NI] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[KS][NI], fm[KS][MI], acc[MI][NI], 0, 0, 0);                  \
    } while (0)
// k-step 0 of a tile's FIRST K-tile in the cross-tile form (KTILE_T0F): C = 0, the accumulators are not zeroed beforehand
#define MMAZ(KS, MI, NI)                                                                                                            \
    do {                                                                                                                          \
        if constexpr (XACC && XT)                                                                                                  \
            asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0"                                                           \
                         : : "v"(fn[KS][NI]), "v"(fm[KS][MI]), "n"(((MI) * 8 + (NI)) * 4), "n"(((MI) * 8 + (NI)) * 4 + 3) : AC_##MI##_##NI); \
        else MMA(KS, MI, NI);                                                                                                      \
    } while (0)
// the wait for K-tile t + 1 in front of barrier #2: N pieces of K-tile t + 2 have been issued behind it (DMA), none (!DMA), or -- the first
// iteration of a cross-tile tile (DMA == 2) -- N pieces plus the previous tile's tail, of which xt_slack operations may stay in flight
#define T4_WAIT_NEXT(DMA, N)                                                                                              \
    do {                                                                                                                  \
        if constexpr ((OTTER_T4_ABL & 8) != 0) { /* ablation build: the K loop never waits for its LDS-DMA */ }           \
        else if ((DMA) == 2) {                                                                                                 \
            if (xt_slack) asm volatile("s_waitcnt vmcnt(%c0)" : : "n"((N) + 32) : "memory");                              \
            else asm volatile("s_waitcnt vmcnt(%c0)" : : "n"(N) : "memory");                                              \
        } else if (DMA) asm volatile("s_waitcnt vmcnt(%c0)" : : "n"(N) : "memory");                                       \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                             \
    } while (0)
// the 128-slot K-tile schedules (tools/gen/gemm_t4_schedule.py inc): KTILE_T0_0 (K-contiguous operands) / KTILE_X0_0 (K-major: split
// transpose reads)
#if (OTTER_T4_ABL & 4)   // ablation build: no workgroup barriers inside the K loop
#define __builtin_amdgcn_s_barrier() ((void)0)
#endif
// the scalar source offsets of the K-tile this iteration's DMA pieces fetch, in three pieces on free slots of the schedule (cross-tile form; nothing otherwise)
#define T4_SETK(N)                                                                                                        \
    do {                                                                                                                  \
        if constexpr (XT && !(OTTER_T4_SETK_BURST)) {                                                                                               \
            if constexpr ((N) == 0) { xs_x = xs_kt + xs_rot; if (xs_x >= nk) xs_x -= nk; }                                \
            else if constexpr ((N) == 1) xs_kp = (xs_x & ~xs_kgm) | ((xs_x + xs_kc) & xs_kgm);                            \
            else {                                                                                                        \
                sa_k = __builtin_amdgcn_readfirstlane(xs_tba + (uint32_t)xs_kp * (TA ? 4u * ksa : ksa));                  \
                sb_k = __builtin_amdgcn_readfirstlane(xs_tbb + (uint32_t)xs_kp * (TB ? 4u * ksb : ksb));                  \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
// K-major schedule (KTILE_X0, plain form): the wrapped K-tile index and the two operands' scalar source offsets of the K-tile whose pieces follow
#define T4_SETKX(N, KT_)                                                                                                  \
    do {                                                                                                                  \
        if constexpr (!XT && (TA || TB)) {                                                                                \
            if constexpr ((N) == 0) { kx_t = (KT_) + krot_plain; if (kx_t >= nk) kx_t -= nk; }                            \
            else {                                                                                                        \
                kx_a = __builtin_amdgcn_readfirstlane((uint32_t)kx_t * (TA ? 4u * ksa : ksa));                            \
                kx_b = __builtin_amdgcn_readfirstlane((uint32_t)kx_t * (TB ? 4u * ksb : ksb));                            \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
#include "gemm_t4_ktile.inc"
#define KTILE_T0 KTILE_T0_0
#define KTILE_T0F KTILE_T0F_0
#define KTILE_X0 KTILE_X0_0
#ifdef OTTER_EXPERIMENTAL  // tools-only: variants 27-29: alternative slot placements T1-T3 of variant 26's K-tile schedule (generated by tools/gen/gemm_t4_schedule.py)
#include "experimental/gemm_t4_placements_t1_t3.inc"
#endif
#define KLOOP(KT)                                   \
    do {                                            \
        int t = 0;                                  \
        for (; t + 2 < nk; t += 2) {                \
            KT(0, t, true, true);                   \
            KT(1, t + 1, true, true);               \
        }                                           \
        KT(0, t, false, true);                      \
        KT(1, t + 1, false, false);                 \
    } while (0)
// XT: dma() fetches what xt_setk() named.  The ring runs on into the next tile: iterations nk-2 / nk-1 request ITS K-tiles 0 / 1 (xt_pre) into
// the buffers the steady state would refill anyway.  Iteration 0 is peeled for its wait: behind a full-tile tail, K-tile 1 is followed in the
// queue by that tail's >= 32 global accesses, which the count may leave in flight (DMA == 2, see T4_WAIT_NEXT); iteration 1's wait for
// K-tile 2 is the first that retires them.  Needs nk >= 4 (the host falls back to the plain form below that).
// (XS_AT: announce the tile / K-tile index for the T4_SETK slots; the A/B build -DOTTER_T4_SETK_BURST=1 computes the offsets in one burst in front of the
//  K-tile instead, as the kernel did until round 6d)
#if OTTER_T4_SETK_BURST
#define XS_TILE(T) ((void)0)
#define XS_AT(T, K_) xt_setk(T, K_)
#else
#define XS_TILE(T) xs_tile(T)
#define XS_AT(T, K_) (xs_kt = (K_))
#endif
#define KLOOP_XT(KT, KTF)                           \
    do {                                            \
        XS_TILE(xt_cur);                            \
        XS_AT(xt_cur, 2);                           \
        KTF(0, 0, 2, true);                         \
        XS_AT(xt_cur, 3);                           \
        KT(1, 1, true, true);                       \
        int t = 2;                                  \
        for (; t + 2 < nk; t += 2) {                \
            XS_AT(xt_cur, t + 2);                   \
            KT(0, t, true, true);                   \
            XS_AT(xt_cur, t + 3);                   \
            KT(1, t + 1, true, true);               \
        }                                           \
        XS_TILE(xt_nxt);                            \
        XS_AT(xt_nxt, 0);                           \
        KT(0, t, xt_pre, true);                     \
        XS_AT(xt_nxt, 1);                           \
        KT(1, t + 1, xt_pre, false);                \
    } while (0)
#ifdef OTTER_EXPERIMENTAL
        static_assert(!XT || SCH == 0, "the cross-tile form runs the default placement only");
        if constexpr (XT) {
            if constexpr (TA || TB) KLOOP_XT(KTILE_X0, KTILE_X0);
            else KLOOP_XT(KTILE_T0, KTILE_T0F);
        } else if constexpr (SCH == 0) KLOOP(KTILE_T0);
        else if constexpr (SCH == 1) KLOOP(KTILE_T1);
        else if constexpr (SCH == 2) KLOOP(KTILE_T2);
        else KLOOP(KTILE_T3);
#undef KTILE_T1
#undef KTILE_T2
#undef KTILE_T3
#else
        static_assert(SCH == 0, "the alternative placements of variant 26 (27-29) are in the OTTER_EXPERIMENTAL build only");
        if constexpr (XT) {
            if constexpr ((TA || TB) && !(OTTER_KMDBG & 4)) KLOOP_XT(KTILE_X0, KTILE_X0);
            else KLOOP_XT(KTILE_T0, KTILE_T0F);
        } else if constexpr ((TA || TB) && !(OTTER_KMDBG & 4)) KLOOP(KTILE_X0);
        else KLOOP(KTILE_T0);
#endif
#if (OTTER_T4_ABL & 4)
#undef __builtin_amdgcn_s_barrier
#endif
#undef KLOOP
#undef KLOOP_XT
#undef T4_SETKX
#undef XS_TILE
#undef XS_AT
#undef T4_SETK
#undef T4_WAIT_NEXT
#undef KTILE_X0
#undef KTILE_T0
#undef KTILE_T0F
#undef KTILE_T0F_0
#undef MMAZ
#undef KTILE_T0_0
#undef KTILE_X0_0
#undef MMA

        // ---- epilogue: both buffers are dead (every fragment read retired before barrier #1 of the last K-tile, no DMA in flight) ----
        // asm MFMAs (K-major instantiations): hipcc does not know their latency, so the wait states between the last MFMAs and the
        // first accumulator reads of the tail are spelled out (a 16x16x32 result is readable 8 passes + 2 after issue)
        // the tail then parks them straight out of the AGPR half (park_stripe(AgprAcc)).
        if constexpr (XACC) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        // XT: 32 pieces in flight (the next tile's K-tiles 0 and 1); K-tile 0's were requested an iteration ago -- this wave's have landed once
        // at most the 16 of K-tile 1 are outstanding (the barrier that publishes them across waves is the one at the top of the next tile)
        if constexpr (XT) { if (xt_pre) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        TMARK(2);
        float part = 0.f;
        const AgprAcc xacc;
        if (full) {
            if constexpr (XT) {
                // the lane index goes through an empty asm per tile: everything the tail derives from it (row numbers, 64-bit addresses, swizzled
                // LDS offsets of every `it`) is then computed HERE and dies here -- hoisted out of the tile loop (they are loop invariant) these
                // values stayed live across the K loop, whose file is full, and were spilled to scratch: reloads + s_waitcnt vmcnt(0) inside iteration 0
                int lane_t = lane;
                asm volatile("" : "+v"(lane_t));
                char* stripe = smem + XT_PARK + wave * XPARK_BYTES;
                if (g.cdt == OTTER_BF16) part = xtail_wave_full<EPI, true>(g, sgate, stripe, m0 + wm * 128, n0 + wn * 128, lane_t);
                else part = xtail_wave_full<EPI, false>(g, sgate, stripe, m0 + wm * 128, n0 + wn * 128, lane_t);
            } else {
                float* blk4 = reinterpret_cast<float*>(smem) + wave * (TAIL_STRIPES * 32 * EPI_LD);
                if constexpr (XACC) {
                    if (g.cdt == OTTER_BF16) part = tail_wave_full<EPI, true>(g, sgate, xacc, blk4, m0 + wm * 128, n0 + wn * 128, lane);
                    else part = tail_wave_full<EPI, false>(g, sgate, xacc, blk4, m0 + wm * 128, n0 + wn * 128, lane);
                } else {
                    if (g.cdt == OTTER_BF16) part = tail_wave_full<EPI, true>(g, sgate, acc, blk4, m0 + wm * 128, n0 + wn * 128, lane);
                    else part = tail_wave_full<EPI, false>(g, sgate, acc, blk4, m0 + wm * 128, n0 + wn * 128, lane);
                }
            }
        } else {
            float* blk = reinterpret_cast<float*>(smem) + wave * (32 * EPI_LD);   // (XT: nothing was requested into the ring: xt_pre is false)
            int lane_t = lane;
            if constexpr (XT) asm volatile("" : "+v"(lane_t));
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if constexpr (XACC) park_stripe(blk, xacc, st, lane_t);
                else park_stripe(blk, acc, st, lane_t);
                __builtin_amdgcn_wave_barrier();
                part += epilogue_stripe<EPI>(g, sgate, blk, m0 + wm * 128 + (st >> 1) * 32, n0 + wn * 128 + (st & 1) * 64, lane_t);
                __builtin_amdgcn_wave_barrier();
            }
        }
        TMARK(3);
        if constexpr (XT) {
            // no drain: the stores stay in flight (iteration 1 of the next tile is the first wait that counts them).  The gate partial's
            // cross-wave sum goes through the first word of each wave's own parking stripe (dead: its tail is over)
            if ((EPI == OTTER_EPI_GATE_BWD || EPI == OTTER_EPI_STORE) && g.partial != nullptr) {
                float* red = reinterpret_cast<float*>(smem + XT_PARK);
                part = wave_sum(part);
                if (lane == 0) red[wave * (XPARK_BYTES / 4)] = part;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (tid == 0) g.partial[vb] = red[0] + red[XPARK_BYTES / 4] + red[2 * (XPARK_BYTES / 4)] + red[3 * (XPARK_BYTES / 4)];
            }
            xt_have = xt_pre;
            xt_slack = xt_pre ? 32 : 0;
        } else {
            block_partial<4, EPI>(g, part, reinterpret_cast<float*>(smem), vb);
            __syncthreads();  // the next tile's prologue DMA overwrites the stripes
        }
        TMARK(4);
        ++tcount;
    }
    if constexpr (XT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last tile's look-ahead K-tile: no LDS-DMA may outlive the workgroup
#undef LDF
#undef LDFA
#undef LDFB
#undef SB
#undef TMARK
}

// ------------------------------------------------------------------------------------------------------------
// bf16 small-grid kernel (variant 25): 128x128 tile, 4 waves of 64x64, BK = 64 stages in a 4-deep LDS-DMA ring.
// Why: the gated block's and the perceiver's projections with N or M = 512 are at most 128-256 tiles -- one block per CU at
// best -- with a LONG reduction (K = 4096): the register-staged double buffer of gemm_bf16_kernel<128,128> pays one global
// latency per K-tile there (66-76 us against 19-27 us for hipBLASLt on 4096x512x4096 / 512x4096x4096 / 512x1024x4096,
// tools/gemm_small.py), and a smaller tile (64x128: all 256 CUs busy) changes nothing (74 us) -- it is latency, not occupancy.
// So: the ring + counted-vmcnt pipeline of variant 17 on a 128x128 tile with the 128-byte-row LDS image: the DMA of stage s+3
// is issued during step s (three K-tiles of prefetch, ~2.5 us to land), one raw barrier per K-tile, slots pinned (one fragment
// read per MFMA slot, a DMA piece on every other one).  Schedule generated by tools/gen/gemm_s4_schedule.py.
// Requires K % 256 == 0 (four stages per unrolled trip) and operands spanning < 4 GB.
// ------------------------------------------------------------------------------------------------------------
template <int EPI, bool CBF16>
__device__ __forceinline__ float tail_wave2_full(const GemmArgs& g, float s, const f32x16_t (&acc)[2][2], float* __restrict__ blk2, int64_t m_wave,
                                                 int64_t n_wave, int lane) {
    using T = TailShape<CBF16>;
    const void* ip; int64_t ild; int idt;
    const bool has_in = tail_input<EPI>(g, ip, ild, idt);
    float part = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        park_block(blk2 + mi * (32 * EPI_LD), acc[mi][0], lane, 0);
        park_block(blk2 + mi * (32 * EPI_LD), acc[mi][1], lane, 32);
    }
    __builtin_amdgcn_wave_barrier();
    auto run = [&](auto inbf) {
        constexpr bool INBF16 = decltype(inbf)::value;
        uint4 r0[T::NIT][2], r1[T::NIT][2];
        if (has_in) {
            tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave, n_wave, lane, r0);
            tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave + 32, n_wave, lane, r1);
        }
        part += tail_stripe_full<EPI, CBF16, INBF16>(g, s, blk2, m_wave, n_wave, lane, r0, has_in);
        part += tail_stripe_full<EPI, CBF16, INBF16>(g, s, blk2 + 32 * EPI_LD, m_wave + 32, n_wave, lane, r1, has_in);
    };
    if (EPI == OTTER_EPI_GELU || idt == OTTER_BF16) run(std::true_type{});
    else run(std::false_type{});
    __builtin_amdgcn_wave_barrier();
    return part;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_s4_kernel(GemmArgs g) {
    constexpr int BM = 128, BN = 128, NT = 256;
    constexpr int STAGE = (BM + BN) * 128;  // 32 KB: [128 A rows ; 128 B rows] x 128 B
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* __restrict__ A = (const bf16_t*)g.A;
    const bf16_t* __restrict__ B = (const bf16_t*)g.B;
    const int nk = (int)(g.K >> 6);  // stages (host guarantees nk % 4 == 0, nk >= 4)
    const float sgate = g.gate ? tanhf(*g.gate) : 1.0f;
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)(uint32_t)((g.M - 1) * g.lda * 2 + g.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)(uint32_t)((g.N - 1) * g.ldb * 2 + g.K * 2), 0x00020000);
    // fragment read bases per k-step (row = w*64 + i*32 + (lane&31), slot (2*ks + (lane>>5)) ^ ((row>>1)&7)); ring slots 0/1 through the
    // ds_read immediate, slots 2/3 (+64 KB) through their own registers
    const int swz = ((lane & 31) >> 1) & 7;
    int ra[4], rb[4], ra_hi[4], rb_hi[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int slot = (2 * ks + (lane >> 5)) ^ swz;
        ra[ks] = (wm * 64 + (lane & 31)) * 128 + (slot << 4);
        rb[ks] = BM * 128 + (wn * 64 + (lane & 31)) * 128 + (slot << 4);
        ra_hi[ks] = ra[ks] + 2 * STAGE;
        rb_hi[ks] = rb[ks] + 2 * STAGE;
        asm volatile("" : "+v"(ra_hi[ks]), "+v"(rb_hi[ks]));
    }
#define SB() __builtin_amdgcn_sched_barrier(0)
#define LDF(dst, base, S_, KS, I)                                                                                          \
    dst = *reinterpret_cast<const bf16x8_t*>(smem + ((S_) < 2 ? base[KS] + (S_) * STAGE : base##_hi[KS] + ((S_) - 2) * STAGE) + (I) * 4096)

    const int ntiles = g.gm * g.gn;
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        int tile_m, tile_n;
        tile_of_block(g, vb, tile_m, tile_n);
        const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
        uint32_t oa[4], ob[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = i * NT + tid, row = c >> 3, phys = c & 7;
            const int slot = phys ^ ((row >> 1) & 7);
            int64_t ga = m0 + row; if (ga > g.M - 1) ga = g.M - 1;
            int64_t gb = n0 + row; if (gb > g.N - 1) gb = g.N - 1;
            oa[i] = (uint32_t)((ga * g.lda + slot * 8) * 2);
            ob[i] = (uint32_t)((gb * g.ldb + slot * 8) * 2);
        }
        // piece p (0..3 = A, 4..7 = B) of the stage holding K columns [step*64, +64) into ring slot S_
        auto dma = [&](int S_, int step, int p) {
            const int wbase = S_ * STAGE + (p >> 2) * (BM * 128) + ((p & 3) * NT + wave * 64) * 16;
            if (p < 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(smem + wbase), 16, (int)oa[p & 3],
                                                         step * 128, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(smem + wbase), 16, (int)ob[p & 3],
                                                         step * 128, 0, 0);
        };
        // ---- prologue: stages 0..2 in flight, 0 and 1 readable ----
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int p = 0; p < 8; ++p) dma(st, st, p);
        f32x16_t acc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8_t fm[4][2], fn[4][2];  // [k-step][32-row block]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            LDF(fn[0][i], rb, 0, 0, i);
            LDF(fm[0][i], ra, 0, 0, i);
        }
#define MMA(KS, MI, NI) acc[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fn[KS][NI], fm[KS][MI], acc[MI][NI], 0, 0, 0)
// ---- GENERATED by tools/gen/gemm_s4_schedule.py (do not edit by hand) ----
#define KSTEP(S, STEPV, DMA, NEXT, VMW)                                                                               \
    do {                                                                                                              \
        constexpr int SN = ((S) + 1) & 3, SD = ((S) + 3) & 3;                                                       \
        MMA(0, 0, 0); SB(); LDF(fn[1][0], rb, S, 1, 0); SB();                                                         \
        MMA(0, 1, 0); SB(); LDF(fm[1][0], ra, S, 1, 0); SB(); if (DMA) dma(SD, (STEPV) + 3, 0); SB();                 \
        MMA(0, 0, 1); SB(); LDF(fm[1][1], ra, S, 1, 1); SB();                                                         \
        MMA(0, 1, 1); SB(); LDF(fn[1][1], rb, S, 1, 1); SB(); if (DMA) dma(SD, (STEPV) + 3, 1); SB();                 \
        MMA(1, 0, 0); SB(); LDF(fn[2][0], rb, S, 2, 0); SB();                                                         \
        MMA(1, 1, 0); SB(); LDF(fm[2][0], ra, S, 2, 0); SB(); if (DMA) dma(SD, (STEPV) + 3, 2); SB();                 \
        MMA(1, 0, 1); SB(); LDF(fm[2][1], ra, S, 2, 1); SB();                                                         \
        MMA(1, 1, 1); SB(); LDF(fn[2][1], rb, S, 2, 1); SB(); if (DMA) dma(SD, (STEPV) + 3, 3); SB();                 \
        MMA(2, 0, 0); SB(); LDF(fn[3][0], rb, S, 3, 0); SB();                                                         \
        MMA(2, 1, 0); SB(); LDF(fm[3][0], ra, S, 3, 0); SB(); if (DMA) dma(SD, (STEPV) + 3, 4); SB();                 \
        MMA(2, 0, 1); SB(); LDF(fm[3][1], ra, S, 3, 1); SB();                                                         \
        MMA(2, 1, 1); SB(); LDF(fn[3][1], rb, S, 3, 1); SB(); if (DMA) dma(SD, (STEPV) + 3, 5); SB();                 \
        MMA(3, 0, 0); SB(); if (NEXT) { LDF(fn[0][0], rb, SN, 0, 0); } SB();                                          \
        MMA(3, 1, 0); SB(); if (NEXT) { LDF(fm[0][0], ra, SN, 0, 0); } SB(); if (DMA) dma(SD, (STEPV) + 3, 6); SB();  \
        MMA(3, 0, 1); SB(); if (NEXT) { LDF(fm[0][1], ra, SN, 0, 1); } SB();                                          \
        MMA(3, 1, 1); SB(); if (NEXT) { LDF(fn[0][1], rb, SN, 0, 1); } SB(); if (DMA) dma(SD, (STEPV) + 3, 7); SB();  \
        asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();                   \
    } while (0)
// ---- end of generated schedule ----
        int st = 0;
        for (; st + 4 < nk; st += 4) {   // full trips: every step issues its stage st+3
            KSTEP(0, st, true, true, 8);
            KSTEP(1, st + 1, true, true, 8);
            KSTEP(2, st + 2, true, true, 8);
            KSTEP(3, st + 3, true, true, 8);
        }
        KSTEP(0, st, true, true, 8);      // last trip: stage nk-1 is issued by its first step, then the queue drains
        KSTEP(1, st + 1, false, true, 0);
        KSTEP(2, st + 2, false, true, 0);
        KSTEP(3, st + 3, false, false, 0);
#undef KSTEP
#undef MMA

        // ---- epilogue: the ring is free (every DMA retired, every wave past the last barrier, every fragment read consumed) ----
        float part = 0.f;
        const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N) && !(g.dbg & 16) && (g.cdt == OTTER_F32 || g.wide);
        if (full) {
            float* blk2 = reinterpret_cast<float*>(smem) + wave * (2 * 32 * EPI_LD);
            if (g.cdt == OTTER_BF16) part = tail_wave2_full<EPI, true>(g, sgate, acc, blk2, m0 + wm * 64, n0 + wn * 64, lane);
            else part = tail_wave2_full<EPI, false>(g, sgate, acc, blk2, m0 + wm * 64, n0 + wn * 64, lane);
        } else {
            float* blk = reinterpret_cast<float*>(smem) + wave * (32 * EPI_LD);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                park_block(blk, acc[mi][0], lane, 0);
                park_block(blk, acc[mi][1], lane, 32);
                __builtin_amdgcn_wave_barrier();
                part += epilogue_stripe<EPI>(g, sgate, blk, m0 + wm * 64 + mi * 32, n0 + wn * 64, lane);
                __builtin_amdgcn_wave_barrier();
            }
        }
        block_partial<4, EPI>(g, part, reinterpret_cast<float*>(smem), vb);
        __syncthreads();  // the next tile's prologue DMA overwrites the stripes
    }
#undef LDF
#undef SB
}

// ------------------------------------------------------------------------------------------------------------
// bf16 small-grid kernel, HALF-HEIGHT tiles (variant 30, round 4): the ring kernel above on a 64 x 128 tile (4 waves of 32 x 64),
// BK = 64 stages of 24 KB in the same 4-deep LDS-DMA ring.  Why: the ring kernel is bound by BYTES IN FLIGHT per CU (three stages of
// prefetch against ~1.5-2 us of latency = ~62 GB/s per CU whatever the tile does with them), and the skinny products of the fusion modules
// (4096 x 512 x 4096 in some order; the resampler's 512-row FFN) are only 32-128 tiles of 128 x 128: 128 or fewer of the 256 CUs stream at
// all.  Half-height tiles double the workgroups (to_q / dWo / dWq: 256) -- 1.5 x the L2 -> LDS bytes in total, on twice the CUs.
// Same LDS image, swizzle, fragment addresses and accumulation order per element as variant 25 (bit-identical results).
// ------------------------------------------------------------------------------------------------------------
template <int EPI, bool CBF16>
__device__ __forceinline__ float tail_wave1_full(const GemmArgs& g, float s, const f32x16_t (&acc)[1][2], float* __restrict__ blk1, int64_t m_wave,
                                                 int64_t n_wave, int lane) {
    using T = TailShape<CBF16>;
    const void* ip; int64_t ild; int idt;
    const bool has_in = tail_input<EPI>(g, ip, ild, idt);
    float part = 0.f;
    park_block(blk1, acc[0][0], lane, 0);
    park_block(blk1, acc[0][1], lane, 32);
    __builtin_amdgcn_wave_barrier();
    auto run = [&](auto inbf) {
        constexpr bool INBF16 = decltype(inbf)::value;
        uint4 r0[T::NIT][2];
        if (has_in) tail_stripe_load<EPI, CBF16, INBF16>(ip, ild, m_wave, n_wave, lane, r0);
        part += tail_stripe_full<EPI, CBF16, INBF16>(g, s, blk1, m_wave, n_wave, lane, r0, has_in);
    };
    if (EPI == OTTER_EPI_GELU || idt == OTTER_BF16) run(std::true_type{});
    else run(std::false_type{});
    __builtin_amdgcn_wave_barrier();
    return part;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_s4h_kernel(GemmArgs g) {
    constexpr int BM = 64, BN = 128, NT = 256;
    constexpr int STAGE = (BM + BN) * 128;  // 24 KB: [64 A rows ; 128 B rows] x 128 B
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const bf16_t* __restrict__ A = (const bf16_t*)g.A;
    const bf16_t* __restrict__ B = (const bf16_t*)g.B;
    const int nk = (int)(g.K >> 6);  // stages (host guarantees nk % 4 == 0, nk >= 4)
    const float sgate = g.gate ? tanhf(*g.gate) : 1.0f;
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)(uint32_t)((g.M - 1) * g.lda * 2 + g.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)(uint32_t)((g.N - 1) * g.ldb * 2 + g.K * 2), 0x00020000);
    // fragment read bases per k-step: A row = wm*32 + (lane&31), B row = wn*64 + i*32 + (lane&31); slot (2*ks + (lane>>5)) ^ ((row>>1)&7)
    const int swz = ((lane & 31) >> 1) & 7;
    int ra[4], rb[4], ra_hi[4], rb_hi[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int slot = (2 * ks + (lane >> 5)) ^ swz;
        ra[ks] = (wm * 32 + (lane & 31)) * 128 + (slot << 4);
        rb[ks] = BM * 128 + (wn * 64 + (lane & 31)) * 128 + (slot << 4);
        ra_hi[ks] = ra[ks] + 2 * STAGE;
        rb_hi[ks] = rb[ks] + 2 * STAGE;
        asm volatile("" : "+v"(ra_hi[ks]), "+v"(rb_hi[ks]));
    }
#define SB() __builtin_amdgcn_sched_barrier(0)
#define LDF(dst, base, S_, KS, I)                                                                                          \
    dst = *reinterpret_cast<const bf16x8_t*>(smem + ((S_) < 2 ? base[KS] + (S_) * STAGE : base##_hi[KS] + ((S_) - 2) * STAGE) + (I) * 4096)

    const int ntiles = g.gm * g.gn;
    for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        int tile_m, tile_n;
        tile_of_block(g, vb, tile_m, tile_n);
        const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
        uint32_t oa[2], ob[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = i * NT + tid, row = c >> 3, phys = c & 7;
            const int slot = phys ^ ((row >> 1) & 7);
            if (i < 2) {
                int64_t ga = m0 + row; if (ga > g.M - 1) ga = g.M - 1;
                oa[i] = (uint32_t)((ga * g.lda + slot * 8) * 2);
            }
            int64_t gb = n0 + row; if (gb > g.N - 1) gb = g.N - 1;
            ob[i] = (uint32_t)((gb * g.ldb + slot * 8) * 2);
        }
        // piece p (0..1 = A, 2..5 = B) of the stage holding K columns [step*64, +64) into ring slot S_
        auto dma = [&](int S_, int step, int p) {
            if (p < 2) {
                const int wbase = S_ * STAGE + (p * NT + wave * 64) * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(smem + wbase), 16, (int)oa[p], step * 128, 0, 0);
            } else {
                const int wbase = S_ * STAGE + BM * 128 + ((p - 2) * NT + wave * 64) * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(smem + wbase), 16, (int)ob[p - 2], step * 128, 0, 0);
            }
        };
        // ---- prologue: stages 0..2 in flight, 0 and 1 readable ----
#pragma unroll
        for (int st = 0; st < 3; ++st)
#pragma unroll
            for (int p = 0; p < 6; ++p) dma(st, st, p);
        f32x16_t acc[1][2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][ni][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8_t fm[4][1], fn[4][2];  // [k-step][32-row block]
        LDF(fn[0][0], rb, 0, 0, 0);
        LDF(fm[0][0], ra, 0, 0, 0);
        LDF(fn[0][1], rb, 0, 0, 1);
#define MMA(KS, MI, NI) acc[MI][NI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fn[KS][NI], fm[KS][MI], acc[MI][NI], 0, 0, 0)
// one K-tile = 4 k-steps x 2 MFMAs; every slot requests fragments of the NEXT k-step (k-step 0 of the next stage during the last one),
// the six DMA pieces of stage s+3 ride on the first six slots
#define KSTEPH(S, STEPV, DMA, NEXT, VMW)                                                                                                   \
    do {                                                                                                                                   \
        constexpr int SN = ((S) + 1) & 3, SD = ((S) + 3) & 3;                                                                              \
        MMA(0, 0, 0); SB(); LDF(fn[1][0], rb, S, 1, 0); SB(); LDF(fm[1][0], ra, S, 1, 0); SB(); if (DMA) dma(SD, (STEPV) + 3, 0); SB();      \
        MMA(0, 0, 1); SB(); LDF(fn[1][1], rb, S, 1, 1); SB(); if (DMA) dma(SD, (STEPV) + 3, 1); SB();                                      \
        MMA(1, 0, 0); SB(); LDF(fn[2][0], rb, S, 2, 0); SB(); LDF(fm[2][0], ra, S, 2, 0); SB(); if (DMA) dma(SD, (STEPV) + 3, 2); SB();      \
        MMA(1, 0, 1); SB(); LDF(fn[2][1], rb, S, 2, 1); SB(); if (DMA) dma(SD, (STEPV) + 3, 3); SB();                                      \
        MMA(2, 0, 0); SB(); LDF(fn[3][0], rb, S, 3, 0); SB(); LDF(fm[3][0], ra, S, 3, 0); SB(); if (DMA) dma(SD, (STEPV) + 3, 4); SB();      \
        MMA(2, 0, 1); SB(); LDF(fn[3][1], rb, S, 3, 1); SB(); if (DMA) dma(SD, (STEPV) + 3, 5); SB();                                      \
        MMA(3, 0, 0); SB(); if (NEXT) { LDF(fn[0][0], rb, SN, 0, 0); LDF(fm[0][0], ra, SN, 0, 0); } SB();                                  \
        MMA(3, 0, 1); SB(); if (NEXT) { LDF(fn[0][1], rb, SN, 0, 1); } SB();                                                              \
        asm volatile("s_waitcnt vmcnt(" #VMW ")" ::: "memory"); __builtin_amdgcn_s_barrier(); SB();                                        \
    } while (0)
        int st = 0;
        for (; st + 4 < nk; st += 4) {   // full trips: every step issues its stage st+3
            KSTEPH(0, st, true, true, 6);
            KSTEPH(1, st + 1, true, true, 6);
            KSTEPH(2, st + 2, true, true, 6);
            KSTEPH(3, st + 3, true, true, 6);
        }
        KSTEPH(0, st, true, true, 6);      // last trip: stage nk-1 is issued by its first step, then the queue drains
        KSTEPH(1, st + 1, false, true, 0);
        KSTEPH(2, st + 2, false, true, 0);
        KSTEPH(3, st + 3, false, false, 0);
#undef KSTEPH
#undef MMA

        // ---- epilogue: the ring is free (every DMA retired, every wave past the last barrier, every fragment read consumed) ----
        float part = 0.f;
        const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N) && !(g.dbg & 16) && (g.cdt == OTTER_F32 || g.wide);
        float* blk = reinterpret_cast<float*>(smem) + wave * (32 * EPI_LD);
        if (full) {
            if (g.cdt == OTTER_BF16) part = tail_wave1_full<EPI, true>(g, sgate, acc, blk, m0 + wm * 32, n0 + wn * 64, lane);
            else part = tail_wave1_full<EPI, false>(g, sgate, acc, blk, m0 + wm * 32, n0 + wn * 64, lane);
        } else {
            park_block(blk, acc[0][0], lane, 0);
            park_block(blk, acc[0][1], lane, 32);
            __builtin_amdgcn_wave_barrier();
            part += epilogue_stripe<EPI>(g, sgate, blk, m0 + wm * 32, n0 + wn * 64, lane);
            __builtin_amdgcn_wave_barrier();
        }
        block_partial<4, EPI>(g, part, reinterpret_cast<float*>(smem), vb);
        __syncthreads();  // the next tile's prologue DMA overwrites the stripes
    }
#undef LDF
#undef SB
}

// ------------------------------------------------------------------------------------------------------------
// exact-f32 MFMA kernel (parity mode): 64x64x32 tile, 4 waves (2x2), one 32x32 accumulator block per wave
// ------------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = 64, BN = 64, BK = 32, LD = BK + 1;
    __shared__ __attribute__((aligned(16))) float sm[2 * 32 * EPI_LD > (BM + BN) * LD ? 2 * 32 * EPI_LD : (BM + BN) * LD];
    float* As = sm;
    float* Bs = sm + BM * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    tile_of_block(g, (int)blockIdx.x, tile_m, tile_n);
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const float* __restrict__ A = (const float*)g.A;
    const float* __restrict__ B = (const float*)g.B;
    const int64_t K = g.K;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int64_t k0 = 0; k0 < K; k0 += BK) {
        // 64 rows x 32 floats = 512 float4 chunks per operand; 2 per thread
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = i * 256 + tid, row = c >> 3, q = c & 7;
            const int64_t k = k0 + q * 4;
            int64_t ga = m0 + row; if (ga > g.M - 1) ga = g.M - 1;
            int64_t gb = n0 + row; if (gb > g.N - 1) gb = g.N - 1;
            float4 va = make_float4(0, 0, 0, 0), vb = make_float4(0, 0, 0, 0);
            if (k < K) {  // K % 4 == 0 is enforced on the host
                va = *reinterpret_cast<const float4*>(A + ga * g.lda + k);
                vb = *reinterpret_cast<const float4*>(B + gb * g.ldb + k);
            }
            float* da = As + row * LD + q * 4;
            float* db = Bs + row * LD + q * 4;
            da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
            db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = Bs[(wn * 32 + (lane & 31)) * LD + kk + (lane >> 5)];  // a-operand = B rows (n)
            const float b = As[(wm * 32 + (lane & 31)) * LD + kk + (lane >> 5)];  // b-operand = A rows (m)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const float s = g.gate ? tanhf(*g.gate) : 1.0f;
    // the two waves of a row pair (wn = 0, 1) share one 64-column stripe
    float* blk = sm + wm * (32 * EPI_LD);
    park_block(blk, acc, lane, wn * 32);
    __syncthreads();
    float part = 0.f;
    {
        // 32 rows x 64 columns handled by the pair's 128 lanes: lane id within the pair = wn*64 + lane
        const int pl = wn * 64 + lane;
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int r = (pl >> 4) + 8 * it;
            const int c = (pl & 15) * 4;
            const int64_t m = m0 + wm * 32 + r, n = n0 + c;
            if (m < g.M && n < g.N) {
                const float4 t = *reinterpret_cast<const float4*>(blk + r * EPI_LD + c);
                float v[4] = {t.x, t.y, t.z, t.w};
                part += epilogue4<EPI>(g, s, m, n, v);
            }
        }
    }
    block_partial<4, EPI>(g, part, sm, (int)blockIdx.x);
}

__global__ void reduce_partials_kernel(const float* __restrict__ partial, int64_t n, const float* __restrict__ gate,
                                       float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    float t = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) t += partial[i];
    t = wave_sum(t);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = red[0] + red[1] + red[2] + red[3];
        if (gate) {
            const float th = tanhf(*gate);
            s *= (1.0f - th * th);
        }
        out[0] = accumulate ? out[0] + s : s;
    }
}

// ---- configuration choice (shared by the launcher and otter_gemm_num_partials) ----
int g_variant = 0;
int g_debug = 0;
int g_order = 0;
int g_narrow_epilogue = 0;  // A/B hook (otter_gemm_set_debug bit 256): force the 4-wide fused tail
// Cross-tile form of variant 26 (round 6): otter_gemm_set_debug bits 14-15 = 0 default / 1 off / 2 on, bits 16-22 = K order + start phase (GemmArgs::korder), bit 23 = use them;
// the process default comes from OTTER_GEMM_XT / OTTER_GEMM_KORDER (read once; A/B switches of bench.py and tools/)
int g_xt_force = 0;
int g_korder_force = -1;
#ifndef OTTER_T4_XT_DEFAULT
#define OTTER_T4_XT_DEFAULT 1
#endif
#ifndef OTTER_T4_KORDER_DEFAULT
#define OTTER_T4_KORDER_DEFAULT 3   // K-tiles of a tile walked from a quarter given by its N panel: -1..-4 % on cold operands (profiles/r06_xt_ab*.txt);
                                    // mode 3 and not the XCD modes: a row of C keeps one summation order wherever it sits in the batch (DP invariant)
#endif
bool t4_xt_on() {
    static int dflt = -1;
    if (dflt < 0) { const char* e = getenv("OTTER_GEMM_XT"); dflt = e ? (e[0] != '0') : OTTER_T4_XT_DEFAULT; }
    return g_xt_force ? g_xt_force == 2 : dflt == 1;
}
int t4_korder() {
    static int dflt = -1;
    if (dflt < 0) { const char* e = getenv("OTTER_GEMM_KORDER"); dflt = e ? (atoi(e) & 127) : OTTER_T4_KORDER_DEFAULT; }
    return g_korder_force >= 0 ? g_korder_force : dflt;
}
enum Cfg { CFG_128 = 1, CFG_256 = 2, CFG_256_GLDS = 3, CFG_MS4 = 4, CFG_MS5 = 5, CFG_PH = 6, CFG_PHC = 7, CFG_WS = 8, CFG_PHB = 9, CFG_PHCB = 10, CFG_PHRB = 11, CFG_MS5B = 12, CFG_PHLB = 13, CFG_PHIB = 14, CFG_PH2B = 15, CFG_PHDB = 16, CFG_Q4 = 17, CFG_R4 = 18, CFG_R4B = 19, CFG_R4C = 20, CFG_R4P = 21, CFG_R4M = 22, CFG_R4N = 23, CFG_S4 = 25, CFG_T4 = 26, CFG_T4B = 27, CFG_T4C = 28, CFG_T4M = 29, CFG_S4H = 30, CFG_F32 = 100 };

// wide: an operand spans >= 4 GB, so the kernels that address it with 32-bit byte offsets are out
int pick_cfg(int64_t M, int64_t N, int64_t K, int ab_dtype, bool wide = false) {
    if (ab_dtype == OTTER_F32) return CFG_F32;
    int v = g_variant;
    if (v == 0) {
        // interleaved A/B medians on MI355X (tools/gemm_ab.py): round 2's one-wave-per-SIMD kernel with the
        // register-resident K-tile (variant 18) beats round 1's balanced 8-wave phased schedule (variant 13) on all three
        // FFN shapes (1.29 / 1.49 / 1.43 PF vs 1.19 / 1.39 / 1.37 on one box); it needs K % 128 == 0, else 13 stays
        if (cdiv64(M, 256) * cdiv64(N, 256) >= 192) v = (K % 128 == 0 && !wide) ? CFG_T4 : CFG_PHLB;
        else if (K % 256 == 0 && K >= 512 && !wide && cdiv64(M, 128) * cdiv64(N, 128) <= 512) v = CFG_S4;   // few tiles, long reduction: the ring
        else v = CFG_128;
    }
    if ((v == CFG_MS4 || v == CFG_MS5 || v == CFG_MS5B) && (K % 32 != 0)) v = CFG_256_GLDS;
    if (v == CFG_WS && K % 64 != 0) v = CFG_256;
    const bool ph = v == CFG_PH || v == CFG_PHC || v == CFG_PHB || v == CFG_PHCB || v == CFG_PHRB || v == CFG_PHLB || v == CFG_PHIB || v == CFG_PH2B || v == CFG_PHDB;
    if (ph && (K % 64 != 0 || wide)) v = (K % 64 == 0) ? CFG_256_GLDS : CFG_256;
    if ((v == CFG_Q4 || v == CFG_R4 || v == CFG_R4B || v == CFG_R4C || v == CFG_R4P || v == CFG_R4M || v == CFG_R4N || v == CFG_T4 || v == CFG_T4B || v == CFG_T4C || v == CFG_T4M) && (K % 128 != 0 || wide)) v = (K % 64 == 0 && !wide) ? CFG_PHLB : ((K % 64 == 0) ? CFG_256_GLDS : CFG_256);
    if ((v == CFG_S4 || v == CFG_S4H) && (K % 256 != 0 || wide)) v = CFG_128;
    if (v == CFG_MS5B && wide) v = CFG_MS5;
    if (v == CFG_256_GLDS && (K % 64 != 0)) v = CFG_256;
    return v;
}
void cfg_tiles(int cfg, int& bm, int& bn) {
    if (cfg == CFG_F32) { bm = 64; bn = 64; }
    else if (cfg == CFG_128 || cfg == CFG_S4) { bm = 128; bn = 128; }
    else if (cfg == CFG_S4H) { bm = 64; bn = 128; }
    else { bm = 256; bn = 256; }
}

// Persistent grids: one workgroup per CU of THIS device (hipDeviceProp_t::multiProcessorCount, read once per process -- one
// process drives one GPU), capped by otter_gemm_set_cu_budget: with a DP reducer live the GEMMs leave a few CUs to RCCL's
// kernels so the all-reduce of bucket i runs beside the dgrad GEMMs of the layers below instead of waiting for the gaps
// between launches (every persistent GEMM otherwise pins all CUs with one 512-register wave per SIMD).
int g_cu_budget = 0;   // 0 = all
int g_persistent = 1;  // 0: the large-grid kernel (variant 26) runs one workgroup per tile (otter_gemm_set_persistent)
unsigned device_cus() {
    static unsigned n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
                ? (unsigned)p.multiProcessorCount : 256u;
    }
    return n;
}
unsigned persistent_cus() {
    const unsigned n = device_cus();
    return (g_cu_budget > 0 && (unsigned)g_cu_budget < n) ? (unsigned)g_cu_budget : n;
}

// ---- profiling hook (bench.py roofline object) ----
struct Prof {
    bool armed = false;
    int64_t M = 0, N = 0, K = 0;
    int max_events = 0, n = 0;
    hipEvent_t* start = nullptr;
    hipEvent_t* stop = nullptr;
    unsigned char* kmajor = nullptr;   // per event: the launch had a K-major operand
} g_prof;

template <typename KernelT>
int set_smem(KernelT kernel, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) OTTER_FAIL(OTTER_ERR_LAUNCH, "hipFuncSetAttribute(%d B LDS): %s", bytes, hipGetErrorString(e));
    return OTTER_OK;
}

template <int BM, int BN, int WM, int WN, bool GLDS, int EPI>
int launch_one(dim3 grid, hipStream_t st, const GemmArgs& g) {
    static bool once = false;
    const int smem = 2 * (BM + BN) * 128;
    if (!once) {
        int rc = set_smem(gemm_bf16_kernel<BM, BN, WM, WN, GLDS, EPI>, smem);
        if (rc) return rc;
        once = true;
    }
    // persistent grid: one resident wave of blocks (256 CUs x blocks that fit per CU), rounded to a multiple of 8
    const unsigned per_cu = (BM == 256) ? 1u : 2u;
    unsigned pg = persistent_cus() * per_cu;
    if (grid.x < pg) pg = grid.x;
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, GLDS, EPI>), dim3(pg), dim3(WM * WN * 64), smem, st, g);
    return OTTER_OK;
}

template <int EPI>
int launch_epi(int cfg, dim3 grid, hipStream_t st, const GemmArgs& g) {
    if (cfg == CFG_F32) {
        hipLaunchKernelGGL(gemm_f32_kernel<EPI>, grid, dim3(256), 0, st, g);
        return OTTER_OK;
    }
    if (cfg == CFG_128) return launch_one<128, 128, 2, 2, false, EPI>(grid, st, g);
    if (cfg == CFG_256) return launch_one<256, 256, 2, 4, false, EPI>(grid, st, g);
#ifdef OTTER_EXPERIMENTAL
    if (cfg == CFG_WS) {
        static bool once = false;
        const int smem = 2 * (256 + 256) * 128;
        if (!once) { int rc = set_smem(gemm_bf16_ws_kernel<EPI>, smem); if (rc) return rc; once = true; }
        unsigned pg = grid.x < persistent_cus() ? grid.x : persistent_cus();
        hipLaunchKernelGGL((gemm_bf16_ws_kernel<EPI>), dim3(pg), dim3(768), smem, st, g);
        return OTTER_OK;
    }
#endif
    if (cfg == CFG_PH || cfg == CFG_PHC || cfg == CFG_PHB || cfg == CFG_PHCB || cfg == CFG_PHRB || cfg == CFG_PHLB || cfg == CFG_PHIB || cfg == CFG_PH2B || cfg == CFG_PHDB) {
        const int smem = 2 * (256 + 256) * 128;
        unsigned pg = grid.x < persistent_cus() ? grid.x : persistent_cus();
#define LAUNCH_PH(CNT_, BUF_)                                                                                              \
    do {                                                                                                                   \
        static bool once = false;                                                                                          \
        if (!once) { int rc = set_smem(gemm_bf16_ph_kernel<EPI, CNT_, BUF_>, smem); if (rc) return rc; once = true; }      \
        hipLaunchKernelGGL((gemm_bf16_ph_kernel<EPI, CNT_, BUF_>), dim3(pg), dim3(512), smem, st, g);                      \
    } while (0)
#ifdef OTTER_EXPERIMENTAL
        if (cfg == CFG_PH) LAUNCH_PH(0, false);
        else if (cfg == CFG_PHC) LAUNCH_PH(1, false);
        else if (cfg == CFG_PHB) LAUNCH_PH(0, true);
        else if (cfg == CFG_PHCB) LAUNCH_PH(1, true);
        else if (cfg == CFG_PHRB) LAUNCH_PH(2, true);
        else if (cfg == CFG_PHLB) LAUNCH_PH(3, true);
        else if (cfg == CFG_PHIB) LAUNCH_PH(4, true);
        else if (cfg == CFG_PH2B) LAUNCH_PH(5, true);
        else LAUNCH_PH(6, true);
#else
        LAUNCH_PH(3, true);   // variant 13, the only phased schedule of the product build (pick_cfg admits no other)
#endif
#undef LAUNCH_PH
        return OTTER_OK;
    }
#ifdef OTTER_EXPERIMENTAL
    if (cfg == CFG_MS4 || cfg == CFG_MS5 || cfg == CFG_MS5B) {
        const int smem = (cfg == CFG_MS4 ? 4 : 5) * 32768;
#define LAUNCH_MS(NS_, BUF_)                                                                                               \
    do {                                                                                                                   \
        static bool once = false;                                                                                          \
        if (!once) { int rc = set_smem(gemm_bf16_ms_kernel<NS_, EPI, BUF_>, smem); if (rc) return rc; once = true; }       \
        hipLaunchKernelGGL((gemm_bf16_ms_kernel<NS_, EPI, BUF_>), grid, dim3(256), smem, st, g);                           \
    } while (0)
        if (cfg == CFG_MS4) LAUNCH_MS(4, false);
        else if (cfg == CFG_MS5) LAUNCH_MS(5, false);
        else LAUNCH_MS(5, true);
#undef LAUNCH_MS
        return OTTER_OK;
    }
    if (cfg == CFG_R4 || cfg == CFG_R4B || cfg == CFG_R4C || cfg == CFG_R4P || cfg == CFG_R4M || cfg == CFG_R4N) {
        const int smem = TAIL_LDS_BYTES > 2 * 65536 ? TAIL_LDS_BYTES : 2 * 65536;  // two K-tile buffers; the tail's parking buffers alias them
        unsigned pg = grid.x < persistent_cus() ? grid.x : persistent_cus();
#define LAUNCH_R4(SCH_, PF_)                                                                                               \
    do {                                                                                                                   \
        static bool once = false;                                                                                          \
        if (!once) { int rc = set_smem(gemm_bf16_r4_kernel<EPI, SCH_, PF_>, smem); if (rc) return rc; once = true; }       \
        hipLaunchKernelGGL((gemm_bf16_r4_kernel<EPI, SCH_, PF_>), dim3(pg), dim3(256), smem, st, g);                       \
    } while (0)
        if (cfg == CFG_R4) LAUNCH_R4(0, false);
        else if (cfg == CFG_R4B) LAUNCH_R4(1, false);
        else if (cfg == CFG_R4C) LAUNCH_R4(2, false);
        else if (cfg == CFG_R4M) LAUNCH_R4(3, false);
        else if (cfg == CFG_R4N) LAUNCH_R4(4, false);
        else LAUNCH_R4(0, true);
#undef LAUNCH_R4
        return OTTER_OK;
    }
#endif
    if (cfg == CFG_T4 || cfg == CFG_T4B || cfg == CFG_T4C || cfg == CFG_T4M) {
        const int smem = TAIL_LDS_BYTES > 2 * 65536 ? TAIL_LDS_BYTES : 2 * 65536;
        unsigned pg = grid.x < persistent_cus() ? grid.x : persistent_cus();
        // one workgroup per tile: asked for by this call (otter_epilogue_args::grid_mode = 2), by the process default
        // (otter_gemm_set_persistent(0)) when the call leaves it open, or by debug bit 13
        const bool per_tile = g.grid_mode == OTTER_GRID_PER_TILE || (g.grid_mode == OTTER_GRID_DEFAULT && !g_persistent);
        if ((g_order & 0x10) || per_tile) pg = grid.x;
#define LAUNCH_T4(SCH_)                                                                                                    \
    do {                                                                                                                   \
        static bool once = false;                                                                                          \
        if (!once) { int rc = set_smem(gemm_bf16_t4_kernel<EPI, SCH_>, smem); if (rc) return rc; once = true; }            \
        hipLaunchKernelGGL((gemm_bf16_t4_kernel<EPI, SCH_>), dim3(pg), dim3(256), smem, st, g);                            \
    } while (0)
#define LAUNCH_T4T(TA_, TB_)                                                                                               \
    do {                                                                                                                   \
        static bool once = false;                                                                                          \
        if (!once) { int rc = set_smem(gemm_bf16_t4_kernel<EPI, 0, TA_, TB_>, smem); if (rc) return rc; once = true; }     \
        hipLaunchKernelGGL((gemm_bf16_t4_kernel<EPI, 0, TA_, TB_>), dim3(pg), dim3(256), smem, st, g);                     \
    } while (0)
// cross-tile form: the ring (2 x 64 KB) + one 8 KB parking stripe per wave above it = all 160 KB of the CU's LDS
#define LAUNCH_T4X(TA_, TB_)                                                                                               \
    do {                                                                                                                   \
        static bool once = false;                                                                                          \
        constexpr int smem_x = 2 * 65536 + 4 * XPARK_BYTES;                                                                \
        if (!once) { int rc = set_smem(gemm_bf16_t4_kernel<EPI, 0, TA_, TB_, true>, smem_x); if (rc) return rc; once = true; } \
        hipLaunchKernelGGL((gemm_bf16_t4_kernel<EPI, 0, TA_, TB_, true>), dim3(pg), dim3(256), smem_x, st, g);             \
    } while (0)
        // (the cross-tile form peels two K-tiles at each end of a tile's K loop: >= 4 K-tiles)
        // K-contiguous operands only: with a K-major operand the cross-tile form measured SLOWER (dW2 404 vs 396 us, profiles/r06_xt_ab*.txt):
        // its tail is shorter, but the stores it leaves in flight hold back the in-order vmcnt waits of the next tile's first K-tiles by more
        if (cfg == CFG_T4 && t4_xt_on() && !g.ta && !g.tb && g.K >= 256) {
            LAUNCH_T4X(false, false);
            return OTTER_OK;
        }
#undef LAUNCH_T4X
        if (g.ta || g.tb) {   // K-major operands: the backward products (dW = dy^T x: both; dx = dy W: B)
            if (g.ta && g.tb) LAUNCH_T4T(true, true);
            else if (g.tb) LAUNCH_T4T(false, true);
            else LAUNCH_T4T(true, false);
            return OTTER_OK;
        }
#undef LAUNCH_T4T
#ifdef OTTER_EXPERIMENTAL
        if (cfg == CFG_T4) LAUNCH_T4(0);
        else if (cfg == CFG_T4B) LAUNCH_T4(1);
        else if (cfg == CFG_T4C) LAUNCH_T4(2);
        else LAUNCH_T4(3);
#else
        LAUNCH_T4(0);
#endif
#undef LAUNCH_T4
        return OTTER_OK;
    }
    if (cfg == CFG_S4H) {
        static bool once = false;
        const int smem = 4 * 24576;  // the ring; the tail's parking buffers (4 waves x 1 stripe) alias it
        if (!once) { int rc = set_smem(gemm_bf16_s4h_kernel<EPI>, smem); if (rc) return rc; once = true; }
        unsigned pg = grid.x < persistent_cus() ? grid.x : persistent_cus();
        hipLaunchKernelGGL((gemm_bf16_s4h_kernel<EPI>), dim3(pg), dim3(256), smem, st, g);
        return OTTER_OK;
    }
    if (cfg == CFG_S4) {
        static bool once = false;
        const int smem = 4 * 32768;  // the ring; the tail's parking buffers (4 waves x 2 stripes = 69632 B) alias it
        if (!once) { int rc = set_smem(gemm_bf16_s4_kernel<EPI>, smem); if (rc) return rc; once = true; }
        unsigned pg = grid.x < persistent_cus() ? grid.x : persistent_cus();
        hipLaunchKernelGGL((gemm_bf16_s4_kernel<EPI>), dim3(pg), dim3(256), smem, st, g);
        return OTTER_OK;
    }
#ifdef OTTER_EXPERIMENTAL
    if (cfg == CFG_Q4) {
        static bool once = false;
        const int smem = TAIL_LDS_BYTES;  // 4 x 32 KB ring; the tail's parking buffers (139264 B) alias it
        if (!once) { int rc = set_smem(gemm_bf16_q4_kernel<EPI>, smem); if (rc) return rc; once = true; }
        unsigned pg = grid.x < persistent_cus() ? grid.x : persistent_cus();
        hipLaunchKernelGGL((gemm_bf16_q4_kernel<EPI>), dim3(pg), dim3(256), smem, st, g);
        return OTTER_OK;
    }
#endif
    return launch_one<256, 256, 2, 4, true, EPI>(grid, st, g);
}

int launch_cfg(int cfg, int kind, dim3 grid, hipStream_t st, const GemmArgs& g) {
    switch (kind) {
        case OTTER_EPI_STORE: return launch_epi<OTTER_EPI_STORE>(cfg, grid, st, g);
        case OTTER_EPI_GELU: return launch_epi<OTTER_EPI_GELU>(cfg, grid, st, g);
        case OTTER_EPI_SCALE_RES: return launch_epi<OTTER_EPI_SCALE_RES>(cfg, grid, st, g);
        default: return launch_epi<OTTER_EPI_GATE_BWD>(cfg, grid, st, g);
    }
}

}  // namespace

extern "C" {

int otter_abi_version(void) { return OTTER_ABI_VERSION; }
const char* otter_last_error(void) { return g_otter_err; }

int otter_device_check(void) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess)
        OTTER_FAIL(OTTER_ERR_LAUNCH, "no HIP device");
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        OTTER_FAIL(OTTER_ERR_UNSUPPORTED, "device is %s, this library is built for gfx950 only", p.gcnArchName);
    return p.multiProcessorCount;
}

int otter_gemm_set_cu_budget(int cus) {
    if (cus < 0) OTTER_FAIL(OTTER_ERR_ARG, "gemm cu budget %d", cus);
    g_cu_budget = (cus > 0 && cus < 8) ? 8 : cus;   // at least one workgroup per XCD
    return (int)persistent_cus();
}

int otter_gemm_set_persistent(int on) {
    g_persistent = on ? 1 : 0;
    return OTTER_OK;
}

int otter_gemm_variant_available(int variant) {
    if (variant < 0 || variant > 30 || variant == 24) return 0;
#ifdef OTTER_EXPERIMENTAL
    return 1;
#else
    // product build: auto (0), the generic 128^2 / 256^2 kernels (1-3), the phased fallback for K % 128 != 0 (13), the small-grid ring
    // (25) and the default (26); every other schedule is compiled into the tools-only experimental library
    return variant <= 3 || variant == CFG_PHLB || variant == CFG_S4 || variant == CFG_T4 || variant == CFG_S4H;
#endif
}

int otter_gemm_set_variant(int variant) {
    if (variant < 0 || variant > 30) OTTER_FAIL(OTTER_ERR_ARG, "gemm variant %d", variant);
    if (!otter_gemm_variant_available(variant))
        OTTER_FAIL(OTTER_ERR_UNSUPPORTED, "gemm variant %d is in the experimental build only (python -m otter_amd.build --experimental)", variant);
    g_variant = variant;
    return OTTER_OK;
}

int otter_gemm_read_timeline(unsigned long long* out, int n) {
    OTTER_REQUIRE(out && n > 0 && n <= 2 * 4 * 8 * 8, "gemm_read_timeline: n");
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_timeline), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) OTTER_FAIL(OTTER_ERR_LAUNCH, "hipMemcpyFromSymbol: %s", hipGetErrorString(e));
    return OTTER_OK;
}

int otter_gemm_set_debug(int flags) {
    g_debug = flags & 255;  // bit 64: tile-phase timeline of variants 18-20 (otter_gemm_read_timeline)
    g_narrow_epilogue = (flags & 256) ? 1 : 0;
    g_order = (flags >> 9) & 31;  // tile-order override (bits 9-12, see tile_of_block); bit 13: non-persistent launch of variant 26
    g_xt_force = (flags >> 14) & 3;                       // cross-tile form of variant 26: 0 default, 1 off, 2 on
    g_korder_force = (flags & (1 << 23)) ? ((flags >> 16) & 127) : -1;   // bit 23: bits 16-22 override the K order / start phase of that form
    return OTTER_OK;
}

int64_t otter_gemm_num_partials(int64_t M, int64_t N, int ab_dtype) {
    int bm, bn;
    cfg_tiles(pick_cfg(M, N, 64, ab_dtype), bm, bn);
    return cdiv64(M, bm) * cdiv64(N, bn);
}

int otter_gemm_kmajor_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int a_kmajor, int b_kmajor, int ab_dtype) {
    if (ab_dtype != OTTER_BF16 || g_variant != 0) return 0;
    if (K < 128) return 0;
    if (K % 128 != 0 && !(a_kmajor && b_kmajor)) return 0;               // K-contiguous operands: whole K-tile pairs (variant 26's pipeline);
                                                                         // both K-major: any K (rows past K read as zeros)
    if (cdiv64(M, 256) * cdiv64(N, 256) < 192) return 0;                 // small grids run the ring kernel on K-contiguous operands
    if (a_kmajor && (M % 8 != 0 || lda % 8 != 0)) return 0;              // 16-byte chunks along the rows
    if (b_kmajor && (N % 8 != 0 || ldb % 8 != 0)) return 0;
    const int64_t span_a = a_kmajor ? ((K - 1) * lda + M) * 2 : ((M - 1) * lda + K) * 2;
    const int64_t span_b = b_kmajor ? ((K - 1) * ldb + N) * 2 : ((N - 1) * ldb + K) * 2;
    return span_a < (int64_t(1) << 32) && span_b < (int64_t(1) << 32);
}

static int gemm_impl(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C, int64_t ldc, int64_t M, int64_t N,
                     int64_t K, int ab_dtype, int c_dtype, const otter_epilogue_args* epi, void* stream);

int otter_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                  int ab_dtype, int c_dtype, const otter_epilogue_args* epi, void* stream) {
    return gemm_impl(A, lda, 0, B, ldb, 0, C, ldc, M, N, K, ab_dtype, c_dtype, epi, stream);
}

int otter_gemm(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C, int64_t ldc, int64_t M, int64_t N,
               int64_t K, int ab_dtype, int c_dtype, const otter_epilogue_args* epi, void* stream) {
    if (a_kmajor || b_kmajor)
        OTTER_REQUIRE(otter_gemm_kmajor_supported(M, N, K, lda, ldb, a_kmajor, b_kmajor, ab_dtype),
                      "gemm: K-major operands need bf16, K %% 128 == 0 (any K when both are K-major), >= 192 tiles of 256x256, M / N / ld %% 8 == 0 and < 4 GB per operand "
                      "(M=%ld N=%ld K=%ld): ask otter_gemm_kmajor_supported first and transpose otherwise", (long)M, (long)N, (long)K);
    return gemm_impl(A, lda, a_kmajor ? 1 : 0, B, ldb, b_kmajor ? 1 : 0, C, ldc, M, N, K, ab_dtype, c_dtype, epi, stream);
}

static bool s4h_off() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OTTER_NO_S4H"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

static int gemm_impl(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor, void* C, int64_t ldc, int64_t M, int64_t N,
                     int64_t K, int ab_dtype, int c_dtype, const otter_epilogue_args* epi, void* stream) {
    OTTER_REQUIRE(A && B && C && epi, "gemm: null pointer");
    OTTER_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    const int kal = ab_dtype == OTTER_BF16 ? 8 : 4;
    OTTER_REQUIRE(((a_kmajor && b_kmajor) || K % kal == 0) && lda % kal == 0 && ldb % kal == 0, "gemm: K=%ld lda=%ld ldb=%ld must be multiples of %d",
                  (long)K, (long)lda, (long)ldb, kal);   // (both operands K-major: K is a row count, any value)
    OTTER_REQUIRE(N % 4 == 0 && ldc % 4 == 0, "gemm: N=%ld ldc=%ld must be multiples of 4", (long)N, (long)ldc);
    OTTER_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0, "gemm: 16-byte alignment");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.cdt = c_dtype;
    g.M = M; g.N = N; g.K = K;
    g.kind = epi->kind; g.accumulate = epi->accumulate; g.gate = epi->gate;
    g.R = epi->R; g.ldr = epi->ldr; g.rdt = epi->r_dtype;
    g.C2 = epi->C2; g.ldc2 = epi->ldc2;
    g.aux = epi->aux; g.ldaux = epi->ldaux; g.auxdt = epi->aux_dtype; g.aux_gelu = epi->aux_is_gelu_input;
    g.partial = epi->partial;
    OTTER_REQUIRE(epi->grid_mode >= OTTER_GRID_DEFAULT && epi->grid_mode <= OTTER_GRID_PER_TILE, "gemm: grid_mode %d", epi->grid_mode);
    g.grid_mode = epi->grid_mode;
    switch (g.kind) {
        case OTTER_EPI_STORE:
            OTTER_REQUIRE(!g.accumulate || c_dtype == OTTER_F32, "gemm: accumulate needs an f32 C");
            OTTER_REQUIRE(!g.partial || c_dtype == OTTER_F32, "gemm: sum-of-squares partials of a plain store need an f32 C");
            break;
        case OTTER_EPI_GELU:
            OTTER_REQUIRE(!g.C2 || g.ldc2 % 4 == 0, "gemm: ldc2 %% 4");
            OTTER_REQUIRE(g.aux_gelu == 0 || g.aux_gelu == 3, "gemm: a GELU launch takes aux_is_gelu_input 0 (C2 = pre-activation) or 3 (C2 = GELU'), got %d", g.aux_gelu);
            break;
        case OTTER_EPI_SCALE_RES:
            OTTER_REQUIRE(g.R && g.ldr % 4 == 0, "gemm: SCALE_RES needs R with ldr %% 4 == 0");
            OTTER_REQUIRE(!g.partial, "gemm: partials are written by the STORE (fp32 C) and GATE_BWD epilogues only");
            break;
        case OTTER_EPI_GATE_BWD:
            OTTER_REQUIRE(g.aux && g.ldaux % 4 == 0, "gemm: GATE_BWD needs aux with ldaux %% 4 == 0");
            OTTER_REQUIRE(g.aux_gelu >= 0 && g.aux_gelu <= 3, "gemm: aux activation %d (0 = identity, 1 = erf GELU, 2 = squared ReLU, 3 = aux is the stashed derivative)", g.aux_gelu);
            OTTER_REQUIRE(g.aux_gelu != 3 || !g.partial, "gemm: a stashed derivative (aux_is_gelu_input 3) cannot give the gate partial sum(acc * f(aux))");
            break;
        default:
            OTTER_FAIL(OTTER_ERR_ARG, "gemm: unknown epilogue %d", g.kind);
    }
    const int64_t esz = ab_dtype == OTTER_BF16 ? 2 : 4;
    const bool kmaj = a_kmajor || b_kmajor;
    const bool wide = !kmaj && (((M - 1) * lda + K) * esz >= (int64_t(1) << 32) || ((N - 1) * ldb + K) * esz >= (int64_t(1) << 32));
    int cfg = kmaj ? CFG_T4 : pick_cfg(M, N, K, ab_dtype, wide);
    // round 4: few 128 x 128 tiles of the ring kernel -> half-height tiles (variant 30), so that the skinny products (128 tiles or fewer on
    // 256 CUs) stream on twice the CUs.  Not for a launch that writes per-block gate partials (their count follows otter_gemm_num_partials).
    // OTTER_NO_S4H=1 (read once): A/B switch.
    if (cfg == CFG_S4 && g_variant == 0 && !s4h_off() && cdiv64(M, 128) * cdiv64(N, 128) <= 160 && !g.partial)
        cfg = CFG_S4H;
    if (cfg == CFG_S4H && g.partial) cfg = CFG_S4;   // (a forced variant 30)
    g.ta = a_kmajor; g.tb = b_kmajor;
    int bm, bn;
    cfg_tiles(cfg, bm, bn);
    g.dbg = g_debug;
    g.order = g_order;
    g.korder = t4_korder();
    // (fp32 outputs stay on the 4-wide tail: there a lane's 4 columns are already a 16-byte store and 16 lanes cover a
    //  whole 256-byte row run; the 8-wide form would split every row into interleaved 16-byte halves: measured +15 %)
    g.wide = (c_dtype == OTTER_BF16 && N % 8 == 0 && ldc % 8 == 0 && (!g.C2 || g.ldc2 % 8 == 0) && (g.kind != OTTER_EPI_SCALE_RES || g.ldr % 8 == 0) &&
              (g.kind != OTTER_EPI_GATE_BWD || g.ldaux % 8 == 0) && !g_narrow_epilogue) ? 1 : 0;
    g.gm = (int)cdiv64(M, bm);
    g.gn = (int)cdiv64(N, bn);
    const dim3 grid((unsigned)(g.gm * g.gn));
    hipStream_t st = (hipStream_t)stream;
    const bool prof = g_prof.armed && g_prof.M == M && g_prof.N == N && g_prof.K == K && g_prof.n < g_prof.max_events;
    if (prof) hipEventRecord(g_prof.start[g_prof.n], st);
    int rc = launch_cfg(cfg, g.kind, grid, st, g);
    if (rc) return rc;
    if (prof) { g_prof.kmajor[g_prof.n] = (unsigned char)(kmaj ? 1 : 0); hipEventRecord(g_prof.stop[g_prof.n++], st); }
    OTTER_CHECK_LAUNCH("gemm");
    return OTTER_OK;
}

int otter_reduce_partials(const float* partial, int64_t n, const float* gate, float* out, int accumulate, void* stream) {
    OTTER_REQUIRE(partial && out && n > 0, "reduce_partials: bad args");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, gate, out, accumulate);
    OTTER_CHECK_LAUNCH("reduce_partials");
    return OTTER_OK;
}

int otter_prof_arm_gemm(int64_t M, int64_t N, int64_t K, int max_events) {
    otter_prof_disarm();
    OTTER_REQUIRE(max_events > 0 && max_events <= 65536, "prof: max_events");
    g_prof.start = new hipEvent_t[max_events];
    g_prof.stop = new hipEvent_t[max_events];
    g_prof.kmajor = new unsigned char[max_events]();
    for (int i = 0; i < max_events; ++i) {
        hipEventCreate(&g_prof.start[i]);
        hipEventCreate(&g_prof.stop[i]);
    }
    g_prof.M = M; g_prof.N = N; g_prof.K = K;
    g_prof.max_events = max_events;
    g_prof.n = 0;
    g_prof.armed = true;
    return OTTER_OK;
}

int otter_prof_disarm(void) {
    if (g_prof.start) {
        for (int i = 0; i < g_prof.max_events; ++i) {
            hipEventDestroy(g_prof.start[i]);
            hipEventDestroy(g_prof.stop[i]);
        }
        delete[] g_prof.start;
        delete[] g_prof.stop;
        delete[] g_prof.kmajor;
    }
    g_prof = Prof();
    return OTTER_OK;
}

int otter_prof_collect_split(int* count, double* total_ms, int* count_kmajor, double* kmajor_ms) {
    OTTER_REQUIRE(count && total_ms, "prof_collect: null");
    double tot = 0.0, km = 0.0;
    int nk = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        hipEventSynchronize(g_prof.stop[i]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]);
        tot += ms;
        if (g_prof.kmajor[i]) { km += ms; ++nk; }
    }
    *count = g_prof.n;
    *total_ms = tot;
    if (count_kmajor) *count_kmajor = nk;
    if (kmajor_ms) *kmajor_ms = km;
    g_prof.n = 0;
    return OTTER_OK;
}

int otter_prof_collect(int* count, double* total_ms) { return otter_prof_collect_split(count, total_ms, nullptr, nullptr); }

}  // extern "C"
