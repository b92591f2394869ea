// norm.hip -- LayerNorm / RMSNorm forward + backward for gfx950.
//
// HBM-bound row kernels: one wave64 per row, the row lives in registers (8 elements = 16 B (bf16) / 32 B (f32) per
// lane per chunk, chunk c of a row covers columns [c*512 + lane*8, +8)), two-pass statistics from registers
// (mean, then sum (x-mean)^2 -- the same formulation as ATen's CPU layer_norm, not E[x^2]-mean^2), wave64 butterfly
// reductions, no LDS.  Algorithmic bytes per row: D*(sizeof x + sizeof y) forward; D*(dy + x + dx) backward, plus a
// second column-parallel pass over dy and x for dgamma/dbeta (deterministic two-stage reduction, no atomics).
#include "common.h"

namespace {

__device__ __forceinline__ void load8(const void* p, int64_t idx, int dt, float (&v)[8]) {
    if (dt == OTTER_BF16) Vec8<bf16_t>::load((const bf16_t*)p + idx, v);
    else Vec8<float>::load((const float*)p + idx, v);
}
__device__ __forceinline__ void store8(void* p, int64_t idx, int dt, const float (&v)[8]) {
    if (dt == OTTER_BF16) Vec8<bf16_t>::store((bf16_t*)p + idx, v);
    else Vec8<float>::store((float*)p + idx, v);
}
__device__ __forceinline__ float round_to(float v, int dt) { return dt == OTTER_BF16 ? bf2f(f2bf(v)) : v; }

// NCH = chunks (of 512 columns) per row held by a wave; D <= NCH*512, D % 8 == 0.
template <int NCH, bool RMS>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const void* __restrict__ x, int xdt, const void* __restrict__ gamma,
                                                       const void* __restrict__ beta, int wdt, void* __restrict__ y, int ydt,
                                                       otter_rowmap ymap, void* __restrict__ y2, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int64_t rows, int D, float eps,
                                                       const void* __restrict__ delta, int ddt, void* __restrict__ xsum) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < D) {
            load8(x, row * D + col, xdt, v[c]);
            if (delta) {  // fused residual add: xsum = x + delta (stored in x's dtype), normalise xsum
                float dl[8];
                load8(delta, row * D + col, ddt, dl);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[c][i] = round_to(v[c][i] + dl[i], xdt);
                store8(xsum, row * D + col, xdt, v[c]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += RMS ? v[c][i] * v[c][i] : v[c][i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
        }
    }
    s = wave_sum(s);
    float mu = 0.f, var;
    if (RMS) {
        var = s / (float)D;
    } else {
        mu = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = c * 512 + lane * 8;
            if (col < D) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = v[c][i] - mu;
                    q += d * d;
                }
            }
        }
        var = wave_sum(q) / (float)D;
    }
    const float rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
    const int64_t orow = map_row(row, ymap);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < D) {
            float g[8], b[8], o[8];
            if (gamma) load8(gamma, col, wdt, g);
            if (beta) load8(beta, col, wdt, b);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = (v[c][i] - mu) * rs;
                if (RMS) t = round_to(t, xdt);  // HF LlamaRMSNorm: weight * normalised.to(input_dtype)
                if (gamma) t *= g[i];
                if (beta) t += b[i];
                o[i] = t;
            }
            store8(y, orow * D + col, ydt, o);
            if (y2) store8(y2, row * D + col, ydt, o);
        }
    }
}

template <int NCH, bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_dx_kernel(const void* __restrict__ dy, int dydt, otter_rowmap dymap,
                                                          const void* __restrict__ x, int xdt, const void* __restrict__ gamma,
                                                          int wdt, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const void* __restrict__ dres, void* __restrict__ dx, int dxdt,
                                                          bf16_t* __restrict__ dx2, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float mu = RMS ? 0.f : mean[row];
    const float rs = rstd[row];
    const int64_t drow = map_row(row, dymap);
    float xh[NCH][8], g[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < D) {
            float xv[8], dv[8], gm[8];
            load8(x, row * D + col, xdt, xv);
            load8(dy, drow * D + col, dydt, dv);
            if (gamma) load8(gamma, col, wdt, gm);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = (xv[i] - mu) * rs;
                if (RMS) t = round_to(t, xdt);
                xh[c][i] = t;
                const float gg = gamma ? dv[i] * gm[i] : dv[i];
                g[c][i] = gg;
                s1 += gg;
                s2 += gg * t;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) xh[c][i] = g[c][i] = 0.f;
        }
    }
    const float m1 = RMS ? 0.f : wave_sum(s1) / (float)D;
    const float m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = c * 512 + lane * 8;
        if (col < D) {
            float o[8], r[8];
            if (dres) load8(dres, row * D + col, dxdt, r);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = (g[c][i] - m1 - xh[c][i] * m2) * rs;
                if (dres) t += r[i];
                o[i] = t;
            }
            store8(dx, row * D + col, dxdt, o);
            if (dx2) Vec8<bf16_t>::store(dx2 + row * D + col, o);  // bf16 copy for a bf16 consumer (no separate cast pass)
        }
    }
}

// column-parallel partial sums for dgamma / dbeta: grid (ceil(D/512), RCH) blocks of four waves.  The waves of a block take every
// fourth row of the block's row chunk (4 x the loads in flight of the one-wave blocks this replaces: 33 -> ~16 us on the 4096 x 4096
// stream) and fold their sums through LDS, so a chunk still produces ONE partial row pair.
template <bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_dw_partial_kernel(const void* __restrict__ dy, int dydt, otter_rowmap dymap,
                                                                  const void* __restrict__ x, int xdt,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  float* __restrict__ part, int64_t rows, int D, int rch) {
    __shared__ float red[3][2][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    const bool active = col < D;
    const int64_t per = cdiv64(rows, rch);
    const int64_t r0 = (int64_t)blockIdx.y * per;
    const int64_t r1 = r0 + per < rows ? r0 + per : rows;
    float ag[8], ab[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ag[i] = ab[i] = 0.f;
    if (active) {
        for (int64_t r = r0 + wave; r < r1; r += 4) {
            float xv[8], dv[8];
            load8(x, r * D + col, xdt, xv);
            load8(dy, map_row(r, dymap) * D + col, dydt, dv);
            const float mu = RMS ? 0.f : mean[r];
            const float rs = rstd[r];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float t = (xv[i] - mu) * rs;
                if (RMS) t = round_to(t, xdt);
                ag[i] += dv[i] * t;
                ab[i] += dv[i];
            }
        }
        if (wave > 0) {
            Vec8<float>::store(&red[wave - 1][0][lane * 8], ag);
            Vec8<float>::store(&red[wave - 1][1][lane * 8], ab);
        }
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            float tg[8], tb[8];
            Vec8<float>::load(&red[w][0][lane * 8], tg);
            Vec8<float>::load(&red[w][1][lane * 8], tb);
#pragma unroll
            for (int i = 0; i < 8; ++i) { ag[i] += tg[i]; ab[i] += tb[i]; }
        }
        Vec8<float>::store(part + ((int64_t)blockIdx.y * 2 + 0) * D + col, ag);
        Vec8<float>::store(part + ((int64_t)blockIdx.y * 2 + 1) * D + col, ab);
    }
}

// final reduction over the RCH partial rows: 64 columns per block, four threads per column (the one-thread-per-column form ran 16
// blocks for D = 4096: 35 us for 4 MB of partials)
__global__ __launch_bounds__(256) void norm_bwd_dw_final_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int D, int rch, int accumulate) {
    __shared__ float red[2][4][64];
    const int c = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float sg = 0.f, sb = 0.f;
    if (col < D)
        for (int k = j; k < rch; k += 4) {
            sg += part[((int64_t)k * 2 + 0) * D + col];
            sb += part[((int64_t)k * 2 + 1) * D + col];
        }
    red[0][j][c] = sg;
    red[1][j][c] = sb;
    __syncthreads();
    if (j == 0 && col < D) {
        sg = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        sb = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        if (dgamma) dgamma[col] = accumulate ? dgamma[col] + sg : sg;
        if (dbeta) dbeta[col] = accumulate ? dbeta[col] + sb : sb;
    }
}

// column sums of a row-mapped matrix: partial[k][col] = sum over the k-th row chunk of src[map(r)][col]; same block shape as above
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ src, int sdt, otter_rowmap map,
                                                             float* __restrict__ part, int64_t rows, int D, int rch) {
    __shared__ float red[3][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + lane * 8;
    const bool active = col < D;
    const int64_t per = cdiv64(rows, rch);
    const int64_t r0 = (int64_t)blockIdx.y * per;
    const int64_t r1 = r0 + per < rows ? r0 + per : rows;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.f;
    if (active) {
        for (int64_t r = r0 + wave; r < r1; r += 4) {
            float v[8];
            load8(src, map_row(r, map) * D + col, sdt, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += v[i];
        }
        if (wave > 0) Vec8<float>::store(&red[wave - 1][lane * 8], a);
    }
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            float t[8];
            Vec8<float>::load(&red[w][lane * 8], t);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += t[i];
        }
        Vec8<float>::store(part + (int64_t)blockIdx.y * D + col, a);
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int D, int rch, int accumulate) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    float s = 0.f;
    if (col < D)
        for (int k = j; k < rch; k += 4) s += part[(int64_t)k * D + col];
    red[j][c] = s;
    __syncthreads();
    if (j == 0 && col < D) {
        s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        out[col] = accumulate ? out[col] + s : s;
    }
}

// ---- coalesced variants of the two row kernels for the training configuration (fp32 residual stream, bf16 branch tensors, D % 512 == 0) ----
// The generic kernels above give a lane 8 CONSECUTIVE columns, so an fp32 row is read as two float4 instructions whose lanes are 32 B
// apart: every 128-B line is requested by both.  Here a lane owns columns [c*512 + 4*lane, +4) and [c*512 + 256 + 4*lane, +4): every
// fp32 instruction covers 1 KB contiguous, every bf16 instruction 512 B contiguous.  The backward keeps x as loaded (fp32) and dy packed
// (bf16) and recomputes x-hat and dy*gamma in the second phase (gamma comes from L2): 96 live registers instead of 128 -> <= 128 VGPRs,
// four waves per SIMD, all 4096 rows of the C2 stream resident at once (the generic backward: 178 VGPRs, two waves per SIMD, two rounds).
__device__ __forceinline__ void ld4(const void* p, int64_t idx, int dt, float (&v)[4]) {
    if (dt == OTTER_BF16) {
        const uint2 r = *reinterpret_cast<const uint2*>((const bf16_t*)p + idx);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    } else {
        const float4 r = *reinterpret_cast<const float4*>((const float*)p + idx);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
}
// streaming loads of the coalesced row kernels (-DOTTER_NORM_NT=mask A/B builds: bit 0 = the fp32 stream x, bit 1 = the bf16 branch dy /
// delta, bit 2 = the incoming fp32 residual gradient): non-temporal = no allocation in L2 / the Infinity Cache for bytes read once.
// Measured (round 4, cold operands, profiles/r04_norm_nt_ab.txt): forward kernels 6 % SLOWER (24.9 -> 26.5, 41.2 -> 43.5 us), backward 3-5 %
// faster (47.6 -> 45.2-46.7), the training step unchanged (127.4-127.8 vs 127.5-128.2 ms) -- default stays 0.
#ifndef OTTER_NORM_NT
#define OTTER_NORM_NT 0
#endif
template <bool NT>
__device__ __forceinline__ void ld4f_s(const float* p, int64_t idx, float (&v)[4]) {
    typedef float f4v_ __attribute__((ext_vector_type(4)));
    if constexpr (NT) {
        const f4v_ r = __builtin_nontemporal_load(reinterpret_cast<const f4v_*>(p + idx));
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    } else {
        const float4 r = *reinterpret_cast<const float4*>(p + idx);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
}
template <bool NT>
__device__ __forceinline__ uint2 ld4bf_raw_s(const bf16_t* p, int64_t idx) {
    typedef unsigned u2v_ __attribute__((ext_vector_type(2)));
    if constexpr (NT) {
        const u2v_ r = __builtin_nontemporal_load(reinterpret_cast<const u2v_*>(p + idx));
        return make_uint2(r.x, r.y);
    } else return *reinterpret_cast<const uint2*>(p + idx);
}
__device__ __forceinline__ void st4bf(bf16_t* p, int64_t idx, const float (&v)[4]) {
    uint2 r;
    r.x = pack2bf(v[0], v[1]);
    r.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(p + idx) = r;
}
__device__ __forceinline__ void st4f(float* p, int64_t idx, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p + idx) = make_float4(v[0], v[1], v[2], v[3]);
}

template <int NCH, bool RMS>
__global__ __launch_bounds__(256) void norm_fwd_c_kernel(const float* __restrict__ x, const void* __restrict__ gamma, const void* __restrict__ beta,
                                                         int wdt, bf16_t* __restrict__ y, otter_rowmap ymap, bf16_t* __restrict__ y2,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int D, float eps,
                                                         const void* __restrict__ delta, int ddt, float* __restrict__ xsum) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nrun = D >> 9;
    const int64_t base = row * D + lane * 4;
    float v[NCH][2][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c < nrun) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t o = base + c * 512 + h * 256;
                ld4f_s<(OTTER_NORM_NT & 1) != 0>(x, o, v[c][h]);
                if (delta) {
                    float dl[4];
                    if ((OTTER_NORM_NT & 2) && ddt == OTTER_BF16) {
                        const uint2 r = ld4bf_raw_s<true>((const bf16_t*)delta, o);
                        dl[0] = __uint_as_float(r.x << 16); dl[1] = __uint_as_float(r.x & 0xffff0000u);
                        dl[2] = __uint_as_float(r.y << 16); dl[3] = __uint_as_float(r.y & 0xffff0000u);
                    } else ld4(delta, o, ddt, dl);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[c][h][i] += dl[i];
                    st4f(xsum, o, v[c][h]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s += RMS ? v[c][h][i] * v[c][h][i] : v[c][h][i];
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[c][h][i] = 0.f;
        }
    }
    s = wave_sum(s);
    float mu = 0.f, var;
    if (RMS) {
        var = s / (float)D;
    } else {
        mu = s / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (c < nrun) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float d = v[c][h][i] - mu;
                        q += d * d;
                    }
            }
        var = wave_sum(q) / (float)D;
    }
    const float rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
    const int64_t obase = map_row(row, ymap) * D + lane * 4;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c < nrun) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = c * 512 + h * 256 + lane * 4;
                float g[4], b[4], o[4];
                if (gamma) ld4(gamma, col, wdt, g);
                if (beta) ld4(beta, col, wdt, b);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float t = (v[c][h][i] - mu) * rs;
                    if (gamma) t *= g[i];
                    if (beta) t += b[i];
                    o[i] = t;
                }
                st4bf(y, obase + c * 512 + h * 256, o);
                if (y2) st4bf(y2, base + c * 512 + h * 256, o);
            }
        }
    }
}

template <int NCH, bool RMS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
void norm_bwd_dx_c_kernel(const bf16_t* __restrict__ dy, otter_rowmap dymap, const float* __restrict__ x, const void* __restrict__ gamma, int wdt,
                          const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dres, float* __restrict__ dx,
                          bf16_t* __restrict__ dx2, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nrun = D >> 9;
    const float mu = RMS ? 0.f : mean[row];
    const float rs = rstd[row];
    const int64_t base = row * D + lane * 4;
    const int64_t dbase = map_row(row, dymap) * D + lane * 4;
    float xr[NCH][2][4];
    uint2 dr[NCH][2];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c < nrun) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                ld4f_s<(OTTER_NORM_NT & 1) != 0>(x, base + c * 512 + h * 256, xr[c][h]);
                dr[c][h] = ld4bf_raw_s<(OTTER_NORM_NT & 2) != 0>(dy, dbase + c * 512 + h * 256);
            }
        }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c < nrun) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float gm[4];
                if (gamma) ld4(gamma, c * 512 + h * 256 + lane * 4, wdt, gm);
                const float dv[4] = {__uint_as_float(dr[c][h].x << 16), __uint_as_float(dr[c][h].x & 0xffff0000u),
                                     __uint_as_float(dr[c][h].y << 16), __uint_as_float(dr[c][h].y & 0xffff0000u)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = (xr[c][h][i] - mu) * rs;
                    const float gg = gamma ? dv[i] * gm[i] : dv[i];
                    s1 += gg;
                    s2 += gg * t;
                }
            }
        }
    const float m1 = RMS ? 0.f : wave_sum(s1) / (float)D;
    const float m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c < nrun) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t o = base + c * 512 + h * 256;
                float gm[4], r[4], out[4];
                if (gamma) ld4(gamma, c * 512 + h * 256 + lane * 4, wdt, gm);
                if (dres) ld4f_s<(OTTER_NORM_NT & 4) != 0>(dres, o, r);
                const float dv[4] = {__uint_as_float(dr[c][h].x << 16), __uint_as_float(dr[c][h].x & 0xffff0000u),
                                     __uint_as_float(dr[c][h].y << 16), __uint_as_float(dr[c][h].y & 0xffff0000u)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = (xr[c][h][i] - mu) * rs;
                    const float gg = gamma ? dv[i] * gm[i] : dv[i];
                    float u = (gg - m1 - t * m2) * rs;
                    if (dres) u += r[i];
                    out[i] = u;
                }
                st4f(dx, o, out);
                if (dx2) st4bf(dx2, o, out);
            }
        }
}

// (Round 4 tried dx AND the dgamma / dbeta partials in one pass -- eight rows per workgroup, contributions folded through a double-buffered
//  LDS stage, one barrier per 512-column chunk: correct (same tests), but 82.6 us against 67.6 us for the two passes below at the C2 stream
//  shape (tools/norm_bench.py, `ln_bwd_dres_dw`): the chunk barriers line the eight waves' loads up, and the dx pass alone already streams
//  at 5.9 TB/s.  The two-pass form stays; the experiment is commit 635b55c "LayerNorm backward: dx + dgamma/dbeta partials in one pass".)
// OTTER_NORM_VARIANT (read once): 0 = generic kernels only, 1 (default) = coalesced kernels where they apply
int norm_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OTTER_NORM_VARIANT"); v = e ? atoi(e) : 1; }
    return v;
}

int pick_nch(int64_t D) {
    int n = 1;
    while ((int64_t)n * 512 < D) n <<= 1;
    return n;
}
int pick_rch(int64_t rows) {
    int64_t r = rows / 32;
    if (r < 1) r = 1;
    if (r > 256) r = 256;
    return (int)r;
}

template <bool RMS>
int launch_fwd(const void* x, int xdt, const void* gamma, const void* beta, int wdt, void* y, int ydt, otter_rowmap ymap,
               void* y2, float* mean, float* rstd, int64_t rows, int64_t D, float eps, hipStream_t st,
               const void* delta = nullptr, int ddt = 0, void* xsum = nullptr) {
    OTTER_REQUIRE(x && y && rows > 0 && D > 0, "norm_fwd: null pointer or empty shape");
    OTTER_REQUIRE(D % 8 == 0 && D <= 8192, "norm_fwd: D=%ld must be a multiple of 8 and <= 8192", (long)D);
    const int nch = pick_nch(D);
    dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
    if (norm_variant() == 1 && xdt == OTTER_F32 && ydt == OTTER_BF16 && D % 512 == 0 && nch <= 8) {
#define LC(N) hipLaunchKernelGGL((norm_fwd_c_kernel<N, RMS>), grid, block, 0, st, (const float*)x, gamma, beta, wdt, (bf16_t*)y, ymap, (bf16_t*)y2, mean, rstd, rows, (int)D, eps, delta, ddt, (float*)xsum)
        switch (nch) {
            case 1: LC(1); break;
            case 2: LC(2); break;
            case 4: LC(4); break;
            default: LC(8); break;
        }
#undef LC
        OTTER_CHECK_LAUNCH("norm_fwd_c");
        return OTTER_OK;
    }
#define L(N) hipLaunchKernelGGL((norm_fwd_kernel<N, RMS>), grid, block, 0, st, x, xdt, gamma, beta, wdt, y, ydt, ymap, y2, mean, rstd, rows, (int)D, eps, delta, ddt, xsum)
    switch (nch) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 4: L(4); break;
        case 8: L(8); break;
        default: L(16); break;
    }
#undef L
    OTTER_CHECK_LAUNCH("norm_fwd");
    return OTTER_OK;
}

template <bool RMS>
int launch_bwd(const void* dy, int dydt, otter_rowmap dymap, const void* x, int xdt, const void* gamma, int wdt,
               const float* mean, const float* rstd, const void* dres, void* dx, int dxdt, bf16_t* dx2, float* dgamma, float* dbeta,
               int accumulate, void* ws, int64_t rows, int64_t D, hipStream_t st) {
    OTTER_REQUIRE(dy && x && rstd && rows > 0, "norm_bwd: null pointer or empty shape");
    OTTER_REQUIRE(D % 8 == 0 && D <= 8192, "norm_bwd: D=%ld must be a multiple of 8 and <= 8192", (long)D);
    const int nch = pick_nch(D);
    const bool coalesced = dx && norm_variant() == 1 && xdt == OTTER_F32 && dydt == OTTER_BF16 && dxdt == OTTER_F32 && D % 512 == 0 && nch <= 8;
    if (coalesced) {
        dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
#define LC(N) hipLaunchKernelGGL((norm_bwd_dx_c_kernel<N, RMS>), grid, block, 0, st, (const bf16_t*)dy, dymap, (const float*)x, gamma, wdt, mean, rstd, (const float*)dres, (float*)dx, dx2, rows, (int)D)
        switch (nch) {
            case 1: LC(1); break;
            case 2: LC(2); break;
            case 4: LC(4); break;
            default: LC(8); break;
        }
#undef LC
        OTTER_CHECK_LAUNCH("norm_bwd_dx_c");
    } else if (dx) {
        dim3 grid((unsigned)cdiv64(rows, 4)), block(256);
#define L(N) hipLaunchKernelGGL((norm_bwd_dx_kernel<N, RMS>), grid, block, 0, st, dy, dydt, dymap, x, xdt, gamma, wdt, mean, rstd, dres, dx, dxdt, dx2, rows, (int)D)
        switch (nch) {
            case 1: L(1); break;
            case 2: L(2); break;
            case 4: L(4); break;
            case 8: L(8); break;
            default: L(16); break;
        }
#undef L
        OTTER_CHECK_LAUNCH("norm_bwd_dx");
    }
    if (dgamma || dbeta) {
        if (!ws) OTTER_FAIL(OTTER_ERR_WORKSPACE, "norm_bwd: workspace required for dgamma/dbeta");
        const int rch = pick_rch(rows);
        dim3 grid((unsigned)cdiv64(D, 512), (unsigned)rch), block(256);
        hipLaunchKernelGGL((norm_bwd_dw_partial_kernel<RMS>), grid, block, 0, st, dy, dydt, dymap, x, xdt, mean, rstd, (float*)ws,
                           rows, (int)D, rch);
        OTTER_CHECK_LAUNCH("norm_bwd_dw_partial");
        hipLaunchKernelGGL(norm_bwd_dw_final_kernel, dim3((unsigned)cdiv64(D, 64)), dim3(256), 0, st, (const float*)ws, dgamma,
                           dbeta, (int)D, rch, accumulate);
        OTTER_CHECK_LAUNCH("norm_bwd_dw_final");
    }
    return OTTER_OK;
}

}  // namespace

extern "C" {

int otter_layernorm_fwd(const void* x, int x_dtype, const void* gamma, const void* beta, int w_dtype, void* y, int y_dtype,
                        otter_rowmap y_map, void* y2, float* mean, float* rstd, int64_t rows, int64_t D, float eps,
                        void* stream) {
    return launch_fwd<false>(x, x_dtype, gamma, beta, w_dtype, y, y_dtype, y_map, y2, mean, rstd, rows, D, eps,
                             (hipStream_t)stream);
}

int otter_add_layernorm_fwd(const void* x, int x_dtype, const void* delta, int delta_dtype, void* xsum, const void* gamma,
                            const void* beta, int w_dtype, void* y, int y_dtype, float* mean, float* rstd, int64_t rows, int64_t D,
                            float eps, void* stream) {
    OTTER_REQUIRE(delta && xsum, "add_layernorm_fwd: delta and xsum are required");
    otter_rowmap id = {0, 0, 0};
    return launch_fwd<false>(x, x_dtype, gamma, beta, w_dtype, y, y_dtype, id, nullptr, mean, rstd, rows, D, eps,
                             (hipStream_t)stream, delta, delta_dtype, xsum);
}

int64_t otter_layernorm_bwd_workspace_bytes(int64_t rows, int64_t D) { return (int64_t)pick_rch(rows) * 2 * D * 4; }

int otter_layernorm_bwd(const void* dy, int dy_dtype, otter_rowmap dy_map, const void* x, int x_dtype, const void* gamma,
                        int w_dtype, const float* mean, const float* rstd, const void* dres, void* dx, int dx_dtype,
                        void* dx_bf16, float* dgamma, float* dbeta, int accumulate, void* ws, int64_t rows, int64_t D, void* stream) {
    OTTER_REQUIRE(mean, "layernorm_bwd: mean is required");
    OTTER_REQUIRE(!dx_bf16 || dx, "layernorm_bwd: dx_bf16 without dx");
    return launch_bwd<false>(dy, dy_dtype, dy_map, x, x_dtype, gamma, w_dtype, mean, rstd, dres, dx, dx_dtype, (bf16_t*)dx_bf16, dgamma,
                             dbeta, accumulate, ws, rows, D, (hipStream_t)stream);
}

int otter_colsum(const void* src, int src_dtype, otter_rowmap src_map, float* out, int accumulate, void* ws, int64_t rows,
                 int64_t D, void* stream) {
    OTTER_REQUIRE(src && out && rows > 0 && D % 8 == 0, "colsum: bad args");
    if (!ws) OTTER_FAIL(OTTER_ERR_WORKSPACE, "colsum: workspace required (otter_layernorm_bwd_workspace_bytes)");
    const int rch = pick_rch(rows);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)cdiv64(D, 512), (unsigned)rch), dim3(256), 0, st, src, src_dtype, src_map,
                       (float*)ws, rows, (int)D, rch);
    OTTER_CHECK_LAUNCH("colsum_partial");
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv64(D, 64)), dim3(256), 0, st, (const float*)ws, out, (int)D, rch,
                       accumulate);
    OTTER_CHECK_LAUNCH("colsum_final");
    return OTTER_OK;
}

int otter_rmsnorm_fwd(const void* x, int x_dtype, const void* w, int w_dtype, void* y, float* rstd, int64_t rows, int64_t D,
                      float eps, void* stream) {
    otter_rowmap id = {0, 0, 0};
    return launch_fwd<true>(x, x_dtype, w, nullptr, w_dtype, y, x_dtype, id, nullptr, nullptr, rstd, rows, D, eps,
                            (hipStream_t)stream);
}

int otter_add_rmsnorm_fwd(const void* x, int x_dtype, const void* delta, int delta_dtype, void* xsum, const void* w, int w_dtype, void* y,
                          int y_dtype, float* rstd, int64_t rows, int64_t D, float eps, void* stream) {
    OTTER_REQUIRE((delta == nullptr) == (xsum == nullptr), "add_rmsnorm_fwd: delta and xsum go together");
    otter_rowmap id = {0, 0, 0};
    return launch_fwd<true>(x, x_dtype, w, nullptr, w_dtype, y, y_dtype, id, nullptr, nullptr, rstd, rows, D, eps, (hipStream_t)stream,
                            delta, delta_dtype, xsum);
}

int otter_rmsnorm_bwd_ex(const void* dy, int dy_dtype, const void* x, int x_dtype, const void* w, int w_dtype, const float* rstd,
                         const void* dres, void* dx, int dx_dtype, void* dx_bf16, float* dw, int accumulate, void* ws, int64_t rows,
                         int64_t D, void* stream) {
    OTTER_REQUIRE(!dx_bf16 || dx, "rmsnorm_bwd_ex: dx_bf16 without dx");
    otter_rowmap id = {0, 0, 0};
    return launch_bwd<true>(dy, dy_dtype, id, x, x_dtype, w, w_dtype, nullptr, rstd, dres, dx, dx_dtype, (bf16_t*)dx_bf16, dw, nullptr,
                            accumulate, ws, rows, D, (hipStream_t)stream);
}

int otter_rmsnorm_bwd(const void* dy, const void* x, int x_dtype, const void* w, int w_dtype, const float* rstd, void* dx,
                      float* dw, int accumulate, void* ws, int64_t rows, int64_t D, void* stream) {
    otter_rowmap id = {0, 0, 0};
    return launch_bwd<true>(dy, x_dtype, id, x, x_dtype, w, w_dtype, nullptr, rstd, nullptr, dx, x_dtype, nullptr, dw, nullptr,
                            accumulate, ws, rows, D, (hipStream_t)stream);
}

}  // extern "C"
