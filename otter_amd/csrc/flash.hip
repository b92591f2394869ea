// flash.hip -- MFMA flash attention (forward, dQ, dK+dV) for the frozen decoder host, head_dim 128, bf16 in / fp32 math.
//
// Replaces, for the training / prefill case, the reference's `scaled_multihead_dot_product_attention`
// (mpt/attention.py:22-84: q k^T * softmax_scale + attn_bias, causal mask, softmax, . v) together with the ALiBi bias of
// mpt/attention.py:447-464 and the key-padding mask of modeling_mpt.py:135-144 -- without materialising the [B,H,S,S]
// scores, and with the bias evaluated in fp32 inside the kernel:  bias[h, j] = slope[h] * (j - (Sk - 1)).
//
// Orientation (gfx950, wave64, v_mfma_f32_32x32x16_bf16; operand A: lane l = row l&31, k-elements 8(l>>5)..+7;
// operand B: lane l = column l&31, same k-elements; C: lane l = column l&31, rows (r&3) + 8(r>>2) + 4(l>>5)):
//   forward / dQ : S^T = K Q^T, so a lane owns ONE query (its column) and 16 keys per 32-key block: the softmax row
//                  statistics (max, sum, LSE, delta) are lane-local scalars, one cross-half exchange per tile.  The C
//                  registers of P^T / dS^T are, as they stand, the B operand of the second product (contraction over
//                  keys): O^T = V^T P^T, dQ^T = K^T dS^T.  Its A operand (V^T / K^T: row = d, k = keys) is read from the
//                  row-major [key][d] LDS tile with the hardware transpose read ds_read_b64_tr_b16.
//   dK / dV      : S = Q K^T, so a lane owns ONE key and 16 queries: P / dS registers are the B operand of the products
//                  that contract over queries, dV^T = dO^T P, dK^T = Q^T dS, whose A operands (row = d, k = queries) are
//                  transpose-read from the [query][d] tiles; K and V fragments of the wave's 32 keys stay in registers.
// The k-dimension of the second products runs over the C-register order (element j of half h of chunk c is row
// 16c + 4h + (j&3) + 8(j>>2)); the transpose reads fetch their rows in the same order, so no data is ever permuted.
// LDS tiles read with ds_read_b128 use 272-B rows, tiles read with the transpose read use 320-B rows (both conflict-free
// for their access pattern: 16 rows x one 16-B slot, resp. 4 rows x 64 B per 32 lanes); a tile needed both ways is
// stored twice.
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

constexpr int HD = 128;       // head dim
constexpr int LDK = HD + 8;   // row-fragment tiles (272-B rows)
constexpr int KMASK_TILES = 128;  // v2 forward / dQ: key-padding masks of up to 128 key tiles (Sk <= 8192) live in LDS
constexpr int LDT = HD + 32;  // transpose-read tiles (320-B rows)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

#ifndef OTTER_FLASH_DMA_SPREAD
#define OTTER_FLASH_DMA_SPREAD 1   // forward (128-wide): DMA pieces of the next tile between the S MFMAs, not in a burst: 41.1-41.3 -> 39.6-40.2 us at C2 (round 6; 0 = A/B build)
#endif
#ifndef OTTER_FLASH_PRIO
#define OTTER_FLASH_PRIO 0   // A/B builds: s_setprio 1 around the MFMA clusters (bit 0 forward, bit 1 dQ, bit 2 dK/dV): the two workgroups of a CU are not
                             // in step with each other, so a wave in its matrix segment may take the issue slots of its neighbour's softmax
#endif
#define FPRIO(BIT_, P_) do { if constexpr ((OTTER_FLASH_PRIO & (BIT_)) != 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(P_); __builtin_amdgcn_sched_barrier(0); } } while (0)
#ifndef OTTER_FLASH_ROWSTORE
#define OTTER_FLASH_ROWSTORE 1   // O, dQ and the per-block dK / dV leave through an LDS transpose as whole 256-byte rows (store_rows_lds), like dK / dV of the
                                 // persistent kernel; 0 (A/B builds) = the 8-byte-per-row stores of store_dt.  Round 4, C2: forward 43.9 -> 41.7 us
#endif
struct Str { int64_t b, s, h; };  // element strides of a [B, S, H, 128] view

struct FlashArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; Str qs, ks, vs;
    bf16_t* o; Str os;
    float* lse;            // [B, H, Sq] natural-log LSE
    const float* slopes;   // [H] or null
    const uint8_t* kvalid; // [B, Sk] or null
    int B, H, Sq, Sk, causal;
    float scale;
    const bf16_t* dout; Str dos;
    float* delta;          // [B, H, Sq]
    bf16_t* dq; bf16_t* dk; bf16_t* dv; Str dqs, dks, dvs;
    int lpt_group;         // heads (PAIR: head pairs) per group of the longest-first block order (divides the number of (b, head) blocks)
    int pair;              // head_dim 64: two heads per workgroup (the PAIR instantiations)
    int fuse_delta;        // round 4: the dQ kernel computes delta = rowsum(dO . O) and the log2 LSE itself and publishes them for dK/dV (no flash_delta launch)
};

// In-kernel timeline of workgroup (3,0,0) / wave 0 of the forward (diagnostics build only: -DOTTER_FLASH_TIMING)
#ifdef OTTER_FLASH_TIMING
__device__ unsigned long long* g_flash_stamps = nullptr;
#define STAMP(slot)                                                                                       \
    do {                                                                                                  \
        unsigned long long t_;                                                                            \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        if (g_flash_stamps && blockIdx.x == 3 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (slot) < 64) \
            g_flash_stamps[slot] = t_;                                                                    \
    } while (0)
// dK/dV kernel: stamps go to LDS (no vmcnt traffic: the kernel's counted s_waitcnt vmcnt(N) must keep meaning its DMA pieces) and are
// copied out after the loop; block 0, wave 0
#define DSTAMP(slot)                                                                                              \
    do {                                                                                                          \
        unsigned long long t_;                                                                                    \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 96) reinterpret_cast<unsigned long long*>(smem + 50688)[slot] = t_; \
    } while (0)
// forward (v2): same, its own LDS area (behind the key-padding masks) and output slots 96..191
#define FSTAMP(slot)                                                                                              \
    do {                                                                                                          \
        unsigned long long t_;                                                                                    \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                 \
        if (LPT && blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 96) reinterpret_cast<unsigned long long*>(smem + 65536 + 1024)[slot] = t_; \
    } while (0)
// whole-launch picture: every workgroup of kernel KID (0 forward, 1 dQ, 2 dK/dV) records begin / end in the 100 MHz constant clock
// (s_memrealtime: the same counter on every XCD) and where it ran (HW_ID, XCC_ID)
__device__ unsigned long long* g_flash_blk = nullptr;
#define BLK_LIN() ((int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z))
#define BLK_BEGIN(KID)                                                                                            \
    do {                                                                                                          \
        unsigned long long t_;                                                                                    \
        asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                             \
        if (g_flash_blk && threadIdx.x == 0 && BLK_LIN() < 4096) {                                                \
            unsigned hw_, xc_;                                                                                    \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                     \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc_));                                    \
            unsigned long long* r_ = g_flash_blk + ((KID) * 4096 + BLK_LIN()) * 4;                                \
            r_[0] = t_; r_[2] = hw_; r_[3] = xc_;                                                                 \
        }                                                                                                         \
    } while (0)
#define BLK_END(KID)                                                                                              \
    do {                                                                                                          \
        unsigned long long t_;                                                                                    \
        asm volatile("s_waitcnt vmcnt(0)\n s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        if (g_flash_blk && threadIdx.x == 0 && BLK_LIN() < 4096) g_flash_blk[((KID) * 4096 + BLK_LIN()) * 4 + 1] = t_; \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#define DSTAMP(slot) do {} while (0)
#define FSTAMP(slot) do {} while (0)
#define BLK_BEGIN(KID) do {} while (0)
#define BLK_END(KID) do {} while (0)
#endif

__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// A operand [32 rows d x 16 k] out of a row-major [k][d] tile with 320-B rows; p already includes the per-lane part
__device__ __forceinline__ bf16x8_t tr_frag(const bf16_t* p) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + 8 * LDT));
    const s16x8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
}
// per-lane offset of the transpose read: lane i of a 16-lane group supplies row (i>>2), columns 4(i&3)..+3 of the
// [4][16] block; groups 0/1 take d-columns 0-15 / 16-31 of the 32-wide d block, groups 2/3 the rows of half h = 1
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

// store the C registers of a [d, row] block product as row-major [row][d] bf16: lane = row, 4 consecutive d per r-group
__device__ __forceinline__ void store_dt(bf16_t* rowp, const f32x16_t (&acc)[4], float mul, int h2) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack2bf(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul);
            w.y = pack2bf(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(rowp + 32 * db + 8 * g + 4 * h2) = w;
        }
}

// head-pair form (see "head pairs" below): d blocks 0-1 belong to the first head of the pair (scaled by mul0), blocks 2-3 to the second
// (mul1), whose 64 columns start `poff` + 64 elements after the first head's
__device__ __forceinline__ void store_dt_pair(bf16_t* rowp, const f32x16_t (&acc)[4], float mul0, float mul1, int h2, int64_t poff) {
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const float mul = db < 2 ? mul0 : mul1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack2bf(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul);
            w.y = pack2bf(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(rowp + 32 * db + 8 * g + 4 * h2 + (db >= 2 ? poff : 0)) = w;
        }
    }
}

// Output rows of one wave (32 rows x 128: dK / dV of the persistent kernel; -DOTTER_FLASH_ROWSTORE=1 A/B builds: O and dQ too) from the
// S^T-orientation accumulators to global memory through LDS: lane (ql, h2) holds
// d = 32 db + 8 g + 4 h2 + e of key ql; written as 8-B pieces into a 16-B-slot-swizzled row-major tile, read back as 16 B per lane,
// four whole rows per store instruction
template <bool PAIR>
__device__ __forceinline__ void store_rows_lds(char* trn, bf16_t* base, int64_t row_stride, int row0, int nrows_valid, const f32x16_t (&acc)[4],
                                               float mul0, float mul1, int lane, int64_t poff) {
    const int ql = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        const float mul = (PAIR && db >= 2) ? mul1 : mul0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack2bf(acc[db][4 * g + 0] * mul, acc[db][4 * g + 1] * mul);
            w.y = pack2bf(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
            *reinterpret_cast<uint2*>(trn + ql * 256 + (((4 * db + g) ^ (ql & 15)) << 4) + 8 * h2) = w;
        }
    }
    const int p = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lane >> 4);
        const uint4 w = *reinterpret_cast<const uint4*>(trn + row * 256 + ((p ^ (row & 15)) << 4));
        if (row0 + row < nrows_valid)
            *reinterpret_cast<uint4*>(base + (int64_t)(row0 + row) * row_stride + 8 * p + ((PAIR && p >= 8) ? poff : 0)) = w;
    }
}

// ------------------------------------------------------------------------------------------------------------
// forward: grid (ceil(Sq/128), H, B), 4 waves, wave w owns queries q0 + 32w .. +31; key tiles of 64
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flash_fwd_kernel(FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);  // [64][LDK]
    bf16_t* Vt = Ks + 64 * LDK;                    // [64][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * 128;
    const int qi = q0 + wave * 32 + ql;
    const int off = a.Sk - a.Sq;
    const bf16_t* qp = a.q + b * a.qs.b + hd * a.qs.h + (int64_t)(qi < a.Sq ? qi : a.Sq - 1) * a.qs.s;
    bf16x8_t qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * c + 8 * h2);
    const bf16_t* kb = a.k + b * a.ks.b + hd * a.ks.h;
    const bf16_t* vb = a.v + b * a.vs.b + hd * a.vs.h;
    const uint8_t* kv = a.kvalid ? a.kvalid + (int64_t)b * a.Sk : nullptr;
    const float sc2 = a.scale * LOG2E, sl2 = a.slopes ? a.slopes[hd] * LOG2E : 0.f;
    int nkt = (a.Sk + 63) >> 6;
    if (a.causal) {
        const int qmax = (q0 + 127 < a.Sq ? q0 + 127 : a.Sq - 1) + off;
        const int lim = qmax < 0 ? 0 : (qmax >> 6) + 1;
        nkt = nkt < lim ? nkt : lim;
    }
    uint4 kr[4], vr[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + 256 * i, row = ch >> 4, c16 = ch & 15, key = kt * 64 + row;
            if (key < a.Sk) {
                kr[i] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * a.ks.s + c16 * 8);
                vr[i] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * a.vs.s + c16 * 8);
            } else {
                kr[i] = make_uint4(0, 0, 0, 0);
                vr[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + 256 * i, row = ch >> 4, c16 = ch & 15;
            *reinterpret_cast<uint4*>(Ks + row * LDK + c16 * 8) = kr[i];
            *reinterpret_cast<uint4*>(Vt + row * LDT + c16 * 8) = vr[i];
        }
    };
    const int troff = tr_lane_off(lane);
    float m = -INFINITY, lsum = 0.f;
    f32x16_t o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = zero16();
    STAMP(0);
    if (nkt > 0) gload(0);
    STAMP(1);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        STAMP(2 + 6 * kt);
        swrite();
        __syncthreads();
        STAMP(3 + 6 * kt);
        if (kt + 1 < nkt) gload(kt + 1);
        const int k0 = kt * 64;
        if (a.causal && k0 > q0 + wave * 32 + 31 + off) continue;  // whole tile above this wave's diagonal
        f32x16_t s[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            s[kbk] = zero16();
            const bf16_t* rowp = Ks + (32 * kbk + ql) * LDK + 8 * h2;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(rowp + 16 * c), qf[c], s[kbk], 0, 0, 0);
        }
        // element (kbk, r) is key k0 + cidx + 4 h2 with the compile-time cidx = 32 kbk + (r&3) + 8 (r>>2): causal / length
        // limits, the ALiBi ramp and the key-validity bit are all "constant + per-tile lane scalar"
        const int kend = a.Sk - 1 - k0, rel = qi + off - k0;
        const int limh = (a.causal && rel < kend ? rel : kend) - 4 * h2;
        const float base = sl2 * (float)(k0 + 4 * h2 - (a.Sk - 1));
        unsigned long long vmh = ~0ull;
        if (kv) {
            const int jj = k0 + lane;
            vmh = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0) >> (4 * h2);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                float x = fmaf(s[kbk][r], sc2, fmaf(sl2, (float)cidx, base));
                const bool ok = cidx <= limh && ((vmh >> cidx) & 1ull);
                x = ok ? x : -INFINITY;
                s[kbk][r] = x;
                mx = fmaxf(mx, x);
            }
        STAMP(4 + 6 * kt);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        STAMP(5 + 6 * kt);
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
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        STAMP(6 + 6 * kt);
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8_t pf = pack8(s[kbk], 8 * c);
                const bf16_t* tp = Vt + troff + (32 * kbk + 16 * c) * LDT;
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tp + 32 * db), pf, o[db], 0, 0, 0);
            }
        STAMP(7 + 6 * kt);
    }
    STAMP(60);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    if (qi < a.Sq) {
        store_dt(a.o + b * a.os.b + hd * a.os.h + (int64_t)qi * a.os.s, o, inv, h2);
        if (h2 == 0) a.lse[((int64_t)b * a.H + hd) * a.Sq + qi] = lsum > 0.f ? m * LN2 + logf(lsum) : -INFINITY;
    }
    STAMP(61);
}

// ------------------------------------------------------------------------------------------------------------
// forward, version 2: same math and orientation; K/V tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds: no staging
// registers, no ds_write pass, rows past Sk are zero-filled by the buffer range check), double-buffered with ONE raw
// barrier per tile, unpadded 256-B rows with XOR swizzles applied on the DMA source offsets:
//   K (row fragments, ds_read_b128): 16-B slot ^= row & 15        -> the 16 rows of a lane group hit 16 distinct slots
//   V (ds_read_b64_tr_b16):          64-B block ^= row & 3        -> the 4 rows of a transpose read hit 4 distinct blocks
// 64 KB of LDS and <= 256 registers per lane: two workgroups per CU, so one wave's softmax VALU / LDS latency runs under
// the co-resident wave's MFMAs.  Less VALU: tiles entirely below the diagonal skip the mask arithmetic, and the
// accumulator is only rescaled when some row's running maximum grew by more than 2^8 (stale maxima are exact: P <= 256).
// ------------------------------------------------------------------------------------------------------------
// LDS-DMA issued through inline asm.  Why not the builtin: hipcc (ROCm 7.2) treats a builtin LDS-DMA as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read it cannot prove disjoint -- in these kernels every fragment read of the CURRENT tile,
// issued right after the DMA of the NEXT one.  That drained the prefetch every iteration (dK/dV kernel: 3 900 cycles per 32-row tile
// against 1 024 of MFMA work, tools/flash_timeline_dkv.py).  An asm statement is invisible to that bookkeeping; completion is counted by
// hand (the explicit `s_waitcnt vmcnt(N)` + raw `s_barrier` at the top of each iteration, which the kernels already had).  M0 (the LDS
// destination) is saved and restored inside the statement; the descriptor and soffset are SGPR operands (s_nop 4 covers a
// v_readfirstlane -> buffer hazard).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
// Longest-first block order in groups of G heads (a.lpt_group): 1-D block index -> (group, rank inside the head, head).  Inside a group the
// blocks are walked rank by rank (rank 0 = the longest block of every head), so the launch never ends on a long block; the group keeps
// the K / V (or Q / dO) panels that the blocks of one head share to what the L2 + Infinity Cache hold -- with ONE group over 512 heads
// of 1 396 tokens (config C5) every concurrently running block belonged to a different head and the forward re-read K / V from HBM for
// each of its 11 query tiles (392 -> 467 us per layer).
struct LptIdx { int bh, rank; };
__device__ __forceinline__ LptIdx lpt_decode(int idx, int nrank, int G) {
    const int per = nrank * G;
    const int grp = idx / per, within = idx - grp * per;
    LptIdx r;
    r.rank = within / G;
    r.bh = grp * G + (within - r.rank * G);
    return r;
}
__device__ __forceinline__ u32x4_t make_rsrc4(const void* p, uint32_t bytes) {
    const uint64_t pa = (uint64_t)p;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)pa);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(pa >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
// Round 4: the LEAN form of csrc/gemm.hip (s_mov m0; s_nop 0; buffer_load ... lds).  M0 is clobbered, not saved / restored: hipcc never keeps
// a value in M0 across statements (every compiler-generated use is preceded by its own s_mov), and nothing else in these kernels uses it.
// The descriptor is built once in the prologue (the v_readfirstlane -> buffer-descriptor hazard is long past) and soffset comes from scalar
// arithmetic.  Per piece 3 instructions instead of 6 + a 5-cycle s_nop: with one or two waves per SIMD every instruction beside an MFMA
// is an issue slot (-DOTTER_FLASH_SAFE_DMA restores the save / settle / restore form for an A/B build).
#ifdef OTTER_FLASH_SAFE_DMA
__device__ __forceinline__ void dma16_asm(u32x4_t r, const char* lds, uint32_t voff, uint32_t soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr(lds)), "v"(voff), "s"(r), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma4_asm(u32x4_t r, const char* lds, uint32_t voff, uint32_t soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr(lds)), "v"(voff), "s"(r), "s"(soff) : "memory");
}
#else
// (diagnostics build: the stamp stores under `threadIdx.x == 0` make hipcc treat the ring-slot address of the persistent dK/dV kernel as
//  divergent and hand a VGPR to the "s" operand; an explicit readfirstlane there only)
#ifdef OTTER_FLASH_TIMING
#define DMA_LDS_ADDR(P_) ((unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr(P_)))
#define DMA_SOFF(S_) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(S_)))
#else
#define DMA_LDS_ADDR(P_) lds_addr(P_)
#define DMA_SOFF(S_) (S_)
#endif
__device__ __forceinline__ void dma16_asm(u32x4_t r, const char* lds, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(DMA_LDS_ADDR(lds)), "v"(voff), "s"(r), "s"(DMA_SOFF(soff)) : "memory");
}
__device__ __forceinline__ void dma4_asm(u32x4_t r, const char* lds, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(DMA_LDS_ADDR(lds)), "v"(voff), "s"(r), "s"(DMA_SOFF(soff)) : "memory");
}
#endif

__device__ __forceinline__ void flash_dma_tile(u32x4_t rk, u32x4_t rv, char* kdst, char* vdst,
                                               const uint32_t (&vk)[4], const uint32_t (&vv)[4], int wave, uint32_t ksoff,
                                               uint32_t vsoff) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dma16_asm(rk, kdst + (wave * 4 + i) * 1024, vk[i], ksoff);
        dma16_asm(rv, vdst + (wave * 4 + i) * 1024, vv[i], vsoff);
    }
}

// Head pairs (PAIR, head_dim 64: Persimmon / Fuyu-8B, fuyu/modeling_persimmon.py:310): one workgroup takes TWO heads whose 64-wide rows
// lie head_stride elements apart in the same token.  The 256-B LDS row of a tile holds head A's 64 columns in slots 0-7 and head B's in
// slots 8-15 (only the DMA source offset of a slot changes), so every tile layout, swizzle and fragment address of the 128-wide kernel
// stays; the contraction over d splits into c = 0-3 (head A) and c = 4-7 (head B), the softmax runs once per head, and d blocks 0-1 /
// 2-3 of the second product take P of head A / B.  Against zero-padding the heads to 128 (round 2): half the MFMA work, half the tile
// traffic and workgroups, no padded q / k / v / o / dO copies.  a.H counts real heads; a pair block is (b, hd) with hd < H/2.
// The pair kernels have ONE tile code path: every tile applies its visibility mask (one 64-bit word per lane, v_bfe_i32 + v_bfi_b32 per
// score).  Separate interior / masked paths as in the 128-wide kernels did not fit the 256 registers of two workgroups per CU (84-392 B of
// scratch per lane, reloaded through VMEM in front of every tile) and measured slower, as did one workgroup per CU with 512 registers
// (C5 shape, forward / backward: 142 / 451 us one path, 153 / 650 two paths, 196 / 505 one workgroup per CU; zero-padded heads 205 / 694).
template <bool LPT, bool PAIR = false, bool ALIBI = true>   // ALIBI = false (pair kernels only): no slopes, the bias arithmetic is compiled out
__global__ __launch_bounds__(256, 2) void flash_fwd2_kernel(FlashArgs a) {
    BLK_BEGIN(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1, 16 KB each
    const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LPT: 1-D grid, LAST query tile first (under the causal mask it sees the most keys): the launch ends on the short blocks
    const int nqb = (a.Sq + 127) >> 7;
    const LptIdx li = lpt_decode((int)blockIdx.x, nqb, a.lpt_group);
    const int HB = PAIR ? a.H >> 1 : a.H, hm = PAIR ? 2 : 1;   // head blocks per batch row; heads per block
    const int b = LPT ? li.bh / HB : blockIdx.z, hd = LPT ? li.bh % HB : blockIdx.y;
    const int q0 = (LPT ? nqb - 1 - li.rank : (int)blockIdx.x) * 128;
    const int qi = q0 + wave * 32 + ql;
    const int off = a.Sk - a.Sq;
    const bf16_t* qp = a.q + b * a.qs.b + hd * hm * a.qs.h + (int64_t)(qi < a.Sq ? qi : a.Sq - 1) * a.qs.s;
    const int64_t qpo = PAIR ? a.qs.h - 64 : 0;   // extra element offset of columns 64.. (head B)
    bf16x8_t qf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * c + 8 * h2 + (c >= 4 ? qpo : 0));
    const bf16_t* kb = a.k + b * a.ks.b + hd * hm * a.ks.h;
    const bf16_t* vb = a.v + b * a.vs.b + hd * hm * a.vs.h;
    const uint8_t* kv = a.kvalid ? a.kvalid + (int64_t)b * a.Sk : nullptr;
    const float sc2 = a.scale * LOG2E, sl2 = a.slopes ? a.slopes[hd * hm] * LOG2E : 0.f;
    const float sl2b = PAIR && a.slopes ? a.slopes[hd * 2 + 1] * LOG2E : 0.f;
    int nkt = (a.Sk + 63) >> 6;
    if (a.causal) {
        const int qmax = (q0 + 127 < a.Sq ? q0 + 127 : a.Sq - 1) + off;
        const int lim = qmax < 0 ? 0 : (qmax >> 6) + 1;
        nkt = nkt < lim ? nkt : lim;
    }
    const uint32_t krow = (uint32_t)(a.ks.s * 2), vrow = (uint32_t)(a.vs.s * 2);
    const uint32_t kpb = PAIR ? (uint32_t)((a.ks.h - 64) * 2) : 0u, vpb = PAIR ? (uint32_t)((a.vs.h - 64) * 2) : 0u;   // source bytes skipped before slot 8
    const u32x4_t rk = make_rsrc4(kb, (uint32_t)((int)((uint32_t)(a.Sk - 1) * krow + 256u + kpb)));
    const u32x4_t rv = make_rsrc4(vb, (uint32_t)((int)((uint32_t)(a.Sk - 1) * vrow + 256u + vpb)));
    uint32_t vk[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = 4 * (wave * 4 + i) + (lane >> 4), p = lane & 15;
        const int sk = p ^ (rl & 15), sv = (((p >> 2) ^ (rl & 3)) << 2) | (p & 3);   // source slot that lands in LDS slot p
        vk[i] = (uint32_t)rl * krow + (uint32_t)(sk << 4) + (sk >= 8 ? kpb : 0u);
        vv[i] = (uint32_t)rl * vrow + (uint32_t)(sv << 4) + (sv >= 8 ? vpb : 0u);
    }
    // read side: K row fragment of row 32 kbk + ql, logical slot 2c + h2 -> byte ((32c) ^ (y << 4)) with y = h2 ^ (ql & 15);
    // V transpose read: lane i of a 16-lane group supplies row 4h + (i>>2) (+ key base, + 8), d block db at 64-B block
    // db ^ (i>>2): one per-lane base, the d block as an XOR on bits 6-7, everything else as immediates
    const int kfo = ql * 256 + ((h2 ^ (ql & 15)) << 4);
    const int gi = lane & 15, gg = lane >> 4;
    const int vto = (4 * (gg >> 1) + (gi >> 2)) * 256 + ((gi >> 2) << 6) + 32 * (gg & 1) + 8 * (gi & 3);
    float m = -INFINITY, lsum = 0.f;
    float mB = -INFINITY, lsumB = 0.f;   // PAIR: statistics of head B
    f32x16_t o[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = zero16();
    // Key-padding masks: one 64-bit word per key tile, parked behind the tile buffers.  Reading the validity bytes inside the loop is a
    // load hipcc counts: it waits vmcnt(0) for it, which also drains the asm LDS-DMA of the next tile (every padded batch paid that).
    unsigned long long* const kmask = reinterpret_cast<unsigned long long*>(smem + 65536);
    const bool kmask_lds = kv != nullptr && nkt <= KMASK_TILES;
    if (kmask_lds) {
        for (int t = wave; t < nkt; t += 4) {
            const int jj = t * 64 + lane;
            const unsigned long long mk = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0);
            if (lane == 0) kmask[t] = mk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // published by the first barrier of the loop
    }
    if (nkt > 0) flash_dma_tile(rk, rv, smem, smem + 32768, vk, vv, wave, 0u, 0u);
    // every load hipcc knows about (the register-resident fragments above) is retired HERE, through the builtin its scoreboard models:
    // otherwise it re-issues its counted waits for them -- down to vmcnt(0) -- in front of the MFMAs of EVERY iteration, and those
    // waits also drain the asm LDS-DMA of the next tile.  (vmcnt(0), expcnt / lgkmcnt untouched: simm16 0x0F70 on gfx9.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    FSTAMP(0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        FSTAMP(1 + 5 * kt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tile kt has landed for every wave; every wave is done with tile kt-1
        asm volatile("" ::: "memory");
        FSTAMP(2 + 5 * kt);
        const int k0 = kt * 64;
        const int wq0 = q0 + wave * 32 + off;  // first query of the wave, in key coordinates
        // OTTER_FLASH_DMA_SPREAD (round 6): the eight DMA pieces of tile kt + 1 are issued between the MFMAs of S = K Q^T instead of in one
        // burst behind the barrier (the burst took ~480 cycles of a 4 400-cycle tile, profiles/r04_flash_block0_timeline.txt): -3 % per launch
        const bool dma_next = kt + 1 < nkt;
        const bool skip_tile = a.causal && k0 > wq0 + 31;   // whole tile above this wave's diagonal
        char* const kdn = smem + (cur ^ 1) * 16384;
        char* const vdn = smem + 32768 + (cur ^ 1) * 16384;
        const uint32_t ksn = (uint32_t)(kt + 1) * 64u * krow, vsn = (uint32_t)(kt + 1) * 64u * vrow;
        if (dma_next && (!OTTER_FLASH_DMA_SPREAD || PAIR || skip_tile)) flash_dma_tile(rk, rv, kdn, vdn, vk, vv, wave, ksn, vsn);
        FSTAMP(3 + 5 * kt);
        if (skip_tile) continue;
        const char* Kc = smem + cur * 16384;
        const char* Vc = smem + 32768 + cur * 16384;
        if constexpr (PAIR) {
            // visibility of the lane's 32 keys of the tile as one bit mask (bit cidx): causal / sequence-end limit and key-padding bits
            unsigned okm_lo, okm_hi;
            {
                const int kend = a.Sk - 1 - k0, rel = qi + off - k0;
                const int limh = (a.causal && rel < kend ? rel : kend) - 4 * h2;
                unsigned long long vmh = ~0ull;
                if (kmask_lds) {
                    vmh = kmask[kt] >> (4 * h2);
                } else if (kv) {
                    const int jj = k0 + lane;
                    vmh = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0) >> (4 * h2);
                }
                const unsigned long long okm = limh < 0 ? 0ull : (limh >= 63 ? vmh : vmh & ((2ull << limh) - 1ull));
                okm_lo = (unsigned)okm;
                okm_hi = (unsigned)(okm >> 32);
            }
            // one head's share of the tile (E: std::integral_constant, the head's register indices must be compile-time)
            auto head = [&](auto E) {
                constexpr int e = decltype(E)::value;
                f32x16_t sp[2];
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk) {
                    sp[kbk] = zero16();
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        sp[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            *reinterpret_cast<const bf16x8_t*>(Kc + kbk * 8192 + (kfo ^ (32 * (c + 4 * e)))), qf[c + 4 * e], sp[kbk], 0, 0, 0);
                }
                const float sle = e ? sl2b : sl2;
                float& me = e ? mB : m;
                float& le = e ? lsumB : lsum;
                const float base = sle * (float)(k0 + 4 * h2 - (a.Sk - 1));
                float mx = -INFINITY;
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                        float x = ALIBI ? fmaf(sp[kbk][r], sc2, fmaf(sle, (float)cidx, base)) : sp[kbk][r] * sc2;
                        {   // v_bfe_i32 + v_bfi_b32: 0 / ~0 from the key's bit, then x or -inf
                            const unsigned mk = (unsigned)__builtin_amdgcn_sbfe((int)(cidx < 32 ? okm_lo : okm_hi), cidx & 31, 1);
                            x = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, x) & mk) | (~mk & 0xFF800000u));
                        }
                        sp[kbk][r] = x;
                        mx = fmaxf(mx, x);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (__ballot(mx > me + 8.0f) != 0ull) {
                    const float mnew = fmaxf(me, mx);
                    const float muse = mnew == -INFINITY ? 0.f : mnew;
                    const float alpha = __builtin_amdgcn_exp2f(me - muse);
                    me = mnew;
                    le *= alpha;
#pragma unroll
                    for (int db = 0; db < 2; ++db)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[2 * e + db][r] *= alpha;
                }
                const float muse = me == -INFINITY ? 0.f : me;
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = __builtin_amdgcn_exp2f(sp[kbk][r] - muse);
                        sp[kbk][r] = p;
                        le += p;
                    }
#pragma unroll
                for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const bf16x8_t pf = pack8(sp[kbk], 8 * c);
#pragma unroll
                        for (int dbl = 0; dbl < 2; ++dbl) {
                            const int db = 2 * e + dbl;
                            const char* tp = Vc + (vto ^ (db << 6)) + (32 * kbk + 16 * c) * 256;
                            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)tp);
                            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tp + 2048));
                            const s16x8_t vfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfr), pf, o[db], 0, 0, 0);
                        }
                    }
            };
            head(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);   // heads apart: interleaved, both heads' score registers are live at once (56 B of scratch)
            head(std::integral_constant<int, 1>{});
            continue;
        }
        f32x16_t s[2];
        FPRIO(1, 1);
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            s[kbk] = zero16();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    *reinterpret_cast<const bf16x8_t*>(Kc + kbk * 8192 + (kfo ^ (32 * c))), qf[c], s[kbk], 0, 0, 0);
                if constexpr (OTTER_FLASH_DMA_SPREAD != 0) {
                    if ((c & 1) && dma_next) {      // piece i of K behind MFMAs 1, 3, 5, 7 of the first key block, of V behind those of the second
                        const int i = c >> 1;
                        if (kbk == 0) dma16_asm(rk, kdn + (wave * 4 + i) * 1024, vk[i], ksn);
                        else dma16_asm(rv, vdn + (wave * 4 + i) * 1024, vv[i], vsn);
                    }
                }
            }
        }
        FPRIO(1, 0);
        const float base = sl2 * (float)(k0 + 4 * h2 - (a.Sk - 1));
        float mx = -INFINITY;
        const bool interior = kv == nullptr && k0 + 63 < a.Sk && (!a.causal || k0 + 63 <= wq0);
        if (interior) {
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                    const float x = fmaf(s[kbk][r], sc2, fmaf(sl2, (float)cidx, base));
                    s[kbk][r] = x;
                    mx = fmaxf(mx, x);
                }
        } else {
            const int kend = a.Sk - 1 - k0, rel = qi + off - k0;
            const int limh = (a.causal && rel < kend ? rel : kend) - 4 * h2;
            unsigned long long vmh = ~0ull;
            if (kmask_lds) {
                vmh = kmask[kt] >> (4 * h2);
            } else if (kv) {
                const int jj = k0 + lane;
                vmh = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0) >> (4 * h2);
            }
            // round 4: visibility of the lane's 32 keys as ONE 64-bit word (causal / sequence-end limit ANDed with the key-padding bits), then
            // v_bfe_i32 + v_bfi_b32 per score -- the head-pair kernels' form -- instead of a 64-bit shift, two compares and a select per score
            // (the diagonal tiles are 2 of the 2-8 tiles of every block at S = 512)
            const unsigned long long okm = limh < 0 ? 0ull : (limh >= 63 ? vmh : vmh & ((2ull << limh) - 1ull));
            const unsigned okm_lo = (unsigned)okm, okm_hi = (unsigned)(okm >> 32);
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                    float x = fmaf(s[kbk][r], sc2, fmaf(sl2, (float)cidx, base));
                    const unsigned mk = (unsigned)__builtin_amdgcn_sbfe((int)(cidx < 32 ? okm_lo : okm_hi), cidx & 31, 1);
                    x = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, x) & mk) | (~mk & 0xFF800000u));
                    s[kbk][r] = x;
                    mx = fmaxf(mx, x);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // lazy rescale: only when some row's maximum grew by more than 8 (log2 units) -- wave-uniform decision
        if (__ballot(mx > m + 8.0f) != 0ull) {
            const float mnew = fmaxf(m, mx);
            const float muse = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = __builtin_amdgcn_exp2f(m - muse);
            m = mnew;
            lsum *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        const float muse = m == -INFINITY ? 0.f : m;
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kbk][r] - muse);
                s[kbk][r] = p;
                lsum += p;
            }
#ifdef OTTER_FLASH_TIMING
        asm volatile("" :: "v"(s[1][15]), "v"(lsum));
#endif
        FSTAMP(4 + 5 * kt);
        FPRIO(1, 1);
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8_t pf = pack8(s[kbk], 8 * c);
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const char* tp = Vc + (vto ^ (db << 6)) + (32 * kbk + 16 * c) * 256;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)tp);
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tp + 2048));
                    const s16x8_t vfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vfr), pf, o[db], 0, 0, 0);
                }
            }
        FPRIO(1, 0);
        FSTAMP(5 + 5 * kt);
    }
    FSTAMP(90);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    if constexpr (PAIR) {
        lsumB += __shfl_xor(lsumB, 32, 64);
        const float invB = lsumB > 0.f ? 1.0f / lsumB : 0.f;
#if OTTER_FLASH_ROWSTORE
        __builtin_amdgcn_s_barrier();
        store_rows_lds<true>(smem + wave * 8192, a.o + b * a.os.b + hd * 2 * a.os.h, a.os.s, q0 + wave * 32, a.Sq, o, inv, invB, lane, a.os.h - 64);
        if (qi < a.Sq && h2 == 0) {
            float* lp = a.lse + ((int64_t)b * a.H + hd * 2) * a.Sq + qi;
            lp[0] = lsum > 0.f ? m * LN2 + logf(lsum) : -INFINITY;
            lp[a.Sq] = lsumB > 0.f ? mB * LN2 + logf(lsumB) : -INFINITY;
        }
#else
        if (qi < a.Sq) {
            store_dt_pair(a.o + b * a.os.b + hd * 2 * a.os.h + (int64_t)qi * a.os.s, o, inv, invB, h2, a.os.h - 64);
            if (h2 == 0) {
                float* lp = a.lse + ((int64_t)b * a.H + hd * 2) * a.Sq + qi;
                lp[0] = lsum > 0.f ? m * LN2 + logf(lsum) : -INFINITY;
                lp[a.Sq] = lsumB > 0.f ? mB * LN2 + logf(lsumB) : -INFINITY;
            }
        }
#endif
        return;
    }
#if OTTER_FLASH_ROWSTORE
    __builtin_amdgcn_s_barrier();   // every wave is done with the last K / V tile: its LDS is free for the row-major staging of O
    store_rows_lds<false>(smem + wave * 8192, a.o + b * a.os.b + hd * a.os.h, a.os.s, q0 + wave * 32, a.Sq, o, inv, inv, lane, 0);
    if (qi < a.Sq && h2 == 0) a.lse[((int64_t)b * a.H + hd) * a.Sq + qi] = lsum > 0.f ? m * LN2 + logf(lsum) : -INFINITY;
#else
    if (qi < a.Sq) {
        store_dt(a.o + b * a.os.b + hd * a.os.h + (int64_t)qi * a.os.s, o, inv, h2);
        if (h2 == 0) a.lse[((int64_t)b * a.H + hd) * a.Sq + qi] = lsum > 0.f ? m * LN2 + logf(lsum) : -INFINITY;
    }
#endif
    BLK_END(0);
#ifdef OTTER_FLASH_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FSTAMP(91);
    __syncthreads();
    if (LPT && blockIdx.x == 0 && threadIdx.x < 96 && g_flash_stamps) g_flash_stamps[96 + threadIdx.x] = reinterpret_cast<unsigned long long*>(smem + 65536 + 1024)[threadIdx.x];
#endif
}

// ------------------------------------------------------------------------------------------------------------
// forward, version 3 (round 4, flash variant 6): SIXTEEN queries per wave on v_mfma_f32_16x16x32_bf16, eight waves per workgroup.
// Why: the counters of version 2 (profiles/r04_pmc_flash.json) show a latency-bound loop -- 41 % of the wave cycles are waits, 17 % issue
// VALU, the matrix pipe is 15 % busy -- with two waves per SIMD (202 registers each).  A wave that owns 16 queries instead of 32 needs
// S^T 16 + O^T 32 + Q 16 registers: <= 128 in all, FOUR waves per SIMD, so twice as many dependent chains (MFMA accumulate, max / shuffle /
// exp) are in flight per SIMD.  Same 64 KB K0 | K1 | V0 | V1 LDS tiles (64 keys), same LDS-DMA double buffering, same LPT block order.
// Orientation (A: lane l = row l&15, k = 8(l>>4)..+7;  B: lane l = column l&15, same k;  C: lane l = column l&15, rows 4(l>>4)+0..3):
//   S^T[key][query] = K Q^T per 16-key block: lane (g = l>>4, j = l&15) holds keys 4g..4g+3 of each of the tile's four blocks for query j
//   (row statistics: two cross-lane exchanges, xor 16 and 32); the P^T registers of blocks 2s, 2s+1 are, as they stand, the B operand of
//   O^T[d][query] += V^T P^T over the 32 keys {16(2s)+4g+r, 16(2s+1)+4g+r}; the matching A operand V^T comes out of the row-major V tile by
//   two ds_read_b64_tr_b16 (rows 16(2s)+4g.., 16(2s+1)+4g.. of the 16-column d block).
// Swizzles (on the DMA source offsets): K as in version 2 (16-B slot ^= row & 15: the 16 lanes of a ds_read_b128 service group hit 16
// distinct slots); V: 32-B unit ^= row & 7 (the 8 rows one pass of the transpose read touches hit 8 distinct units).
// ------------------------------------------------------------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool LPT>
__global__ __launch_bounds__(512, 4) void flash_fwd3_kernel(FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1, 16 KB each
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = (a.Sq + 127) >> 7;
    const LptIdx li = lpt_decode((int)blockIdx.x, nqb, a.lpt_group);
    const int b = LPT ? li.bh / a.H : blockIdx.z, hd = LPT ? li.bh % a.H : blockIdx.y;
    const int q0 = (LPT ? nqb - 1 - li.rank : (int)blockIdx.x) * 128;
    const int qi = q0 + wave * 16 + j;
    const int off = a.Sk - a.Sq;
    const bf16_t* qp = a.q + b * a.qs.b + hd * a.qs.h + (int64_t)(qi < a.Sq ? qi : a.Sq - 1) * a.qs.s;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 32 * ks + 8 * g);
    const bf16_t* kb = a.k + b * a.ks.b + hd * a.ks.h;
    const bf16_t* vb = a.v + b * a.vs.b + hd * a.vs.h;
    const uint8_t* kv = a.kvalid ? a.kvalid + (int64_t)b * a.Sk : nullptr;
    const float sc2 = a.scale * LOG2E, sl2 = a.slopes ? a.slopes[hd] * LOG2E : 0.f;
    int nkt = (a.Sk + 63) >> 6;
    if (a.causal) {
        const int qmax = (q0 + 127 < a.Sq ? q0 + 127 : a.Sq - 1) + off;
        const int lim = qmax < 0 ? 0 : (qmax >> 6) + 1;
        nkt = nkt < lim ? nkt : lim;
    }
    const uint32_t krow = (uint32_t)(a.ks.s * 2), vrow = (uint32_t)(a.vs.s * 2);
    const u32x4_t rk = make_rsrc4(kb, (uint32_t)((int)((uint32_t)(a.Sk - 1) * krow + 256u)));
    const u32x4_t rv = make_rsrc4(vb, (uint32_t)((int)((uint32_t)(a.Sk - 1) * vrow + 256u)));
    // DMA: a tile is 16 pieces of 1 KB (4 rows x 256 B) per operand; wave w issues pieces 2w, 2w+1 of K and of V
    uint32_t vk[2], vv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = 4 * (wave * 2 + i) + (lane >> 4), p = lane & 15;
        const int sk = p ^ (rl & 15), sv = (((p >> 1) ^ (rl & 7)) << 1) | (p & 1);   // source slot that lands in LDS slot p
        vk[i] = (uint32_t)rl * krow + (uint32_t)(sk << 4);
        vv[i] = (uint32_t)rl * vrow + (uint32_t)(sv << 4);
    }
    auto dma_tile = [&](char* kdst, char* vdst, uint32_t ksoff, uint32_t vsoff) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            dma16_asm(rk, kdst + (wave * 2 + i) * 1024, vk[i], ksoff);
            dma16_asm(rv, vdst + (wave * 2 + i) * 1024, vv[i], vsoff);
        }
    };
    // read side.  K fragment (key block kb, k-step ks): row 16 kb + j, logical slot 4 ks + g -> byte (kfo ^ (ks << 6)) + 4096 kb.
    // V transpose read (32-key step s, d block db): lane i = j of group g supplies row 32 s + 4 g + (j >> 2) (+ 16), unit db ^ (row & 7)
    const int kfo = j * 256 + ((g ^ j) << 4);
    const int vto = (4 * g + (j >> 2)) * 256 + ((4 * (g & 1) + (j >> 2)) << 5) + 8 * (j & 3);
    float m = -INFINITY, lsum = 0.f;
    f32x4_t o[8];
#pragma unroll
    for (int db = 0; db < 8; ++db) o[db] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned long long* const kmask = reinterpret_cast<unsigned long long*>(smem + 65536);
    const bool kmask_lds = kv != nullptr && nkt <= KMASK_TILES;
    if (kmask_lds) {
        for (int t = wave; t < nkt; t += 8) {
            const int jj = t * 64 + lane;
            const unsigned long long mk = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0);
            if (lane == 0) kmask[t] = mk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // published by the first barrier of the loop
    }
    if (nkt > 0) dma_tile(smem, smem + 32768, 0u, 0u);
    __builtin_amdgcn_s_waitcnt(0x0F70);    // every load hipcc knows about is retired here (see version 2)
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // tile kt has landed for every wave; every wave is done with tile kt-1
        asm volatile("" ::: "memory");
        if (kt + 1 < nkt)
            dma_tile(smem + (cur ^ 1) * 16384, smem + 32768 + (cur ^ 1) * 16384, (uint32_t)(kt + 1) * 64u * krow, (uint32_t)(kt + 1) * 64u * vrow);
        const int k0 = kt * 64;
        const int wq0 = q0 + wave * 16 + off;  // first query of the wave, in key coordinates
        if (a.causal && k0 > wq0 + 15) continue;  // whole tile above this wave's diagonal
        const char* Kc = smem + cur * 16384;
        const char* Vc = smem + 32768 + cur * 16384;
        f32x4_t s[4];
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk) {
            s[kbk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s[kbk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(Kc + kbk * 4096 + (kfo ^ (ks << 6))), qf[ks], s[kbk], 0, 0, 0);
        }
        const float base = sl2 * (float)(k0 + 4 * g - (a.Sk - 1));
        float mx = -INFINITY;
        const bool interior = kv == nullptr && k0 + 63 < a.Sk && (!a.causal || k0 + 63 <= wq0);
        if (interior) {
#pragma unroll
            for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = fmaf(s[kbk][r], sc2, fmaf(sl2, (float)(16 * kbk + r), base));
                    s[kbk][r] = x;
                    mx = fmaxf(mx, x);
                }
        } else {
            // visibility of the lane's 16 keys (16 kbk + 4 g + r) as one word: causal / sequence-end limit ANDed with the key-padding bits
            const int kend = a.Sk - 1 - k0, rel = qi + off - k0;
            const int limh = (a.causal && rel < kend ? rel : kend) - 4 * g;
            unsigned long long vmh = ~0ull;
            if (kmask_lds) {
                vmh = kmask[kt] >> (4 * g);
            } else if (kv) {
                const int jj = k0 + lane;
                vmh = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0) >> (4 * g);
            }
            const unsigned long long okm = limh < 0 ? 0ull : (limh >= 63 ? vmh : vmh & ((2ull << limh) - 1ull));
            const unsigned okm_lo = (unsigned)okm, okm_hi = (unsigned)(okm >> 32);
#pragma unroll
            for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cidx = 16 * kbk + r;
                    float x = fmaf(s[kbk][r], sc2, fmaf(sl2, (float)cidx, base));
                    const unsigned mk = (unsigned)__builtin_amdgcn_sbfe((int)(cidx < 32 ? okm_lo : okm_hi), cidx & 31, 1);
                    x = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, x) & mk) | (~mk & 0xFF800000u));
                    s[kbk][r] = x;
                    mx = fmaxf(mx, x);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // lazy rescale: only when some row's maximum grew by more than 8 (log2 units) -- wave-uniform decision
        if (__ballot(mx > m + 8.0f) != 0ull) {
            const float mnew = fmaxf(m, mx);
            const float muse = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = __builtin_amdgcn_exp2f(m - muse);
            m = mnew;
            lsum *= alpha;
#pragma unroll
            for (int db = 0; db < 8; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[db][r] *= alpha;
        }
        const float muse = m == -INFINITY ? 0.f : m;
#pragma unroll
        for (int kbk = 0; kbk < 4; ++kbk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kbk][r] - muse);
                s[kbk][r] = p;
                lsum += p;
            }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8_t pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pf[r] = (__bf16)s[2 * st][r]; pf[4 + r] = (__bf16)s[2 * st + 1][r]; }
#pragma unroll
            for (int db = 0; db < 8; ++db) {
                const char* tp = Vc + st * 8192 + (vto ^ (db << 5));
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)tp);
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tp + 4096));
                const s16x8_t vfr = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vfr), pf, o[db], 0, 0, 0);
            }
        }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
    if (qi < a.Sq) {
        bf16_t* rowp = a.o + b * a.os.b + hd * a.os.h + (int64_t)qi * a.os.s;
#pragma unroll
        for (int db = 0; db < 8; ++db) {
            uint2 w;
            w.x = pack2bf(o[db][0] * inv, o[db][1] * inv);
            w.y = pack2bf(o[db][2] * inv, o[db][3] * inv);
            *reinterpret_cast<uint2*>(rowp + 16 * db + 4 * g) = w;
        }
        if (g == 0) a.lse[((int64_t)b * a.H + hd) * a.Sq + qi] = lsum > 0.f ? m * LN2 + logf(lsum) : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------------------
// delta[b,h,q] = sum_d dO . O   (16 lanes per row)
// ------------------------------------------------------------------------------------------------------------
template <int LPR>   // lanes per row: head_dim / 8
__global__ __launch_bounds__(256) void flash_delta_kernel(FlashArgs a) {
    const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + (threadIdx.x / LPR);
    const int c = threadIdx.x & (LPR - 1);
    const int64_t nrows = (int64_t)a.B * a.H * a.Sq;
    float acc = 0.f;
    if (row < nrows) {
        const int qi = (int)(row % a.Sq);
        const int hd = (int)((row / a.Sq) % a.H);
        const int b = (int)(row / ((int64_t)a.Sq * a.H));
        float x[8], y[8];
        Vec8<bf16_t>::load(a.dout + b * a.dos.b + hd * a.dos.h + (int64_t)qi * a.dos.s + 8 * c, x);
        Vec8<bf16_t>::load(a.o + b * a.os.b + hd * a.os.h + (int64_t)qi * a.os.s + 8 * c, y);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += x[i] * y[i];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row < nrows && c == 0) {
        a.delta[row] = acc;
        const float l = a.lse[row];
        a.delta[nrows + row] = l == -INFINITY ? INFINITY : l * LOG2E;  // exp2(x - inf) = 0: a dead row contributes nothing
    }
}

// ------------------------------------------------------------------------------------------------------------
// dQ: same decomposition as the forward; per key tile  S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);  // [64][LDK]
    bf16_t* Vs = Ks + 64 * LDK;                    // [64][LDK]
    bf16_t* Kt = Vs + 64 * LDK;                    // [64][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * 128;
    const int qi = q0 + wave * 32 + ql;
    const int qc = qi < a.Sq ? qi : a.Sq - 1;
    const int off = a.Sk - a.Sq;
    const bf16_t* qp = a.q + b * a.qs.b + hd * a.qs.h + (int64_t)qc * a.qs.s;
    const bf16_t* dop = a.dout + b * a.dos.b + hd * a.dos.h + (int64_t)qc * a.dos.s;
    bf16x8_t qf[8], dof[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        qf[c] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * c + 8 * h2);
        dof[c] = *reinterpret_cast<const bf16x8_t*>(dop + 16 * c + 8 * h2);
    }
    const int64_t srow = ((int64_t)b * a.H + hd) * a.Sq + qc;
    const float lse_n = a.lse[srow];
    const bool live = qi < a.Sq && lse_n != -INFINITY;
    const float lse2 = lse_n * LOG2E, dl = a.delta[srow];
    const bf16_t* kb = a.k + b * a.ks.b + hd * a.ks.h;
    const bf16_t* vb = a.v + b * a.vs.b + hd * a.vs.h;
    const uint8_t* kv = a.kvalid ? a.kvalid + (int64_t)b * a.Sk : nullptr;
    const float sc2 = a.scale * LOG2E, sl2 = a.slopes ? a.slopes[hd] * LOG2E : 0.f;
    int nkt = (a.Sk + 63) >> 6;
    if (a.causal) {
        const int qmax = (q0 + 127 < a.Sq ? q0 + 127 : a.Sq - 1) + off;
        const int lim = qmax < 0 ? 0 : (qmax >> 6) + 1;
        nkt = nkt < lim ? nkt : lim;
    }
    uint4 kr[4], vr[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + 256 * i, row = ch >> 4, c16 = ch & 15, key = kt * 64 + row;
            if (key < a.Sk) {
                kr[i] = *reinterpret_cast<const uint4*>(kb + (int64_t)key * a.ks.s + c16 * 8);
                vr[i] = *reinterpret_cast<const uint4*>(vb + (int64_t)key * a.vs.s + c16 * 8);
            } else {
                kr[i] = make_uint4(0, 0, 0, 0);
                vr[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = tid + 256 * i, row = ch >> 4, c16 = ch & 15;
            *reinterpret_cast<uint4*>(Ks + row * LDK + c16 * 8) = kr[i];
            *reinterpret_cast<uint4*>(Kt + row * LDT + c16 * 8) = kr[i];
            *reinterpret_cast<uint4*>(Vs + row * LDK + c16 * 8) = vr[i];
        }
    };
    const int troff = tr_lane_off(lane);
    f32x16_t dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = zero16();
    if (nkt > 0) gload(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        swrite();
        __syncthreads();
        if (kt + 1 < nkt) gload(kt + 1);
        const int k0 = kt * 64;
        if (a.causal && k0 > q0 + wave * 32 + 31 + off) continue;
        const int kend = a.Sk - 1 - k0, rel = qi + off - k0;
        const int limh = live ? (a.causal && rel < kend ? rel : kend) - 4 * h2 : -1;
        const float base = sl2 * (float)(k0 + 4 * h2 - (a.Sk - 1));
        unsigned long long vmh = ~0ull;
        if (kv) {
            const int jj = k0 + lane;
            vmh = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0) >> (4 * h2);
        }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            f32x16_t s = zero16(), dp = zero16();
            const bf16_t* krow = Ks + (32 * kbk + ql) * LDK + 8 * h2;
            const bf16_t* vrow = Vs + (32 * kbk + ql) * LDK + 8 * h2;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(krow + 16 * c), qf[c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(vrow + 16 * c), dof[c], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                const float x = fmaf(s[r], sc2, fmaf(sl2, (float)cidx, base));
                const bool ok = cidx <= limh && ((vmh >> cidx) & 1ull);
                const float p = ok ? __builtin_amdgcn_exp2f(x - lse2) : 0.f;
                s[r] = p * (dp[r] - dl);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8_t dsf = pack8(s, 8 * c);
                const bf16_t* tp = Kt + troff + (32 * kbk + 16 * c) * LDT;
#pragma unroll
                for (int db = 0; db < 4; ++db) dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tp + 32 * db), dsf, dq[db], 0, 0, 0);
            }
        }
    }
    if (qi < a.Sq) store_dt(a.dq + b * a.dqs.b + hd * a.dqs.h + (int64_t)qi * a.dqs.s, dq, a.scale, h2);
}

// ------------------------------------------------------------------------------------------------------------
// dK, dV: grid (ceil(Sk/128), H, B), wave w owns keys k0 + 32w .. +31 (K, V fragments in registers); query tiles of 32
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);  // [32][LDK]
    bf16_t* dOs = Qs + 32 * LDK;                   // [32][LDK]
    bf16_t* Qt = dOs + 32 * LDK;                   // [32][LDT]
    bf16_t* dOt = Qt + 32 * LDT;                   // [32][LDT]
    float* lse_s = reinterpret_cast<float*>(dOt + 32 * LDT);  // [32] (log2 domain; +inf marks a dead row)
    float* dl_s = lse_s + 32;                                  // [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, hd = blockIdx.y, k0 = blockIdx.x * 128;
    const int kw = k0 + wave * 32, kj = kw + ql;
    const int off = a.Sk - a.Sq;
    const int kc = kj < a.Sk ? kj : a.Sk - 1;
    const bf16_t* kp = a.k + b * a.ks.b + hd * a.ks.h + (int64_t)kc * a.ks.s;
    const bf16_t* vp = a.v + b * a.vs.b + hd * a.vs.h + (int64_t)kc * a.vs.s;
    bf16x8_t kf[8], vf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        kf[c] = *reinterpret_cast<const bf16x8_t*>(kp + 16 * c + 8 * h2);
        vf[c] = *reinterpret_cast<const bf16x8_t*>(vp + 16 * c + 8 * h2);
    }
    bool kok = kj < a.Sk;
    if (a.kvalid) kok = kok && a.kvalid[(int64_t)b * a.Sk + kc] != 0;
    const float sc2 = a.scale * LOG2E, sl2 = a.slopes ? a.slopes[hd] * LOG2E : 0.f;
    const float bias2 = sl2 * (float)(kj - (a.Sk - 1));
    const bf16_t* qb = a.q + b * a.qs.b + hd * a.qs.h;
    const bf16_t* dob = a.dout + b * a.dos.b + hd * a.dos.h;
    const float* lseb = a.lse + ((int64_t)b * a.H + hd) * a.Sq;
    const float* dlb = a.delta + ((int64_t)b * a.H + hd) * a.Sq;
    const int nqt = (a.Sq + 31) >> 5;
    int qt0 = 0;
    if (a.causal) {
        const int imin = k0 - off;
        qt0 = imin > 0 ? (imin >> 5) : 0;
    }
    uint4 qr[2], dr[2];
    float sr = 0.f, sd = 0.f;
    auto gload = [&](int qt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = tid + 256 * i, row = ch >> 4, c16 = ch & 15, qrow = qt * 32 + row;
            if (qrow < a.Sq) {
                qr[i] = *reinterpret_cast<const uint4*>(qb + (int64_t)qrow * a.qs.s + c16 * 8);
                dr[i] = *reinterpret_cast<const uint4*>(dob + (int64_t)qrow * a.dos.s + c16 * 8);
            } else {
                qr[i] = make_uint4(0, 0, 0, 0);
                dr[i] = make_uint4(0, 0, 0, 0);
            }
        }
        if (tid < 32) {
            const int qrow = qt * 32 + tid;
            const float l = qrow < a.Sq ? lseb[qrow] : -INFINITY;
            sr = l == -INFINITY ? INFINITY : l * LOG2E;   // dead row -> exp2(x - inf) = 0
            sd = qrow < a.Sq ? dlb[qrow] : 0.f;
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = tid + 256 * i, row = ch >> 4, c16 = ch & 15;
            *reinterpret_cast<uint4*>(Qs + row * LDK + c16 * 8) = qr[i];
            *reinterpret_cast<uint4*>(Qt + row * LDT + c16 * 8) = qr[i];
            *reinterpret_cast<uint4*>(dOs + row * LDK + c16 * 8) = dr[i];
            *reinterpret_cast<uint4*>(dOt + row * LDT + c16 * 8) = dr[i];
        }
        if (tid < 32) { lse_s[tid] = sr; dl_s[tid] = sd; }
    };
    const int troff = tr_lane_off(lane);
    f32x16_t dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk[db] = zero16(); dv[db] = zero16(); }
    if (qt0 < nqt) gload(qt0);
    for (int qt = qt0; qt < nqt; ++qt) {
        __syncthreads();
        swrite();
        __syncthreads();
        if (qt + 1 < nqt) gload(qt + 1);
        const int i0 = qt * 32;
        if (a.causal && i0 + 31 + off < kw) continue;  // every query of the tile precedes this wave's keys
        f32x16_t s = zero16(), dp = zero16();
        {
            const bf16_t* qrow = Qs + ql * LDK + 8 * h2;
            const bf16_t* drow = dOs + ql * LDK + 8 * h2;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(qrow + 16 * c), kf[c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(drow + 16 * c), vf[c], dp, 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 8 * g + 4 * h2);
            const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 8 * g + 4 * h2);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                const int i = i0 + 8 * g + 4 * h2 + e;
                const bool ok = kok && i < a.Sq && (!a.causal || kj <= i + off);
                const float p = ok ? __builtin_amdgcn_exp2f(s[r] * sc2 + bias2 - lv[e]) : 0.f;
                s[r] = p;
                dp[r] = p * (dp[r] - dvv[e]);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bf16x8_t pf = pack8(s, 8 * c), dsf = pack8(dp, 8 * c);
            const bf16_t* tq = Qt + troff + (16 * c) * LDT;
            const bf16_t* td = dOt + troff + (16 * c) * LDT;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dv[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(td + 32 * db), pf, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(tq + 32 * db), dsf, dk[db], 0, 0, 0);
            }
        }
    }
    if (kj < a.Sk) {
        store_dt(a.dk + b * a.dks.b + hd * a.dks.h + (int64_t)kj * a.dks.s, dk, a.scale, h2);
        store_dt(a.dv + b * a.dvs.b + hd * a.dvs.h + (int64_t)kj * a.dvs.s, dv, 1.0f, h2);
    }
}

// ------------------------------------------------------------------------------------------------------------
// backward, version 2 (LDS-DMA tiles, one raw barrier per tile, double buffering; see flash_fwd2_kernel).
// A tile that is read BOTH as row fragments (ds_read_b128) and through the transpose read uses the swizzle
//   16-B slot ^= pi(row & 15),  pi(x) = ((x & 3) << 2) | (x >> 2)
// pi is a bijection on 0..15 (16 rows of a ds_read_b128 lane group -> 16 distinct slots) whose upper two bits are
// row & 3 (the 4 rows of a transpose read -> 4 distinct 64-B blocks): one LDS copy is conflict-free for both.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pi16(int x) { return ((x & 3) << 2) | ((x >> 2) & 3); }

// per-lane byte offsets of the two transpose reads (rows +0 and +8 of a 16-row k chunk) inside a pi-swizzled tile
__device__ __forceinline__ void tr_pi_offsets(int lane, int& o1, int& o2) {
    const int gi = lane & 15, gg = lane >> 4, j = gi >> 2, h = gg >> 1;
    const int lo = 2 * (gg & 1) + ((gi & 3) >> 1);
    o1 = (4 * h + j) * 256 + (j << 6) + 16 * (lo ^ h) + 8 * (gi & 1);
    o2 = (4 * h + j + 8) * 256 + (j << 6) + 16 * (lo ^ h ^ 2) + 8 * (gi & 1);
}
__device__ __forceinline__ bf16x8_t tr_pi_frag(const char* tile, int o1, int o2, int db, int rowbase) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + (o1 ^ (db << 6)) + rowbase * 256));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + (o2 ^ (db << 6)) + rowbase * 256));
    const s16x8_t r = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, r);
}

template <bool LPT, bool PAIR = false, bool ALIBI = true>   // PAIR: two 64-wide heads per workgroup, see flash_fwd2_kernel
__global__ __launch_bounds__(256, 2) void flash_bwd_dq2_kernel(FlashArgs a) {
    BLK_BEGIN(1);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1, 16 KB each
    const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nqb = (a.Sq + 127) >> 7;   // LPT: see flash_fwd2_kernel
    const LptIdx li = lpt_decode((int)blockIdx.x, nqb, a.lpt_group);
    const int HB = PAIR ? a.H >> 1 : a.H, hm = PAIR ? 2 : 1;
    const int b = LPT ? li.bh / HB : blockIdx.z, hd = LPT ? li.bh % HB : blockIdx.y;
    const int q0 = (LPT ? nqb - 1 - li.rank : (int)blockIdx.x) * 128;
    const int qi = q0 + wave * 32 + ql;
    const int qc = qi < a.Sq ? qi : a.Sq - 1;
    const int off = a.Sk - a.Sq;
    const bf16_t* qp = a.q + b * a.qs.b + hd * hm * a.qs.h + (int64_t)qc * a.qs.s;
    const bf16_t* dop = a.dout + b * a.dos.b + hd * hm * a.dos.h + (int64_t)qc * a.dos.s;
    const int64_t qpo = PAIR ? a.qs.h - 64 : 0, dopo = PAIR ? a.dos.h - 64 : 0;
    bf16x8_t qf[8], dof[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        qf[c] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * c + 8 * h2 + (c >= 4 ? qpo : 0));
        dof[c] = *reinterpret_cast<const bf16x8_t*>(dop + 16 * c + 8 * h2 + (c >= 4 ? dopo : 0));
    }
    const int64_t nrows = (int64_t)a.B * a.H * a.Sq;
    const int64_t srow = ((int64_t)b * a.H + hd * hm) * a.Sq + qc;
    float lse2, dl, lse2B = 0.f, dlB = 0.f;
    if (a.fuse_delta) {
        // delta[q] = sum_d dO[q, d] O[q, d] from the dO fragments this lane already holds and the matching O fragments (the lane pair
        // ql / ql + 32 covers a row: one cross-half add), and the log2-domain LSE; written out for the dK/dV kernel, which runs AFTER this
        // one in that mode.  Replaces the flash_delta launch (18 us per layer at C2: 67 MB read for 1 MB of results).
        const bf16_t* op = a.o + b * a.os.b + hd * hm * a.os.h + (int64_t)qc * a.os.s;
        const int64_t opo = PAIR ? a.os.h - 64 : 0;
        float accA = 0.f, accB = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(op + 16 * c + 8 * h2 + (c >= 4 ? opo : 0));
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) t = fmaf(static_cast<float>(dof[c][i]), static_cast<float>(of[i]), t);
            if (PAIR && c >= 4) accB += t; else accA += t;
        }
        accA += __shfl_xor(accA, 32, 64);
        const float lA = a.lse[srow];
        dl = accA;
        lse2 = qi < a.Sq ? (lA == -INFINITY ? INFINITY : lA * LOG2E) : INFINITY;
        if (PAIR) {
            accB += __shfl_xor(accB, 32, 64);
            const float lB = a.lse[srow + a.Sq];
            dlB = accB;
            lse2B = qi < a.Sq ? (lB == -INFINITY ? INFINITY : lB * LOG2E) : INFINITY;
        }
        if (h2 == 0 && qi < a.Sq) {
            a.delta[srow] = dl;
            a.delta[nrows + srow] = lse2;
            if (PAIR) { a.delta[srow + a.Sq] = dlB; a.delta[nrows + srow + a.Sq] = lse2B; }
        }
    } else {
        lse2 = qi < a.Sq ? a.delta[nrows + srow] : INFINITY;
        dl = a.delta[srow];
        if (PAIR) { lse2B = qi < a.Sq ? a.delta[nrows + srow + a.Sq] : INFINITY; dlB = a.delta[srow + a.Sq]; }
    }
    const bf16_t* kb = a.k + b * a.ks.b + hd * hm * a.ks.h;
    const bf16_t* vb = a.v + b * a.vs.b + hd * hm * a.vs.h;
    const uint8_t* kv = a.kvalid ? a.kvalid + (int64_t)b * a.Sk : nullptr;
    const float sc2 = a.scale * LOG2E, sl2 = a.slopes ? a.slopes[hd * hm] * LOG2E : 0.f;
    const float sl2B = PAIR && a.slopes ? a.slopes[hd * 2 + 1] * LOG2E : 0.f;
    int nkt = (a.Sk + 63) >> 6;
    if (a.causal) {
        const int qmax = (q0 + 127 < a.Sq ? q0 + 127 : a.Sq - 1) + off;
        const int lim = qmax < 0 ? 0 : (qmax >> 6) + 1;
        nkt = nkt < lim ? nkt : lim;
    }
    const uint32_t krow = (uint32_t)(a.ks.s * 2), vrow = (uint32_t)(a.vs.s * 2);
    const uint32_t kpb = PAIR ? (uint32_t)((a.ks.h - 64) * 2) : 0u, vpb = PAIR ? (uint32_t)((a.vs.h - 64) * 2) : 0u;
    const u32x4_t rk = make_rsrc4(kb, (uint32_t)((int)((uint32_t)(a.Sk - 1) * krow + 256u + kpb)));
    const u32x4_t rv = make_rsrc4(vb, (uint32_t)((int)((uint32_t)(a.Sk - 1) * vrow + 256u + vpb)));
    uint32_t vk[4], vv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = 4 * (wave * 4 + i) + (lane >> 4), p = lane & 15;
        const int sk = p ^ pi16(rl & 15), sv = p ^ (rl & 15);
        vk[i] = (uint32_t)rl * krow + (uint32_t)(sk << 4) + (sk >= 8 ? kpb : 0u);
        vv[i] = (uint32_t)rl * vrow + (uint32_t)(sv << 4) + (sv >= 8 ? vpb : 0u);
    }
    const int kfo = ql * 256 + ((h2 ^ pi16(ql & 15)) << 4);
    const int vfo = ql * 256 + ((h2 ^ (ql & 15)) << 4);
    int to1, to2;
    tr_pi_offsets(lane, to1, to2);
    f32x16_t dq[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dq[db] = zero16();
    // Key-padding masks: one 64-bit word per key tile, parked behind the tile buffers.  Reading the validity bytes inside the loop is a
    // load hipcc counts: it waits vmcnt(0) for it, which also drains the asm LDS-DMA of the next tile (every padded batch paid that).
    unsigned long long* const kmask = reinterpret_cast<unsigned long long*>(smem + 65536);
    const bool kmask_lds = kv != nullptr && nkt <= KMASK_TILES;
    if (kmask_lds) {
        for (int t = wave; t < nkt; t += 4) {
            const int jj = t * 64 + lane;
            const unsigned long long mk = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0);
            if (lane == 0) kmask[t] = mk;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // published by the first barrier of the loop
    }
    if (nkt > 0) flash_dma_tile(rk, rv, smem, smem + 32768, vk, vv, wave, 0u, 0u);
    // every load hipcc knows about (the register-resident fragments above) is retired HERE, through the builtin its scoreboard models:
    // otherwise it re-issues its counted waits for them -- down to vmcnt(0) -- in front of the MFMAs of EVERY iteration, and those
    // waits also drain the asm LDS-DMA of the next tile.  (vmcnt(0), expcnt / lgkmcnt untouched: simm16 0x0F70 on gfx9.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int k0 = kt * 64;
        const int wq0 = q0 + wave * 32 + off;
        // (the forward's spread of the DMA pieces over the MFMAs was tried here too in round 6: 137-138 vs 133-134 us per backward -- not kept)
        const bool dma_next = kt + 1 < nkt;
        const bool skip_tile = a.causal && k0 > wq0 + 31;
        char* const kdn = smem + (cur ^ 1) * 16384;
        char* const vdn = smem + 32768 + (cur ^ 1) * 16384;
        const uint32_t ksn = (uint32_t)(kt + 1) * 64u * krow, vsn = (uint32_t)(kt + 1) * 64u * vrow;
        if (dma_next) flash_dma_tile(rk, rv, kdn, vdn, vk, vv, wave, ksn, vsn);
        if (skip_tile) continue;
        const char* Kc = smem + cur * 16384;
        const char* Vc = smem + 32768 + cur * 16384;
        const int kend = a.Sk - 1 - k0, rel = qi + off - k0;
        const int limh = (a.causal && rel < kend ? rel : kend) - 4 * h2;
        const float base = sl2 * (float)(k0 + 4 * h2 - (a.Sk - 1)) - lse2;
        const bool interior = kv == nullptr && k0 + 63 < a.Sk && (!a.causal || k0 + 63 <= wq0);
        unsigned long long vmh = ~0ull;
        if (kmask_lds) {
            vmh = kmask[kt] >> (4 * h2);
        } else if (kv) {
            const int jj = k0 + lane;
            vmh = __ballot(jj < a.Sk && kv[jj < a.Sk ? jj : 0] != 0) >> (4 * h2);
        }
        if constexpr (PAIR) {
            // bit cidx: the lane's key cidx of the tile is visible (see the forward)
            const unsigned long long okm = limh < 0 ? 0ull : (limh >= 63 ? vmh : vmh & ((2ull << limh) - 1ull));
            const unsigned okm_lo = (unsigned)okm, okm_hi = (unsigned)(okm >> 32);
            auto part = [&](auto KBK, auto E) {
                constexpr int kbk = decltype(KBK)::value, e = decltype(E)::value;
                f32x16_t s = zero16(), dp = zero16();
#pragma unroll
                for (int c = 4 * e; c < 4 * e + 4; ++c) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Kc + kbk * 8192 + (kfo ^ (32 * c))), qf[c], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Vc + kbk * 8192 + (vfo ^ (32 * c))), dof[c], dp, 0, 0, 0);
                }
                const float sle = e ? sl2B : sl2, dle = e ? dlB : dl;
                const float be = sle * (float)(k0 + 4 * h2 - (a.Sk - 1)) - (e ? lse2B : lse2);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                    float p = __builtin_amdgcn_exp2f(ALIBI ? fmaf(s[r], sc2, fmaf(sle, (float)cidx, be)) : fmaf(s[r], sc2, be));
                    const unsigned mk = (unsigned)__builtin_amdgcn_sbfe((int)(cidx < 32 ? okm_lo : okm_hi), cidx & 31, 1);
                    p = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p) & mk);
                    s[r] = p * (dp[r] - dle);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const bf16x8_t dsf = pack8(s, 8 * c);
#pragma unroll
                    for (int dbl = 0; dbl < 2; ++dbl)
                        dq[2 * e + dbl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pi_frag(Kc, to1, to2, 2 * e + dbl, 32 * kbk + 16 * c), dsf,
                                                                                  dq[2 * e + dbl], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // one (key block, head) at a time (without it: same time, 450 vs 451 us per backward)
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            part(I0{}, I0{}); part(I0{}, I1{});
            part(I1{}, I0{}); part(I1{}, I1{});
            continue;
        }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
            f32x16_t s = zero16(), dp = zero16();
            FPRIO(2, 1);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Kc + kbk * 8192 + (kfo ^ (32 * c))), qf[c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(Vc + kbk * 8192 + (vfo ^ (32 * c))), dof[c], dp, 0, 0, 0);
            }
            FPRIO(2, 0);
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, fmaf(sl2, (float)cidx, base)));
                    s[r] = p * (dp[r] - dl);
                }
            } else {
                // (round 4: one visibility word per lane and tile, v_bfe_i32 + v_and_b32 per score: see the forward)
                const unsigned long long okm = limh < 0 ? 0ull : (limh >= 63 ? vmh : vmh & ((2ull << limh) - 1ull));
                const unsigned okm_lo = (unsigned)okm, okm_hi = (unsigned)(okm >> 32);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cidx = 32 * kbk + (r & 3) + 8 * (r >> 2);
                    float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc2, fmaf(sl2, (float)cidx, base)));
                    const unsigned mk = (unsigned)__builtin_amdgcn_sbfe((int)(cidx < 32 ? okm_lo : okm_hi), cidx & 31, 1);
                    p = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, p) & mk);
                    s[r] = p * (dp[r] - dl);
                }
            }
            FPRIO(2, 1);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8_t dsf = pack8(s, 8 * c);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_pi_frag(Kc, to1, to2, db, 32 * kbk + 16 * c), dsf, dq[db], 0, 0, 0);
            }
            FPRIO(2, 0);
        }
    }
    if constexpr (PAIR) {
#if OTTER_FLASH_ROWSTORE
        __builtin_amdgcn_s_barrier();
        store_rows_lds<true>(smem + wave * 8192, a.dq + b * a.dqs.b + hd * 2 * a.dqs.h, a.dqs.s, q0 + wave * 32, a.Sq, dq, a.scale, a.scale, lane, a.dqs.h - 64);
#else
        if (qi < a.Sq) store_dt_pair(a.dq + b * a.dqs.b + hd * 2 * a.dqs.h + (int64_t)qi * a.dqs.s, dq, a.scale, a.scale, h2, a.dqs.h - 64);
#endif
        return;
    }
#if OTTER_FLASH_ROWSTORE
    __builtin_amdgcn_s_barrier();
    store_rows_lds<false>(smem + wave * 8192, a.dq + b * a.dqs.b + hd * a.dqs.h, a.dqs.s, q0 + wave * 32, a.Sq, dq, a.scale, a.scale, lane, 0);
#else
    if (qi < a.Sq) store_dt(a.dq + b * a.dqs.b + hd * a.dqs.h + (int64_t)qi * a.dqs.s, dq, a.scale, h2);
#endif
    BLK_END(1);
}

// PERS (round 4): ONE workgroup per (batch, head) walks ALL key blocks of that head (grid = B * H, one per CU at C2).  The launch picture
// (tools/flash_launch_picture.py, profiles/r04_flash_launch_picture.txt) showed every workgroup of this kernel paying ~10 us of fixed cost
// -- cold K / V fragment loads, pipeline fill, 32 sixteen-byte-per-row stores and their drain, the dispatch gap -- around 1.4 us per query
// tile, four workgroups one after the other on each CU: 40 of 101 us.  Here the Q / dO ring simply keeps running across key blocks (the
// tile after the last one of block r is the first one of block r+1), the next block's K / V rows are DMA-staged into wave-private LDS one
// whole block ahead, and dK / dV leave through an LDS transpose as full 256-B rows.
constexpr int DKV_STG = 52224;              // PERS: K | V of the NEXT key block, 2 x 8 KB per wave
constexpr int DKV_TRN = DKV_STG + 65536;    // PERS: 8 KB per wave for the row-major dK / dV tile on its way out
constexpr int DKV_PERS_SMEM = DKV_TRN + 32768;

template <int MINB, bool LPT, bool PAIR = false, bool PERS = false>   // PAIR: two 64-wide heads per workgroup, see flash_fwd2_kernel
__global__ __launch_bounds__(256, MINB) void flash_bwd_dkv2_kernel(FlashArgs a) {
    BLK_BEGIN(2);
    // 3-stage ring, two query tiles in flight (a 32-row tile is only ~1k MFMA cycles of work, less than the DMA latency):
    // Q[3] | dO[3] (8 KB each) | lse2[3][64] | delta[3][64]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DSTAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LPT: 1-D grid walked key block by key block -- under the causal mask key block 0 has the most query tiles, so the longest blocks
    // are dispatched first and the launch ends on the short ones
    const int nkb = (a.Sk + 127) >> 7;
    const LptIdx li = lpt_decode((int)blockIdx.x, nkb, a.lpt_group);
    const int HB = PAIR ? a.H >> 1 : a.H, hm = PAIR ? 2 : 1;
    const int b = PERS ? (int)blockIdx.x / HB : (LPT ? li.bh / HB : (int)blockIdx.z), hd = PERS ? (int)blockIdx.x % HB : (LPT ? li.bh % HB : (int)blockIdx.y);
    int k0 = PERS ? 0 : (LPT ? li.rank : (int)blockIdx.x) * 128;   // PERS: key block 0 first (the most query tiles under the causal mask)
    int kw = k0 + wave * 32, kj = kw + ql;
    const int off = a.Sk - a.Sq;
    const int kc = kj < a.Sk ? kj : a.Sk - 1;
    const bf16_t* kp = a.k + b * a.ks.b + hd * hm * a.ks.h + (int64_t)kc * a.ks.s;
    const bf16_t* vp = a.v + b * a.vs.b + hd * hm * a.vs.h + (int64_t)kc * a.vs.s;
    const int64_t kpo = PAIR ? a.ks.h - 64 : 0, vpo = PAIR ? a.vs.h - 64 : 0;
    bf16x8_t kf[8], vf[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        kf[c] = *reinterpret_cast<const bf16x8_t*>(kp + 16 * c + 8 * h2 + (c >= 4 ? kpo : 0));
        vf[c] = *reinterpret_cast<const bf16x8_t*>(vp + 16 * c + 8 * h2 + (c >= 4 ? vpo : 0));
    }
    bool kok = kj < a.Sk;
    if (a.kvalid) kok = kok && a.kvalid[(int64_t)b * a.Sk + kc] != 0;
    unsigned kvbits = 0xffffffffu;   // PERS: this lane's key-padding bit of every key block (<= 32 blocks), read once
    if constexpr (PERS) {
        if (a.kvalid) {
            kvbits = 0u;
            for (int r = 0; r < nkb; ++r) {
                const int kk = r * 128 + wave * 32 + ql;
                if (a.kvalid[(int64_t)b * a.Sk + (kk < a.Sk ? kk : a.Sk - 1)] != 0) kvbits |= 1u << r;
            }
        }
    }
    const float sc2 = a.scale * LOG2E, sl2 = a.slopes ? a.slopes[hd * hm] * LOG2E : 0.f;
    const float sl2B = PAIR && a.slopes ? a.slopes[hd * 2 + 1] * LOG2E : 0.f;
    // (__fmul_rn: a product the compiler may not contract into the `bias2 - lse` of the tile loop -- with -ffp-contract=fast it did so in
    //  the per-block instantiations, where bias2 is loop-invariant, and not in PERS: last-bit differences between the two forms)
    float bias2 = __fmul_rn(sl2, (float)(kj - (a.Sk - 1)));
    float bias2B = __fmul_rn(sl2B, (float)(kj - (a.Sk - 1)));
    const bf16_t* qb = a.q + b * a.qs.b + hd * hm * a.qs.h;
    const bf16_t* dob = a.dout + b * a.dos.b + hd * hm * a.dos.h;
    const int64_t nrows = (int64_t)a.B * a.H * a.Sq;
    const float* dlb = a.delta + ((int64_t)b * a.H + hd * hm) * a.Sq;
    const float* lsb = dlb + nrows;
    const int nqt = (a.Sq + 31) >> 5;
    auto first_tile_of = [&](int kb0) {   // first query tile that sees any key of the 128-key block starting at kb0
        const int imin = kb0 - off;
        return a.causal && imin > 0 ? (imin >> 5) : 0;
    };
    int qt0 = first_tile_of(k0);
    // PERS: first tile of the NEXT key block (the ring runs on into it); past the last block: tiles beyond Sq (the DMA reads zeros)
    int qtn = PERS ? (nkb > 1 ? first_tile_of(128) : nqt + 2) : 0;
    int rank = 0;
    const uint32_t qrow = (uint32_t)(a.qs.s * 2), dorow = (uint32_t)(a.dos.s * 2);
    const uint32_t qpb = PAIR ? (uint32_t)((a.qs.h - 64) * 2) : 0u, dpb = PAIR ? (uint32_t)((a.dos.h - 64) * 2) : 0u;
    const u32x4_t rq = make_rsrc4(qb, (uint32_t)((int)((uint32_t)(a.Sq - 1) * qrow + 256u + qpb)));
    const u32x4_t rdo = make_rsrc4(dob, (uint32_t)((int)((uint32_t)(a.Sq - 1) * dorow + 256u + dpb)));
    // PAIR: the statistics words of a tile are lanes 0-31 = head A's 32 rows, lanes 32-63 = head B's (the next row of the [B,H,Sq] arrays);
    // head A's rows past Sq then read head B's first words instead of zeros: finite (delta) or +inf (log2 lse of a dead row), and masked
    const u32x4_t rls = make_rsrc4(lsb, (uint32_t)(a.Sq * 4 * hm));
    const u32x4_t rdl = make_rsrc4(dlb, (uint32_t)(a.Sq * 4 * hm));
    const uint32_t stat_voff = PAIR ? (uint32_t)((lane & 31) * 4 + (lane >> 5) * a.Sq * 4) : (uint32_t)lane * 4u;
    uint32_t vq[2], vd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = 4 * (wave * 2 + i) + (lane >> 4), p = lane & 15;
        const int sq = p ^ pi16(rl & 15);
        vq[i] = (uint32_t)rl * qrow + (uint32_t)(sq << 4) + (sq >= 8 ? qpb : 0u);
        vd[i] = (uint32_t)rl * dorow + (uint32_t)(sq << 4) + (sq >= 8 ? dpb : 0u);
    }
    char* const stat = smem + 49152;
    // PERS: staging of the next key block's K / V rows (this wave's 32 keys, 2 x 8 DMA pieces of four 256-B rows), same slot swizzle as the
    // Q / dO tiles so the fragments come out through the same `qfo ^ 32c` addresses
    char* const stK = smem + DKV_STG + wave * 16384;
    char* const stV = stK + 8192;
    const uint32_t krow = (uint32_t)(a.ks.s * 2), vrow = (uint32_t)(a.vs.s * 2);
    const uint32_t kpb = PAIR ? (uint32_t)((a.ks.h - 64) * 2) : 0u, vpb = PAIR ? (uint32_t)((a.vs.h - 64) * 2) : 0u;
    u32x4_t rks = {0u, 0u, 0u, 0u}, rvs = {0u, 0u, 0u, 0u};
    if constexpr (PERS) {
        rks = make_rsrc4(a.k + b * a.ks.b + hd * hm * a.ks.h, (uint32_t)((int)((uint32_t)(a.Sk - 1) * krow + 256u + kpb)));
        rvs = make_rsrc4(a.v + b * a.vs.b + hd * hm * a.vs.h, (uint32_t)((int)((uint32_t)(a.Sk - 1) * vrow + 256u + vpb)));
    }
    auto stage_kv = [&](int kw_next) {
        const int r4 = lane >> 4, pp = lane & 15;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sq = pp ^ ((r4 << 2) | (i & 3));   // = p ^ pi16((4 i + r4) & 15)
            dma16_asm(rks, stK + i * 1024, (uint32_t)r4 * krow + (uint32_t)(sq << 4) + (sq >= 8 ? kpb : 0u), (uint32_t)(kw_next + 4 * i) * krow);
            dma16_asm(rvs, stV + i * 1024, (uint32_t)r4 * vrow + (uint32_t)(sq << 4) + (sq >= 8 ? vpb : 0u), (uint32_t)(kw_next + 4 * i) * vrow);
        }
    };
    auto issue = [&](int qt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            dma16_asm(rq, smem + buf * 8192 + (wave * 2 + i) * 1024, vq[i], (uint32_t)qt * 32u * qrow);
            dma16_asm(rdo, smem + 24576 + buf * 8192 + (wave * 2 + i) * 1024, vd[i], (uint32_t)qt * 32u * dorow);
        }
        if (wave == 0) {  // 64 floats each (the upper 32 belong to the next tile; rows past Sq read as 0 and are masked)
            dma4_asm(rls, stat + buf * 256, stat_voff, (uint32_t)qt * 128u);
            dma4_asm(rdl, stat + 768 + buf * 256, stat_voff, (uint32_t)qt * 128u);
        }
    };
    const int qfo = ql * 256 + ((h2 ^ pi16(ql & 15)) << 4);
    int to1, to2;
    tr_pi_offsets(lane, to1, to2);
    f32x16_t dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk[db] = zero16(); dv[db] = zero16(); }
    if constexpr (PERS) {
        if (nkb > 1) stage_kv(kw + 128);   // older than every ring piece: the loop's counted waits cover it
    }
    issue(qt0, 0);       // unconditional (see the loop): rows past Sq read as zeros
    issue(qt0 + 1, 1);
    // every load hipcc knows about (the register-resident fragments above) is retired HERE, through the builtin its scoreboard models:
    // otherwise it re-issues its counted waits for them -- down to vmcnt(0) -- in front of the MFMAs of EVERY iteration, and those
    // waits also drain the asm LDS-DMA of the next tile.  (vmcnt(0), expcnt / lgkmcnt untouched: simm16 0x0F70 on gfx9.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // Software pipeline (one wave per SIMD: nothing else covers a latency).  Per tile t:
    //   top     transpose reads of tile t issued (they land under the S / dP MFMAs)
    //           S / dP MFMAs of tile t -- their row fragments were read during the previous iteration's dV / dK MFMAs;
    //           the DMA of tile t+2 is issued into the slot of tile t-1 while the matrix pipe works through the queue
    //           softmax of tile t (VALU)
    //   middle  counted wait for the DMA of tile t+1 + barrier (every wave has also finished READING tile t: lgkmcnt(0) first)
    //           row-fragment reads of tile t+1 issued (they land under ...)
    //   bottom  dV / dK MFMAs of tile t
    // The barrier that opens the loop in the straightforward form moved to the middle; tile t's slot is dead after it.
    bf16x8_t qr[8], dr[8];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // prologue: both tiles (the builtin wait above covers them anyway)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const char* Qn = smem;
        const char* Dn = smem + 24576;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qr[c] = *reinterpret_cast<const bf16x8_t*>(Qn + (qfo ^ (32 * c)));
            dr[c] = *reinterpret_cast<const bf16x8_t*>(Dn + (qfo ^ (32 * c)));
        }
    }
    // one DMA piece of tile qt (pieces 0-3: Q / dO halves of this wave's rows; 4-5: the lse / delta words, wave 0 only)
    auto issue_piece = [&](int qt, int buf, int p) {
        if (p < 4) {
            const int i = p >> 1;
            if (p & 1) dma16_asm(rdo, smem + 24576 + buf * 8192 + (wave * 2 + i) * 1024, vd[i], (uint32_t)qt * 32u * dorow);
            else dma16_asm(rq, smem + buf * 8192 + (wave * 2 + i) * 1024, vq[i], (uint32_t)qt * 32u * qrow);
        } else {
            if (p == 4) dma4_asm(rls, stat + buf * 256, stat_voff, (uint32_t)qt * 128u);
            else dma4_asm(rdl, stat + 768 + buf * 256, stat_voff, (uint32_t)qt * 128u);
        }
    };
    // dK / dV MFMAs through asm with AGPR accumulators ("+a"); s_nop 1 covers a VALU-written (cvt_pk) B operand; the _MEM form also orders
    // the statement against the compiler's LDS reads (phase 3 places two row-fragment reads behind every MFMA)
#define MFMA_ACC(ACC_, A_, B_) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC_) : "v"(A_), "v"(B_))
#define MFMA_ACC_MEM(ACC_, A_, B_) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC_) : "v"(A_), "v"(B_) : "memory")
    int cur = 0;
    bool first = false;   // PERS: first tile of a key block other than block 0 (its successor tile was waited for at the switch)
    for (;;) {
    for (int qt = qt0; qt < nqt; ++qt, cur = cur == 2 ? 0 : cur + 1) {
        DSTAMP(1 + 5 * (qt - qt0));
        const int i0 = qt * 32;
        // tile after next: PERS runs on into the next key block's first tiles
        const int qt2 = PERS ? (qt + 2 < nqt ? qt + 2 : qtn + (qt + 2 - nqt)) : qt + 2;
        // (no per-wave skip of tiles that precede the wave's keys: the branch makes hipcc shuttle the 128 accumulator
        //  registers between VGPRs and AGPRs on every iteration, which costs more than the <= 3 masked tiles it saves)
        const char* Qc = smem + cur * 8192;
        const char* Dc = smem + 24576 + cur * 8192;
        const float* lse_s = reinterpret_cast<const float*>(stat + cur * 256);
        const float* dl_s = reinterpret_cast<const float*>(stat + 768 + cur * 256);
        // The DMA of tile qt+2, the mid-iteration wait + barrier and the row-fragment reads of tile qt+1 are UNCONDITIONAL: past the last
        // tile the source rows lie beyond the descriptor's range (the DMA writes zeros into a dead slot) and the fragments read are never
        // used.  Guarding each piece / read on qt+2 < nqt cost 19 scalar branches + ~40 scalar ALU instructions per iteration, and with one
        // wave per SIMD every instruction, scalar or not, is an issue slot of four cycles.
        const int nb = cur == 0 ? 2 : cur - 1;   // slot of tile qt-1 = slot of tile qt+2
        if constexpr (PAIR) {
            // Same pipeline, one head after the other inside the tile: per head 8 S / dP MFMAs (c = 4e .. 4e+3) with that head's eight
            // transpose-read fragments (d blocks 2e, 2e+1) and -- head A -- the DMA pieces of tile qt+2 between them, the softmax (rows 8-15
            // one stage per dV / dK MFMA of the first half), then the four second-half MFMAs; head B's carry the row-fragment reads of tile
            // qt+1 behind the mid-iteration barrier.  The visibility window (lo, hi) is the same for both heads.
            const bool interior = i0 + 31 < a.Sq && (!a.causal || kw + 31 <= i0 + off);  // wave-uniform
            int lo = 0, hi = 32;
            if (!interior) {
                const int vis = a.causal ? kj - off - i0 : 0;
                lo = vis > 0 ? vis : 0;
                hi = a.Sq - i0 < 32 ? a.Sq - i0 : 32;
            }
            if (!kok) hi = 0;
            const unsigned span = hi > lo ? (unsigned)(hi - lo) : 0u;
            const int idx0 = 4 * h2 - lo;
            const int nx = cur == 2 ? 0 : cur + 1;
            const char* Qn = smem + nx * 8192;
            const char* Dn = smem + 24576 + nx * 8192;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float bias_e = e ? bias2B : bias2;
                bf16x8_t tD[2][2], tQ[2][2];
                f32x16_t s, dp;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int c = 4 * e + c4;
                    if (c4 == 0) {
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(qr[c]), "v"(kf[c]));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(dp) : "v"(dr[c]), "v"(vf[c]));
                    } else {
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(qr[c]), "v"(kf[c]));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dp) : "v"(dr[c]), "v"(vf[c]));
                    }
                    {
                        const int cc = c4 >> 1;
                        if ((c4 & 1) == 0) {
                            tD[cc][0] = tr_pi_frag(Dc, to1, to2, 2 * e, 16 * cc);
                            tD[cc][1] = tr_pi_frag(Dc, to1, to2, 2 * e + 1, 16 * cc);
                        } else {
                            tQ[cc][0] = tr_pi_frag(Qc, to1, to2, 2 * e, 16 * cc);
                            tQ[cc][1] = tr_pi_frag(Qc, to1, to2, 2 * e + 1, 16 * cc);
                        }
                    }
                    if (e == 0) issue_piece(qt2, nb, c4);
                    else if (c4 == 0 && wave == 0) { issue_piece(qt2, nb, 4); issue_piece(qt2, nb, 5); }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s), "+v"(dp));
#define DKV_ELEM(R_, LV_, DV_)                                                                         \
    do {                                                                                               \
        const float pe_ = __builtin_amdgcn_exp2f(fmaf(s[R_], sc2, bias_e - (LV_)));                     \
        const bool ok_ = (unsigned)(idx0 + 8 * ((R_) >> 2) + ((R_) & 3)) < span;                       \
        const float p_ = ok_ ? pe_ : 0.f;                                                              \
        s[R_] = p_;                                                                                    \
        dp[R_] = p_ * (dp[R_] - (DV_));                                                                \
    } while (0)
                {
                    float lv[8], dvv[8];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 32 * e + 8 * g + 4 * h2);
                        const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 32 * e + 8 * g + 4 * h2);
                        lv[4 * g] = l4.x; lv[4 * g + 1] = l4.y; lv[4 * g + 2] = l4.z; lv[4 * g + 3] = l4.w;
                        dvv[4 * g] = d4.x; dvv[4 * g + 1] = d4.y; dvv[4 * g + 2] = d4.z; dvv[4 * g + 3] = d4.w;
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) DKV_ELEM(r, lv[r], dvv[r]);
                }
#undef DKV_ELEM
                const bf16x8_t pf0 = pack8(s, 0), dsf0 = pack8(dp, 0);
                __builtin_amdgcn_sched_barrier(0);
                {
                    float lv[8], dvv[8], t[8];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 32 * e + 8 * (g + 2) + 4 * h2);
                        const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 32 * e + 8 * (g + 2) + 4 * h2);
                        lv[4 * g] = l4.x; lv[4 * g + 1] = l4.y; lv[4 * g + 2] = l4.z; lv[4 * g + 3] = l4.w;
                        dvv[4 * g] = d4.x; dvv[4 * g + 1] = d4.y; dvv[4 * g + 2] = d4.z; dvv[4 * g + 3] = d4.w;
                    }
#define PIN8() asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]))
                    asm volatile("" : "+v"(s), "+v"(dp));
                    MFMA_ACC(dv[2 * e], tD[0][0], pf0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = fmaf(s[8 + j], sc2, bias_e - lv[j]);
                    PIN8();
                    MFMA_ACC(dk[2 * e], tQ[0][0], dsf0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = __builtin_amdgcn_exp2f(t[j]);
                    PIN8();
                    MFMA_ACC(dv[2 * e + 1], tD[0][1], pf0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = (unsigned)(idx0 + 8 * ((8 + j) >> 2) + (j & 3)) < span ? t[j] : 0.f;
                    PIN8();
                    MFMA_ACC(dk[2 * e + 1], tQ[0][1], dsf0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        s[8 + j] = t[j];
                        dp[8 + j] = t[j] * (dp[8 + j] - dvv[j]);
                    }
#undef PIN8
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bf16x8_t pf1 = pack8(s, 8), dsf1 = pack8(dp, 8);
                if (e == 1) {
                    if (PERS && first) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); first = false; }
                    else if (wave == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i & 1) MFMA_ACC_MEM(dk[2 * e + (i >> 1)], tQ[1][i >> 1], dsf1);
                    else MFMA_ACC_MEM(dv[2 * e + (i >> 1)], tD[1][i >> 1], pf1);
                    if (e == 1) {
                        qr[2 * i] = *reinterpret_cast<const bf16x8_t*>(Qn + (qfo ^ (32 * (2 * i))));
                        dr[2 * i] = *reinterpret_cast<const bf16x8_t*>(Dn + (qfo ^ (32 * (2 * i))));
                        qr[2 * i + 1] = *reinterpret_cast<const bf16x8_t*>(Qn + (qfo ^ (32 * (2 * i + 1))));
                        dr[2 * i + 1] = *reinterpret_cast<const bf16x8_t*>(Dn + (qfo ^ (32 * (2 * i + 1))));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            continue;
        }
        // ---- phase 1: S^T / dP^T (16 MFMAs, VGPR accumulators through asm: see below).  Every MFMA pair is followed by its share of the
        // other work of the iteration -- two transpose-read fragments of THIS tile (needed in phase 3) and one DMA piece of tile qt+2 --
        // and a sched_barrier pins that order: one wave per SIMD issues in order, so anything issued in a burst (48 LDS reads, 4-6 DMA
        // pieces of ~80 cycles each) leaves the matrix pipe idle for its whole issue time (tools/flash_timeline_dkv.py: 1 260 cycles for
        // this phase against 512 of MFMA work when reads and DMA came first / in the middle).
        // S^T and dP^T accumulate in VGPRs: with the builtin hipcc puts them in a[0:31], which hold a quarter of the dK / dV accumulators,
        // and moves those 32 registers out to VGPRs and back around them on every tile (96 v_accvgpr_* per iteration).  Hazards the
        // assembler does not pad: 18+ wait states after the last MFMA before VALU reads (the operands come from LDS reads the compiler waits
        // for and from long-lived registers: no VALU-write -> MFMA-read hazard here).
        bf16x8_t tD[2][4], tQ[2][4];
        f32x16_t s, dp;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(qr[0]), "v"(kf[0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(dp) : "v"(dr[0]), "v"(vf[0]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(qr[c]), "v"(kf[c]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dp) : "v"(dr[c]), "v"(vf[c]));
            }
            {   // fragments 2c, 2c+1 of the list  tD[0][0..3], tQ[0][0..3], tD[1][0..3], tQ[1][0..3]
                const int cc = c >> 2, db0 = (c & 1) * 2;
                if (((c >> 1) & 1) == 0) {
                    tD[cc][db0] = tr_pi_frag(Dc, to1, to2, db0, 16 * cc);
                    tD[cc][db0 + 1] = tr_pi_frag(Dc, to1, to2, db0 + 1, 16 * cc);
                } else {
                    tQ[cc][db0] = tr_pi_frag(Qc, to1, to2, db0, 16 * cc);
                    tQ[cc][db0 + 1] = tr_pi_frag(Qc, to1, to2, db0 + 1, 16 * cc);
                }
            }
            if (c < 4) issue_piece(qt2, nb, c);
            else if (c == 4 && wave == 0) { issue_piece(qt2, nb, 4); issue_piece(qt2, nb, 5); }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s), "+v"(dp));
        DSTAMP(2 + 5 * (qt - qt0));
        // ---- phase 2: softmax of rows 0-7 (element r = 4g + e is query row i0 + 8g + 4h2 + e), then rows 8-15 one element per dV / dK
        // MFMA of the first half (c = 0), then the second half's MFMAs with the row-fragment reads of tile qt+1 between them.
        // Visibility of a row for this lane's key: lo <= idx < hi, one unsigned compare + v_cndmask (the short-circuit form compiled into
        // 29 EXEC-mask branches per iteration); interior tiles without a key-padding mask skip the mask arithmetic.
        const bool interior = i0 + 31 < a.Sq && (!a.causal || kw + 31 <= i0 + off);  // wave-uniform
        int lo = 0, hi = 32;
        if (!interior) {
            const int vis = a.causal ? kj - off - i0 : 0;     // first visible row, relative to the tile
            lo = vis > 0 ? vis : 0;
            hi = a.Sq - i0 < 32 ? a.Sq - i0 : 32;
        }
        if (!kok) hi = 0;
        const unsigned span = hi > lo ? (unsigned)(hi - lo) : 0u;
        const int idx0 = 4 * h2 - lo;
        // (macro, not a lambda: s[r] / dp[r] must stay register indices)
#define DKV_ELEM(R_, FAST_, LV_, DV_)                                                                  \
    do {                                                                                               \
        const float pe_ = __builtin_amdgcn_exp2f(fmaf(s[R_], sc2, bias2 - (LV_)));                      \
        const bool ok_ = (FAST_) || (unsigned)(idx0 + 8 * ((R_) >> 2) + ((R_) & 3)) < span;            \
        const float p_ = ok_ ? pe_ : 0.f;                                                              \
        s[R_] = p_;                                                                                    \
        dp[R_] = p_ * (dp[R_] - (DV_));                                                                \
    } while (0)
        {
            float lv[8], dvv[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 8 * g + 4 * h2);
                const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 8 * g + 4 * h2);
                lv[4 * g] = l4.x; lv[4 * g + 1] = l4.y; lv[4 * g + 2] = l4.z; lv[4 * g + 3] = l4.w;
                dvv[4 * g] = d4.x; dvv[4 * g + 1] = d4.y; dvv[4 * g + 2] = d4.z; dvv[4 * g + 3] = d4.w;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) DKV_ELEM(r, false, lv[r], dvv[r]);
        }
        const bf16x8_t pf0 = pack8(s, 0), dsf0 = pack8(dp, 0);
        __builtin_amdgcn_sched_barrier(0);
        DSTAMP(3 + 5 * (qt - qt0));
        {
            float lv[8], dvv[8];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 8 * (g + 2) + 4 * h2);
                const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 8 * (g + 2) + 4 * h2);
                lv[4 * g] = l4.x; lv[4 * g + 1] = l4.y; lv[4 * g + 2] = l4.z; lv[4 * g + 3] = l4.w;
                dvv[4 * g] = d4.x; dvv[4 * g + 1] = d4.y; dvv[4 * g + 2] = d4.z; dvv[4 * g + 3] = d4.w;
            }
            // (one code path: a wave-uniform `interior` branch around these MFMAs made hipcc carry the 128 dK / dV accumulators through
            //  both arms -- 128 AGPR copies + 168 B of scratch per lane, whose reloads are VMEM loads that drain the DMA; an interior tile
            //  simply has lo = 0, hi = 32)
            // Rows 8-15 in two groups of four, each stage of a group (argument, exp2, mask, dS) behind one dV / dK MFMA.  Everything is pinned:
            // the MFMAs are asm (the builtin is pure to the IR optimiser, which gathered all eight in front of the softmax whatever the
            // source order and the sched_barriers said), and an empty asm re-defines a stage's four results before the next MFMA, so a
            // stage can neither rise above the previous MFMA nor sink below the next.  Four independent elements per stage: one element
            // per MFMA (first attempt) serialised each element's dependent chain -- 125 cycles per pair instead of ~60.
#define DKV_MFMA0(I_)                                                              \
    do {                                                                           \
        if ((I_) & 1) MFMA_ACC(dk[(I_) >> 1], tQ[0][(I_) >> 1], dsf0);             \
        else MFMA_ACC(dv[(I_) >> 1], tD[0][(I_) >> 1], pf0);                       \
    } while (0)
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                const int r0 = 8 + 4 * grp;
                float t0, t1, t2, t3;
                asm volatile("" : "+v"(s), "+v"(dp));
                DKV_MFMA0(4 * grp + 0);
                t0 = fmaf(s[r0 + 0], sc2, bias2 - lv[4 * grp + 0]);
                t1 = fmaf(s[r0 + 1], sc2, bias2 - lv[4 * grp + 1]);
                t2 = fmaf(s[r0 + 2], sc2, bias2 - lv[4 * grp + 2]);
                t3 = fmaf(s[r0 + 3], sc2, bias2 - lv[4 * grp + 3]);
                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                DKV_MFMA0(4 * grp + 1);
                t0 = __builtin_amdgcn_exp2f(t0); t1 = __builtin_amdgcn_exp2f(t1); t2 = __builtin_amdgcn_exp2f(t2); t3 = __builtin_amdgcn_exp2f(t3);
                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                DKV_MFMA0(4 * grp + 2);
                t0 = (unsigned)(idx0 + 8 * ((r0 + 0) >> 2) + ((r0 + 0) & 3)) < span ? t0 : 0.f;
                t1 = (unsigned)(idx0 + 8 * ((r0 + 1) >> 2) + ((r0 + 1) & 3)) < span ? t1 : 0.f;
                t2 = (unsigned)(idx0 + 8 * ((r0 + 2) >> 2) + ((r0 + 2) & 3)) < span ? t2 : 0.f;
                t3 = (unsigned)(idx0 + 8 * ((r0 + 3) >> 2) + ((r0 + 3) & 3)) < span ? t3 : 0.f;
                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                DKV_MFMA0(4 * grp + 3);
                s[r0 + 0] = t0; s[r0 + 1] = t1; s[r0 + 2] = t2; s[r0 + 3] = t3;
                dp[r0 + 0] = t0 * (dp[r0 + 0] - dvv[4 * grp + 0]);
                dp[r0 + 1] = t1 * (dp[r0 + 1] - dvv[4 * grp + 1]);
                dp[r0 + 2] = t2 * (dp[r0 + 2] - dvv[4 * grp + 2]);
                dp[r0 + 3] = t3 * (dp[r0 + 3] - dvv[4 * grp + 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef DKV_MFMA0
        }
#undef DKV_ELEM
        const bf16x8_t pf1 = pack8(s, 8), dsf1 = pack8(dp, 8);
        DSTAMP(4 + 5 * (qt - qt0));
        // tile qt+1 has landed when at most the pieces of tile qt+2 are outstanding (wave 0 also carries the two statistics pieces);
        // lgkmcnt(0): this wave's LDS reads of tile qt (transpose reads, lse / delta) have returned, so after the barrier its slot is dead
        // (PERS, first tile of a later key block: tile qt+1 was waited for -- vmcnt(0) -- at the switch, BEFORE the dK / dV stores and the
        //  staging DMA were issued; a counted wait here would have to count those)
        if (PERS && first) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); first = false; }
        else if (wave == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // ---- phase 3: dV / dK MFMAs of the second half, the 16 row-fragment reads of tile qt+1 (for the NEXT iteration's phase 1) two per MFMA
        const int nx = cur == 2 ? 0 : cur + 1;
        const char* Qn = smem + nx * 8192;
        const char* Dn = smem + 24576 + nx * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i & 1) MFMA_ACC_MEM(dk[i >> 1], tQ[1][i >> 1], dsf1);
            else MFMA_ACC_MEM(dv[i >> 1], tD[1][i >> 1], pf1);
            qr[i] = *reinterpret_cast<const bf16x8_t*>(Qn + (qfo ^ (32 * i)));
            dr[i] = *reinterpret_cast<const bf16x8_t*>(Dn + (qfo ^ (32 * i)));
            __builtin_amdgcn_sched_barrier(0);
        }
        DSTAMP(5 + 5 * (qt - qt0));
    }
    // the accumulators were last written by asm MFMAs hipcc knows nothing about: 18+ wait states before it reads them out of the AGPRs
    asm volatile("s_nop 15\n\ts_nop 3" : "+a"(dk[0]), "+a"(dk[1]), "+a"(dk[2]), "+a"(dk[3]), "+a"(dv[0]), "+a"(dv[1]), "+a"(dv[2]), "+a"(dv[3]));
    if constexpr (!PERS) break;
    else {
        // ---- switch to the next key block of this head
        const bool last = rank + 1 >= nkb;
        DSTAMP(80 + 3 * (rank & 3));
        // everything in flight lands first (the next tile's pieces were issued most of an iteration ago): the next iteration then needs no
        // counted wait, whatever number of stores / staging pieces is issued below
        if (!last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        char* const trn = smem + DKV_TRN + wave * 8192;
        const int64_t dkpo = PAIR ? a.dks.h - 64 : 0, dvpo = PAIR ? a.dvs.h - 64 : 0;
        store_rows_lds<PAIR>(trn, a.dk + b * a.dks.b + hd * hm * a.dks.h, a.dks.s, kw, a.Sk, dk, a.scale, a.scale, lane, dkpo);
        store_rows_lds<PAIR>(trn, a.dv + b * a.dvs.b + hd * hm * a.dvs.h, a.dvs.s, kw, a.Sk, dv, 1.0f, 1.0f, lane, dvpo);
        DSTAMP(81 + 3 * (rank & 3));
        if (last) break;
        ++rank;
        k0 += 128; kw += 128; kj += 128;
        kok = kj < a.Sk && ((kvbits >> rank) & 1u) != 0u;
        bias2 = __fmul_rn(sl2, (float)(kj - (a.Sk - 1)));
        bias2B = __fmul_rn(sl2B, (float)(kj - (a.Sk - 1)));
        qt0 = qtn;
        qtn = rank + 1 < nkb ? first_tile_of(k0 + 128) : nqt + 2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            kf[c] = *reinterpret_cast<const bf16x8_t*>(stK + (qfo ^ (32 * c)));
            vf[c] = *reinterpret_cast<const bf16x8_t*>(stV + (qfo ^ (32 * c)));
        }
#pragma unroll
        for (int db = 0; db < 4; ++db) { dk[db] = zero16(); dv[db] = zero16(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging rows are in registers before the next block's rows may land on them
        if (rank + 1 < nkb) stage_kv(kw + 128);
        first = true;
        DSTAMP(82 + 3 * ((rank - 1) & 3));
    }
    }
#undef MFMA_ACC
#undef MFMA_ACC_MEM
    if constexpr (PERS) {
        BLK_END(2);
#ifdef OTTER_FLASH_TIMING
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < 96 && g_flash_stamps) g_flash_stamps[threadIdx.x] = reinterpret_cast<unsigned long long*>(smem + 50688)[threadIdx.x];
#endif
        return;
    }
    DSTAMP(90);
#if OTTER_FLASH_ROWSTORE
    {   // the Q / dO ring is dead once every wave has left the loop: 8 KB of it per wave stage the row-major dK / dV tiles.
        // The loop issues the LDS-DMA pieces of tiles qt+2 unconditionally (out-of-range tiles read zeros) and its last counted wait
        // leaves up to 4-6 of them in flight: every wave drains its OWN pieces before the barrier, so that no late DMA write of
        // another wave can land on the staging rows between store_rows_lds's ds_write and ds_read.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        char* const trn = smem + wave * 8192;
        const int64_t dkpo = PAIR ? a.dks.h - 64 : 0, dvpo = PAIR ? a.dvs.h - 64 : 0;
        store_rows_lds<PAIR>(trn, a.dk + b * a.dks.b + hd * hm * a.dks.h, a.dks.s, kw, a.Sk, dk, a.scale, a.scale, lane, dkpo);
        store_rows_lds<PAIR>(trn, a.dv + b * a.dvs.b + hd * hm * a.dvs.h, a.dvs.s, kw, a.Sk, dv, 1.0f, 1.0f, lane, dvpo);
    }
    if constexpr (PAIR) return;
#else
    if constexpr (PAIR) {
        if (kj < a.Sk) {
            store_dt_pair(a.dk + b * a.dks.b + hd * 2 * a.dks.h + (int64_t)kj * a.dks.s, dk, a.scale, a.scale, h2, a.dks.h - 64);
            store_dt_pair(a.dv + b * a.dvs.b + hd * 2 * a.dvs.h + (int64_t)kj * a.dvs.s, dv, 1.0f, 1.0f, h2, a.dvs.h - 64);
        }
        return;
    }
    if (kj < a.Sk) {
        store_dt(a.dk + b * a.dks.b + hd * a.dks.h + (int64_t)kj * a.dks.s, dk, a.scale, h2);
        store_dt(a.dv + b * a.dvs.b + hd * a.dvs.h + (int64_t)kj * a.dvs.s, dv, 1.0f, h2);
    }
#endif
    BLK_END(2);
#ifdef OTTER_FLASH_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DSTAMP(91);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 96 && g_flash_stamps) g_flash_stamps[threadIdx.x] = reinterpret_cast<unsigned long long*>(smem + 50688)[threadIdx.x];
#endif
}

int fill_args(const otter_flash_desc* d, FlashArgs& a, bool bwd) {
    OTTER_REQUIRE(d && d->q && d->k && d->v && d->o && d->lse, "flash: null pointer");
    OTTER_REQUIRE(d->head_dim == HD || d->head_dim == 64, "flash: head_dim %d (128, or 64 with an even number of heads)", d->head_dim);
    OTTER_REQUIRE(d->B > 0 && d->H > 0 && d->Sq > 0 && d->Sk > 0, "flash: empty shape");
    const bool pair = d->head_dim == 64;
    OTTER_REQUIRE(!pair || d->H % 2 == 0, "flash: head_dim 64 runs two heads per workgroup and needs an even head count (got %d)", d->H);
    const otter_flash_view* vs[] = {&d->qv, &d->kv, &d->vv, &d->ov, &d->dov, &d->dqv, &d->dkv, &d->dvv};
    for (int i = 0; i < (bwd ? 8 : 4); ++i)
        OTTER_REQUIRE(vs[i]->batch_stride % 8 == 0 && vs[i]->seq_stride % 8 == 0 && vs[i]->head_stride % 8 == 0,
                      "flash: strides must be multiples of 8 elements");
    OTTER_REQUIRE((((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)d->o) & 15) == 0, "flash: 16-byte alignment");
    memset(&a, 0, sizeof(a));
    auto st = [](const otter_flash_view& v) { return Str{v.batch_stride, v.seq_stride, v.head_stride}; };
    a.q = (const bf16_t*)d->q; a.k = (const bf16_t*)d->k; a.v = (const bf16_t*)d->v; a.o = (bf16_t*)d->o;
    a.qs = st(d->qv); a.ks = st(d->kv); a.vs = st(d->vv); a.os = st(d->ov);
    a.lse = d->lse; a.slopes = d->alibi_slopes; a.kvalid = d->key_valid;
    a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Sk = d->Sk; a.causal = d->causal; a.scale = d->scale;
    a.pair = pair ? 1 : 0;
    if (pair) {   // the second head of a pair must lie inside the first head's token row, after its 64 columns
        for (int i = 0; i < (bwd ? 8 : 4); ++i)
            OTTER_REQUIRE(vs[i]->head_stride >= 64 && vs[i]->head_stride + 64 <= vs[i]->seq_stride, "flash: head_dim 64 wants 64 <= head_stride <= seq_stride - 64");
    }
    {   // heads per LPT group: the K + V (= Q + dO) panels of a group, Sk x 128 x 2 B x 2 per head (or head pair), within 64 MB; a divisor
        // of the block count and a multiple of 8 when there is one (blocks are dealt round-robin to the 8 XCDs: same head -> same XCD)
        const int64_t nbh = (int64_t)d->B * (pair ? d->H / 2 : d->H), per_head = (int64_t)(d->Sk > d->Sq ? d->Sk : d->Sq) * 512;
        int64_t g = nbh;
        const int64_t cap = (int64_t(64) << 20) / (per_head > 0 ? per_head : 1);
        if (g > cap) {
            g = 0;
            for (int64_t c = cap - cap % 8; c >= 8; c -= 8)
                if (nbh % c == 0) { g = c; break; }
            if (g == 0) g = nbh;
        }
        // (experiment hook: OTTER_FLASH_LPT_GROUP=n forces the group size when it divides the block count)
        static const int forced = [] { const char* e = getenv("OTTER_FLASH_LPT_GROUP"); return e ? atoi(e) : 0; }();
        if (forced > 0 && nbh % forced == 0) g = forced;
        a.lpt_group = (int)g;
    }
    if (bwd) {
        OTTER_REQUIRE(d->dout && d->delta && d->dq && d->dk && d->dv, "flash bwd: null pointer");
        OTTER_REQUIRE((((uintptr_t)d->dout | (uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & 15) == 0, "flash bwd: 16-byte alignment");
        a.dout = (const bf16_t*)d->dout; a.dos = st(d->dov); a.delta = d->delta;
        a.dq = (bf16_t*)d->dq; a.dk = (bf16_t*)d->dk; a.dv = (bf16_t*)d->dv;
        a.dqs = st(d->dqv); a.dks = st(d->dkv); a.dvs = st(d->dvv);
    }
    return OTTER_OK;
}

int g_flash_variant = 0;
// longest-processing-time-first block order (default for causal launches; variants 2 / 3 keep the plain 3-D grid for A/B runs):
// at C2 (B=8, 32 heads, S=512) the forward went 54.3 -> 43.7 us and the backward 230 -> 184.5 us per layer with it
inline bool flash_lpt(const FlashArgs& a) { return a.causal && g_flash_variant != 2 && g_flash_variant != 3; }

// One dK/dV workgroup per (batch, head or head pair) that walks all key blocks (flash_bwd_dkv2_kernel<..., PERS>): when those workgroups fill
// the chip evenly (a multiple of the CU count, or many rounds of them), every key block has at least two query tiles (the ring runs two
// tiles ahead across the block switch) and the key-padding bits of all blocks fit one word.  Flash variant 7 switches it off (A/B).
inline int device_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return n;
}
inline bool dkv_persistent(const FlashArgs& a) {
    if (g_flash_variant == 7 || g_flash_variant == 2 || g_flash_variant == 3 || g_flash_variant == 5) return false;
    const int nkb = (a.Sk + 127) / 128, nqt = (a.Sq + 31) / 32;
    if (nkb < 2 || nkb > 32 || a.Sq != a.Sk) return false;   // (the backward is a training launch: square; nothing else is tested in this form)
    const int imin = (nkb - 1) * 128 - (a.Sk - a.Sq);
    const int qt_last = a.causal && imin > 0 ? imin >> 5 : 0;
    if (nqt - qt_last < 2) return false;
    if (g_flash_variant == 8) return true;   // tests: the persistent form on small launches too
    const int nwg = a.B * (a.pair ? a.H / 2 : a.H), cus = device_cu_count();
    return nwg % cus == 0 || nwg >= 8 * cus;
}

template <typename K>
int set_smem(K kern, int bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) OTTER_FAIL(OTTER_ERR_LAUNCH, "flash: hipFuncSetAttribute(%d): %s", bytes, hipGetErrorString(e));
    return OTTER_OK;
}

}  // namespace

extern "C" {

int otter_flash_set_variant(int v) {
    OTTER_REQUIRE(v >= 0 && v <= 8, "flash variant %d (8 = 0 with the persistent dK/dV form on every launch it can address; 7 = 0 with one dK/dV workgroup per key block instead of the persistent per-head form; 6 = forward version 3: 16 queries per wave, 16x16x32 MFMA, four waves per SIMD; 0 = default: LDS-DMA v2, LPT block order, delta inside the dQ kernel; 1 = register-staged v1; 2 = v2, plain grid; 3 = 2 + dK/dV at two workgroups per CU; 4 = 0 with the separate flash_delta launch (round-3 order); 5 = 0 + dK/dV at two workgroups per CU)", v);
    g_flash_variant = v;
    return OTTER_OK;
}

int otter_flash_attn_fwd(const otter_flash_desc* d, void* stream) {
    FlashArgs a;
    int rc = fill_args(d, a, false);
    if (rc) return rc;
    // the DMA path addresses a head's K / V with 32-bit offsets from its base
    const bool v2 = g_flash_variant != 1 && (int64_t)a.Sk * a.ks.s * 2 < (int64_t(1) << 31) && (int64_t)a.Sk * a.vs.s * 2 < (int64_t(1) << 31);
    OTTER_REQUIRE(v2 || !a.pair, "flash: head_dim 64 only on the LDS-DMA kernels (variant != 1, K / V panels under 2 GB)");
    if (v2 && a.pair) {
        const int smem = 65536 + KMASK_TILES * 8;
        static bool once = false;
        if (!once) {
            rc = set_smem(flash_fwd2_kernel<false, true, true>, smem); if (rc) return rc;
            rc = set_smem(flash_fwd2_kernel<true, true, true>, smem); if (rc) return rc;
            rc = set_smem(flash_fwd2_kernel<false, true, false>, smem); if (rc) return rc;
            rc = set_smem(flash_fwd2_kernel<true, true, false>, smem); if (rc) return rc;
            once = true;
        }
        const unsigned nqb = (unsigned)((a.Sq + 127) / 128), hb = (unsigned)(a.H / 2);
        const dim3 g1(nqb * hb * a.B), g3(nqb, hb, a.B);
        if (a.slopes) {
            if (flash_lpt(a)) hipLaunchKernelGGL((flash_fwd2_kernel<true, true, true>), g1, dim3(256), smem, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((flash_fwd2_kernel<false, true, true>), g3, dim3(256), smem, (hipStream_t)stream, a);
        } else {
            if (flash_lpt(a)) hipLaunchKernelGGL((flash_fwd2_kernel<true, true, false>), g1, dim3(256), smem, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((flash_fwd2_kernel<false, true, false>), g3, dim3(256), smem, (hipStream_t)stream, a);
        }
    } else if (v2 && g_flash_variant == 6) {
        const int smem = 65536 + KMASK_TILES * 8;
        static bool once = false;
        if (!once) {
            rc = set_smem(flash_fwd3_kernel<false>, smem); if (rc) return rc;
            rc = set_smem(flash_fwd3_kernel<true>, smem); if (rc) return rc;
            once = true;
        }
        const unsigned nqb = (unsigned)((a.Sq + 127) / 128);
        if (flash_lpt(a)) hipLaunchKernelGGL((flash_fwd3_kernel<true>), dim3(nqb * a.H * a.B), dim3(512), smem, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((flash_fwd3_kernel<false>), dim3(nqb, a.H, a.B), dim3(512), smem, (hipStream_t)stream, a);
    } else if (v2) {
#ifdef OTTER_FLASH_TIMING
        const int smem = 65536 + KMASK_TILES * 8 + 1024;
#else
        const int smem = 65536 + KMASK_TILES * 8;
#endif
        static bool once = false;
        if (!once) {
            rc = set_smem(flash_fwd2_kernel<false, false, true>, smem); if (rc) return rc;
            rc = set_smem(flash_fwd2_kernel<true, false, true>, smem); if (rc) return rc;
            once = true;
        }
        const unsigned nqb = (unsigned)((a.Sq + 127) / 128);
        if (flash_lpt(a)) hipLaunchKernelGGL((flash_fwd2_kernel<true, false, true>), dim3(nqb * a.H * a.B), dim3(256), smem, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((flash_fwd2_kernel<false, false, true>), dim3(nqb, a.H, a.B), dim3(256), smem, (hipStream_t)stream, a);
    } else {
        const int smem = 64 * LDK * 2 + 64 * LDT * 2;
        static bool once = false;
        if (!once) { rc = set_smem(flash_fwd_kernel, smem); if (rc) return rc; once = true; }
        hipLaunchKernelGGL(flash_fwd_kernel, dim3((a.Sq + 127) / 128, a.H, a.B), dim3(256), smem, (hipStream_t)stream, a);
    }
    OTTER_CHECK_LAUNCH("flash_fwd");
    return OTTER_OK;
}

int otter_flash_attn_bwd(const otter_flash_desc* d, void* stream) {
    FlashArgs a;
    int rc = fill_args(d, a, true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nrows = (int64_t)a.B * a.H * a.Sq;
    const int64_t lim = int64_t(1) << 31;
    const bool v2 = g_flash_variant != 1 && (int64_t)a.Sk * a.ks.s * 2 < lim && (int64_t)a.Sk * a.vs.s * 2 < lim &&
                    (int64_t)a.Sq * a.qs.s * 2 < lim && (int64_t)a.Sq * a.dos.s * 2 < lim;
    // round 4: on the LDS-DMA kernels the dQ kernel computes delta / the log2 LSE itself and runs FIRST (dK/dV reads what it published);
    // variant 4 keeps the separate flash_delta launch and the old order (A/B)
    a.fuse_delta = (v2 && g_flash_variant != 4) ? 1 : 0;
    if (!a.fuse_delta) {
        if (a.pair) hipLaunchKernelGGL(flash_delta_kernel<8>, dim3((unsigned)((nrows + 31) / 32)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(flash_delta_kernel<16>, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, st, a);
        OTTER_CHECK_LAUNCH("flash_delta");
    }
    OTTER_REQUIRE(v2 || !a.pair, "flash: head_dim 64 only on the LDS-DMA kernels (variant != 1, panels under 2 GB)");
    if (v2 && a.pair) {
        const int smem_kv = 49152 + 1536, smem_q = 65536 + KMASK_TILES * 8;
        static bool once = false;
        if (!once) {
            rc = set_smem(flash_bwd_dkv2_kernel<1, false, true>, smem_kv); if (rc) return rc;
            rc = set_smem(flash_bwd_dkv2_kernel<1, true, true>, smem_kv); if (rc) return rc;
            rc = set_smem((flash_bwd_dkv2_kernel<1, true, true, true>), DKV_PERS_SMEM); if (rc) return rc;
            rc = set_smem(flash_bwd_dq2_kernel<false, true, true>, smem_q); if (rc) return rc;
            rc = set_smem(flash_bwd_dq2_kernel<true, true, true>, smem_q); if (rc) return rc;
            rc = set_smem(flash_bwd_dq2_kernel<false, true, false>, smem_q); if (rc) return rc;
            rc = set_smem(flash_bwd_dq2_kernel<true, true, false>, smem_q); if (rc) return rc;
            once = true;
        }
        const unsigned nkb = (unsigned)((a.Sk + 127) / 128), nqb = (unsigned)((a.Sq + 127) / 128), hb = (unsigned)(a.H / 2);
        auto launch_dkv = [&]() {
            if (dkv_persistent(a)) { hipLaunchKernelGGL((flash_bwd_dkv2_kernel<1, true, true, true>), dim3(hb * a.B), dim3(256), DKV_PERS_SMEM, st, a); return; }
            if (flash_lpt(a)) hipLaunchKernelGGL((flash_bwd_dkv2_kernel<1, true, true>), dim3(nkb * hb * a.B), dim3(256), smem_kv, st, a);
            else hipLaunchKernelGGL((flash_bwd_dkv2_kernel<1, false, true>), dim3(nkb, hb, a.B), dim3(256), smem_kv, st, a);
        };
        auto launch_dq = [&]() {
            if (a.slopes) {
                if (flash_lpt(a)) hipLaunchKernelGGL((flash_bwd_dq2_kernel<true, true, true>), dim3(nqb * hb * a.B), dim3(256), smem_q, st, a);
                else hipLaunchKernelGGL((flash_bwd_dq2_kernel<false, true, true>), dim3(nqb, hb, a.B), dim3(256), smem_q, st, a);
            } else {
                if (flash_lpt(a)) hipLaunchKernelGGL((flash_bwd_dq2_kernel<true, true, false>), dim3(nqb * hb * a.B), dim3(256), smem_q, st, a);
                else hipLaunchKernelGGL((flash_bwd_dq2_kernel<false, true, false>), dim3(nqb, hb, a.B), dim3(256), smem_q, st, a);
            }
        };
        if (a.fuse_delta) { launch_dq(); OTTER_CHECK_LAUNCH("flash_bwd_dq"); launch_dkv(); OTTER_CHECK_LAUNCH("flash_bwd_dkv"); }
        else { launch_dkv(); OTTER_CHECK_LAUNCH("flash_bwd_dkv"); launch_dq(); OTTER_CHECK_LAUNCH("flash_bwd_dq"); }
        return OTTER_OK;
    }
    if (v2) {
        #ifdef OTTER_FLASH_TIMING
        const int smem_kv = 49152 + 1536 + 1024, smem_q = 65536 + KMASK_TILES * 8;
#else
        const int smem_kv = 49152 + 1536, smem_q = 65536 + KMASK_TILES * 8;
#endif
        static bool once = false;
        if (!once) {
            rc = set_smem(flash_bwd_dkv2_kernel<1, false, false>, smem_kv); if (rc) return rc;
            rc = set_smem(flash_bwd_dkv2_kernel<2, false, false>, smem_kv); if (rc) return rc;
            rc = set_smem(flash_bwd_dkv2_kernel<1, true, false>, smem_kv); if (rc) return rc;
            rc = set_smem(flash_bwd_dkv2_kernel<2, true, false>, smem_kv); if (rc) return rc;
            rc = set_smem((flash_bwd_dkv2_kernel<1, true, false, true>), DKV_PERS_SMEM); if (rc) return rc;
            rc = set_smem((flash_bwd_dq2_kernel<false, false, true>), smem_q); if (rc) return rc;
            rc = set_smem((flash_bwd_dq2_kernel<true, false, true>), smem_q); if (rc) return rc;
            once = true;
        }
        const unsigned nkb = (unsigned)((a.Sk + 127) / 128);
        const unsigned nqb = (unsigned)((a.Sq + 127) / 128);
        auto launch_dkv = [&]() {
            if (dkv_persistent(a)) { hipLaunchKernelGGL((flash_bwd_dkv2_kernel<1, true, false, true>), dim3(a.H * a.B), dim3(256), DKV_PERS_SMEM, st, a); return; }
            if (g_flash_variant == 3) hipLaunchKernelGGL((flash_bwd_dkv2_kernel<2, false, false>), dim3(nkb, a.H, a.B), dim3(256), smem_kv, st, a);
            else if (flash_lpt(a) && g_flash_variant != 5) hipLaunchKernelGGL((flash_bwd_dkv2_kernel<1, true, false>), dim3(nkb * a.H * a.B), dim3(256), smem_kv, st, a);
            else if (g_flash_variant == 5 && a.causal) hipLaunchKernelGGL((flash_bwd_dkv2_kernel<2, true, false>), dim3(nkb * a.H * a.B), dim3(256), smem_kv, st, a);
            else hipLaunchKernelGGL((flash_bwd_dkv2_kernel<1, false, false>), dim3(nkb, a.H, a.B), dim3(256), smem_kv, st, a);
        };
        auto launch_dq = [&]() {
            if (flash_lpt(a)) hipLaunchKernelGGL((flash_bwd_dq2_kernel<true, false, true>), dim3(nqb * a.H * a.B), dim3(256), smem_q, st, a);
            else hipLaunchKernelGGL((flash_bwd_dq2_kernel<false, false, true>), dim3(nqb, a.H, a.B), dim3(256), smem_q, st, a);
        };
        if (a.fuse_delta) { launch_dq(); OTTER_CHECK_LAUNCH("flash_bwd_dq"); launch_dkv(); OTTER_CHECK_LAUNCH("flash_bwd_dkv"); }
        else { launch_dkv(); OTTER_CHECK_LAUNCH("flash_bwd_dkv"); launch_dq(); OTTER_CHECK_LAUNCH("flash_bwd_dq"); }
        return OTTER_OK;
    }
    const int smem_kv = (2 * 32 * LDK + 2 * 32 * LDT) * 2 + 64 * 4;
    const int smem_q = (2 * 64 * LDK + 64 * LDT) * 2;
    static bool once = false;
    if (!once) {
        rc = set_smem(flash_bwd_dkv_kernel, smem_kv); if (rc) return rc;
        rc = set_smem(flash_bwd_dq_kernel, smem_q); if (rc) return rc;
        once = true;
    }
    hipLaunchKernelGGL(flash_bwd_dkv_kernel, dim3((a.Sk + 127) / 128, a.H, a.B), dim3(256), smem_kv, st, a);
    OTTER_CHECK_LAUNCH("flash_bwd_dkv");
    hipLaunchKernelGGL(flash_bwd_dq_kernel, dim3((a.Sq + 127) / 128, a.H, a.B), dim3(256), smem_q, st, a);
    OTTER_CHECK_LAUNCH("flash_bwd_dq");
    return OTTER_OK;
}

#ifdef OTTER_FLASH_TIMING
int otter_flash_set_stamps(void* dev_ptr) {  // diagnostics builds only; not part of include/otter_hip.h
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_flash_stamps), &dev_ptr, sizeof(void*));
    return e == hipSuccess ? 0 : -1;
}
int otter_flash_set_block_stamps(void* dev_ptr) {  // [3][4096][4] u64: begin, end (10 ns ticks), HW_ID, XCC_ID per workgroup
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_flash_blk), &dev_ptr, sizeof(void*));
    return e == hipSuccess ? 0 : -1;
}
#endif

}  // extern "C"
