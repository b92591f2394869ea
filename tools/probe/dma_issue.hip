// Round 6d probe: what an LDS-DMA piece (s_mov m0 + buffer_load_dwordx4 ... lds, 1 KB per wave) costs a wave that otherwise issues back-to-back
// v_mfma_f32_16x16x32_bf16 -- with ONE such wave per SIMD (the layout of gemm variant 26: 128 MFMAs + 16 pieces per K-tile and wave) and with TWO
// waves per SIMD that split the same work (64 MFMAs + 8 pieces each, the second wave's pieces half a period later).  Same MFMAs and the same
// 64 KB of DMA per CU and iteration in both modes; source = a 64 KB L2-resident buffer; nobody waits for the pieces beyond a bounded queue.
// Build: hipcc --offload-arch=gfx950 -O3 dma_issue.hip -o dma_issue.bin ; run: ./dma_issue.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void dma16(u32x4_t r, unsigned lds, unsigned voff, unsigned soff) {
    if constexpr (MODE == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
    else if constexpr (MODE == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %2, %3 lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
    else if constexpr (MODE == 2) asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, 1\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b64 exec, -1" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
    else if constexpr (MODE == 3) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
}
template <int Q>
__device__ __forceinline__ void dma16_imm_nom0(u32x4_t r, unsigned voff, unsigned soff) {
    if constexpr (Q == 0) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(r), "s"(soff) : "memory");
    else if constexpr (Q == 1) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:1024 lds" : : "v"(voff), "s"(r), "s"(soff) : "memory");
    else if constexpr (Q == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:2048 lds" : : "v"(voff), "s"(r), "s"(soff) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:3072 lds" : : "v"(voff), "s"(r), "s"(soff) : "memory");
}
template <int Q>
__device__ __forceinline__ void dma16_imm(u32x4_t r, unsigned lds, unsigned voff, unsigned soff) {
    if constexpr (Q == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
    else if constexpr (Q == 1) asm volatile("buffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
    else if constexpr (Q == 2) asm volatile("buffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
    else asm volatile("buffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds" : : "s"(lds), "v"(voff), "s"(r), "s"(soff) : "memory");
}

// NB = accumulator blocks per wave (64: one wave per SIMD, 32: two), PIECES = DMA pieces per iteration and wave, PHASE = slot offset of the pieces
template <int NB, int PIECES, bool DMA, int MODE = 0, int FILL = 0>
__global__ __launch_bounds__(NB == 64 ? 256 : 512) void k(const uint4* __restrict__ src, const void* __restrict__ dsrc, unsigned long long* __restrict__ out, int iters) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int NA = NB == 64 ? 8 : 4;
    bf16x8_t a[NA], b[8];
#pragma unroll
    for (int i = 0; i < NA; ++i) { uint4 x = src[((tid + blockIdx.x * 512) * 16 + i) & 0xffff]; a[i] = *reinterpret_cast<bf16x8_t*>(&x); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint4 y = src[((tid + blockIdx.x * 512) * 16 + 8 + i) & 0xffff]; b[i] = *reinterpret_cast<bf16x8_t*>(&y); }
    f32x4_t acc[NA][8];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t rs;
    {
        const uint64_t p = (uint64_t)dsrc;
        rs[0] = (unsigned)p; rs[1] = (unsigned)(p >> 32) & 0xffffu; rs[2] = 65536u; rs[3] = 0x00020000u;
        rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
        rs[2] = __builtin_amdgcn_readfirstlane(rs[2]); rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned voff = (unsigned)lane * 16u;
    unsigned sfill = 0; uint4 lfill = {0, 0, 0, 0};
    const int phase = (NB == 32 && wave >= 4) ? (64 / PIECES) / 2 : 0;
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int p = 0;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int ks = 0; ks < (NB == 64 ? 2 : 2); ++ks) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int slot = (i * 8 + j) * 2 + ks;                 // 0 .. 2 * NA * 8 - 1
                    if constexpr (FILL != 0) {
                        if ((slot & 3) == 1) {
                            if constexpr (FILL == 1) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
                            else if constexpr (FILL == 2) asm volatile("s_nop 0" ::: "memory");
                            else if constexpr (FILL == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sfill));
                            else if constexpr (FILL == 4) { asm volatile("ds_read_b128 %0, %1" : "=v"(lfill) : "v"(voff) : "memory"); }
                            else if constexpr (FILL == 5) { asm volatile("s_waitcnt lgkmcnt(15)\n\ts_waitcnt lgkmcnt(15)\n\ts_waitcnt lgkmcnt(15)" ::: "memory"); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    constexpr int NS = NA * 16;
                    if constexpr (MODE == 6) {
                        if (DMA && ((slot + 1 + NS - phase) % (NS / PIECES)) == 0) {     // one slot before a piece
                            const int q = (((slot + 1) % NS) / (NS / PIECES)) % PIECES;
                            if ((q & 3) == 0) {
                                const unsigned l4 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((it & 1) * 64 + wave * PIECES + q) * 1024));
                                asm volatile("s_mov_b32 m0, %0" : : "s"(l4) : "memory");
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    if (DMA && ((slot + NS - phase) % (NS / PIECES)) == 0) {
                        const int q = (slot / (NS / PIECES)) % PIECES;
                        if constexpr (MODE == 6) {
                            const unsigned s4 = __builtin_amdgcn_readfirstlane((unsigned)((wave * PIECES + (q & ~3)) * 1024) & 0xffffu);
                            if ((q & 3) == 0) dma16_imm_nom0<0>(rs, voff, s4); else if ((q & 3) == 1) dma16_imm_nom0<1>(rs, voff, s4);
                            else if ((q & 3) == 2) dma16_imm_nom0<2>(rs, voff, s4); else dma16_imm_nom0<3>(rs, voff, s4);
                        } else
                        if constexpr (MODE == 4) {
                            const unsigned l4 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((it & 1) * 64 + wave * PIECES + (q & ~3)) * 1024));
                            const unsigned s4 = __builtin_amdgcn_readfirstlane((unsigned)((wave * PIECES + (q & ~3)) * 1024) & 0xffffu);
                            if ((q & 3) == 0) dma16_imm<0>(rs, l4, voff, s4); else if ((q & 3) == 1) dma16_imm<1>(rs, l4, voff, s4);
                            else if ((q & 3) == 2) dma16_imm<2>(rs, l4, voff, s4); else dma16_imm<3>(rs, l4, voff, s4);
                        } else
                        dma16<MODE>(rs, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((it & 1) * 64 + wave * PIECES + q) * 1024)), voff, __builtin_amdgcn_readfirstlane((unsigned)((wave * PIECES + q) * 1024) & 0xffffu));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        if (DMA) asm volatile("s_waitcnt vmcnt(%c0)" : : "n"(PIECES) : "memory");   // bounded queue: the previous iteration's pieces have landed
        (void)p;
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = c1 - c0;
    if (sum == 12345.678f || sfill == 0x7fffffffu || lfill.x == 0x12345u) out[4096 + tid] = (unsigned long long)sum;
    if (FILL == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

static uint16_t bf16_of(float f) { union { float f; uint32_t u; } c; c.f = f; uint32_t u = c.u; return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <int NB, int PIECES, bool DMA, int MODE = 0, int FILL = 0>
static void run(const char* name, const uint4* d, const void* ds, unsigned long long* o, int nblk, const char* data) {
    const int iters = 20000;
    auto kern = k<NB, PIECES, DMA, MODE, FILL>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(NB == 64 ? 256 : 512), 140 * 1024, 0, d, ds, o, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(NB == 64 ? 256 : 512), 140 * 1024, 0, d, ds, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    // per iteration and SIMD: 128 MFMAs = 2048 cycles of matrix pipe in both modes
    const double flops = (double)nblk * 4 * 128 * 16 * 16 * 32 * 2 * (double)iters;
    printf("%-44s %-6s %9.1f cycles per iteration (128 MFMAs per SIMD = 2048)   %7.1f TFLOP/s  %.3f GHz\n", name, data, (double)h[0] / iters, flops / ms / 1e9,
           (double)h[0] / (ms * 1e6));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;
    uint16_t* h = (uint16_t*)malloc(65536 * 16);
    uint4* d; void* ds; unsigned long long* o;
    hipMalloc(&d, 65536 * 16); hipMalloc(&ds, 1 << 20); hipMalloc(&o, 8 * (4096 + 512 * 256)); hipMemset(ds, 0, 1 << 20);
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (int i = 0; i < 65536 * 8; ++i) {
            float u = 0.f; for (int q = 0; q < 12; ++q) u += rand() / (float)RAND_MAX; u -= 6.f;
            h[i] = data ? bf16_of(u * 0.05f) : 0;
        }
        hipMemcpy(d, h, 65536 * 16, hipMemcpyHostToDevice);
        const char* dn = data ? "random" : "zeros";
        for (int rep = 0; rep < 2; ++rep) {
            run<64, 16, false>("1 wave / SIMD, MFMAs only", d, ds, o, nblk, dn);
            run<64, 16, false, 0, 1>("1 wave / SIMD, MFMAs + 32 satisfied s_waitcnt", d, ds, o, nblk, dn);
            run<64, 16, false, 0, 5>("1 wave / SIMD, MFMAs + 32 x 3 s_waitcnt", d, ds, o, nblk, dn);
            run<64, 16, false, 0, 2>("1 wave / SIMD, MFMAs + 32 s_nop", d, ds, o, nblk, dn);
            run<64, 16, false, 0, 3>("1 wave / SIMD, MFMAs + 32 s_add", d, ds, o, nblk, dn);
            run<64, 16, false, 0, 4>("1 wave / SIMD, MFMAs + 32 ds_read_b128", d, ds, o, nblk, dn);
            run<64, 16, true>("1 wave / SIMD, 16 pieces per 128 MFMAs", d, ds, o, nblk, dn);
            run<64, 16, true, 1>("1 wave / SIMD, 16 pieces, no VGPR address", d, ds, o, nblk, dn);
            run<64, 16, true, 2>("1 wave / SIMD, 16 pieces, one lane active", d, ds, o, nblk, dn);
            run<64, 16, true, 3>("1 wave / SIMD, 16 pieces of dword (256 B)", d, ds, o, nblk, dn);
            run<64, 16, true, 4>("1 wave / SIMD, 16 pieces, M0 once per four", d, ds, o, nblk, dn);
            run<64, 16, true, 6>("1 wave / SIMD, 16 pieces, M0 per four, a slot early", d, ds, o, nblk, dn);
            run<32, 8, false>("2 waves / SIMD, MFMAs only", d, ds, o, nblk, dn);
            run<32, 8, true>("2 waves / SIMD, 8 pieces per 64 MFMAs each", d, ds, o, nblk, dn);
        }
    }
    return 0;
}
