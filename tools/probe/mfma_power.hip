// Sustained matrix-pipe rate under the power cap: v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16, one wave per SIMD, 256 accumulator
// registers per wave either way, operands = random bf16 (N(0,1)-like) or zeros.  Prints TFLOP/s and the clock implied by back-to-back issue
// (32x32x16: 8 passes = 32 cycles... measured as flops / (flops per CU-cycle)).  Build: hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    bf16x8_t a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 x = src[(tid * 16 + i) & 0xffff], y = src[(tid * 16 + 8 + i) & 0xffff];
        a[i] = *reinterpret_cast<bf16x8_t*>(&x);
        b[i] = *reinterpret_cast<bf16x8_t*>(&y);
    }
    float sum = 0.f;
    if (MODE == 0) {
        f32x16_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks * 4 + i], b[ks * 4 + j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    } else {
        f32x4_t acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    }
    if (sum == 12345.678f) out[tid] = sum;
}

static uint16_t bf16_of(float f) { union { float f; uint32_t u; } c; c.f = f; uint32_t u = c.u; return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;
    uint16_t* h = (uint16_t*)malloc(65536 * 16);
    uint4* d; float* o;
    hipMalloc(&d, 65536 * 16); hipMalloc(&o, 4 * 256 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int data = 0; data < 2; ++data) {
        srand(1);
        for (int i = 0; i < 65536 * 8; ++i) {
            float u = 0.f; for (int q = 0; q < 12; ++q) u += rand() / (float)RAND_MAX; u -= 6.f;
            h[i] = data ? bf16_of(u * 0.05f) : 0;
        }
        hipMemcpy(d, h, 65536 * 16, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep)
            for (int mode = 0; mode < 2; ++mode) {
                // flops per wave and iteration: mode 0: 32 MFMAs x 32*32*16*2; mode 1: 64 x 16*16*32*2  (identical: 1 048 576)
                const int iters = 400000;   // 32 MFMA x 32 cycles (or 64 x 16) = 2048 cycles per iteration -> ~0.4 s at 2 GHz
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, d, o, 1000); else hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, d, o, 1000);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, d, o, iters); else hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, d, o, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = (double)nblk * 4 * iters * 1048576.0 * 2 / 2;  // 32*32*16*2 = 32768 flop x 32 = 1 048 576
                const double cycles = (double)iters * 2048;
                printf("{\"data\": \"%s\", \"mfma\": \"%s\", \"blocks\": %d, \"ms\": %.2f, \"tflops\": %.1f, \"implied_ghz\": %.3f}\n", data ? "random" : "zeros",
                       mode ? "16x16x32" : "32x32x16", nblk, ms, flops / ms / 1e9, cycles / ms / 1e6);
                fflush(stdout);
            }
    }
    return 0;
}
