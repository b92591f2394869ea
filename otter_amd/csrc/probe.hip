// probe.hip -- machine calibration for bench.py's `roofline` object: the matrix-pipe rate this part sustains RIGHT NOW.
//
// The bench line prices its kernels against the 2.5 PFLOP/s dense bf16 peak of MI355X_MICROARCH.md.  That figure is 256 CUs x 4 SIMDs x
// 1024 FLOP per cycle at ~2.4 GHz; a part under its power cap does not hold that clock on real operands (round 2: back-to-back MFMAs on
// random bf16 sustain ~2.07 PF, on zeros ~2.5 PF, tools/probe/mfma_power.hip), and boxes of one pool differ by ~5 %.  otter_probe_mfma
// runs the product GEMM's own instruction (v_mfma_f32_16x16x32_bf16, 64 independent accumulator blocks = 256 accumulator registers per
// wave, one wave per SIMD, every CU) back to back on caller-provided operand bits and reports, from inside the kernel, the shader-clock
// cycles (s_memtime) and the 100 MHz wall-clock ticks (s_memrealtime) the loop took: FLOP/s = flops / ticks, clock = cycles / ticks.
// No memory traffic inside the loop, nothing else on the chip: an upper bound for any MFMA-bound kernel on this box at this moment.
#include "common.h"

typedef __bf16 probe_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float probe_f32x4_t __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void probe_mfma_kernel(const uint4* __restrict__ src, unsigned long long* __restrict__ out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    probe_bf16x8_t a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 x = src[(tid * 16 + i) & 0xffff], y = src[(tid * 16 + 8 + i) & 0xffff];
        a[i] = __builtin_bit_cast(probe_bf16x8_t, x);
        b[i] = __builtin_bit_cast(probe_bf16x8_t, y);
    }
    probe_f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = probe_f32x4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_readcyclecounter();        // s_memtime: shader clock
    const unsigned long long r0 = wall_clock64();                       // s_memrealtime: 100 MHz
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0) {          // wave 0 of every workgroup: [2 * block] = shader cycles, [2 * block + 1] = 100 MHz ticks
        out[2 * blockIdx.x] = c1 - c0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (sum == 12345.678f) out[2 * gridDim.x + tid] = (unsigned long long)sum;   // keeps the accumulators alive; never taken for sane operands
}

}  // namespace

extern "C" {

int otter_probe_mfma(const void* operands, void* out, int iters, int n_workgroups, void* stream) {
    OTTER_REQUIRE(operands && out && iters > 0 && n_workgroups > 0 && n_workgroups <= 4096, "probe_mfma: operands (1 MiB), out, iters > 0, 1..4096 workgroups");
    hipLaunchKernelGGL(probe_mfma_kernel, dim3((unsigned)n_workgroups), dim3(256), 0, (hipStream_t)stream, (const uint4*)operands,
                       (unsigned long long*)out, iters);
    OTTER_CHECK_LAUNCH("probe_mfma");
    return OTTER_OK;
}

}  // extern "C"
