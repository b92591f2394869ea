// Issue cost of the VALU instructions the flash-attention softmax is made of, one wave per SIMD and two waves per SIMD (gfx950):
// cycles per instruction for v_fma_f32, v_pk_fma_f32, v_max3_f32, v_exp_f32, v_cvt_pk_bf16_f32, measured with s_memtime around N
// independent chains; and the same VALU stream beside a co-resident wave that issues back-to-back MFMAs (does VALU of wave A run under
// the MFMAs of wave B on one SIMD?).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define REP 64
// MODE 0 fma, 1 pk_fma, 2 max3, 3 exp2, 4 cvt_pk_bf16, 5 pk_add, 6 pk_mul; role: waves with (wave & rolemask) != 0 run MFMAs instead
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mfma_partner) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.001f * (float)(lane + i);
    const float a = 1.0001f, b = 0.0003f;
    const bool partner = mfma_partner && wave >= 4;   // waves 4-7 are the second wave of each SIMD
    f32x16_t acc = {0};
    bf16x8_t fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (float)lane); fb[i] = (__bf16)(0.02f * (float)i); }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (partner) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < REP / 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < REP; ++r) {
                if constexpr (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r & 15]) : "v"(a), "v"(b)); }
                else if constexpr (MODE == 1) {
                    f2 v = {x[(2 * r) & 15], x[(2 * r + 1) & 15]};
                    const f2 aa = {a, a}, bb = {b, b};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
                    x[(2 * r) & 15] = v.x; x[(2 * r + 1) & 15] = v.y;
                } else if constexpr (MODE == 2) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[r & 15]) : "v"(a), "v"(b)); }
                else if constexpr (MODE == 3) { asm volatile("v_exp_f32 %0, %0" : "+v"(x[r & 15])); }
                else if constexpr (MODE == 4) {
                    unsigned o;
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(x[r & 15]), "v"(x[(r + 1) & 15]));
                    x[r & 15] = __uint_as_float(o);
                } else if constexpr (MODE == 5) {
                    f2 v = {x[(2 * r) & 15], x[(2 * r + 1) & 15]};
                    const f2 aa = {a, a};
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(aa));
                    x[(2 * r) & 15] = v.x; x[(2 * r + 1) & 15] = v.y;
                } else {
                    f2 v = {x[(2 * r) & 15], x[(2 * r + 1) & 15]};
                    const f2 aa = {a, a};
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(aa));
                    x[(2 * r) & 15] = v.x; x[(2 * r + 1) & 15] = v.y;
                }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int nthreads, int partner) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&cyc, 8 * sizeof(unsigned long long));
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(nthreads), 0, 0, out, cyc, iters, partner);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(nthreads), 0, 0, out, cyc, iters, partner);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on gfx950; readcyclecounter = s_memtime -> report raw ticks per instruction and let the ratio speak
    printf("%-18s threads %3d partner-mfma %d : wave0 %.3f ticks/instr", name, nthreads, partner, (double)h[0] / (iters * REP));
    if (nthreads > 256) printf("   wave4 %.3f ticks/%s", (double)h[4] / (iters * (partner ? REP / 4 : REP)), partner ? "mfma" : "instr");
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int nt = cfg == 0 ? 256 : 512, pa = cfg == 2;
        run<0>("v_fma_f32", nt, pa);
        run<1>("v_pk_fma_f32", nt, pa);
        run<5>("v_pk_add_f32", nt, pa);
        run<6>("v_pk_mul_f32", nt, pa);
        run<2>("v_max3_f32", nt, pa);
        run<3>("v_exp_f32", nt, pa);
        run<4>("v_cvt_pk_bf16_f32", nt, pa);
    }
    return 0;
}
