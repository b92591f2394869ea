// ds_read_b64_tr_b16 semantics probe (build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe.bin; prints which LDS element each lane receives)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;   // element value = its index; row r (stride 64), col c -> 64 r + c
    __syncthreads();
    int l = threadIdx.x, i = l & 15, g = l >> 4;
    // group g reads rows 4g..4g+3 (stride 64 elements), cols 0..15; lane i supplies row 4g + (i>>2), cols 4(i&3)..
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + (4 * g + (i >> 2)) * 64 + (i & 3) * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d); short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
    return 0;
}
