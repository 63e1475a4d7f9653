// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements does lane l receive?  LDS holds element index i at i.
// Case A: lane address = 8 * l bytes (row-major [*][16] bf16 image, rows of 32 B).
// Case B: row stride 64 B: lane (g = l >> 4, i = l & 15) -> address g * 256 + (i >> 2) * 64 + (i & 3) * 8.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
__global__ void k(int* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)((char*)lds + 8 * l));
    const int g = l >> 4, i = l & 15;
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)((char*)lds + g * 256 + (i >> 2) * 64 + (i & 3) * 8));
    for (int j = 0; j < 4; ++j) { out[l * 8 + j] = a[j]; out[l * 8 + 4 + j] = b[j]; }
}
int main() {
    int* d; hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d A:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 8 + j]);
        printf("   B:"); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 8 + 4 + j]); printf("\n");
    }
    return 0;
}
