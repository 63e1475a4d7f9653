#include <hip/hip_runtime.h>
// MODE register (hwreg id 1), bit 23 = FP16_OVFL: overflowing fp16 VALU results clamp to +-65504 instead of inf
#define OG_HWREG_FP16_OVFL (1 | (23 << 6) | (0 << 11))
__global__ void k(const float* x, _Float16* y, int ovfl) {
    if (ovfl) __builtin_amdgcn_s_setreg(OG_HWREG_FP16_OVFL, 1);
    y[threadIdx.x] = (_Float16)x[threadIdx.x];
}
int main() {
    float hx[4] = {1.0f, 70000.f, -1e9f, 65519.f};
    float* dx; _Float16* dy; hipMalloc(&dx, 16); hipMalloc(&dy, 8);
    hipMemcpy(dx, hx, 16, hipMemcpyHostToDevice);
    for (int o = 0; o < 2; ++o) {
        hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, dx, dy, o);
        _Float16 hy[4]; hipMemcpy(hy, dy, 8, hipMemcpyDeviceToHost);
        printf("ovfl=%d: %g %g %g %g\n", o, (float)hy[0], (float)hy[1], (float)hy[2], (float)hy[3]);
    }
    return 0;
}
