// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O2 tools/pmc_calib.cpp -o tools/pmc_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o cal -- tools/pmc_calib     (then WRITE_SIZE)
// Every kernel touches each byte of a 1 GiB buffer (4x the Infinity Cache) exactly once:
//   calib_stream_b128   16 B per lane, consecutive lanes consecutive float4s        (the guide's pattern: x2)
//   calib_gather_b32    one dword per lane, lanes = consecutive pixels, one channel per instruction -- the
//                       pixel gather of conv_igemm3_kernel (buffer load, per-lane offset = pixel, scalar
//                       offset = channel)
//   calib_write_b32     one dword store per lane, same shape (the conv epilogue: a lane writes its pixel,
//                       channel after channel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void calib_stream_b128(const float4* __restrict__ x, float* __restrict__ out, long n4) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = x[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_gather_b32(const float* __restrict__ x, float* __restrict__ out, int C, long HW) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;       // one pixel per lane
    if (p >= HW) return;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += x[(long)c * HW + p];
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void calib_write_b32(float* __restrict__ y, int C, long HW) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    for (int c = 0; c < C; ++c) y[(long)c * HW + p] = (float)c;
}

int main() {
    const int C = 256;
    const long HW = 1 << 20;                       // 256 channels x 1 Mi pixels x 4 B = 1 GiB
    const long n = (long)C * HW;
    float *x, *out;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(x, 0, n * 4));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_stream_b128, dim3(4096), dim3(256), 0, 0, (const float4*)x, out, n / 4);
        hipLaunchKernelGGL(calib_gather_b32, dim3((unsigned)(HW / 256)), dim3(256), 0, 0, (const float*)x, out, C, HW);
        hipLaunchKernelGGL(calib_write_b32, dim3((unsigned)(HW / 256)), dim3(256), 0, 0, x, C, HW);
    }
    CK(hipDeviceSynchronize());
    printf("known bytes per launch: %ld\n", n * 4);
    return 0;
}
