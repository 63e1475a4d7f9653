// Does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs?  (fp16x2's low piece l of a small element is subnormal when the
// tensor scale leaves the element far below the maximum; a producer-side split with a loose scale bound relies on them.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/denorm_probe tools/denorm_probe.hip && tools/denorm_probe
// One wave: A[m][k] = a for all m, k; B[k][n] = b; D[m][n] = 16 * a * b expected, for a few (a, b) pairs around the
// fp16 subnormal range (smallest normal 2^-14 = 6.1e-5, smallest subnormal 2^-24 = 6.0e-8).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* ab, float* out, int n) {
    for (int i = 0; i < n; ++i) {
        const _Float16 a = (_Float16)ab[2 * i], b = (_Float16)ab[2 * i + 1];
        f16x8 va, vb;
        for (int j = 0; j < 8; ++j) { va[j] = a; vb[j] = b; }
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
        if (threadIdx.x == 0) { out[3 * i] = (float)a; out[3 * i + 1] = (float)b; out[3 * i + 2] = acc[0]; }
    }
}

int main() {
    const int n = 8;
    float h[2 * n] = {1.0f, 1.0f,              // sanity: 16
                      6.1035156e-5f, 1.0f,     // smallest normal 2^-14
                      3.0517578e-5f, 1.0f,     // 2^-15: subnormal
                      5.9604645e-8f, 1.0f,     // 2^-24: smallest subnormal
                      5.9604645e-8f, 1024.0f,  // subnormal x large
                      3.0517578e-5f, 3.0517578e-5f,   // subnormal x subnormal = 2^-30 (fine in fp32)
                      1.7881393e-7f, 2.0f,     // 3 * 2^-24
                      9.5367432e-7f, 0.5f};    // 2^-20
    float *d_ab, *d_out; float out[3 * n];
    hipMalloc(&d_ab, sizeof(h)); hipMalloc(&d_out, sizeof(out));
    hipMemcpy(d_ab, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_ab, d_out, n);
    hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        const double want = 16.0 * (double)out[3 * i] * (double)out[3 * i + 1];
        printf("a %.9e  b %.9e  mfma %.9e  expected %.9e  %s\n", out[3 * i], out[3 * i + 1], out[3 * i + 2], want,
               fabs(out[3 * i + 2] - want) <= 1e-6 * fabs(want) ? "ok" : "DIFFERS");
    }
    return 0;
}
