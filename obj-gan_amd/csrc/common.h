// Shared helpers for the objgan_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OG_WAVE 64

// Return convention of every entry point (mirrors the reference FFI, which
// returns 1 on success and 0 on an argument error -- reference
// image_generation/models/roi_align/src/roi_align_cuda.c:18-22 -- and replaces
// its exit(-1) on a launch failure by a negative hipError code).
#define OG_OK 1
#define OG_BAD_ARGS 0

// First statement of every exported entry point: HIP's last-error slot is per thread and sticky, so
// an error left behind by somebody else's call (torch probing devices, a profiler attaching) would
// otherwise be reported as the failure of our next launch.
#define OG_ENTRY() ((void)hipGetLastError())

static inline int og_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? OG_OK : -(int)e;
}

// Development switches.  The shipped library reads no environment: OG_KNOB(fn, "NAME", default) defines fn() as
// the constant.  A development build (OBJGAN_DEV=1 python -m objgan_hip.build, i.e. -DOG_DEV) reads the integer
// from the environment once instead, for A/B runs of tools/conv_bench and bench.py on the GPU box.
#ifdef OG_DEV
#include <stdlib.h>
static inline int og_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}
#define OG_KNOB(fn, name, dflt) \
    static int fn() { static int v = -0x7fffffff; if (v == -0x7fffffff) v = og_env_int(name, dflt); return v; }
#else
#define OG_KNOB(fn, name, dflt) static inline int fn() { return dflt; }
#endif

static inline int og_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Grid for HBM-bound grid-stride kernels: enough workgroups to fill 256 CUs
// (8 XCDs x 32) several times over, capped so the tail stays short.
static inline int og_stream_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

__device__ __forceinline__ float og_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float og_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (<= 16 waves); `red` is >= 16 floats of LDS.
__device__ __forceinline__ float og_block_sum(float v, float* red) {
    v = og_wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float r = (lane < nw) ? red[lane] : 0.f;
    r = og_wave_sum(r);
    return r;
}
