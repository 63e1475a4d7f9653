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

// fp16x2 kernels convert fp32 -> fp16 under a scale derived from the tensor's maximum.  With a correct maximum nothing
// overflows (max |x| * 2^s < 2^15); MODE.FP16_OVFL (hwreg 1, bit 23) makes an overflowing conversion clamp to +-65504
// instead of producing inf, so that a STALE / too-small maximum degrades the result gracefully (saturated operands,
// finite outputs) instead of poisoning it with inf / NaN.  Per-wave state, set in the kernel prologue, one scalar
// instruction (tools/ovfl_probe.hip; tests/test_kernels_gpu.py::test_fp16x2_survives_wrong_maxima).
__device__ __forceinline__ void og_fp16_saturate() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }

// ---- partial maxima of |x| (scale input of the fp16x2 convolution arithmetic) ------------------------------------
// A tensor's maximum travels as OG_AMAX_SLOTS non-negative floats whose maximum is max |x| (consumers reduce them in
// their prologue: 16 loads per lane).  Producers fill them in one of two ways, both free of float atomics, of ordering
// assumptions and of extra launches: a launch of G <= OG_AMAX_SLOTS workgroups OWNS its slots (og_amax_own: workgroup b
// stores its maximum into slot b and zeroes slots b + G, b + 2G, ...); a launch with more workgroups adds its maxima with
// integer atomicMax on the float's bit pattern (og_amax_atomic) into slots an EARLIER kernel of the same sequence zeroed.
#define OG_AMAX_SLOTS 1024
__device__ __forceinline__ float og_block_max(float m, float* red /* >= 4 floats of LDS */) {
    m = og_wave_max(m);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = m;
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
    float v = red[0];
    for (int w = 1; w < nw; ++w) v = fmaxf(v, red[w]);
    return v;
}
__device__ __forceinline__ void og_amax_own(float block_max, float* __restrict__ amax, int bid, int nblocks) {
    if (threadIdx.x == 0) {
        amax[bid] = block_max;
        for (int k = bid + nblocks; k < OG_AMAX_SLOTS; k += nblocks) amax[k] = 0.f;
    }
}
__device__ __forceinline__ void og_amax_atomic(float block_max, float* __restrict__ amax, unsigned bid) {
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned*>(amax) + (bid & (OG_AMAX_SLOTS - 1)), __float_as_uint(block_max));
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
