// Stand-alone rate probe of v_mfma_f32_32x32x16_bf16 on gfx950.
//
// What it answers: how fast does the bf16 matrix pipe run (a) as a pipe -- independent accumulator chains, constant
// operands -- and (b) in the issue pattern of the bf16x3 convolution loop (6 MFMAs per 16-deep K step and row group,
// operands changing every step), for zero / random / split-fp32 (h, m, l) operand data, at 1 and 2 waves per SIMD?
// It reports TFLOP/s of bf16 MFMA work for the first launch and for the last of six back-to-back launches (the chip's
// power management settles within ~0.2 s), the shader cycles (s_memtime) a wave spends per MFMA of its own (two waves of a
// SIMD do not share the pipe evenly: issue is oldest-first), and the shader clock (s_memtime against the constant 100 MHz
// s_memrealtime).  MFMA_RATE_WAVES=1 prints the start / end times of the waves of the first two workgroups.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_rate tools/mfma_rate.hip && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

typedef __attribute__((__vector_size__(8 * sizeof(short)))) short bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// NCH independent accumulator chains (row groups), SIX products per chain and iteration (the bf16x3 K step), NSET operand
// sets cycled (operands toggle between consecutive MFMAs when NSET > 1).  RUN = consecutive MFMAs on the SAME accumulator:
// 1 = product-major / chain-minor (consecutive MFMAs never share an accumulator: the order of conv_igemm3_kernel in round 3),
// 6 = chain-major (a row group's six products back to back).  PIN: a sched_barrier behind every MFMA (hipcc re-orders
// independent MFMAs chain-minor by itself).  NT threads per workgroup: 256 = one wave per SIMD and workgroup, 512 = two.
template <int NCH, int NSET, int RUN, int PIN, int NT>
__global__ __launch_bounds__(NT) void mfma_probe(const bf16x8* __restrict__ opnd, float* __restrict__ out, int iters,
                                                  unsigned long long* __restrict__ cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[NSET], b[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        a[s] = opnd[(2 * s) * 64 + lane];
        b[s] = opnd[(2 * s + 1) * 64 + lane];
    }
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[c][j] = 0.f;
    __syncthreads();
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 6 / RUN; ++g)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int r = 0; r < RUN; ++r) {
                    const int q = g * RUN + r;
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(q + c) % NSET], b[(q * 2 + c) % NSET], acc[c], 0, 0, 0);
                    if (PIN == 1 || (PIN == 2 && r == RUN - 1 && c == NCH - 1)) __builtin_amdgcn_sched_barrier(0);
                }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[c][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) {           // per wave: shader-clock start / end, constant 100 MHz clock start / end
        unsigned long long* c = cyc + ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 4;
        c[0] = t0; c[1] = t1; c[2] = w0; c[3] = wall_clock64();
    }
}

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

static float frand() { return (float)((double)rand() / RAND_MAX) * 2.f - 1.f; }

// data 0: zeros; 1: uniform random [-1, 1) bf16; 2: sets (0,1,2) = (h, m, l) of random fp32 values, the bf16x3 operand mix
static void fill(std::vector<unsigned short>& v, int nset, int data) {
    v.assign((size_t)nset * 2 * 64 * 8, 0);
    if (data == 0) return;
    for (int op = 0; op < 2; ++op)
        for (int e = 0; e < 64 * 8; ++e) {
            if (data == 1) {
                for (int s = 0; s < nset; ++s) v[((size_t)(2 * s + op) * 64 * 8) + e] = f2bf(frand());
            } else {
                for (int s0 = 0; s0 < nset; s0 += 3) {
                    float x = frand() * (op ? 1.f : 0.05f);
                    unsigned short h = f2bf(x); float r = x - bf2f(h);
                    unsigned short m = f2bf(r); float r2 = r - bf2f(m);
                    unsigned short l = f2bf(r2);
                    unsigned short hml[3] = {h, m, l};
                    for (int k = 0; k < 3 && s0 + k < nset; ++k) v[((size_t)(2 * (s0 + k) + op) * 64 * 8) + e] = hml[k];
                }
            }
        }
}

template <int NCH, int NSET, int RUN, int PIN, int NT>
static void run(const char* name, int wgs_per_cu, int data, bf16x8* d_op, float* d_out, unsigned long long* d_cyc, int ncu) {
    std::vector<unsigned short> h;
    fill(h, NSET, data);
    CK(hipMemcpy(d_op, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int blocks = ncu * wgs_per_cu;
    const int wps = wgs_per_cu * NT / 256;               // waves per SIMD
    const int per_iter = NCH * 6;
    const int iters = (int)(3.0e6 / per_iter / wps);     // ~3 M MFMAs per SIMD: ~40-75 ms per launch
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = (double)blocks * (NT / 64) * iters * per_iter * 2.0 * 32 * 32 * 16;
    float ms_first = 0, ms_last = 0;
    const int nw = blocks * (NT / 64);
    std::vector<unsigned long long> c((size_t)nw * 4);
    const int REPS = 6;                                  // back to back: ~0.3-0.4 s of sustained load per line
    for (int rep = 0; rep < REPS; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_probe<NCH, NSET, RUN, PIN, NT>), dim3(blocks), dim3(NT), 0, 0, d_op, d_out, iters, d_cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) ms_first = ms;
        ms_last = ms;
    }
    CK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
    // per wave: cycles per MFMA of its own (min / mean / max over waves) and the shader clock = cycles / 100 MHz ticks
    double cmin = 1e30, cmax = 0, csum = 0, mhz = 0;
    unsigned long long first = ~0ull, last = 0;
    for (int w = 0; w < nw; ++w) {
        const double cy = (double)(c[w * 4 + 1] - c[w * 4 + 0]);
        const double per = cy / ((double)iters * per_iter);
        cmin = std::min(cmin, per); cmax = std::max(cmax, per); csum += per;
        mhz += cy / ((double)(c[w * 4 + 3] - c[w * 4 + 2]) / 100.0);
        first = std::min(first, c[w * 4 + 2]); last = std::max(last, c[w * 4 + 3]);
    }
    printf("%-38s w/SIMD %d %-6s first %6.1f TF sustained %6.1f TF | cyc per own MFMA min %5.1f mean %5.1f max %5.1f | clock %4.0f MHz | span %5.1f ms\n",
           name, wps, data == 0 ? "zeros" : data == 1 ? "random" : "hml", flops / (ms_first * 1e-3) * 1e-12,
           flops / (ms_last * 1e-3) * 1e-12, cmin, csum / nw, cmax, mhz / nw, (double)(last - first) / 1e5);
    if (getenv("MFMA_RATE_WAVES")) {                    // start / end of the waves of workgroup 0 and 1, in us from the first start
        for (int w = 0; w < 2 * (NT / 64) && w < nw; ++w)
            printf("    wg %d wave %d: %9.1f .. %9.1f us\n", w / (NT / 64), w % (NT / 64), (double)(c[w * 4 + 2] - first) / 100.0,
                   (double)(c[w * 4 + 3] - first) / 100.0);
    }
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("# %s, %d CUs, max clock %d MHz.  v_mfma_f32_32x32x16_bf16; TF = bf16 MFMA TFLOP/s (divide by 6 for bf16x3 fp32-equivalent)\n",
           p.name, ncu, p.clockRate / 1000);
    bf16x8* d_op; float* d_out; unsigned long long* d_cyc;
    CK(hipMalloc(&d_op, 64 * 1024)); CK(hipMalloc(&d_out, (size_t)ncu * 2 * 512 * 4 * 4)); CK(hipMalloc(&d_cyc, (size_t)ncu * 2 * 8 * 4 * 8));
    for (int pass = 0; pass < 2; ++pass)          // the whole table twice: order / thermal history effects show as differences
    for (int data = 0; data < 3; ++data) {
        if (data == 1 && pass == 1) continue;
        printf("## pass %d, operand data: %s\n", pass, data == 0 ? "zeros" : data == 1 ? "uniform random bf16" : "(h, m, l) splits of random fp32");
        run<1, 1, 6, 0, 256>("1 chain, constant operands", 1, data, d_op, d_out, d_cyc, ncu);
        run<4, 1, 1, 0, 256>("4 chains, constant operands", 1, data, d_op, d_out, d_cyc, ncu);
        run<3, 6, 1, 1, 256>("3 rowgrp, product-major (RUN 1)", 1, data, d_op, d_out, d_cyc, ncu);
        run<3, 6, 6, 1, 256>("3 rowgrp, rowgroup-major (RUN 6)", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 1, 1, 256>("6 rowgrp, product-major (RUN 1)", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 6, 1, 256>("6 rowgrp, rowgroup-major (RUN 6)", 1, data, d_op, d_out, d_cyc, ncu);
        run<4, 1, 1, 0, 256>("4 chains, constant operands, 2 WG/CU", 2, data, d_op, d_out, d_cyc, ncu);
        run<3, 6, 1, 1, 256>("3 rowgrp, RUN 1, 2 WG/CU", 2, data, d_op, d_out, d_cyc, ncu);
        run<3, 6, 2, 1, 256>("3 rowgrp, RUN 2, 2 WG/CU", 2, data, d_op, d_out, d_cyc, ncu);
        run<3, 6, 3, 1, 256>("3 rowgrp, RUN 3, 2 WG/CU", 2, data, d_op, d_out, d_cyc, ncu);
        run<3, 6, 6, 1, 256>("3 rowgrp, RUN 6, 2 WG/CU", 2, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 1, 1, 512>("6 rowgrp, RUN 1, 8-wave WG", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 1, 0, 512>("6 rowgrp, RUN 1 unpinned, 8-wave WG", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 1, 2, 512>("6 rowgrp, RUN 1 pinned per iteration, 8-wave", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 6, 0, 512>("6 rowgrp, RUN 6 unpinned, 8-wave WG", 1, data, d_op, d_out, d_cyc, ncu);
        run<3, 6, 1, 0, 256>("3 rowgrp, RUN 1 unpinned, 2 WG/CU", 2, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 2, 1, 512>("6 rowgrp, RUN 2, 8-wave WG", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 3, 1, 512>("6 rowgrp, RUN 3, 8-wave WG", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 6, 6, 1, 512>("6 rowgrp, RUN 6, 8-wave WG", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 2, 1, 1, 512>("6 rowgrp, RUN 1, 2 operand sets, 8-wave", 1, data, d_op, d_out, d_cyc, ncu);
        run<6, 2, 6, 1, 512>("6 rowgrp, RUN 6, 2 operand sets, 8-wave", 1, data, d_op, d_out, d_cyc, ncu);
    }
    return 0;
}
