// Stand-alone micro-benchmark of the conv kernels of libobjgan_hip.so on the hot-path shapes
// (development aid; no Python / torch start-up cost on the GPU box).
//
//   hipcc --offload-arch=gfx950 -O2 tools/conv_bench.cpp -Iinclude -L obj-gan_amd/objgan_hip -lobjgan_hip \
//         -Wl,-rpath,'$ORIGIN/../obj-gan_amd/objgan_hip' -o tools/conv_bench
//   tools/conv_bench [filter] [iters] [math: 0 fp32 | 1 bf16 inputs | 2 bf16x3 | 4 fp16x2 | 5 fp16x2 on records | 6: 5 + weight gradient on pre-split dy]
//   (math 5: forward / data gradient read the pre-split fp16 record of their pixel operand; the record passes are timed
//    on their own line; the weight gradient runs math 4; `hash` columns: FNV-1a of the output bits -- equal between
//    math 4 and math 5 when the two are bit-identical)
//
// For every shape: forward, data gradient and weight gradient are timed with hipEvents and
// reported as algorithmic TFLOP/s (2*N*OH*OW*Cout*Cin*k*k); a sample of output elements is
// checked against a double-precision host evaluation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "objgan_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int N, Cin, H, W, Cout, k, s, p, refl, up; };

static const Shape SHAPES[] = {
    {"res1_128 194->388 3x3 refl", 16, 194, 128, 128, 388, 3, 1, 1, 1, 0},
    {"res2_128 194->194 3x3 refl", 16, 194, 128, 128, 194, 3, 1, 1, 1, 0},
    {"res1_64  194->388 3x3 refl", 16, 194, 64, 64, 388, 3, 1, 1, 1, 0},
    {"res1_32  194->388 3x3 refl", 16, 194, 32, 32, 388, 3, 1, 1, 1, 0},
    {"up_256   194->96 3x3 up", 16, 194, 128, 128, 96, 3, 1, 1, 0, 1},
    {"hmap_256 80->24 3x3 refl", 16, 80, 256, 256, 24, 3, 1, 1, 1, 0},
    {"shp_512  80->12 3x3 refl", 16, 80, 512, 512, 12, 3, 1, 1, 1, 0},
    {"patd_l1  3->96 4x4 s2 @256", 16, 3, 256, 256, 96, 4, 2, 1, 0, 0},
    {"objd_l1  15->96 4x4 s2 @512", 16, 15, 512, 512, 96, 4, 2, 1, 0, 0},
    {"objd_l2  96->192 4x4 s2 @256", 16, 96, 256, 256, 192, 4, 2, 1, 0, 0},
    {"objd_l3  192->384 4x4 s2 @128", 16, 192, 128, 128, 384, 4, 2, 1, 0, 0},
    {"d_l4     384->768 4x4 s2 @64", 16, 384, 64, 64, 768, 4, 2, 1, 0, 0},
    {"d_l4s    384->768 4x4 s2 @32", 16, 384, 32, 32, 768, 4, 2, 1, 0, 0},
    {"joint    1024->768 3x3 @16", 16, 1024, 16, 16, 768, 3, 1, 1, 0, 0},
    {"rgb_256  48->3 3x3", 16, 48, 256, 256, 3, 3, 1, 1, 0, 0},
    {"outlogit 768->1 4x4 s2 @4", 13, 768, 4, 4, 1, 4, 2, 0, 0, 0},
    {"roi_code 384->384 4x4 @5", 160, 384, 5, 5, 384, 4, 1, 1, 0, 0},
    {"incep_17 192->192 1x7 (as 3x3)", 16, 192, 17, 17, 192, 3, 1, 1, 0, 0},
};

static unsigned g_seed = 12345u;
static int g_math = 0;      // 0 fp32, 1 bf16 inputs, 2 bf16x3 (fp32 split three ways on the bf16 MFMA) (argv[3])
// full 24-bit significands: the l pieces of the bf16x3 split must not be zero
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffffff) / 8388608.0f - 1.0f; }

static int reflect(int i, int L) { if (i < 0) i = -i; if (i >= L) i = 2 * (L - 1) - i; return i; }

int main(int argc, char** argv) {
    const char* filt = argc > 1 ? argv[1] : "";
    const int iters = argc > 2 ? atoi(argv[2]) : 5;
    g_math = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int g_math_arg = g_math;
    for (const Shape& sh : SHAPES) {
        g_math = g_math_arg;
        if (filt[0] && !strstr(sh.name, filt)) continue;
        const int LH = sh.up ? 2 * sh.H : sh.H, LW = sh.up ? 2 * sh.W : sh.W;
        const int OH = (LH + 2 * sh.p - sh.k) / sh.s + 1, OW = (LW + 2 * sh.p - sh.k) / sh.s + 1;
        const int T = sh.k * sh.k;
        const size_t nx = (size_t)sh.N * sh.Cin * sh.H * sh.W, ny = (size_t)sh.N * sh.Cout * OH * OW;
        const size_t nw = (size_t)sh.Cout * sh.Cin * T;
        std::vector<float> hx(nx), hw(nw), hy(ny), hg(ny);
        for (auto& v : hx) v = frand();
        for (auto& v : hw) v = frand() * 0.05f;
        for (auto& v : hg) v = frand();
        float *dx, *dw, *dy, *dg, *dgx, *dgw, *wt;
        const int TH = sh.refl ? LH + 2 : LH, TW = sh.refl ? LW + 2 : LW;   // dgrad target (padded when reflect)
        const size_t ngx = (size_t)sh.N * sh.Cin * TH * TW;
        long nwt = objgan_conv_packed_floats(sh.Cout > sh.Cin ? sh.Cout : sh.Cin, sh.Cout > sh.Cin ? sh.Cout : sh.Cin, T);
        if (nwt < (long)(sh.Cout + 1) * 256) nwt = (long)(sh.Cout + 1) * 256;
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dg, ny * 4));
        CK(hipMalloc(&dgx, ngx * 4)); CK(hipMalloc(&dgw, nw * 4)); CK(hipMalloc(&wt, nwt * 4));
        // one scratch buffer large enough for every split-K / weight-gradient workspace of this shape
        float* wsb;
        const long nwsb = 192L << 20;     // 768 MB of floats (math 6: + the fp16 pair of dy)
        CK(hipMalloc(&wsb, nwsb * 4));
        float* mxx; float* mxg;          // math 4 (fp16x2): per-workgroup maxima of x and of dy (objgan_absmax_partials)
        CK(hipMalloc(&mxx, 1024 * 4)); CK(hipMalloc(&mxg, 1024 * 4));
        float *recx = nullptr, *recg = nullptr;   // math 5: the fp16 records of x and dy
        const int wmath = g_math >= 6 ? g_math : 5;     // math 6: as 5, the weight gradient ALSO reads dy as its fp16 pair; 7: two column groups per wave
        if (g_math >= 6) g_math = 5;
        const int kmath = g_math == 5 ? 4 : g_math;     // arithmetic of the calls that have no record form
        if (g_math == 5) {
            CK(hipMalloc(&recx, objgan_h2_records_floats(sh.N, sh.Cin, (long)sh.H * sh.W) * 4));
            CK(hipMalloc(&recg, objgan_h2_records_floats(sh.N, sh.Cout, (long)OH * OW) * 4));
        }
        auto prep_x = [&]() {
            objgan_absmax_partials(dx, (long)nx, mxx, st);
            if (objgan_h2_records(dx, mxx, recx, sh.N, sh.Cin, (long)sh.H * sh.W, st) != 1) { fprintf(stderr, "records x\n"); exit(1); }
        };
        auto prep_g = [&]() {
            objgan_absmax_partials(dg, (long)ny, mxg, st);
            if (objgan_h2_records(dg, mxg, recg, sh.N, sh.Cout, (long)OH * OW, st) != 1) { fprintf(stderr, "records g\n"); exit(1); }
        };
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dg, hg.data(), ny * 4, hipMemcpyHostToDevice));
        const double flops = 2.0 * sh.N * OH * OW * (double)sh.Cout * sh.Cin * T;

        std::vector<int> dh(T), dwv(T), stp(T);
        for (int kh = 0; kh < sh.k; ++kh) for (int kw = 0; kw < sh.k; ++kw) {
            dh[kh * sh.k + kw] = kh - sh.p; dwv[kh * sh.k + kw] = kw - sh.p; stp[kh * sh.k + kw] = kh * sh.k + kw;
        }
        // math 5 only where the MFMA implicit-GEMM kernel runs (bank layout class 5); the thin / first-generation kernels keep math 4
        auto recm = [&](int C_, int H_, int W_, int M_, int Tg_, int PH_, int PW_) {
            if (g_math != 5) return g_math;
            return (objgan_conv_bank_layout(sh.N, C_, H_, W_, M_, Tg_, PH_, PW_, 0, 5) & 255) == 5 ? 5 : 4;
        };
        auto fwd = [&]() {
            const int fm = recm(sh.Cin, sh.H, sh.W, sh.Cout, T, OH, OW);
            if (g_math == 4) objgan_absmax_partials(dx, (long)nx, mxx, st);       // (the weight gradient reuses it)
            int rc = objgan_conv_igemm(fm == 5 ? recx : dx, dw, nullptr, dy, wt, sh.N, sh.Cin, sh.H, sh.W, sh.up, sh.refl, sh.Cout, sh.Cin, T, 0,
                                       T, dh.data(), dwv.data(), stp.data(), OH, OW, sh.s, OH, OW, 1, 1, 0, 0, 0, 0, 0, fm, nullptr, mxx, nullptr, wsb, nwsb, st);
            if (rc != 1) { fprintf(stderr, "fwd rc=%d\n", rc); exit(1); }
        };
        auto dgrad = [&]() {
            if (g_math == 4) objgan_absmax_partials(dg, (long)ny, mxg, st);       // (shared with the weight gradient)
            const float* dgk = dg;
            if (sh.s == 1) {
                const int pe = sh.refl ? 0 : sh.p;
                std::vector<int> h2(T), w2(T);
                for (int kh = 0; kh < sh.k; ++kh) for (int kw = 0; kw < sh.k; ++kw) { h2[kh * sh.k + kw] = pe - kh; w2[kh * sh.k + kw] = pe - kw; }
                const int dm = recm(sh.Cout, OH, OW, sh.Cin, T, TH, TW);
                dgk = dm == 5 ? recg : dg;
                int rc = objgan_conv_igemm(dgk, dw, nullptr, dgx, wt, sh.N, sh.Cout, OH, OW, 0, 0, sh.Cout, sh.Cin, T, 1,
                                           T, h2.data(), w2.data(), stp.data(), TH, TW, 1, TH, TW, 1, 1, 0, 0, 0, 0, 0, dm, nullptr, mxg, nullptr, wsb, nwsb, st);
                if (rc != 1) { fprintf(stderr, "dgrad rc=%d\n", rc); exit(1); }
            } else if (sh.k % 2 == 0 && LH % 2 == 0 && sh.Cin > 32) {
                std::vector<int> h2, w2, s2;
                for (int ph = 0; ph < 2; ++ph) for (int pw = 0; pw < 2; ++pw)
                    for (int kh = 0; kh < sh.k; ++kh) if (((ph + sh.p - kh) % 2 + 2) % 2 == 0)
                        for (int kw = 0; kw < sh.k; ++kw) if (((pw + sh.p - kw) % 2 + 2) % 2 == 0) {
                            h2.push_back((ph + sh.p - kh) / 2); w2.push_back((pw + sh.p - kw) / 2); s2.push_back(kh * sh.k + kw);
                        }
                dgk = g_math == 5 ? recg : dg;
                int rc = objgan_conv_dgrad_s2_phases(dgk, dw, dgx, wt, sh.N, sh.Cout, OH, OW, sh.Cin, T, (int)h2.size() / 4,
                                                     h2.data(), w2.data(), s2.data(), LH / 2, LW / 2, 0, g_math, mxg, wsb, nwsb, st);
                if (rc != 1) { fprintf(stderr, "dgrad phases rc=%d\n", rc); exit(1); }
            } else {
                CK(hipMemsetAsync(dgx, 0, ngx * 4, st));
                for (int ph = 0; ph < 2; ++ph) for (int pw = 0; pw < 2; ++pw) {
                    std::vector<int> h2, w2, s2;
                    for (int kh = 0; kh < sh.k; ++kh) if (((ph + sh.p - kh) % 2 + 2) % 2 == 0)
                        for (int kw = 0; kw < sh.k; ++kw) if (((pw + sh.p - kw) % 2 + 2) % 2 == 0) {
                            h2.push_back((ph + sh.p - kh) / 2); w2.push_back((pw + sh.p - kw) / 2); s2.push_back(kh * sh.k + kw);
                        }
                    const int PHg = (LH - ph + 1) / 2, PWg = (LW - pw + 1) / 2;
                    const int dm = recm(sh.Cout, OH, OW, sh.Cin, (int)h2.size(), PHg, PWg);
                    dgk = dm == 5 ? recg : dg;
                    int rc = objgan_conv_igemm(dgk, dw, nullptr, dgx, wt, sh.N, sh.Cout, OH, OW, 0, 0, sh.Cout, sh.Cin, T, 1,
                                               (int)h2.size(), h2.data(), w2.data(), s2.data(), PHg, PWg, 1, LH, LW, 2, 2, ph, pw, 0, 1, 0, dm, nullptr, mxg, nullptr, wsb, nwsb, st);
                    if (rc != 1) { fprintf(stderr, "dgrad2 rc=%d\n", rc); exit(1); }
                }
            }
        };
        auto wgrad = [&]() {
            // math 5: x as its record where the record form of the weight gradient takes the geometry
            const bool wrec = g_math == 5 && objgan_conv_wgrad_rec_ok(sh.N, sh.Cin, sh.H, sh.W, sh.Cout, OH, OW, sh.k);
            int rc = objgan_conv_wgrad(wrec ? recx : dx, dg, dgw, sh.N, sh.Cin, sh.H, sh.W, sh.up, sh.refl, sh.Cout, OH, OW, sh.k, sh.s, sh.p,
                                       wrec ? wmath : kmath, 0, mxx, mxg, wsb, nwsb, st);
            if (rc != 1) { fprintf(stderr, "wgrad rc=%d\n", rc); exit(1); }
        };
        auto timeit = [&](auto&& fn) {
            fn(); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) fn();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            return (double)ms / iters;
        };
        double tpx = 0, tpg = 0;
        if (g_math == 5) { tpx = timeit(prep_x); tpg = timeit(prep_g); }
        const double tf = timeit(fwd), td = timeit(dgrad), tw = timeit(wgrad);
        auto fnv = [](const float* v, size_t n) {
            unsigned long long h = 1469598103934665603ull;
            const unsigned* u = reinterpret_cast<const unsigned*>(v);
            for (size_t i = 0; i < n; ++i) { h ^= u[i]; h *= 1099511628211ull; }
            return h;
        };

        // ---- spot checks (double precision on the host)
        CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
        std::vector<float> hgw(nw);
        CK(hipMemcpy(hgw.data(), dgw, nw * 4, hipMemcpyDeviceToHost));
        double maxerr_f = 0, maxref_f = 0, se_f = 0, sr_f = 0;
        for (int smp = 0; smp < 256; ++smp) {
            g_seed = g_seed * 1664525u + 1013904223u; const int n = (g_seed >> 8) % sh.N;
            g_seed = g_seed * 1664525u + 1013904223u; const int co = (g_seed >> 8) % sh.Cout;
            g_seed = g_seed * 1664525u + 1013904223u; int oh = (g_seed >> 8) % OH;
            g_seed = g_seed * 1664525u + 1013904223u; int ow = (g_seed >> 8) % OW;
            if (smp < 8) { oh = (smp & 1) ? OH - 1 : 0; ow = (smp & 2) ? OW - 1 : 0; }
            double acc = 0;
            for (int ci = 0; ci < sh.Cin; ++ci) for (int kh = 0; kh < sh.k; ++kh) for (int kw = 0; kw < sh.k; ++kw) {
                int ih = oh * sh.s - sh.p + kh, iw = ow * sh.s - sh.p + kw;
                if (sh.refl) { ih = reflect(ih, LH); iw = reflect(iw, LW); }
                else if (ih < 0 || ih >= LH || iw < 0 || iw >= LW) continue;
                if (sh.up) { ih >>= 1; iw >>= 1; }
                acc += (double)hx[(((size_t)n * sh.Cin + ci) * sh.H + ih) * sh.W + iw] * hw[((size_t)co * sh.Cin + ci) * T + kh * sh.k + kw];
            }
            const double got = hy[(((size_t)n * sh.Cout + co) * OH + oh) * OW + ow];
            maxerr_f = fmax(maxerr_f, fabs(got - acc)); maxref_f = fmax(maxref_f, fabs(acc));
            se_f += (got - acc) * (got - acc); sr_f += acc * acc;
        }
        double maxerr_w = 0, maxref_w = 0, se_w = 0, sr_w = 0;
        for (int smp = 0; smp < 12; ++smp) {
            g_seed = g_seed * 1664525u + 1013904223u; const int co = (g_seed >> 8) % sh.Cout;
            g_seed = g_seed * 1664525u + 1013904223u; const int ci = (g_seed >> 8) % sh.Cin;
            g_seed = g_seed * 1664525u + 1013904223u; const int t = (g_seed >> 8) % T;
            const int kh = t / sh.k, kw = t % sh.k;
            double acc = 0;
            for (int n = 0; n < sh.N; ++n) for (int oh = 0; oh < OH; ++oh) for (int ow = 0; ow < OW; ++ow) {
                int ih = oh * sh.s - sh.p + kh, iw = ow * sh.s - sh.p + kw;
                if (sh.refl) { ih = reflect(ih, LH); iw = reflect(iw, LW); }
                else if (ih < 0 || ih >= LH || iw < 0 || iw >= LW) continue;
                if (sh.up) { ih >>= 1; iw >>= 1; }
                acc += (double)hx[(((size_t)n * sh.Cin + ci) * sh.H + ih) * sh.W + iw] * hg[(((size_t)n * sh.Cout + co) * OH + oh) * OW + ow];
            }
            const double got = hgw[((size_t)co * sh.Cin + ci) * T + t];
            maxerr_w = fmax(maxerr_w, fabs(got - acc)); maxref_w = fmax(maxref_w, fabs(acc));
            se_w += (got - acc) * (got - acc); sr_w += acc * acc;
        }
        double maxerr_d = 0, maxref_d = 0, se_d = 0, sr_d = 0;
        {
            std::vector<float> hgx(ngx);
            CK(hipMemcpy(hgx.data(), dgx, ngx * 4, hipMemcpyDeviceToHost));
            for (int smp = 0; smp < 128; ++smp) {
                g_seed = g_seed * 1664525u + 1013904223u; const int n = (g_seed >> 8) % sh.N;
                g_seed = g_seed * 1664525u + 1013904223u; const int ci = (g_seed >> 8) % sh.Cin;
                g_seed = g_seed * 1664525u + 1013904223u; int ih = (g_seed >> 8) % TH;
                g_seed = g_seed * 1664525u + 1013904223u; int iw = (g_seed >> 8) % TW;
                if (smp < 4) { ih = (smp & 1) ? TH - 1 : 0; iw = (smp & 2) ? TW - 1 : 0; }
                // dgx is w.r.t. the (reflect-)padded / upsampled logical input [TH x TW]
                const int pe = sh.refl ? 0 : sh.p;
                double acc = 0;
                for (int co = 0; co < sh.Cout; ++co) for (int kh = 0; kh < sh.k; ++kh) for (int kw = 0; kw < sh.k; ++kw) {
                    const int nh = ih + pe - kh, nw_ = iw + pe - kw;
                    if (nh < 0 || nw_ < 0 || nh % sh.s || nw_ % sh.s) continue;
                    const int oh = nh / sh.s, ow = nw_ / sh.s;
                    if (oh >= OH || ow >= OW) continue;
                    acc += (double)hg[(((size_t)n * sh.Cout + co) * OH + oh) * OW + ow] * hw[((size_t)co * sh.Cin + ci) * T + kh * sh.k + kw];
                }
                const double got = hgx[(((size_t)n * sh.Cin + ci) * TH + ih) * TW + iw];
                maxerr_d = fmax(maxerr_d, fabs(got - acc)); maxref_d = fmax(maxref_d, fabs(acc));
                se_d += (got - acc) * (got - acc); sr_d += acc * acc;
            }
        }
        // err: max |got - fp64| / max |fp64| over the samples; l2: sqrt(sum (got - fp64)^2 / sum fp64^2)
        printf("%-32s fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF | err f %.1e d %.1e w %.1e | l2 f %.2e d %.2e w %.2e\n",
               sh.name, tf, flops / tf / 1e9, td, flops / td / 1e9, tw, flops / tw / 1e9,
               maxerr_f / (maxref_f + 1e-30), maxerr_d / (maxref_d + 1e-30), maxerr_w / (maxref_w + 1e-30),
               sqrt(se_f / (sr_f + 1e-300)), sqrt(se_d / (sr_d + 1e-300)), sqrt(se_w / (sr_w + 1e-300)));
        {
            std::vector<float> hgx2(ngx);
            CK(hipMemcpy(hgx2.data(), dgx, ngx * 4, hipMemcpyDeviceToHost));
            printf("    hash fwd %016llx dgrad %016llx wgrad %016llx", fnv(hy.data(), ny), fnv(hgx2.data(), ngx), fnv(hgw.data(), nw));
            if (g_math == 5) printf("  | record passes (absmax + split): x %.3f ms  dy %.3f ms", tpx, tpg);
            printf("\n");
        }
        fflush(stdout);
        if (recx) hipFree(recx);
        if (recg) hipFree(recg);
        hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dg); hipFree(dgx); hipFree(dgw); hipFree(wt); hipFree(wsb); hipFree(mxx); hipFree(mxg);
    }
    return 0;
}
