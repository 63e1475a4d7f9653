// Normalisation + gated/leaky activations, forward and backward, for gfx950.
//
// Covers, as fused HBM-bound passes, what the reference expresses as separate
// nn.BatchNorm2d / nn.BatchNorm1d (train mode, batch statistics) / nn.InstanceNorm2d
// (affine=False) modules followed by GLU / LeakyReLU(0.2) / a residual add:
//   reference image_generation/model.py:19-27 (GLU), :43-49 (upBlock: conv -> BN -> GLU),
//   :63-81 (HmapResBlock: conv -> IN -> GLU -> conv -> IN -> +x), :486-518 (fc -> BN1d -> GLU),
//   :589-617 / :1119-1123 (conv -> IN -> LeakyReLU), :989-995, :1008-1012 (conv -> BN -> LeakyReLU).
//
// A "group" is the set of elements that share statistics: one (n, c) plane for
// InstanceNorm, one channel c across the batch for BatchNorm.  Statistics are accumulated
// as shifted sums  sum(x - K), sum((x - K)^2)  with K = first element of the group (kills the
// cancellation of the naive E[x^2] - E[x]^2 form), wave-shuffle reduced, one atomic pair
// per workgroup.  The apply pass normalises, applies gamma/beta, the activation and the
// optional residual in one read + one write; GLU reads both gate halves and writes C/2
// channels, so the 2C-channel normalised tensor never exists in HBM.
#include "common.h"

#define OG_NORM_NONE 0
#define OG_NORM_LRELU 1
#define OG_NORM_GLU 2

struct NormGeom {
    int N, C, HW;
    int per_channel;   // 1 = BatchNorm (group = c), 0 = InstanceNorm (group = n*C + c)
};

__device__ __forceinline__ float og_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// Statistics workspace of one call: [2G totals][2 * G * P partial pairs].  Every workgroup of a statistics launch
// stores the partial sums of its slice into its own slot; norm_partials_sum_kernel (one thread per group, launched
// where round 2 had the memset of the atomics' target) adds the P slots of a group in slot order: bit-reproducible
// statistics.  (A single-launch form -- ticket + last arriver sums -- was measured first: the agent-scope release
// fence every workgroup needs writes the XCD's L2 back, 150-200 us per launch against 25.)
struct NormWs { float* sums; float* part; int P; };
__device__ __forceinline__ void og_store_partial2(const NormWs& ws, int g, int idx, float s1, float s2) {
    if (ws.P == 1) { ws.sums[2 * g] = s1; ws.sums[2 * g + 1] = s2; return; }
    float* slot = ws.part + ((size_t)g * ws.P + idx) * 2;
    slot[0] = s1; slot[1] = s2;
}
// (amax != nullptr: also zeroes the OG_AMAX_SLOTS maxima slots the apply kernel behind it fills with og_amax_atomic)
__global__ __launch_bounds__(256) void norm_partials_sum_kernel(NormWs ws, int G, float* __restrict__ amax) {
    if (amax && blockIdx.x == 0)
        for (int k = threadIdx.x; k < OG_AMAX_SLOTS; k += 256) amax[k] = 0.f;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float* p = ws.part + (size_t)g * ws.P * 2;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < ws.P; ++k) { a += p[2 * k]; b += p[2 * k + 1]; }
    ws.sums[2 * g] = a; ws.sums[2 * g + 1] = b;
}

// ---- statistics ------------------------------------------------------------------------
// grid = (G, S).  Every workgroup stores its partial pair into its own slot; norm_partials_sum_kernel adds the slots in
// order (sums is fully written there: nothing to zero, no atomics).
__global__ __launch_bounds__(256) void norm_stats_kernel(const float* __restrict__ x,
                                                         NormWs ws, NormGeom gm) {
    __shared__ float red[16];
    const int g = blockIdx.x;
    const int planes = gm.per_channel ? gm.N : 1;
    const size_t base = gm.per_channel ? (size_t)g * gm.HW : (size_t)g * gm.HW;
    const size_t pstride = (size_t)gm.C * gm.HW;
    const long total = (long)planes * gm.HW;
    const float K = x[base];
    const long chunk = (total + gridDim.y - 1) / gridDim.y;
    const long e0 = (long)blockIdx.y * chunk;
    const long e1 = min(total, e0 + chunk);
    float s1 = 0.f, s2 = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const long p = e / gm.HW;
        const long i = e - p * gm.HW;
        const float d = x[base + p * pstride + i] - K;
        s1 += d;
        s2 += d * d;
    }
    s1 = og_block_sum(s1, red);
    s2 = og_block_sum(s2, red);
    if (threadIdx.x == 0) og_store_partial2(ws, g, blockIdx.y, s1, s2);
}

// mean/rstd per group; optional BatchNorm running-statistics update (momentum, unbiased var)
__global__ __launch_bounds__(256) void norm_finalize_kernel(
    const float* __restrict__ x, const float* __restrict__ sums, float* __restrict__ mean,
    float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var,
    NormGeom gm, int G, float eps, float momentum) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float cnt = gm.per_channel ? (float)gm.N * gm.HW : (float)gm.HW;
    const float K = x[(size_t)g * gm.HW];
    const float m1 = sums[2 * g] / cnt;
    float var = sums[2 * g + 1] / cnt - m1 * m1;
    var = fmaxf(var, 0.f);
    const float mu = K + m1;
    mean[g] = mu;
    rstd[g] = 1.0f / sqrtf(var + eps);
    if (running_mean) {
        const float unbiased = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
        running_mean[g] = (1.f - momentum) * running_mean[g] + momentum * mu;
        running_var[g] = (1.f - momentum) * running_var[g] + momentum * unbiased;
    }
}

// ---- forward apply ---------------------------------------------------------------------
// out channels: C (NONE / LRELU) or C/2 (GLU).  residual (same shape as out) optional.
__global__ __launch_bounds__(256) void norm_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ residual, float* __restrict__ y, NormGeom gm, int mode) {
    const int Co = mode == OG_NORM_GLU ? gm.C / 2 : gm.C;
    const long total = (long)gm.N * Co * gm.HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const long plane = e / gm.HW;          // n*Co + c
        const int i = (int)(e - plane * gm.HW);
        const int n = (int)(plane / Co);
        const int c = (int)(plane - (long)n * Co);
        const int ga = gm.per_channel ? c : n * gm.C + c;
        const size_t xa = ((size_t)n * gm.C + c) * gm.HW + i;
        float za = (x[xa] - mean[ga]) * rstd[ga];
        if (gamma) za = za * gamma[c] + beta[c];
        float v;
        if (mode == OG_NORM_GLU) {
            const int cb = c + Co;
            const int gb = gm.per_channel ? cb : n * gm.C + cb;
            float zb = (x[xa + (size_t)Co * gm.HW] - mean[gb]) * rstd[gb];
            if (gamma) zb = zb * gamma[cb] + beta[cb];
            v = za * og_sigmoid(zb);
        } else if (mode == OG_NORM_LRELU) {
            v = za > 0.f ? za : 0.2f * za;
        } else {
            v = za;
        }
        if (residual) v += residual[e];
        y[e] = v;
    }
}

// ---- backward --------------------------------------------------------------------------
// dz (gradient w.r.t. the normalised+affine value z) recomputed from x, dy and the mode.
__device__ __forceinline__ float norm_dz(const float* __restrict__ x, const float* __restrict__ dy,
                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         const NormGeom& gm, int mode, int n, int c, int i,
                                         float& xhat_out) {
    const int g = gm.per_channel ? c : n * gm.C + c;
    const size_t xa = ((size_t)n * gm.C + c) * gm.HW + i;
    const float xhat = (x[xa] - mean[g]) * rstd[g];
    xhat_out = xhat;
    if (mode == OG_NORM_NONE) return dy[xa];
    float z = xhat;
    if (gamma) z = z * gamma[c] + beta[c];
    if (mode == OG_NORM_LRELU) return dy[xa] * (z > 0.f ? 1.f : 0.2f);
    // GLU: channels [0, C/2) are the value half, [C/2, C) the gate half
    const int Ch = gm.C / 2;
    const bool is_gate = c >= Ch;
    const int co = is_gate ? c - Ch : c;
    const int cp = is_gate ? c - Ch : c + Ch;     // partner channel
    const int gp = gm.per_channel ? cp : n * gm.C + cp;
    const size_t xp = ((size_t)n * gm.C + cp) * gm.HW + i;
    float zp = (x[xp] - mean[gp]) * rstd[gp];
    if (gamma) zp = zp * gamma[cp] + beta[cp];
    const float d = dy[((size_t)n * Ch + co) * gm.HW + i];
    if (!is_gate) return d * og_sigmoid(zp);
    const float sg = og_sigmoid(z);
    return d * zp * sg * (1.f - sg);
}

// grid = (G, S): bsums[g*2] += sum dz, bsums[g*2+1] += sum dz*xhat   (zero on entry)
__global__ __launch_bounds__(256) void norm_bwd_stats_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    NormWs ws, NormGeom gm, int mode) {
    __shared__ float red[16];
    const int g = blockIdx.x;
    const int planes = gm.per_channel ? gm.N : 1;
    const long total = (long)planes * gm.HW;
    const long chunk = (total + gridDim.y - 1) / gridDim.y;
    const long e0 = (long)blockIdx.y * chunk;
    const long e1 = min(total, e0 + chunk);
    float s1 = 0.f, s2 = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const int p = (int)(e / gm.HW);
        const int i = (int)(e - (long)p * gm.HW);
        const int n = gm.per_channel ? p : g / gm.C;
        const int c = gm.per_channel ? g : g - n * gm.C;
        float xhat;
        const float dz = norm_dz(x, dy, mean, rstd, gamma, beta, gm, mode, n, c, i, xhat);
        s1 += dz;
        s2 += dz * xhat;
    }
    s1 = og_block_sum(s1, red);
    s2 = og_block_sum(s2, red);
    if (threadIdx.x == 0) og_store_partial2(ws, g, blockIdx.y, s1, s2);
}

// dx = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat))
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ bsums, float* __restrict__ dx, NormGeom gm, int mode) {
    const long total = (long)gm.N * gm.C * gm.HW;
    const float inv_cnt = 1.0f / (gm.per_channel ? (float)gm.N * gm.HW : (float)gm.HW);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const long plane = e / gm.HW;
        const int i = (int)(e - plane * gm.HW);
        const int n = (int)(plane / gm.C);
        const int c = (int)(plane - (long)n * gm.C);
        const int g = gm.per_channel ? c : n * gm.C + c;
        float xhat;
        const float dz = norm_dz(x, dy, mean, rstd, gamma, beta, gm, mode, n, c, i, xhat);
        const float gw = gamma ? gamma[c] : 1.f;
        dx[e] = gw * rstd[g] * (dz - bsums[2 * g] * inv_cnt - xhat * bsums[2 * g + 1] * inv_cnt);
    }
}

// ---- plane-structured variants (HW % 4 == 0, HW >= 256) ----------------------------------------
// One workgroup = one (n, c) plane (or a chunk of it): the group / channel constants are scalars,
// the element loop is pure 16-byte streaming -- no per-element index arithmetic.  GLU handles the
// value and the gate plane of an output channel together, so x and dy are read exactly once.
#define OG_NORM_CHUNK 8192          // elements of one plane per workgroup

__device__ __forceinline__ float og_group_mean(const float* p, int g) { return p[g]; }

// grid = (N*C, chunks): sums[g*2 + {0,1}] += shifted sums of this plane chunk (zero on entry)
__global__ __launch_bounds__(256) void norm_stats_plane_kernel(const float* __restrict__ x,
                                                               NormWs ws, NormGeom gm) {
    __shared__ float red[16];
    const int plane = blockIdx.x;                       // n*C + c
    const int c = plane % gm.C;
    const int g = gm.per_channel ? c : plane;
    const float K = x[(size_t)g * gm.HW];               // first element of the group (n = 0 plane)
    const float4* xp = reinterpret_cast<const float4*>(x + (size_t)plane * gm.HW);
    const int i0 = blockIdx.y * (OG_NORM_CHUNK / 4);
    const int i1 = min(gm.HW / 4, i0 + OG_NORM_CHUNK / 4);
    float s1 = 0.f, s2 = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float4 v = xp[i];
        const float a = v.x - K, b = v.y - K, cc = v.z - K, d = v.w - K;
        s1 += (a + b) + (cc + d);
        s2 += (a * a + b * b) + (cc * cc + d * d);
    }
    s1 = og_block_sum(s1, red);
    s2 = og_block_sum(s2, red);
    // slot of this workgroup within its group: BatchNorm groups collect the N images x chunks, InstanceNorm the chunks
    const int idx = gm.per_channel ? (plane / gm.C) * (int)gridDim.y + (int)blockIdx.y : (int)blockIdx.y;
    if (threadIdx.x == 0) og_store_partial2(ws, g, idx, s1, s2);
}

// Statistics of group g from its shifted sums (what norm_finalize_kernel computes); when `fin.sums` is set
// the apply kernel does it itself and the workgroup of image 0 / chunk 0 publishes mean, rstd and the
// BatchNorm running statistics -- one launch less per layer.
struct NormFin { const float* sums; float* mean_out; float* rstd_out; float* running_mean; float* running_var;
                 float eps, momentum; };

__device__ __forceinline__ void og_group_stats(const float* __restrict__ x, const NormFin& fin, const NormGeom& gm,
                                               int g, bool publish, float& mu, float& rs) {
    const float cnt = gm.per_channel ? (float)gm.N * gm.HW : (float)gm.HW;
    const float K = x[(size_t)g * gm.HW];
    const float m1 = fin.sums[2 * g] / cnt;
    float var = fin.sums[2 * g + 1] / cnt - m1 * m1;
    var = fmaxf(var, 0.f);
    mu = K + m1;
    rs = 1.0f / sqrtf(var + fin.eps);
    if (publish && threadIdx.x == 0) {
        fin.mean_out[g] = mu;
        fin.rstd_out[g] = rs;
        if (fin.running_mean) {
            const float unbiased = cnt > 1.f ? var * cnt / (cnt - 1.f) : var;
            fin.running_mean[g] = (1.f - fin.momentum) * fin.running_mean[g] + fin.momentum * mu;
            fin.running_var[g] = (1.f - fin.momentum) * fin.running_var[g] + fin.momentum * unbiased;
        }
    }
}

// grid = (N*Co, chunks)
template <int MODE>
__global__ __launch_bounds__(256) void norm_apply_plane_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ residual, float* __restrict__ y, NormGeom gm, NormFin fin, float* __restrict__ amax) {
    __shared__ float amax_red[4];
    float vmax = 0.f;
    const int Co = MODE == OG_NORM_GLU ? gm.C / 2 : gm.C;
    const int plane = blockIdx.x;                       // n*Co + c
    const int n = plane / Co;
    const int c = plane - n * Co;
    const int ga = gm.per_channel ? c : n * gm.C + c;
    const bool publish = blockIdx.y == 0 && (!gm.per_channel || n == 0);
    float mua, rsa;
    if (fin.sums) og_group_stats(x, fin, gm, ga, publish, mua, rsa); else { mua = mean[ga]; rsa = rstd[ga]; }
    float sa = rsa, ta = -mua * sa;                     // z = x*sa + ta
    if (gamma) { ta = ta * gamma[c] + beta[c]; sa *= gamma[c]; }
    float sb = 0.f, tb = 0.f;
    if (MODE == OG_NORM_GLU) {
        const int cb = c + Co;
        const int gb = gm.per_channel ? cb : n * gm.C + cb;
        float mub, rsb;
        if (fin.sums) og_group_stats(x, fin, gm, gb, publish, mub, rsb); else { mub = mean[gb]; rsb = rstd[gb]; }
        sb = rsb; tb = -mub * sb;
        if (gamma) { tb = tb * gamma[cb] + beta[cb]; sb *= gamma[cb]; }
    }
    const float4* xa = reinterpret_cast<const float4*>(x + ((size_t)n * gm.C + c) * gm.HW);
    const float4* xb = reinterpret_cast<const float4*>(x + ((size_t)n * gm.C + c + Co) * gm.HW);
    const float4* rp = residual ? reinterpret_cast<const float4*>(residual + (size_t)plane * gm.HW) : nullptr;
    float4* yp = reinterpret_cast<float4*>(y + (size_t)plane * gm.HW);
    const int i0 = blockIdx.y * (OG_NORM_CHUNK / 4);
    const int i1 = min(gm.HW / 4, i0 + OG_NORM_CHUNK / 4);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float4 a = xa[i];
        float z[4] = {a.x * sa + ta, a.y * sa + ta, a.z * sa + ta, a.w * sa + ta};
        float v[4];
        if (MODE == OG_NORM_GLU) {
            const float4 b = xb[i];
            const float zb[4] = {b.x * sb + tb, b.y * sb + tb, b.z * sb + tb, b.w * sb + tb};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[j] * og_sigmoid(zb[j]);
        } else if (MODE == OG_NORM_LRELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[j] > 0.f ? z[j] : 0.2f * z[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[j];
        }
        if (rp) { const float4 r = rp[i]; v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        yp[i] = make_float4(v[0], v[1], v[2], v[3]);
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (amax) og_amax_atomic(og_block_max(vmax, amax_red), amax, blockIdx.x + blockIdx.y * gridDim.x);
}

// Backward, shared element math: from x (both halves for GLU) and dy produce dz and xhat of the
// value channel (a) and, for GLU, of the gate channel (b).
template <int MODE>
__device__ __forceinline__ void norm_dz_pair(float xa, float xb, float d, float ma, float ra, float mb, float rb,
                                             float gaa, float baa, float gab, float bab, bool affine,
                                             float& dza, float& xha, float& dzb, float& xhb) {
    xha = (xa - ma) * ra;
    float za = affine ? xha * gaa + baa : xha;
    if (MODE == OG_NORM_NONE) { dza = d; dzb = 0.f; xhb = 0.f; return; }
    if (MODE == OG_NORM_LRELU) { dza = d * (za > 0.f ? 1.f : 0.2f); dzb = 0.f; xhb = 0.f; return; }
    xhb = (xb - mb) * rb;
    const float zb = affine ? xhb * gab + bab : xhb;
    const float sg = og_sigmoid(zb);
    dza = d * sg;
    dzb = d * za * sg * (1.f - sg);
}

// grid = (N*Co, chunks); bsums zero on entry
template <int MODE>
__global__ __launch_bounds__(256) void norm_bwd_stats_plane_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    NormWs ws, NormGeom gm) {
    __shared__ float red[16];
    const int Co = MODE == OG_NORM_GLU ? gm.C / 2 : gm.C;
    const int plane = blockIdx.x;
    const int n = plane / Co;
    const int c = plane - n * Co;
    const int cb = c + Co;
    const int ga = gm.per_channel ? c : n * gm.C + c;
    const int gb = gm.per_channel ? cb : n * gm.C + cb;
    const bool affine = gamma != nullptr;
    const float ma = mean[ga], ra = rstd[ga];
    const float mb = MODE == OG_NORM_GLU ? mean[gb] : 0.f, rb = MODE == OG_NORM_GLU ? rstd[gb] : 0.f;
    const float gaa = affine ? gamma[c] : 1.f, baa = affine ? beta[c] : 0.f;
    const float gab = (affine && MODE == OG_NORM_GLU) ? gamma[cb] : 1.f, bab = (affine && MODE == OG_NORM_GLU) ? beta[cb] : 0.f;
    const float4* xa = reinterpret_cast<const float4*>(x + ((size_t)n * gm.C + c) * gm.HW);
    const float4* xb = reinterpret_cast<const float4*>(x + ((size_t)n * gm.C + cb) * gm.HW);
    const float4* dp = reinterpret_cast<const float4*>(dy + (size_t)plane * gm.HW);
    const int i0 = blockIdx.y * (OG_NORM_CHUNK / 4);
    const int i1 = min(gm.HW / 4, i0 + OG_NORM_CHUNK / 4);
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float4 va = xa[i];
        const float4 vb = MODE == OG_NORM_GLU ? xb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vd = dp[i];
        const float xs[4] = {va.x, va.y, va.z, va.w}, ys[4] = {vb.x, vb.y, vb.z, vb.w}, ds[4] = {vd.x, vd.y, vd.z, vd.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dza, xha, dzb, xhb;
            norm_dz_pair<MODE>(xs[j], ys[j], ds[j], ma, ra, mb, rb, gaa, baa, gab, bab, affine, dza, xha, dzb, xhb);
            a1 += dza; a2 += dza * xha;
            if (MODE == OG_NORM_GLU) { b1 += dzb; b2 += dzb * xhb; }
        }
    }
    a1 = og_block_sum(a1, red);
    a2 = og_block_sum(a2, red);
    if (MODE == OG_NORM_GLU) { b1 = og_block_sum(b1, red); b2 = og_block_sum(b2, red); }
    const int idx = gm.per_channel ? n * (int)gridDim.y + (int)blockIdx.y : (int)blockIdx.y;
    if (threadIdx.x == 0) {
        og_store_partial2(ws, ga, idx, a1, a2);
        if (MODE == OG_NORM_GLU) og_store_partial2(ws, gb, idx, b1, b2);
    }
}

// grid = (N*Co, chunks): dx = gamma * rstd * (dz - mean(dz) - xhat * mean(dz * xhat))
template <int MODE>
__global__ __launch_bounds__(256) void norm_bwd_apply_plane_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ bsums, float* __restrict__ dx, NormGeom gm,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ amax) {
    __shared__ float amax_red[4];
    float vmax = 0.f;
    const int Co = MODE == OG_NORM_GLU ? gm.C / 2 : gm.C;
    const int plane = blockIdx.x;
    const int n = plane / Co;
    const int c = plane - n * Co;
    const int cb = c + Co;
    const int ga = gm.per_channel ? c : n * gm.C + c;
    const int gb = gm.per_channel ? cb : n * gm.C + cb;
    const bool affine = gamma != nullptr;
    const float inv_cnt = 1.0f / (gm.per_channel ? (float)gm.N * gm.HW : (float)gm.HW);
    // BatchNorm parameter gradients (dgamma = sum dz*xhat, dbeta = sum dz): published by image 0 / chunk 0
    if (dgamma && gm.per_channel && n == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        dbeta[c] = bsums[2 * c]; dgamma[c] = bsums[2 * c + 1];
        if (MODE == OG_NORM_GLU) { dbeta[cb] = bsums[2 * cb]; dgamma[cb] = bsums[2 * cb + 1]; }
    }
    const float ma = mean[ga], ra = rstd[ga];
    const float mb = MODE == OG_NORM_GLU ? mean[gb] : 0.f, rb = MODE == OG_NORM_GLU ? rstd[gb] : 0.f;
    const float gaa = affine ? gamma[c] : 1.f, baa = affine ? beta[c] : 0.f;
    const float gab = (affine && MODE == OG_NORM_GLU) ? gamma[cb] : 1.f, bab = (affine && MODE == OG_NORM_GLU) ? beta[cb] : 0.f;
    const float ka1 = bsums[2 * ga] * inv_cnt, ka2 = bsums[2 * ga + 1] * inv_cnt;
    const float kb1 = MODE == OG_NORM_GLU ? bsums[2 * gb] * inv_cnt : 0.f, kb2 = MODE == OG_NORM_GLU ? bsums[2 * gb + 1] * inv_cnt : 0.f;
    const float wa = gaa * ra, wb = gab * rb;
    const float4* xa = reinterpret_cast<const float4*>(x + ((size_t)n * gm.C + c) * gm.HW);
    const float4* xb = reinterpret_cast<const float4*>(x + ((size_t)n * gm.C + cb) * gm.HW);
    const float4* dp = reinterpret_cast<const float4*>(dy + (size_t)plane * gm.HW);
    float4* oa = reinterpret_cast<float4*>(dx + ((size_t)n * gm.C + c) * gm.HW);
    float4* ob = reinterpret_cast<float4*>(dx + ((size_t)n * gm.C + cb) * gm.HW);
    const int i0 = blockIdx.y * (OG_NORM_CHUNK / 4);
    const int i1 = min(gm.HW / 4, i0 + OG_NORM_CHUNK / 4);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float4 va = xa[i];
        const float4 vb = MODE == OG_NORM_GLU ? xb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vd = dp[i];
        const float xs[4] = {va.x, va.y, va.z, va.w}, ys[4] = {vb.x, vb.y, vb.z, vb.w}, ds[4] = {vd.x, vd.y, vd.z, vd.w};
        float ra4[4], rb4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dza, xha, dzb, xhb;
            norm_dz_pair<MODE>(xs[j], ys[j], ds[j], ma, ra, mb, rb, gaa, baa, gab, bab, affine, dza, xha, dzb, xhb);
            ra4[j] = wa * (dza - ka1 - xha * ka2);
            rb4[j] = wb * (dzb - kb1 - xhb * kb2);
        }
        oa[i] = make_float4(ra4[0], ra4[1], ra4[2], ra4[3]);
        if (MODE == OG_NORM_GLU) ob[i] = make_float4(rb4[0], rb4[1], rb4[2], rb4[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fmaxf(fabsf(ra4[j]), MODE == OG_NORM_GLU ? fabsf(rb4[j]) : 0.f));
    }
    if (amax) og_amax_atomic(og_block_max(vmax, amax_red), amax, blockIdx.x + blockIdx.y * gridDim.x);
}


// ---- InstanceNorm, one kernel per direction -------------------------------------------------------------
// The statistics of an InstanceNorm group are those of ONE plane, so a workgroup can own its plane(s) from
// the first read to the last write: pass 1 accumulates the shifted sums (block reduction, no atomics, no
// zeroed workspace), pass 2 re-reads the plane -- 4 .. 64 KB that the workgroup itself has just pulled
// through L2 / the Infinity Cache -- and applies.  Against the three-kernel path (memset + statistics +
// finalize + apply) this is one launch instead of four and one HBM read of x instead of two.
// grid = N*Co workgroups (GLU: the value and the gate plane of an output channel together).
__device__ __forceinline__ void og_block_sum2(float& a, float& b, float* red) {
    a = og_wave_sum(a);
    b = og_wave_sum(b);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[wid] = a; red[4 + wid] = b; }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
}

template <int MODE>
__global__ __launch_bounds__(256) void in_fwd_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ residual, float* __restrict__ y,
    float* __restrict__ mean, float* __restrict__ rstd, NormGeom gm, float eps, float* __restrict__ amax) {
    __shared__ float red[8];
    float vmax = 0.f;
    const int Co = MODE == OG_NORM_GLU ? gm.C / 2 : gm.C;
    const int plane = blockIdx.x;                       // n*Co + c
    const int n = plane / Co;
    const int c = plane - n * Co;
    const int ga = n * gm.C + c, gb = ga + Co;
    const float4* xa = reinterpret_cast<const float4*>(x + (size_t)ga * gm.HW);
    const float4* xb = reinterpret_cast<const float4*>(x + (size_t)gb * gm.HW);
    const int n4 = gm.HW / 4;
    const float inv = 1.0f / (float)gm.HW;
    const float Ka = x[(size_t)ga * gm.HW];
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = xa[i];
        const float a = v.x - Ka, b = v.y - Ka, cc = v.z - Ka, d = v.w - Ka;
        s1 += (a + b) + (cc + d);
        s2 += (a * a + b * b) + (cc * cc + d * d);
    }
    og_block_sum2(s1, s2, red);
    float m1 = s1 * inv;
    float var = fmaxf(s2 * inv - m1 * m1, 0.f);
    const float mua = Ka + m1, ra = 1.0f / sqrtf(var + eps);
    float mub = 0.f, rb = 0.f;
    if (MODE == OG_NORM_GLU) {
        const float Kb = x[(size_t)gb * gm.HW];
        s1 = 0.f; s2 = 0.f;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 v = xb[i];
            const float a = v.x - Kb, b = v.y - Kb, cc = v.z - Kb, d = v.w - Kb;
            s1 += (a + b) + (cc + d);
            s2 += (a * a + b * b) + (cc * cc + d * d);
        }
        og_block_sum2(s1, s2, red);
        m1 = s1 * inv;
        var = fmaxf(s2 * inv - m1 * m1, 0.f);
        mub = Kb + m1; rb = 1.0f / sqrtf(var + eps);
    }
    if (threadIdx.x == 0) {
        mean[ga] = mua; rstd[ga] = ra;
        if (MODE == OG_NORM_GLU) { mean[gb] = mub; rstd[gb] = rb; }
    }
    const float sa = ra, ta = -mua * ra, sb = rb, tb = -mub * rb;      // z = x*s + t
    const float4* rp = residual ? reinterpret_cast<const float4*>(residual + (size_t)plane * gm.HW) : nullptr;
    float4* yp = reinterpret_cast<float4*>(y + (size_t)plane * gm.HW);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 a = xa[i];
        const float z[4] = {a.x * sa + ta, a.y * sa + ta, a.z * sa + ta, a.w * sa + ta};
        float v[4];
        if (MODE == OG_NORM_GLU) {
            const float4 b = xb[i];
            const float zb[4] = {b.x * sb + tb, b.y * sb + tb, b.z * sb + tb, b.w * sb + tb};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[j] * og_sigmoid(zb[j]);
        } else if (MODE == OG_NORM_LRELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[j] > 0.f ? z[j] : 0.2f * z[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = z[j];
        }
        if (rp) { const float4 r = rp[i]; v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w; }
        yp[i] = make_float4(v[0], v[1], v[2], v[3]);
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (amax) og_amax_atomic(og_block_max(vmax, red), amax, blockIdx.x);      // (slots zeroed by the caller)
}

template <int MODE>
__global__ __launch_bounds__(256) void in_bwd_fused_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, NormGeom gm, float* __restrict__ amax) {
    __shared__ float red[8];
    float vmax = 0.f;
    const int Co = MODE == OG_NORM_GLU ? gm.C / 2 : gm.C;
    const int plane = blockIdx.x;
    const int n = plane / Co;
    const int c = plane - n * Co;
    const int ga = n * gm.C + c, gb = ga + Co;
    const float inv_cnt = 1.0f / (float)gm.HW;
    const float ma = mean[ga], ra = rstd[ga];
    const float mb = MODE == OG_NORM_GLU ? mean[gb] : 0.f, rb = MODE == OG_NORM_GLU ? rstd[gb] : 0.f;
    const float4* xa = reinterpret_cast<const float4*>(x + (size_t)ga * gm.HW);
    const float4* xb = reinterpret_cast<const float4*>(x + (size_t)gb * gm.HW);
    const float4* dp = reinterpret_cast<const float4*>(dy + (size_t)plane * gm.HW);
    float4* oa = reinterpret_cast<float4*>(dx + (size_t)ga * gm.HW);
    float4* ob = reinterpret_cast<float4*>(dx + (size_t)gb * gm.HW);
    const int n4 = gm.HW / 4;
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 va = xa[i];
        const float4 vb = MODE == OG_NORM_GLU ? xb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vd = dp[i];
        const float xs[4] = {va.x, va.y, va.z, va.w}, ys[4] = {vb.x, vb.y, vb.z, vb.w}, ds[4] = {vd.x, vd.y, vd.z, vd.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dza, xha, dzb, xhb;
            norm_dz_pair<MODE>(xs[j], ys[j], ds[j], ma, ra, mb, rb, 1.f, 0.f, 1.f, 0.f, false, dza, xha, dzb, xhb);
            a1 += dza; a2 += dza * xha;
            if (MODE == OG_NORM_GLU) { b1 += dzb; b2 += dzb * xhb; }
        }
    }
    og_block_sum2(a1, a2, red);
    if (MODE == OG_NORM_GLU) og_block_sum2(b1, b2, red);
    const float ka1 = a1 * inv_cnt, ka2 = a2 * inv_cnt, kb1 = b1 * inv_cnt, kb2 = b2 * inv_cnt;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 va = xa[i];
        const float4 vb = MODE == OG_NORM_GLU ? xb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vd = dp[i];
        const float xs[4] = {va.x, va.y, va.z, va.w}, ys[4] = {vb.x, vb.y, vb.z, vb.w}, ds[4] = {vd.x, vd.y, vd.z, vd.w};
        float ra4[4], rb4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float dza, xha, dzb, xhb;
            norm_dz_pair<MODE>(xs[j], ys[j], ds[j], ma, ra, mb, rb, 1.f, 0.f, 1.f, 0.f, false, dza, xha, dzb, xhb);
            ra4[j] = ra * (dza - ka1 - xha * ka2);
            rb4[j] = rb * (dzb - kb1 - xhb * kb2);
        }
        oa[i] = make_float4(ra4[0], ra4[1], ra4[2], ra4[3]);
        if (MODE == OG_NORM_GLU) ob[i] = make_float4(rb4[0], rb4[1], rb4[2], rb4[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fmaxf(fabsf(ra4[j]), MODE == OG_NORM_GLU ? fabsf(rb4[j]) : 0.f));
    }
    if (amax) og_amax_atomic(og_block_max(vmax, red), amax, blockIdx.x);      // (slots zeroed by the caller)
}

// dgamma[c] = bsums[2c+1], dbeta[c] = bsums[2c]   (BatchNorm only)
__global__ void norm_affine_grad_kernel(const float* __restrict__ bsums, float* __restrict__ dgamma,
                                        float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { dbeta[c] = bsums[2 * c]; dgamma[c] = bsums[2 * c + 1]; }
}

// ---- plain activations on conv outputs (forward is fused in the conv epilogue) ----------
// kind 1: LeakyReLU(0.2) from the OUTPUT y (sign(y) == sign(z));  2: tanh;  3: sigmoid;  4: ReLU
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy,
                                                      const float* __restrict__ y,
                                                      float* __restrict__ dz, long total, int kind,
                                                      float* __restrict__ amax) {
    __shared__ float amax_red[4];
    float vmax = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        const float o = y[e], d = dy[e];
        float r;
        if (kind == 1) r = d * (o > 0.f ? 1.f : 0.2f);
        else if (kind == 2) r = d * (1.f - o * o);
        else if (kind == 3) r = d * o * (1.f - o);
        else r = o > 0.f ? d : 0.f;
        dz[e] = r;
        vmax = fmaxf(vmax, fabsf(r));
    }
    if (amax) og_amax_own(og_block_max(vmax, amax_red), amax, blockIdx.x, gridDim.x);      // grid <= OG_AMAX_SLOTS
}

// per-channel sum over (N, HW): out[c] = sum_n sum_i x[n, c, i]   (conv bias gradient); grid (C, S): S partial sums
// per channel into ws[c][S], added in order by channel_partials_sum_kernel
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x,
                                                          float* __restrict__ out, int N, int C, int HW,
                                                          float* __restrict__ part) {
    __shared__ float red[16];
    const int c = blockIdx.x;
    const long total = (long)N * HW;
    const long chunk = (total + gridDim.y - 1) / gridDim.y;
    const long e0 = (long)blockIdx.y * chunk;
    const long e1 = min(total, e0 + chunk);
    float s = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += blockDim.x) {
        const long n = e / HW;
        const long i = e - n * HW;
        s += x[((size_t)n * C + c) * HW + i];
    }
    s = og_block_sum(s, red);
    if (threadIdx.x == 0) {
        if (gridDim.y == 1) out[c] = s; else part[(size_t)c * gridDim.y + blockIdx.y] = s;
    }
}
__global__ __launch_bounds__(256) void channel_partials_sum_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                   int C, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f;
    for (int k = 0; k < S; ++k) a += part[(size_t)c * S + k];
    out[c] = a;
}

#define OG_IN_FUSED_MAX 65536        // largest plane (elements) the one-kernel InstanceNorm takes: 256 KB, two passes
OG_KNOB(og_norm_nofuse, "OG_NORM_NOFUSE", 0)       // development builds: 1 = the three-kernel path everywhere

static inline int norm_splits(int G, long per_group) {
    // aim for >= 4 workgroups per CU overall, >= 1024 elements per workgroup
    long want = (256L * 4 + G - 1) / G;
    long maxs = (per_group + 1023) / 1024;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

// partial sums per group of the statistics kernels (plane-structured or generic launch)
static inline bool norm_planes(int N, int C, int HW) { return (HW % 4 == 0) && HW >= 256 && (long)N * C < 2000000; }
static inline int norm_partials(int N, int C, int HW, int per_channel) {
    const int G = per_channel ? C : N * C;
    if (norm_planes(N, C, HW)) return (per_channel ? N : 1) * og_cdiv(HW, OG_NORM_CHUNK);
    return norm_splits(G, per_channel ? (long)N * HW : HW);
}
static inline NormWs norm_ws(float* buf, int G, int P) { return NormWs{buf, buf + 2 * (size_t)G, P}; }
static inline void norm_sum_partials(const NormWs& ws, int G, hipStream_t s, float* amax = nullptr) {
    if (ws.P > 1) hipLaunchKernelGGL(norm_partials_sum_kernel, dim3(og_cdiv(G, 256)), dim3(256), 0, s, ws, G, amax);
}

extern "C" {

// Non-zero if objgan_norm_forward / objgan_norm_backward with these sizes fill `amax` (the OG_AMAX_SLOTS = 1024 partial maxima of
// |y| resp. |dx|: the scale input of the fp16x2 convolutions, see objgan_absmax_partials) inside their own launches: the
// plane-structured statistics + apply path with more than one statistics workgroup per group.  Host-only.
// Returns 2 for the one-kernel InstanceNorm path: its workgroups add their maxima with integer atomicMax and no
// earlier launch of the call could zero the slots -- the caller passes them zero-filled.
int objgan_norm_amax_supported(int N, int C, int HW, int per_channel, int affine) {
    if (N <= 0 || C <= 0 || HW <= 0 || !norm_planes(N, C, HW)) return 0;
    if (!per_channel && !affine && HW <= OG_IN_FUSED_MAX && !og_norm_nofuse()) return 2;      // one-kernel InstanceNorm
    return norm_partials(N, C, HW, per_channel) > 1 ? 1 : 0;
}

// Floats of statistics workspace (`sums` of objgan_norm_forward, `bsums` of objgan_norm_backward) for these sizes:
// [2G totals][2 * G * P partial pairs], P = workgroups per group of the statistics launch.  Host-only.
long objgan_norm_ws_floats(int N, int C, int HW, int per_channel) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    const long G = per_channel ? C : (long)N * C;
    return 2 * G + 2 * G * norm_partials(N, C, HW, per_channel);
}

// Forward: statistics + apply.  Workspaces: sums [objgan_norm_ws_floats] (totals first), mean [G], rstd [G].
// per_channel = 1 BatchNorm (G = C), 0 InstanceNorm (G = N*C).  gamma/beta/running_* may be
// null.  mode: 0 none, 1 LeakyReLU(0.2), 2 GLU (y has C/2 channels).  residual may be null.
int objgan_norm_forward(const float* x, float* y, const float* residual,
                        const float* gamma, const float* beta,
                        float* running_mean, float* running_var,
                        float* sums, float* mean, float* rstd,
                        int N, int C, int HW, int per_channel, int mode,
                        float eps, float momentum, float* amax, void* stream) {
    OG_ENTRY();
    if (mode == OG_NORM_GLU && (C & 1)) return OG_BAD_ARGS;
    if (N <= 0 || C <= 0 || HW <= 0) return OG_OK;
    if (amax && !objgan_norm_amax_supported(N, C, HW, per_channel, gamma != nullptr)) return OG_BAD_ARGS;
    hipStream_t s = (hipStream_t)stream;
    NormGeom gm{N, C, HW, per_channel};
    const int G = per_channel ? C : N * C;
    const long per_group = per_channel ? (long)N * HW : HW;
    const bool planes = norm_planes(N, C, HW);
    const int chunks = og_cdiv(HW, OG_NORM_CHUNK);
    const int Co = mode == OG_NORM_GLU ? C / 2 : C;
    if (planes && !per_channel && !gamma && HW <= OG_IN_FUSED_MAX && !og_norm_nofuse()) {
        // InstanceNorm (no affine parameters on this path): one workgroup owns its plane(s) end to end
        dim3 grid(N * Co);
        if (mode == OG_NORM_GLU)
            hipLaunchKernelGGL((in_fwd_fused_kernel<OG_NORM_GLU>), grid, dim3(256), 0, s, x, residual, y, mean, rstd, gm, eps, amax);
        else if (mode == OG_NORM_LRELU)
            hipLaunchKernelGGL((in_fwd_fused_kernel<OG_NORM_LRELU>), grid, dim3(256), 0, s, x, residual, y, mean, rstd, gm, eps, amax);
        else
            hipLaunchKernelGGL((in_fwd_fused_kernel<OG_NORM_NONE>), grid, dim3(256), 0, s, x, residual, y, mean, rstd, gm, eps, amax);
        return og_launch_status();
    }
    const NormWs ws = norm_ws(sums, G, norm_partials(N, C, HW, per_channel));
    if (planes) {
        hipLaunchKernelGGL(norm_stats_plane_kernel, dim3(N * C, chunks), dim3(256), 0, s, x, ws, gm);
    } else {
        dim3 grid(G, norm_splits(G, per_group));
        hipLaunchKernelGGL(norm_stats_kernel, grid, dim3(256), 0, s, x, ws, gm);
    }
    norm_sum_partials(ws, G, s, amax);          // (also zeroes the maxima slots the apply kernel fills)
    if (planes) {
        dim3 grid(N * Co, chunks);
        const NormFin fin{sums, mean, rstd, running_mean, running_var, eps, momentum};
        if (mode == OG_NORM_GLU)
            hipLaunchKernelGGL((norm_apply_plane_kernel<OG_NORM_GLU>), grid, dim3(256), 0, s, x, mean, rstd, gamma, beta, residual, y, gm, fin, amax);
        else if (mode == OG_NORM_LRELU)
            hipLaunchKernelGGL((norm_apply_plane_kernel<OG_NORM_LRELU>), grid, dim3(256), 0, s, x, mean, rstd, gamma, beta, residual, y, gm, fin, amax);
        else
            hipLaunchKernelGGL((norm_apply_plane_kernel<OG_NORM_NONE>), grid, dim3(256), 0, s, x, mean, rstd, gamma, beta, residual, y, gm, fin, amax);
        return og_launch_status();
    }
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(og_cdiv(G, 256)), dim3(256), 0, s, x, sums, mean,
                       rstd, running_mean, running_var, gm, G, eps, momentum);
    const long total = (long)N * Co * HW;
    hipLaunchKernelGGL(norm_apply_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, s, x, mean,
                       rstd, gamma, beta, residual, y, gm, mode);
    return og_launch_status();
}

// Apply only, with given statistics (eval-mode BatchNorm: running_mean / 1/sqrt(running_var + eps) --
// generator sampling with the EMA weights, reference evaluator.py; trainer.save_img_results).
int objgan_norm_apply(const float* x, float* y, const float* residual, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, int N, int C, int HW, int per_channel, int mode,
                      void* stream) {
    OG_ENTRY();
    if (mode == OG_NORM_GLU && (C & 1)) return OG_BAD_ARGS;
    if (N <= 0 || C <= 0 || HW <= 0) return OG_OK;
    hipStream_t s = (hipStream_t)stream;
    NormGeom gm{N, C, HW, per_channel};
    const int Co = mode == OG_NORM_GLU ? C / 2 : C;
    if ((HW % 4 == 0) && HW >= 256 && (long)N * C < 2000000) {
        dim3 grid(N * Co, og_cdiv(HW, OG_NORM_CHUNK));
        const NormFin fin{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f};
        if (mode == OG_NORM_GLU)
            hipLaunchKernelGGL((norm_apply_plane_kernel<OG_NORM_GLU>), grid, dim3(256), 0, s, x, mean, rstd, gamma, beta, residual, y, gm, fin, (float*)nullptr);
        else if (mode == OG_NORM_LRELU)
            hipLaunchKernelGGL((norm_apply_plane_kernel<OG_NORM_LRELU>), grid, dim3(256), 0, s, x, mean, rstd, gamma, beta, residual, y, gm, fin, (float*)nullptr);
        else
            hipLaunchKernelGGL((norm_apply_plane_kernel<OG_NORM_NONE>), grid, dim3(256), 0, s, x, mean, rstd, gamma, beta, residual, y, gm, fin, (float*)nullptr);
        return og_launch_status();
    }
    const long total = (long)N * Co * HW;
    hipLaunchKernelGGL(norm_apply_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, s, x, mean,
                       rstd, gamma, beta, residual, y, gm, mode);
    return og_launch_status();
}

// Backward.  dy has the shape of y (C/2 channels for GLU).  bsums [2G] workspace (zeroed
// here).  dgamma/dbeta may be null.  (The residual gradient is dy itself.)
int objgan_norm_backward(const float* x, const float* dy, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, float* bsums,
                         float* dx, float* dgamma, float* dbeta,
                         int N, int C, int HW, int per_channel, int mode, float* amax, void* stream) {
    OG_ENTRY();
    if (mode == OG_NORM_GLU && (C & 1)) return OG_BAD_ARGS;
    if (N <= 0 || C <= 0 || HW <= 0) return OG_OK;
    if (amax && !objgan_norm_amax_supported(N, C, HW, per_channel, gamma != nullptr)) return OG_BAD_ARGS;
    hipStream_t s = (hipStream_t)stream;
    NormGeom gm{N, C, HW, per_channel};
    const int G = per_channel ? C : N * C;
    const long per_group = per_channel ? (long)N * HW : HW;
    const bool planes = norm_planes(N, C, HW);
    if (planes && !per_channel && !gamma && HW <= OG_IN_FUSED_MAX && !og_norm_nofuse()) {
        const int Co = mode == OG_NORM_GLU ? C / 2 : C;
        dim3 grid(N * Co);
        if (mode == OG_NORM_GLU)
            hipLaunchKernelGGL((in_bwd_fused_kernel<OG_NORM_GLU>), grid, dim3(256), 0, s, x, dy, mean, rstd, dx, gm, amax);
        else if (mode == OG_NORM_LRELU)
            hipLaunchKernelGGL((in_bwd_fused_kernel<OG_NORM_LRELU>), grid, dim3(256), 0, s, x, dy, mean, rstd, dx, gm, amax);
        else
            hipLaunchKernelGGL((in_bwd_fused_kernel<OG_NORM_NONE>), grid, dim3(256), 0, s, x, dy, mean, rstd, dx, gm, amax);
        return og_launch_status();
    }
    const NormWs ws = norm_ws(bsums, G, norm_partials(N, C, HW, per_channel));
    if (planes) {
        const int Co = mode == OG_NORM_GLU ? C / 2 : C;
        dim3 grid(N * Co, og_cdiv(HW, OG_NORM_CHUNK));
#define OG_NB(MODE)                                                                                         \
        hipLaunchKernelGGL((norm_bwd_stats_plane_kernel<MODE>), grid, dim3(256), 0, s, x, dy, mean, rstd,   \
                           gamma, beta, ws, gm);                                                            \
        norm_sum_partials(ws, G, s, amax);                                                                  \
        hipLaunchKernelGGL((norm_bwd_apply_plane_kernel<MODE>), grid, dim3(256), 0, s, x, dy, mean, rstd,   \
                           gamma, beta, bsums, dx, gm, dgamma, dbeta, amax);
        if (mode == OG_NORM_GLU) { OG_NB(OG_NORM_GLU) } else if (mode == OG_NORM_LRELU) { OG_NB(OG_NORM_LRELU) } else { OG_NB(OG_NORM_NONE) }
#undef OG_NB
        return og_launch_status();
    } else {
        dim3 grid(G, norm_splits(G, per_group));
        hipLaunchKernelGGL(norm_bwd_stats_kernel, grid, dim3(256), 0, s, x, dy, mean, rstd, gamma, beta,
                           ws, gm, mode);
        norm_sum_partials(ws, G, s);
        const long total = (long)N * C * HW;
        hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(og_stream_grid(total, 256)), dim3(256), 0, s, x,
                           dy, mean, rstd, gamma, beta, bsums, dx, gm, mode);
    }
    if (dgamma && per_channel)
        hipLaunchKernelGGL(norm_affine_grad_kernel, dim3(og_cdiv(C, 256)), dim3(256), 0, s, bsums,
                           dgamma, dbeta, C);
    return og_launch_status();
}

// amax (may be NULL): the OG_AMAX_SLOTS partial maxima of |dz| (see objgan_absmax_partials)
int objgan_act_backward(const float* dy, const float* y, float* dz, long total, int kind, float* amax,
                        void* stream) {
    OG_ENTRY();
    if (kind < 1 || kind > 4) return OG_BAD_ARGS;
    if (total <= 0) return OG_OK;
    int grid = og_stream_grid(total, 256);
    if (amax && grid > OG_AMAX_SLOTS) grid = OG_AMAX_SLOTS;      // every workgroup owns a slot
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, y, dz, total, kind, amax);
    return og_launch_status();
}

// out[c] = sum over n, i of x[n, c, i]: fully written (one workgroup per channel, or S partial slots per channel in `ws`
// summed in slot order by channel_partials_sum_kernel); nothing is pre-zeroed
// floats of workspace objgan_channel_sum needs: [C * S partial sums]
long objgan_channel_sum_ws_floats(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    const int S = norm_splits(C, (long)N * HW);
    return S > 1 ? (long)C * S : 0;
}

int objgan_channel_sum(const float* x, float* out, int N, int C, int HW, float* ws, void* stream) {
    OG_ENTRY();
    if (N <= 0 || C <= 0 || HW <= 0) return OG_OK;
    hipStream_t s = (hipStream_t)stream;
    const int S = norm_splits(C, (long)N * HW);
    if (S > 1 && !ws) return OG_BAD_ARGS;
    dim3 grid(C, S);
    hipLaunchKernelGGL(channel_sum_kernel, grid, dim3(256), 0, s, x, out, N, C, HW, ws);
    if (S > 1) hipLaunchKernelGGL(channel_partials_sum_kernel, dim3(og_cdiv(C, 256)), dim3(256), 0, s, ws, out, C, S);
    return og_launch_status();
}

}  // extern "C"
