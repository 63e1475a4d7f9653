// Frozen bidirectional LSTM text encoder forward (reference image_generation/model.py:85-179,
// RNN_ENCODER: nn.Embedding -> nn.LSTM(batch_first, bidirectional) on a packed sequence).
//
// The encoder is evaluated once per training step on B <= 32 captions of <= 12 words
// (trainer.py:367-369); it is latency-bound, not throughput-bound.  One workgroup owns one
// (caption, direction) and walks its time steps inside the kernel: the word vector and the hidden
// state live in LDS, thread j owns gate pre-activation j of the 4H gates and streams column j of
// the TRANSPOSED weight matrices ([I][4H], [H][4H]: consecutive threads read consecutive floats),
// threads < H then apply the cell update.  Packed-sequence semantics: direction 0 runs t = 0..len-1,
// direction 1 runs t = len-1..0, positions >= len of the output are zero, the final hidden state
// of each direction is returned (PyTorch gate order i, f, g, o).
#include "common.h"

__device__ __forceinline__ float lstm_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// grid = (B, 2), block = 4H threads (<= 1024); dynamic LDS = (I + 2H + 4H) floats
__global__ void lstm_bidir_fwd_kernel(const float* __restrict__ table, const long* __restrict__ captions,
                                      const int* __restrict__ lens,
                                      const float* __restrict__ wt_ih, const float* __restrict__ wt_hh,
                                      const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                      float* __restrict__ out, float* __restrict__ hn,
                                      int L, int Lout, int I, int H, int ntoken) {
    extern __shared__ float sm[];
    float* xs = sm;             // [I]
    float* hs = xs + I;         // [H]
    float* cs = hs + H;         // [H]
    float* gs = cs + H;         // [4H]
    const int b = blockIdx.x, dir = blockIdx.y;
    const int j = threadIdx.x;
    const int G = 4 * H;
    int len = lens[b];
    len = len < 0 ? 0 : (len > L ? L : len);
    const float* wi = wt_ih + (size_t)dir * I * G;
    const float* wh = wt_hh + (size_t)dir * H * G;
    const float bias = j < G ? b_ih[dir * G + j] + b_hh[dir * G + j] : 0.f;
    if (j < H) { hs[j] = 0.f; cs[j] = 0.f; }
    float* ob = out + ((size_t)b * 2 * H + (size_t)dir * H) * Lout;
    // positions past the caption length are zero (pad_packed_sequence + post_process_words)
    for (int e = j; e < H * Lout; e += blockDim.x) {
        const int t = e % Lout;
        if (t >= len) ob[(size_t)(e / Lout) * Lout + t] = 0.f;
    }
    for (int s = 0; s < len; ++s) {
        const int t = dir == 0 ? s : len - 1 - s;
        long tok = captions[(size_t)b * L + t];
        tok = tok < 0 ? 0 : (tok >= ntoken ? ntoken - 1 : tok);
        __syncthreads();
        for (int i = j; i < I; i += blockDim.x) xs[i] = table[(size_t)tok * I + i];
        __syncthreads();
        if (j < G) {
            float acc = bias;
            for (int i = 0; i < I; ++i) acc = fmaf(wi[(size_t)i * G + j], xs[i], acc);
            for (int k = 0; k < H; ++k) acc = fmaf(wh[(size_t)k * G + j], hs[k], acc);
            gs[j] = acc;
        }
        __syncthreads();
        if (j < H) {
            const float ig = lstm_sigmoid(gs[j]);
            const float fg = lstm_sigmoid(gs[H + j]);
            const float gg = tanhf(gs[2 * H + j]);
            const float og = lstm_sigmoid(gs[3 * H + j]);
            const float c = fg * cs[j] + ig * gg;
            const float h = og * tanhf(c);
            cs[j] = c;
            hs[j] = h;
            if (t < Lout) ob[(size_t)j * Lout + t] = h;
        }
    }
    __syncthreads();
    if (j < H) hn[(size_t)b * 2 * H + dir * H + j] = hs[j];
}

extern "C" {

// table [ntoken][I]; captions [B][L] int64 (0 = pad); lens [B] int32; wt_ih [2][I][4H] and
// wt_hh [2][H][4H] are the TRANSPOSED weight_ih_l0(_reverse) / weight_hh_l0(_reverse); b_* [2][4H].
// out [B][2H][Lout] (words_emb layout of the reference), hn [B][2H] (sent_emb).
int objgan_lstm_bidir_forward(const float* table, const long* captions, const int* lens,
                              const float* wt_ih, const float* wt_hh, const float* b_ih, const float* b_hh,
                              float* out, float* hn, int B, int L, int Lout, int I, int H, int ntoken,
                              void* stream) {
    OG_ENTRY();
    if (4 * H > 1024 || H < 1 || I < 1 || L < 1 || Lout < 1 || ntoken < 1) return OG_BAD_ARGS;
    if (B <= 0) return OG_OK;
    const int threads = (4 * H + 63) / 64 * 64;
    const size_t lds = sizeof(float) * (size_t)(I + 6 * H);
    if (lds > 64 * 1024) return OG_BAD_ARGS;
    hipLaunchKernelGGL(lstm_bidir_fwd_kernel, dim3(B, 2), dim3(threads), lds, (hipStream_t)stream,
                       table, captions, lens, wt_ih, wt_hh, b_ih, b_hh, out, hn, L, Lout, I, H, ntoken);
    return og_launch_status();
}

}  // extern "C"
