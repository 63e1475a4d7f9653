/*
 * objgan_hip.h -- C ABI of libobjgan_hip.so: the MI355X (gfx950) native kernels behind the
 * Obj-GAN image_generation G+D training step.
 *
 * This is the drop-in boundary of the hot path.  The reference's only native boundary is the
 * cffi module `models/roi_align/_ext/roi_align` (reference
 * image_generation/models/roi_align/src/roi_align.h:1-5, src/roi_align_cuda.h:1-5):
 *
 *     int roi_align_forward_cuda (int aligned_height, int aligned_width, float spatial_scale,
 *                                 THCudaTensor* features, THCudaTensor* rois, THCudaTensor* output);
 *     int roi_align_backward_cuda(int aligned_height, int aligned_width, float spatial_scale,
 *                                 THCudaTensor* top_grad, THCudaTensor* rois, THCudaTensor* bottom_grad);
 *
 * objgan_roi_align_forward / _backward replace exactly those two (same argument meaning; the
 * TH tensor handles become raw device pointers + sizes, the implicit THC "current stream"
 * becomes an explicit hipStream_t).  Everything else the reference step obtains from
 * cuDNN / cuBLAS / THC through PyTorch (conv, norms, attention bmm+softmax, masked max,
 * bilinear resize, Adam) is exported here with the same conventions so that the host side
 * (Python, ctypes) stays a thin binding.
 *
 * Conventions (identical to the reference FFI, roi_align_cuda.c:7-40):
 *   - the library never allocates, frees or retains memory; the caller owns every buffer,
 *     including scratch / workspace buffers named in the signatures;
 *   - all tensors are contiguous fp32, NCHW, device memory;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, reads no environment
 *     variable (development builds, -DOG_DEV, do) and keeps no global state -- with one opt-in exception: the
 *     measurement window of objgan_prof_enable / _collect / _dump (a host-side table of hipEvents, off by default,
 *     touched by the calling thread only);
 *   - return value: 1 = launched, 0 = argument error (the reference returns 0 when
 *     rois.size(1) != 5), < 0 = -(hipError_t) of a failed launch (the reference calls exit(-1)).
 */
#ifndef OBJGAN_HIP_H
#define OBJGAN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ROIAlign (reference models/roi_align/src/roi_align_kernel.cu:15-143) ------------------ */
/* features [B, C, H, W], rois [num_rois, roi_cols = 5] = (batch_idx, x1, y1, x2, y2),
 * output [num_rois, C, aligned_height, aligned_width].  Returns 0 if roi_cols != 5. */
int objgan_roi_align_forward(const float* features, const float* rois, float* output,
                             int num_rois, int roi_cols, int channels, int height, int width,
                             int aligned_height, int aligned_width, float spatial_scale,
                             void* stream);
/* bottom_grad [batch_size, C, H, W] must be zero-filled by the caller (it is accumulated with
 * atomics, exactly like the reference kernel). */
int objgan_roi_align_backward(const float* top_grad, const float* rois, float* bottom_grad,
                              int batch_size, int num_rois, int roi_cols, int channels,
                              int height, int width, int aligned_height, int aligned_width,
                              float spatial_scale, void* stream);

/* The same adjoint with a FIXED summation order (bit-reproducible): the taps of an image's ROI samples are sorted once
 * by anchor pixel (stable) into a table in `ws`, and every (pixel, channel) gathers its taps in a fixed order.  The
 * reference's kernel (and objgan_roi_align_backward) leaves the order of its atomicAdds unspecified.  ws: at least
 * objgan_roi_align_backward_ws_floats(...) floats; shapes the query returns 0 for (more than 512 rois, maps over
 * ~5K pixels) and ws == NULL take objgan_roi_align_backward's path. */
long objgan_roi_align_backward_ws_floats(int batch_size, int num_rois, int channels, int height, int width,
                                         int aligned_height, int aligned_width);
int objgan_roi_align_backward_ordered(const float* top_grad, const float* rois, float* bottom_grad,
                                      int batch_size, int num_rois, int roi_cols, int channels,
                                      int height, int width, int aligned_height, int aligned_width,
                                      float spatial_scale, float* ws, long ws_floats, void* stream);
/* avg_pool2d(kernel_size=2, stride=1) tail of RoIAlignAvg (modules/roi_align.py:26-29). */
int objgan_avgpool2s1_forward(const float* in, float* out, long planes, int ih, int iw, void* stream);
int objgan_avgpool2s1_backward(const float* grad_out, float* grad_in, long planes, int ih, int iw,
                               void* stream);

/* ---- implicit-GEMM convolution on MFMA (reference model.py conv stacks, via cuDNN) ---------- */
long objgan_conv_packed_floats(int M, int C, int T);
/* layout of the packed bank objgan_conv_igemm uses for these arguments -- low byte: layout class (0..5), bits 8..:
 * channel chunks per K group of the row-major classes 1 / 3 / 4 / 5 -- : part of the key of any caller-side bank cache
 * (the same filter is served by different kernels and K orders at different sizes) */
int objgan_conv_bank_layout(int N, int C, int H, int W, int M, int Tg, int PH, int PW, int act, int math);
/* y[n,m,a*osh+ooh,b*osw+oow] = act(bias[m] + sum_{c,t} Wp[m][c*Tg+t] * x[n,c,a*s+dh[t],b*s+dw[t]])
 * for (a,b) in PH x PW.  w: [Cout][Cin][Torig]; transpose=0 -> m=cout,c=cin; 1 -> m=cin,c=cout.
 * src_tap[t]: which of the Torig taps GEMM tap t uses (-1 = zero).  1 <= Tg <= 32.
 * upsample=1: taps address a nearest-x2 upsampled view of x; pad_mode 0 zeros, 1 reflect.
 * act: 0 none, 1 LeakyReLU(0.2), 2 tanh, 3 sigmoid, 4 ReLU.  wt: scratch of
 * objgan_conv_packed_floats(M, C, Tg) floats.  y_prezeroed=1 tells the library that y already
 * holds zeros (lets partial-coverage launches, i.e. stride-2 dgrad phases, use split-K).
 * wt_packed=1: wt still holds the packed bank written by an earlier call with the same w, taps,
 * transpose flag, math and input size class (the caller caches it while w is unchanged); 0: pack now.
 * math: 0 = fp32 MFMA (exact fp32 fmaf chains); 1 = mixed precision (BASELINE config 5): operands rounded
 * to bf16 (RNE), fp32 accumulation, fp32 tensors at the boundary -- the call first writes a bf16 channel-blocked copy
 * [N][ceil16(C)/16][H][W][16] of x into ws and the matrix kernel reads its pixel operand from that copy as 16-byte vectors
 * (without ws: channel-strided fp32 gathers converted in registers, same values); 2 = "bf16x3": fp32 results on
 * the bf16 matrix pipe -- every fp32 operand is split exactly into three bf16 pieces (24 = 8 + 8 + 8 significand
 * bits) and the six largest of the nine partial products are accumulated in fp32; what is dropped is below 2^-24
 * of a product.  Outputs with <= 32 channels always run on the fp32 VALU kernels. */
int objgan_conv_igemm(const float* x, const float* w, const float* bias, float* y, float* wt,
                      int N, int C, int H, int W, int upsample, int pad_mode,
                      int Cout, int Cin, int Torig, int transpose,
                      int Tg, const int* dh, const int* dw, const int* src_tap,
                      int PH, int PW, int stride,
                      int OHf, int OWf, int osh, int osw, int ooh, int oow,
                      int act, int y_prezeroed, int wt_packed, int math, float* ring, const float* xmax, float* ymax,
                      float* ws, long ws_floats, void* stream);
/* math: 0 fp32 operands on the fp32 MFMA; 1 operands rounded to bf16; 2 "bf16x3" (fp32 operands split exactly three ways
 * on the bf16 MFMA, six products); 4 "fp16x2" (fp32 operands as two fp16 pieces of x * 2^s on the fp16 MFMA, three
 * products: residual <= 2^-23 |x| for every element within 2^-10 of its tensor's maximum, 2^-39 of that maximum below;
 * the filter bank's scale comes from partial maxima the pack path leaves behind the bank).  xmax: math 4 only -- the 1024
 * partial maxima of |x| (the kernel derives 2^s from them), written by objgan_absmax_partials or by the producer of x
 * itself (objgan_norm_forward / objgan_norm_backward / objgan_act_backward `amax`, this call's `ymax`); NULL otherwise.
 * ymax (may be NULL): 1024 ZERO-FILLED floats that receive the partial maxima of |y| -- from the kernel's epilogue where the
 * launch writes final values, from a pass over y otherwise. */
int objgan_absmax_partials(const float* x, long n, float* out1024, void* stream);
/* math 5: the arithmetic of math 4 with the pixel operand handed over as its PRE-SPLIT fp16 record instead of the fp32
 * tensor: `x` points to rec[n][c / 16][h | l][pixel][c % 16] fp16 with x * 2^s = h + l (channels C..ceil16(C)-1 zero), s
 * derived from the same `xmax` slots the call is given (they still undo the scale in the epilogue).  The kernel reads two
 * 16-byte loads per lane and K step instead of eight channel-strided dwords + the split on the VALU, and short block rows run
 * two 32-pixel groups per wave; results are bit-identical to math 4.  Only where objgan_conv_bank_layout answers class 5
 * (the MFMA implicit-GEMM kernel); the filter bank is the math-4 bank.  objgan_h2_records writes a record in one pass
 * (4 B read + 4 B written per element); objgan_h2_records_floats: its size in 4-byte units (= N * ceil16(C) * HW). */
long objgan_h2_records_floats(int N, int C, long HW);
int objgan_h2_records(const float* x, const float* xmax1024, void* rec, int N, int C, long HW, void* stream);
/* ws: the bf16 channel-blocked copy of x (math 1), then the split-K workspace.  Small-grid / long-K launches are split along K: every split stores its partial output into
 * its own slot of ws and a second kernel sums the slots in split order (+ bias, activation) -- bit-reproducible, no
 * zero-fill of y, no atomics.  objgan_conv_igemm_ws_floats (host-only, same geometry arguments; ring != 0 when a ring
 * buffer is passed) says how many floats a call needs (math 0 / 2: 0 for most); a call that needs them and gets fewer returns 0. */
long objgan_conv_igemm_ws_floats(int N, int C, int H, int W, int upsample, int pad_mode,
                                 int Cout, int Cin, int Torig, int transpose, int Tg,
                                 int PH, int PW, int stride, int OHf, int OWf, int osh, int osw,
                                 int act, int y_prezeroed, int math, int ring);
/* ring (may be NULL): data gradient of a ReflectionPad2d(1) convolution without the padded intermediate -- the
 * GEMM runs over the padded pixel grid (PH = OHf + 2, PW = OWf + 2, y is the UNPADDED gradient), interior
 * pixels are stored into y, the one-pixel border into ring [N*M][2*PW + 2*PH]; objgan_reflect_ring_fold then
 * adds the border back at its mirror positions.  Only for calls whose objgan_conv_bank_layout class (low byte) is
 * 1, 3, 4 or 5. */
int objgan_reflect_ring_fold(const float* ring, float* y, long planes, int H, int W, void* stream);
/* Data gradient of a stride-2 convolution whose four output parity phases have the same number
 * of taps (k=4, pad 1, even sizes): one launch for all phases.  x = dY [N,Cout,OH,OW],
 * y = dX [N,Cin,2*PH,2*PW], fully written (no pre-zeroing).  dh/dw/src_tap: 4 x Tg entries, phase
 * p = (row parity << 1) | column parity.  wt: 4*ceil(1.5*Cin*Tg*ceil16(Cout)) + 1024 floats (four phase banks,
 * then the partial maxima of |w| the fp16x2 pack leaves for its scale); wt_packed as above.
 * ws: objgan_conv_dgrad_s2_phases_ws_floats floats (math 1: the bf16 channel-blocked copy of dY, see objgan_conv_igemm; else 0). */
long objgan_conv_dgrad_s2_phases_ws_floats(int N, int Cout, int OH, int OW, int math);
/* Data gradient of a 4x4 / stride-2 / pad-1 convolution w.r.t. an input of Cin <= 32 channels (the first convolution of the
 * shape / object discriminators, reference model.py:1119-1128, 1217-1220), all four output parity phases in ONE launch of the
 * fp32 VALU kernel: dy [N, Cout, OH, OW] is read twice (the per-phase form read it four times), dx [N, Cin, 2 OH, 2 OW] is fully
 * written.  wt: objgan_conv_dgrad_s2_thin_floats(Cout, Cin) floats, packed by the call unless wt_packed;
 * objgan_conv_pack_job_thin_phase: the pack job of one of its four phase banks for objgan_conv_pack_jobs_run. */
long objgan_conv_dgrad_s2_thin_floats(int Cout, int Cin);
int objgan_conv_pack_job_thin_phase(void* job, const float* w, float* wt, int Cout, int Cin, int phase);
int objgan_conv_dgrad_s2_thin(const float* dy, const float* w, float* dx, float* wt, int N, int Cout, int OH, int OW,
                              int Cin, int wt_packed, void* stream);
int objgan_conv_dgrad_s2_phases(const float* x, const float* w, float* y, float* wt,
                                int N, int Cout, int OH, int OW, int Cin, int Torig,
                                int Tg, const int* dh, const int* dw, const int* src_tap,
                                int PH, int PW, int wt_packed, int math, const float* xmax, float* ws, long ws_floats,
                                void* stream);
/* dw[co][ci][kh][kw] = (accumulate ? dw : 0) + sum dy * x; ksize in {1,3,4}.  The reduction over pixels is split
 * across workgroups; the partial tiles go through ws (objgan_conv_wgrad_ws_floats floats, host-only query) and are
 * summed in split order: the weight gradient is bit-reproducible and dw needs no zero-fill.  math 1: the workspace also
 * holds the bf16 copies the kernel reads (x channel-blocked as in objgan_conv_igemm, dy in its own layout); without a
 * workspace the call gathers the fp32 tensors and rounds them in registers (same values). */
long objgan_conv_wgrad_ws_floats(int N, int Cin, int H, int W, int upsample, int pad_mode,
                                 int Cout, int OH, int OW, int ksize, int stride, int pad, int math);
int objgan_conv_wgrad(const float* x, const float* dy, float* dw,
                      int N, int Cin, int H, int W, int upsample, int pad_mode,
                      int Cout, int OH, int OW, int ksize, int stride, int pad, int math,
                      int accumulate, const float* xmax, const float* dymax, float* ws, long ws_floats, void* stream);
/* (math 4: xmax / dymax = objgan_absmax_partials of x / dy; launches planned on the LDS-staged or first-generation
 * kernels run bf16x3 instead -- both are fp32-result arithmetics) */
/* math 5: `x` is the fp16 RECORD of the source (objgan_h2_records under xmax; dy stays the fp32 tensor): the lanes read
 * half records and transpose them through LDS (ds_read_b64_tr_b16) instead of gathering 32 (channel, tap) planes per
 * wave from the fp32 tensor and splitting on the VALU.  Same products per 16-pixel step as math 4 (fp32 results; the
 * pixel splits differ, so the two are equal up to summation order).  Only where objgan_conv_wgrad_rec_ok says so
 * (OH * OW % 32 == 0, OH, OW <= 256, (H - 1) * W < 65535).
 * math 6 (round 6): as 5, and dy is ALSO read pre-split -- its fp16 pair (plane h, plane l, same layout) is written into
 * the workspace by one pass and the K loop carries no operand split; bit-identical to math 5.  math 7 (round 6): as 5 with
 * two (tap, 32-channel) column groups per wave and block rows of at most 128 rows (launches of fewer than 16 384 pixels
 * take math 5's kernel).  Both measured no faster in the training step (profiles/r06_ab_variants.txt): kept for the
 * record, not used by the Python layer's default. */
int objgan_conv_wgrad_rec_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ksize);
/* bf16 mode (math 1, BASELINE config 5) with the operand copy made ONCE per tensor (round 6).  math 1 writes a bf16
 * channel-blocked copy [N][Cp/16][HW][16] of its pixel operand into the call's workspace on every call: the forward
 * convolution and the weight gradient of a layer each copy the same x, every branch of an Inception block copies its shared
 * input (32.9 ms of 216 per B = 32 step were such copies).  objgan_nhwc_bf16 makes that copy on its own
 * (objgan_nhwc_bf16_floats floats; 0 = too large, stay on math 1), and math 3 is math 1 with `x` pointing to it:
 * objgan_conv_igemm(..., math = 3) where objgan_conv_bank_layout(..., math = 1) answers class 3, objgan_conv_wgrad(...,
 * math = 3) where objgan_conv_wgrad_bfb_ok says so (dy is still copied per call).  Same values, same kernels: bit-identical
 * to math 1. */
long objgan_nhwc_bf16_floats(int N, int C, long HW);
int objgan_nhwc_bf16(const float* x, float* out, int N, int C, long HW, void* stream);
int objgan_conv_wgrad_bfb_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ksize);

/* ---- frozen text encoder (reference model.py:85-179 RNN_ENCODER: Embedding + bidirectional LSTM on
 * a packed sequence).  table [ntoken][I]; captions [B][L] int64; lens [B] int32; wt_ih [2][I][4H] and
 * wt_hh [2][H][4H] are the transposed weight_ih_l0(_reverse) / weight_hh_l0(_reverse); b_* [2][4H]
 * (gate order i, f, g, o).  out [B][2H][Lout] = words_emb (zero past each length), hn [B][2H] = sent_emb. */
int objgan_lstm_bidir_forward(const float* table, const long* captions, const int* lens,
                              const float* wt_ih, const float* wt_hh, const float* b_ih, const float* b_hh,
                              float* out, float* hn, int B, int L, int Lout, int I, int H, int ntoken,
                              void* stream);

/* ---- normalisation + GLU / LeakyReLU / residual (BatchNorm train mode, InstanceNorm) -------- */
/* `sums` (forward) / `bsums` (backward): statistics workspace of objgan_norm_ws_floats(N, C, HW, per_channel) floats
 * (host-only query): every workgroup of the statistics launch stores its partial pair into its own slot, a second
 * one-thread-per-group kernel (norm_partials_sum_kernel) adds the slots in slot order -- bit-reproducible statistics, no
 * fp32 atomics, nothing to pre-zero; the totals are the first 2G floats. */
long objgan_norm_ws_floats(int N, int C, int HW, int per_channel);
int objgan_norm_forward(const float* x, float* y, const float* residual,
                        const float* gamma, const float* beta,
                        float* running_mean, float* running_var,
                        float* sums, float* mean, float* rstd,
                        int N, int C, int HW, int per_channel, int mode,
                        float eps, float momentum, float* amax, void* stream);
/* amax (may be NULL; only where objgan_norm_amax_supported is non-zero -- 2: the caller passes the slots ZERO-FILLED): the 1024 partial maxima of |y| (forward) / |dx|
 * (backward), filled inside the call's own launches -- the scale input of an fp16x2 convolution that reads the tensor
 * next, instead of a separate pass over it (objgan_absmax_partials). */
int objgan_norm_amax_supported(int N, int C, int HW, int per_channel, int affine);
/* apply with given statistics (eval-mode BatchNorm: mean = running_mean, rstd = 1/sqrt(running_var+eps)) */
int objgan_norm_apply(const float* x, float* y, const float* residual, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, int N, int C, int HW, int per_channel, int mode,
                      void* stream);
int objgan_norm_backward(const float* x, const float* dy, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, float* bsums,
                         float* dx, float* dgamma, float* dbeta,
                         int N, int C, int HW, int per_channel, int mode, float* amax, void* stream);
int objgan_act_backward(const float* dy, const float* y, float* dz, long total, int kind, float* amax, void* stream);
/* out[c] = sum over (n, i) of x[n, c, i] (bias gradients); ws: objgan_channel_sum_ws_floats floats (ordered combine) */
long objgan_channel_sum_ws_floats(int N, int C, int HW);
int objgan_channel_sum(const float* x, float* out, int N, int C, int HW, float* ws, void* stream);

/* ---- attention (reference GlobalAttention.py:32-181, miscc/utils.py:401-413) ---------------- */
int objgan_attn_general_forward(const float* x, const float* src, const unsigned char* mask,
                                float* wc, float* attn, int B, int idf, int Q, int L, void* stream);
/* dsrc is fully written (no zero-fill): every wave stores its partial tile into ws
 * (objgan_attn_general_backward_ws_floats floats, host-only query) and a second kernel sums the waves in order */
long objgan_attn_general_backward_ws_floats(int B, int idf, int Q, int L);
int objgan_attn_general_backward(const float* x, const float* src, const float* attn,
                                 const float* dwc, const float* dattn, float* dx, float* dsrc,
                                 int B, int idf, int Q, int L, float* ws, void* stream);
int objgan_attn_bu_forward(const float* tgt, const float* ctx1, const float* src,
                           const unsigned char* mask, float* wc, float* attn,
                           int B, int d2, int idf, int R, int L, int normalize, float eps,
                           void* stream);
int objgan_attn_bu_backward(const float* dwc, const float* attn, float* dsrc,
                            int B, int idf, int R, int L, void* stream);
int objgan_masked_max_forward(const float* f, const float* m, float* out, int B, int num, int R,
                              int P, long m_stride_b, long m_stride_r, long m_stride_c, void* stream);
/* df is fully written (no zero-fill); ws: objgan_masked_max_backward_ws_floats floats (ordered two-level sum) */
long objgan_masked_max_backward_ws_floats(int B, int num, int R, int P);
int objgan_masked_max_backward(const float* f, const float* m, const float* dout, float* df,
                               int B, int num, int R, int P, long m_stride_b, long m_stride_r,
                               long m_stride_c, float* ws, void* stream);
int objgan_softmax_strided_forward(const float* x, float* y, long outer, int dim, long inner,
                                   float scale, const int* lens, int nlens,
                                   const unsigned char* rowvalid, void* stream);
int objgan_softmax_strided_backward(const float* y, const float* dy, float* dx, long outer, int dim,
                                    long inner, float scale, void* stream);

/* ---- resize, gradient folds, optimiser ------------------------------------------------------ */
int objgan_bilinear_forward(const float* x, float* y, long planes, int ih, int iw, int oh, int ow,
                            void* stream);
int objgan_bilinear_backward(const float* dy, float* dx, long planes, int ih, int iw, int oh, int ow,
                             void* stream);
int objgan_sum2x2(const float* dy, float* dx, long planes, int h, int w, void* stream);
int objgan_reflect_fold(const float* dxp, float* dx, long planes, int h, int w, void* stream);
/* nn.BCELoss()(p, constant target t) on the small probability maps of the discriminator heads (reference
 * miscc/losses.py:182-204 and every other adversarial term): mean over n elements -> out[0]; log clamped at
 * -100 and gradient (p - t) / max(p (1 - p), 1e-12) like torch.  g: the upstream scalar gradient (device). */
int objgan_bce_const_forward(const float* p, float* out, int n, float t, void* stream);
int objgan_bce_const_backward(const float* p, const float* g, float* dp, int n, float t, void* stream);
/* Batched re-packing of cached filter banks.  A job is an opaque blob (objgan_conv_pack_job_bytes() bytes) that
 * says "pack w into wt exactly as objgan_conv_igemm -- or phase `phase` of objgan_conv_dgrad_s2_phases --
 * does for these arguments"; a caller that keeps packed banks (wt_packed = 1) stores the blobs of a network
 * back to back in device memory and refreshes all of them with ONE launch after an optimizer step (the
 * per-use re-pack was 499 launches per training step).  Replaces nothing in the reference: cuDNN transforms
 * filters internally. */
int objgan_conv_pack_job_bytes();
int objgan_conv_pack_job(void* job, const float* w, float* wt, int N, int C, int H, int W, int Cout, int Cin,
                         int Torig, int transpose, int Tg, const int* src_tap, int PH, int PW, int act, int math);
int objgan_conv_pack_job_phase(void* job, const float* w, float* wt, int Cout, int Cin, int Torig, int Tg,
                               const int* src_tap_phase, int phase, int math);
int objgan_conv_pack_jobs_run(const void* jobs_dev, int njobs, void* stream);
/* Batched small matrix product C[b] = A[b] . B[b] with arbitrary element strides (transposes are free): the
 * per-image region-context products of the DAMSM word loss (reference GlobalAttention.py:62-68 inside the
 * caption loop of losses.py:87-127: B torch.bmm calls) in one launch, likewise both of its gradients. */
int objgan_bmm_strided(const float* A, const float* B, float* C, int batch, int M, int N, int K,
                       long sab, long sam, long sak, long sbb, long sbk, long sbn,
                       long scb, long scm, long scn, void* stream);
/* Layout-map stem of the object discriminators without the 512x512 lift (reference
 * image_generation/model.py:1217-1226: shp_code(F.interpolate(seg, 512, bilinear, align_corners))).  The channel
 * contraction runs at low resolution (a 1x1 convolution C -> 9*Mo through objgan_conv_igemm); these two entry
 * points apply the separable per-axis operator "shift by the tap offset o reflect-pad o bilinear lift" to its
 * result, forward and adjoint.  z [N, 9*Mo, h, w] (channel (dh*3+dw)*Mo + co), y / dy [N, Mo, SH, SW],
 * scratch N*3*Mo*h*SW floats.  Tables live in device memory and are built by the caller:
 *   forward  [3][S] per axis: source indices i0, i1 and the weight l1 of i1 (l0 = 1 - l1), per tap offset;
 *   backward CSR transposes: off [3][len+1], idx, wt. */
int objgan_lift_taps_forward(const float* z, const float* bias, float* y, float* scratch, int N, int Mo, int h,
                             int w, int SH, int SW, const int* ci0, const int* ci1, const float* cl1,
                             const int* ri0, const int* ri1, const float* rl1, void* stream);
int objgan_lift_taps_backward(const float* dy, float* dz, float* scratch, int N, int Mo, int h, int w, int SH,
                              int SW, const int* roff, const int* ridx, const float* rwt, const int* coff,
                              const int* cidx, const float* cwt, void* stream);
/* Pooling layers of the frozen Inception-v3 encoder (reference image_generation/model.py:203-287:
 * F.max_pool2d(k3, s2), F.avg_pool2d(k3, s1, p1), F.avg_pool2d(k8)); planes = N*C NCHW planes.
 * mode 0 = max (no padding; idx receives the plane-local arg-max of every output for the backward pass,
 * first maximum in row-major order like torch), 1 = average with divisor k*k (count_include_pad). */
int objgan_pool2d_forward(const float* x, float* y, int* idx, long planes, int H, int W, int OH, int OW,
                          int k, int s, int p, int mode, void* stream);
int objgan_pool2d_backward(const float* dy, const int* idx, float* dx, long planes, int H, int W, int OH,
                           int OW, int k, int s, int p, int mode, void* stream);
/* torch.optim.Adam update over flat arenas; the gradient is pre-multiplied by grad_scale
 * (1/world_size under data parallelism: the RCCL all-reduce is a plain sum).  Hyper-parameters are
 * doubles like torch's python floats: 1 - beta and lr / (1 - beta1^t) are evaluated in double. */
int objgan_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1,
                     double beta2, double eps, int step, float grad_scale, void* stream);
/* The same update, taken only when the DEVICE flag flag[0] > 0, with the step counter on the device
 * (state: 3 doubles {steps, beta1^steps, beta2^steps}, initialised {0, 1, 1}; coef: 3 floats scratch).
 * Replaces the host-side `if float(errObjSSD) > 0:` of reference image_generation/trainer.py:429,440
 * under data parallelism: the flag rides behind the gradient arena through the all-reduce, so every
 * rank takes or skips the same update without a device->host read.  grad_scale < 0: the gradient is divided by
 * flag[0] -- after the all-reduce the number of ranks that contributed one -- instead of a fixed 1 / world size. */
int objgan_adam_step_gated(float* p, const float* g, float* m, float* v, long n, double lr, double beta1,
                           double beta2, double eps, double* state, const float* flag, float* coef,
                           float grad_scale, void* stream);
int objgan_ema_update(float* avg, const float* p, long n, float decay, float one_minus_decay,
                      void* stream);

/* ---- training images on the device (the loader side of the path) -------------------------
 * Replaces, per batch, the host-side `transforms.Resize((s, s))(img)` + ToTensor + Normalize(0.5, 0.5) of
 * reference image_generation/miscc/load.py:141-150 (Pillow's antialiased bilinear Image.resize, once per
 * branch size and image): the decoded 8-bit RGB images of a batch arrive back to back in one byte buffer
 * (rows of W*3 bytes; image b at byte offs[b], hs[b] x ws[b] pixels; offs / hs / ws are DEVICE arrays) and
 * out [B, 3, S, S] is bit for bit what Pillow + torch compute on the host.  Hmax >= every height,
 * kmax >= objgan_resize_pil_kmax(largest side of the batch, S) (the tap count Pillow allots per pixel).
 * coef_scratch: B*2*S*(kmax+2) ints, tmp_scratch: B*Hmax*S*3 bytes. */
int objgan_resize_pil_kmax(int max_in, int S);
int objgan_resize_pil_rgb8(const unsigned char* src, const long* offs, const int* hs, const int* ws, int B,
                           int Hmax, int kmax, int S, int* coef_scratch, unsigned char* tmp_scratch, float* out,
                           void* stream);

/* ---- training images: baseline JPEG decode on the device ------------------------------
 * Replaces the host decode `Image.open(BytesIO(img_bytes)).convert('RGB')` of reference
 * image_generation/miscc/load.py:141-151 (get_imgs; Pillow = libjpeg-turbo with the library defaults: JDCT_ISLOW, fancy
 * upsampling, RGB output): the FILE bytes cross PCIe, the decoded image never exists on the host.  Bit for bit Pillow's
 * output (integer arithmetic throughout: jdhuff.c entropy decode, jidctint.c inverse DCT, jdsample.c triangle-filter
 * upsampling, jdcolor.c fixed-point colour conversion).
 *
 * objgan_jpeg_parse     HOST ONLY.  Markers, quantisation and Huffman tables of one file -> a descriptor of
 *                       objgan_jpeg_desc_bytes() bytes whose first ten ints are {width, height, components, hmax, vmax,
 *                       MCUs per row, MCU rows, restart interval, reason, scan offset}.  Returns 1 for a file the device
 *                       decodes (baseline / sequential Huffman, 8 bit, one interleaved scan, grey or YCbCr 4:4:4 / 4:2:2 /
 *                       4:2:0) and 0 with reason != 0 for anything else (1 not JPEG, 2 progressive / arithmetic / lossless,
 *                       3 precision, 4 components, 5 sampling, 6 scan, 7 tables, 8 truncated, 9 RGB-coded): the caller routes that file
 *                       to its host decoder knowingly.
 * objgan_jpeg_plan      HOST ONLY.  Lays a batch out: descs[i] gets its file's byte offset in the batch buffer
 *                       (file_offsets[i], multiples of 16), its output byte offset, its workspace slices and the slice of
 *                       the entropy index handed in (nsegs[i] entries; NULL / 0: none).
 *                       -> workspace bytes (0: a refused descriptor in the batch).
 * objgan_jpeg_decode    files: the batch's files back to back (device, buffer padded to 16 bytes); descs_host: the planned
 *                       descriptors (launch geometry), descs_dev: a device copy of the same bytes (what the kernels read);
 *                       out: image i as [height][width][3] RGB bytes at its output offset; ws: objgan_jpeg_plan's byte
 *                       count.  The Huffman scan of a file is serial: a file without index is walked by ONE lane, which
 *                       writes its state at every MCU-row start to index_out (objgan_jpeg_seg_bytes() bytes per MCU row,
 *                       image i at the sum of the MCU rows of the images before it; NULL: not wanted); handed back in
 *                       as index_in (NULL: none) the same file is decoded by one lane per MCU row -- a training set is
 *                       decoded once per epoch, so the index of the first epoch serves all later ones. */
long objgan_jpeg_desc_bytes(void);
long objgan_jpeg_seg_bytes(void);
int objgan_jpeg_parse(const unsigned char* file, long nbytes, void* desc_out);
long objgan_jpeg_plan(void* descs, int n, const long* file_offsets, const long* out_offsets, const int* nsegs);
int objgan_jpeg_decode(const unsigned char* files, const void* descs_host, const void* descs_dev, int n,
                       unsigned char* out, void* ws, long ws_bytes, const void* index_in, void* index_out, void* stream);

/* ---- per-box instance masks on the device (the loader side of the path) --------------------
 * Replaces the four `skimage.transform.resize(mask, [s, s])` calls per box of reference
 * image_generation/miscc/load.py:160-176 (s = 32, 64, 128, 256 from the 64 x 64 instance mask): `count` square
 * float64 masks src[count][n][n] (n <= 64) -> out[k][count][sizes[k]][sizes[k]] float64 for nsizes <= 4 sizes (out: HOST
 * array of device pointers).  skimage's defaults as scipy.ndimage evaluates them (Gaussian anti-aliasing when
 * shrinking, order-1 zoom, mode 'mirror', grid_mode, clip to the input range), in float64 and in scipy's operation
 * order: bit for bit scipy's results.  taps[k * 17 ..]: the ntaps[k] = 2 radius + 1 <= 17 normalised weights of the
 * anti-aliasing filter of size k (host array; ntaps[k] = 0 when sizes[k] >= n). */
int objgan_mask_resize(const double* src, int count, int n, int nsizes, const int* sizes, double* const* out,
                       const int* ntaps, const double* taps, void* stream);

/* ---- measurement aid (bench.py roofline leg): hipEvent-bracketed conv launches ------------------ */
int objgan_prof_enable(int on);
int objgan_prof_collect(double* ms, double* flops, long* count);   /* arrays of 192 categories (48..95: fp16x2 instances; 96..191: fp16x2 on records, one / two pixel groups) */
/* per-launch records of the current window (call before objgan_prof_collect, which resets it):
 * meta[10*i..] = {kind 0 GEMM / 1 weight gradient / 2 thin, tile height, rows, K channels, taps, images,
 * pixel-grid rows, pixel-grid columns, stride (negative: strided output phases), K splits} */
int objgan_prof_dump(float* ms, double* flops, int* meta, int max_records, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* OBJGAN_HIP_H */
