/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (obj-gan_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Plain-C restatement of the reference ROIAlign for the parity tests of the gfx950 kernel.
 *
 *   oracle_roi_align_forward   follows reference image_generation/models/roi_align/src/
 *                              roi_align.c:80-136 (ROIAlignForwardCpu), which is also the forward
 *                              of the CUDA kernel roi_align_kernel.cu:15-70.
 *   oracle_roi_align_backward  follows the CUDA kernel roi_align_kernel.cu:94-143.  The
 *                              reference's own CPU backward (roi_align.c:138-190) is NOT a
 *                              specification: its in-bounds test is inverted (it accumulates
 *                              only for out-of-range samples) and it is unreachable from Python
 *                              (functions/roi_align.py:38 asserts grad_output.is_cuda).
 *
 * Pinning: oracle_roi_align_forward is checked bit-for-bit against the reference's own
 * roi_align.c compiled into oracle/_ref/libref_roi_align.so (tests/test_oracle_cpu.py) and
 * against the committed golden vectors tests/golden/roi_align_*.npz generated from it.
 *
 * Written independently of the reference text: geometry is derived once per (roi, sample) and
 * the channel loop is innermost-but-one; the arithmetic (float/double promotion order) is what
 * must -- and does -- match.
 */
#include <math.h>
#include <stddef.h>

typedef struct {
    int valid;
    int hstart, wstart;
    float h_ratio, w_ratio;
} sample_t;

/* Types matter: all roi quantities are float; the literals `1.` / `0.` of the reference are
 * double, so `end - start + 1.`, `size / (aligned - 1.)` and `(1. - ratio)` are evaluated in
 * double and rounded to float when assigned; `ph * bin + start` is a float multiply then a
 * float add (this file must be built without FMA contraction: -ffp-contract=off). */
static sample_t roi_sample(const float* roi, float spatial_scale, int height, int width,
                           int aligned_height, int aligned_width, int ph, int pw) {
    sample_t s;
    float roi_start_w = roi[1] * spatial_scale;
    float roi_start_h = roi[2] * spatial_scale;
    float roi_end_w = roi[3] * spatial_scale;
    float roi_end_h = roi[4] * spatial_scale;
    float roi_width = fmaxf((float)((double)(roi_end_w - roi_start_w) + 1.0), 0.0f);
    float roi_height = fmaxf((float)((double)(roi_end_h - roi_start_h) + 1.0), 0.0f);
    float bin_size_h = (float)((double)roi_height / ((double)aligned_height - 1.0));
    float bin_size_w = (float)((double)roi_width / ((double)aligned_width - 1.0));
    volatile float hm = (float)ph * bin_size_h;   /* volatile: forbid fusing into an FMA */
    volatile float wm = (float)pw * bin_size_w;
    float h = hm + roi_start_h;
    float w = wm + roi_start_w;
    s.hstart = (int)fminf((float)floor((double)h), (float)(height - 2));
    s.wstart = (int)fminf((float)floor((double)w), (float)(width - 2));
    s.valid = !(h < 0 || h >= height || w < 0 || w >= width);
    s.h_ratio = h - (float)s.hstart;
    s.w_ratio = w - (float)s.wstart;
    return s;
}

void oracle_roi_align_forward(const float* feat, const float* rois, float* out, int num_rois,
                              int channels, int height, int width, int aligned_height,
                              int aligned_width, float spatial_scale) {
    for (int n = 0; n < num_rois; ++n) {
        const float* roi = rois + (size_t)n * 5;
        int img_start = (int)(((roi[0] * (float)channels) * (float)height) * (float)width);
        for (int ph = 0; ph < aligned_height; ++ph)
            for (int pw = 0; pw < aligned_width; ++pw) {
                sample_t s = roi_sample(roi, spatial_scale, height, width, aligned_height,
                                        aligned_width, ph, pw);
                for (int c = 0; c < channels; ++c) {
                    size_t o = (((size_t)n * channels + c) * aligned_height + ph) * aligned_width + pw;
                    if (!s.valid) { out[o] = 0.0f; continue; }
                    const float* p = feat + img_start + ((size_t)c * height + s.hstart) * width + s.wstart;
                    /* C typing of the reference expression (roi_align.c:131-134), term by term:
                     *   ul * (1. - hr) * (1. - wr)   float*double -> double, * double
                     *   ur * (1. - hr) * wr          float*double -> double, * float->double
                     *   dl * hr * (1. - wr)          float*float  -> FLOAT,  then * double
                     *   dr * hr * wr                 float*float*float -> FLOAT
                     * summed left to right in double, rounded to float by the store. */
                    double omh = 1.0 - (double)s.h_ratio, omw = 1.0 - (double)s.w_ratio;
                    double t1 = ((double)p[0] * omh) * omw;
                    double t2 = ((double)p[1] * omh) * (double)s.w_ratio;
                    volatile float f3 = p[width] * s.h_ratio;
                    double t3 = (double)f3 * omw;
                    volatile float f4a = p[width + 1] * s.h_ratio;
                    volatile float f4 = f4a * s.w_ratio;
                    double v = ((t1 + t2) + t3) + (double)f4;
                    out[o] = (float)v;
                }
            }
    }
}

/* bottom_grad must be zero-filled by the caller; contributions are added in (n, c, ph, pw) order
 * (the CUDA kernel adds them atomically in an unspecified order: compare with a tolerance). */
void oracle_roi_align_backward(const float* top_grad, const float* rois, float* bottom_grad,
                               int num_rois, int channels, int height, int width,
                               int aligned_height, int aligned_width, float spatial_scale) {
    for (int n = 0; n < num_rois; ++n) {
        const float* roi = rois + (size_t)n * 5;
        int img_start = (int)(((roi[0] * (float)channels) * (float)height) * (float)width);
        for (int c = 0; c < channels; ++c)
            for (int ph = 0; ph < aligned_height; ++ph)
                for (int pw = 0; pw < aligned_width; ++pw) {
                    sample_t s = roi_sample(roi, spatial_scale, height, width, aligned_height,
                                            aligned_width, ph, pw);
                    if (!s.valid) continue;
                    size_t o = (((size_t)n * channels + c) * aligned_height + ph) * aligned_width + pw;
                    float* p = bottom_grad + img_start + ((size_t)c * height + s.hstart) * width + s.wstart;
                    /* C typing of roi_align_kernel.cu:137-140: `(1. - h_ratio)` is double,
                     * `(1 - w_ratio)` is FLOAT (int literal); the two h_ratio terms are all-float. */
                    float g = top_grad[o];
                    double omh = 1.0 - (double)s.h_ratio;
                    volatile float omw = 1.0f - s.w_ratio;
                    volatile float g_h = g * s.h_ratio;
                    volatile float c3 = g_h * omw;
                    volatile float c4 = g_h * s.w_ratio;
                    p[0] += (float)(((double)g * omh) * (double)omw);
                    p[1] += (float)(((double)g * omh) * (double)s.w_ratio);
                    p[width] += c3;
                    p[width + 1] += c4;
                }
    }
}
