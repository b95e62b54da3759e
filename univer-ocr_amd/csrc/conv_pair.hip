// The Monochrome block as ONE forward and ONE backward launch (C-ABI entry points uocr_conv_pair_fwd / _bwd):
//     x (1 ch) -> conv3x3 (1 -> 16, pad 1) -> LeakyReLU -> conv3x3 (16 -> 1, pad 1) -> [Sigmoid] -> y
// (reference: my_model/model.py:108-135 built from nn/layers/convolutional.py:62-145 and
// nn/layers/layers.py:377-418).
//
// Run layer by layer the 16-channel activation a1 (268 MB at 32x256x512) is written once and read three times per
// train step and its gradient (another 268 MB) is written once and read once: 1.6 GB of HBM traffic for 34 MB of real
// input and output.  Here neither tensor leaves the chip: a1 is recomputed from the 1-channel input where it is
// needed, and every 16-channel contraction runs on the matrix cores.  The kernels: conv_pair_strip.hip (float32,
// v_mfma_f32_16x16x4_f32, exact f32 FMA chains) and conv_pair_strip_h.hip (binary16 storage, v_mfma_f32_16x16x16_f16):
// column strips walked down the rows, window operands in registers, tap sums of the backward-data / forward output in
// rolling registers.  (Rounds 1-2 used 16 x 32 tiles with an LDS scatter / gather of the 9 tap sums per position and a
// second kernel for the tile borders: 131 / 265 us where the strips take 97 / 157.)
#include "conv_pair.h"
#include "uocr_common.h"

namespace {

constexpr int C = 16;

int check_pair(uocr_ctx* ctx, int dtype, int n, int h, int w, int cmid, int act2, double alpha1) {
    if (UOCR_DTYPE_BASE(dtype) != UOCR_F32 && UOCR_DTYPE_BASE(dtype) != UOCR_F16)
        UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: float32 / float16 only");
    if (cmid != C) UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: 16 middle channels only (got %d)", cmid);
    if (act2 != UOCR_ACT_NONE && act2 != UOCR_ACT_SIGMOID)
        UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: output activation must be none or sigmoid");
    // the forward kernels take LeakyReLU as max(z, alpha z)
    if (!(alpha1 >= 0.0 && alpha1 <= 1.0))
        UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "conv_pair: LeakyReLU slope must lie in [0, 1] (got %g)", alpha1);
    UOCR_REQUIRE(ctx, n > 0 && h > 0 && w > 0 && n <= 65535);
    return UOCR_OK;
}

}  // namespace

extern "C" int uocr_conv_pair_fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w1, const void* b1,
                                  const void* w2, const void* b2, void* y, int n, int h, int w, int cmid,
                                  double pad_value1, int use_bias1, int use_bias2, double alpha1, int act2) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && w1 && b1 && w2 && b2 && y);
    int rc = check_pair(ctx, dtype, n, h, w, cmid, act2, alpha1);
    if (rc != UOCR_OK) return rc;
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F32)
        return uocr_pair_strip_fwd_f32(ctx, (const float*)x, (const float*)w1, (const float*)b1, (const float*)w2,
                                       (const float*)b2, (float*)y, n, h, w, (float)pad_value1, use_bias1, use_bias2,
                                       (float)alpha1, act2);
    return uocr_pair_strip_fwd_f16(ctx, x, (const float*)w1, (const float*)b1, (const float*)w2, (const float*)b2, y, n, h, w,
                                   (float)pad_value1, use_bias1, use_bias2, (float)alpha1, act2);
}

extern "C" int uocr_conv_pair_bwd(uocr_ctx* ctx, int dtype, const void* x, const void* y, const void* dy,
                                  const void* w1, const void* b1, const void* w2, void* dw1, void* db1, void* dw2,
                                  void* db2, void* dx, int n, int h, int w, int cmid, double pad_value1,
                                  int use_bias1, int use_bias2, double alpha1, int act2, int accumulate) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && y && dy && w1 && b1 && w2 && dw1 && db1 && dw2 && db2);
    int rc = check_pair(ctx, dtype, n, h, w, cmid, act2, alpha1);
    if (rc != UOCR_OK) return rc;
    if (UOCR_DTYPE_BASE(dtype) == UOCR_F32)
        return uocr_pair_strip_bwd_f32(ctx, (const float*)x, (const float*)y, (const float*)dy, (const float*)w1,
                                       (const float*)b1, (const float*)w2, (float*)dw1, (float*)db1, (float*)dw2,
                                       (float*)db2, (float*)dx, n, h, w, (float)pad_value1, use_bias1, use_bias2,
                                       (float)alpha1, act2 == UOCR_ACT_SIGMOID, accumulate, 1.f);
    return uocr_pair_strip_bwd_f16(ctx, x, y, dy, (const float*)w1, (const float*)b1, (const float*)w2, (float*)dw1,
                                   (float*)db1, (float*)dw2, (float*)db2, dx, n, h, w, (float)pad_value1, use_bias1,
                                   use_bias2, (float)alpha1, act2 == UOCR_ACT_SIGMOID, accumulate,
                                   (float)uocr_grad_unscale(dtype));
}
