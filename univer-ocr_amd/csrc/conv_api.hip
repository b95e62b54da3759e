// C-ABI entry points of Convolutional2D: argument validation + dispatch between the
// shape-specialised kernels and the generic ones (conv.hip).
#include "conv_dims.h"

namespace {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_dims(uocr_ctx* ctx, const ConvDims& d) {
    UOCR_REQUIRE(ctx, d.n > 0 && d.h > 0 && d.w > 0 && d.cin > 0 && d.cout > 0);
    UOCR_REQUIRE(ctx, d.kh > 0 && d.kw > 0 && d.sh > 0 && d.sw > 0 && d.ph >= 0 && d.pw >= 0);
    UOCR_REQUIRE(ctx, d.oh > 0 && d.ow > 0);
    // every window must lie inside the padded input (convolutional.py:290-301 output shape)
    UOCR_REQUIRE(ctx, (d.oh - 1) * d.sh + d.kh <= d.h + 2 * d.ph);
    UOCR_REQUIRE(ctx, (d.ow - 1) * d.sw + d.kw <= d.w + 2 * d.pw);
    return UOCR_OK;
}

}  // namespace

extern "C" {

int uocr_conv2d_fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y, int n, int h,
                    int wd, int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int oh, int ow,
                    double pad_value, int use_bias, int act, double act_alpha) {
    UOCR_CHECK_CTX(ctx);
    const ConvDims d{n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, oh, ow};
    int rc = check_dims(ctx, d);
    if (rc) return rc;
    UOCR_REQUIRE(ctx, x && w && y && (b || !use_bias));
    UOCR_REQUIRE(ctx, act >= UOCR_ACT_NONE && act <= UOCR_ACT_SIGMOID);
    if (uocr_conv_h16_eligible(ctx, dtype, d, 0) && uocr_aligned_act(x, dtype) && uocr_aligned_act(y, dtype))
        return uocr_conv_fwd_h16(ctx, x, w, b, y, d, pad_value, use_bias, act, act_alpha);
    if (uocr_conv_t32_eligible(ctx, dtype, d, 0) && aligned16(x) && aligned16(y))
        return uocr_conv_fwd_t32(ctx, x, w, b, y, d, pad_value, use_bias, act, act_alpha);
    if (uocr_conv_h3_eligible(ctx, dtype, d) && aligned16(x) && aligned16(y))
        return uocr_conv_fwd_h3(ctx, x, w, b, y, d, pad_value, use_bias, act, act_alpha);
    if (uocr_conv_tiled_eligible(ctx, dtype, d) && uocr_aligned_act(x, dtype) && uocr_aligned_act(y, dtype))
        return uocr_conv_fwd_tiled(ctx, dtype, x, w, b, y, d, pad_value, use_bias, act, act_alpha);
    if (uocr_conv_fast_eligible(ctx, dtype, d, x, y, w))
        return uocr_conv_fwd_fast(ctx, dtype, x, w, b, y, d, pad_value, use_bias, act, act_alpha);
    if (uocr_conv_mfma_eligible(ctx, dtype, d, 0))
        return uocr_conv_fwd_mfma(ctx, x, w, b, y, d, pad_value, use_bias, act, act_alpha);
    return uocr_conv_fwd_generic(ctx, dtype, x, w, b, y, d, pad_value, use_bias, act, act_alpha);
}

int uocr_conv2d_bwd_data(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx, int n, int h, int wd,
                         int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int oh, int ow,
                         const void* x_act, int act, double act_alpha) {
    UOCR_CHECK_CTX(ctx);
    const ConvDims d{n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, oh, ow};
    int rc = check_dims(ctx, d);
    if (rc) return rc;
    UOCR_REQUIRE(ctx, dy && w && dx);
    UOCR_REQUIRE(ctx, act == UOCR_ACT_NONE || (x_act && (act == UOCR_ACT_SIGMOID || (act == UOCR_ACT_LEAKY && act_alpha > 0))));
    const ActMask mask{act == UOCR_ACT_NONE ? nullptr : x_act, act, act_alpha};
    if (uocr_conv_h16_eligible(ctx, dtype, d, 1) && uocr_aligned_act(dy, dtype) && uocr_aligned_act(dx, dtype) &&
        (!mask.y || uocr_aligned_act(mask.y, dtype)))
        return uocr_conv_dgrad_h16(ctx, dy, w, dx, d, mask);
    if (uocr_conv_t32_eligible(ctx, dtype, d, 1) && aligned16(dy) && aligned16(dx) && (!mask.y || aligned16(mask.y)))
        return uocr_conv_dgrad_t32(ctx, dy, w, dx, d, mask);
    if (uocr_conv_fast_eligible(ctx, dtype, d, dy, dx, w)) return uocr_conv_dgrad_fast(ctx, dtype, dy, w, dx, d, mask);
    if (uocr_conv_mfma_eligible(ctx, dtype, d, 1)) return uocr_conv_dgrad_mfma(ctx, dy, w, dx, d, mask);
    return uocr_conv_dgrad_generic(ctx, dtype, dy, w, dx, d, mask);
}

int uocr_conv2d_bwd_weight(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, int n, int h,
                           int wd, int cin, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int oh, int ow,
                           double pad_value, int use_bias, int accumulate) {
    UOCR_CHECK_CTX(ctx);
    const ConvDims d{n, h, wd, cin, cout, kh, kw, sh, sw, ph, pw, oh, ow};
    int rc = check_dims(ctx, d);
    if (rc) return rc;
    UOCR_REQUIRE(ctx, x && dy && dw && db);
    if (uocr_conv_wgrad_h16_eligible(ctx, dtype, d) && uocr_aligned_act(x, dtype) && uocr_aligned_act(dy, dtype))
        return uocr_conv_wgrad_h16(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    if (uocr_conv_wgrad_t32_eligible(ctx, dtype, d) && aligned16(x) && aligned16(dy))
        return uocr_conv_wgrad_t32(ctx, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    if (uocr_conv_wgrad_s2_h16_eligible(ctx, dtype, d) && uocr_aligned_act(x, dtype) && uocr_aligned_act(dy, dtype))
        return uocr_conv_wgrad_s2_h16(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    if (uocr_conv_fast_eligible(ctx, dtype, d, x, dy, dw))
        return uocr_conv_wgrad_fast(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    if (uocr_conv_mfma_eligible(ctx, dtype, d, 2))
        return uocr_conv_wgrad_mfma(ctx, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    return uocr_conv_wgrad_generic(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
}

}  // extern "C"
