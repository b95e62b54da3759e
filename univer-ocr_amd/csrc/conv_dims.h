// Shared between the conv translation units (generic, fast paths, MFMA implicit GEMM, API).
#pragma once
#include "uocr_common.h"

struct ConvDims {
    int n, h, w, cin, cout, kh, kw, sh, sw, ph, pw, oh, ow;
};

// generic direct kernels (conv.hip): any shape, f32 / f64
int uocr_conv_fwd_generic(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y,
                          const ConvDims& d, double pad_value, int use_bias, int act, double act_alpha);
int uocr_conv_dgrad_generic(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx, const ConvDims& d);
int uocr_conv_wgrad_generic(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db,
                            const ConvDims& d, double pad_value, int use_bias, int accumulate);
