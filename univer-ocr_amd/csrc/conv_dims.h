// Shared between the conv translation units (generic, fast paths, MFMA implicit GEMM, API).
#pragma once
#include "uocr_common.h"

struct ConvDims {
    int n, h, w, cin, cout, kh, kw, sh, sw, ph, pw, oh, ow;
};

// optional epilogue of backward-data: dx *= act'(y) evaluated from the activation OUTPUT y that is
// this conv's input tensor (same shape as dx).  Folds the backward pass of a preceding fused
// conv+LeakyReLU/Sigmoid into this kernel's store (act = UOCR_ACT_NONE: plain dx).
struct ActMask {
    const void* y;
    int act;
    double alpha;
};

// f32 MFMA implicit GEMM (gemm_mfma.hip); which: 0 fwd, 1 dgrad, 2 wgrad
bool uocr_conv_mfma_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, int which);
int uocr_conv_fwd_mfma(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                       double pad_value, int use_bias, int act, double act_alpha);
int uocr_conv_dgrad_mfma(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d,
                         const ActMask& mask);
int uocr_conv_wgrad_mfma(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                         double pad_value, int use_bias, int accumulate);
// binary16-MFMA kernels of the small-channel convs (conv_h16.hip), UOCR_F16 only; which: 0 fwd, 1 dgrad
bool uocr_conv_h16_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, int which);
int uocr_conv_fwd_h16(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                      double pad_value, int use_bias, int act, double act_alpha);
int uocr_conv_dgrad_h16(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d, const ActMask& mask);
bool uocr_upconv_h16_eligible(uocr_ctx* ctx, int dtype, int cin, int cout);
int uocr_upconv_fwd_h16(uocr_ctx* ctx, const void* x_low, const void* w, const void* b, void* y, int n, int hl, int wl,
                        int use_bias, int act, double act_alpha);
int uocr_upconv_dgrad_h16(uocr_ctx* ctx, const void* dy, const void* w, void* dx_low, int n, int hl, int wl,
                          const void* mask_y, int mask_act, double mask_alpha);
// ... their weight gradients (conv_h16w.hip)
bool uocr_conv_wgrad_h16_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d);
int uocr_conv_wgrad_h16(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                        double pad_value, int use_bias, int accumulate);
bool uocr_conv_wgrad_s2_h16_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d);
int uocr_conv_wgrad_s2_h16(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                           double pad_value, int use_bias, int accumulate);
int uocr_upconv_wgrad_h16(uocr_ctx* ctx, const void* x_low, const void* dy, float* partial, size_t partial_floats,
                          int n, int hl, int wl, int ch, int* nblocks);
// float32 vertical-Toeplitz MFMA kernels of the small-channel convs (conv_t32.hip); which: 0 fwd, 1 dgrad
bool uocr_conv_t32_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, int which);
int uocr_conv_fwd_t32(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                      double pad_value, int use_bias, int act, double act_alpha);
int uocr_conv_dgrad_t32(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d, const ActMask& mask);
// ... their weight gradients (conv_t32w.hip)
bool uocr_conv_wgrad_t32_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d);
int uocr_conv_wgrad_t32(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                        double pad_value, int use_bias, int accumulate);
bool uocr_upconv_t32_eligible(uocr_ctx* ctx, int dtype, int cin, int cout);
int uocr_upconv_dgrad_t32(uocr_ctx* ctx, const void* dy, const void* w, void* dx_low, int n, int hl, int wl, int ch,
                          const void* mask_y, int mask_act, double mask_alpha);
// float32 5x5 4 -> 2 forward on error-compensated binary16 MFMAs (conv_h3.hip; experiment, ctx option h3)
bool uocr_conv_h3_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d);
int uocr_conv_fwd_h3(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                     double pad_value, int use_bias, int act, double act_alpha);
// LDS-tiled forward for the 5x5 stride-1 4-channel convs (conv_tiled.hip), f32 / f16 storage
bool uocr_conv_tiled_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d);
int uocr_conv_fwd_tiled(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                        double pad_value, int use_bias, int act, double act_alpha);
// shape-specialised direct kernels for the skinny my_model convs (conv_fast.hip), f32 / f16 storage
bool uocr_conv_fast_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, const void* p0, const void* p1,
                             const void* p2);
int uocr_conv_fwd_fast(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y,
                       const ConvDims& d, double pad_value, int use_bias, int act, double act_alpha);
int uocr_conv_dgrad_fast(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx, const ConvDims& d,
                         const ActMask& mask);
int uocr_conv_wgrad_fast(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db,
                         const ConvDims& d, double pad_value, int use_bias, int accumulate);
// generic direct kernels (conv.hip): any shape, f32 / f64
int uocr_conv_fwd_generic(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y,
                          const ConvDims& d, double pad_value, int use_bias, int act, double act_alpha);
int uocr_conv_dgrad_generic(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx, const ConvDims& d,
                            const ActMask& mask);
int uocr_conv_wgrad_generic(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db,
                            const ConvDims& d, double pad_value, int use_bias, int accumulate);
