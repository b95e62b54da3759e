// Internal entry points of the fused Monochrome block (not part of the C ABI): the column-strip kernels of
// conv_pair_strip.hip, called by uocr_conv_pair_fwd / uocr_conv_pair_bwd (conv_pair.hip).
#pragma once
#include "uocr_common.h"

int uocr_pair_strip_bwd_f32(uocr_ctx* ctx, const float* x, const float* y, const float* dy, const float* w1,
                            const float* b1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                            float* dx, int n, int h, int w, float pad1, int use_b1, int use_b2, float alpha,
                            bool sig, int accumulate, float unscale);

int uocr_pair_strip_fwd_f32(uocr_ctx* ctx, const float* x, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* y, int n, int h, int w, float pad1, int use_b1, int use_b2,
                            float alpha, int act2);

int uocr_pair_strip_bwd_f16(uocr_ctx* ctx, const void* x, const void* y, const void* dy, const float* w1,
                            const float* b1, const float* w2, float* dw1, float* db1, float* dw2, float* db2,
                            void* dx, int n, int h, int w, float pad1, int use_b1, int use_b2, float alpha,
                            bool sig, int accumulate, float unscale);

int uocr_pair_strip_fwd_f16(uocr_ctx* ctx, const void* x, const float* w1, const float* b1, const float* w2,
                            const float* b2, void* y, int n, int h, int w, float pad1, int use_b1, int use_b2,
                            float alpha, int act2);
