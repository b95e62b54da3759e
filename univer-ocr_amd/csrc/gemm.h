// Strided GEMM descriptor shared by dense.hip (generic LDS-tiled kernel) and gemm_mfma.hip.
#pragma once
#include "uocr_common.h"

struct GemmArgs {
    const void* a;      // A(i,p) = a[i*a_rs + p*a_cs]
    const void* b;      // B(p,j) = b[p*b_rs + j*b_cs]
    void* c;            // C(i,j) = c[i*ldc + j]
    long a_rs, a_cs, b_rs, b_cs, ldc;
    int m, n, depth;    // C is m x n, summed over depth
    int a_ones_col;     // A's LAST column (p == depth-1) is all ones and not stored  ([x,1], layers.py:336)
    int a_ones_row;     // A's LAST row (i == m-1) is all ones and not stored         ([x,1]^T, layers.py:345)
    int accumulate;     // C += instead of C =
    // fused activations of the dense layers (both zero / null in a plain GEMM; not combined with accumulate):
    int act;            // C = act(C)                      -- forward of FullyConnected + LeakyRelu / Sigmoid
    double act_alpha;
    const void* mask_y; // C *= act'(mask_y[i*ldc + j])    -- dx through the fused activation that produced the
    int mask_act;       //                                    layer's input (mask_y = that input)
    double mask_alpha;
};

int uocr_gemm_generic(uocr_ctx* ctx, int dtype, const GemmArgs& g);
bool uocr_gemm_mfma_eligible(uocr_ctx* ctx, int dtype, const GemmArgs& g);
int uocr_gemm_mfma(uocr_ctx* ctx, const GemmArgs& g);
// dispatcher: MFMA path when eligible, else generic
int uocr_gemm(uocr_ctx* ctx, int dtype, const GemmArgs& g);
// deferred weight-gradient group of a ctx (gemm_mfma.hip): freed with the ctx
void uocr_gemm_defer_free(uocr_ctx* ctx);
int uocr_gemm_defer_begin(uocr_ctx* ctx);
int uocr_gemm_defer_flush(uocr_ctx* ctx, int keep_open);
