// GEMM dispatch: f32 MFMA kernel for shapes that fill its tiles, generic LDS-tiled kernel otherwise.
#include "gemm.h"

int uocr_gemm(uocr_ctx* ctx, int dtype, const GemmArgs& g) {
    if (uocr_gemm_mfma_eligible(ctx, dtype, g)) return uocr_gemm_mfma(ctx, g);
    return uocr_gemm_generic(ctx, dtype, g);
}
