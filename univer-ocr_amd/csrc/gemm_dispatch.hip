// GEMM dispatch: f32 MFMA kernel for shapes that fill its tiles, generic LDS-tiled kernel otherwise.
#include "gemm.h"

#include "univer_hip.h"

int uocr_gemm(uocr_ctx* ctx, int dtype, const GemmArgs& g) {
    if (uocr_gemm_mfma_eligible(ctx, dtype, g)) return uocr_gemm_mfma(ctx, g);   // activations in its epilogue
    int rc = uocr_gemm_generic(ctx, dtype, g);
    if (rc) return rc;
    // the generic kernel has no epilogue: fused activations become elementwise passes over C (ldc == n there)
    const size_t count = (size_t)g.m * g.n;
    if (g.act != UOCR_ACT_NONE) {
        UOCR_REQUIRE(ctx, g.ldc == g.n && !g.accumulate);
        rc = uocr_act_fwd(ctx, dtype, g.act, g.act_alpha, g.c, g.c, count);
        if (rc) return rc;
    }
    if (g.mask_act != UOCR_ACT_NONE) {
        UOCR_REQUIRE(ctx, g.ldc == g.n && !g.accumulate && g.mask_y);
        rc = uocr_act_bwd_from_output(ctx, dtype, g.mask_act, g.mask_alpha, g.mask_y, g.c, g.c, count);
    }
    return rc;
}
