// One launch for the finish kernels of a backward pass (finish_group.h) + the C-ABI entry points of the deferred
// weight-gradient group (uocr_wgrad_defer_begin / _flush; the GEMM half lives in gemm_mfma.hip).
#include <algorithm>

#include "finish_group.h"
#include "gemm.h"

namespace {

constexpr int FIN_MAX = 8;
constexpr size_t FIN_REGION_BYTES = 24u << 20;      // block partials of one net's backward pass: < 8 MB at 32 x 256 x 512

struct FinishGroupArgs {
    int count;
    FinishDesc d[FIN_MAX];
};

struct FinishDefer {
    bool on = false;
    int count = 0;
    size_t used = 0;
    float* region = nullptr;
    FinishGroupArgs args{};
};

inline FinishDefer* fin_of(uocr_ctx* ctx) { return (FinishDefer*)ctx->finish_defer; }

__device__ __forceinline__ void fin_out(float* dst, double s, bool live, float unscale, int accumulate) {
    s = live ? s * (double)unscale : 0.0;                  // UOCR_F16_SCALED(k): 2^-k, else 1
    *dst = accumulate ? (float)((double)*dst + s) : (float)s;
}

// block -> (finish q, 8 consecutive columns); 32 row segments per column, eight loads in flight per thread, segments
// added in order (the shape of pair_strip_finish: 256-thread blocks are schedulable inside the page step)
__global__ __launch_bounds__(256) void finish_group_kernel(FinishGroupArgs g) {
    constexpr int FC = 8, NSEG = 32;
    __shared__ double seg[NSEG][FC];
    int q = 0;
    while (q + 1 < g.count && (int)blockIdx.x >= g.d[q + 1].first_block) ++q;          // (block-uniform)
    const FinishDesc& d = g.d[q];
    const int o = threadIdx.x % FC, sg = threadIdx.x / FC, j = ((int)blockIdx.x - d.first_block) * FC + o;
    double s = 0.0;
    if (j < d.ncols) {
        const float* src = d.partial + (size_t)(j / d.group_cols) * d.group_stride + (j % d.group_cols);
        const int per = (d.nblocks + NSEG - 1) / NSEG, b0 = sg * per, b1 = min(d.nblocks, b0 + per);
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[(size_t)(b + k) * d.row_stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (double)v[k];
        }
        for (; b < b1; ++b) s += (double)src[(size_t)b * d.row_stride];
    }
    seg[sg][o] = s;
    __syncthreads();
    if (sg != 0 || j >= d.ncols) return;
#pragma unroll
    for (int k = 1; k < NSEG; ++k) s += seg[k][o];
    switch (d.kind) {
        case FIN_COLS: {
            const int ndw = d.p[0];
            if (j < ndw) fin_out(d.dw + j, s, true, d.unscale, d.accumulate);
            else fin_out(d.db + (j - ndw), s, d.use_bias, d.unscale, d.accumulate);
            break;
        }
        case FIN_TAPROWS: {
            const int NP = d.p[0], NW = d.p[1], NB = d.p[2];
            const int ky = j / NP, idx = j % NP;
            if (idx < NW) fin_out(d.dw + ky * NW + idx, s, true, d.unscale, d.accumulate);
            else if (ky == 0 && idx < NW + NB) fin_out(d.db + (idx - NW), s, d.use_bias, d.unscale, d.accumulate);
            break;
        }
        case FIN_FAST: {
            const int NP = d.p[0], NW = d.p[1], KW = d.p[2], CIN = d.p[3], COUT = d.p[4], KYR = d.p[5], COB = d.p[6];
            const int OCG = COUT / COB;
            const int grp = j / NP, a = j % NP;
            const int kyg = grp / OCG, ocg = grp % OCG;
            if (a < NW) {
                const int oc = a % COB, t = a / COB;
                const int c = t % CIN, tap = t / CIN;
                const int kx = tap % KW, kyl = tap / KW;
                fin_out(d.dw + (((size_t)(kyg * KYR + kyl) * KW + kx) * CIN + c) * COUT + ocg * COB + oc, s, true, d.unscale,
                        d.accumulate);
            } else if (kyg == 0 && a - NW < COB) {
                fin_out(d.db + ocg * COB + (a - NW), s, d.use_bias, d.unscale, d.accumulate);
            }
            break;
        }
        default: {                                            // FIN_PAIR
            if (j < 144) fin_out(d.dw + j, s, true, d.unscale, d.accumulate);                      // dW1^T[tap][ch]
            else if (j < 160) fin_out(d.db + (j - 144), s, d.use_bias, d.unscale, d.accumulate);   // row 9: the ones copy
            else if (j >= 256 && j < 256 + 144) fin_out(d.dw2 + (j - 256), s, true, d.unscale, d.accumulate);
            else if (j == 512) fin_out(d.db2, s, d.use_bias2, d.unscale, d.accumulate);
            break;
        }
    }
}

int fin_flush(uocr_ctx* ctx) {
    FinishDefer* f = fin_of(ctx);
    if (!f || f->count == 0) {
        if (f) f->used = 0;
        return UOCR_OK;
    }
    FinishGroupArgs& g = f->args;
    g.count = f->count;
    int blocks = 0;
    for (int q = 0; q < g.count; ++q) {
        g.d[q].first_block = blocks;
        blocks += (g.d[q].ncols + 7) / 8;
    }
    hipLaunchKernelGGL(finish_group_kernel, dim3(blocks), dim3(256), 0, ctx->stream, g);
    UOCR_LAUNCH_CHECK(ctx);
    f->count = 0;
    f->used = 0;              // (the stream orders the next group's producers behind this launch)
    return UOCR_OK;
}

}  // namespace

float* uocr_partial_buffer(uocr_ctx* ctx, size_t bytes, int* rc) {
    *rc = UOCR_OK;
    FinishDefer* f = fin_of(ctx);
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (f && f->on && f->region && f->count < FIN_MAX && f->used + need <= FIN_REGION_BYTES) {
        float* p = (float*)((char*)f->region + f->used);
        f->used += need;
        return p;
    }
    *rc = uocr_need_workspace(ctx, bytes);
    return *rc == UOCR_OK ? (float*)ctx->workspace : nullptr;
}

bool uocr_finish_defer(uocr_ctx* ctx, const FinishDesc& d) {
    FinishDefer* f = fin_of(ctx);
    if (!f || !f->on || !f->region || f->count >= FIN_MAX) return false;
    // only partials that live in the deferred region survive until the flush
    const char* p = (const char*)d.partial;
    if (p < (const char*)f->region || p >= (const char*)f->region + FIN_REGION_BYTES) return false;
    f->args.d[f->count++] = d;
    return true;
}

void uocr_finish_defer_free(uocr_ctx* ctx) {
    FinishDefer* f = fin_of(ctx);
    if (!f) return;
    if (f->region) hipFree(f->region);
    delete f;
    ctx->finish_defer = nullptr;
}

extern "C" int uocr_wgrad_defer_begin(uocr_ctx* ctx) {
    UOCR_CHECK_CTX(ctx);
    if (!ctx->finish_defer) {
        FinishDefer* f = new FinishDefer();
        if (hipMalloc((void**)&f->region, FIN_REGION_BYTES) != hipSuccess) f->region = nullptr;     // (finishes then stay separate)
        ctx->finish_defer = f;
    }
    FinishDefer* f = fin_of(ctx);
    if (f->on) UOCR_FAIL(ctx, UOCR_ERR_ARG, "uocr_wgrad_defer_begin: a deferred group is already open");
    int rc = uocr_gemm_defer_begin(ctx);
    if (rc) return rc;
    f->on = true;
    f->count = 0;
    f->used = 0;
    return UOCR_OK;
}

extern "C" int uocr_wgrad_defer_flush(uocr_ctx* ctx, int keep_open) {
    UOCR_CHECK_CTX(ctx);
    FinishDefer* f = fin_of(ctx);
    if (!f || !f->on) UOCR_FAIL(ctx, UOCR_ERR_ARG, "uocr_wgrad_defer_flush: no deferred group is open");
    int rc = uocr_gemm_defer_flush(ctx, keep_open);
    const int rc2 = fin_flush(ctx);
    if (rc == UOCR_OK) rc = rc2;
    f->on = keep_open != 0 && rc == UOCR_OK;
    return rc;
}
