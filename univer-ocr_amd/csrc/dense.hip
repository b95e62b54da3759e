// FullyConnected (web_app/components/nn/layers/layers.py:307-363) as three strided GEMMs.
//   forward   y  = [x, 1] . w                     (layers.py:335-339)
//   backward  dx = dy . w[:-1]^T                  (layers.py:341-344)
//             dw += [x, 1]^T . dy                 (layers.py:345-346)
// The reference materialises [x, 1] with a concatenate per call; here the column of ones is
// virtual (a flag on the A operand), so x is read in place.
//
// This file holds the shape-generic LDS-tiled GEMM (float32 / float64, any strides); the f32 MFMA
// path for large shapes is dispatched in front of it (gemm_mfma.hip).
#include "gemm.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

// C[i,j] (+)= sum_p A(i,p) * B(p,j);  A(i,p) = a[i*a_rs + p*a_cs] except the virtual ones
// row/column, B(p,j) = b[p*b_rs + j*b_cs].  256 threads, 4x4 outputs per thread.
template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
    __shared__ T As[TK][TM + 4];
    __shared__ T Bs[TK][TN + 4];
    const T* __restrict__ a = (const T*)g.a;
    const T* __restrict__ b = (const T*)g.b;
    T* __restrict__ c = (T*)g.c;
    const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    T acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[r][s] = T(0);

    for (int p0 = 0; p0 < g.depth; p0 += TK) {
        // stage A tile (TM x TK) and B tile (TK x TN): 1024 elements each, 4 per thread
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int lin = threadIdx.x + e * 256;
            // A: pick the fast axis for coalescing
            int ai, ap;
            if (g.a_cs == 1) { ap = lin % TK; ai = lin / TK; } else { ai = lin % TM; ap = lin / TM; }
            const int gi = i0 + ai, gp = p0 + ap;
            T av = T(0);
            if (gi < g.m && gp < g.depth) {
                if ((g.a_ones_col && gp == g.depth - 1) || (g.a_ones_row && gi == g.m - 1)) av = T(1);
                else av = a[(size_t)gi * g.a_rs + (size_t)gp * g.a_cs];
            }
            As[ap][ai] = av;
            int bj, bp;
            if (g.b_cs == 1) { bj = lin % TN; bp = lin / TN; } else { bp = lin % TK; bj = lin / TK; }
            const int gj = j0 + bj, gq = p0 + bp;
            T bv = T(0);
            if (gj < g.n && gq < g.depth) bv = b[(size_t)gq * g.b_rs + (size_t)gj * g.b_cs];
            Bs[bp][bj] = bv;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < TK; ++p) {
            T av[4], bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = As[p][ty * 4 + r];
#pragma unroll
            for (int s = 0; s < 4; ++s) bv[s] = Bs[p][tx * 4 + s];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[r][s] += av[r] * bv[s];
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gi = i0 + ty * 4 + r;
        if (gi >= g.m) continue;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int gj = j0 + tx * 4 + s;
            if (gj >= g.n) continue;
            T* dst = c + (size_t)gi * g.ldc + gj;
            *dst = g.accumulate ? *dst + acc[r][s] : acc[r][s];
        }
    }
}

}  // namespace

int uocr_gemm_generic(uocr_ctx* ctx, int dtype, const GemmArgs& g) {
    const dim3 grid((g.n + TN - 1) / TN, (g.m + TM - 1) / TM), block(256);
    UOCR_DISPATCH(ctx, dtype, {
        hipLaunchKernelGGL((gemm_generic_kernel<T>), grid, block, 0, ctx->stream, g);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

extern "C" {

int uocr_dense_fwd_act(uocr_ctx* ctx, int dtype, const void* x, const void* w, void* y, int m, int n_in, int n_out,
                       int act, double act_alpha) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && w && y && m > 0 && n_in > 0 && n_out > 0);
    UOCR_REQUIRE(ctx, act >= UOCR_ACT_NONE && act <= UOCR_ACT_SIGMOID);
    GemmArgs g{};
    g.a = x; g.a_rs = n_in; g.a_cs = 1; g.a_ones_col = 1;
    g.b = w; g.b_rs = n_out; g.b_cs = 1;
    g.c = y; g.ldc = n_out;
    g.m = m; g.n = n_out; g.depth = n_in + 1;
    g.act = act; g.act_alpha = act_alpha;
    return uocr_gemm(ctx, dtype, g);
}

int uocr_dense_fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w, void* y, int m, int n_in, int n_out) {
    return uocr_dense_fwd_act(ctx, dtype, x, w, y, m, n_in, n_out, UOCR_ACT_NONE, 0.0);
}

int uocr_dense_bwd_act(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* dy, void* dx, void* dw,
                       int m, int n_in, int n_out, int accumulate, int x_act, double x_act_alpha) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && w && dy && (dx || dw) && m > 0 && n_in > 0 && n_out > 0);
    UOCR_REQUIRE(ctx, x_act == UOCR_ACT_NONE || x_act == UOCR_ACT_SIGMOID || (x_act == UOCR_ACT_LEAKY && x_act_alpha > 0));
    if (dx) {   // dx[m, n_in] = dy[m, n_out] . w[:n_in, :]^T  (* act'(x) when x is the output of a fused activation)
        GemmArgs g{};
        g.a = dy; g.a_rs = n_out; g.a_cs = 1;
        g.b = w; g.b_rs = 1; g.b_cs = n_out;
        g.c = dx; g.ldc = n_in;
        g.m = m; g.n = n_in; g.depth = n_out;
        if (x_act != UOCR_ACT_NONE) {
            g.mask_y = x; g.mask_act = x_act; g.mask_alpha = x_act_alpha;
        }
        int rc = uocr_gemm(ctx, dtype, g);
        if (rc) return rc;
    }
    if (!dw) return UOCR_OK;
    GemmArgs g{};   // dw[n_in + 1, n_out] (+)= [x, 1]^T . dy
    g.a = x; g.a_rs = 1; g.a_cs = n_in; g.a_ones_row = 1;
    g.b = dy; g.b_rs = n_out; g.b_cs = 1;
    g.c = dw; g.ldc = n_out;
    g.m = n_in + 1; g.n = n_out; g.depth = m;
    g.accumulate = accumulate;
    return uocr_gemm(ctx, dtype, g);
}

int uocr_dense_bwd(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* dy, void* dx, void* dw, int m,
                   int n_in, int n_out, int accumulate) {
    return uocr_dense_bwd_act(ctx, dtype, x, w, dy, dx, dw, m, n_in, n_out, accumulate, UOCR_ACT_NONE, 0.0);
}

}  // extern "C"
