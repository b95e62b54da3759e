// Convolutional2D on gfx950: shape-generic direct kernels (any kernel size / stride / padding /
// padding_value / channel count, float32 or float64).  These are the correctness baseline every
// specialised kernel is checked against; shape-specialised fast paths are dispatched in front of
// them by uocr_conv2d_* (see conv_fast.hip).
//
// Reference (paths relative to web_app/components/nn/layers/):
//   forward ........ convolutional.py:62-99   (GPU kernel :153-195)
//   backward dx .... convolutional.py:101-145 (GPU kernel :203-219, 239-250)
//   backward dw/db . convolutional.py:116-138 (GPU kernel :221-237, 252-265, 274-284)
// Layout: x (n,h,w,cin), w (kh,kw,cin,cout), y/dy (n,oh,ow,cout), all C-contiguous.
#include "conv_dims.h"

namespace {

template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
    T v[V];
};

template <typename T>
__device__ __forceinline__ T dev_exp_(T x);
template <>
__device__ __forceinline__ float dev_exp_<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double dev_exp_<double>(double x) { return exp(x); }

template <typename T>
__device__ __forceinline__ T apply_act(T v, int act, T alpha) {
    switch (act) {
        case UOCR_ACT_RELU: return v * (v >= T(0) ? T(1) : T(0));
        case UOCR_ACT_LEAKY: return v * ((v >= T(0) ? T(1) : T(0)) + alpha * (v < T(0) ? T(1) : T(0)));
        case UOCR_ACT_SIGMOID: return T(1) / (T(1) + dev_exp_<T>(-v));
        default: return v;
    }
}

// thread = (output pixel, block of CB consecutive output channels); channel block fastest, so a
// wave stores 64*CB contiguous elements.  Sum order = (ky, kx, ic), bias last: the order of the
// reference's [patch, 1].[w; b] dot product (convolutional.py:92-95).
// TS = activation storage type (T itself, or _Float16 with T = float: UOCR_F16), T = arithmetic / parameter type
template <typename TS, typename T, int CB>
__global__ __launch_bounds__(256) void conv_fwd_generic(const TS* __restrict__ x, const T* __restrict__ w,
                                                        const T* __restrict__ bias, TS* __restrict__ y, ConvDims d,
                                                        T pad_value, int use_bias, int act, T act_alpha) {
    const int nob = d.cout / CB;
    const size_t total = (size_t)d.n * d.oh * d.ow * nob;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int oc0 = (int)(idx % nob) * CB;
        size_t pix = idx / nob;
        const int ox = (int)(pix % d.ow);
        size_t t = pix / d.ow;
        const int oy = (int)(t % d.oh);
        const int b = (int)(t / d.oh);
        T acc[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[j] = T(0);
        const int iy0 = oy * d.sh - d.ph, ix0 = ox * d.sw - d.pw;
        for (int ky = 0; ky < d.kh; ++ky) {
            const int iy = iy0 + ky;
            for (int kx = 0; kx < d.kw; ++kx) {
                const int ix = ix0 + kx;
                const bool inside = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
                const TS* xp = x + (((size_t)b * d.h + iy) * d.w + ix) * d.cin;
                const T* wp = w + ((size_t)(ky * d.kw + kx) * d.cin) * d.cout + oc0;
                for (int ic = 0; ic < d.cin; ++ic) {
                    const T xv = inside ? (T)xp[ic] : pad_value;
#pragma unroll
                    for (int j = 0; j < CB; ++j) acc[j] += xv * wp[(size_t)ic * d.cout + j];
                }
            }
        }
        Pack<TS, CB> out;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            T v = acc[j];
            if (use_bias) v += bias[oc0 + j];
            out.v[j] = (TS)apply_act(v, act, act_alpha);
        }
        *reinterpret_cast<Pack<TS, CB>*>(y + pix * d.cout + oc0) = out;
    }
}

// thread = (input pixel, block of CB input channels).  dx[b,y,x,ic] = sum over the output pixels
// whose window covers (y,x): dy[b,gy,gx,:] . w[ky,kx,ic,:].  Windows are visited in raster order
// (gy, gx ascending = ky, kx descending), the order the reference scatter-adds them (:121-134).
template <typename TS, typename T, int CB>
__global__ __launch_bounds__(256) void conv_dgrad_generic(const TS* __restrict__ dy, const T* __restrict__ w,
                                                          TS* __restrict__ dx, ConvDims d, const TS* __restrict__ mask_y,
                                                          int mask_act, T mask_alpha) {
    const int nib = d.cin / CB;
    const size_t total = (size_t)d.n * d.h * d.w * nib;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ic0 = (int)(idx % nib) * CB;
        size_t pix = idx / nib;
        const int ix = (int)(pix % d.w);
        size_t t = pix / d.w;
        const int iy = (int)(t % d.h);
        const int b = (int)(t / d.h);
        T acc[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[j] = T(0);
        for (int ky = d.kh - 1; ky >= 0; --ky) {
            const int ty = iy + d.ph - ky;
            if (ty < 0 || ty % d.sh) continue;
            const int gy = ty / d.sh;
            if (gy >= d.oh) continue;
            for (int kx = d.kw - 1; kx >= 0; --kx) {
                const int tx = ix + d.pw - kx;
                if (tx < 0 || tx % d.sw) continue;
                const int gx = tx / d.sw;
                if (gx >= d.ow) continue;
                const TS* gp = dy + (((size_t)b * d.oh + gy) * d.ow + gx) * d.cout;
                const T* wp = w + ((size_t)(ky * d.kw + kx) * d.cin + ic0) * d.cout;
                for (int oc = 0; oc < d.cout; ++oc) {
                    const T g = (T)gp[oc];
#pragma unroll
                    for (int j = 0; j < CB; ++j) acc[j] += g * wp[(size_t)j * d.cout + oc];
                }
            }
        }
        Pack<TS, CB> out;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            T v = acc[j];
            if (mask_act != UOCR_ACT_NONE)
                v *= act_grad_from_output<T>((T)mask_y[pix * d.cin + ic0 + j], mask_act, mask_alpha);
            out.v[j] = (TS)v;
        }
        *reinterpret_cast<Pack<TS, CB>*>(dx + pix * d.cin + ic0) = out;
    }
}

// stage 1 of dw/db: block `blk` owns output pixels [p0,p1); thread owns pair q = (k, oc) with
// k = (ky,kx,ic) flattened, k == K = the bias row (its "x" is 1, convolutional.py:125).
// float64 accumulation whatever T is: the sum runs over n*oh*ow (4.2 M at 32x256x512) terms.
template <typename TS, typename T>
__global__ __launch_bounds__(256) void conv_wgrad_partial_generic(const TS* __restrict__ x, const TS* __restrict__ dy,
                                                                  double* __restrict__ partial, ConvDims d,
                                                                  T pad_value, int npairs, int pix_per_block) {
    const size_t npix = (size_t)d.n * d.oh * d.ow;
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    const size_t p1 = min(npix, p0 + (size_t)pix_per_block);
    const int K = d.kh * d.kw * d.cin;
    for (int q = threadIdx.x; q < npairs; q += blockDim.x) {
        const int k = q / d.cout, oc = q % d.cout;
        const bool is_bias = (k == K);
        const int ic = k % d.cin;
        const int kx = (k / d.cin) % d.kw;
        const int ky = k / (d.cin * d.kw);
        int ox = (int)(p0 % d.ow);
        size_t t = p0 / d.ow;
        int oy = (int)(t % d.oh);
        int b = (int)(t / d.oh);
        double acc = 0.0;
        for (size_t p = p0; p < p1; ++p) {
            const double g = (double)dy[p * d.cout + oc];
            double xv = 1.0;
            if (!is_bias) {
                const int iy = oy * d.sh - d.ph + ky, ix = ox * d.sw - d.pw + kx;
                const bool inside = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
                xv = inside ? (double)x[(((size_t)b * d.h + iy) * d.w + ix) * d.cin + ic] : (double)pad_value;
            }
            acc += xv * g;
            if (++ox == d.ow) {
                ox = 0;
                if (++oy == d.oh) {
                    oy = 0;
                    ++b;
                }
            }
        }
        partial[(size_t)blockIdx.x * npairs + q] = acc;
    }
}

// stage 2: dw[q] / db[q-K*cout] (+)= sum over blocks, fixed order (deterministic)
template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_finish(const double* __restrict__ partial, T* __restrict__ dw,
                                                         T* __restrict__ db, int npairs, int nblocks, int kc,
                                                         int accumulate, double unscale) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= npairs) return;
    double acc = 0.0;
    for (int blk = 0; blk < nblocks; ++blk) acc += partial[(size_t)blk * npairs + q];
    acc *= unscale;                                      // UOCR_F16_SCALED(k): 2^-k, else 1
    T* dst = q < kc ? dw + q : db + (q - kc);
    *dst = accumulate ? (T)((double)*dst + acc) : (T)acc;
}

template <typename TS, typename T>
int fwd_generic(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                double pad_value, int use_bias, int act, double act_alpha) {
    int cb = (d.cout % 4 == 0) ? 4 : (d.cout % 2 == 0 ? 2 : 1);
    while (cb > 1 && (reinterpret_cast<uintptr_t>(y) % (sizeof(TS) * cb))) cb >>= 1;
    const size_t total = (size_t)d.n * d.oh * d.ow * (d.cout / cb);
    const dim3 grid(uocr_blocks_for(total, 256, 1u << 20)), block(256);
#define LAUNCH_FWD(CB)                                                                                      \
    hipLaunchKernelGGL((conv_fwd_generic<TS, T, CB>), grid, block, 0, ctx->stream, (const TS*)x, (const T*)w, \
                       (const T*)b, (TS*)y, d, (T)pad_value, use_bias, act, (T)act_alpha)
    if (cb == 4) LAUNCH_FWD(4);
    else if (cb == 2) LAUNCH_FWD(2);
    else LAUNCH_FWD(1);
#undef LAUNCH_FWD
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

template <typename TS, typename T>
int dgrad_generic(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d, const ActMask& mask) {
    int cb = (d.cin % 4 == 0) ? 4 : (d.cin % 2 == 0 ? 2 : 1);
    while (cb > 1 && (reinterpret_cast<uintptr_t>(dx) % (sizeof(TS) * cb))) cb >>= 1;
    const size_t total = (size_t)d.n * d.h * d.w * (d.cin / cb);
    const dim3 grid(uocr_blocks_for(total, 256, 1u << 20)), block(256);
#define LAUNCH_DG(CB) \
    hipLaunchKernelGGL((conv_dgrad_generic<TS, T, CB>), grid, block, 0, ctx->stream, (const TS*)dy, (const T*)w, (TS*)dx, d, \
                       (const TS*)mask.y, mask.act, (T)mask.alpha)
    if (cb == 4) LAUNCH_DG(4);
    else if (cb == 2) LAUNCH_DG(2);
    else LAUNCH_DG(1);
#undef LAUNCH_DG
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

template <typename TS, typename T>
int wgrad_generic(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                  double pad_value, int use_bias, int accumulate) {
    const int K = d.kh * d.kw * d.cin;
    const int kc = K * d.cout;
    const int npairs = kc + (use_bias ? d.cout : 0);
    const size_t npix = (size_t)d.n * d.oh * d.ow;
    // blocks: ~512 pixels each, at most 2048, partial buffer at most half the workspace
    size_t nblk = (npix + 511) / 512;
    if (nblk > 2048) nblk = 2048;
    const size_t cap = (ctx->workspace_bytes / 2) / ((size_t)npairs * sizeof(double));
    if (cap < 1) return uocr_need_workspace(ctx, 2 * (size_t)npairs * sizeof(double));
    if (nblk > cap) nblk = cap;
    const int ppb = (int)((npix + nblk - 1) / nblk);
    nblk = (npix + ppb - 1) / ppb;
    double* partial = (double*)ctx->workspace;
    hipLaunchKernelGGL((conv_wgrad_partial_generic<TS, T>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream,
                       (const TS*)x, (const TS*)dy, partial, d, (T)pad_value, npairs, ppb);
    UOCR_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL((conv_wgrad_finish<T>), dim3((npairs + 255) / 256), dim3(256), 0, ctx->stream,
                       (const double*)partial, (T*)dw, (T*)db, npairs, (int)nblk, kc, accumulate,
                       uocr_grad_unscale(dtype));
    UOCR_LAUNCH_CHECK(ctx);
    if (!use_bias && !accumulate && db)
        UOCR_HIP(ctx, hipMemsetAsync(db, 0, (size_t)d.cout * sizeof(T), ctx->stream));
    return UOCR_OK;
}

}  // namespace

int uocr_conv_fwd_generic(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y,
                          const ConvDims& d, double pad_value, int use_bias, int act, double act_alpha) {
    UOCR_DISPATCH_ACT(ctx, dtype, { return fwd_generic<TS, T>(ctx, x, w, b, y, d, pad_value, use_bias, act, act_alpha); });
    return UOCR_OK;
}

int uocr_conv_dgrad_generic(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx, const ConvDims& d,
                            const ActMask& mask) {
    UOCR_DISPATCH_ACT(ctx, dtype, { return dgrad_generic<TS, T>(ctx, dy, w, dx, d, mask); });
    return UOCR_OK;
}

int uocr_conv_wgrad_generic(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db,
                            const ConvDims& d, double pad_value, int use_bias, int accumulate) {
    UOCR_DISPATCH_ACT(ctx, dtype, {
        return wgrad_generic<TS, T>(ctx, dtype, x, dy, dw, db, d, pad_value, use_bias, accumulate);
    });
    return UOCR_OK;
}
