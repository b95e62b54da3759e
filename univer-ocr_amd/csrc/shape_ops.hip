// HBM-bound index-shuffling kernels: MaxPool2D, Upsample2D, Conv2DToBatchedFixedWidthed, Concat.
// Roofline: HBM bandwidth.  One thread per OUTPUT element of the pass (gather formulation, so no
// atomics), channel index fastest so a wave's accesses are contiguous in NHWC.
//
// Reference semantics restated (paths relative to web_app/components/nn/layers/):
//   MaxPool2D .................... maxpool.py:24-90 (NumPy path; numba path :92-202 differs on
//                                  all-negative partial windows -- the NumPy path is the oracle)
//   Upsample2D ................... upsample.py:21-110
//   Conv2DToBatchedFixedWidthed .. convolutional.py:330-373
//   Concat ....................... layers.py:240-284
#include "uocr_common.h"

namespace {

struct PoolDims {
    int n, h, w, c, kh, kw, sh, sw, ph, pw, oh, ow;
};

// y = max over the window clipped to the zero-padded extent [0,h+2ph) x [0,w+2pw); padded cells
// count as 0 (maxpool.py:36-40); the mask cell (kh x kw bytes per window, window-major) marks
// every position equal to the max, positions outside the padded extent stay 0 (:47-53).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ mask, PoolDims d) {
    const size_t total = (size_t)d.n * d.oh * d.ow * d.c;
    const int hp = d.h + 2 * d.ph, wp = d.w + 2 * d.pw;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % d.c);
        size_t t = idx / d.c;
        const int ox = (int)(t % d.ow);
        t /= d.ow;
        const int oy = (int)(t % d.oh);
        const int b = (int)(t / d.oh);
        const int py0 = oy * d.sh, px0 = ox * d.sw;   // padded coordinates
        T best = T(0);
        bool any = false;
        for (int ky = 0; ky < d.kh; ++ky) {
            const int py = py0 + ky;
            if (py >= hp) break;
            const int iy = py - d.ph;
            for (int kx = 0; kx < d.kw; ++kx) {
                const int px = px0 + kx;
                if (px >= wp) break;
                const int ix = px - d.pw;
                const bool inside = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
                const T v = inside ? x[(((size_t)b * d.h + iy) * d.w + ix) * d.c + ch] : T(0);
                if (!any) best = v;
                else if (best == best && (v > best || v != v)) best = v;   // NaN sticks, as np.max
                any = true;
            }
        }
        y[idx] = best;
        for (int ky = 0; ky < d.kh; ++ky) {
            const int py = py0 + ky, iy = py - d.ph;
            for (int kx = 0; kx < d.kw; ++kx) {
                const int px = px0 + kx, ix = px - d.pw;
                uint8_t m = 0;
                if (py < hp && px < wp) {
                    const bool inside = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
                    const T v = inside ? x[(((size_t)b * d.h + iy) * d.w + ix) * d.c + ch] : T(0);
                    m = (v == best) ? 1 : 0;
                }
                mask[(((size_t)b * d.oh * d.kh + (size_t)oy * d.kh + ky) * ((size_t)d.ow * d.kw) +
                      (size_t)ox * d.kw + kx) * d.c + ch] = m;
            }
        }
    }
}

// dx[b,y,x,c] = sum over windows (oy,ox) covering (y,x) whose mask bit at that position is set of
// dy[b,oy,ox,c] / (#set bits of the window).  Windows are visited in raster order (oy, ox
// ascending) = the order in which the reference accumulates (maxpool.py:73-83).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                          T* __restrict__ dx, PoolDims d) {
    const size_t total = (size_t)d.n * d.h * d.w * d.c;
    const size_t mrow = (size_t)d.ow * d.kw * d.c;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % d.c);
        size_t t = idx / d.c;
        const int ix = (int)(t % d.w);
        t /= d.w;
        const int iy = (int)(t % d.h);
        const int b = (int)(t / d.h);
        const int py = iy + d.ph, px = ix + d.pw;
        T acc = T(0);
        for (int ky = d.kh - 1; ky >= 0; --ky) {
            const int ty = py - ky;
            if (ty < 0 || ty % d.sh) continue;
            const int oy = ty / d.sh;
            if (oy >= d.oh) continue;
            for (int kx = d.kw - 1; kx >= 0; --kx) {
                const int tx = px - kx;
                if (tx < 0 || tx % d.sw) continue;
                const int ox = tx / d.sw;
                if (ox >= d.ow) continue;
                const uint8_t* cell = mask + ((size_t)b * d.oh * d.kh + (size_t)oy * d.kh) * mrow +
                                      (size_t)ox * d.kw * d.c + ch;
                if (!cell[(size_t)ky * mrow + (size_t)kx * d.c]) continue;
                int cnt = 0;
                for (int a = 0; a < d.kh; ++a)
                    for (int e = 0; e < d.kw; ++e) cnt += cell[(size_t)a * mrow + (size_t)e * d.c];
                acc += dy[(((size_t)b * d.oh + oy) * d.ow + ox) * d.c + ch] / (T)cnt;
            }
        }
        dx[idx] = acc;
    }
}

// 2x2 / stride 2 / no padding / even h, w / c % 4 == 0 (float32): the windows tile the image, so the mask
// has exactly the layout of x (mask[b][2oy+ky][2ox+kx][ch]) and every input pixel belongs to one window.
// One thread per (window, channel quad): 4 x 16-B loads, one 16-B store of y, four 4-B mask stores;
// the same comparison chain as maxpool_fwd_kernel (NaN sticks).
__device__ __forceinline__ float pool_max4(float a, float b, float c, float d) {
    float best = a;
    if (best == best && (b > best || b != b)) best = b;
    if (best == best && (c > best || c != c)) best = c;
    if (best == best && (d > best || d != d)) best = d;
    return best;
}

__device__ __forceinline__ uint32_t pool_mask4(const float4& v, const float4& m) {
    return (v.x == m.x ? 1u : 0u) | (v.y == m.y ? 0x100u : 0u) | (v.z == m.z ? 0x10000u : 0u) |
           (v.w == m.w ? 0x1000000u : 0u);
}

__global__ __launch_bounds__(256) void maxpool2_fwd_vec(const float4* __restrict__ x, float4* __restrict__ y,
                                                        uint32_t* __restrict__ mask, size_t windows, int ow, int cq) {
    // windows = n*oh*ow*cq; x / mask rows hold 2*ow*cq quads
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < windows;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cq);
        const size_t t = idx / cq;
        const int ox = (int)(t % ow);
        const size_t row = t / ow;                      // b*oh + oy
        const size_t rq = (size_t)2 * ow * cq;          // quads per input row
        const size_t i00 = (row * 2) * rq + (size_t)(2 * ox) * cq + q;
        const float4 a = x[i00], b = x[i00 + cq], c = x[i00 + rq], d = x[i00 + rq + cq];
        float4 m;
        m.x = pool_max4(a.x, b.x, c.x, d.x);
        m.y = pool_max4(a.y, b.y, c.y, d.y);
        m.z = pool_max4(a.z, b.z, c.z, d.z);
        m.w = pool_max4(a.w, b.w, c.w, d.w);
        y[idx] = m;
        mask[i00] = pool_mask4(a, m);
        mask[i00 + cq] = pool_mask4(b, m);
        mask[i00 + rq] = pool_mask4(c, m);
        mask[i00 + rq + cq] = pool_mask4(d, m);
    }
}

__device__ __forceinline__ float4 pool_scatter4(uint32_t m, const float4& g) {
    return make_float4((m & 0xffu) ? g.x : 0.f, (m & 0xff00u) ? g.y : 0.f, (m & 0xff0000u) ? g.z : 0.f,
                       (m & 0xff000000u) ? g.w : 0.f);
}

__global__ __launch_bounds__(256) void maxpool2_bwd_vec(const float4* __restrict__ dy, const uint32_t* __restrict__ mask,
                                                        float4* __restrict__ dx, size_t windows, int ow, int cq) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < windows;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(idx % cq);
        const size_t t = idx / cq;
        const int ox = (int)(t % ow);
        const size_t row = t / ow;
        const size_t rq = (size_t)2 * ow * cq;
        const size_t i00 = (row * 2) * rq + (size_t)(2 * ox) * cq + q;
        const uint32_t ma = mask[i00], mb = mask[i00 + cq], mc = mask[i00 + rq], md = mask[i00 + rq + cq];
        const uint32_t cnt = ma + mb + mc + md;         // per byte: 1..4 set bits (the max itself is always set)
        float4 g = dy[idx];
        g.x /= (float)(cnt & 0xffu);
        g.y /= (float)((cnt >> 8) & 0xffu);
        g.z /= (float)((cnt >> 16) & 0xffu);
        g.w /= (float)(cnt >> 24);
        dx[i00] = pool_scatter4(ma, g);
        dx[i00 + cq] = pool_scatter4(mb, g);
        dx[i00 + rq] = pool_scatter4(mc, g);
        dx[i00 + rq + cq] = pool_scatter4(md, g);
    }
}

bool maxpool2_vec_ok(int dtype, const PoolDims& d, const void* a, const void* b, const void* c) {
    return dtype == UOCR_F32 && d.kh == 2 && d.kw == 2 && d.sh == 2 && d.sw == 2 && d.ph == 0 && d.pw == 0 &&
           d.h == 2 * d.oh && d.w == 2 * d.ow && d.c % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 &&
           ((uintptr_t)c % 4) == 0;
}

// y[b, Y, X, c] = x[b, Y / sy, X / sx, c]   (upsample.py:21-25)
template <typename T>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h,
                                                           int w, int c, int sy, int sx) {
    const int oh = h * sy, ow = w * sx;
    const size_t total = (size_t)n * oh * ow * c;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        size_t t = idx / c;
        const int X = (int)(t % ow);
        t /= ow;
        const int Y = (int)(t % oh);
        const int b = (int)(t / oh);
        y[idx] = x[(((size_t)b * h + Y / sy) * w + X / sx) * c + ch];
    }
}

// dx[b,y,x,c] = sum_{j<sy, i<sx} dy[b, y*sy + j, x*sx + i, c]   (upsample.py:27-39)
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int n, int h,
                                                           int w, int c, int sy, int sx) {
    const int ow = w * sx, oh = h * sy;
    const size_t total = (size_t)n * h * w * c;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        size_t t = idx / c;
        const int ix = (int)(t % w);
        t /= w;
        const int iy = (int)(t % h);
        const int b = (int)(t / h);
        T acc = T(0);
        for (int j = 0; j < sy; ++j)
            for (int i = 0; i < sx; ++i)
                acc += dy[(((size_t)b * oh + iy * sy + j) * ow + ix * sx + i) * c + ch];
        dx[idx] = acc;
    }
}

// 2x2 nearest upsampling of the my_model decoders (float32, 1 or 4 channels): one thread per 16 B of input
// -- 4 pixels of a 1-channel row or one 4-channel pixel -- instead of one thread (and a 64-bit div/mod
// chain) per output element.  `quads` = n*h*w*c/4.
template <int CH>
__global__ __launch_bounds__(256) void upsample2_fwd_vec(const float4* __restrict__ x, float4* __restrict__ y,
                                                         size_t quads, int h, int row_quads) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < quads;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t row = idx / row_quads;              // b*h + iy
        const int k = (int)(idx - row * row_quads);
        const float4 v = x[idx];
        float4 lo, hi;
        if constexpr (CH == 4) {
            lo = v;
            hi = v;
        } else {
            lo = make_float4(v.x, v.x, v.y, v.y);
            hi = make_float4(v.z, v.z, v.w, v.w);
        }
        float4* out = y + (row * 2) * (size_t)(2 * row_quads) + 2 * k;     // output row 2*(b*h + iy)
        out[0] = lo;
        out[1] = hi;
        out[2 * row_quads] = lo;
        out[2 * row_quads + 1] = hi;
    }
}

// same summation order as upsample_bwd_kernel: ((dy00 + dy01) + dy10) + dy11 starting from 0
template <int CH>
__global__ __launch_bounds__(256) void upsample2_bwd_vec(const float4* __restrict__ dy, float4* __restrict__ dx,
                                                         size_t quads, int h, int row_quads) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < quads;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t row = idx / row_quads;
        const int k = (int)(idx - row * row_quads);
        const float4* in = dy + (row * 2) * (size_t)(2 * row_quads) + 2 * k;
        const float4 a0 = in[0], a1 = in[1], b0 = in[2 * row_quads], b1 = in[2 * row_quads + 1];
        float4 r;
        if constexpr (CH == 4) {
            r.x = ((0.f + a0.x) + a1.x + b0.x) + b1.x;
            r.y = ((0.f + a0.y) + a1.y + b0.y) + b1.y;
            r.z = ((0.f + a0.z) + a1.z + b0.z) + b1.z;
            r.w = ((0.f + a0.w) + a1.w + b0.w) + b1.w;
        } else {
            r.x = ((0.f + a0.x) + a0.y + b0.x) + b0.y;
            r.y = ((0.f + a0.z) + a0.w + b0.z) + b0.w;
            r.z = ((0.f + a1.x) + a1.y + b1.x) + b1.y;
            r.w = ((0.f + a1.z) + a1.w + b1.z) + b1.w;
        }
        dx[idx] = r;
    }
}

bool upsample2_vec_ok(int dtype, const void* a, const void* b, int wd, int c, int sy, int sx) {
    return dtype == UOCR_F32 && sy == 2 && sx == 2 && (c == 1 || c == 4) && (wd * c) % 4 == 0 &&
           ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0;
}

// y[(b*w + col), r, j, c] = xpad[b, r, col + j, c], xpad = x shifted right by width/2 inside a
// zero row of length w + width (convolutional.py:335-349)
template <typename T>
__global__ __launch_bounds__(256) void fixed_width_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h,
                                                              int w, int c, int width) {
    const int hw = width / 2;
    const size_t total = (size_t)n * w * h * width * c;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        size_t t = idx / c;
        const int j = (int)(t % width);
        t /= width;
        const int r = (int)(t % h);
        t /= h;
        const int col = (int)(t % w);
        const int b = (int)(t / w);
        const int src = col + j - hw;
        y[idx] = (src >= 0 && src < w) ? x[(((size_t)b * h + r) * w + src) * c + ch] : T(0);
    }
}

// dx[b, r, xcol, c] = sum_j dy[(b*w + (xcol + hw - j)), r, j, c] over windows inside [0, w)
// (convolutional.py:351-362; the reference scatter-adds window by window, col ascending = j descending)
template <typename T>
__global__ __launch_bounds__(256) void fixed_width_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int n,
                                                              int h, int w, int c, int width) {
    const int hw = width / 2;
    const size_t total = (size_t)n * h * w * c;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % c);
        size_t t = idx / c;
        const int xc = (int)(t % w);
        t /= w;
        const int r = (int)(t % h);
        const int b = (int)(t / h);
        T acc = T(0);
        for (int j = width - 1; j >= 0; --j) {
            const int col = xc + hw - j;
            if (col < 0 || col >= w) continue;
            acc += dy[((((size_t)b * w + col) * h + r) * width + j) * c + ch];
        }
        dx[idx] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void copy2d_kernel(T* __restrict__ dst, size_t dst_ld, const T* __restrict__ src,
                                                     size_t src_ld, size_t rows, size_t cols) {
    const size_t total = rows * cols;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t r = idx / cols, cc = idx % cols;
        dst[r * dst_ld + cc] = src[r * src_ld + cc];
    }
}

inline bool pool_dims_ok(const PoolDims& d) {
    return d.n > 0 && d.h > 0 && d.w > 0 && d.c > 0 && d.kh > 0 && d.kw > 0 && d.sh > 0 && d.sw > 0 && d.ph >= 0 &&
           d.pw >= 0 && d.oh > 0 && d.ow > 0;
}

}  // namespace

extern "C" {

int uocr_maxpool2d_fwd(uocr_ctx* ctx, int dtype, const void* x, void* y, uint8_t* mask, int n, int h, int wd, int c,
                       int kh, int kw, int sh, int sw, int ph, int pw, int oh, int ow) {
    UOCR_CHECK_CTX(ctx);
    const PoolDims d{n, h, wd, c, kh, kw, sh, sw, ph, pw, oh, ow};
    UOCR_REQUIRE(ctx, x && y && mask && pool_dims_ok(d));
    // every window must start inside the padded extent (true for maxpool.py:204-216 shapes)
    UOCR_REQUIRE(ctx, (oh - 1) * sh < h + 2 * ph && (ow - 1) * sw < wd + 2 * pw);
    const size_t total = (size_t)n * oh * ow * c;
    if (maxpool2_vec_ok(dtype, d, x, y, mask)) {
        const size_t windows = total / 4;
        hipLaunchKernelGGL(maxpool2_fwd_vec, dim3(uocr_blocks_for(windows, 256, UOCR_MAX_GRID * 4)), dim3(256), 0,
                           ctx->stream, (const float4*)x, (float4*)y, (uint32_t*)mask, windows, ow, c / 4);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(uocr_blocks_for(total, 256, UOCR_MAX_GRID * 4)), dim3(256), 0,
                           ctx->stream, (const T*)x, (T*)y, mask, d);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_maxpool2d_bwd(uocr_ctx* ctx, int dtype, const void* dy, const uint8_t* mask, void* dx, int n, int h, int wd,
                       int c, int kh, int kw, int sh, int sw, int ph, int pw, int oh, int ow) {
    UOCR_CHECK_CTX(ctx);
    const PoolDims d{n, h, wd, c, kh, kw, sh, sw, ph, pw, oh, ow};
    UOCR_REQUIRE(ctx, dy && dx && mask && pool_dims_ok(d));
    const size_t total = (size_t)n * h * wd * c;
    if (maxpool2_vec_ok(dtype, d, dy, dx, mask)) {
        const size_t windows = (size_t)n * oh * ow * c / 4;
        hipLaunchKernelGGL(maxpool2_bwd_vec, dim3(uocr_blocks_for(windows, 256, UOCR_MAX_GRID * 4)), dim3(256), 0,
                           ctx->stream, (const float4*)dy, (const uint32_t*)mask, (float4*)dx, windows, ow, c / 4);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(uocr_blocks_for(total, 256, UOCR_MAX_GRID * 4)), dim3(256), 0,
                           ctx->stream, (const T*)dy, mask, (T*)dx, d);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_upsample2d_fwd(uocr_ctx* ctx, int dtype, const void* x, void* y, int n, int h, int wd, int c, int sy,
                        int sx) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && y && n > 0 && h > 0 && wd > 0 && c > 0 && sy > 0 && sx > 0);
    const size_t total = (size_t)n * h * sy * wd * sx * c;
    if (upsample2_vec_ok(dtype, x, y, wd, c, sy, sx)) {
        const size_t quads = (size_t)n * h * wd * c / 4;
        const dim3 grid(uocr_blocks_for(quads, 256, UOCR_MAX_GRID * 4));
        if (c == 4)
            hipLaunchKernelGGL((upsample2_fwd_vec<4>), grid, dim3(256), 0, ctx->stream, (const float4*)x, (float4*)y,
                               quads, h, wd * c / 4);
        else
            hipLaunchKernelGGL((upsample2_fwd_vec<1>), grid, dim3(256), 0, ctx->stream, (const float4*)x, (float4*)y,
                               quads, h, wd * c / 4);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((upsample_fwd_kernel<T>), dim3(uocr_blocks_for(total, 256, UOCR_MAX_GRID * 4)), dim3(256),
                           0, ctx->stream, (const T*)x, (T*)y, n, h, wd, c, sy, sx);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_upsample2d_bwd(uocr_ctx* ctx, int dtype, const void* dy, void* dx, int n, int h, int wd, int c, int sy,
                        int sx) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, dy && dx && n > 0 && h > 0 && wd > 0 && c > 0 && sy > 0 && sx > 0);
    const size_t total = (size_t)n * h * wd * c;
    if (upsample2_vec_ok(dtype, dy, dx, wd, c, sy, sx)) {
        const size_t quads = total / 4;
        const dim3 grid(uocr_blocks_for(quads, 256, UOCR_MAX_GRID * 4));
        if (c == 4)
            hipLaunchKernelGGL((upsample2_bwd_vec<4>), grid, dim3(256), 0, ctx->stream, (const float4*)dy, (float4*)dx,
                               quads, h, wd * c / 4);
        else
            hipLaunchKernelGGL((upsample2_bwd_vec<1>), grid, dim3(256), 0, ctx->stream, (const float4*)dy, (float4*)dx,
                               quads, h, wd * c / 4);
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((upsample_bwd_kernel<T>), dim3(uocr_blocks_for(total, 256, UOCR_MAX_GRID * 4)), dim3(256),
                           0, ctx->stream, (const T*)dy, (T*)dx, n, h, wd, c, sy, sx);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_fixed_width_fwd(uocr_ctx* ctx, int dtype, const void* x, void* y, int n, int h, int wd, int c, int width) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, x && y && n > 0 && h > 0 && wd > 0 && c > 0 && width > 0 && wd >= width);
    const size_t total = (size_t)n * wd * h * width * c;
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((fixed_width_fwd_kernel<T>), dim3(uocr_blocks_for(total, 256, UOCR_MAX_GRID * 4)),
                           dim3(256), 0, ctx->stream, (const T*)x, (T*)y, n, h, wd, c, width);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_fixed_width_bwd(uocr_ctx* ctx, int dtype, const void* dy, void* dx, int n, int h, int wd, int c,
                         int width) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, dy && dx && n > 0 && h > 0 && wd > 0 && c > 0 && width > 0 && wd >= width);
    const size_t total = (size_t)n * h * wd * c;
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((fixed_width_bwd_kernel<T>), dim3(uocr_blocks_for(total, 256, UOCR_MAX_GRID * 4)),
                           dim3(256), 0, ctx->stream, (const T*)dy, (T*)dx, n, h, wd, c, width);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_copy_2d(uocr_ctx* ctx, int dtype, void* dst, size_t dst_ld, const void* src, size_t src_ld, size_t rows,
                 size_t cols) {
    UOCR_CHECK_CTX(ctx);
    if (!rows || !cols) return UOCR_OK;
    UOCR_REQUIRE(ctx, dst && src && dst_ld >= cols && src_ld >= cols);
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((copy2d_kernel<T>), dim3(uocr_blocks_for(rows * cols, 256, UOCR_MAX_GRID * 4)), dim3(256),
                           0, ctx->stream, (T*)dst, dst_ld, (const T*)src, src_ld, rows, cols);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

}  // extern "C"
