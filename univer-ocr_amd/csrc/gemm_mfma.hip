// float32 MFMA GEMM core for gfx950 and the implicit-GEMM forms of Convolutional2D / FullyConnected
// built on it.  Roofline: MFMA (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD, 157 TF peak).
//
//   C[M x N] = A[M x D] . B[D x N]   (D = "depth", the summed index)
//
// Block tile BM x 64 (BM = 128: 4 waves of 32x64; BM = 64: 2x2 waves of 32x32), depth tile 32,
// 256 threads.  A and B tiles are staged global -> registers -> LDS (double buffered, one barrier
// per depth tile): the loads of tile t+1 are issued before the MFMAs of tile t and written to LDS
// after them.  A loaded along the depth is m-major in LDS with a row stride of 33 floats
// (conflict-free ds_read_b32 for the 32x32x2 A fragment: lane l reads A[l&31][k + (l>>5)]); A
// loaded along M (the dw forms) is depth-major with BM-float rows, stored as the float4s it was
// loaded as; B is depth-major with 64-float rows (lane l reads B[k + (l>>5)][l&31]) when it is a row-major [D][N]
// matrix in memory, n-major with stride 33 when it lies along the depth (BDepthContig: the weights of the dx forms);
// A is produced by a loader functor, which is where the im2col / transposed-conv / bias-column
// logic lives -- nothing is materialised in HBM.
//
// Split-D: when M x N has too few tiles to fill 256 CUs (dw of the 64-channel convs: 15 tiles),
// gridDim.z blocks each sum a slice of D into a float32 slab in the workspace and a second kernel
// adds the slabs in a fixed order (deterministic) and applies the epilogue.
// hipcc-flags: -mllvm -amdgpu-mfma-vgpr-form=1
// (accumulators stay in VGPRs: with the AGPR form the compiler copies all 32 of them in and out of the
// AGPRs around every depth tile -- 64 v_accvgpr moves against 32 MFMAs)
#include <array>
#include <type_traits>

#include "conv_dims.h"
#include "gemm.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BN = 64;
constexpr int BD = 32;          // depth tile
constexpr int LDA = BD + 1;     // LDS row stride of the A tile (floats)

struct alignas(16) F4 {
    float v[4];
};

__device__ __forceinline__ F4 f4_zero() { return F4{{0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ F4 f4_fill(float x) { return F4{{x, x, x, x}}; }

// The vector loaders never branch around a global load: an element group that lies outside the operand is
// loaded from the operand's base address instead and tagged, and the tag is resolved when the registers are
// written to LDS, AFTER the MFMAs of the current tile.  (With `cond ? load : constant` the compiler merges
// the two values in the destination registers right after the load -- an s_waitcnt vmcnt(0) in front of the
// MFMA loop, which serialises every depth tile's global latency with its MFMAs.)
enum : int { LD_KEEP = 0, LD_ZERO = 1, LD_E0 = 2, LD_PAD = 3 };   // E0 = (1, 0, 0, 0): the ones row / column

__device__ __forceinline__ F4 ld_resolve(F4 v, int code, float pad) {
    const float first = code == LD_PAD ? pad : (code == LD_E0 ? 1.f : 0.f);
    const float rest = code == LD_PAD ? pad : 0.f;
    F4 r;
    r.v[0] = code == LD_KEEP ? v.v[0] : first;
#pragma unroll
    for (int q = 1; q < 4; ++q) r.v[q] = code == LD_KEEP ? v.v[q] : rest;
    return r;
}

// n / d for 0 <= n < 2^31 as one mulhi + shift (d fixed per launch): a runtime 32-bit division is ~40 VALU
// instructions, and the dw loader needs two per 16-byte load.
struct FastDiv {
    unsigned mul, shift;
    int d;
    FastDiv() = default;
    explicit FastDiv(int divisor) : mul(0), shift(0), d(divisor) {
        if (divisor > 1) {
            int lg = 0;
            while ((1ll << lg) < divisor) ++lg;                    // ceil(log2 d)
            const int p = 31 + lg;
            mul = (unsigned)(((1ull << p) + (unsigned)divisor - 1) / (unsigned)divisor);
            shift = (unsigned)(p - 32);
        }
    }
    __device__ __forceinline__ int div(int n) const { return d == 1 ? n : (int)(__umulhi((unsigned)n, mul) >> shift); }
};

// ---------------------------------------------------------------------------------------------
// epilogue
// ---------------------------------------------------------------------------------------------
struct Epilogue {
    float* c;            // C[i*ldc + j]
    long ldc;
    const float* bias;   // per column, may be null
    int act;             // UOCR_ACT_*
    float alpha;
    int accumulate;      // C += value
    // conv dw/db: rows >= split_row go to c2[j] instead (the bias row of [x~,1]^T.dy)
    int split_row;       // -1 = disabled
    float* c2;
    // conv dx: value *= act'(mask_y[i*ldc + j]) (backward of the activation that produced this conv's input)
    const float* mask_y;
    int mask_act;
    float mask_alpha;
};

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
        case UOCR_ACT_RELU: return v * (v >= 0.f ? 1.f : 0.f);
        case UOCR_ACT_LEAKY: return v * ((v >= 0.f ? 1.f : 0.f) + alpha * (v < 0.f ? 1.f : 0.f));
        case UOCR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

__device__ __forceinline__ void epilogue_store(const Epilogue& e, int i, int j, float v) {
    if (e.bias) v += e.bias[j];
    v = act_apply(v, e.act, e.alpha);
    if (e.mask_act != UOCR_ACT_NONE) v *= act_grad_from_output<float>(e.mask_y[(long)i * e.ldc + j], e.mask_act, e.mask_alpha);
    float* dst = (e.split_row >= 0 && i >= e.split_row) ? e.c2 + j : e.c + (long)i * e.ldc + j;
    *dst = e.accumulate ? *dst + v : v;
}

// ---------------------------------------------------------------------------------------------
// B operand: row-major [D][N]
// ---------------------------------------------------------------------------------------------
struct BRowMajor {
    static constexpr bool DEPTH_CONTIG = false;
    const float* b;
    long ld;
    int rows, n;       // valid extent
    int vec_ok;        // rows 16-byte aligned (ld % 4 == 0, base aligned) and n % 4 == 0: no partial groups
    __device__ __forceinline__ F4 load(int d, int j, int& code) const {
        const bool in = d < rows && j < n;
        code = in ? LD_KEEP : LD_ZERO;
        return *reinterpret_cast<const F4*>(in ? b + (long)d * ld + j : b);
    }
    __device__ __forceinline__ F4 load_slow(int d, int j) const {
        if (d >= rows || j >= n) return f4_zero();
        const float* p = b + (long)d * ld + j;
        if (vec_ok && j + 3 < n) return *reinterpret_cast<const F4*>(p);
        F4 r = f4_zero();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j + q < n) r.v[q] = p[q];
        return r;
    }
};

// B read along the depth -- the weights of the two dx forms, which are stored with the summed index
// contiguous: dense dx B(p, j) = w[j * n_out + p] (one segment), conv dx B((kk, oc), ic) =
// w[(kk * cin + ic) * cout + oc] (one segment of cout per tap).  Staged like a depth-contiguous A: 16-B
// loads along the depth, n-major LDS tile with stride 33.  (These operands used to be transposed into the
// workspace by a separate kernel in front of every dx GEMM: five launches per step of the Char net.)
struct BDepthContig {
    static constexpr bool DEPTH_CONTIG = true;
    const float* b;
    long ld;           // distance between columns j
    int n, depth;      // valid extent
    int rows;          // conv: columns per tap (cin); dense: 0
    FastDiv by_seg;    // depth per segment (conv: cout, dense: depth)
    int vec_ok;        // 16-byte aligned groups: base aligned, ld % 4 == 0, segment % 4 == 0
    __device__ __forceinline__ const float* at(int d0, int j) const {
        const int kk = by_seg.div(d0);
        return b + ((long)kk * rows + j) * ld + (d0 - kk * by_seg.d);
    }
    __device__ __forceinline__ F4 load(int j, int d0, int& code) const {
        const bool in = j < n && d0 < depth;
        code = in ? LD_KEEP : LD_ZERO;
        return *reinterpret_cast<const F4*>(in ? at(d0, j) : b);
    }
    __device__ __forceinline__ F4 load_slow(int j, int d0) const {
        F4 r = f4_zero();
        if (j >= n) return r;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (d0 + q < depth) r.v[q] = *at(d0 + q, j);
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// A loaders.  DEPTH_CONTIG loaders give A(i, d0..d0+3); the others give A(i0..i0+3, d).
// ---------------------------------------------------------------------------------------------
// dense forward / dense dx: A(i,d) = a[i*ld + d]; optional virtual last column of ones ([x,1])
struct ARowMajor {
    static constexpr bool DEPTH_CONTIG = true;
    const float* a;
    long ld;
    int m, stored;     // stored = number of stored columns; column `stored` is the ones column
    int ones_col;
    int vec_ok;        // rows 16-byte aligned and stored % 4 == 0
    static constexpr bool HAS_SLOW = true;
    struct Row {
        const float* p;
        bool ok;
    };
    __device__ __forceinline__ float fill() const { return 0.f; }
    __device__ __forceinline__ Row prep(int i) const { return Row{a + (long)i * ld, i < m}; }
    __device__ __forceinline__ F4 load(const Row& r, int tile, int kq, int& code) const {
        const int d0 = tile * BD + kq * 4;
        const bool in = r.ok && d0 < stored;
        code = in ? LD_KEEP : ((r.ok && ones_col && d0 == stored) ? LD_E0 : LD_ZERO);
        return *reinterpret_cast<const F4*>(in ? r.p + d0 : a);
    }
    __device__ __forceinline__ F4 load_slow(const Row& r, int tile, int kq) const {
        const int d0 = tile * BD + kq * 4;
        if (!r.ok) return f4_zero();
        if (vec_ok && d0 + 3 < stored) return *reinterpret_cast<const F4*>(r.p + d0);
        F4 out = f4_zero();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int d = d0 + q;
            if (d < stored) out.v[q] = r.p[d];
            else if (ones_col && d == stored) out.v[q] = 1.f;
        }
        return out;
    }
};

// dense dw: A(i,d) = a[d*ld + i] (= x^T); optional virtual last ROW of ones ([x,1]^T)
struct AColMajor {
    static constexpr bool DEPTH_CONTIG = false;
    const float* a;
    long ld;
    int stored_rows;   // rows i < stored_rows are stored; row `stored_rows` is the ones row
    int ones_row;
    int depth;
    int vec_ok;        // rows 16-byte aligned and stored_rows % 4 == 0
    static constexpr bool HAS_SLOW = true;
    struct Row {
        int i0;
    };
    __device__ __forceinline__ float fill() const { return 0.f; }
    __device__ __forceinline__ Row prep(int i0) const { return Row{i0}; }
    __device__ __forceinline__ F4 load(const Row& r, int d, int& code) const {
        const bool in = d < depth && r.i0 < stored_rows;
        code = in ? LD_KEEP : ((d < depth && ones_row && r.i0 == stored_rows) ? LD_E0 : LD_ZERO);
        return *reinterpret_cast<const F4*>(in ? a + (long)d * ld + r.i0 : a);
    }
    __device__ __forceinline__ F4 load_slow(const Row& r, int d) const {
        if (d >= depth) return f4_zero();
        const float* p = a + (long)d * ld + r.i0;
        if (vec_ok && r.i0 + 3 < stored_rows) return *reinterpret_cast<const F4*>(p);
        F4 out = f4_zero();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = r.i0 + q;
            if (i < stored_rows) out.v[q] = p[q];
            else if (ones_row && i == stored_rows) out.v[q] = 1.f;
        }
        return out;
    }
};

// conv forward: rows = output pixels, depth = (ky,kx,ic); requires cin % 32 == 0
struct AConvFwd {
    static constexpr bool DEPTH_CONTIG = true;
    const float* x;
    ConvDims d;
    float pad;
    int m;             // n*oh*ow
    // (every integer division of the loaders is a multiply-high by a constant prepared on the host: the hardware has no
    // integer divide, the emulation is ~30 instructions, and on gfx950 a loader's vector instructions are paid in MFMA time)
    int cpt;           // depth tiles per tap = cin / BD
    FastDiv by_ow, by_oh, by_cpt, by_kw;
    AConvFwd(const float* x_, const ConvDims& d_, float pad_, int m_)
        : x(x_), d(d_), pad(pad_), m(m_), cpt(d_.cin / BD), by_ow(d_.ow), by_oh(d_.oh), by_cpt(d_.cin / BD), by_kw(d_.kw) {}
    struct Row {
        long base;     // offset of x[b, iy0, ix0, 0] (may be negative)
        int iy0, ix0;
        bool ok;
    };
    __device__ __forceinline__ Row prep(int i) const {
        Row r;
        r.ok = i < m;
        const int p = r.ok ? i : 0;
        const int t = by_ow.div(p), ox = p - t * d.ow;
        const int b = by_oh.div(t), oy = t - b * d.oh;
        r.iy0 = oy * d.sh - d.ph;
        r.ix0 = ox * d.sw - d.pw;
        r.base = (((long)b * d.h + r.iy0) * d.w + r.ix0) * d.cin;
        return r;
    }
    static constexpr bool HAS_SLOW = false;
    __device__ __forceinline__ float fill() const { return pad; }
    __device__ __forceinline__ F4 load(const Row& r, int tile, int kq, int& code) const {
        // the tap of a depth tile is the same for the whole block: scalar work
        const int kk = by_cpt.div(tile), ky = by_kw.div(kk), kx = kk - ky * d.kw;
        const int tap_off = (ky * d.w + kx) * d.cin + (tile - kk * cpt) * BD;
        const int iy = r.iy0 + ky, ix = r.ix0 + kx;
        const bool inside = (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
        code = !r.ok ? LD_ZERO : (inside ? LD_KEEP : LD_PAD);
        return *reinterpret_cast<const F4*>(code == LD_KEEP ? x + r.base + (tap_off + kq * 4) : x);
    }
};

// conv dx: rows = input pixels, depth = (ky,kx,oc); requires cout % 32 == 0
struct AConvDgrad {
    static constexpr bool DEPTH_CONTIG = true;
    const float* dy;
    ConvDims d;
    int m;             // n*h*w
    int cpt;           // depth tiles per tap = cout / BD
    FastDiv by_w, by_h, by_cpt, by_kw, by_sh, by_sw;
    AConvDgrad(const float* dy_, const ConvDims& d_, int m_)
        : dy(dy_), d(d_), m(m_), cpt(d_.cout / BD), by_w(d_.w), by_h(d_.h), by_cpt(d_.cout / BD), by_kw(d_.kw), by_sh(d_.sh),
          by_sw(d_.sw) {}
    struct Row {
        long base;     // offset of dy[b, 0, 0, 0]
        int y, x;
        bool ok;
    };
    __device__ __forceinline__ Row prep(int i) const {
        Row r;
        r.ok = i < m;
        const int p = r.ok ? i : 0;
        const int t = by_w.div(p);
        r.x = p - t * d.w;
        const int b = by_h.div(t);
        r.y = t - b * d.h;
        r.base = (long)b * d.oh * d.ow * d.cout;
        return r;
    }
    static constexpr bool HAS_SLOW = false;
    __device__ __forceinline__ float fill() const { return 0.f; }
    __device__ __forceinline__ F4 load(const Row& r, int tile, int kq, int& code) const {
        const int kk = by_cpt.div(tile), ky = by_kw.div(kk), kx = kk - ky * d.kw;      // (scalar: the block's tap)
        const int c0 = (tile - kk * cpt) * BD + kq * 4;
        const int ty = r.y + d.ph - ky, tx = r.x + d.pw - kx;
        const int gy = by_sh.div(ty), gx = by_sw.div(tx);      // (meaningless for negative ty / tx: caught by ty >= 0)
        const bool hit = r.ok && ty >= 0 && tx >= 0 && gy * d.sh == ty && gx * d.sw == tx && gy < d.oh && gx < d.ow;
        code = hit ? LD_KEEP : LD_ZERO;
        return *reinterpret_cast<const F4*>(hit ? dy + r.base + ((gy * d.ow + gx) * d.cout + c0) : dy);
    }
    // A strided transposed conv multiplies structural zeros: input row y only receives tap rows ky with
    // (y + ph - ky) divisible by the stride and inside the output.  When the BM rows of a block lie in ONE
    // image row (Char net: w = 64 = BM) whole depth tiles are zero for the block and are skipped
    // (stride (2,1), kh = 5: 2 or 3 of 5 tap rows remain).
    __device__ __forceinline__ bool tile_is_zero(int m0, int bm, int tile) const {
        if (d.sh == 1) return false;        // only the top / bottom image rows would gain: not worth the test
        const int last = min(m0 + bm, m) - 1;
        const int r0 = by_w.div(m0), r1 = by_w.div(last);
        if (r0 != r1) return false;
        const int y = r0 - by_h.div(r0) * d.h;
        const int ky = by_kw.div(by_cpt.div(tile));
        const int ty = y + d.ph - ky;
        if (ty < 0) return true;
        const int gy = by_sh.div(ty);
        return gy * d.sh != ty || gy >= d.oh;
    }
};

// conv dw/db: rows = (ky,kx,ic) [+ one bias row], depth = output pixels; requires cin % 4 == 0
struct AConvWgrad {
    static constexpr bool DEPTH_CONTIG = false;
    const float* x;
    ConvDims d;
    float pad;
    int K;             // kh*kw*cin
    int use_bias;
    int depth;         // n*oh*ow
    FastDiv by_ow, by_oh;
    struct Row {
        int ky, kx, ic0;
        int kind;      // 0 = weights rows, 1 = bias row group, 2 = beyond
    };
    __device__ __forceinline__ Row prep(int i0) const {
        Row r;
        if (i0 < K) {
            const int kk = i0 / d.cin;
            r.ic0 = i0 - kk * d.cin;
            r.ky = kk / d.kw;
            r.kx = kk - r.ky * d.kw;
            r.kind = 0;
        } else {
            r.ky = r.kx = r.ic0 = 0;
            r.kind = (i0 == K && use_bias) ? 1 : 2;
        }
        return r;
    }
    static constexpr bool HAS_SLOW = false;
    __device__ __forceinline__ float fill() const { return pad; }
    __device__ __forceinline__ F4 load(const Row& r, int p, int& code) const {
        const bool live = p < depth;
        const int t = by_ow.div(live ? p : 0), ox = (live ? p : 0) - t * d.ow;
        const int b = by_oh.div(t), oy = t - b * d.oh;
        const int iy = oy * d.sh - d.ph + r.ky, ix = ox * d.sw - d.pw + r.kx;
        const bool inside = iy >= 0 && iy < d.h && ix >= 0 && ix < d.w;
        code = (!live || r.kind == 2) ? LD_ZERO : (r.kind == 1 ? LD_E0 : (inside ? LD_KEEP : LD_PAD));
        return *reinterpret_cast<const F4*>(code == LD_KEEP ? x + (((long)b * d.h + iy) * d.w + ix) * d.cin + r.ic0 : x);
    }
    // The same load from a CURSOR: the output pixel (b, oy, ox) of a depth index and the element offset of its
    // window origin are carried from depth tile to depth tile (+BD pixels: at most one row wrap when ow >= BD)
    // instead of being rebuilt with two divisions and a 64-bit multiply chain per 16-byte load -- on gfx950 f32
    // MFMAs and vector instructions do not overlap, so every VALU instruction of the loader
    // is paid in MFMA time.  Usable when ow >= BD and the tensor has < 2^31 elements (`cursor_ok`).
    struct Cursor {
        int ox, oy, b;
        int base;          // ((b*h + oy*sh) * w + ox*sw) * cin
    };
    int cursor_ok;
    __device__ __forceinline__ Cursor cursor_at(int p) const {
        Cursor c;
        const int q = min(p, depth);       // (beyond the depth: b == n, which `load_at` reads as "not live")
        const int t = by_ow.div(q);
        c.ox = q - t * d.ow;
        c.b = by_oh.div(t);
        c.oy = t - c.b * d.oh;
        c.base = ((c.b * d.h + c.oy * d.sh) * d.w + c.ox * d.sw) * d.cin;
        return c;
    }
    __device__ __forceinline__ void advance(Cursor& c) const {          // += BD pixels
        c.ox += BD;
        c.base += BD * d.sw * d.cin;
        if (c.ox >= d.ow) {
            c.ox -= d.ow;
            c.oy += 1;
            c.base += (d.sh * d.w - d.ow * d.sw) * d.cin;
            if (c.oy == d.oh) {
                c.oy = 0;
                c.b += 1;
                c.base += (d.h - d.oh * d.sh) * d.w * d.cin;
            }
        }
    }
    __device__ __forceinline__ F4 load_at(const Row& r, const Cursor& c, int& code) const {
        const int iy = c.oy * d.sh - d.ph + r.ky, ix = c.ox * d.sw - d.pw + r.kx;
        const bool inside = (unsigned)iy < (unsigned)d.h && (unsigned)ix < (unsigned)d.w;
        code = (c.b >= d.n || r.kind == 2) ? LD_ZERO : (r.kind == 1 ? LD_E0 : (inside ? LD_KEEP : LD_PAD));
        const int off = c.base + ((r.ky - d.ph) * d.w + (r.kx - d.pw)) * d.cin + r.ic0;
        return *reinterpret_cast<const F4*>(x + (code == LD_KEEP ? off : 0));
    }
};

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
// LDS of one block: the A tile (double buffered), then the B tile
template <int BM>
constexpr int gemm_a_floats() { return 2 * BM * LDA; }
template <typename BLoader>
constexpr int gemm_b_floats() { return 2 * (BLoader::DEPTH_CONTIG ? BN * LDA : BD * BN); }

// Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  Blocks that share
// operand rows (the M tiles of one depth slab in dw; neighbouring pixel tiles, which share halo rows, in a
// conv forward) should share an L2: XCD c takes the c-th contiguous eighth of the logical block ids.
__device__ __forceinline__ void gemm_block_coords(int lin, int gx, int gy, int gz, int xcd_remap, int& bx, int& by, int& bz) {
    if (xcd_remap) {
        const int total = gx * gy * gz;
        const int c = lin & 7, q = total >> 3, r = total & 7;
        lin = c * q + min(c, r) + (lin >> 3);
    }
    bx = lin % gx;
    const int rest = lin / gx;
    by = rest % gy;
    bz = rest / gy;
}

// one block's share of C = A . B: output tile (bx, by), depth slice bz; As / Bs: this block's LDS
template <int BM, typename ALoader, typename BLoader>
__device__ __forceinline__ void gemm_tile(const ALoader& A, const BLoader& B, const Epilogue& ep, int M, int N, int ntiles,
                                          int tiles_per_split, float* slabs, int bx, int by, int bz, float* As,
                                          float* Bs) {
    constexpr int WM = BM / 32;            // waves along M: 4 or 2
    constexpr int WN = 4 / WM;             // waves along N: 1 or 2
    constexpr int NB = (BN / WN) / 32;     // 32x32 blocks per wave along N: 2 or 1
    constexpr int A_PASSES = BM / 32;      // depth-contiguous staging: 32 rows per pass
    constexpr int AM_GROUPS = BM / 4;      // m-contiguous staging: float4 groups along M
    constexpr int AM_ROWS = 256 / AM_GROUPS;   // depth rows covered per pass: 8 or 16
    constexpr int AM_PASSES = BD / AM_ROWS;    // 4 or 2
    constexpr int A_REGS = ALoader::DEPTH_CONTIG ? A_PASSES : AM_PASSES;
    constexpr int ASZ = BM * LDA;                                             // floats per buffer
    constexpr int BSZ = BLoader::DEPTH_CONTIG ? BN * LDA : BD * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = bx * BM, n0 = by * BN;
    const int t_begin = bz * tiles_per_split;
    const int t_end = min(ntiles, t_begin + tiles_per_split);

    // ---- staging assignment ----
    typename ALoader::Row rows[ALoader::DEPTH_CONTIG ? A_PASSES : 1];
    int a_kq = 0, a_r = 0, a_iq = 0, a_pr = 0;
    if constexpr (ALoader::DEPTH_CONTIG) {
        a_kq = tid & 7;
        a_r = tid >> 3;
#pragma unroll
        for (int s = 0; s < A_PASSES; ++s) rows[s] = A.prep(m0 + a_r + 32 * s);
    } else {
        a_iq = tid % AM_GROUPS;
        a_pr = tid / AM_GROUPS;
        rows[0] = A.prep(m0 + a_iq * 4);
    }
    const int b_nq = tid & 15, b_kr = tid >> 4;       // row-major B: float4 along n at depths b_kr, b_kr + 16
    const int b_kq = tid & 7, b_r = tid >> 3;         // depth-contiguous B: float4 along the depth, columns b_r, b_r + 32

    F4 areg[A_REGS], breg[2];
    int acode[A_REGS], bcode[2];
    bool a_slow = false;
    if constexpr (ALoader::HAS_SLOW) a_slow = !A.vec_ok;
    constexpr bool HAS_CURSOR = requires { typename ALoader::Cursor; };
    struct NoCursor {};
    auto cursors = [&] {
        if constexpr (HAS_CURSOR) return std::array<typename ALoader::Cursor, AM_PASSES>{};
        else return NoCursor{};
    }();
    bool use_cursor = false;
    if constexpr (HAS_CURSOR) {
        use_cursor = A.cursor_ok != 0;
        if (use_cursor) {
#pragma unroll
            for (int sidx = 0; sidx < AM_PASSES; ++sidx) cursors[sidx] = A.cursor_at(t_begin * BD + a_pr + AM_ROWS * sidx);
        }
    }
    auto load_tile = [&](int t) {
        if (a_slow) {             // unaligned rows or ragged groups: element loads (block-uniform choice)
            if constexpr (ALoader::HAS_SLOW) {
#pragma unroll
                for (int s = 0; s < A_REGS; ++s) {
                    if constexpr (ALoader::DEPTH_CONTIG) areg[s] = A.load_slow(rows[s], t, a_kq);
                    else areg[s] = A.load_slow(rows[0], t * BD + a_pr + AM_ROWS * s);
                    acode[s] = LD_KEEP;
                }
            }
        } else if constexpr (ALoader::DEPTH_CONTIG) {
#pragma unroll
            for (int s = 0; s < A_PASSES; ++s) areg[s] = A.load(rows[s], t, a_kq, acode[s]);
        } else {
            bool done = false;
            if constexpr (HAS_CURSOR) {
                if (use_cursor) {           // tiles are visited in order from t_begin (no tile skipping for this loader)
#pragma unroll
                    for (int s = 0; s < AM_PASSES; ++s) {
                        areg[s] = A.load_at(rows[0], cursors[s], acode[s]);
                        A.advance(cursors[s]);
                    }
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int s = 0; s < AM_PASSES; ++s) areg[s] = A.load(rows[0], t * BD + a_pr + AM_ROWS * s, acode[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (BLoader::DEPTH_CONTIG) {
                if (B.vec_ok) breg[s] = B.load(n0 + b_r + 32 * s, t * BD + b_kq * 4, bcode[s]);
                else {
                    breg[s] = B.load_slow(n0 + b_r + 32 * s, t * BD + b_kq * 4);
                    bcode[s] = LD_KEEP;
                }
            } else if (B.vec_ok) breg[s] = B.load(t * BD + b_kr + 16 * s, n0 + b_nq * 4, bcode[s]);
            else {
                breg[s] = B.load_slow(t * BD + b_kr + 16 * s, n0 + b_nq * 4);
                bcode[s] = LD_KEEP;
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * ASZ;
        // (interior tiles: every group of the wave was loaded as it is -- one vote instead of 8 selects per group)
        int codes = bcode[0] | bcode[1];
#pragma unroll
        for (int s = 0; s < A_REGS; ++s) codes |= acode[s];
        if (__builtin_amdgcn_ballot_w64(codes != LD_KEEP) != 0) {
#pragma unroll
            for (int s = 0; s < A_REGS; ++s) areg[s] = ld_resolve(areg[s], acode[s], A.fill());
#pragma unroll
            for (int s = 0; s < 2; ++s) breg[s] = ld_resolve(breg[s], bcode[s], 0.f);
        }
        if constexpr (ALoader::DEPTH_CONTIG) {
#pragma unroll
            for (int s = 0; s < A_PASSES; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) as[(a_r + 32 * s) * LDA + a_kq * 4 + q] = areg[s].v[q];
        } else {
#pragma unroll
            for (int s = 0; s < AM_PASSES; ++s)      // depth-major [BD][BM]: the float4 goes in as it was loaded
                *reinterpret_cast<F4*>(&as[(a_pr + AM_ROWS * s) * BM + a_iq * 4]) = areg[s];
        }
        if constexpr (BLoader::DEPTH_CONTIG) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) Bs[buf * BSZ + (b_r + 32 * s) * LDA + b_kq * 4 + q] = breg[s].v[q];
        } else {
            *reinterpret_cast<F4*>(&Bs[buf * BSZ + b_kr * BN + b_nq * 4]) = breg[0];
            *reinterpret_cast<F4*>(&Bs[buf * BSZ + (b_kr + 16) * BN + b_nq * 4]) = breg[1];
        }
    };

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    // which 32-row band of the tile a wave owns rotates with the slab index: in the last M tile of a ragged GEMM
    // (dw of a 64-channel 3x3 conv: 577 rows = 4.5 tiles) some bands lie beyond M and do no MFMAs; without the
    // rotation those idle waves always sit on the same SIMDs of every CU (wave w runs on SIMD w)
    const int wm = (wave % WM + bz) % WM, wn = wave / WM;
    const int fi = wm * 32 + (lane & 31);        // A fragment row of this lane
    const int fk = lane >> 5;                    // depth offset inside a 2-deep MFMA step
    const int fj = wn * (BN / WN) + (lane & 31); // B fragment column (first 32x32 block)
    const bool band_live = m0 + wm * 32 < M;     // a wave whose 32 rows lie beyond M only helps with the staging

    // depth tiles that are zero for the whole block (block-uniform) are stepped over
    auto next_tile = [&](int t) {
        if constexpr (requires { A.tile_is_zero(0, 0, 0); }) {
            while (t < t_end && A.tile_is_zero(m0, BM, t)) ++t;
        }
        return t;
    };
    int t = next_tile(t_begin);
    if (t < t_end) {
        load_tile(t);
        store_tile(0);
    }
    __syncthreads();
    for (int buf = 0; t < t_end; buf ^= 1) {
        const int tn = next_tile(t + 1);
        const bool more = tn < t_end;
        if (more) load_tile(tn);
        if (band_live) {
            // fragments of depth step k+2 are read from LDS while the MFMAs of step k run
            // one base address per operand and tile (buffer + lane part), made opaque so that the compiler keeps it in
            // a VGPR and addresses the 16 depth steps with immediate offsets (it otherwise re-adds the buffer offset
            // to sixteen hoisted per-step registers: a v_add3 in front of every ds_read, i.e. VALU time taken from
            // the MFMAs, section 5a of DESIGN.md)
            // (indices into the __shared__ arrays, not pointers: a pointer through the asm loses its address space and
            // the reads become flat loads)
            int ia = buf * ASZ + (ALoader::DEPTH_CONTIG ? fi * LDA + fk : fk * BM + fi);
            int ib = buf * BSZ + (BLoader::DEPTH_CONTIG ? fj * LDA + fk : fk * BN + fj);
            asm volatile("" : "+v"(ia), "+v"(ib));
            auto frag_a = [&](int k) { return As[ia + (ALoader::DEPTH_CONTIG ? k : k * BM)]; };
            auto frag_b = [&](int k, int nb) {
                return Bs[ib + (BLoader::DEPTH_CONTIG ? nb * 32 * LDA + k : k * BN + nb * 32)];
            };
            float a = frag_a(0), b[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = frag_b(0, nb);
#pragma unroll
            for (int k = 0; k < BD; k += 2) {
                float an = 0.f, bn[NB];
                if (k + 2 < BD) {
                    an = frag_a(k + 2);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) bn[nb] = frag_b(k + 2, nb);
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[nb], acc[nb], 0, 0, 0);
                if (k + 2 < BD) {
                    a = an;
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) b[nb] = bn[nb];
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        t = tn;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int j = n0 + fj + nb * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < M && j < N) {
                if (slabs) slabs[((long)bz * M + i) * N + j] = acc[nb][r];
                else epilogue_store(ep, i, j, acc[nb][r]);
            }
        }
    }
}

template <int BM, typename ALoader, typename BLoader>
__global__ __launch_bounds__(256) void mfma_gemm_kernel(ALoader A, BLoader B, Epilogue ep, int M, int N,
                                                        int ntiles, int tiles_per_split, float* slabs,
                                                        int xcd_remap) {
    __shared__ __attribute__((aligned(16))) float As[gemm_a_floats<BM>()];
    __shared__ __attribute__((aligned(16))) float Bs[gemm_b_floats<BLoader>()];
    int bx, by, bz;
    gemm_block_coords(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y, gridDim.z,
                      xcd_remap, bx, by, bz);
    gemm_tile<BM, ALoader, BLoader>(A, B, ep, M, N, ntiles, tiles_per_split, slabs, bx, by, bz, As, Bs);
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nsplit, int M, int N,
                                                          Epilogue ep) {
    const long total = (long)M * N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        // eight slabs' loads in flight at a time, added in slab order (the sum is the same fmaf-free chain
        // z = 0, 1, 2, ... whatever the batching)
        float v = 0.f;
        int z = 0;
        for (; z + 8 <= nsplit; z += 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = slabs[(long)(z + k) * total + idx];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
        for (; z < nsplit; ++z) v += slabs[(long)z * total + idx];
        epilogue_store(ep, (int)(idx / N), (int)(idx % N), v);
    }
}

// out[c*rows + r] = in[r*cols + c]  (weights are small: <= 2 MB)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int rows, int cols) {
    const long total = (long)rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx / rows), r = (int)(idx % rows);
        out[idx] = in[(long)r * cols + c];
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// workspace layout for this file: [0, half) split-D slabs, [half, end) transposed weights
inline float* ws_slabs(uocr_ctx* ctx) { return (float*)ctx->workspace; }
inline float* ws_aux(uocr_ctx* ctx) { return (float*)((char*)ctx->workspace + ctx->workspace_bytes / 2); }
inline size_t ws_half(uocr_ctx* ctx) { return ctx->workspace_bytes / 2; }

template <typename ALoader, typename BLoader>
int launch_mfma(uocr_ctx* ctx, const ALoader& A, const BLoader& B, const Epilogue& ep, int M, int N, int depth,
                bool allow_split) {
    const int ntiles = (depth + BD - 1) / BD;
    // 128-row tiles (two MFMAs per A fragment read) when they still fill the chip: on their own, or -- very deep
    // GEMMs, dw of a wide conv -- together with the depth split
    const long tiles128 = (long)((M + 127) / 128) * ((N + BN - 1) / BN);
    const long deep = (allow_split && ctx->opt_split > 0) ? min(256, ntiles / 32) : 1;
    int bm = (tiles128 >= 512 || tiles128 * deep >= 512) ? 128 : 64;
    if (ctx->opt_bm == 64 || ctx->opt_bm == 128) bm = ctx->opt_bm;
    const int gm = (M + bm - 1) / bm, gn = (N + BN - 1) / BN;
    int nsplit = 1;
    // Small GEMMs are latency-bound per block (a depth tile is ~1k MFMA cycles against a 1-2 us global
    // load round trip), so aim for ~4 resident blocks per CU: split the depth until there are ~1024 blocks
    // (a fused reduction by the last-arriving block was tried: the agent-scope release/acquire it needs
    // writes back and invalidates the XCD's L2 per block -- the Char net went from 0.57 to 2.55 ms/step)
    // "split_blocks" = 1024 (the default) means: as many blocks as are RESIDENT at once -- 3 per CU with 128-row
    // tiles (50 KB of LDS each), 4 with 64-row tiles (33 KB).  A grid of 1025 blocks of the 128-row kernel ran as
    // one full round of 768 plus a second round a third full: dw of the wide conv 59 -> see DESIGN.md section 5.
    const int target = ctx->opt_split == 1024 ? ctx->cu_count * (bm == 128 ? 3 : 4) : ctx->opt_split;
    if (allow_split && target > 0 && gm * gn < target && ntiles >= 4) {
        nsplit = ctx->opt_split == 1024 ? max(1, target / (gm * gn)) : (target + gm * gn - 1) / (gm * gn);
        // at most 32 slabs of >= 2 depth tiles, or up to 256 when every slab still has >= 32 tiles (very deep
        // GEMMs with few output tiles: dw of a wide conv, 5 tiles x 131072 depth tiles -- ~4 blocks per CU
        // keep the CUs evenly loaded where 2.5 would not)
        const int cap = max(min(32, ntiles / 2), min(256, ntiles / 32));
        if (nsplit > cap) nsplit = cap;
        const size_t per = (size_t)M * N * sizeof(float);
        if ((size_t)nsplit * per > ws_half(ctx)) nsplit = (int)(ws_half(ctx) / per);
        if (nsplit < 1) nsplit = 1;
        if (nsplit < ctx->opt_split_min) nsplit = 1;
    }
    const int tps = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + tps - 1) / tps;
    float* slabs = nsplit > 1 ? ws_slabs(ctx) : nullptr;
    const dim3 grid(gm, gn, nsplit), block(256);
    if (bm == 128)
        hipLaunchKernelGGL((mfma_gemm_kernel<128, ALoader, BLoader>), grid, block, 0, ctx->stream, A, B, ep, M, N, ntiles, tps,
                           slabs, ctx->opt_xcd);
    else
        hipLaunchKernelGGL((mfma_gemm_kernel<64, ALoader, BLoader>), grid, block, 0, ctx->stream, A, B, ep, M, N, ntiles, tps,
                           slabs, ctx->opt_xcd);
    UOCR_LAUNCH_CHECK(ctx);
    if (nsplit > 1) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(uocr_blocks_for((size_t)M * N, 256, 1024)), dim3(256), 0,
                           ctx->stream, (const float*)slabs, nsplit, M, N, ep);
        UOCR_LAUNCH_CHECK(ctx);
    }
    return UOCR_OK;
}

// ---------------------------------------------------------------------------------------------
// Deferred weight-gradient GEMMs: ONE launch for all of them (uocr_wgrad_defer_begin / _flush)
//
// The weight gradients of a backward pass do not depend on each other and nothing reads them before the pass ends; the
// Char net's five (conv_2, conv_3, the windows + dense_1 layer as a conv; dense_2, dense_3) are GEMMs of 15-130 output
// tiles each, run one after the other: five launches of a few hundred latency-bound blocks + five reduce launches, 125
// of the net's 440 us.  Between begin and flush the eligible weight-gradient GEMMs are only RECORDED; flush splits
// their depths so that all of them together fill the chip and runs them as one grid (block -> problem by a prefix
// table, two loader types) followed by one reduce launch.  Same tiles, same slab order per problem: the sums are those
// of the separate launches with the same split.
// ---------------------------------------------------------------------------------------------
constexpr int GROUP_MAX = 8, GROUP_MAX_TYPE = 4;

struct GroupShape {
    int M, N, ntiles, tps, gx, gy, gz, first_block;
    float* slabs;
    Epilogue ep;
};
struct GroupArgs {
    int count, xcd_remap;
    int type[GROUP_MAX];            // 0: AConvWgrad, 1: AColMajor
    int index[GROUP_MAX];           // into conv_a / col_a
    GroupShape shape[GROUP_MAX];
    BRowMajor b[GROUP_MAX];
    AConvWgrad conv_a[GROUP_MAX_TYPE];
    AColMajor col_a[GROUP_MAX_TYPE];
};

__global__ __launch_bounds__(256) void mfma_gemm_group_kernel(GroupArgs g) {
    __shared__ __attribute__((aligned(16))) float As[gemm_a_floats<64>()];
    __shared__ __attribute__((aligned(16))) float Bs[gemm_b_floats<BRowMajor>()];
    int p = 0;
    while (p + 1 < g.count && (int)blockIdx.x >= g.shape[p + 1].first_block) ++p;     // (block-uniform)
    const GroupShape& sh = g.shape[p];
    int bx, by, bz;
    gemm_block_coords((int)blockIdx.x - sh.first_block, sh.gx, sh.gy, sh.gz, g.xcd_remap, bx, by, bz);
    if (g.type[p] == 0)
        gemm_tile<64, AConvWgrad, BRowMajor>(g.conv_a[g.index[p]], g.b[p], sh.ep, sh.M, sh.N, sh.ntiles, sh.tps, sh.slabs, bx,
                                             by, bz, As, Bs);
    else
        gemm_tile<64, AColMajor, BRowMajor>(g.col_a[g.index[p]], g.b[p], sh.ep, sh.M, sh.N, sh.ntiles, sh.tps, sh.slabs, bx,
                                            by, bz, As, Bs);
}

struct GroupReduceArgs {
    int count;
    long first[GROUP_MAX + 1];      // element offsets of the problems that were split
    const float* slabs[GROUP_MAX];
    int nsplit[GROUP_MAX], N[GROUP_MAX];
    Epilogue ep[GROUP_MAX];
};

__global__ __launch_bounds__(256) void slab_reduce_group_kernel(GroupReduceArgs a) {
    const long total = a.first[a.count];
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        int p = 0;
        while (p + 1 < a.count && g >= a.first[p + 1]) ++p;
        const long idx = g - a.first[p], size = a.first[p + 1] - a.first[p];
        const float* slabs = a.slabs[p];
        const int nsplit = a.nsplit[p];
        float v = 0.f;                                  // eight slabs' loads in flight, added in slab order (slab_reduce_kernel)
        int z = 0;
        for (; z + 8 <= nsplit; z += 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = slabs[(long)(z + k) * size + idx];
#pragma unroll
            for (int k = 0; k < 8; ++k) v += t[k];
        }
        for (; z < nsplit; ++z) v += slabs[(long)z * size + idx];
        epilogue_store(a.ep[p], (int)(idx / a.N[p]), (int)(idx % a.N[p]), v);
    }
}

// what is recorded between begin and flush
struct GemmDefer {
    bool on = false;
    int count = 0, nconv = 0, ncol = 0;
    GroupArgs args{};
    int depth[GROUP_MAX]{};
};

inline GemmDefer* defer_of(uocr_ctx* ctx) { return (GemmDefer*)ctx->gemm_defer; }

// true: recorded (the caller returns UOCR_OK without launching)
template <typename ALoader>
bool defer_record(uocr_ctx* ctx, const ALoader& A, const BRowMajor& B, const Epilogue& ep, int M, int N, int depth) {
    GemmDefer* d = defer_of(ctx);
    if (!d || !d->on || d->count >= GROUP_MAX) return false;
    constexpr bool conv = std::is_same<ALoader, AConvWgrad>::value;
    if ((conv ? d->nconv : d->ncol) >= GROUP_MAX_TYPE) return false;
    // only the small problems (64-row tiles, few of them) gain from sharing a launch
    const long tiles = (long)((M + 63) / 64) * ((N + BN - 1) / BN);
    if (tiles >= 2L * ctx->cu_count || (size_t)M * N * sizeof(float) * 2 > ws_half(ctx)) return false;
    const int p = d->count++;
    GroupArgs& g = d->args;
    g.type[p] = conv ? 0 : 1;
    if constexpr (conv) {
        g.index[p] = d->nconv;
        g.conv_a[d->nconv++] = A;
    } else {
        g.index[p] = d->ncol;
        g.col_a[d->ncol++] = A;
    }
    g.b[p] = B;
    g.shape[p].M = M;
    g.shape[p].N = N;
    g.shape[p].ep = ep;
    d->depth[p] = depth;
    return true;
}

int defer_flush(uocr_ctx* ctx) {
    GemmDefer* d = defer_of(ctx);
    if (!d || d->count == 0) return UOCR_OK;
    GroupArgs& g = d->args;
    g.count = d->count;
    g.xcd_remap = ctx->opt_xcd;
    // depth splits: the problems share ~4 resident blocks per CU in proportion to their work (tile-iterations)
    double work[GROUP_MAX], total_work = 0.0;
    for (int p = 0; p < g.count; ++p) {
        GroupShape& sh = g.shape[p];
        sh.gx = (sh.M + 63) / 64;
        sh.gy = (sh.N + BN - 1) / BN;
        sh.ntiles = (d->depth[p] + BD - 1) / BD;
        work[p] = (double)sh.gx * sh.gy * sh.ntiles;
        total_work += work[p];
    }
    const double target = ctx->opt_split <= 0 ? 0.0 : ctx->opt_group_blocks > 0 ? (double)ctx->opt_group_blocks : 4.0 * ctx->cu_count;
    size_t slab_floats = 0;
    int nblocks = 0;
    GroupReduceArgs r{};
    for (int p = 0; p < g.count; ++p) {
        GroupShape& sh = g.shape[p];
        const int tiles = sh.gx * sh.gy;
        int nsplit = (int)(target * work[p] / total_work / tiles + 0.5);
        const int cap = std::max(std::min(32, sh.ntiles / 2), std::min(256, sh.ntiles / 32));
        nsplit = std::max(1, std::min(nsplit, cap));
        const size_t per = (size_t)sh.M * sh.N;
        while (nsplit > 1 && (slab_floats + (size_t)nsplit * per) * sizeof(float) > ws_half(ctx)) --nsplit;
        if (nsplit < ctx->opt_split_min) nsplit = 1;
        sh.tps = (sh.ntiles + nsplit - 1) / nsplit;
        nsplit = (sh.ntiles + sh.tps - 1) / sh.tps;
        sh.gz = nsplit;
        sh.first_block = nblocks;
        nblocks += tiles * nsplit;
        sh.slabs = nullptr;
        if (nsplit > 1) {
            sh.slabs = ws_slabs(ctx) + slab_floats;
            slab_floats += (size_t)nsplit * per;
            const int q = r.count++;
            r.first[q + 1] = r.first[q] + (long)per;
            r.slabs[q] = sh.slabs;
            r.nsplit[q] = nsplit;
            r.N[q] = sh.N;
            r.ep[q] = sh.ep;
        }
    }
    hipLaunchKernelGGL(mfma_gemm_group_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, g);
    UOCR_LAUNCH_CHECK(ctx);
    if (r.count > 0) {
        hipLaunchKernelGGL(slab_reduce_group_kernel, dim3(uocr_blocks_for((size_t)r.first[r.count], 256, 2048)), dim3(256),
                           0, ctx->stream, r);
        UOCR_LAUNCH_CHECK(ctx);
    }
    d->count = d->nconv = d->ncol = 0;
    return UOCR_OK;
}

inline Epilogue plain_epilogue(float* c, long ldc, int accumulate) {
    return Epilogue{c, ldc, nullptr, UOCR_ACT_NONE, 0.f, accumulate, -1, nullptr, nullptr, UOCR_ACT_NONE, 0.f};
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// entry points used by the dispatchers (gemm_dispatch.hip, conv_api.hip)
// ---------------------------------------------------------------------------------------------
bool uocr_gemm_mfma_eligible(uocr_ctx* ctx, int dtype, const GemmArgs& g) {
    if (dtype != UOCR_F32 || ctx->opt_mfma == 0) return false;
    const bool a_row = (g.a_cs == 1), a_col = (g.a_rs == 1 && !a_row);
    if (!(a_row || a_col)) return false;
    if (!(g.b_cs == 1 || g.b_rs == 1)) return false;
    if (g.b_cs != 1 && (g.b_cs != g.depth || (size_t)g.n * g.depth * sizeof(float) > ws_half(ctx))) return false;
    if (ctx->opt_mfma == 2) return true;
    return (double)g.m * g.n * g.depth >= 4.0e6 && g.n >= 32 && g.m >= 32;
}

int uocr_gemm_mfma(uocr_ctx* ctx, const GemmArgs& g) {
    Epilogue ep0 = plain_epilogue((float*)g.c, g.ldc, g.accumulate);
    ep0.act = g.act;                                  // fused activations of the dense layers (gemm.h)
    ep0.alpha = (float)g.act_alpha;
    ep0.mask_y = (const float*)g.mask_y;
    ep0.mask_act = g.mask_act;
    ep0.mask_alpha = (float)g.mask_alpha;
    if (g.b_cs != 1 && g.b_rs == 1 && g.a_cs == 1) {
        // dense dx: B(p, j) = w[j * n_out + p] is read along the depth as it lies in memory
        const int stored = g.a_ones_col ? g.depth - 1 : g.depth;
        ARowMajor A{(const float*)g.a, g.a_rs, g.m, stored, g.a_ones_col,
                    (g.a_rs % 4 == 0 && stored % 4 == 0 && aligned16(g.a)) ? 1 : 0};
        BDepthContig B{(const float*)g.b, g.b_cs, g.n, g.depth, 0, FastDiv(g.depth),
                       (g.b_cs % 4 == 0 && g.depth % 4 == 0 && aligned16(g.b)) ? 1 : 0};
        return launch_mfma(ctx, A, B, ep0, g.m, g.n, g.depth, true);
    }
    // otherwise B must be row-major [depth][n]; a transposed B is transposed into the workspace
    BRowMajor B;
    if (g.b_cs == 1) {
        B = BRowMajor{(const float*)g.b, g.b_rs, g.depth, g.n, (g.b_rs % 4 == 0 && g.n % 4 == 0 && aligned16(g.b)) ? 1 : 0};
    } else {
        float* bt = ws_aux(ctx);     // B(p,j) = b[p + j*b_cs]: stored as [n][b_cs]; take its transpose
        // in[r = j][c = p] with cols = b_cs (only the first `depth` columns are used) -> out[p*n + j]
        hipLaunchKernelGGL(transpose_kernel, dim3(uocr_blocks_for((size_t)g.n * g.depth, 256, 1024)), dim3(256), 0,
                           ctx->stream, (const float*)g.b, bt, g.n, (int)g.b_cs);
        UOCR_LAUNCH_CHECK(ctx);
        // transpose_kernel wrote out[c*rows + r] for c < cols = b_cs; rows = n  => out[p*n + j]
        B = BRowMajor{bt, g.n, g.depth, g.n, (g.n % 4 == 0 && aligned16(bt)) ? 1 : 0};
    }
    const Epilogue ep = ep0;
    if (g.a_cs == 1) {
        const int stored = g.a_ones_col ? g.depth - 1 : g.depth;
        ARowMajor A{(const float*)g.a, g.a_rs, g.m, stored, g.a_ones_col,
                    (g.a_rs % 4 == 0 && stored % 4 == 0 && aligned16(g.a)) ? 1 : 0};
        return launch_mfma(ctx, A, B, ep, g.m, g.n, g.depth, true);
    }
    const int stored_rows = g.a_ones_row ? g.m - 1 : g.m;
    AColMajor A{(const float*)g.a, g.a_cs, stored_rows, g.a_ones_row, g.depth,
                (g.a_cs % 4 == 0 && stored_rows % 4 == 0 && aligned16(g.a)) ? 1 : 0};
    // [x, 1]^T . dy, a dense layer's weight gradient: recorded when a deferred group is open (and B lies in place)
    if (g.b_cs == 1 && g.a_ones_row && defer_record(ctx, A, B, ep, g.m, g.n, g.depth)) return UOCR_OK;
    return launch_mfma(ctx, A, B, ep, g.m, g.n, g.depth, true);
}

bool uocr_conv_mfma_eligible(uocr_ctx* ctx, int dtype, const ConvDims& d, int which /*0 fwd, 1 dgrad, 2 wgrad*/) {
    if (dtype != UOCR_F32 || ctx->opt_mfma == 0) return false;
    if (d.cout % 4 || d.cin % 4) return false;
    if (which == 0 && d.cin % BD) return false;
    if (which == 1 && d.cout % BD) return false;
    if (which == 1 && (size_t)d.kh * d.kw * d.cin * d.cout * sizeof(float) > ws_half(ctx)) return false;
    if (ctx->opt_mfma == 2) return true;
    return d.cin >= 16 && d.cout >= 16;
}

int uocr_conv_fwd_mfma(uocr_ctx* ctx, const void* x, const void* w, const void* b, void* y, const ConvDims& d,
                       double pad_value, int use_bias, int act, double act_alpha) {
    const int M = d.n * d.oh * d.ow, K = d.kh * d.kw * d.cin;
    AConvFwd A((const float*)x, d, (float)pad_value, M);
    BRowMajor B{(const float*)w, d.cout, K, d.cout, aligned16(w) ? 1 : 0};
    Epilogue ep{(float*)y, d.cout, use_bias ? (const float*)b : nullptr, act, (float)act_alpha, 0, -1, nullptr,
                nullptr,    UOCR_ACT_NONE, 0.f};
    return launch_mfma(ctx, A, B, ep, M, d.cout, K, (size_t)M * d.cout * 4 * 8 <= ws_half(ctx));
}

int uocr_conv_dgrad_mfma(uocr_ctx* ctx, const void* dy, const void* w, void* dx, const ConvDims& d,
                         const ActMask& mask) {
    const int M = d.n * d.h * d.w, D = d.kh * d.kw * d.cout;
    // B((kk, oc), ic) = w[(kk * cin + ic) * cout + oc]: the weights as they lie in memory, read along oc
    AConvDgrad A((const float*)dy, d, M);
    BDepthContig B{(const float*)w, d.cout, d.cin, D, d.cin, FastDiv(d.cout), aligned16(w) ? 1 : 0};
    Epilogue ep = plain_epilogue((float*)dx, d.cin, 0);
    ep.mask_y = (const float*)mask.y;
    ep.mask_act = mask.act;
    ep.mask_alpha = (float)mask.alpha;
    // A strided transposed conv only meets the tap rows ky with (iy + ph - ky) % sh == 0 and an output row inside the
    // image; the kernel skips the depth tiles of the others.  When few tiles per block are live, a depth split only
    // adds slabs of zeros and a reduce launch (Char conv_3, 5 x 3 stride (2, 1) on 5 rows: 1 of 5 tap rows live,
    // 27.5 us split 6 ways, 18.5 us unsplit), so the split decision counts the live tiles
    long live = 0;
    for (int iy = 0; iy < d.h; ++iy)
        for (int ky = 0; ky < d.kh; ++ky) {
            const int t = iy + d.ph - ky;
            live += t >= 0 && t % d.sh == 0 && t / d.sh < d.oh;
        }
    const long live_tiles = ((long)D + BD - 1) / BD * live / ((long)d.h * d.kh);
    const bool split_ok = (size_t)M * d.cin * 4 * 8 <= ws_half(ctx) && live_tiles >= 12;
    return launch_mfma(ctx, A, B, ep, M, d.cin, D, split_ok);
}

int uocr_conv_wgrad_mfma(uocr_ctx* ctx, const void* x, const void* dy, void* dw, void* db, const ConvDims& d,
                         double pad_value, int use_bias, int accumulate) {
    const int K = d.kh * d.kw * d.cin, P = d.n * d.oh * d.ow;
    const int M = K + (use_bias ? 1 : 0);
    const int cursor_ok = d.ow >= BD && (long)(d.n + 1) * d.h * d.w * d.cin < (1l << 31) - (long)d.kh * d.w * d.cin;
    AConvWgrad A{(const float*)x, d, (float)pad_value, K, use_bias, P, FastDiv(d.ow), FastDiv(d.oh), cursor_ok};
    BRowMajor B{(const float*)dy, d.cout, P, d.cout, aligned16(dy) ? 1 : 0};
    Epilogue ep{(float*)dw, d.cout, nullptr, UOCR_ACT_NONE, 0.f, accumulate, use_bias ? K : -1, (float*)db,
                nullptr,    UOCR_ACT_NONE, 0.f};
    if (!defer_record(ctx, A, B, ep, M, d.cout, P)) {
        int rc = launch_mfma(ctx, A, B, ep, M, d.cout, P, true);
        if (rc) return rc;
    }
    if (!use_bias && !accumulate) UOCR_HIP(ctx, hipMemsetAsync(db, 0, (size_t)d.cout * sizeof(float), ctx->stream));
    return UOCR_OK;
}

void uocr_gemm_defer_free(uocr_ctx* ctx) {
    delete (GemmDefer*)ctx->gemm_defer;
    ctx->gemm_defer = nullptr;
}

int uocr_gemm_defer_begin(uocr_ctx* ctx) {
    if (!ctx->gemm_defer) ctx->gemm_defer = new GemmDefer();
    GemmDefer* d = (GemmDefer*)ctx->gemm_defer;
    d->on = true;
    d->count = d->nconv = d->ncol = 0;
    return UOCR_OK;
}

int uocr_gemm_defer_flush(uocr_ctx* ctx, int keep_open) {
    GemmDefer* d = (GemmDefer*)ctx->gemm_defer;
    if (!d || !d->on) return UOCR_OK;
    const int rc = defer_flush(ctx);
    d->on = keep_open != 0 && rc == UOCR_OK;
    return rc;
}
