// Shared internals of libuniver_hip.so (gfx950 only).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "univer_hip.h"

struct uocr_ctx {
    int device;
    hipStream_t stream;
    bool owns_stream;
    void* workspace;
    size_t workspace_bytes;
    const double* snap_src;   // uocr_ctx_set_loss_snapshot: loss slots the fused optimizer kernels copy into ...
    int snap_count;
    double* snap_ring;        // ... row (counter++ % snap_ring_len) of this ring, or snap_src == null
    int snap_ring_len;
    unsigned* snap_counter;
    void* gemm_defer;    // recorded weight-gradient GEMMs of an open deferred group (gemm_mfma.hip), or null
    void* finish_defer;  // recorded finish kernels + their partial region (finish_group.hip), or null
    unsigned* sync;      // UOCR_SYNC_WORDS arrival counters of the single-launch reductions (loss.hip): zero between launches
    int cu_count;
    int opt_mfma;        // 0 = never, 1 = auto (default), 2 = whenever eligible (tests)
    int opt_fast;        // 0 = generic kernels only, 1 = shape-specialised fast paths (default)
    int opt_tiled;       // 0 = no LDS-tiled conv kernels, 1 = use them where instantiated (default)
    int opt_split;       // MFMA GEMMs split their depth until there are about this many blocks (0 = never)
    int opt_split_min;   // ... but only when that takes at least this many slabs (a 2-way split rarely pays its reduce)
    int opt_xcd;         // 1 = MFMA GEMM blocks are renumbered so that neighbours share an XCD (L2)
    int opt_bm;          // 0 = choose the MFMA GEMM row tile automatically, 64 / 128 = force it (experiments)
    int opt_h16;         // 1 = binary16-MFMA kernels for the small-channel convs in UOCR_F16 mode (default)
    int opt_t32;         // float32 vertical-Toeplitz MFMA kernels for the small-channel convs: bit 0 forward, bit 1 backward-data
    int opt_pair_band;   // rows per band of the strip kernels (0 = about one block per CU)
    int opt_pair_pf;     // row prefetch of the pair forward kernels (-1 auto / 0 / 1 / 2, see conv_pair_strip.hip)
    int opt_h3;          // 1 = the float32 Line output conv forward on error-compensated binary16 MFMAs (conv_h3.hip; experiment)
    int opt_group_blocks; // blocks of a deferred weight-gradient group (0 = four per CU)
    int opt_wgrad_bands;  // row bands per tap / channel group of the direct weight-gradient kernels (0 = by accumulator count)
    int opt_pair_g;      // groups of 16 columns per wave of the strip kernels: 4 (8 waves per block) or 2 (16 waves)
    char err[512];
};

constexpr int UOCR_SYNC_WORDS = 1 << 18;   // 1 MB
// bits of the "t32" option whose kernels this library contains (conv_t32.hip, conv_t32w.hip)
#ifdef UOCR_EXPERIMENTS
constexpr int UOCR_T32_BUILT = 255;
#else
constexpr int UOCR_T32_BUILT = 2;
#endif

#define UOCR_FAIL(ctx, code, ...)                                   \
    do {                                                            \
        if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); \
        return (code);                                              \
    } while (0)

#define UOCR_HIP(ctx, call)                                                              \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess)                                                            \
            UOCR_FAIL(ctx, UOCR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                               \
    } while (0)

#define UOCR_LAUNCH_CHECK(ctx)                                                           \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess)                                                            \
            UOCR_FAIL(ctx, UOCR_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                               \
    } while (0)

#define UOCR_REQUIRE(ctx, cond)                                                          \
    do {                                                                                 \
        if (!(cond)) UOCR_FAIL(ctx, UOCR_ERR_ARG, "bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
    } while (0)

#define UOCR_CHECK_CTX(ctx) \
    do {                    \
        if (!(ctx)) return UOCR_ERR_ARG; \
    } while (0)

// dtype dispatch: BODY sees the type alias T
#define UOCR_DISPATCH(ctx, dtype, ...)                                   \
    do {                                                                 \
        if ((dtype) == UOCR_F32) { using T = float; __VA_ARGS__; }       \
        else if ((dtype) == UOCR_F64) { using T = double; __VA_ARGS__; } \
        else UOCR_FAIL(ctx, UOCR_ERR_DTYPE, "unknown dtype %d", (int)(dtype)); \
    } while (0)

// activation-tensor dispatch: BODY sees TS = the storage type in HBM and T = the type to compute in
// (UOCR_F16: binary16 storage, float32 arithmetic; the gradient-scale bits of dtype are ignored here)
#define UOCR_DISPATCH_ACT(ctx, dtype, ...)                                                        \
    do {                                                                                          \
        const int base_ = UOCR_DTYPE_BASE(dtype);                                                 \
        if (base_ == UOCR_F32) { using TS = float; using T = float; __VA_ARGS__; }                \
        else if (base_ == UOCR_F64) { using TS = double; using T = double; __VA_ARGS__; }         \
        else if (base_ == UOCR_F16) { using TS = _Float16; using T = float; __VA_ARGS__; }        \
        else UOCR_FAIL(ctx, UOCR_ERR_DTYPE, "unknown dtype %d", (int)(dtype));                    \
    } while (0)

// storage-type dispatch for kernels that convert every element on load / store themselves: BODY sees T
#define UOCR_DISPATCH_STORAGE(ctx, dtype, ...)                                                    \
    do {                                                                                          \
        const int base_ = UOCR_DTYPE_BASE(dtype);                                                 \
        if (base_ == UOCR_F32) { using T = float; __VA_ARGS__; }                                  \
        else if (base_ == UOCR_F64) { using T = double; __VA_ARGS__; }                            \
        else if (base_ == UOCR_F16) { using T = _Float16; __VA_ARGS__; }                          \
        else UOCR_FAIL(ctx, UOCR_ERR_DTYPE, "unknown dtype %d", (int)(dtype));                    \
    } while (0)

// 2^k of UOCR_F16_SCALED(k): the loss kernels multiply their gradient by it, the bwd_weight kernels divide
// dw / db by it (1 for the other dtypes)
static inline double uocr_grad_scale(int dtype) {
    return UOCR_DTYPE_BASE(dtype) == UOCR_F16 ? (double)(1u << (UOCR_DTYPE_GRAD_SCALE_LOG2(dtype) & 31)) : 1.0;
}
static inline double uocr_grad_unscale(int dtype) { return 1.0 / uocr_grad_scale(dtype); }

static inline int uocr_need_workspace(uocr_ctx* ctx, size_t bytes) {
    if (bytes > ctx->workspace_bytes)
        UOCR_FAIL(ctx, UOCR_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", bytes,
                  ctx->workspace_bytes);
    return UOCR_OK;
}

static inline unsigned uocr_blocks_for(size_t items, unsigned per_block, unsigned cap) {
    size_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// grid cap for grid-stride HBM-bound kernels: 256 CUs x 8 blocks of 256 threads
constexpr unsigned UOCR_MAX_GRID = 2048;
constexpr int UOCR_WAVE = 64;

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_reduce_max(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// block-wide sum of doubles for blocks of up to 1024 threads; result valid in thread 0
__device__ __forceinline__ double block_reduce_sum(double v, double* smem /* >= 16 doubles */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_reduce_sum(v);
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
    v = (threadIdx.x < (unsigned)nw) ? smem[threadIdx.x] : 0.0;
    if (wid == 0) v = wave_reduce_sum(v);
    return v;
}

// ---- single-launch sums (loss.hip: the cross-entropies; elementwise.hip: the regulariser sums of the fused optimizer tail)
// Blocks hand small float64 partials to each other INSIDE a launch.  gfx950 has one L2 per XCD and they are not
// coherent with each other, so every handed-off word is written by an agent-scope atomic store (write-through, sc1),
// drained (s_waitcnt vmcnt(0)) before the writer's arrival is counted, and read by agent-scope atomic loads (sc1:
// past this CU's L1 and this XCD's L2 copy); no fences.  Counters live in ctx->sync, are zero between launches and
// are put back to zero by the last block that touches them.
__device__ __forceinline__ void pub_store(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double pub_load(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned sync_arrive(unsigned* c) {            // the payload stores of this thread are drained first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned sync_peek(unsigned* c) {
    return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sync_clear(unsigned* c) {
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The last block of a launch to arrive at `counter` adds up `count` published partials (block-strided, fixed order)
// and stores scale * sum; every thread of every block calls this (it contains barriers).
__device__ __forceinline__ void last_block_sum(unsigned* counter, double* partial, double mine, int count, double scale,
                                               double* out, double* smem /* >= 17 doubles */) {
    if (threadIdx.x == 0) {
        pub_store(partial + blockIdx.x, mine);
        const unsigned t = sync_arrive(counter);
        smem[16] = t == gridDim.x - 1 ? 1.0 : 0.0;
    }
    __syncthreads();
    if (smem[16] == 0.0) return;                       // block-uniform
    double acc = 0.0;
    if (count <= 4 * (int)blockDim.x) {                // (the usual case: all of a thread's partials requested at once)
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + k * blockDim.x;
            t[k] = i < count ? pub_load(partial + i) : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += t[k];
    } else {
        for (int i = threadIdx.x; i < count; i += blockDim.x) acc += pub_load(partial + i);
    }
    acc = block_reduce_sum(acc, smem);
    if (threadIdx.x == 0) {
        *out = scale * acc;
        sync_clear(counter);
    }
}

// ---- activation-tensor element access ---------------------------------------------------------------------
// TA = float (UOCR_F32) or _Float16 (UOCR_F16: binary16 in HBM, float32 in registers).  1 / 2 / 4 consecutive
// elements per call as ONE memory instruction (4 / 8 / 16 bytes of float, 2 / 4 / 8 bytes of binary16).
using uocr_h2 = __attribute__((ext_vector_type(2))) _Float16;
using uocr_h4 = __attribute__((ext_vector_type(4))) _Float16;

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const _Float16* p) { return (float)*p; }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const _Float16* p) {
    const uocr_h2 v = *reinterpret_cast<const uocr_h2*>(p);
    return make_float2((float)v.x, (float)v.y);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const _Float16* p) {
    const uocr_h4 v = *reinterpret_cast<const uocr_h4*>(p);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(_Float16* p, float v) { *p = (_Float16)v; }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void st2(_Float16* p, float2 v) {
    uocr_h2 h;
    h.x = (_Float16)v.x;
    h.y = (_Float16)v.y;
    *reinterpret_cast<uocr_h2*>(p) = h;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(_Float16* p, float4 v) {
    uocr_h4 h;
    h.x = (_Float16)v.x;
    h.y = (_Float16)v.y;
    h.z = (_Float16)v.z;
    h.w = (_Float16)v.w;
    *reinterpret_cast<uocr_h4*>(p) = h;
}

// float32 / binary16 activation kernels: BODY sees TA (the float64 mode has its own generic kernels)
#define UOCR_DISPATCH_TA(ctx, dtype, ...)                                                          \
    do {                                                                                           \
        const int base_ = UOCR_DTYPE_BASE(dtype);                                                  \
        if (base_ == UOCR_F32) { using TA = float; __VA_ARGS__; }                                  \
        else if (base_ == UOCR_F16) { using TA = _Float16; __VA_ARGS__; }                          \
        else UOCR_FAIL(ctx, UOCR_ERR_DTYPE, "dtype %d: this kernel exists for float32 / float16 only", (int)(dtype)); \
    } while (0)

// alignment an activation pointer needs for the widest (4-element) access of the dtype
static inline bool uocr_aligned_act(const void* p, int dtype) {
    return (reinterpret_cast<uintptr_t>(p) & (UOCR_DTYPE_BASE(dtype) == UOCR_F16 ? 7u : 15u)) == 0;
}

// d act(x) / dx expressed through the activation OUTPUT y (leaky: alpha > 0 so sign(y) == sign(x))
template <typename T>
__device__ __forceinline__ T act_grad_from_output(T y, int act, T alpha) {
    if (act == UOCR_ACT_LEAKY) return y >= T(0) ? T(1) : alpha;
    if (act == UOCR_ACT_SIGMOID) return y * (T(1) - y);
    return T(1);
}

// A tile staging loop `for (i = tid; i < COUNT; i += NT) lds[i] = in_image(i) ? load(i) : fill` compiles to ONE load, a wait
// and a store per trip: COUNT / NT global round trips in a row in front of every tile (ten for the dy tile of the upsample +
// conv backward-data kernels).  Here a thread issues up to BATCH loads (clamped addresses, no branch) before it waits for the
// first one; `addr(i, inside)` gives element i's clamped source pointer and whether it lies inside the image, `put(i, v,
// inside)` stores it (or the fill value).
template <int COUNT, int NT, int BATCH, typename V, typename Addr, typename Put>
__device__ __forceinline__ void stage_batched(int tid, Addr addr, Put put) {
    constexpr int K = (COUNT + NT - 1) / NT;
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += BATCH) {
        V v[BATCH];
        bool in[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (k0 + k < K) v[k] = *addr(min(tid + NT * (k0 + k), COUNT - 1), in[k]);
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
            if (k0 + k < K) {
                const int i = tid + NT * (k0 + k);
                if (i < COUNT) put(i, v[k], in[k]);
            }
    }
}

// Tile coordinates of a persistent block that walks t = blockIdx.x, += gridDim.x over tiles_x * tiles_y * images tiles:
// the three integer divisions (no hardware divide: ~20 dependent instructions each, in front of the tile's first load)
// are made once, for the first tile and for the stride; every further tile is three scalar adds with carries.
struct TileWalk {
    int strip, trow, img;
    int qx, qy, qz;
    __device__ __forceinline__ TileWalk(int first, int step, int tiles_x, int tiles_y) {
        strip = first % tiles_x;
        int r = first / tiles_x;
        trow = r % tiles_y;
        img = r / tiles_y;
        qx = step % tiles_x;
        r = step / tiles_x;
        qy = r % tiles_y;
        qz = r / tiles_y;
    }
    __device__ __forceinline__ void next(int tiles_x, int tiles_y) {
        strip += qx;
        int carry = strip >= tiles_x;
        strip -= carry ? tiles_x : 0;
        trow += qy + carry;
        carry = trow >= tiles_y;
        trow -= carry ? tiles_y : 0;
        img += qz + carry;
    }
};

// out = (accumulate ? out : 0) + scale * sum(partial[0..count)), one block, deterministic order
int uocr_finish_sum(uocr_ctx* ctx, const double* partial, int count, double scale, double* out, int accumulate);
