// HBM-bound elementwise kernels: activations, fan-out sums, fills, casts, regularizers,
// fused optimizer steps, NaN check.  Roofline: HBM bandwidth; every kernel moves 16 B per lane
// per access (float4 / double2) over a grid-stride loop capped at 2048 blocks (8 per CU).
//
// Reference semantics restated (paths relative to web_app/components/nn/):
//   Relu / LeakyRelu / Sigmoid ........ layers/layers.py:377-418
//   sum of fan-out gradients .......... models.py:218
//   L1 / L2 ........................... regularizations.py:15-26, applied layers.py:147-155
//   Adam / Momentum / RMSProp ......... optimizers.py:47-98
//   nan_weights ....................... layers/layers.py:139-140
#include <type_traits>

#include "uocr_common.h"

namespace {

template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
    T v[V];
};

template <typename T>
constexpr int vec_width() {
    return 16 / (int)sizeof(T);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- generic N-input map kernel ----------------------------------------------------------
// TS = storage type in HBM, T = the type the functor computes in (TS itself, or float for binary16 storage:
// UOCR_F16 keeps activations as _Float16 and does every operation in float32)
template <typename TS, int NIN>
struct MapArgs {
    TS* out;
    const TS* in[NIN];
};

template <typename TS, typename T, int NIN, int V, typename Op>
__global__ __launch_bounds__(256) void map_kernel(MapArgs<TS, NIN> a, size_t n, Op op) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nvec = n / V;
    for (size_t i = tid; i < nvec; i += stride) {
        Pack<TS, V> x[NIN];
#pragma unroll
        for (int k = 0; k < NIN; ++k) x[k] = *reinterpret_cast<const Pack<TS, V>*>(a.in[k] + i * V);
        Pack<TS, V> r;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            T args[NIN];
#pragma unroll
            for (int k = 0; k < NIN; ++k) args[k] = (T)x[k].v[j];
            r.v[j] = (TS)op(args);
        }
        *reinterpret_cast<Pack<TS, V>*>(a.out + i * V) = r;
    }
    for (size_t i = nvec * V + tid; i < n; i += stride) {
        T args[NIN];
#pragma unroll
        for (int k = 0; k < NIN; ++k) args[k] = (T)a.in[k][i];
        a.out[i] = (TS)op(args);
    }
}

template <typename T, typename TS, int NIN, typename Op>
int launch_map(uocr_ctx* ctx, MapArgs<TS, NIN> a, size_t n, Op op) {
    if (n == 0) return UOCR_OK;
    bool al = aligned16(a.out);
    for (int k = 0; k < NIN; ++k) al = al && aligned16(a.in[k]);
    constexpr int V = vec_width<TS>();
    const unsigned grid = uocr_blocks_for((n + V - 1) / V, 256, UOCR_MAX_GRID);
    if (al)
        hipLaunchKernelGGL((map_kernel<TS, T, NIN, V, Op>), dim3(grid), dim3(256), 0, ctx->stream, a, n, op);
    else
        hipLaunchKernelGGL((map_kernel<TS, T, NIN, 1, Op>), dim3(uocr_blocks_for(n, 256, UOCR_MAX_GRID)),
                           dim3(256), 0, ctx->stream, a, n, op);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

template <typename T>
__device__ __forceinline__ T dev_exp(T x);
template <>
__device__ __forceinline__ float dev_exp<float>(float x) { return expf(x); }
template <>
__device__ __forceinline__ double dev_exp<double>(double x) { return exp(x); }

// ---- activation functors -------------------------------------------------------------------
// mask = (x >= 0) [+ alpha * (x < 0)], y = x * mask: NaN and -0.0 behave as in NumPy.
template <typename T>
struct ReluFwd {
    __device__ T operator()(const T* a) const { return a[0] * (a[0] >= T(0) ? T(1) : T(0)); }
};
template <typename T>
struct LeakyFwd {
    T alpha;
    __device__ T operator()(const T* a) const {
        const T m = (a[0] >= T(0) ? T(1) : T(0)) + alpha * (a[0] < T(0) ? T(1) : T(0));
        return a[0] * m;
    }
};
template <typename T>
struct SigmoidFwd {
    __device__ T operator()(const T* a) const { return T(1) / (T(1) + dev_exp<T>(-a[0])); }
};
// backward: args = {x (stashed input), dy}
template <typename T>
struct ReluBwd {
    __device__ T operator()(const T* a) const { return a[1] * (a[0] >= T(0) ? T(1) : T(0)); }
};
template <typename T>
struct LeakyBwd {
    T alpha;
    __device__ T operator()(const T* a) const {
        const T m = (a[0] >= T(0) ? T(1) : T(0)) + alpha * (a[0] < T(0) ? T(1) : T(0));
        return a[1] * m;
    }
};
// e^{-x}/(1+e^{-x})^2 is even in x: evaluate it at |x| so e <= 1 never overflows (the reference
// formula, layers.py:412-415, overflows to NaN for x < -709 in float64 / x < -88 in float32).
template <typename T>
struct SigmoidBwd {
    __device__ T operator()(const T* a) const {
        const T e = dev_exp<T>(-(a[0] < T(0) ? -a[0] : a[0]));
        const T d = e + T(1);
        return a[1] * e / (d * d);
    }
};

// backward from the layer's OUTPUT y (used when the activation is fused into the producing conv and
// its input is never stored): leaky  y >= 0 <=> x >= 0 for alpha > 0;  sigmoid'  = y (1 - y)
template <typename T>
struct LeakyBwdFromOut {
    T alpha;
    __device__ T operator()(const T* a) const { return a[1] * (a[0] >= T(0) ? T(1) : alpha); }
};
template <typename T>
struct SigmoidBwdFromOut {
    __device__ T operator()(const T* a) const { return a[1] * a[0] * (T(1) - a[0]); }
};

template <typename T>
struct AddOp {
    __device__ T operator()(const T* a) const { return a[0] + a[1]; }
};
template <typename T>
struct AxpyOp {  // args = {y, x}
    T alpha;
    __device__ T operator()(const T* a) const { return a[0] + alpha * a[1]; }
};
template <typename T>
struct ScaleOp {
    T alpha;
    __device__ T operator()(const T* a) const { return a[0] * alpha; }
};

template <typename T, int V>
__global__ __launch_bounds__(256) void fill_kernel(T* x, T value, size_t n) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nvec = n / V;
    Pack<T, V> r;
#pragma unroll
    for (int j = 0; j < V; ++j) r.v[j] = value;
    for (size_t i = tid; i < nvec; i += stride) *reinterpret_cast<Pack<T, V>*>(x + i * V) = r;
    for (size_t i = nvec * V + tid; i < n; i += stride) x[i] = value;
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* src, D* dst, D scale, bool use_scale, size_t n) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += stride) {
        D v = (D)src[i];
        dst[i] = use_scale ? v * scale : v;
    }
}

// dst = (TS)((T)src * scale): the scalar fallback of uocr_u8_to_float
template <typename S, typename TS, typename T>
__global__ __launch_bounds__(256) void convert_scaled_kernel(const S* src, TS* dst, T scale, size_t n) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += stride) dst[i] = (TS)((T)src[i] * scale);
}

// page images arrive as uint8 (datasets.py:16-19 divides by 255): 16 pixels = one 16-byte load per lane, stored
// as 16 values of the activation type (float64 computes the product in double like the reference)
template <typename TS>
__global__ __launch_bounds__(256) void u8_to_float_vec_kernel(const uint8_t* __restrict__ src, TS* __restrict__ dst,
                                                              double scale, size_t ngroups) {
    using T = typename std::conditional<std::is_same<TS, double>::value, double, float>::type;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const T sc = (T)scale;
    for (size_t g = tid; g < ngroups; g += stride) {
        const uint4 raw = reinterpret_cast<const uint4*>(src)[g];
        const uint32_t words[4] = {raw.x, raw.y, raw.z, raw.w};
        constexpr int PER = 16 / (int)sizeof(TS);                      // elements per 16-byte store
#pragma unroll
        for (int q = 0; q < 16 / PER; ++q) {
            Pack<TS, PER> out;
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                const int k = q * PER + e;
                out.v[e] = (TS)((T)((words[k >> 2] >> (8 * (k & 3))) & 0xffu) * sc);
            }
            *reinterpret_cast<Pack<TS, PER>*>(dst + g * 16 + q * PER) = out;
        }
    }
}

// ---- regularizers: grad += dR/dw, partial sums of R in float64 ------------------------------
template <typename T, int KIND /*1 = L1, 2 = L2*/>
__global__ __launch_bounds__(256) void reg_kernel(const T* w, T* grad, size_t n, T strength, double* partial) {
    __shared__ double smem[16];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (size_t i = tid; i < n; i += stride) {
        const T v = w[i];
        if (KIND == 2) {
            grad[i] += strength * T(2) * v;
            acc += (double)v * (double)v;
        } else {
            const T s = v > T(0) ? T(1) : (v < T(0) ? T(-1) : (v == T(0) ? T(0) : v));
            grad[i] += strength * s;
            acc += (double)(v < T(0) ? -v : v);
        }
    }
    acc = block_reduce_sum(acc, smem);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// final reduction of `count` float64 partials by ONE block; out = scale * sum (+ out if accumulate)
__global__ __launch_bounds__(256) void finish_sum_kernel(const double* partial, int count, double scale,
                                                         double* out, int accumulate) {
    __shared__ double smem[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < count; i += blockDim.x) acc += partial[i];
    acc = block_reduce_sum(acc, smem);
    if (threadIdx.x == 0) *out = (accumulate ? *out : 0.0) + scale * acc;
}

// ---- optimizers ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T dev_sqrt(T x);
template <>
__device__ __forceinline__ float dev_sqrt<float>(float x) { return sqrtf(x); }
template <>
__device__ __forceinline__ double dev_sqrt<double>(double x) { return sqrt(x); }

template <typename T, int V>
__global__ __launch_bounds__(256) void adam_kernel(T* w, const T* g, T* v, T* a, size_t n, T lr, T b1, T b2,
                                                   T eps) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nvec = n / V;
    const T c1 = T(1) - b1, c2 = T(1) - b2;
    for (size_t i = tid; i < nvec; i += stride) {
        Pack<T, V> pw = *reinterpret_cast<const Pack<T, V>*>(w + i * V);
        const Pack<T, V> pg = *reinterpret_cast<const Pack<T, V>*>(g + i * V);
        Pack<T, V> pv = *reinterpret_cast<const Pack<T, V>*>(v + i * V);
        Pack<T, V> pa = *reinterpret_cast<const Pack<T, V>*>(a + i * V);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            pv.v[j] = b1 * pv.v[j] + c1 * pg.v[j];
            pa.v[j] = b2 * pa.v[j] + c2 * (pg.v[j] * pg.v[j]);
            pw.v[j] -= lr / (dev_sqrt<T>(pa.v[j]) + eps) * pv.v[j];
        }
        *reinterpret_cast<Pack<T, V>*>(w + i * V) = pw;
        *reinterpret_cast<Pack<T, V>*>(v + i * V) = pv;
        *reinterpret_cast<Pack<T, V>*>(a + i * V) = pa;
    }
    for (size_t i = nvec * V + tid; i < n; i += stride) {
        const T gi = g[i];
        const T vi = b1 * v[i] + c1 * gi;
        const T ai = b2 * a[i] + c2 * (gi * gi);
        v[i] = vi;
        a[i] = ai;
        w[i] -= lr / (dev_sqrt<T>(ai) + eps) * vi;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void momentum_kernel(T* w, const T* g, T* v, size_t n, T lr, T mu) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += stride) {
        const T vi = mu * v[i] - lr * g[i];
        v[i] = vi;
        w[i] += vi;
    }
}

// Momentum step with the regulariser gradients and the gradient reset folded in (the tail of a train step:
// layers.py:147-155 regularize, optimizers.py:75-78 update, layers.py:20-21 clear_grad = 4 launches -> 2).
// Same float operations in the same order as reg_kernel followed by momentum_kernel.
struct RegRanges {
    long long lo[4], hi[4];
    int kind[4];          // 1 = L1, 2 = L2
    double strength[4];
    int n;
};

// OPT 0: Momentum (s1 = velocity; p0 = lr, p1 = momentum); OPT 1: Adam (s1 = velocity, s2 = accumulated;
// p0 = lr, p1 = beta1, p2 = beta2, p3 = eps) -- the update expressions of momentum_kernel / adam_kernel
template <typename T, int OPT>
__device__ __forceinline__ void opt_fused_one(T& wi, T& gi, T& vi, T& ai, long long i, const RegRanges& rr,
                                              bool near_range, double (&acc)[4], T p0, T p1, T p2, T p3) {
    if (near_range) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < rr.n && i >= rr.lo[r] && i < rr.hi[r]) {
                const T strength = (T)rr.strength[r];
                if (rr.kind[r] == 2) {
                    gi += strength * T(2) * wi;
                    acc[r] += (double)wi * (double)wi;
                } else {
                    const T s = wi > T(0) ? T(1) : (wi < T(0) ? T(-1) : (wi == T(0) ? T(0) : wi));
                    gi += strength * s;
                    acc[r] += (double)(wi < T(0) ? -wi : wi);
                }
            }
        }
    }
    if constexpr (OPT == 0) {
        vi = p1 * vi - p0 * gi;
        wi = wi + vi;
    } else {
        vi = p1 * vi + (T(1) - p1) * gi;
        ai = p2 * ai + (T(1) - p2) * (gi * gi);
        wi = wi - p0 / (dev_sqrt<T>(ai) + p3) * vi;
    }
}

// VEC = 4: every thread moves 4 consecutive parameters per trip as one vector per array (the arrays are
// 4-element aligned: the launcher checks); VEC = 1 is the fallback for unaligned views.  The regulariser
// ranges are tested once per vector, element by element only where a range is touched.
// uocr_ctx_set_loss_snapshot: the thread that ends the step's last kernel (the one that stores the regularisation loss)
// copies the net's loss slots into the next row of a ring -- the per-step snapshot of the losses without a copy launch
// of its own (a launch costs its lane 8-10 us inside the page step).  The row index is a device counter: a replayed
// HIP graph advances it by itself.
struct LossSnapshot {
    const double* src;       // null: off
    int count;
    double* ring;            // [ring_len][count]
    int ring_len;
    unsigned* counter;
};
__device__ __forceinline__ void loss_snapshot(const LossSnapshot& s) {
    if (!s.src) return;
    const unsigned k = atomicAdd(s.counter, 1u) % (unsigned)s.ring_len;
    for (int i = 0; i < s.count; ++i) s.ring[(size_t)k * s.count + i] = s.src[i];
}

template <typename T, int OPT, int VEC>
__global__ __launch_bounds__(256) void opt_fused_kernel(T* __restrict__ w, T* __restrict__ g, T* __restrict__ s1,
                                                        T* __restrict__ s2, size_t n, T p0, T p1, T p2, T p3,
                                                        RegRanges rr, double* partial /* [grid][4] */,
                                                        double* __restrict__ loss_out, unsigned* counter,
                                                        int zero_grad, const double* __restrict__ hyper, LossSnapshot snap) {
    if (hyper) {      // hyper-parameters from device memory: a captured HIP graph follows lr / beta changes
        p0 = (T)hyper[0];
        p1 = (T)hyper[1];
        p2 = (T)hyper[2];
        p3 = (T)hyper[3];
    }
    struct alignas(sizeof(T) * VEC) Vec {
        T v[VEC];
    };
    __shared__ double smem[4][4];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t groups = n / VEC;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (size_t q = tid; q < groups; q += stride) {
        const long long i0 = (long long)(q * VEC);
        Vec wv = *reinterpret_cast<const Vec*>(w + i0);
        Vec gv = *reinterpret_cast<const Vec*>(g + i0);
        Vec sv = *reinterpret_cast<const Vec*>(s1 + i0);
        Vec av = sv;
        if constexpr (OPT == 1) av = *reinterpret_cast<const Vec*>(s2 + i0);
        bool near_range = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) near_range |= r < rr.n && i0 < rr.hi[r] && i0 + VEC > rr.lo[r];
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            opt_fused_one<T, OPT>(wv.v[k], gv.v[k], sv.v[k], av.v[k], i0 + k, rr, near_range, acc, p0, p1, p2, p3);
        if (zero_grad) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) gv.v[k] = T(0);
        }
        *reinterpret_cast<Vec*>(w + i0) = wv;
        *reinterpret_cast<Vec*>(g + i0) = gv;
        *reinterpret_cast<Vec*>(s1 + i0) = sv;
        if constexpr (OPT == 1) *reinterpret_cast<Vec*>(s2 + i0) = av;
    }
    for (size_t i = groups * VEC + tid; i < n; i += stride) {      // the n % VEC trailing parameters
        T wi = w[i], gi = g[i], vi = s1[i], ai = OPT == 1 ? s2[i] : T(0);
        opt_fused_one<T, OPT>(wi, gi, vi, ai, (long long)i, rr, rr.n > 0, acc, p0, p1, p2, p3);
        w[i] = wi;
        g[i] = zero_grad ? T(0) : gi;
        s1[i] = vi;
        if constexpr (OPT == 1) s2[i] = ai;
    }
    if (rr.n == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) loss_snapshot(snap);
        return;
    }
    // one block reduction for all ranges: waves through shuffles, the four waves through LDS (fixed order)
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double t = r < rr.n ? wave_reduce_sum(acc[r]) : 0.0;
        if (lane == 0) smem[wv_id][r] = t;
    }
    __syncthreads();
    if (gridDim.x == 1) {                                  // small nets: the one block IS the whole sum, no finish launch
        if (threadIdx.x == 0) {
            double total = 0.0;
            for (int r = 0; r < rr.n; ++r) total += rr.strength[r] * (smem[0][r] + smem[1][r] + smem[2][r] + smem[3][r]);
            *loss_out = total;
            loss_snapshot(snap);
        }
        return;
    }
    // several blocks: publish the block's four sums; the last block to arrive adds all of them up (block order) --
    // no finish launch
    __shared__ int last;
    if (threadIdx.x < 4)
        pub_store(partial + (size_t)blockIdx.x * 4 + threadIdx.x,
                  smem[0][threadIdx.x] + smem[1][threadIdx.x] + smem[2][threadIdx.x] + smem[3][threadIdx.x]);
    if (threadIdx.x == 0) last = sync_arrive(counter) == gridDim.x - 1;       // (threads 0-3 are one wave: drained together)
    __syncthreads();
    if (!last) return;
    // every partial this thread adds is requested before the first one is waited for (agent-scope loads take a trip to the
    // memory side each; one range after the other, one block at a time was 3 x 3 of them in a row), block order kept
    __shared__ double red[16];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    {
        double t[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = threadIdx.x + 256 * k;
                t[k][r] = (i < (int)gridDim.x && r < rr.n) ? pub_load(partial + (size_t)i * 4 + r) : 0.0;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] += t[k][r];
    }
    double total = 0.0;
    for (int r = 0; r < rr.n; ++r) {
        const double sum = block_reduce_sum(a[r], red);
        __syncthreads();
        total += rr.strength[r] * sum;
    }
    if (threadIdx.x == 0) {
        *loss_out = total;
        sync_clear(counter);
        loss_snapshot(snap);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rmsprop_kernel(T* w, const T* g, T* a, size_t n, T lr, T rho, T eps) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n; i += stride) {
        const T gi = g[i];
        const T ai = rho * a[i] + (T(1) - rho) * (gi * gi);
        a[i] = ai;
        w[i] -= lr / (dev_sqrt<T>(ai) + eps) * gi;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void has_nan_kernel(const T* x, size_t n, int32_t* flag) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool found = false;
    for (size_t i = tid; i < n; i += stride) found |= (x[i] != x[i]);
    if (__any(found) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

template <typename T>
int reg_impl(uocr_ctx* ctx, int kind, const void* w, void* grad, size_t n, double strength, double* loss_out,
             int accumulate) {
    const unsigned grid = uocr_blocks_for(n, 256 * 4, 512);
    int rc = uocr_need_workspace(ctx, grid * sizeof(double));
    if (rc) return rc;
    double* partial = (double*)ctx->workspace;
    if (kind == 2)
        hipLaunchKernelGGL((reg_kernel<T, 2>), dim3(grid), dim3(256), 0, ctx->stream, (const T*)w, (T*)grad, n,
                           (T)strength, partial);
    else
        hipLaunchKernelGGL((reg_kernel<T, 1>), dim3(grid), dim3(256), 0, ctx->stream, (const T*)w, (T*)grad, n,
                           (T)strength, partial);
    UOCR_LAUNCH_CHECK(ctx);
    return uocr_finish_sum(ctx, partial, (int)grid, strength, loss_out, accumulate);
}

}  // namespace

int uocr_finish_sum(uocr_ctx* ctx, const double* partial, int count, double scale, double* out, int accumulate) {
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, partial, count, scale, out, accumulate);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

extern "C" {

int uocr_act_fwd(uocr_ctx* ctx, int dtype, int kind, double alpha, const void* x, void* y, size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, x && y);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        MapArgs<TS, 1> a{(TS*)y, {(const TS*)x}};
        switch (kind) {
            case UOCR_ACT_RELU: return launch_map<T>(ctx, a, count, ReluFwd<T>{});
            case UOCR_ACT_LEAKY: return launch_map<T>(ctx, a, count, LeakyFwd<T>{(T)alpha});
            case UOCR_ACT_SIGMOID: return launch_map<T>(ctx, a, count, SigmoidFwd<T>{});
            default: UOCR_FAIL(ctx, UOCR_ERR_ARG, "unknown activation kind %d", kind);
        }
    });
    return UOCR_OK;
}

int uocr_act_bwd(uocr_ctx* ctx, int dtype, int kind, double alpha, const void* x, const void* dy, void* dx,
                 size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, x && dy && dx);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        MapArgs<TS, 2> a{(TS*)dx, {(const TS*)x, (const TS*)dy}};
        switch (kind) {
            case UOCR_ACT_RELU: return launch_map<T>(ctx, a, count, ReluBwd<T>{});
            case UOCR_ACT_LEAKY: return launch_map<T>(ctx, a, count, LeakyBwd<T>{(T)alpha});
            case UOCR_ACT_SIGMOID: return launch_map<T>(ctx, a, count, SigmoidBwd<T>{});
            default: UOCR_FAIL(ctx, UOCR_ERR_ARG, "unknown activation kind %d", kind);
        }
    });
    return UOCR_OK;
}

int uocr_act_bwd_from_output(uocr_ctx* ctx, int dtype, int kind, double alpha, const void* y, const void* dy,
                             void* dx, size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, y && dy && dx);
    UOCR_REQUIRE(ctx, kind == UOCR_ACT_SIGMOID || (kind == UOCR_ACT_LEAKY && alpha > 0.0));
    UOCR_DISPATCH_ACT(ctx, dtype, {
        MapArgs<TS, 2> a{(TS*)dx, {(const TS*)y, (const TS*)dy}};
        if (kind == UOCR_ACT_LEAKY) return launch_map<T>(ctx, a, count, LeakyBwdFromOut<T>{(T)alpha});
        return launch_map<T>(ctx, a, count, SigmoidBwdFromOut<T>{});
    });
    return UOCR_OK;
}

int uocr_add(uocr_ctx* ctx, int dtype, const void* a, const void* b, void* out, size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, a && b && out);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        MapArgs<TS, 2> m{(TS*)out, {(const TS*)a, (const TS*)b}};
        return launch_map<T>(ctx, m, count, AddOp<T>{});
    });
    return UOCR_OK;
}

int uocr_axpy(uocr_ctx* ctx, int dtype, double alpha, const void* x, void* y, size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, x && y);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        MapArgs<TS, 2> m{(TS*)y, {(const TS*)y, (const TS*)x}};
        return launch_map<T>(ctx, m, count, AxpyOp<T>{(T)alpha});
    });
    return UOCR_OK;
}

int uocr_scale(uocr_ctx* ctx, int dtype, double alpha, void* x, size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, x != nullptr);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        MapArgs<TS, 1> m{(TS*)x, {(const TS*)x}};
        return launch_map<T>(ctx, m, count, ScaleOp<T>{(T)alpha});
    });
    return UOCR_OK;
}

int uocr_fill(uocr_ctx* ctx, int dtype, void* x, double value, size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, x != nullptr);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        constexpr int V = vec_width<TS>();
        if (aligned16(x))
            hipLaunchKernelGGL((fill_kernel<TS, V>), dim3(uocr_blocks_for((count + V - 1) / V, 256, UOCR_MAX_GRID)),
                               dim3(256), 0, ctx->stream, (TS*)x, (TS)value, count);
        else
            hipLaunchKernelGGL((fill_kernel<TS, 1>), dim3(uocr_blocks_for(count, 256, UOCR_MAX_GRID)), dim3(256), 0,
                               ctx->stream, (TS*)x, (TS)value, count);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_convert(uocr_ctx* ctx, int src_dtype, const void* src, int dst_dtype, void* dst, size_t count) {
    UOCR_CHECK_CTX(ctx);
    src_dtype = UOCR_DTYPE_BASE(src_dtype);
    dst_dtype = UOCR_DTYPE_BASE(dst_dtype);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, src && dst);
    const dim3 grid(uocr_blocks_for(count, 256, UOCR_MAX_GRID)), block(256);
    if (src_dtype == UOCR_F32 && dst_dtype == UOCR_F64)
        hipLaunchKernelGGL((convert_kernel<float, double>), grid, block, 0, ctx->stream, (const float*)src,
                           (double*)dst, 1.0, false, count);
    else if (src_dtype == UOCR_F64 && dst_dtype == UOCR_F32)
        hipLaunchKernelGGL((convert_kernel<double, float>), grid, block, 0, ctx->stream, (const double*)src,
                           (float*)dst, 1.0f, false, count);
    else if (src_dtype == UOCR_F32 && dst_dtype == UOCR_F16)
        hipLaunchKernelGGL((convert_kernel<float, _Float16>), grid, block, 0, ctx->stream, (const float*)src,
                           (_Float16*)dst, (_Float16)1, false, count);
    else if (src_dtype == UOCR_F16 && dst_dtype == UOCR_F32)
        hipLaunchKernelGGL((convert_kernel<_Float16, float>), grid, block, 0, ctx->stream, (const _Float16*)src,
                           (float*)dst, 1.0f, false, count);
    else if (src_dtype == UOCR_F64 && dst_dtype == UOCR_F16)
        hipLaunchKernelGGL((convert_kernel<double, _Float16>), grid, block, 0, ctx->stream, (const double*)src,
                           (_Float16*)dst, (_Float16)1, false, count);
    else if (src_dtype == UOCR_F16 && dst_dtype == UOCR_F64)
        hipLaunchKernelGGL((convert_kernel<_Float16, double>), grid, block, 0, ctx->stream, (const _Float16*)src,
                           (double*)dst, 1.0, false, count);
    else if (src_dtype == dst_dtype && (src_dtype == UOCR_F32 || src_dtype == UOCR_F64 || src_dtype == UOCR_F16))
        return uocr_d2d(ctx, dst, src, count * (src_dtype == UOCR_F32 ? 4 : src_dtype == UOCR_F64 ? 8 : 2));
    else
        UOCR_FAIL(ctx, UOCR_ERR_DTYPE, "unsupported conversion %d -> %d", src_dtype, dst_dtype);
    UOCR_LAUNCH_CHECK(ctx);
    return UOCR_OK;
}

int uocr_u8_to_float(uocr_ctx* ctx, int dtype, const uint8_t* src, void* dst, double scale, size_t count) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, src && dst);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        if ((count & 15) == 0 && aligned16(src) && aligned16(dst))           // 16 pixels per lane per trip
            hipLaunchKernelGGL((u8_to_float_vec_kernel<TS>), dim3(uocr_blocks_for(count / 16, 256, UOCR_MAX_GRID)),
                               dim3(256), 0, ctx->stream, src, (TS*)dst, scale, count / 16);
        else
            hipLaunchKernelGGL((convert_scaled_kernel<uint8_t, TS, T>), dim3(uocr_blocks_for(count, 256, UOCR_MAX_GRID)),
                               dim3(256), 0, ctx->stream, src, (TS*)dst, (T)scale, count);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_l2_reg(uocr_ctx* ctx, int dtype, const void* w, void* grad, size_t count, double strength,
                double* loss_out, int accumulate_loss) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, w && grad && loss_out && count > 0);
    UOCR_DISPATCH(ctx, dtype, { return reg_impl<T>(ctx, 2, w, grad, count, strength, loss_out, accumulate_loss); });
    return UOCR_OK;
}

int uocr_l1_reg(uocr_ctx* ctx, int dtype, const void* w, void* grad, size_t count, double strength,
                double* loss_out, int accumulate_loss) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, w && grad && loss_out && count > 0);
    UOCR_DISPATCH(ctx, dtype, { return reg_impl<T>(ctx, 1, w, grad, count, strength, loss_out, accumulate_loss); });
    return UOCR_OK;
}

int uocr_adam_step(uocr_ctx* ctx, int dtype, void* w, const void* g, void* v, void* a, size_t count, double lr,
                   double beta1, double beta2, double eps) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, w && g && v && a);
    UOCR_DISPATCH(ctx, dtype, {
        constexpr int V = vec_width<T>();
        if (aligned16(w) && aligned16(g) && aligned16(v) && aligned16(a))
            hipLaunchKernelGGL((adam_kernel<T, V>), dim3(uocr_blocks_for((count + V - 1) / V, 256, UOCR_MAX_GRID)),
                               dim3(256), 0, ctx->stream, (T*)w, (const T*)g, (T*)v, (T*)a, count, (T)lr, (T)beta1,
                               (T)beta2, (T)eps);
        else
            hipLaunchKernelGGL((adam_kernel<T, 1>), dim3(uocr_blocks_for(count, 256, UOCR_MAX_GRID)), dim3(256), 0,
                               ctx->stream, (T*)w, (const T*)g, (T*)v, (T*)a, count, (T)lr, (T)beta1, (T)beta2,
                               (T)eps);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_momentum_step(uocr_ctx* ctx, int dtype, void* w, const void* g, void* v, size_t count, double lr,
                       double momentum) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, w && g && v);
    UOCR_DISPATCH(ctx, dtype, {
        hipLaunchKernelGGL((momentum_kernel<T>), dim3(uocr_blocks_for(count, 256, UOCR_MAX_GRID)), dim3(256), 0,
                           ctx->stream, (T*)w, (const T*)g, (T*)v, count, (T)lr, (T)momentum);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

static int launch_opt_fused(uocr_ctx* ctx, int dtype, int opt, void* w, void* g, void* s1, void* s2, size_t count,
                            double p0, double p1, double p2, double p3, int nranges, const long long* lo,
                            const long long* hi, const int* kind, const double* strength, double* reg_loss_out,
                            int zero_grad, const double* hyper_dev) {
    UOCR_REQUIRE(ctx, nranges >= 0 && nranges <= 4 && (nranges == 0 || (lo && hi && kind && strength)));
    UOCR_REQUIRE(ctx, nranges == 0 || reg_loss_out);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, w && g && s1 && (opt == 0 || s2));
    RegRanges rr{};
    rr.n = nranges;
    for (int r = 0; r < nranges; ++r) {
        UOCR_REQUIRE(ctx, (kind[r] == 1 || kind[r] == 2) && lo[r] >= 0 && lo[r] <= hi[r] && (size_t)hi[r] <= count);
        rr.lo[r] = lo[r];
        rr.hi[r] = hi[r];
        rr.kind[r] = kind[r];
        rr.strength[r] = strength[r];
    }
    // 4 parameters per thread per trip when the arrays allow vector access
    const auto aligned = [](const void* p, size_t to) { return (reinterpret_cast<uintptr_t>(p) % to) == 0; };
    const size_t vec_bytes = 4 * (dtype == UOCR_F64 ? 8 : 4);
    const bool vec = aligned(w, vec_bytes) && aligned(g, vec_bytes) && aligned(s1, vec_bytes) &&
                     (opt == 0 || aligned(s2, vec_bytes));
    // with regulariser sums every block ends in a ticket on ONE counter (same-address atomics are served one after the other)
    // and the last one adds all partials: a block per CU (802 k parameters, 3 ranges: 35.9 us at 784 blocks, 32.5 at 512,
    // 30.2 at 256, 29.9 at 128 -- of which ~18 us are the benchmark's own loss-slot launches)
    const unsigned grid = uocr_blocks_for(vec ? (count + 3) / 4 : count, 256, nranges > 0 ? 256u : 1024u);
    int rc = uocr_need_workspace(ctx, (size_t)grid * 4 * sizeof(double));
    if (rc) return rc;
    double* partial = (double*)ctx->workspace;
    const LossSnapshot snap{ctx->snap_src, ctx->snap_count, ctx->snap_ring, ctx->snap_ring_len, ctx->snap_counter};
    UOCR_DISPATCH(ctx, dtype, {
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, ctx->stream, (T*)w, (T*)g, (T*)s1, (T*)s2, count, (T)p0,
                               (T)p1, (T)p2, (T)p3, rr, partial, reg_loss_out, ctx->sync + 3, zero_grad, hyper_dev, snap);
        };
        if (opt == 0 && vec) launch(opt_fused_kernel<T, 0, 4>);
        else if (opt == 0) launch(opt_fused_kernel<T, 0, 1>);
        else if (vec) launch(opt_fused_kernel<T, 1, 4>);
        else launch(opt_fused_kernel<T, 1, 1>);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_momentum_step_fused(uocr_ctx* ctx, int dtype, void* w, void* g, void* v, size_t count, double lr,
                             double momentum, int nranges, const long long* lo, const long long* hi, const int* kind,
                             const double* strength, double* reg_loss_out, int zero_grad, const double* hyper_dev) {
    UOCR_CHECK_CTX(ctx);
    return launch_opt_fused(ctx, dtype, 0, w, g, v, nullptr, count, lr, momentum, 0.0, 0.0, nranges, lo, hi, kind,
                            strength, reg_loss_out, zero_grad, hyper_dev);
}

int uocr_adam_step_fused(uocr_ctx* ctx, int dtype, void* w, void* g, void* v, void* a, size_t count, double lr,
                         double beta1, double beta2, double eps, int nranges, const long long* lo,
                         const long long* hi, const int* kind, const double* strength, double* reg_loss_out,
                         int zero_grad, const double* hyper_dev) {
    UOCR_CHECK_CTX(ctx);
    return launch_opt_fused(ctx, dtype, 1, w, g, v, a, count, lr, beta1, beta2, eps, nranges, lo, hi, kind, strength,
                            reg_loss_out, zero_grad, hyper_dev);
}

int uocr_ctx_set_loss_snapshot(uocr_ctx* ctx, const double* slots, int count, double* ring, int ring_len,
                               unsigned* counter) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, !slots || (count > 0 && ring && ring_len > 0 && counter));
    ctx->snap_src = slots;
    ctx->snap_count = count;
    ctx->snap_ring = ring;
    ctx->snap_ring_len = ring_len;
    ctx->snap_counter = counter;
    return UOCR_OK;
}

int uocr_rmsprop_step(uocr_ctx* ctx, int dtype, void* w, const void* g, void* a, size_t count, double lr,
                      double rho, double eps) {
    UOCR_CHECK_CTX(ctx);
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, w && g && a);
    UOCR_DISPATCH(ctx, dtype, {
        hipLaunchKernelGGL((rmsprop_kernel<T>), dim3(uocr_blocks_for(count, 256, UOCR_MAX_GRID)), dim3(256), 0,
                           ctx->stream, (T*)w, (const T*)g, (T*)a, count, (T)lr, (T)rho, (T)eps);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_has_nan(uocr_ctx* ctx, int dtype, const void* x, size_t count, int32_t* flag_out) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, flag_out != nullptr);
    UOCR_HIP(ctx, hipMemsetAsync(flag_out, 0, sizeof(int32_t), ctx->stream));
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, x != nullptr);
    UOCR_DISPATCH(ctx, dtype, {
        hipLaunchKernelGGL((has_nan_kernel<T>), dim3(uocr_blocks_for(count, 256, UOCR_MAX_GRID)), dim3(256), 0,
                           ctx->stream, (const T*)x, count, flag_out);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

}  // extern "C"
