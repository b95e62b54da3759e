// Loss kernels: wavefront / block reductions in float64, gradient written in the tensor dtype,
// loss scalar written once (float64) to a device slot -- no host sync (the reference does
// float(loss) per call: losses.py:25,42,57,73).  Roofline: HBM bandwidth.
//
//   SegmentationDice2D / SegmentationJaccard2D ... losses.py:9-42
//   SigmoidCrossEntropy .......................... losses.py:45-57
//   SoftmaxCrossEntropy .......................... losses.py:60-73
#include <algorithm>
#include <type_traits>

#include "uocr_common.h"

namespace {

// store a gradient value: binary16 saturates at its largest finite number instead of overflowing to inf (the power-of-two
// gradient scale of UOCR_F16 is static; a sparse channel early in training can exceed 65504 / 2^k -- an inf would
// become NaN in dw and in the weights)
template <typename T, typename V>
__device__ __forceinline__ T grad_store(V v) {
    if constexpr (std::is_same<T, _Float16>::value) {
        const float f = (float)v;
        return (T)fminf(fmaxf(f, -65504.f), 65504.f);
    } else {
        return (T)v;
    }
}

constexpr double SEG_EPS = 1e-8;   // losses.py:17,36

template <typename T, int V>
struct alignas(16) VecOf {
    T v[V];
};

// Dice / Jaccard stay TWO launches (sums, then gradient).  A one-launch form was built and measured in round 3 (blocks
// keep their chunk of both tensors in registers, publish their partial sums, wait inside the launch for the image's
// other blocks on an arrival counter, then write the gradient: both tensors cross HBM once): 25.2 us against 22.2 for
// the two launches on 32 x 256 x 512 x 1 alone (the wait -- atomic round trip, poll, sc1 re-read of the partials -- is
// longer than the launch boundary it replaces and the chip idles through it), and the page step went from 0.84 to
// 0.96 ms: blocks that wait hold their CU slots while the other lanes' kernels want them.  Dropped.
//
// stage 1: block (chunk, pair=(b,ch)) sums p*g, p, g over its pixel range of image b, channel ch
template <typename T>
__global__ __launch_bounds__(256) void seg_partial_kernel(const T* __restrict__ pred, const T* __restrict__ gt,
                                                          double* __restrict__ partial, int hw, int c, int nchunks) {
    __shared__ double smem[16];
    const int chunk = blockIdx.x, pair = blockIdx.y;
    const int b = pair / c, ch = pair % c;
    const int per = (hw + nchunks - 1) / nchunks;
    const int p0 = chunk * per;
    int p1 = min(hw, p0 + per);
    const T* pp = pred + (size_t)b * hw * c + ch;
    const T* gp = gt + (size_t)b * hw * c + ch;
    double s_pg = 0.0, s_p = 0.0, s_g = 0.0;
    if constexpr (sizeof(T) <= 4) {
        // one channel, 16-byte aligned rows: V = 4 (float) or 8 (binary16) pixels per 16-byte load (the per-chunk
        // sums only change their order of addition within a thread)
        constexpr int V = 16 / (int)sizeof(T);
        if (c == 1 && (hw % V) == 0 && (per % V) == 0 && ((uintptr_t)pred & 15) == 0 && ((uintptr_t)gt & 15) == 0) {
            const VecOf<T, V>* p4 = reinterpret_cast<const VecOf<T, V>*>(pp);
            const VecOf<T, V>* g4 = reinterpret_cast<const VecOf<T, V>*>(gp);
            for (int q = p0 / V + threadIdx.x; q < p1 / V; q += blockDim.x) {
                const VecOf<T, V> pv = p4[q], gv = g4[q];
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    s_pg += (double)pv.v[k] * (double)gv.v[k];
                    s_p += (double)pv.v[k];
                    s_g += (double)gv.v[k];
                }
            }
            p1 = p0;                                     // nothing left for the scalar loop
        }
    }
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const double pv = (double)pp[(size_t)p * c], gv = (double)gp[(size_t)p * c];
        s_pg += pv * gv;
        s_p += pv;
        s_g += gv;
    }
    s_pg = block_reduce_sum(s_pg, smem);
    s_p = block_reduce_sum(s_p, smem);
    s_g = block_reduce_sum(s_g, smem);
    if (threadIdx.x == 0) {
        double* o = partial + ((size_t)pair * nchunks + chunk) * 3;
        o[0] = s_pg;
        o[1] = s_p;
        o[2] = s_g;
    }
}

// stage 2: block (chunk, pair) first sums its pair's chunk partials (num, den -> the gradient is LINEAR in
// the label: dice r = (-2/den) g + 2 num/den^2 ; jaccard r = -((den + num)/den^2) g + num/den^2, the same
// values as losses.py:24,41), then writes its pixel range; block (0,0) also adds up the loss of all pairs.
// out_act == UOCR_ACT_SIGMOID: pred is the output of a Sigmoid and the gradient is taken w.r.t. that
// Sigmoid's INPUT (r * p (1 - p)) -- folds the backward pass of a fused conv+Sigmoid output layer.
template <int KIND>
__device__ __forceinline__ void seg_num_den(double s_pg, double s_p, double s_g, double& num, double& den) {
    num = s_pg + SEG_EPS;
    den = KIND == UOCR_LOSS_DICE ? s_p + s_g + 2 * SEG_EPS             // losses.py:19-20
                                 : s_p + s_g - num + 2 * SEG_EPS;      // losses.py:38
}

template <typename T, int KIND>
__global__ __launch_bounds__(256) void seg_grad_kernel(const T* __restrict__ pred, const T* __restrict__ gt,
                                                       const double* __restrict__ partial, T* __restrict__ grad,
                                                       double* __restrict__ loss_out, int hw, int c, int nchunks,
                                                       int npairs, int out_act, double gscale) {
    __shared__ double smem[16];
    __shared__ double coef[2];
    const int chunk = blockIdx.x, pair = blockIdx.y;
    if (chunk == 0 && pair == 0) {                       // the loss: sum over all pairs
        double loss = 0.0;
        for (int q = threadIdx.x; q < npairs; q += blockDim.x) {
            double s_pg = 0.0, s_p = 0.0, s_g = 0.0;
            for (int k = 0; k < nchunks; ++k) {
                const double* o = partial + ((size_t)q * nchunks + k) * 3;
                s_pg += o[0];
                s_p += o[1];
                s_g += o[2];
            }
            double num, den;
            seg_num_den<KIND>(s_pg, s_p, s_g, num, den);
            loss += KIND == UOCR_LOSS_DICE ? 1.0 - 2.0 * num / den : 1.0 - num / den;   // losses.py:22,40
        }
        loss = block_reduce_sum(loss, smem);
        if (threadIdx.x == 0) *loss_out = loss;
        __syncthreads();
    }
    if (!grad) return;
    if (threadIdx.x == 0) {                              // nchunks <= 64: a serial sum in chunk order
        double s_pg = 0.0, s_p = 0.0, s_g = 0.0;
        for (int k = 0; k < nchunks; ++k) {
            const double* o = partial + ((size_t)pair * nchunks + k) * 3;
            s_pg += o[0];
            s_p += o[1];
            s_g += o[2];
        }
        double num, den;
        seg_num_den<KIND>(s_pg, s_p, s_g, num, den);
        if (KIND == UOCR_LOSS_DICE) {
            coef[0] = -2.0 / den;
            coef[1] = 2.0 * num / (den * den);
        } else {
            coef[0] = -(den + num) / (den * den);
            coef[1] = num / (den * den);
        }
        coef[0] *= gscale;                               // UOCR_F16_SCALED(k): grad * 2^k (exact), else 1
        coef[1] *= gscale;
    }
    __syncthreads();
    const double ca = coef[0], cb = coef[1];
    const int b = pair / c, ch = pair % c;
    const int per = (hw + nchunks - 1) / nchunks;
    const int p0 = chunk * per;
    int p1 = min(hw, p0 + per);
    const size_t base = (size_t)b * hw * c + ch;
    if constexpr (sizeof(T) <= 4) {
        constexpr int V = 16 / (int)sizeof(T);
        if (c == 1 && (hw % V) == 0 && (per % V) == 0 && ((uintptr_t)pred & 15) == 0 && ((uintptr_t)gt & 15) == 0 &&
            ((uintptr_t)grad & 15) == 0) {
            const VecOf<T, V>* p4 = reinterpret_cast<const VecOf<T, V>*>(pred + base);
            const VecOf<T, V>* g4 = reinterpret_cast<const VecOf<T, V>*>(gt + base);
            VecOf<T, V>* o4 = reinterpret_cast<VecOf<T, V>*>(grad + base);
            const bool sig = out_act == UOCR_ACT_SIGMOID;
            for (int q = p0 / V + threadIdx.x; q < p1 / V; q += blockDim.x) {
                const VecOf<T, V> gv = g4[q];
                double r[V];
#pragma unroll
                for (int k = 0; k < V; ++k) r[k] = ca * (double)gv.v[k] + cb;
                if (sig) {
                    const VecOf<T, V> pv = p4[q];
#pragma unroll
                    for (int k = 0; k < V; ++k) r[k] *= (double)pv.v[k] * (1.0 - (double)pv.v[k]);
                }
                VecOf<T, V> out;
#pragma unroll
                for (int k = 0; k < V; ++k) out.v[k] = grad_store<T>(r[k]);
                o4[q] = out;
            }
            p1 = p0;
        }
    }
    for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const size_t idx = base + (size_t)p * c;
        double r = ca * (double)gt[idx] + cb;
        if (out_act == UOCR_ACT_SIGMOID) {
            const double pv = (double)pred[idx];
            r *= pv * (1.0 - pv);
        }
        grad[idx] = grad_store<T>(r);
    }
}

// ---- c = 1 / 2 / 4 channels (all my_model nets): one block walks a contiguous pixel range of image b with
// 16-byte accesses (4 float / 8 binary16 elements = V / C pixels x C channels) and keeps the three sums of every
// channel in registers; partial[(b*C + ch)][chunk][3] as above.  The generic kernels read one channel of an
// interleaved tensor with stride C: a quarter of every line for the Line net's two maps.
template <typename T, int C>
__global__ __launch_bounds__(256) void seg_partial_vec_kernel(const T* __restrict__ pred, const T* __restrict__ gt,
                                                              double* __restrict__ partial, int hw, int nchunks) {
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ double smem[16];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const size_t nvec = (size_t)hw * C / V;                          // vectors per image (hw * C % V == 0: launcher)
    const size_t per = (nvec + nchunks - 1) / nchunks;
    const size_t v0 = (size_t)chunk * per, v1 = min(nvec, v0 + per);
    const VecOf<T, V>* p4 = reinterpret_cast<const VecOf<T, V>*>(pred + (size_t)b * hw * C);
    const VecOf<T, V>* g4 = reinterpret_cast<const VecOf<T, V>*>(gt + (size_t)b * hw * C);
    // float partial sums per thread (<= 64 terms of |x| <= 1 each per trip batch), double across trips
    double s_pg[C], s_p[C], s_g[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) s_pg[ch] = s_p[ch] = s_g[ch] = 0.0;
    auto add = [&](const VecOf<T, V>& pv, const VecOf<T, V>& gv) {
        float a_pg[C], a_p[C], a_g[C];
#pragma unroll
        for (int ch = 0; ch < C; ++ch) a_pg[ch] = a_p[ch] = a_g[ch] = 0.f;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float pf = (float)pv.v[k], gf = (float)gv.v[k];
            a_pg[k % C] += pf * gf;
            a_p[k % C] += pf;
            a_g[k % C] += gf;
        }
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            s_pg[ch] += (double)a_pg[ch];
            s_p[ch] += (double)a_p[ch];
            s_g[ch] += (double)a_g[ch];
        }
    };
    // U vectors of each tensor in flight per thread (2 x 16 bytes per trip left the kernel waiting for HBM: 2 blocks of
    // 256 threads per CU hold 16 KB in flight; the vectors are added in the order of the one-at-a-time loop)
    constexpr int U = 4;
    size_t q = v0 + threadIdx.x;
    for (; q + (U - 1) * blockDim.x < v1; q += U * blockDim.x) {
        VecOf<T, V> pv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pv[u] = p4[q + u * blockDim.x];
            gv[u] = g4[q + u * blockDim.x];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) add(pv[u], gv[u]);
    }
    for (; q < v1; q += blockDim.x) add(p4[q], g4[q]);
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        const double t_pg = block_reduce_sum(s_pg[ch], smem);
        const double t_p = block_reduce_sum(s_p[ch], smem);
        const double t_g = block_reduce_sum(s_g[ch], smem);
        if (threadIdx.x == 0) {
            double* o = partial + ((size_t)(b * C + ch) * nchunks + chunk) * 3;
            o[0] = t_pg;
            o[1] = t_p;
            o[2] = t_g;
        }
    }
}

// The gradient kernel has its OWN, finer chunking (`nchunks` blocks per image, up to 512: an HBM stream wants many
// blocks; the sums kernel ends in block reductions and is better off with fewer, `pchunks`): wave 0 adds the
// pchunks partial sums of the image's channels (lanes over the chunks, then the wave tree: a fixed order).
template <typename T, int C, int KIND>
__global__ __launch_bounds__(256) void seg_grad_vec_kernel(const T* __restrict__ pred, const T* __restrict__ gt,
                                                           const double* __restrict__ partial, T* __restrict__ grad,
                                                           double* __restrict__ loss_out, int hw, int pchunks,
                                                           int nchunks, int npairs, int out_act, double gscale) {
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ double smem[16];
    __shared__ float coef[C][2];
    // grid row 0 is the loss row: its block 0 adds up the loss of all (image, channel) pairs -- wave w takes pairs w,
    // w + 4, ...: lanes over the chunks, then the wave tree -- while the rows behind it stream; the other blocks of the
    // row leave at once.  (As a prologue of block (0, 0), with one thread per pair walking the chunks, it was the
    // critical path of the whole launch: 12 of 34 us at 8 x 1024 x 2048.)
    if (blockIdx.y == 0) {
        if (blockIdx.x != 0) return;
        // groups of gs = 16 / 32 / 64 lanes (the smallest that holds the chunks) take one pair each: 256 / gs pairs per
        // round; butterfly sums inside a group (every lane ends with the total, a fixed tree)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int gs = pchunks <= 16 ? 16 : pchunks <= 32 ? 32 : 64;
        const int sub = threadIdx.x % gs, grp = threadIdx.x / gs, ngrp = 256 / gs;
        double loss = 0.0;
        for (int q0 = 0; q0 < npairs; q0 += ngrp) {
            const int q = q0 + grp;
            double s_pg = 0.0, s_p = 0.0, s_g = 0.0;
            if (q < npairs)
                for (int k = sub; k < pchunks; k += gs) {
                    const double* o = partial + ((size_t)q * pchunks + k) * 3;
                    s_pg += o[0];
                    s_p += o[1];
                    s_g += o[2];
                }
            for (int off = gs >> 1; off > 0; off >>= 1) {
                s_pg += __shfl_xor(s_pg, off, 64);
                s_p += __shfl_xor(s_p, off, 64);
                s_g += __shfl_xor(s_g, off, 64);
            }
            double num, den;
            seg_num_den<KIND>(s_pg, s_p, s_g, num, den);
            if (sub == 0 && q < npairs) loss += KIND == UOCR_LOSS_DICE ? 1.0 - 2.0 * num / den : 1.0 - num / den;   // losses.py:22,40
        }
        // the group leaders' terms: lanes 0, gs, 2 gs, ... of the wave
        loss = wave_reduce_sum(loss);
        if (lane == 0) smem[wv] = loss;
        __syncthreads();
        if (threadIdx.x == 0) *loss_out = smem[0] + smem[1] + smem[2] + smem[3];
        return;
    }
    const int chunk = blockIdx.x, b = blockIdx.y - 1;
    if (!grad) return;
    if (threadIdx.x < 64) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            double s_pg = 0.0, s_p = 0.0, s_g = 0.0;
            for (int k = threadIdx.x; k < pchunks; k += 64) {
                const double* o = partial + ((size_t)(b * C + ch) * pchunks + k) * 3;
                s_pg += o[0];
                s_p += o[1];
                s_g += o[2];
            }
            s_pg = wave_reduce_sum(s_pg);
            s_p = wave_reduce_sum(s_p);
            s_g = wave_reduce_sum(s_g);
            if (threadIdx.x == 0) {
                double num, den, ca, cb;
                seg_num_den<KIND>(s_pg, s_p, s_g, num, den);
                if (KIND == UOCR_LOSS_DICE) {
                    ca = -2.0 / den;
                    cb = 2.0 * num / (den * den);
                } else {
                    ca = -(den + num) / (den * den);
                    cb = num / (den * den);
                }
                // the gradient is linear in the label with these two coefficients (computed in float64, applied in
                // float32: the label is 0 / 1 and the result is stored in float32 or binary16 anyway)
                coef[ch][0] = (float)(ca * gscale);
                coef[ch][1] = (float)(cb * gscale);
            }
        }
    }
    __syncthreads();
    float ca[C], cb[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        ca[ch] = coef[ch][0];
        cb[ch] = coef[ch][1];
    }
    const size_t nvec = (size_t)hw * C / V;
    const size_t per = (nvec + nchunks - 1) / nchunks;
    const size_t v0 = (size_t)chunk * per, v1 = min(nvec, v0 + per);
    const size_t base = (size_t)b * hw * C;
    const VecOf<T, V>* p4 = reinterpret_cast<const VecOf<T, V>*>(pred + base);
    const VecOf<T, V>* g4 = reinterpret_cast<const VecOf<T, V>*>(gt + base);
    VecOf<T, V>* o4 = reinterpret_cast<VecOf<T, V>*>(grad + base);
    const bool sig = out_act == UOCR_ACT_SIGMOID;
    // loads of U trips in flight (see seg_partial_vec_kernel); the Sigmoid case is a loop of its own: a per-load
    // "register or load" choice on a run-time flag makes hipcc branch around every load and wait for each
    constexpr int U = 4;
    auto run = [&](auto sigtag) {
        constexpr bool SIG = decltype(sigtag)::value;
        auto one = [&](const VecOf<T, V>& gv, const VecOf<T, V>& pv) {
            float r[V];
#pragma unroll
            for (int k = 0; k < V; ++k) r[k] = ca[k % C] * (float)gv.v[k] + cb[k % C];
            if constexpr (SIG) {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float pf = (float)pv.v[k];
                    r[k] *= pf * (1.f - pf);
                }
            }
            VecOf<T, V> out;
#pragma unroll
            for (int k = 0; k < V; ++k) out.v[k] = grad_store<T>(r[k]);
            return out;
        };
        size_t q = v0 + threadIdx.x;
        for (; q + (U - 1) * blockDim.x < v1; q += U * blockDim.x) {
            VecOf<T, V> gv[U], pv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                gv[u] = g4[q + u * blockDim.x];
                if constexpr (SIG) pv[u] = p4[q + u * blockDim.x];
                else pv[u] = gv[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) o4[q + u * blockDim.x] = one(gv[u], pv[u]);
        }
        for (; q < v1; q += blockDim.x) {
            const VecOf<T, V> gv = g4[q];
            o4[q] = one(gv, SIG ? p4[q] : gv);
        }
    };
    if (sig) run(std::true_type{});
    else run(std::false_type{});
}

// SoftmaxCrossEntropy (losses.py:60-73).  A row of the Char net has 162 classes: one wave per row leaves 2/3 of
// the third trip idle and, worse, did all its transcendental work in float64 (~50 instructions per exp).  Here a
// row is handled by RL = 16 / 32 / 64 lanes (the smallest that covers it in <= 4 trips), so a wave works on
// 64 / RL rows at once, and float32 / binary16 tensors are exponentiated in float32 (relative error 1e-7, the
// tolerance of the float32 mode is 1e-5); the row losses are still summed in float64.  float64 tensors keep
// float64 arithmetic throughout.
template <typename T, typename CT, int RL>
__global__ __launch_bounds__(256) void softmax_ce_kernel(const T* __restrict__ pred, const T* __restrict__ gt,
                                                         T* __restrict__ grad, double* block_loss, unsigned* counter,
                                                         double* __restrict__ loss_out, int m, int c, double gscale) {
    constexpr int RPW = 64 / RL;                          // rows per wave
    __shared__ double smem[17];
    __shared__ double rows[4][RPW];
    const int lane = threadIdx.x & 63, sub = lane % RL;
    const int row = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * RPW + lane / RL;
    const bool live = row < m;
    const int rr = live ? row : m - 1;
    const T* x = pred + (size_t)rr * c;
    const T* g = gt + (size_t)rr * c;
    CT mx = -INFINITY;
    for (int j = sub; j < c; j += RL) mx = fmax(mx, (CT)x[j]);
#pragma unroll
    for (int off = RL / 2; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    CT se = 0;
    for (int j = sub; j < c; j += RL) se += std::is_same<CT, float>::value ? (CT)expf((float)((CT)x[j] - mx)) : (CT)exp((double)((CT)x[j] - mx));
#pragma unroll
    for (int off = RL / 2; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
    const CT lse = std::is_same<CT, float>::value ? (CT)logf((float)se) : (CT)log((double)se);
    const CT inv = CT(1) / se, scale = (CT)(gscale / (double)m);
    double loss = 0.0;
    for (int j = sub; j < c; j += RL) {
        const CT z = (CT)x[j] - mx;
        const CT gv = (CT)g[j];
        if (gv != CT(0)) loss -= (double)(gv * (z - lse));
        if (grad && live) {
            const CT e = std::is_same<CT, float>::value ? (CT)expf((float)z) : (CT)exp((double)z);
            grad[(size_t)row * c + j] = grad_store<T>((e * inv - gv) * scale);
        }
    }
#pragma unroll
    for (int off = RL / 2; off > 0; off >>= 1) loss += __shfl_xor(loss, off, 64);
    // the block's rows in row order, then the blocks in block order by the last block to arrive (one launch: no
    // finish kernel; the sum groups the rows by block, still a fixed order)
    if (sub == 0) rows[threadIdx.x >> 6][lane / RL] = live ? loss : 0.0;
    __syncthreads();
    double mine = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int r = 0; r < RPW; ++r) mine += rows[w][r];
    }
    last_block_sum(counter, block_loss, mine, (int)gridDim.x, 1.0 / (double)m, loss_out, smem);
}

template <typename T>
__global__ __launch_bounds__(256) void sigmoid_ce_kernel(const T* __restrict__ pred, const T* __restrict__ gt,
                                                         T* __restrict__ grad, double* partial, unsigned* counter,
                                                         double* __restrict__ loss_out, size_t total, double inv_m,
                                                         double gscale) {
    __shared__ double smem[17];
    double acc = 0.0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const double x = (double)pred[idx], g = (double)gt[idx];
        const double p = 1.0 / (1.0 + exp(-x));
        acc -= g * log(p) + (1.0 - g) * log(1.0 - p);
        if (grad) grad[idx] = grad_store<T>((g * (p - 1.0) + (1.0 - g) * p) * inv_m * gscale);
    }
    acc = block_reduce_sum(acc, smem);
    __syncthreads();
    last_block_sum(counter, partial, acc, (int)gridDim.x, inv_m, loss_out, smem);
}

}  // namespace

extern "C" {

int uocr_seg_loss(uocr_ctx* ctx, int dtype, int kind, const void* pred, const void* gt, void* grad,
                  double* loss_out, int n, int hw, int c, int out_act) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, pred && gt && loss_out && n > 0 && hw > 0 && c > 0);
    UOCR_REQUIRE(ctx, kind == UOCR_LOSS_DICE || kind == UOCR_LOSS_JACCARD);
    UOCR_REQUIRE(ctx, out_act == UOCR_ACT_NONE || out_act == UOCR_ACT_SIGMOID);
    const int npairs = n * c;
    int nchunks = (hw + 8191) / 8192;
    if (nchunks > 64) nchunks = 64;
    const size_t part_bytes = (size_t)npairs * nchunks * 3 * sizeof(double);
    int rc = uocr_need_workspace(ctx, part_bytes);
    if (rc) return rc;
    double* partial = (double*)ctx->workspace;
    UOCR_REQUIRE(ctx, (size_t)hw * c < (size_t)INT32_MAX && npairs <= 65535);
    const double gscale = uocr_grad_scale(dtype);
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int elem = UOCR_DTYPE_BASE(dtype) == UOCR_F16 ? 2 : UOCR_DTYPE_BASE(dtype) == UOCR_F32 ? 4 : 8;
    const bool vec = elem <= 4 && (c == 1 || c == 2 || c == 4) && ((size_t)hw * c * elem) % 16 == 0 && al16(pred) &&
                     al16(gt) && (!grad || al16(grad)) && n < 65535;
    if (vec) {
        // channel-interleaved 16-byte kernels.  The sums kernel keeps the chunking of the generic kernels (<= 64 chunks
        // of >= 8192 pixels); the gradient kernel streams in chunks of 1024 vectors (256 threads x 4 in flight), at most
        // 512 per image
        const size_t nvec = (size_t)hw * c * elem / 16;
        const int gchunks = (int)std::min<size_t>(512, std::max<size_t>(1, (nvec + 1023) / 1024));
        // (more than 64 chunks for the sums kernel: 128 / 256 measured the same at 8 x 1024 x 2048, tools/dev/dice_sweep.py)
        auto run = [&](auto tag, auto ctag) {
            using T = decltype(tag);
            constexpr int C = decltype(ctag)::value;
            hipLaunchKernelGGL((seg_partial_vec_kernel<T, C>), dim3(nchunks, n), dim3(256), 0, ctx->stream, (const T*)pred,
                               (const T*)gt, partial, hw, nchunks);
            const dim3 ggrid = grad ? dim3(gchunks, n + 1) : dim3(1, 1);       // row 0: the loss
            if (kind == UOCR_LOSS_DICE)
                hipLaunchKernelGGL((seg_grad_vec_kernel<T, C, UOCR_LOSS_DICE>), ggrid, dim3(256), 0, ctx->stream,
                                   (const T*)pred, (const T*)gt, (const double*)partial, (T*)grad, loss_out, hw, nchunks,
                                   gchunks, npairs, out_act, gscale);
            else
                hipLaunchKernelGGL((seg_grad_vec_kernel<T, C, UOCR_LOSS_JACCARD>), ggrid, dim3(256), 0, ctx->stream,
                                   (const T*)pred, (const T*)gt, (const double*)partial, (T*)grad, loss_out, hw, nchunks,
                                   gchunks, npairs, out_act, gscale);
        };
        auto by_c = [&](auto tag) {
            if (c == 1) run(tag, std::integral_constant<int, 1>{});
            else if (c == 2) run(tag, std::integral_constant<int, 2>{});
            else run(tag, std::integral_constant<int, 4>{});
        };
        if (elem == 4) by_c(float{});
        else by_c(_Float16{});
        UOCR_LAUNCH_CHECK(ctx);
        return UOCR_OK;
    }
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((seg_partial_kernel<T>), dim3(nchunks, npairs), dim3(256), 0, ctx->stream, (const T*)pred,
                           (const T*)gt, partial, hw, c, nchunks);
        UOCR_LAUNCH_CHECK(ctx);
        // without a gradient only block (0,0) has work
        const dim3 grid = grad ? dim3(nchunks, npairs) : dim3(1, 1);
        if (kind == UOCR_LOSS_DICE)
            hipLaunchKernelGGL((seg_grad_kernel<T, UOCR_LOSS_DICE>), grid, dim3(256), 0, ctx->stream, (const T*)pred,
                               (const T*)gt, (const double*)partial, (T*)grad, loss_out, hw, c, nchunks, npairs,
                               out_act, gscale);
        else
            hipLaunchKernelGGL((seg_grad_kernel<T, UOCR_LOSS_JACCARD>), grid, dim3(256), 0, ctx->stream,
                               (const T*)pred, (const T*)gt, (const double*)partial, (T*)grad, loss_out, hw, c,
                               nchunks, npairs, out_act, gscale);
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_softmax_ce(uocr_ctx* ctx, int dtype, const void* pred, const void* gt, void* grad, double* loss_out,
                    int m, int c) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, pred && gt && loss_out && m > 0 && c > 0);
    int rc = uocr_need_workspace(ctx, ((size_t)m / 4 + 1) * sizeof(double));
    if (rc) return rc;
    double* block_loss = (double*)ctx->workspace;
    const double gscale = uocr_grad_scale(dtype);
    UOCR_DISPATCH_ACT(ctx, dtype, {
        auto launch = [&](auto rl_tag) {
            constexpr int RL = decltype(rl_tag)::value;
            const int rows_per_block = 4 * (64 / RL);
            hipLaunchKernelGGL((softmax_ce_kernel<TS, T, RL>), dim3((m + rows_per_block - 1) / rows_per_block), dim3(256),
                               0, ctx->stream, (const TS*)pred, (const TS*)gt, (TS*)grad, block_loss, ctx->sync + 1, loss_out,
                               m, c, gscale);
        };
        if (c <= 64) launch(std::integral_constant<int, 16>{});
        else if (c <= 128) launch(std::integral_constant<int, 32>{});
        else launch(std::integral_constant<int, 64>{});
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

int uocr_sigmoid_ce(uocr_ctx* ctx, int dtype, const void* pred, const void* gt, void* grad, double* loss_out,
                    int m, size_t count) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, pred && gt && loss_out && m > 0 && count > 0);
    const unsigned grid = uocr_blocks_for(count, 256 * 4, 512);
    int rc = uocr_need_workspace(ctx, grid * sizeof(double));
    if (rc) return rc;
    double* partial = (double*)ctx->workspace;
    UOCR_DISPATCH_STORAGE(ctx, dtype, {
        hipLaunchKernelGGL((sigmoid_ce_kernel<T>), dim3(grid), dim3(256), 0, ctx->stream, (const T*)pred,
                           (const T*)gt, (T*)grad, partial, ctx->sync + 2, loss_out, count, 1.0 / (double)m,
                           uocr_grad_scale(dtype));
        UOCR_LAUNCH_CHECK(ctx);
    });
    return UOCR_OK;
}

}  // extern "C"
