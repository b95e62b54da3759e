/*
 * univer_hip.h -- C ABI of libuniver_hip.so, the MI355X (gfx950) backend of the
 * univer-ocr self-written deep-learning framework (KerkDovan/univer-ocr).
 *
 * The reference has no FFI: its device seam is `CP` (web_app/components/nn/gpu.py:5-29) plus the
 * per-layer closure pair `_forward_gpu/_backward_gpu` (nn/layers/layers.py:169-197), implemented
 * with CuPy and numba.cuda JIT kernels.  Every entry point below replaces one of those closures
 * (or one CuPy expression on the hot path) and cites the reference lines it stands in for.
 * Paths are relative to /root/reference/web_app/components/nn/.
 *
 * Conventions
 *  - plain pointers and sizes only; all tensor pointers are DEVICE pointers, caller-owned,
 *    C-contiguous, NHWC for images, (kh,kw,Cin,Cout) for conv weights, (n_in+1,n_out) with the
 *    bias as the LAST ROW for dense weights (layers.py:326-338);
 *  - `dtype` selects the arithmetic/storage type of every tensor argument of the call:
 *    UOCR_F32 (the production type), UOCR_F64 (the reference's own type, used by the parity
 *    and numeric-gradient tests) or UOCR_F16 (BASELINE configs[4], the HBM-bound high-resolution
 *    regime): ACTIVATION tensors (x, y, dy, dx, pred, gt, grad, activation masks) are IEEE binary16
 *    in HBM, PARAMETER tensors (w, b, dw, db, optimizer state) stay float32, every sum is
 *    accumulated in float32 (float64 for the long reductions, as in the other modes).  Activation
 *    GRADIENTS of the F16 mode carry a power-of-two scale so that Dice gradients of a 2-Mpixel page
 *    (~1e-6) do not fall below binary16's range: dtype = UOCR_F16_SCALED(k) makes the loss entry
 *    points write grad * 2^k and the bwd_weight entry points multiply dw/db by 2^-k (exact);
 *    every other entry point is linear in the gradient and ignores k (UOCR_F16 == k = 0);
 *  - every function returns 0 (UOCR_OK) or a negative UOCR_ERR_* code and never throws;
 *    uocr_last_error(ctx) returns a human readable message for the last failure on that ctx;
 *  - all work is enqueued asynchronously on the ctx's HIP stream; nothing synchronises unless
 *    its name ends in _sync.  No call allocates device memory after uocr_ctx_create /
 *    uocr_ctx_reserve_workspace, so a sequence of calls can be captured into a HIP graph;
 *  - loss scalars are written as float64 to a caller-provided device slot and fetched by the
 *    caller when it wants them (the reference does `float(loss)` = one D2H sync per loss).
 */
#ifndef UNIVER_HIP_H
#define UNIVER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UOCR_ABI_VERSION 4

typedef struct uocr_ctx uocr_ctx;

enum { UOCR_F32 = 0, UOCR_F64 = 1, UOCR_F16 = 2 };
/* F16 with activation gradients scaled by 2^k, 0 <= k <= 30 (see the conventions above) */
#define UOCR_F16_SCALED(k) (UOCR_F16 | ((k) << 8))
#define UOCR_DTYPE_BASE(dtype) ((dtype) & 0xff)
#define UOCR_DTYPE_GRAD_SCALE_LOG2(dtype) (((dtype) >> 8) & 0xff)
enum { UOCR_ACT_NONE = 0, UOCR_ACT_RELU = 1, UOCR_ACT_LEAKY = 2, UOCR_ACT_SIGMOID = 3 };
enum { UOCR_LOSS_DICE = 0, UOCR_LOSS_JACCARD = 1 };
enum {
    UOCR_OK = 0,
    UOCR_ERR_ARG = -1,         /* null pointer, negative size, inconsistent shape */
    UOCR_ERR_DTYPE = -2,       /* dtype is not UOCR_F32 / UOCR_F64 / UOCR_F16 (or F16 where a kernel has no F16 form) */
    UOCR_ERR_HIP = -3,         /* a HIP runtime call failed (see uocr_last_error) */
    UOCR_ERR_WORKSPACE = -4,   /* ctx workspace too small: call uocr_ctx_reserve_workspace */
    UOCR_ERR_UNSUPPORTED = -5, /* shape outside what the kernels implement */
    UOCR_ERR_RCCL = -6         /* librccl.so.1 missing, or an RCCL call failed (see uocr_last_error) */
};

/* ---- context, memory, events (stand in for CP / cupy, gpu.py:5-29) ------------------------- */
int uocr_abi_version(void);
int uocr_ctx_create(int device, size_t workspace_bytes, uocr_ctx** out);
/* a context whose own stream runs on a subset of the compute units (hipExtStreamCreateWithCUMask; bit i of the mask =
 * compute unit i in the driver's numbering, which deals consecutive bits round-robin over the 8 XCDs of an MI355X: bits
 * with i % 8 in a set = those whole XCDs).  Independent nets trained on concurrent lanes (PageTrainer) can be given
 * disjoint partitions; kernels that size their grid by the CU count use the partition's. */
int uocr_ctx_create_cu_mask(int device, size_t workspace_bytes, const uint32_t* cu_mask, int mask_words, uocr_ctx** out);
int uocr_ctx_destroy(uocr_ctx* ctx);
/* run on a caller-owned hipStream_t (0 = the legacy default stream) instead of the ctx's own */
int uocr_ctx_set_stream(uocr_ctx* ctx, void* hip_stream);
void* uocr_ctx_get_stream(uocr_ctx* ctx);
/* kernel selection knobs (for parity tests and A/B timing): "mfma" = 0 never / 1 auto / 2 whenever
 * eligible; "fast_paths" = 0 generic kernels only / 1 shape-specialised kernels (default);
 * "tiled" = 0 / 1 (default) LDS-tiled conv kernels where instantiated; MFMA GEMM tuning:
 * "split_blocks" (1024: split the depth until about this many blocks exist, 0 = never),
 * "split_min" (3: smallest number of slabs worth a reduce pass), "gemm_bm" (0 auto / 64 / 128 row
 * tile), "xcd_remap" (1: neighbouring tiles are numbered onto the same XCD / L2), "h16" (1: UOCR_F16
 * convolutions with 1-4 channels run on binary16 MFMAs -- the float32 weights enter the matrix cores
 * rounded to binary16, accumulation stays float32; 0: float32 vector arithmetic on the binary16 data),
 * "t32" (bit mask, default 2: which float32 small-channel convolutions use the float32-MFMA Toeplitz
 * kernels: 1 forward / 2 backward-data / 4 upsample+conv backward-data of the 4-channel layers,
 * 8 / 16 / 32 the same for 1-channel layers, 64 / 128 the float32-MFMA weight gradients of the stride-1 /
 * stride-2 5x5 convolutions -- the default library contains bit 2 only, the others need a library built with
 * UOCR_BUILD_EXPERIMENTS=1 ./build.sh and are refused with UOCR_ERR_UNSUPPORTED otherwise; likewise "h3" = 1, the
 * float32 Line output conv forward on error-compensated binary16 MFMAs); the Monochrome pair kernels:
 * "pair_band" (rows per band, 0 auto), "pair_g" (4 / 2 groups of 16 columns per wave), "pair_pf" (row prefetch form
 * of the forward kernels, -1 auto); "wgrad_bands" (row bands per tap / channel group of the direct weight-gradient
 * kernels, 0 = 64 or 512 by the kernel's accumulator count).  Results do not depend on any of them beyond float32
 * summation order ("h16": beyond the binary16 rounding of the weight operands; "h3": 22 significant bits). */
int uocr_ctx_set_option(uocr_ctx* ctx, const char* key, int value);
int uocr_ctx_reserve_workspace(uocr_ctx* ctx, size_t bytes);   /* synchronises; not capturable */
const char* uocr_last_error(uocr_ctx* ctx);
int uocr_malloc(uocr_ctx* ctx, size_t bytes, void** out);                       /* cupy.zeros/asarray */
int uocr_free(uocr_ctx* ctx, void* ptr);
int uocr_memset_zero(uocr_ctx* ctx, void* ptr, size_t bytes);                   /* Param.clear_grad, layers.py:20-21 */
int uocr_h2d(uocr_ctx* ctx, void* dst, const void* src_host, size_t bytes);     /* CP.copy, gpu.py:19-23 */
int uocr_d2h_sync(uocr_ctx* ctx, void* dst_host, const void* src, size_t bytes);/* CP.asnumpy, gpu.py:25-29 */
int uocr_d2d(uocr_ctx* ctx, void* dst, const void* src, size_t bytes);
int uocr_stream_sync(uocr_ctx* ctx);                                            /* cuda.synchronize() */
int uocr_event_create(void** out_event);
int uocr_event_destroy(void* event);
int uocr_event_record(uocr_ctx* ctx, void* event);
/* the ctx's stream waits (on the device) for `event`, recorded on any other ctx's stream: orders lanes */
int uocr_stream_wait_event(uocr_ctx* ctx, void* event);
int uocr_event_elapsed_ms_sync(void* start, void* stop, float* out_ms);
int uocr_event_synchronize(void* event);                                        /* host waits for the event */
/* HIP-graph capture and replay of a call sequence on the ctx's stream.  The reference pays a Python dispatch and
 * a cuda.synchronize() per layer (layers/layers.py:179-197, convolutional.py:192,278); a train step recorded once
 * between begin_capture and end_capture (every entry point of this header that is not marked "not capturable" or
 * "_sync" may be called in between; no allocation happens inside the library) is replayed with one uocr_graph_launch.
 * Buffers named in the captured calls must stay allocated for the life of the graph; by-value arguments are frozen
 * (optimizer hyper-parameters that change go through the `hyper` device arrays).  Capture mode: relaxed (the caller's
 * allocator may call hipMalloc while capturing).  Other streams join a capture the HIP way: an event recorded on the
 * capturing stream and waited for with uocr_stream_wait_event. */
int uocr_graph_begin_capture(uocr_ctx* ctx);
int uocr_graph_end_capture(uocr_ctx* ctx, void** out_graph_exec);
int uocr_graph_launch(uocr_ctx* ctx, void* graph_exec);
int uocr_graph_destroy(void* graph_exec);
/* Per-step snapshots of a net's losses without a launch of their own.  The reference reads every loss on the host right
 * after the kernel that made it (losses.py:25, models.py:246-254); here losses stay in device slots that the next step
 * overwrites.  With a snapshot configured, the fused optimizer entry points (uocr_momentum_step_fused /
 * uocr_adam_step_fused: the last kernel of a train step) end by copying slots[0..count) into row (n % ring_len) of
 * ring[ring_len][count], n = the number of such calls since `counter` (one device word, zeroed by the caller) was set
 * up -- a device-side count, so a replayed HIP graph advances it by itself.  slots == NULL switches it off. */
int uocr_ctx_set_loss_snapshot(uocr_ctx* ctx, const double* slots, int count, double* ring, int ring_len,
                               unsigned* counter);
/* Deferred weight gradients.  The reference computes dW of a layer right where it computes dX (convolutional.py:101-145,
 * layers.py:341-347); nothing reads dW before the backward pass ends (models.py:226-254).  Between begin and flush the
 * weight-gradient GEMMs of uocr_conv2d_bwd_weight / uocr_dense_bwd(_act) that are small enough to leave the chip
 * mostly idle on their own (the Char net's, my_model/model.py:250-304) are only recorded -- the call returns at once --
 * and flush runs all of them as ONE grid and one reduction; every other call is unaffected.  Operands named in
 * recorded calls must stay valid and unchanged until the flush.  keep_open != 0: flush what is recorded and keep
 * recording (a gradient bucket must be complete now).  Same sums as the separate launches up to float32 summation
 * order (the depth splits are chosen for the group). */
int uocr_wgrad_defer_begin(uocr_ctx* ctx);
int uocr_wgrad_defer_flush(uocr_ctx* ctx, int keep_open);
/* name, CU count, HBM bytes of the ctx's device (train.py:70-90 prints the numba equivalents) */
int uocr_device_info(uocr_ctx* ctx, char* name_out, size_t name_cap, int* cu_count, size_t* hbm_bytes);

/* ---- Convolutional2D (layers/convolutional.py) -------------------------------------------- */
/* y[b,oy,ox,:] = sum_{ky,kx,ic} x~[b,oy*sh-ph+ky,ox*sw-pw+kx,ic] * w[ky,kx,ic,:] (+ b if use_bias),
 * x~ = x inside, pad_value outside (convolutional.py:62-99; GPU kernel :153-195).  OH/OW follow
 * convolutional.py:290-301 and are passed explicitly.  `act` fuses a following activation layer
 * (UOCR_ACT_NONE = plain conv). */
int uocr_conv2d_fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* b, void* y,
                    int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw,
                    int ph, int pw, int oh, int ow, double pad_value, int use_bias,
                    int act, double act_alpha);
/* dx (unpadded input shape), overwritten (convolutional.py:101-145 dx part; GPU :203-219,239-250) */
int uocr_conv2d_bwd_data(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx,
                         int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw,
                         int ph, int pw, int oh, int ow,
                         /* optional epilogue dx *= act'(x_act): x_act = this conv's INPUT when that input is
                          * the output of a (fused) LeakyReLU / Sigmoid -- folds that layer's backward
                          * (layers.py:400-402, 412-415) into this store; act = UOCR_ACT_NONE: off */
                         const void* x_act, int act, double act_alpha);
/* dw (+)= x~^T.dy, db (+)= sum dy (only when use_bias: bias_vec = bias*ones, convolutional.py:113,125);
 * the padded border contributes pad_value to dw (:124-128; GPU :221-237).  accumulate!=0 adds
 * into dw/db (`self.w.grad += dw_total`, :137-138), 0 overwrites. */
int uocr_conv2d_bwd_weight(uocr_ctx* ctx, int dtype, const void* x, const void* dy, void* dw, void* db,
                           int n, int h, int wd, int cin, int cout, int kh, int kw, int sh, int sw,
                           int ph, int pw, int oh, int ow, double pad_value, int use_bias,
                           int accumulate);

/* The Monochrome block (my_model/model.py:108-135) as one forward and one backward kernel, float32 / UOCR_F16:
 *   y = act2( conv3x3( LeakyReLU_alpha1( conv3x3(x; w1,b1) ); w2,b2 ) ),  x,y: (n,h,w,1), w1: (3,3,1,16),
 *   w2: (3,3,16,1), both convs stride 1 / padding 1 (convolutional.py:62-99 twice + layers.py:377-418),
 *   conv_2's padding value 0.  The 16-channel activation and its gradient are recomputed in registers /
 *   LDS instead of crossing HBM.  act2 = UOCR_ACT_NONE or UOCR_ACT_SIGMOID.
 * bwd: dy = gradient w.r.t. y (AFTER act2; the kernel multiplies by act2'(y)); dw/db as
 *   uocr_conv2d_bwd_weight (accumulate!=0 adds), dx may be NULL (page-input gradient not wanted).
 * UOCR_ERR_UNSUPPORTED for float64, cmid != 16 or a LeakyReLU slope outside [0, 1] (the kernels take it as
 * max(z, alpha z)): the caller then runs the layers one by one. */
int uocr_conv_pair_fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w1, const void* b1,
                       const void* w2, const void* b2, void* y, int n, int h, int wd, int cmid,
                       double pad_value1, int use_bias1, int use_bias2, double alpha1, int act2);
int uocr_conv_pair_bwd(uocr_ctx* ctx, int dtype, const void* x, const void* y, const void* dy,
                       const void* w1, const void* b1, const void* w2, void* dw1, void* db1, void* dw2,
                       void* db2, void* dx, int n, int h, int wd, int cmid, double pad_value1,
                       int use_bias1, int use_bias2, double alpha1, int act2, int accumulate);

/* Upsample2D(2) + Convolutional2D(5x5, stride 1, padding 2, padding value 0, 4 -> 4 channels) as one op on
 * the LOW-RES tensor x_low (n,hl,wl,4) -> y (n,2hl,2wl,4): the decoder blocks `up_i` of the Line net
 * (my_model/model.py:194-247 = upsample.py:21-39 + convolutional.py:62-145).  Each output parity phase sees
 * a 3x3 block of source pixels with the 5x5 taps summed in groups, so the upsampled tensor is never built
 * (float32; results equal the two-layer path to rounding, not bit for bit).  Arguments as uocr_conv2d_*,
 * with hl, wl the LOW-RES size; bwd_data returns the gradient w.r.t. x_low (= upsample backward of the conv's
 * dx).  UOCR_ERR_UNSUPPORTED for any other shape or dtype: the caller then runs the two layers. */
/* weff (may be NULL): 576 floats owned by the caller.  The float32 4-channel forward writes the per-phase 3 x 3 weights
 * there; handed to bwd_data of the same layer while w is unchanged (the same train step), it saves that call a launch. */
int uocr_upconv2x_fwd(uocr_ctx* ctx, int dtype, const void* x_low, const void* w, const void* b, void* y,
                      int n, int hl, int wl, int cin, int cout, int kh, int kw, int ph, int pw,
                      int use_bias, int act, double act_alpha, void* weff);
int uocr_upconv2x_bwd_data(uocr_ctx* ctx, int dtype, const void* dy, const void* w, void* dx_low,
                           int n, int hl, int wl, int cin, int cout, int kh, int kw, int ph, int pw,
                           const void* x_act, int act, double act_alpha, const void* weff);
int uocr_upconv2x_bwd_weight(uocr_ctx* ctx, int dtype, const void* x_low, const void* dy, void* dw, void* db,
                             int n, int hl, int wl, int cin, int cout, int kh, int kw, int ph, int pw,
                             int use_bias, int accumulate);

/* ---- MaxPool2D (layers/maxpool.py; the NumPy path :24-90 is the semantics) ---------------- */
/* y = window max with zero padding; mask (uint8, shape (n, kh*oh, kw*ow, c), window-major) marks
 * every element equal to the max; windows running past the padded extent (ceil_mode) shrink. */
int uocr_maxpool2d_fwd(uocr_ctx* ctx, int dtype, const void* x, void* y, uint8_t* mask,
                       int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                       int oh, int ow);
/* dx[b,y,x,c] = sum over windows containing (y,x) with mask set of dy/ties (maxpool.py:62-90) */
int uocr_maxpool2d_bwd(uocr_ctx* ctx, int dtype, const void* dy, const uint8_t* mask, void* dx,
                       int n, int h, int wd, int c, int kh, int kw, int sh, int sw, int ph, int pw,
                       int oh, int ow);

/* ---- Upsample2D (layers/upsample.py:21-110) ----------------------------------------------- */
int uocr_upsample2d_fwd(uocr_ctx* ctx, int dtype, const void* x, void* y,
                        int n, int h, int wd, int c, int sy, int sx);
int uocr_upsample2d_bwd(uocr_ctx* ctx, int dtype, const void* dy, void* dx,
                        int n, int h, int wd, int c, int sy, int sx);   /* h,wd = INPUT (dx) size */

/* ---- Relu / LeakyRelu / Sigmoid (layers/layers.py:377-418) -------------------------------- */
int uocr_act_fwd(uocr_ctx* ctx, int dtype, int kind, double alpha, const void* x, void* y, size_t count);
/* x = the layer's stashed INPUT (the reference stashes the mask / X, layers.py:379,396,409) */
int uocr_act_bwd(uocr_ctx* ctx, int dtype, int kind, double alpha, const void* x, const void* dy,
                 void* dx, size_t count);

/* same gradient computed from the layer's OUTPUT y (conv + activation fused: the pre-activation is
 * never stored).  kind = UOCR_ACT_LEAKY (alpha > 0: sign(y) == sign(x)) or UOCR_ACT_SIGMOID (y(1-y)) */
int uocr_act_bwd_from_output(uocr_ctx* ctx, int dtype, int kind, double alpha, const void* y, const void* dy,
                             void* dx, size_t count);

/* ---- FullyConnected (layers/layers.py:307-363) -------------------------------------------- */
int uocr_dense_fwd(uocr_ctx* ctx, int dtype, const void* x, const void* w, void* y,
                   int m, int n_in, int n_out);
/* dx = dy . w[:-1]^T ; dw (+)= [x,1]^T . dy ; dx or dw may be NULL to skip it (the two halves may then run on
 * different ctx: the weight gradient is off the critical path of a backward pass) */
int uocr_dense_bwd(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* dy,
                   void* dx, void* dw, int m, int n_in, int n_out, int accumulate);
/* FullyConnected followed by LeakyRelu / Sigmoid (my_model/model.py:250-304: dense_1, dense_2) as one
 * GEMM: y = act([x,1] . w) (layers.py:335-339 + :377-418); and the backward of a layer whose input x IS such an
 * activation's output: dx = (dy . w[:-1]^T) * act'(x), act' taken from the output as in uocr_act_bwd_from_output */
int uocr_dense_fwd_act(uocr_ctx* ctx, int dtype, const void* x, const void* w, void* y,
                       int m, int n_in, int n_out, int act, double act_alpha);
int uocr_dense_bwd_act(uocr_ctx* ctx, int dtype, const void* x, const void* w, const void* dy,
                       void* dx, void* dw, int m, int n_in, int n_out, int accumulate,
                       int x_act, double x_act_alpha);

/* ---- Conv2DToBatchedFixedWidthed (layers/convolutional.py:330-373) ------------------------ */
int uocr_fixed_width_fwd(uocr_ctx* ctx, int dtype, const void* x, void* y,
                         int n, int h, int wd, int c, int width);
int uocr_fixed_width_bwd(uocr_ctx* ctx, int dtype, const void* dy, void* dx,
                         int n, int h, int wd, int c, int width);

/* ---- Concat / fan-out sums / fills (layers.py:240-284; models.py:218) --------------------- */
/* dst[r*dst_ld + c] = src[r*src_ld + c], r<rows, c<cols (element strides): one call per Concat input */
int uocr_copy_2d(uocr_ctx* ctx, int dtype, void* dst, size_t dst_ld, const void* src, size_t src_ld,
                 size_t rows, size_t cols);
int uocr_add(uocr_ctx* ctx, int dtype, const void* a, const void* b, void* out, size_t count);
int uocr_axpy(uocr_ctx* ctx, int dtype, double alpha, const void* x, void* y, size_t count); /* y += alpha*x */
int uocr_scale(uocr_ctx* ctx, int dtype, double alpha, void* x, size_t count);
int uocr_fill(uocr_ctx* ctx, int dtype, void* x, double value, size_t count);
int uocr_convert(uocr_ctx* ctx, int src_dtype, const void* src, int dst_dtype, void* dst, size_t count);
/* dst = u8 * scale (page images arrive as uint8, datasets.py:16-19 divides by 255) */
int uocr_u8_to_float(uocr_ctx* ctx, int dtype, const uint8_t* src, void* dst, double scale, size_t count);

/* ---- losses (losses.py) : grad tensor + float64 loss scalar at *loss_out (device) --------- */
/* out_act = UOCR_ACT_SIGMOID: pred is the output of a Sigmoid layer (layers.py:407-418) and grad is taken
 * w.r.t. that layer's INPUT (its backward folded in); UOCR_ACT_NONE: grad w.r.t. pred (losses.py:24,41) */
int uocr_seg_loss(uocr_ctx* ctx, int dtype, int kind /*UOCR_LOSS_DICE|JACCARD*/, const void* pred,
                  const void* gt, void* grad /*may be NULL*/, double* loss_out,
                  int n, int hw, int c, int out_act);                                   /* losses.py:9-42 */
int uocr_softmax_ce(uocr_ctx* ctx, int dtype, const void* pred, const void* gt, void* grad /*may be NULL*/,
                    double* loss_out, int m, int c);                       /* losses.py:60-73 */
int uocr_sigmoid_ce(uocr_ctx* ctx, int dtype, const void* pred, const void* gt, void* grad /*may be NULL*/,
                    double* loss_out, int m, size_t count);                /* losses.py:45-57 */

/* ---- regularizers (regularizations.py:15-26, applied at layers.py:147-155) ----------------- */
/* grad += d/dw, *loss_out (+)= strength*sum(...)  (accumulate_loss=0 overwrites the slot) */
int uocr_l2_reg(uocr_ctx* ctx, int dtype, const void* w, void* grad, size_t count, double strength,
                double* loss_out, int accumulate_loss);
int uocr_l1_reg(uocr_ctx* ctx, int dtype, const void* w, void* grad, size_t count, double strength,
                double* loss_out, int accumulate_loss);

/* ---- optimizers (optimizers.py:47-98), fused in-place updates ------------------------------ */
/* v=b1*v+(1-b1)g; a=b2*a+(1-b2)g^2; w-=lr/(sqrt(a)+eps)*v   -- NO bias correction (:56-61) */
int uocr_adam_step(uocr_ctx* ctx, int dtype, void* w, const void* g, void* v, void* a, size_t count,
                   double lr, double beta1, double beta2, double eps);
/* v=mu*v-lr*g; w+=v  (:75-78; mu=0 is plain SGD) */
int uocr_momentum_step(uocr_ctx* ctx, int dtype, void* w, const void* g, void* v, size_t count,
                       double lr, double momentum);
/* a=rho*a+(1-rho)g^2; w-=lr/(sqrt(a)+eps)*g  (:92-95) */
/* The tail of a train step in one pass over the flat parameter buffer: for up to 4 index ranges [lo, hi)
 * grad += dR/dw (kind 1 = L1, 2 = L2: regularizations.py:15-26 as applied in layers.py:147-155), then the
 * Momentum update (optimizers.py:75-78), then grad = 0 when zero_grad (Param.clear_grad, layers.py:20-21);
 * *reg_loss_out = sum_r strength_r * R_r (float64 device slot).  lo / hi / kind / strength are HOST arrays. */
int uocr_momentum_step_fused(uocr_ctx* ctx, int dtype, void* w, void* g, void* v, size_t count, double lr,
                             double momentum, int nranges, const long long* lo, const long long* hi,
                             const int* kind, const double* strength, double* reg_loss_out, int zero_grad,
                             /* optional DEVICE array {lr, momentum, -, -}: when given the kernel reads the
                              * hyper-parameters from it instead of the by-value arguments, so a call captured in
                              * a HIP graph follows the trainer's learning-rate decay (my_model/trainer.py:260);
                              * NULL = use the arguments */
                             const double* hyper_dev);
/* the same with the Adam update of optimizers.py:56-61 (no bias correction, as the reference) */
int uocr_adam_step_fused(uocr_ctx* ctx, int dtype, void* w, void* g, void* v, void* a, size_t count, double lr,
                         double beta1, double beta2, double eps, int nranges, const long long* lo,
                         const long long* hi, const int* kind, const double* strength, double* reg_loss_out,
                         int zero_grad, const double* hyper_dev /* {lr, beta1, beta2, eps} on the device, or NULL */);
int uocr_rmsprop_step(uocr_ctx* ctx, int dtype, void* w, const void* g, void* a, size_t count,
                      double lr, double rho, double eps);
/* *flag_out (int32, device) = 1 if any element is NaN else 0  (nan_weights, layers.py:139-140) */
int uocr_has_nan(uocr_ctx* ctx, int dtype, const void* x, size_t count, int32_t* flag_out);

/* ---- data parallel over the GPUs of one node: RCCL over xGMI ------------------------------------
 * The reference has no multi-GPU path; BASELINE.json adds one to the step loop my_model/trainer.py:213-233
 * -> nn/model_system.py:104-118 -> nn/models.py:250-254: between compute_loss_and_gradients and update_grads
 * every rank's flat gradient buffer is summed.  One process per GPU, one communicator per process and device.
 * librccl.so.1 is bound at run time (no link-time dependency; single-GPU use needs no RCCL).
 *   rank 0:   uocr_dp_get_unique_id(id)  -> ship the 128 bytes to the other ranks by ANY channel (file, socket,
 *             MPI, torch.distributed, the launcher's environment)
 *   all:      uocr_dp_init(ctx, rank, world, id)                 (collective; blocks until every rank called)
 *             uocr_dp_broadcast(ctx, weights, n, UOCR_F32, 0)    identical replicas
 *   per step: uocr_dp_allreduce_sum(ctx, grads, n, UOCR_F32)     in place, asynchronous on THAT ctx's stream
 *   all:      uocr_dp_finalize(ctx)
 * Collectives of the communicator must be issued in the same order on every rank and must not overlap each
 * other: issue them through one ctx (a communication lane), or chain the issuing streams with
 * uocr_event_record / uocr_stream_wait_event. */
#define UOCR_DP_UNIQUE_ID_BYTES 128
int uocr_dp_get_unique_id(void* out_bytes /* UOCR_DP_UNIQUE_ID_BYTES */);
int uocr_dp_init(uocr_ctx* ctx, int rank, int world, const void* unique_id_bytes);
int uocr_dp_info(uocr_ctx* ctx, int* rank, int* world /* 0 = no communicator */);
int uocr_dp_allreduce_sum(uocr_ctx* ctx, void* buf, size_t count, int dtype);
int uocr_dp_broadcast(uocr_ctx* ctx, void* buf, size_t count, int dtype, int root);
int uocr_dp_finalize(uocr_ctx* ctx);
int uocr_dp_version(int* out_version);   /* ncclGetVersion of the RCCL this process bound (e.g. 22703), for run records */

#ifdef __cplusplus
}
#endif
#endif /* UNIVER_HIP_H */
