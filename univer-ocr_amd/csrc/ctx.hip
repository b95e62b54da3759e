// Context, memory and event entry points of the C ABI (include/univer_hip.h).
// These stand in for the reference's `CP` backend switch (nn/gpu.py:5-29): CP.copy = H2D,
// CP.asnumpy = D2H, cupy.zeros = malloc + memset, cuda.synchronize() = stream sync.
#include <cstdlib>

#include "finish_group.h"
#include "gemm.h"
#include "uocr_common.h"

extern "C" {

int uocr_abi_version(void) { return UOCR_ABI_VERSION; }

int uocr_ctx_create(int device, size_t workspace_bytes, uocr_ctx** out) {
    if (!out || device < 0) return UOCR_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device >= count) return UOCR_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return UOCR_ERR_HIP;
    uocr_ctx* ctx = new uocr_ctx();
    ctx->device = device;
    ctx->err[0] = 0;
    ctx->workspace = nullptr;
    ctx->workspace_bytes = 0;
    ctx->sync = nullptr;
    ctx->gemm_defer = nullptr;
    ctx->snap_src = nullptr;
    ctx->snap_count = 0;
    ctx->snap_ring = nullptr;
    ctx->snap_ring_len = 0;
    ctx->snap_counter = nullptr;
    ctx->finish_defer = nullptr;
    ctx->owns_stream = true;
    ctx->opt_mfma = 1;
    ctx->opt_fast = 1;
    ctx->opt_tiled = 1;
    ctx->opt_split = 1024;
    ctx->opt_split_min = 3;
    ctx->opt_bm = 0;
    ctx->opt_h16 = 1;
    ctx->opt_t32 = 2;    // measured: only the 4-channel backward-data beats its vector kernel in the step (conv_t32.hip,
                         // conv_t32w.hip: bits 64 / 128 = the float32-MFMA weight gradients, 37.0 -> 36.1 k images/s with both)
    ctx->opt_xcd = 1;
    ctx->opt_pair_band = 0;
    ctx->opt_pair_g = 4;
    ctx->opt_group_blocks = 0;
    ctx->opt_wgrad_bands = 0;
    ctx->opt_h3 = 0;
#ifdef UOCR_EXPERIMENTS
    if (const char* e = getenv("UOCR_H3")) ctx->opt_h3 = atoi(e);            // development override (tools/dev/h3_ab.sh)
#endif
    ctx->opt_pair_pf = -1;   // auto: float32 one step ahead (49.5 us; 50.8 pinned in the loop, 51.8 three ahead), binary16 three (72.1; 73.8 / 75.5)
    if (const char* e = getenv("UOCR_PAIR_PF")) ctx->opt_pair_pf = atoi(e);   // development override (tools/dev/pf_ab.sh)
    if (const char* e = getenv("UOCR_WGRAD_BANDS")) ctx->opt_wgrad_bands = atoi(e);   // development override (tools/dev/bands_ab.sh)
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return UOCR_ERR_HIP;
    }
    hipDeviceProp_t prop;
    ctx->cu_count = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
    if (workspace_bytes) {
        if (hipMalloc(&ctx->workspace, workspace_bytes) != hipSuccess) {
            hipStreamDestroy(ctx->stream);
            delete ctx;
            return UOCR_ERR_HIP;
        }
        ctx->workspace_bytes = workspace_bytes;
    }
    // arrival counters: zeroed once here; every kernel that uses them leaves them at zero
    if (hipMalloc((void**)&ctx->sync, UOCR_SYNC_WORDS * sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(ctx->sync, 0, UOCR_SYNC_WORDS * sizeof(unsigned), ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        if (ctx->sync) hipFree(ctx->sync);
        if (ctx->workspace) hipFree(ctx->workspace);
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return UOCR_ERR_HIP;
    }
    *out = ctx;
    return UOCR_OK;
}

int uocr_ctx_create_cu_mask(int device, size_t workspace_bytes, const uint32_t* cu_mask, int mask_words, uocr_ctx** out) {
    if (!out || !cu_mask || mask_words <= 0) return UOCR_ERR_ARG;
    int rc = uocr_ctx_create(device, workspace_bytes, out);
    if (rc != UOCR_OK) return rc;
    uocr_ctx* ctx = *out;
    hipStream_t masked = nullptr;
    if (hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask_words, cu_mask) != hipSuccess) {
        uocr_ctx_destroy(ctx);
        *out = nullptr;
        return UOCR_ERR_HIP;
    }
    hipStreamDestroy(ctx->stream);
    ctx->stream = masked;
    int cus = 0;
    for (int i = 0; i < mask_words; ++i) cus += __builtin_popcount(cu_mask[i]);
    if (cus > 0 && cus < ctx->cu_count) ctx->cu_count = cus;     // grids of persistent kernels follow the partition
    return UOCR_OK;
}

int uocr_ctx_destroy(uocr_ctx* ctx) {
    UOCR_CHECK_CTX(ctx);
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->workspace) hipFree(ctx->workspace);
    if (ctx->sync) hipFree(ctx->sync);
    uocr_gemm_defer_free(ctx);
    uocr_finish_defer_free(ctx);
    if (ctx->owns_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return UOCR_OK;
}

int uocr_ctx_set_stream(uocr_ctx* ctx, void* hip_stream) {
    UOCR_CHECK_CTX(ctx);
    if (ctx->owns_stream) {
        UOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        UOCR_HIP(ctx, hipStreamDestroy(ctx->stream));
        ctx->owns_stream = false;
    }
    ctx->stream = (hipStream_t)hip_stream;
    return UOCR_OK;
}

int uocr_ctx_set_option(uocr_ctx* ctx, const char* key, int value) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, key != nullptr);
    if (!strcmp(key, "mfma")) ctx->opt_mfma = value;
    else if (!strcmp(key, "fast_paths")) ctx->opt_fast = value;
    else if (!strcmp(key, "tiled")) ctx->opt_tiled = value;
    else if (!strcmp(key, "split_blocks")) ctx->opt_split = value;
    else if (!strcmp(key, "split_min")) ctx->opt_split_min = value;
    else if (!strcmp(key, "gemm_bm")) ctx->opt_bm = value;
    else if (!strcmp(key, "xcd_remap")) ctx->opt_xcd = value;
    else if (!strcmp(key, "h16")) ctx->opt_h16 = value;
    else if (!strcmp(key, "t32")) {
        if (value & ~UOCR_T32_BUILT)
            UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "option t32 = %d asks for kernels this library was built without "
                      "(UOCR_BUILD_EXPERIMENTS=1 ./build.sh)", value);
        ctx->opt_t32 = value;
    }
    else if (!strcmp(key, "pair_band")) ctx->opt_pair_band = value;
    else if (!strcmp(key, "pair_g")) ctx->opt_pair_g = value;
    else if (!strcmp(key, "group_blocks")) ctx->opt_group_blocks = value;
    else if (!strcmp(key, "wgrad_bands") && value >= 0) ctx->opt_wgrad_bands = value;
    else if (!strcmp(key, "h3")) {
#ifndef UOCR_EXPERIMENTS
        if (value) UOCR_FAIL(ctx, UOCR_ERR_UNSUPPORTED, "option h3: this library was built without conv_h3 "
                             "(UOCR_BUILD_EXPERIMENTS=1 ./build.sh)");
#endif
        ctx->opt_h3 = value;
    }
    else if (!strcmp(key, "pair_pf")) ctx->opt_pair_pf = value;
    else UOCR_FAIL(ctx, UOCR_ERR_ARG, "unknown option '%s'", key);
    return UOCR_OK;
}

void* uocr_ctx_get_stream(uocr_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int uocr_ctx_reserve_workspace(uocr_ctx* ctx, size_t bytes) {
    UOCR_CHECK_CTX(ctx);
    if (bytes <= ctx->workspace_bytes) return UOCR_OK;
    UOCR_HIP(ctx, hipSetDevice(ctx->device));
    UOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->workspace) UOCR_HIP(ctx, hipFree(ctx->workspace));
    ctx->workspace = nullptr;
    ctx->workspace_bytes = 0;
    UOCR_HIP(ctx, hipMalloc(&ctx->workspace, bytes));
    ctx->workspace_bytes = bytes;
    return UOCR_OK;
}

const char* uocr_last_error(uocr_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int uocr_malloc(uocr_ctx* ctx, size_t bytes, void** out) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, out != nullptr);
    UOCR_HIP(ctx, hipSetDevice(ctx->device));
    UOCR_HIP(ctx, hipMalloc(out, bytes ? bytes : 1));
    return UOCR_OK;
}

int uocr_free(uocr_ctx* ctx, void* ptr) {
    UOCR_CHECK_CTX(ctx);
    if (ptr) UOCR_HIP(ctx, hipFree(ptr));
    return UOCR_OK;
}

int uocr_memset_zero(uocr_ctx* ctx, void* ptr, size_t bytes) {
    UOCR_CHECK_CTX(ctx);
    if (!bytes) return UOCR_OK;
    UOCR_REQUIRE(ctx, ptr != nullptr);
    UOCR_HIP(ctx, hipMemsetAsync(ptr, 0, bytes, ctx->stream));
    return UOCR_OK;
}

int uocr_h2d(uocr_ctx* ctx, void* dst, const void* src_host, size_t bytes) {
    UOCR_CHECK_CTX(ctx);
    if (!bytes) return UOCR_OK;
    UOCR_REQUIRE(ctx, dst && src_host);
    UOCR_HIP(ctx, hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return UOCR_OK;
}

int uocr_d2h_sync(uocr_ctx* ctx, void* dst_host, const void* src, size_t bytes) {
    UOCR_CHECK_CTX(ctx);
    if (!bytes) return UOCR_OK;
    UOCR_REQUIRE(ctx, dst_host && src);
    UOCR_HIP(ctx, hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    UOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UOCR_OK;
}

int uocr_d2d(uocr_ctx* ctx, void* dst, const void* src, size_t bytes) {
    UOCR_CHECK_CTX(ctx);
    if (!bytes) return UOCR_OK;
    UOCR_REQUIRE(ctx, dst && src);
    UOCR_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return UOCR_OK;
}

int uocr_stream_sync(uocr_ctx* ctx) {
    UOCR_CHECK_CTX(ctx);
    UOCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return UOCR_OK;
}

int uocr_event_create(void** out_event) {
    if (!out_event) return UOCR_ERR_ARG;
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return UOCR_ERR_HIP;
    *out_event = (void*)ev;
    return UOCR_OK;
}

int uocr_event_destroy(void* event) {
    if (!event) return UOCR_ERR_ARG;
    return hipEventDestroy((hipEvent_t)event) == hipSuccess ? UOCR_OK : UOCR_ERR_HIP;
}

int uocr_event_record(uocr_ctx* ctx, void* event) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, event != nullptr);
    UOCR_HIP(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
    return UOCR_OK;
}

int uocr_stream_wait_event(uocr_ctx* ctx, void* event) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, event != nullptr);
    UOCR_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
    return UOCR_OK;
}

int uocr_event_elapsed_ms_sync(void* start, void* stop, float* out_ms) {
    if (!start || !stop || !out_ms) return UOCR_ERR_ARG;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return UOCR_ERR_HIP;
    return hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? UOCR_OK
                                                                                           : UOCR_ERR_HIP;
}

int uocr_event_synchronize(void* event) {
    if (!event) return UOCR_ERR_ARG;
    return hipEventSynchronize((hipEvent_t)event) == hipSuccess ? UOCR_OK : UOCR_ERR_HIP;
}

int uocr_graph_begin_capture(uocr_ctx* ctx) {
    UOCR_CHECK_CTX(ctx);
    UOCR_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    return UOCR_OK;
}

int uocr_graph_end_capture(uocr_ctx* ctx, void** out_graph_exec) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, out_graph_exec != nullptr);
    *out_graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    UOCR_HIP(ctx, hipStreamEndCapture(ctx->stream, &graph));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) UOCR_FAIL(ctx, UOCR_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *out_graph_exec = (void*)exec;
    return UOCR_OK;
}

int uocr_graph_launch(uocr_ctx* ctx, void* graph_exec) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, graph_exec != nullptr);
    UOCR_HIP(ctx, hipGraphLaunch((hipGraphExec_t)graph_exec, ctx->stream));
    return UOCR_OK;
}

int uocr_graph_destroy(void* graph_exec) {
    if (!graph_exec) return UOCR_ERR_ARG;
    return hipGraphExecDestroy((hipGraphExec_t)graph_exec) == hipSuccess ? UOCR_OK : UOCR_ERR_HIP;
}

int uocr_device_info(uocr_ctx* ctx, char* name_out, size_t name_cap, int* cu_count, size_t* hbm_bytes) {
    UOCR_CHECK_CTX(ctx);
    hipDeviceProp_t prop;
    UOCR_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name_out && name_cap) {
        strncpy(name_out, prop.name, name_cap - 1);
        name_out[name_cap - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return UOCR_OK;
}

}  // extern "C"
