// Data-parallel entry points of the C ABI: RCCL all-reduce / broadcast of flat parameter and gradient buffers
// over xGMI, one process per GPU.  The reference has no multi-GPU path (no collective call sites anywhere);
// these are the MI355X-native addition BASELINE.json names, at the place of the reference's step loop where a
// gradient exists and the optimizer has not run yet (nn/models.py:250-254: between compute_loss_and_gradients
// and update_grads).
//
// RCCL is bound at RUN time (dlopen of "librccl.so.1"): a single-GPU user of the library needs no RCCL, and a
// process that has already loaded an RCCL (PyTorch-ROCm ships one) keeps exactly that one -- two copies with
// two HIP runtimes in one process is what goes wrong otherwise.  One communicator per process and device; the
// collective is enqueued on the stream of the ctx it is called through, so the caller decides where it runs
// (a communication lane of its own, or the lane of the net whose gradients it reduces) and orders it with
// uocr_event_record / uocr_stream_wait_event.  Collectives of the ONE communicator must be issued in the same
// order on every rank and never run concurrently: issue them from one stream, or chain them with events.
#include <dlfcn.h>

#include <mutex>

#include "uocr_common.h"

namespace {

// the slice of rccl.h this file uses (types restated so that the build needs no RCCL headers either)
using ncclComm_t = void*;
struct ncclUniqueId {
    char internal[UOCR_DP_UNIQUE_ID_BYTES];
};
enum { ncclSuccess = 0 };
enum { ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 };   // ncclDataType_t
enum { ncclSum = 0 };                                          // ncclRedOp_t

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    char why[256] = {0};
};

Rccl g_rccl;
std::mutex g_mu;

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};
constexpr int MAX_DEVICES = 64;
Comm g_comm[MAX_DEVICES];

bool load_rccl() {
    if (g_rccl.handle) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) {
        g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 not found: %s", dlerror());
        return false;
    }
    auto sym = [&](const char* s) { return dlsym(g_rccl.handle, s); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))sym("ncclGetVersion");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce ||
        !g_rccl.Broadcast) {
        snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so.1 lacks a required symbol");
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
        return false;
    }
    return true;
}

const char* rccl_error(int code) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "?"; }

int nccl_type(int dtype) {
    switch (dtype & 0xff) {
        case UOCR_F32: return ncclFloat32;
        case UOCR_F64: return ncclFloat64;
        case UOCR_F16: return ncclFloat16;
        default: return -1;
    }
}

#define UOCR_RCCL(ctx, call)                                                                            \
    do {                                                                                                \
        int r_ = (call);                                                                                \
        if (r_ != ncclSuccess)                                                                          \
            UOCR_FAIL(ctx, UOCR_ERR_RCCL, "%s failed: %s (%s:%d)", #call, rccl_error(r_), __FILE__, __LINE__); \
    } while (0)

}  // namespace

extern "C" {

int uocr_dp_get_unique_id(void* out_bytes) {
    if (!out_bytes) return UOCR_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!load_rccl()) return UOCR_ERR_RCCL;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return UOCR_ERR_RCCL;
    memcpy(out_bytes, id.internal, UOCR_DP_UNIQUE_ID_BYTES);
    return UOCR_OK;
}

int uocr_dp_init(uocr_ctx* ctx, int rank, int world, const void* unique_id_bytes) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, unique_id_bytes && world >= 1 && rank >= 0 && rank < world);
    UOCR_REQUIRE(ctx, ctx->device < MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_mu);
    if (!load_rccl()) UOCR_FAIL(ctx, UOCR_ERR_RCCL, "%s", g_rccl.why);
    Comm& c = g_comm[ctx->device];
    if (c.comm) UOCR_FAIL(ctx, UOCR_ERR_ARG, "uocr_dp_init: device %d already has a communicator", ctx->device);
    UOCR_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, unique_id_bytes, UOCR_DP_UNIQUE_ID_BYTES);
    UOCR_RCCL(ctx, g_rccl.CommInitRank(&c.comm, world, id, rank));
    c.rank = rank;
    c.world = world;
    return UOCR_OK;
}

int uocr_dp_info(uocr_ctx* ctx, int* rank, int* world) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, ctx->device < MAX_DEVICES);
    const Comm& c = g_comm[ctx->device];
    if (rank) *rank = c.comm ? c.rank : 0;
    if (world) *world = c.comm ? c.world : 0;     // 0 = no communicator
    return UOCR_OK;
}

int uocr_dp_allreduce_sum(uocr_ctx* ctx, void* buf, size_t count, int dtype) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, ctx->device < MAX_DEVICES);
    const Comm& c = g_comm[ctx->device];
    if (!c.comm) UOCR_FAIL(ctx, UOCR_ERR_RCCL, "uocr_dp_allreduce_sum before uocr_dp_init");
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, buf != nullptr);
    const int t = nccl_type(dtype);
    if (t < 0) UOCR_FAIL(ctx, UOCR_ERR_DTYPE, "unknown dtype %d", dtype);
    UOCR_RCCL(ctx, g_rccl.AllReduce(buf, buf, count, t, ncclSum, c.comm, ctx->stream));
    return UOCR_OK;
}

int uocr_dp_broadcast(uocr_ctx* ctx, void* buf, size_t count, int dtype, int root) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, ctx->device < MAX_DEVICES);
    const Comm& c = g_comm[ctx->device];
    if (!c.comm) UOCR_FAIL(ctx, UOCR_ERR_RCCL, "uocr_dp_broadcast before uocr_dp_init");
    if (!count) return UOCR_OK;
    UOCR_REQUIRE(ctx, buf != nullptr && root >= 0 && root < c.world);
    const int t = nccl_type(dtype);
    if (t < 0) UOCR_FAIL(ctx, UOCR_ERR_DTYPE, "unknown dtype %d", dtype);
    UOCR_RCCL(ctx, g_rccl.Broadcast(buf, buf, count, t, root, c.comm, ctx->stream));
    return UOCR_OK;
}

int uocr_dp_finalize(uocr_ctx* ctx) {
    UOCR_CHECK_CTX(ctx);
    UOCR_REQUIRE(ctx, ctx->device < MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_mu);
    Comm& c = g_comm[ctx->device];
    if (!c.comm) return UOCR_OK;
    UOCR_HIP(ctx, hipSetDevice(ctx->device));
    UOCR_HIP(ctx, hipDeviceSynchronize());
    const int r = g_rccl.CommDestroy(c.comm);
    c = Comm{};
    if (r != ncclSuccess) UOCR_FAIL(ctx, UOCR_ERR_RCCL, "ncclCommDestroy failed: %s", rccl_error(r));
    return UOCR_OK;
}

int uocr_dp_version(int* out_version) {
    if (!out_version) return UOCR_ERR_ARG;
    *out_version = 0;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!load_rccl()) return UOCR_ERR_RCCL;
    if (!g_rccl.GetVersion || g_rccl.GetVersion(out_version) != ncclSuccess) return UOCR_ERR_RCCL;
    return UOCR_OK;
}

}  // extern "C"
