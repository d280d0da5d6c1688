// gs_rccl.cpp -- RCCL, loaded on first use (single-process users never touch it), the one-rank self-test of the
// ghost-row exchange's call pattern, which libraries the process is bound to, and the error message of the last failure.
#include <atomic>
#include <chrono>
#include "gs_internal.h"

namespace gsi {

thread_local std::string g_last_error;

int32_t fail(int32_t code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static std::atomic<Rccl *> g_rccl_loaded{nullptr}; // what rccl() found, for callers that must not trigger the load

Rccl *rccl_if_loaded() { return g_rccl_loaded.load(); }

Rccl *rccl()
{
    // function-local static: initialised once, thread-safe (C++11)
    static Rccl *const instance = []() -> Rccl * {
        static Rccl r;
        // GS_RCCL_LIBRARY names the library to bind instead of the system's librccl (a custom
        // RCCL build; the tests' shared-memory transport double, tests/cpp/shm_transport.cpp)
        const char *user = std::getenv("GS_RCCL_LIBRARY");
        for (const char *name : {user, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (!name || !*name) continue;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle || name == user) break; // an explicit choice never falls back silently
        }
        if (!r.handle) return nullptr;
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p) ok = false;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
            return nullptr;
        }
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.handle, "ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.handle, "ncclCommUserRank"));
        r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(dlsym(r.handle, "ncclCommCuDevice"));
        g_rccl_loaded.store(&r);
        return &r;
    }();
    return instance;
}

} // namespace gsi

using namespace gsi;

extern "C" {

const char *gs_last_error(void) { return g_last_error.c_str(); }

int32_t gs_get_unique_id(void *out128)
{
    if (!out128) return fail(GS_ERR_INVALID, "null output");
    Rccl *R = rccl();
    if (!R) return fail(GS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    ncclUniqueId id;
    GS_NCCL(R, R->GetUniqueId(&id));
    std::memcpy(out128, &id, sizeof id);
    return GS_OK;
}

int32_t gs_rccl_selftest(int32_t device, uint64_t floats)
{
    Rccl *R = rccl();
    if (!R) return fail(GS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
    if (floats == 0 || floats > (1ull << 28)) return fail(GS_ERR_INVALID, "message of %llu floats", (unsigned long long)floats);
    GS_HIP(hipSetDevice(device));
    ncclUniqueId id;
    GS_NCCL(R, R->GetUniqueId(&id));
    ncclComm_t comm = nullptr;
    GS_NCCL(R, R->CommInitRank(&comm, 1, id, 0));
    float *src = nullptr, *dst = nullptr;
    hipStream_t stream = nullptr;
    std::vector<float> host(floats), back(floats);
    for (uint64_t i = 0; i < floats; ++i) host[i] = (float)(i % 65521) * 0.25f + 1.0f;
    int32_t st = GS_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (st == GS_OK && e != hipSuccess) st = fail(GS_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    int least = 0, greatest = 0;
    step(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    step(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    step(hipMalloc(reinterpret_cast<void **>(&src), floats * sizeof(float)), "hipMalloc");
    step(hipMalloc(reinterpret_cast<void **>(&dst), floats * sizeof(float)), "hipMalloc");
    step(hipMemcpy(src, host.data(), floats * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy");
    step(hipMemset(dst, 0, floats * sizeof(float)), "hipMemset");
    if (st == GS_OK) {
        // the call pattern of push_halo: one group, a send and the matching receive, on the halo stream
        ncclResult_t r = R->GroupStart();
        if (r == ncclSuccess) r = R->Send(src, (size_t)floats, ncclFloat, 0, comm, stream);
        if (r == ncclSuccess) r = R->Recv(dst, (size_t)floats, ncclFloat, 0, comm, stream);
        const ncclResult_t e = R->GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) st = fail(GS_ERR_RCCL, "grouped ncclSend / ncclRecv to self failed: %s", R->GetErrorString(r));
    }
    step(hipStreamSynchronize(stream), "hipStreamSynchronize");
    step(hipMemcpy(back.data(), dst, floats * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy");
    if (st == GS_OK && std::memcmp(back.data(), host.data(), floats * sizeof(float)) != 0)
        st = fail(GS_ERR_RCCL, "the message came back altered");
    if (stream) (void)hipStreamDestroy(stream);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    R->CommDestroy(comm);
    (void)hipGetLastError();
    return st;
}

// The ghost-row exchange's transport UNDER LOAD, as far as one GPU can show it (gs_hip.h: gs_debug_exchange_probe_*): a
// one-rank communicator, a high-priority stream like a slab's halo stream, `messages` send / receive pairs to itself
// in one group per exchange -- or, mode 1, the same bytes as device-to-device copies, the route of in-process chains.
// The caller decides what else the chip is doing (tools/rccl_under_load.py: nothing, or the interior kernel of a
// 2^28-cell slab on the compute stream of a context).
struct gs_exchange_probe {
    int device = 0, mode = 0, messages = 0;
    uint64_t floats = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float *src = nullptr, *dst = nullptr;
};

int32_t gs_debug_exchange_probe_destroy(gs_exchange_probe *p)
{
    if (!p) return GS_OK;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->e0) (void)hipEventDestroy(p->e0);
    if (p->e1) (void)hipEventDestroy(p->e1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->src) (void)hipFree(p->src);
    if (p->dst) (void)hipFree(p->dst);
    if (p->comm) {
        Rccl *R = rccl();
        if (R) R->CommDestroy(p->comm);
    }
    (void)hipGetLastError();
    delete p;
    return GS_OK;
}

int32_t gs_debug_exchange_probe_create(int32_t device, int32_t mode, int32_t messages, uint64_t floats, gs_exchange_probe **out)
{
    if (!out) return fail(GS_ERR_INVALID, "null output");
    *out = nullptr;
    if (mode < 0 || mode > 1 || messages < 1 || messages > 8 || floats == 0 || floats > (1ull << 26))
        return fail(GS_ERR_INVALID, "mode 0 / 1, 1 to 8 messages of at most 2^26 floats");
    gs_exchange_probe *p = new (std::nothrow) gs_exchange_probe();
    if (!p) return fail(GS_ERR_NOMEM, "out of host memory");
    p->device = device; p->mode = mode; p->messages = messages; p->floats = floats;
    int32_t st = GS_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (st == GS_OK && e != hipSuccess) st = fail(GS_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    step(hipSetDevice(device), "hipSetDevice");
    if (st == GS_OK && mode == 0) {
        Rccl *R = rccl();
        if (!R) st = fail(GS_ERR_RCCL, "librccl could not be loaded: %s", dlerror());
        ncclUniqueId id;
        ncclResult_t r = st == GS_OK ? R->GetUniqueId(&id) : ncclSuccess;
        if (st == GS_OK && r == ncclSuccess) r = R->CommInitRank(&p->comm, 1, id, 0);
        if (st == GS_OK && r != ncclSuccess) st = fail(GS_ERR_RCCL, "one-rank communicator: %s", R->GetErrorString(r));
    }
    int least = 0, greatest = 0;
    step(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    step(hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    step(hipEventCreate(&p->e0), "hipEventCreate");
    step(hipEventCreate(&p->e1), "hipEventCreate");
    const size_t bytes = (size_t)messages * floats * sizeof(float);
    step(hipMalloc(reinterpret_cast<void **>(&p->src), bytes), "hipMalloc");
    step(hipMalloc(reinterpret_cast<void **>(&p->dst), bytes), "hipMalloc");
    step(hipMemset(p->src, 0x3c, bytes), "hipMemset");
    step(hipMemset(p->dst, 0, bytes), "hipMemset");
    step(hipDeviceSynchronize(), "hipDeviceSynchronize");
    if (st != GS_OK) { gs_debug_exchange_probe_destroy(p); return st; }
    *out = p;
    return GS_OK;
}

// One exchange: enqueued now, waited for.  host_ms: from the first enqueue call to the end of the wait (what a pass's
// boundary has to hide); device_ms: between two events around the exchange on its stream (start of its first kernel or
// copy to the end of the last); the difference is how long the exchange waited for a place on the chip.
int32_t gs_debug_exchange_probe_run(gs_exchange_probe *p, float *host_ms, float *device_ms)
{
    if (!p) return fail(GS_ERR_INVALID, "null probe");
    GS_HIP(hipSetDevice(p->device));
    const auto t0 = std::chrono::steady_clock::now();
    GS_HIP(hipEventRecord(p->e0, p->stream));
    if (p->mode == 0) {
        Rccl *R = rccl();
        ncclResult_t r = R->GroupStart();
        for (int m = 0; m < p->messages && r == ncclSuccess; ++m) {
            r = R->Send(p->src + (size_t)m * p->floats, (size_t)p->floats, ncclFloat, 0, p->comm, p->stream);
            if (r == ncclSuccess) r = R->Recv(p->dst + (size_t)m * p->floats, (size_t)p->floats, ncclFloat, 0, p->comm, p->stream);
        }
        const ncclResult_t e = R->GroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) return fail(GS_ERR_RCCL, "grouped ncclSend / ncclRecv to self failed: %s", R->GetErrorString(r));
    } else {
        for (int m = 0; m < p->messages; ++m)
            GS_HIP(hipMemcpyAsync(p->dst + (size_t)m * p->floats, p->src + (size_t)m * p->floats, (size_t)p->floats * sizeof(float),
                                  hipMemcpyDeviceToDevice, p->stream));
    }
    GS_HIP(hipEventRecord(p->e1, p->stream));
    GS_HIP(hipEventSynchronize(p->e1));
    const auto t1 = std::chrono::steady_clock::now();
    if (host_ms) *host_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
    if (device_ms) GS_HIP(hipEventElapsedTime(device_ms, p->e0, p->e1));
    return GS_OK;
}

// Which HIP runtime and which RCCL this process's libgs_hip.so is bound to (dladdr of an entry point of each), with
// their versions.  A process that imported torch first resolves libamdhip64.so.7 and librccl.so.1 by SONAME to the
// copies torch bundles -- the runtime that owns the device pointers the planes live at is then the one RCCL moves them
// with; a torch-free process gets /opt/rocm's.
int32_t gs_runtime_info(int32_t load_rccl, char *out, size_t cap)
{
    if (!out || cap == 0) return fail(GS_ERR_INVALID, "null output");
    Dl_info hip_so{}, rccl_so{};
    int hip_version = 0, rccl_version = 0;
    (void)dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &hip_so);
    if (hipRuntimeGetVersion(&hip_version) != hipSuccess) { hip_version = 0; (void)hipGetLastError(); }
    Rccl *R = load_rccl ? rccl() : rccl_if_loaded(); // load_rccl = 0: reported if this process has loaded it already
    if (R) {
        (void)dladdr(reinterpret_cast<const void *>(R->Send), &rccl_so);
        auto get_version = reinterpret_cast<ncclResult_t (*)(int *)>(dlsym(R->handle, "ncclGetVersion"));
        if (get_version) (void)get_version(&rccl_version);
    }
    // paths go into JSON strings: quotes and backslashes escaped, control characters dropped
    auto json_string = [](const char *p) {
        std::string o;
        for (; p && *p; ++p) {
            if (*p == '"' || *p == '\\') o += '\\';
            if ((unsigned char)*p >= 0x20) o += *p;
        }
        return o;
    };
    const char *user = std::getenv("GS_RCCL_LIBRARY");
    const std::string hip_path = json_string(hip_so.dli_fname), rccl_path = json_string(rccl_so.dli_fname);
    const int n = std::snprintf(out, cap, "{\"hip\": \"%s\", \"hip_runtime_version\": %d, \"rccl\": %s%s%s, \"rccl_version\": %d, "
                                          "\"rccl_named_by_GS_RCCL_LIBRARY\": %s}",
                                hip_path.c_str(), hip_version, rccl_so.dli_fname ? "\"" : "", rccl_so.dli_fname ? rccl_path.c_str() : "null",
                                rccl_so.dli_fname ? "\"" : "", rccl_version, user && *user ? "true" : "false");
    if (n < 0 || (size_t)n >= cap) {
        out[0] = '\0';
        return fail(GS_ERR_INVALID, "gs_runtime_info: the object needs %d bytes, the buffer holds %zu", n + 1, cap);
    }
    return GS_OK;
}

int32_t gs_ctx_comm_info(const gs_ctx *ctx, int32_t *rccl_ranks, int32_t *rccl_rank, int32_t *rccl_device)
{
    if (!ctx) return fail(GS_ERR_INVALID, "null context");
    int n = 0, r = -1, d = -1;
    if (ctx->comm) {
        Rccl *R = rccl();
        if (!R) return fail(GS_ERR_RCCL, "RCCL is not loaded");
        if (R->CommCount) GS_NCCL(R, R->CommCount(ctx->comm, &n));
        if (R->CommUserRank) GS_NCCL(R, R->CommUserRank(ctx->comm, &r));
        if (R->CommCuDevice) GS_NCCL(R, R->CommCuDevice(ctx->comm, &d));
    }
    if (rccl_ranks) *rccl_ranks = n;
    if (rccl_rank) *rccl_rank = r;
    if (rccl_device) *rccl_device = d;
    return GS_OK;
}

} // extern "C"
