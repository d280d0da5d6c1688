// shm_transport.cpp -- TEST DOUBLE for librccl: the eight nccl* entry points libgs_hip.so binds
// (gs_internal.h: struct Rccl), implemented over a POSIX shared-memory mailbox per ordered rank pair.
//
// Why: a 1-GPU box cannot give every rank its own GPU and RCCL refuses two ranks on one device, so
// the multi-process leg of the library (rank-local slab, K-row ncclSend/ncclRecv groups, ghost-depth
// tracking across processes) would otherwise first run in the driver's multi-GPU bench.  With
// GS_RCCL_LIBRARY pointing here, N processes share device 0 and everything except RCCL itself runs
// for real.  Signatures come from <rccl/rccl.h>, so a mismatch with the real library's ABI is a
// compile error here.
//
// Semantics kept: point-to-point messages between a pair are matched in issue order; operations are
// ordered after the work already enqueued on `stream` and before whatever is enqueued later.  They
// are implemented synchronously (stream sync + blocking copies through host memory) at
// ncclGroupEnd: all sends first, then all receives, which cannot deadlock because a mailbox holds
// kSlots undelivered messages and a group never sends more than that to one peer.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr int kSlots = 8;
constexpr size_t kSlotBytes = 1u << 20; // one message: K rows of one plane

struct Mailbox {
    std::atomic<uint64_t> head; // messages written
    std::atomic<uint64_t> tail; // messages consumed
    uint64_t bytes[kSlots];
    alignas(64) unsigned char data[kSlots][kSlotBytes];
};

struct Shared {
    std::atomic<int> attached;
    Mailbox box[kMaxRanks][kMaxRanks]; // [src][dst]
};

struct Comm {
    Shared *sh = nullptr;
    int rank = 0, n = 0;
    std::string name;
};

struct Op {
    bool send;
    void *ptr;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t dtype_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

bool wait_until(const std::function<bool()> &ready)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (!ready()) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false; // peer died
    }
    return true;
}

ncclResult_t run_ops()
{
    std::vector<Op> ops;
    ops.swap(g_ops);
    for (const Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (const Op &o : ops) {
        if (!o.send) continue;
        if (o.bytes > kSlotBytes) return ncclInvalidArgument;
        Mailbox &m = o.comm->sh->box[o.comm->rank][o.peer];
        if (!wait_until([&] { return m.head.load(std::memory_order_acquire) - m.tail.load(std::memory_order_acquire) < kSlots; }))
            return ncclSystemError;
        const uint64_t h = m.head.load(std::memory_order_relaxed);
        if (hipMemcpy(m.data[h % kSlots], o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        m.bytes[h % kSlots] = o.bytes;
        m.head.store(h + 1, std::memory_order_release);
    }
    for (const Op &o : ops) {
        if (o.send) continue;
        Mailbox &m = o.comm->sh->box[o.peer][o.comm->rank];
        if (!wait_until([&] { return m.head.load(std::memory_order_acquire) > m.tail.load(std::memory_order_acquire); }))
            return ncclSystemError;
        const uint64_t t = m.tail.load(std::memory_order_relaxed);
        if (m.bytes[t % kSlots] != o.bytes) {
            std::fprintf(stderr, "shm_transport: rank %d expected %zu bytes from %d, message has %llu\n",
                         o.comm->rank, o.bytes, o.peer, (unsigned long long)m.bytes[t % kSlots]);
            return ncclInvalidArgument; // sender and receiver disagree on the message size
        }
        if (hipMemcpy(o.ptr, m.data[t % kSlots], o.bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
        m.tail.store(t + 1, std::memory_order_release);
    }
    return ncclSuccess;
}

ncclResult_t post(bool send, void *ptr, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || peer < 0 || peer >= c->n || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{send, ptr, count * dtype_size(t), peer, c, stream});
    return g_depth > 0 ? ncclSuccess : run_ops();
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "/gs_shm_%d_%lld", (int)getpid(),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(Shared)) != 0) { close(fd); return ncclSystemError; }
    void *p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); // zero-filled by the kernel
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    Comm *c = new Comm;
    c->sh = static_cast<Shared *>(p);
    c->rank = rank;
    c->n = nranks;
    c->name = id.internal;
    // like the real call: returns once every rank has joined
    c->sh->attached.fetch_add(1, std::memory_order_acq_rel);
    if (!wait_until([&] { return c->sh->attached.load(std::memory_order_acquire) >= nranks; })) return ncclSystemError;
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    munmap(c->sh, sizeof(Shared));
    shm_unlink(c->name.c_str()); // the first caller removes the name; the mappings stay valid
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(true, const_cast<void *>(buf), count, t, peer, comm, stream);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(false, buf, count, t, peer, comm, stream);
}

ncclResult_t ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? run_ops() : ncclSuccess;
}

// what gs_ctx_comm_info reads
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    const Comm *c = reinterpret_cast<const Comm *>(comm);
    if (!c || !count) return ncclInvalidArgument;
    *count = c->n;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank)
{
    const Comm *c = reinterpret_cast<const Comm *>(comm);
    if (!c || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}

ncclResult_t ncclCommCuDevice(const ncclComm_t comm, int *device)
{
    if (!comm || !device) return ncclInvalidArgument;
    return hipGetDevice(device) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "shm_transport: HIP call failed";
    case ncclSystemError: return "shm_transport: shared memory / peer timeout";
    case ncclInvalidArgument: return "shm_transport: invalid argument or message size mismatch";
    case ncclInvalidUsage: return "shm_transport: invalid usage";
    default: return "shm_transport: error";
    }
}

} // extern "C"
