// mock_rccl.hip -- TEST DOUBLE of the eleven RCCL entry points libptgpu.so resolves at run time (csrc/pt_comm.hip), built as
// librccl.so.1 and put in front of the real one by tests/test_gpu_parity.py (LD_LIBRARY_PATH of a child process that never imports
// torch). Purpose: execute the N > 1 paths of the C ABI -- rank offsets, gather slots, root selection, grouped collectives of several
// communicators -- on a box with ONE GPU, where the real RCCL can only ever form a one-rank communicator.
//
// What it is: every rank of a clique may live on the same device (ncclCommInitAll accepts repeats). When a rank posts a collective
// its STREAM blocks (hipStreamWaitValue32 on a per-call signal) exactly as it would behind RCCL's kernel waiting for its peers; once
// all ranks of the clique have posted the matching call (at ncclGroupEnd when inside a group) the collective is executed as
// device-to-device copies on an internal stream that first waits for everything each rank had enqueued before its call, and the
// signals are released. What it is NOT: RCCL. The HOST never blocks here, so it cannot show the deadlock one thread gets from
// ungrouped multi-communicator calls, and nothing exercises xGMI, IPC or the real library's kernels: real RCCL at N > 1 stays
// unexecuted until the driver's scaling run.
// It does check what RCCL checks about arguments: equal counts / types / roots across ranks, rank and root ranges, in-place layout.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum Kind { kAllGather, kGather, kAllReduce };
struct Op {
    Kind kind;
    const void *send;
    void *recv;
    size_t count;
    ncclDataType_t type;
    ncclRedOp_t red;
    int root;
    hipStream_t stream;
    hipEvent_t before;    // everything the rank had enqueued before the call
    uint32_t *signal;     // the rank's stream waits for *signal == 1
};
struct Clique {
    int n = 0;
    std::vector<int> devices;
    std::vector<std::vector<Op>> posted;   // per rank, in call order
    size_t resolved = 0;
    hipStream_t stream = nullptr;
    uint64_t *tmp = nullptr;   // staging of the all-reduce
    int joined = 0, alive = 0;
    unsigned long long executed[3] = {0, 0, 0};
};
std::mutex g_mu;
std::map<std::string, Clique *> g_by_id;
thread_local std::vector<Clique *> t_touched;   // cliques this thread posted to inside its open group
thread_local int t_depth = 0;
char g_err[256] = "mock rccl: no error";

}  // namespace

struct ncclComm {
    Clique *clique;
    int rank;
};

namespace {

size_t type_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

__global__ void sum_u64(const uint64_t *in, uint64_t *out, int n, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t s = 0;
        for (int r = 0; r < n; ++r) s += in[(size_t)r * count + i];
        out[i] = s;
    }
}

ncclResult_t bad(const char *what) {
    snprintf(g_err, sizeof g_err, "mock rccl: %s", what);
    return ncclInvalidArgument;
}
#define HIPOK(e) do { if ((e) != hipSuccess) { snprintf(g_err, sizeof g_err, "mock rccl: %s failed", #e); return ncclUnhandledCudaError; } } while (0)

ncclResult_t execute(Clique *q, size_t k) {
    const Op &o0 = q->posted[0][k];
    if (getenv("MOCK_RCCL_LOG")) fprintf(stderr, "[mock rccl] executing call %zu kind %d\n", k, (int)o0.kind);
    for (int r = 1; r < q->n; ++r) {
        const Op &o = q->posted[r][k];
        if (o.kind != o0.kind || o.count != o0.count || o.type != o0.type || o.root != o0.root || o.red != o0.red) return bad("ranks disagree about a collective (kind / count / type / root)");
    }
    HIPOK(hipSetDevice(q->devices[0]));
    if (!q->stream) HIPOK(hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking));
    for (int r = 0; r < q->n; ++r) {   // the internal stream waits for whatever each rank enqueued before its call
        HIPOK(hipStreamWaitEvent(q->stream, q->posted[r][k].before, 0));
        HIPOK(hipEventDestroy(q->posted[r][k].before));
    }
    const size_t bytes = o0.count * type_size(o0.type);
    if (o0.kind == kAllGather || o0.kind == kGather) {
        for (int d = 0; d < q->n; ++d) {
            if (o0.kind == kGather && d != o0.root) continue;
            char *recv = static_cast<char *>(q->posted[d][k].recv);
            if (!recv) return bad("receiving rank passed a NULL receive buffer");
            for (int s = 0; s < q->n; ++s) {
                const void *send = q->posted[s][k].send;
                if (recv + (size_t)s * bytes != send) HIPOK(hipMemcpyAsync(recv + (size_t)s * bytes, send, bytes, hipMemcpyDeviceToDevice, q->stream));
            }
        }
    } else {
        if (o0.type != ncclUint64 || o0.red != ncclSum) return bad("the test double reduces ncclUint64 / ncclSum only");
        if (o0.count > 16) return bad("the test double reduces at most 16 elements");
        if (!q->tmp) HIPOK(hipMalloc((void **)&q->tmp, (size_t)(q->n + 1) * 16 * 8));
        for (int s = 0; s < q->n; ++s) HIPOK(hipMemcpyAsync(q->tmp + (size_t)s * o0.count, q->posted[s][k].send, bytes, hipMemcpyDeviceToDevice, q->stream));
        hipLaunchKernelGGL(sum_u64, dim3(1), dim3(64), 0, q->stream, q->tmp, q->tmp + (size_t)q->n * o0.count, q->n, o0.count);
        for (int d = 0; d < q->n; ++d) HIPOK(hipMemcpyAsync(q->posted[d][k].recv, q->tmp + (size_t)q->n * o0.count, bytes, hipMemcpyDeviceToDevice, q->stream));
    }
    for (int r = 0; r < q->n; ++r) HIPOK(hipStreamWriteValue32(q->stream, q->posted[r][k].signal, 1u, 0));   // ... and releases every rank's stream
    q->executed[o0.kind] += 1;
    return ncclSuccess;
}

ncclResult_t resolve(Clique *q) {
    for (;;) {
        for (int r = 0; r < q->n; ++r)
            if (q->posted[r].size() <= q->resolved) return ncclSuccess;   // somebody has not posted call number `resolved` yet
        if (ncclResult_t e = execute(q, q->resolved)) return e;
        q->resolved += 1;
    }
}

ncclResult_t post(ncclComm_t c, Op op) {
    if (!c || !c->clique) return bad("NULL communicator");
    std::lock_guard<std::mutex> lock(g_mu);
    if (getenv("MOCK_RCCL_LOG")) fprintf(stderr, "[mock rccl] rank %d posts kind %d count %zu root %d depth %d\n", c->rank, (int)op.kind, op.count, op.root, t_depth);
    Clique *q = c->clique;
    if (op.kind == kGather && (op.root < 0 || op.root >= q->n)) return bad("root out of range");
    if (!op.send) return bad("NULL send buffer");
    // the rank's stream: remember what came before the call, then block until the collective has run
    HIPOK(hipSetDevice(q->devices[c->rank]));
    // (nothing here may touch the NULL stream or synchronise the device: another rank's stream may already be blocked in this
    // very collective, and it lives on the same device. The signal is cleared BEFORE the event the executor waits for, so its
    // release cannot be overtaken by the clearing write.)
    HIPOK(hipExtMallocWithFlags((void **)&op.signal, 8, hipMallocSignalMemory));
    HIPOK(hipStreamWriteValue32(op.stream, op.signal, 0u, 0));
    HIPOK(hipEventCreateWithFlags(&op.before, hipEventDisableTiming));
    HIPOK(hipEventRecord(op.before, op.stream));
    HIPOK(hipStreamWaitValue32(op.stream, op.signal, 1u, hipStreamWaitValueEq, 0xffffffffu));
    q->posted[c->rank].push_back(op);
    if (t_depth > 0) {
        bool seen = false;
        for (Clique *t : t_touched) seen = seen || t == q;
        if (!seen) t_touched.push_back(q);
        return ncclSuccess;
    }
    return resolve(q);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
    if (!version) return ncclInvalidArgument;
    *version = NCCL_MAJOR * 10000 + 9900;   // (x.99.00: a version no real RCCL reports, so a test can tell the double was loaded)
    return ncclSuccess;
}
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    static unsigned long long counter = 0;
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "mock-rccl-%llu", ++counter);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks <= 0 || rank < 0 || rank >= nranks) return bad("bad rank / world");
    std::lock_guard<std::mutex> lock(g_mu);
    const std::string key(id.internal, sizeof id.internal);
    Clique *&q = g_by_id[key];
    if (!q) {
        q = new Clique();
        q->n = nranks;
        q->devices.assign(nranks, 0);
        q->posted.resize(nranks);
    }
    if (q->n != nranks) return bad("ranks disagree about the world size");
    int dev = 0;
    (void)hipGetDevice(&dev);
    q->devices[rank] = dev;
    q->joined += 1, q->alive += 1;
    *comm = new ncclComm{q, rank};
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    if (!comms || ndev <= 0) return bad("bad device list");
    std::lock_guard<std::mutex> lock(g_mu);
    Clique *q = new Clique();
    q->n = ndev;
    q->posted.resize(ndev);
    for (int i = 0; i < ndev; ++i) q->devices.push_back(devlist ? devlist[i] : i);   // (repeats are fine: that is the point)
    q->joined = q->alive = ndev;
    for (int i = 0; i < ndev; ++i) comms[i] = new ncclComm{q, i};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    std::lock_guard<std::mutex> lock(g_mu);
    Clique *q = comm->clique;
    if (q && --q->alive == 0) {
        if (q->stream) (void)hipStreamSynchronize(q->stream), (void)hipStreamDestroy(q->stream);
        (void)hipFree(q->tmp);
        for (auto it = g_by_id.begin(); it != g_by_id.end();) it = it->second == q ? g_by_id.erase(it) : std::next(it);
        delete q;
    }
    delete comm;
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    if (!recvbuff) return bad("NULL receive buffer");
    return post(comm, Op{kAllGather, sendbuff, recvbuff, sendcount, datatype, ncclSum, -1, stream, nullptr, nullptr});
}
ncclResult_t ncclGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
    return post(comm, Op{kGather, sendbuff, recvbuff, sendcount, datatype, ncclSum, root, stream, nullptr, nullptr});
}
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!recvbuff) return bad("NULL receive buffer");
    return post(comm, Op{kAllReduce, sendbuff, recvbuff, count, datatype, op, -1, stream, nullptr, nullptr});
}
ncclResult_t ncclGroupStart() {
    t_depth += 1;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return bad("ncclGroupEnd without ncclGroupStart");
    if (--t_depth > 0) return ncclSuccess;
    std::lock_guard<std::mutex> lock(g_mu);
    ncclResult_t rc = ncclSuccess;
    for (Clique *q : t_touched)
        if (ncclResult_t e = resolve(q)) rc = e;
    t_touched.clear();
    return rc;
}
const char *ncclGetErrorString(ncclResult_t result) { return result == ncclSuccess ? "no error" : g_err; }

// test hooks (not RCCL): how deep this thread's group nesting is (must be 0 after every ABI call), and how many collectives ran
int mock_rccl_group_depth() { return t_depth; }

}  // extern "C"
