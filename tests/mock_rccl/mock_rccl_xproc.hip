// mock_rccl_xproc.hip -- a second TEST DOUBLE of the eleven RCCL entry points libptgpu.so resolves at run time (csrc/pt_comm.hip), this
// one ACROSS PROCESSES: the ranks of a communicator are separate processes (one per rank, as `python -m torch.distributed.run` starts
// them) that may all sit on the same GPU. Purpose: run bench.py's own N > 1 code path -- the unique id travelling over the launcher's
// process group, pt_comm_create per rank, shard renders, pt_comm_gather_frame on a second stream, the sharded == single self-check -- on
// a box with ONE GPU, where real RCCL refuses two ranks on one device ("Duplicate GPU detected"). Loaded through PTGPU_RCCL_LIBRARY.
//
// What it is: every collective is BLOCKING and goes through host memory -- the caller's stream is synchronised, the rank's send buffer is
// copied into a POSIX shared-memory segment named by the unique id, the ranks meet at a barrier in that segment, each receiving rank
// copies what it is owed back to its device, and a second barrier frees the segment for the next call. What it is NOT: RCCL -- nothing
// here exercises xGMI, IPC handles, or the real library's kernels and stream semantics; real RCCL with more than one rank stays
// unexecuted until the driver's scaling run (bench.py's self-check makes that run prove itself).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 24u << 20;   // per rank and call: a 2400 x 1600 float3 frame split in two still fits
struct Seg {
    std::atomic<uint32_t> joined, left;
    std::atomic<uint32_t> count, sense;   // sense-reversing barrier
    std::atomic<uint32_t> failed;
    uint32_t pad[11];
    unsigned char data[kMaxRanks][kSlotBytes];
};

char g_err[256] = "mock rccl (xproc): no error";
thread_local int t_depth = 0;

struct Op {
    int kind;   // 0 all-gather, 1 gather, 2 all-reduce (u64 sum)
    const void *send;
    void *recv;
    size_t bytes;
    int root;
    hipStream_t stream;
    struct ncclComm *comm;
};
thread_local std::vector<Op> t_ops;

}  // namespace

struct ncclComm {
    Seg *seg;
    int rank, world;
    uint32_t my_sense;
    char name[64];
};

namespace {

ncclResult_t bad(const char *what) {
    snprintf(g_err, sizeof g_err, "mock rccl (xproc): %s", what);
    return ncclInvalidArgument;
}
double now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
bool barrier(ncclComm *c) {
    Seg *s = c->seg;
    c->my_sense ^= 1u;
    if (s->count.fetch_add(1u) + 1u == (uint32_t)c->world) {
        s->count.store(0u);
        s->sense.store(c->my_sense);
        return true;
    }
    const double t0 = now();
    while (s->sense.load() != c->my_sense) {
        if (s->failed.load() || now() - t0 > 120.0) {
            s->failed.store(1u);
            snprintf(g_err, sizeof g_err, "mock rccl (xproc): rank %d waited 120 s for its peers", c->rank);
            return false;
        }
        usleep(50);
    }
    return true;
}
ncclResult_t run(const Op &o) {
    ncclComm *c = o.comm;
    if (o.bytes > kSlotBytes) return bad("message larger than the test double's slot");
    if (hipStreamSynchronize(o.stream) != hipSuccess) return bad("hipStreamSynchronize failed");
    if (hipMemcpy(c->seg->data[c->rank], o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return bad("copy to the segment failed");
    if (!barrier(c)) return ncclSystemError;
    if (o.kind == 2) {
        if (o.bytes != 8) return bad("the test double reduces one ncclUint64");
        uint64_t sum = 0;
        for (int r = 0; r < c->world; ++r) {
            uint64_t v;
            memcpy(&v, c->seg->data[r], 8);
            sum += v;
        }
        if (hipMemcpy(o.recv, &sum, 8, hipMemcpyHostToDevice) != hipSuccess) return bad("copy from the segment failed");
    } else if (o.kind == 0 || c->rank == o.root) {
        if (!o.recv) return bad("receiving rank passed a NULL receive buffer");
        for (int r = 0; r < c->world; ++r)
            if (hipMemcpy(static_cast<char *>(o.recv) + (size_t)r * o.bytes, c->seg->data[r], o.bytes, hipMemcpyHostToDevice) != hipSuccess) return bad("copy from the segment failed");
    }
    if (!barrier(c)) return ncclSystemError;   // (nobody overwrites a slot that a peer is still reading)
    return ncclSuccess;
}
ncclResult_t post(const Op &o) {
    if (t_depth > 0) {
        t_ops.push_back(o);
        return ncclSuccess;
    }
    return run(o);
}
size_t type_size(ncclDataType_t t) {
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *v) {
    *v = 29901;   // (major 2: what libptgpu.so checks; 99xx marks the double, xx01 the cross-process one)
    return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t) { return g_err; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/mock_rccl_%d_%lld", (int)getpid(), (long long)(now() * 1e6));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return bad("bad rank / world (the double holds up to 8 ranks)");
    id.internal[sizeof id.internal - 1] = 0;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return bad("shm_open failed");
    if (ftruncate(fd, sizeof(Seg)) != 0) return bad("ftruncate failed");   // (a fresh segment reads as zeros: counters start at 0)
    void *p = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return bad("mmap failed");
    ncclComm *c = new ncclComm{static_cast<Seg *>(p), rank, world, 0u, ""};
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->seg->joined.fetch_add(1u);
    const double t0 = now();
    while (c->seg->joined.load() < (uint32_t)world) {   // every rank has the segment mapped before anybody uses it
        if (now() - t0 > 120.0) return bad("peers did not join within 120 s");
        usleep(100);
    }
    *out = c;
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return bad("ncclCommInitAll: the cross-process double has one rank per process"); }
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->seg->left.fetch_add(1u) + 1u == (uint32_t)c->world) shm_unlink(c->name);
    munmap(c->seg, sizeof(Seg));
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    return post(Op{0, send, recv, count * type_size(type), -1, stream, comm});
}
ncclResult_t ncclGather(const void *send, void *recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream) {
    if (root < 0 || root >= comm->world) return bad("gather root out of range");
    return post(Op{1, send, recv, count * type_size(type), root, stream, comm});
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (type != ncclUint64 || op != ncclSum || count != 1) return bad("the test double reduces one ncclUint64 with ncclSum");
    return post(Op{2, send, recv, 8, -1, stream, comm});
}
ncclResult_t ncclGroupStart() {
    t_depth += 1;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return bad("ncclGroupEnd without ncclGroupStart");
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    for (const Op &o : ops) {   // (every rank issued the same sequence: the calls meet in order)
        const ncclResult_t r = run(o);
        if (r != ncclSuccess) return r;
    }
    return ncclSuccess;
}

}  // extern "C"
