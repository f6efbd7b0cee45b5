// pt_comm.hip -- multi-GPU frames behind the C ABI. One frame is split by rows (row y -> rank y % world,
// disjoint pixels as scene.rs:90-93); the only exchange is ONE ncclAllGather / ncclGather of the float3 shards plus an 8-byte
// ncclAllReduce of the ray count (scene.rs:118-120). xGMI is point-to-point and the message is small (11.5 MB at 1200x800),
// so one collective, no ring tuning.
//
// RCCL is resolved at RUN time, when the first pt_comm_* function is called: PTGPU_RCCL_LIBRARY if set, else the library already
// loaded into the process (a PyTorch process has its own librccl.so, and two RCCL builds behind one soname must not meet), else the system's.
// The render entry points therefore work on machines without RCCL, and libptgpu.so has no link-time dependency on it.
#include "pt_host.h"

#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>   // types and constants only: every call goes through the table below

#include <mutex>
#include <new>

using namespace pthostside;

namespace {

struct Rccl {
    void *handle = nullptr;
    int version = 0;
    char path[256] = "";
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// nullptr + pt_last_error when RCCL cannot be had
const Rccl *rccl() {
    static std::mutex mu;
    static Rccl R;
    static bool tried = false, ok = false;
    static char why[384] = "";
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        // 1. whatever RCCL the process already runs (torch.distributed's own copy, for instance, which carries no soname): the
        //    first loaded object whose file name says librccl
        char loaded[320] = "";
        dl_iterate_phdr(
            [](struct dl_phdr_info *info, size_t, void *out) -> int {
                if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) {
                    snprintf(static_cast<char *>(out), 320, "%s", info->dlpi_name);
                    return 1;
                }
                return 0;
            },
            loaded);
        // 0. an explicit choice: PTGPU_RCCL_LIBRARY=/path/to/librccl.so pins the RCCL build this library talks to, whatever else the process
        //    has loaded (a deployment with several ROCm installations; the tests' cross-process double, tests/mock_rccl/mock_rccl_xproc.hip)
        const char *pinned = getenv("PTGPU_RCCL_LIBRARY");
        if (pinned && pinned[0]) {
            R.handle = dlopen(pinned, RTLD_NOW | RTLD_LOCAL);
            if (!R.handle) loaded[0] = 0;   // (fall through to the error below: a pinned library that cannot be loaded is not replaced silently)
        }
        if (!R.handle && !(pinned && pinned[0]) && loaded[0]) R.handle = dlopen(loaded, RTLD_NOW | RTLD_NOLOAD);
        const char *rocm = getenv("ROCM_PATH");
        char full[2][320];
        snprintf(full[0], sizeof full[0], "%s/lib/librccl.so.1", rocm ? rocm : "/opt/rocm");
        snprintf(full[1], sizeof full[1], "/opt/rocm/lib/librccl.so.1");
        const char *later[] = {"librccl.so.1", full[0], full[1]};
        for (const char *n : later)   // 2. the loader's search path, then the ROCm installation
            if (!R.handle && !(pinned && pinned[0])) R.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!R.handle) {
            snprintf(why, sizeof why, "RCCL is not available (dlopen %s: %s)", (pinned && pinned[0]) ? pinned : "librccl.so.1", dlerror());
        } else {
            bool all = true;
#define PT_RCCL_SYM(field, sym) all = ((R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.handle, sym))) != nullptr) && all
            PT_RCCL_SYM(GetVersion, "ncclGetVersion");
            PT_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
            PT_RCCL_SYM(CommInitRank, "ncclCommInitRank");
            PT_RCCL_SYM(CommInitAll, "ncclCommInitAll");
            PT_RCCL_SYM(CommDestroy, "ncclCommDestroy");
            PT_RCCL_SYM(AllGather, "ncclAllGather");
            PT_RCCL_SYM(Gather, "ncclGather");
            PT_RCCL_SYM(AllReduce, "ncclAllReduce");
            PT_RCCL_SYM(GroupStart, "ncclGroupStart");
            PT_RCCL_SYM(GroupEnd, "ncclGroupEnd");
            PT_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef PT_RCCL_SYM
            Dl_info info;
            if (R.GetVersion && dladdr(reinterpret_cast<void *>(R.GetVersion), &info) && info.dli_fname) snprintf(R.path, sizeof R.path, "%s", info.dli_fname);
            if (!all) {
                snprintf(why, sizeof why, "%s lacks an entry point this library calls", R.path[0] ? R.path : "librccl");
            } else if (R.GetVersion(&R.version) != ncclSuccess || R.version / 10000 != NCCL_MAJOR) {
                // only the C signatures of the eleven calls above are relied on, and they are the same across RCCL 2.x: the
                // header's minor version need not match the runtime's (PyTorch 2.10 ships 2.26, ROCm 7.2 has 2.27)
                snprintf(why, sizeof why, "%s reports RCCL version code %d, this library was written against major version %d", R.path, R.version, NCCL_MAJOR);
            } else {
                ok = true;
            }
        }
    }
    if (!ok) {
        fail(PT_ERR_UNSUPPORTED, "%s", why);
        return nullptr;
    }
    return &R;
}

}  // namespace

struct pt_comm {
    const Rccl *R = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0;
    uint32_t rank = 0, world = 1;
    float *d_gather = nullptr;       // [world][ceil(H / world)][W][3]; this rank's shard is rendered in place in slot `rank`
    size_t gather_floats = 0;
    // what slot `rank` holds after a pt_render_sharded call: the rows of frame `slot_frame` at slot_w x slot_h. A rank that
    // does not receive the frame (root >= 0) blends its next progressive frame against these rows, not against its stale buffer.
    bool slot_valid = false;
    uint32_t slot_frame = 0, slot_w = 0, slot_h = 0;
};

#define NCCL_TRY(R, call)                                                                                 \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) return fail(PT_ERR_HIP, "%s failed: %s", #call, (R)->GetErrorString(r_)); \
    } while (0)

namespace {

__global__ void shard_pack_kernel(const float *full, float *shard, uint32_t row_floats, uint32_t rows, uint32_t index, uint32_t count) {
    const size_t n = (size_t)rows * row_floats;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i / row_floats, x = i - j * row_floats;
        shard[i] = full[(j * count + index) * row_floats + x];
    }
}

// row y of the frame = gathered[y % count][y / count]
__global__ void shard_unpack_kernel(const float *gathered, float *full, uint32_t row_floats, uint32_t height, uint32_t count, uint32_t prow) {
    const size_t n = (size_t)height * row_floats;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t y = i / row_floats, x = i - y * row_floats;
        full[i] = gathered[((y % count) * prow + y / count) * row_floats + x];
    }
}

uint32_t copy_grid(size_t n) { return (uint32_t)std::min<size_t>((n + 255) / 256, 4096); }

// (the clearing memset runs on the CALLER's stream: a NULL-stream hipMemset is not ordered against a non-blocking stream, and what is
// enqueued next on `stream` writes this rank's rows into the buffer -- found by tests/mock_rccl: whole shards arrived as zeros)
int comm_ensure(pt_comm *c, uint32_t width, uint32_t height, hipStream_t stream) {
    const uint32_t prow = (height + c->world - 1) / c->world;
    const size_t need = (size_t)c->world * prow * width * 3u;
    if (need <= c->gather_floats) return PT_OK;
    (void)hipFree(c->d_gather);
    c->d_gather = nullptr;
    c->gather_floats = 0;
    c->slot_valid = false;
    HIP_TRY(hipMalloc((void **)&c->d_gather, need * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->d_gather, 0, need * sizeof(float), stream));   // ranks with one row less send a zero row
    HIP_TRY(hipStreamSynchronize(stream));   // (once per growth: a later call may bring ANOTHER stream, which nothing would order behind this memset)
    c->gather_floats = need;
    return PT_OK;
}

// The exchange of one rank, in two halves so that several ranks driven by ONE thread can share a group (RCCL: collectives of
// different communicators issued by one thread must sit inside one ncclGroupStart / ncclGroupEnd, or the first rank's call
// waits for peers that this thread has not reached yet). The shard already sits in slot `rank` of c->d_gather.
int comm_post(pt_comm *c, uint32_t width, uint32_t height, uint64_t *d_ray_count, int root, hipStream_t stream) {
    const uint32_t prow = (height + c->world - 1) / c->world;
    const size_t slot = (size_t)prow * width * 3u;
    const Rccl *R = c->R;
    if (root < 0)
        NCCL_TRY(R, R->AllGather(c->d_gather + (size_t)c->rank * slot, c->d_gather, slot, ncclFloat, c->comm, stream));
    else
        NCCL_TRY(R, R->Gather(c->d_gather + (size_t)c->rank * slot, c->d_gather, slot, ncclFloat, root, c->comm, stream));
    NCCL_TRY(R, R->AllReduce(d_ray_count, d_ray_count, 1, ncclUint64, ncclSum, c->comm, stream));
    return PT_OK;
}
int comm_unpack(pt_comm *c, uint32_t width, uint32_t height, float *d_rgb_full, int root, hipStream_t stream) {
    if (root >= 0 && (uint32_t)root != c->rank) return PT_OK;   // this rank does not receive the frame
    const uint32_t prow = (height + c->world - 1) / c->world;
    const size_t n = (size_t)height * width * 3u;
    hipLaunchKernelGGL(shard_unpack_kernel, dim3(copy_grid(n)), dim3(256), 0, stream, c->d_gather, d_rgb_full, width * 3u, height, c->world, prow);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}
// one group around the posts of `n` ranks; the group is closed on EVERY path (an open group would swallow the thread's later calls)
template <typename Post>
int comm_grouped(const Rccl *R, uint32_t n, Post post) {
    NCCL_TRY(R, R->GroupStart());
    int rc = PT_OK;
    for (uint32_t i = 0; i < n && rc == PT_OK; ++i) rc = post(i);
    const ncclResult_t e = R->GroupEnd();
    if (rc != PT_OK) return rc;   // (pt_last_error holds the post's message)
    if (e != ncclSuccess) return fail(PT_ERR_HIP, "ncclGroupEnd failed: %s", R->GetErrorString(e));
    return PT_OK;
}
int comm_exchange(pt_comm *c, uint32_t width, uint32_t height, float *d_rgb_full, uint64_t *d_ray_count, int root, hipStream_t stream) {
    const bool have_frame = root < 0 || (uint32_t)root == c->rank;
    if (have_frame && !d_rgb_full) return fail(PT_ERR_INVALID_ARG, "d_rgb_full is NULL on a rank that receives the frame");
    // (a one-rank communicator goes through the same calls: that is what a 1-GPU box can test with the real RCCL)
    if (int rc = comm_grouped(c->R, 1u, [&](uint32_t) { return comm_post(c, width, height, d_ray_count, root, stream); })) return rc;
    return comm_unpack(c, width, height, d_rgb_full, root, stream);
}

// Argument checks of one rank of pt_render_sharded[_all], nothing enqueued: the all-ranks form runs them for EVERY rank before the
// first render goes out (a rank failing in mid-loop would leave the ranks before it rendered for a frame that is never exchanged).
// The blend reads the previous frame (scene.rs:114-116): this rank's rows. After an all-gather every rank's full buffer is
// current, and they are packed out of it. A rank that does NOT receive the frame (root >= 0, another rank) has a stale
// buffer; its gather slot still holds exactly the rows it rendered for the frame before, so those are kept instead.
int sharded_check_one(const pt_scene *s, const pt_comm *c, const pt_params *params, uint32_t frame_num, const float *d_rgb_full_inout, const uint64_t *d_ray_count,
                      int root, bool *keep_slot_out) {
    if (!s || !c || !d_ray_count) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (s->device != c->device) return fail(PT_ERR_INVALID_ARG, "scene lives on device %d, communicator on %d", s->device, c->device);
    if (root >= (int)c->world) return fail(PT_ERR_INVALID_ARG, "root %d out of range (%u ranks)", root, c->world);
    const bool receives = root < 0 || (uint32_t)root == c->rank;
    if (receives && !d_rgb_full_inout) return fail(PT_ERR_INVALID_ARG, "d_rgb_full_inout is NULL on a rank that receives the frame");
    const bool slot_is_previous = c->slot_valid && frame_num > 0 && c->slot_frame + 1u == frame_num && c->slot_w == params->width && c->slot_h == params->height;
    const bool keep = slot_is_previous && !receives;
    if (!keep && !d_rgb_full_inout) return fail(PT_ERR_INVALID_ARG, "d_rgb_full_inout is NULL and the communicator does not hold this rank's previous rows");
    if (keep_slot_out) *keep_slot_out = keep;
    return PT_OK;
}

// prepares the rank's slot and enqueues its render (shared by the one-rank and the all-ranks forms of pt_render_sharded)
int sharded_render_one(pt_scene *s, pt_comm *c, const pt_params *params, const pt_camera *cam, uint32_t frame_num, float *d_rgb_full_inout, uint64_t *d_ray_count,
                       int root, hipStream_t stream) {
    bool keep_slot = false;
    if (int rc = sharded_check_one(s, c, params, frame_num, d_rgb_full_inout, d_ray_count, root, &keep_slot)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = comm_ensure(c, params->width, params->height, stream)) return rc;
    const uint32_t prow = (params->height + c->world - 1) / c->world;
    float *slot = c->d_gather + (size_t)c->rank * prow * params->width * 3u;
    if (!keep_slot)
        if (int rc = pt_shard_pack(d_rgb_full_inout, slot, params->width, params->height, c->rank, c->world, stream)) return rc;
    c->slot_valid = false;
    if (int rc = launch(s, params, cam, frame_num, c->rank, c->world, slot, d_ray_count, stream)) return rc;
    c->slot_valid = true, c->slot_frame = frame_num, c->slot_w = params->width, c->slot_h = params->height;
    return PT_OK;
}

// the communicators of a pt_*_all call: one clique, one process, in rank order
int check_clique(pt_comm *const *comms, uint32_t n) {
    if (!comms || n == 0) return fail(PT_ERR_INVALID_ARG, "no communicators");
    for (uint32_t i = 0; i < n; ++i) {
        if (!comms[i]) return fail(PT_ERR_INVALID_ARG, "comms[%u] is NULL", i);
        if (comms[i]->world != n || comms[i]->rank != i) return fail(PT_ERR_INVALID_ARG, "comms[%u] is rank %u of %u: pass all %u communicators of pt_comm_create_all in rank order", i, comms[i]->rank, comms[i]->world, comms[i]->world);
    }
    return PT_OK;
}

}  // namespace

extern "C" int pt_comm_runtime(int *version_code_out, char *path_out, size_t path_capacity) {
    const Rccl *R = rccl();
    if (!R) return PT_ERR_UNSUPPORTED;
    if (version_code_out) *version_code_out = R->version;
    if (path_out && path_capacity) snprintf(path_out, path_capacity, "%s", R->path);
    return PT_OK;
}

extern "C" int pt_comm_unique_id(uint8_t id_out[PT_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    if (!id_out) return fail(PT_ERR_INVALID_ARG, "id_out is NULL");
    const Rccl *R = rccl();
    if (!R) return PT_ERR_UNSUPPORTED;
    ncclUniqueId id;
    NCCL_TRY(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return PT_OK;
}

extern "C" int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], uint32_t rank, uint32_t world, int device, pt_comm **comm_out) {
    if (!id || !comm_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    *comm_out = nullptr;
    if (world == 0 || rank >= world) return fail(PT_ERR_INVALID_ARG, "bad rank %u of %u", rank, world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(PT_ERR_INVALID_ARG, "device %d out of range (%d devices)", device, ndev);
    const Rccl *R = rccl();
    if (!R) return PT_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(device));
    pt_comm *c = new (std::nothrow) pt_comm();
    if (!c) return fail(PT_ERR_INVALID_ARG, "out of host memory");
    c->R = R, c->device = device, c->rank = rank, c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = R->CommInitRank(&c->comm, (int)world, uid, (int)rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(PT_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString(r));
    }
    *comm_out = c;
    return PT_OK;
}

extern "C" int pt_comm_create_all(const int *devices, uint32_t n, pt_comm **comms_out) {
    if (!devices || !comms_out || n == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / no devices");
    for (uint32_t i = 0; i < n; ++i) comms_out[i] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    for (uint32_t i = 0; i < n; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return fail(PT_ERR_INVALID_ARG, "device %d out of range (%d devices)", devices[i], ndev);
    const Rccl *R = rccl();
    if (!R) return PT_ERR_UNSUPPORTED;
    std::vector<ncclComm_t> cs(n, nullptr);
    NCCL_TRY(R, R->CommInitAll(cs.data(), (int)n, devices));
    for (uint32_t i = 0; i < n; ++i) {
        pt_comm *c = new (std::nothrow) pt_comm();
        if (!c) {
            for (uint32_t j = 0; j < n; ++j) {
                if (j < i) comms_out[j]->comm = nullptr, delete comms_out[j], comms_out[j] = nullptr;
                (void)R->CommDestroy(cs[j]);
            }
            return fail(PT_ERR_INVALID_ARG, "out of host memory");
        }
        c->R = R, c->comm = cs[i], c->device = devices[i], c->rank = i, c->world = n;
        comms_out[i] = c;
    }
    return PT_OK;
}

extern "C" void pt_comm_destroy(pt_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && c->R) (void)c->R->CommDestroy(c->comm);
    (void)hipFree(c->d_gather);
    delete c;
}

extern "C" int pt_comm_rank(const pt_comm *c, uint32_t *rank_out, uint32_t *world_out) {
    if (!c) return fail(PT_ERR_INVALID_ARG, "comm is NULL");
    if (rank_out) *rank_out = c->rank;
    if (world_out) *world_out = c->world;
    return PT_OK;
}

extern "C" int pt_shard_pack(const float *d_rgb_full, float *d_rgb_shard, uint32_t width, uint32_t height, uint32_t shard_index, uint32_t shard_count,
                             void *hip_stream) {
    if (!d_rgb_full || !d_rgb_shard || width == 0 || height == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / empty frame");
    if (shard_count == 0 || shard_index >= shard_count) return fail(PT_ERR_INVALID_ARG, "bad shard %u/%u", shard_index, shard_count);
    const uint32_t rows = pt_shard_rows(height, shard_index, shard_count);
    if (rows == 0) return PT_OK;
    const size_t n = (size_t)rows * width * 3u;
    hipLaunchKernelGGL(shard_pack_kernel, dim3(copy_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), d_rgb_full, d_rgb_shard, width * 3u, rows,
                       shard_index, shard_count);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

extern "C" int pt_shard_unpack_all(const float *d_gathered, float *d_rgb_full, uint32_t width, uint32_t height, uint32_t shard_count, void *hip_stream) {
    if (!d_gathered || !d_rgb_full || width == 0 || height == 0 || shard_count == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / empty frame");
    const size_t n = (size_t)height * width * 3u;
    hipLaunchKernelGGL(shard_unpack_kernel, dim3(copy_grid(n)), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), d_gathered, d_rgb_full, width * 3u, height,
                       shard_count, (height + shard_count - 1) / shard_count);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

// copy a rank's compact shard into its gather slot
static int stage_shard(pt_comm *c, uint32_t width, uint32_t height, const float *d_rgb_shard, hipStream_t stream) {
    HIP_TRY(hipSetDevice(c->device));
    if (int rc = comm_ensure(c, width, height, stream)) return rc;
    const uint32_t prow = (height + c->world - 1) / c->world, rows = pt_shard_rows(height, c->rank, c->world);
    c->slot_valid = false;   // (the slot now holds whatever the caller rendered: only pt_render_sharded knows which frame that is)
    if (rows)
        HIP_TRY(hipMemcpyAsync(c->d_gather + (size_t)c->rank * prow * width * 3u, d_rgb_shard, (size_t)rows * width * 3u * sizeof(float), hipMemcpyDeviceToDevice,
                               stream));
    return PT_OK;
}

extern "C" int pt_comm_gather_frame(pt_comm *c, uint32_t width, uint32_t height, const float *d_rgb_shard, float *d_rgb_full, uint64_t *d_ray_count, int root,
                                    void *hip_stream) {
    if (!c || !d_rgb_shard || !d_ray_count || width == 0 || height == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / empty frame");
    if (root >= (int)c->world) return fail(PT_ERR_INVALID_ARG, "root %d out of range (%u ranks)", root, c->world);
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    if (int rc = stage_shard(c, width, height, d_rgb_shard, stream)) return rc;
    return comm_exchange(c, width, height, d_rgb_full, d_ray_count, root, stream);
}

extern "C" int pt_comm_gather_frame_all(pt_comm *const *comms, uint32_t n, uint32_t width, uint32_t height, const float *const *d_rgb_shards, float *const *d_rgb_fulls,
                                        uint64_t *const *d_ray_counts, int root, void *const *hip_streams) {
    if (int rc = check_clique(comms, n)) return rc;
    if (!d_rgb_shards || !d_rgb_fulls || !d_ray_counts || width == 0 || height == 0) return fail(PT_ERR_INVALID_ARG, "NULL argument / empty frame");
    if (root >= (int)n) return fail(PT_ERR_INVALID_ARG, "root %d out of range (%u ranks)", root, n);
    const auto stream_of = [&](uint32_t i) { return reinterpret_cast<hipStream_t>(hip_streams ? hip_streams[i] : nullptr); };
    for (uint32_t i = 0; i < n; ++i) {
        const bool receives = root < 0 || (uint32_t)root == i;
        if (!d_rgb_shards[i] || !d_ray_counts[i] || (receives && !d_rgb_fulls[i])) return fail(PT_ERR_INVALID_ARG, "NULL buffer for rank %u", i);
        if (int rc = stage_shard(comms[i], width, height, d_rgb_shards[i], stream_of(i))) return rc;
    }
    if (int rc = comm_grouped(comms[0]->R, n, [&](uint32_t i) {
            (void)hipSetDevice(comms[i]->device);
            return comm_post(comms[i], width, height, d_ray_counts[i], root, stream_of(i));
        }))
        return rc;
    for (uint32_t i = 0; i < n; ++i) {
        HIP_TRY(hipSetDevice(comms[i]->device));
        if (int rc = comm_unpack(comms[i], width, height, d_rgb_fulls[i], root, stream_of(i))) return rc;
    }
    return PT_OK;
}

extern "C" int pt_render_sharded(pt_scene *s, pt_comm *c, const pt_params *params, const pt_camera *cam, uint32_t frame_num, float *d_rgb_full_inout,
                                 uint64_t *d_ray_count, int root, void *hip_stream) {
    if (!s || !c || !params || !cam || !d_ray_count) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0) return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    hipStream_t stream = reinterpret_cast<hipStream_t>(hip_stream);
    if (int rc = sharded_render_one(s, c, params, cam, frame_num, d_rgb_full_inout, d_ray_count, root, stream)) return rc;
    return comm_exchange(c, params->width, params->height, d_rgb_full_inout, d_ray_count, root, stream);
}

extern "C" int pt_render_sharded_all(pt_scene *const *scenes, pt_comm *const *comms, uint32_t n, const pt_params *params, const pt_camera *cam, uint32_t frame_num,
                                     float *const *d_rgb_full_inout, uint64_t *const *d_ray_counts, int root, void *const *hip_streams) {
    if (int rc = check_clique(comms, n)) return rc;
    if (!scenes || !params || !cam || !d_rgb_full_inout || !d_ray_counts) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0) return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    if (root >= (int)n) return fail(PT_ERR_INVALID_ARG, "root %d out of range (%u ranks)", root, n);
    const auto stream_of = [&](uint32_t i) { return reinterpret_cast<hipStream_t>(hip_streams ? hip_streams[i] : nullptr); };
    // every rank's arguments are checked before anything is enqueued for any of them
    for (uint32_t i = 0; i < n; ++i)
        if (int rc = sharded_check_one(scenes[i], comms[i], params, frame_num, d_rgb_full_inout[i], d_ray_counts[i], root, nullptr)) return rc;
    // every rank's pack + render first (nothing here waits for a peer), then ALL ranks' collectives inside one group, then the unpacks
    for (uint32_t i = 0; i < n; ++i)
        if (int rc = sharded_render_one(scenes[i], comms[i], params, cam, frame_num, d_rgb_full_inout[i], d_ray_counts[i], root, stream_of(i))) {
            // (a launch failure in mid-loop: the ranks before this one rendered a frame that will not be exchanged -- their slots hold no
            //  frame the next call may build on)
            for (uint32_t j = 0; j < n; ++j) comms[j]->slot_valid = false;
            return rc;
        }
    if (int rc = comm_grouped(comms[0]->R, n, [&](uint32_t i) {
            (void)hipSetDevice(comms[i]->device);
            return comm_post(comms[i], params->width, params->height, d_ray_counts[i], root, stream_of(i));
        }))
        return rc;
    for (uint32_t i = 0; i < n; ++i) {
        HIP_TRY(hipSetDevice(comms[i]->device));
        if (int rc = comm_unpack(comms[i], params->width, params->height, d_rgb_full_inout[i], root, stream_of(i))) return rc;
    }
    return PT_OK;
}
