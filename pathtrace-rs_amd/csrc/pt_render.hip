// pt_render.hip -- the render entry points of the C ABI over launch() (pt_launch.hip): device-resident buffer, one multi-GPU
// shard, and the reference's own contract -- Scene::update on a HOST buffer (offline.rs:27-34 times exactly that call).
//
// Host-buffer pipeline (pageable memory, the default): the kernels render into a pinned + mapped copy of the caller's buffer
// over PCIe (each lane reads the previous frame's 12 bytes and writes the new ones when its pixel completes), so there is no
// D2H phase after the kernel. The copies between the caller's pages and the pinned ones are done by a few helper threads:
//   * in:  under the MEASURING launch (the first sample of every pixel, which touches no pixel buffer). A buffer that holds
//          +0.0f everywhere -- offline.rs:25 allocates exactly that -- is not copied at all: the scan that establishes it
//          runs under the measuring launch as well, and the frame kernel blends against 0.0f (same arithmetic, same bits);
//   * out: after the frame kernel, in parallel.
// A buffer registered with pt_buffer_register is rendered in place, without either copy.
#include "pt_host.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace pthostside;

namespace {

// ---- helper threads for the host-side copies --------------------------------------------------------------------------
// Started on first use, parked on a condition variable in between, joined at process exit. The calling thread always takes
// part, so a pool of zero helpers (PTGPU_HOST_THREADS=0 in development builds, or a single-CPU host) is just a loop.
class HostPool {
  public:
    static HostPool &get() {
        static HostPool pool;
        return pool;
    }
    // fn(chunk) for chunk in [0, n): chunks are claimed from a shared counter by the helpers and the caller
    void run(size_t n, const std::function<void(size_t)> &fn) {
        if (n == 0) return;
        std::lock_guard<std::mutex> one_job(job_mu_);   // one job at a time (pt_render calls of different scenes may race here)
        std::unique_lock<std::mutex> lock(mu_);
        start_helpers();
        fn_ = &fn, n_ = n, next_.store(0), done_.store(0);
        ++epoch_;
        cv_.notify_all();
        lock.unlock();
        work();
        lock.lock();
        idle_.wait(lock, [&] { return done_.load() == n_ && active_ == 0; });
        fn_ = nullptr;
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
            cv_.notify_all();
        }
        for (std::thread &t : threads_) t.join();
    }

  private:
    void start_helpers() {
        if (started_) return;
        started_ = true;
        int n = dev_knobs().host_threads;
        if (n < 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            n = hw > 1 ? (int)std::min<unsigned>(7u, hw - 1u) : 0;   // (11.5 MB: 0.34 ms with 4 threads, memory-bound beyond ~8)
        }
        for (int i = 0; i < n; ++i) threads_.emplace_back([this] { helper(); });
    }
    void work() {
        for (;;) {
            const size_t i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
            done_.fetch_add(1);
        }
    }
    void helper() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
            cv_.wait(lock, [&] { return stop_ || epoch_ != seen; });
            if (stop_) return;
            seen = epoch_;
            if (next_.load() >= n_) continue;   // woke after the job's last chunk was claimed: the poster may already be gone
            ++active_;
            lock.unlock();
            work();
            lock.lock();
            --active_;
            idle_.notify_all();
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_, idle_;
    std::vector<std::thread> threads_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0}, done_{0};
    uint64_t epoch_ = 0;
    int active_ = 0;
    bool started_ = false, stop_ = false;
};

constexpr size_t kCopyChunk = 256u * 1024u;

void parallel_copy(void *dst, const void *src, size_t bytes) {
    const size_t n = (bytes + kCopyChunk - 1) / kCopyChunk;
    HostPool::get().run(n, [&](size_t i) {
        const size_t off = i * kCopyChunk;
        memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, std::min(kCopyChunk, bytes - off));
    });
}

// every byte zero, i.e. every float +0.0f (-0.0f has its sign bit set: such a buffer takes the general path)
bool parallel_all_zero(const void *p, size_t bytes) {
    std::atomic<bool> zero{true};
    const size_t n = (bytes + kCopyChunk - 1) / kCopyChunk;
    HostPool::get().run(n, [&](size_t i) {
        if (!zero.load(std::memory_order_relaxed)) return;
        const size_t off = i * kCopyChunk, len = std::min(kCopyChunk, bytes - off);
        const unsigned char *b = static_cast<const unsigned char *>(p) + off;
        uint64_t acc = 0;
        size_t k = 0;
        for (; k + 32 <= len; k += 32) {
            uint64_t w[4];
            memcpy(w, b + k, 32);
            acc |= w[0] | w[1] | w[2] | w[3];
        }
        for (; k < len; ++k) acc |= b[k];
        if (acc) zero.store(false, std::memory_order_relaxed);
    });
    return zero.load();
}

// Host buffers the caller registered (pt_buffer_register): pt_render then renders straight into them over PCIe.
struct RegisteredBuffer {
    void *host;
    size_t bytes;
};
std::vector<RegisteredBuffer> g_registered;   // (registration is rare and process-wide; guarded by g_reg_mutex)
std::mutex g_reg_mutex;

bool registered_device_ptr(const void *host, size_t bytes, void **dev_out) {
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (const RegisteredBuffer &r : g_registered) {
        const char *b = static_cast<const char *>(r.host), *p = static_cast<const char *>(host);
        if (p >= b && p + bytes <= b + r.bytes) {
            void *d = nullptr;
            if (hipHostGetDevicePointer(&d, r.host, 0) != hipSuccess || !d) return false;
            *dev_out = static_cast<char *>(d) + (p - b);
            return true;
        }
    }
    return false;
}

int read_ray_count(pt_scene *s, uint64_t *out) {
    unsigned long long rc64 = 0;
    HIP_TRY(hipMemcpy(&rc64, s->d_ray_count, sizeof rc64, hipMemcpyDeviceToHost));   // (synchronises the null stream)
    *out = rc64;
    return PT_OK;
}

}  // namespace

namespace pthostside {
// device frame (pt_scene_prepare's throw-away frame; unpinned fallback of pt_render) + the pinned, mapped copy of the caller's buffer
int ensure_frame_buffers(pt_scene *s, size_t floats) {
    if (floats <= s->frame_floats) return PT_OK;
    (void)hipFree(s->d_frame);
    (void)hipHostFree(s->h_stage);
    s->d_frame = nullptr, s->h_stage = nullptr, s->h_stage_dev = nullptr, s->frame_floats = 0;
    HIP_TRY(hipMalloc((void **)&s->d_frame, floats * sizeof(float)));
    void *dev = nullptr;   // (a failure to pin is not fatal: pt_render then stages through d_frame with plain hipMemcpy)
    if (hipHostMalloc((void **)&s->h_stage, floats * sizeof(float), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
        s->h_stage = nullptr;
    } else if (hipHostGetDevicePointer(&dev, s->h_stage, 0) != hipSuccess || !dev) {
        (void)hipHostFree(s->h_stage);
        s->h_stage = nullptr;
    } else {
        s->h_stage_dev = static_cast<float *>(dev);
    }
    s->frame_floats = floats;
    return PT_OK;
}
}  // namespace pthostside

extern "C" int pt_render_device(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, float *d_rgb_inout, uint64_t *d_ray_count,
                                void *hip_stream) {
    return launch(s, params, cam, frame_num, 0, 1, d_rgb_inout, d_ray_count, reinterpret_cast<hipStream_t>(hip_stream));
}

extern "C" int pt_render_shard_device(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, uint32_t shard_index, uint32_t shard_count,
                                      float *d_rgb_shard_inout, uint64_t *d_ray_count, void *hip_stream) {
    return launch(s, params, cam, frame_num, shard_index, shard_count, d_rgb_shard_inout, d_ray_count, reinterpret_cast<hipStream_t>(hip_stream));
}

extern "C" int pt_scene_prepare(pt_scene *s, const pt_params *params) {
    if (!s || !params) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0) return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    if ((uint64_t)params->width * params->height > 0x3fffffffull) return fail(PT_ERR_INVALID_ARG, "frame too large");
    HIP_TRY(hipSetDevice(s->device));
    if (int rc = ensure_frame_buffers(s, (size_t)params->width * params->height * 3u)) return rc;
    if (params->use_bvh && s->bvh_root < 0) return PT_OK;   // (pt_render will report the missing tree)
    // one throw-away frame with the caller's geometry of launch (samples only scale the work, except that the two-launch frame
    // needs kTwoLaunchMinSamples of them to be scheduled at all): allocates every lazily sized buffer, loads the code objects
    pt_params p = *params;
    p.samples = params->samples >= ptsel::kTwoLaunchMinSamples ? ptsel::kTwoLaunchMinSamples : 1u;
    pt_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.lower_left_corner[0] = cam.lower_left_corner[1] = cam.lower_left_corner[2] = -1.0f;
    cam.horizontal[0] = 2.0f;
    cam.vertical[1] = 2.0f;
    HIP_TRY(hipMemsetAsync(s->d_frame, 0, (size_t)params->width * params->height * 3u * sizeof(float), nullptr));
    if (int rc = launch(s, &p, &cam, 0, 0, 1, s->d_frame, reinterpret_cast<uint64_t *>(s->d_ray_count), nullptr)) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    s->ev_valid = false;
    s->hint_valid = false;
    HostPool::get().run(16, [](size_t) {});   // starts the copy helpers now rather than inside the first pt_render
    return PT_OK;
}

extern "C" int pt_buffer_register(void *host_ptr, size_t bytes) {
    if (!host_ptr || bytes == 0) return fail(PT_ERR_INVALID_ARG, "NULL buffer / zero size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device available");
    HIP_TRY(hipHostRegister(host_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    g_registered.push_back(RegisteredBuffer{host_ptr, bytes});
    return PT_OK;
}

extern "C" int pt_buffer_unregister(void *host_ptr) {
    if (!host_ptr) return fail(PT_ERR_INVALID_ARG, "NULL buffer");
    {
        std::lock_guard<std::mutex> lock(g_reg_mutex);
        size_t i = 0;
        while (i < g_registered.size() && g_registered[i].host != host_ptr) ++i;
        if (i == g_registered.size()) return fail(PT_ERR_INVALID_ARG, "buffer was not registered with pt_buffer_register");
        g_registered.erase(g_registered.begin() + (long)i);
    }
    HIP_TRY(hipHostUnregister(host_ptr));
    return PT_OK;
}

extern "C" int pt_render(pt_scene *s, const pt_params *params, const pt_camera *cam, uint32_t frame_num, float *rgb_inout, uint64_t *ray_count_out) {
    if (!s || !params || !cam || !rgb_inout || !ray_count_out) return fail(PT_ERR_INVALID_ARG, "NULL argument");
    if (params->width == 0 || params->height == 0 || params->samples == 0) return fail(PT_ERR_INVALID_ARG, "width/height/samples must be non-zero");
    if ((uint64_t)params->width * params->height > 0x3fffffffull) return fail(PT_ERR_INVALID_ARG, "frame too large");
    HIP_TRY(hipSetDevice(s->device));
    const size_t floats = (size_t)params->width * params->height * 3u, bytes = floats * sizeof(float);
    uint64_t *const d_rays = reinterpret_cast<uint64_t *>(s->d_ray_count);
    void *mapped = nullptr;
    if (registered_device_ptr(rgb_inout, bytes, &mapped)) {
        // registered (pinned + mapped) caller buffer: the kernel reads the previous frame and writes the new one in place,
        // pixel by pixel as lanes finish them -- the transfers ride under the render, nothing is staged or copied afterwards
        if (int rc = launch(s, params, cam, frame_num, 0, 1, static_cast<float *>(mapped), d_rays, nullptr)) return rc;
        return read_ray_count(s, ray_count_out);
    }
    if (int rc = ensure_frame_buffers(s, floats)) return rc;
    if (s->h_stage) {
        // pageable caller buffer: render into its pinned + mapped copy; the copy-in (or the scan that shows it unnecessary)
        // runs on the host while the GPU executes the measuring launch
        typedef std::chrono::steady_clock Clock;
        const auto ms = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
        const Clock::time_point t0 = Clock::now();
        Clock::time_point t_in0 = t0, t_in1 = t0;
        const BeforeFrame copy_in = [&](bool *prev_zero) -> int {
            t_in0 = Clock::now();
            *prev_zero = parallel_all_zero(rgb_inout, bytes);
            if (!*prev_zero) parallel_copy(s->h_stage, rgb_inout, bytes);
            t_in1 = Clock::now();
            return PT_OK;
        };
        if (int rc = launch(s, params, cam, frame_num, 0, 1, s->h_stage_dev, d_rays, nullptr, &copy_in)) return rc;
        const Clock::time_point t1 = Clock::now();
        if (int rc = read_ray_count(s, ray_count_out)) return rc;   // (the kernel has finished: its PCIe writes are visible)
        const Clock::time_point t2 = Clock::now();
        parallel_copy(rgb_inout, s->h_stage, bytes);
        const Clock::time_point t3 = Clock::now();
        s->host_ms[0] = ms(t_in0, t_in1), s->host_ms[1] = ms(t1, t2), s->host_ms[2] = ms(t2, t3), s->host_ms[3] = ms(t0, t3);
        return PT_OK;
    }
    // no pinned memory to be had: stage through the device frame with synchronous copies
    HIP_TRY(hipMemcpy(s->d_frame, rgb_inout, bytes, hipMemcpyHostToDevice));
    if (int rc = launch(s, params, cam, frame_num, 0, 1, s->d_frame, d_rays, nullptr)) return rc;
    HIP_TRY(hipMemcpy(rgb_inout, s->d_frame, bytes, hipMemcpyDeviceToHost));
    return read_ray_count(s, ray_count_out);
}
