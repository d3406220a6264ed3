// multi_device.cpp — see multi_device.h.
#include "multi_device.h"

#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>

namespace bert_hip {

void shard_bounds(const int32_t *cu, int n, int n_shards, std::vector<int> &bounds) {
    bounds.assign((size_t)n_shards + 1, n);
    bounds[0] = 0;
    const long long total = n > 0 ? (long long)cu[n] - cu[0] : 0;
    int start = 0;
    for (int r = 0; r + 1 < n_shards; ++r) {
        // first sentence boundary whose token offset reaches total * (r+1) / n_shards, or the one before it if that is closer
        const double target = (double)total * (r + 1) / n_shards;
        int lo = 0, hi = n;                                   // smallest end with cu[end] - cu[0] >= target
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if ((double)((long long)cu[mid] - cu[0]) >= target) hi = mid; else lo = mid + 1;
        }
        int end = lo;
        if (end > 0 && std::fabs((double)((long long)cu[end - 1] - cu[0]) - target) <= std::fabs((double)((long long)cu[end < n ? end : n] - cu[0]) - target))
            --end;
        end = end < start ? start : (end > n ? n : end);
        bounds[r + 1] = end;
        start = end;
    }
}

// ---- persistent shard workers
namespace {
std::atomic<long> g_threads_created{0};

// guards one call of eval: no exception crosses a thread boundary (or, on the caller's thread, the C ABI one level up)
int guarded_eval(const std::function<int(int, int, int)> &eval, int r, int b0, int b1, std::string &msg) {
    try {
        return eval(r, b0, b1);
    } catch (const std::exception &e) {
        msg = e.what();
    } catch (...) {
        msg = "unknown exception";
    }
    return -9;
}
}  // namespace

struct ShardWorkers::Impl {
    struct Slot {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        const std::function<int(int, int, int)> *job = nullptr;     // non-null: work to do
        int r = 0, b0 = 0, b1 = 0, rc = 0;
        bool done = false, quit = false;
        std::string msg;
    };
    std::vector<std::unique_ptr<Slot>> slots;

    static void loop(Slot *s) {
        std::unique_lock<std::mutex> lk(s->m);
        for (;;) {
            s->cv.wait(lk, [&] { return s->job || s->quit; });
            if (s->quit) return;
            const auto *job = s->job;
            lk.unlock();
            std::string msg;
            const int rc = guarded_eval(*job, s->r, s->b0, s->b1, msg);
            lk.lock();
            s->rc = rc;
            s->msg = std::move(msg);
            s->job = nullptr;
            s->done = true;
            s->cv.notify_all();
        }
    }
};

ShardWorkers::ShardWorkers(int n_workers) : impl_(nullptr) {
    try {
        impl_ = new Impl;
        for (int k = 0; k < n_workers; ++k) {
            std::unique_ptr<Impl::Slot> s(new Impl::Slot);
            s->th = std::thread(Impl::loop, s.get());         // std::system_error: stop here, the rest run on the caller
            g_threads_created.fetch_add(1);
            impl_->slots.push_back(std::move(s));
            threads_.push_back(nullptr);
        }
    } catch (...) {
    }
}

ShardWorkers::~ShardWorkers() {
    if (!impl_) return;
    for (auto &s : impl_->slots) {
        { std::lock_guard<std::mutex> lk(s->m); s->quit = true; }
        s->cv.notify_all();
        if (s->th.joinable()) s->th.join();
    }
    delete impl_;
}

long ShardWorkers::threads_created() { return g_threads_created.load(); }

int ShardWorkers::run(const std::vector<int> &bounds, const std::function<int(int, int, int)> &eval, std::string *err) {
    const int n_shards = (int)bounds.size() - 1, n_workers = impl_ ? (int)impl_->slots.size() : 0;
    std::vector<int> rc((size_t)(n_shards > 0 ? n_shards : 0), 0);
    std::vector<std::string> msgs(rc.size());
    // hand shard r to worker r - 1 ...
    for (int r = 1; r < n_shards && r - 1 < n_workers; ++r) {
        if (bounds[r + 1] <= bounds[r]) continue;
        Impl::Slot &s = *impl_->slots[r - 1];
        std::lock_guard<std::mutex> lk(s.m);
        s.r = r; s.b0 = bounds[r]; s.b1 = bounds[r + 1]; s.done = false; s.rc = 0; s.msg.clear();
        s.job = &eval;
        s.cv.notify_all();
    }
    // ... shard 0 and every shard without a worker run here ...
    for (int r = 0; r < n_shards; ++r)
        if ((r == 0 || r - 1 >= n_workers) && bounds[r + 1] > bounds[r]) rc[r] = guarded_eval(eval, r, bounds[r], bounds[r + 1], msgs[r]);
    // ... and every worker is waited for before anything is returned (or rethrown): its job refers to the caller's frame
    for (int r = 1; r < n_shards && r - 1 < n_workers; ++r) {
        if (bounds[r + 1] <= bounds[r]) continue;
        Impl::Slot &s = *impl_->slots[r - 1];
        std::unique_lock<std::mutex> lk(s.m);
        s.cv.wait(lk, [&] { return s.done; });
        rc[r] = s.rc;
        msgs[r] = s.msg;
    }
    for (int r = 0; r < n_shards; ++r)
        if (rc[r]) {
            if (err && !msgs[r].empty()) *err = msgs[r];
            return rc[r];
        }
    return 0;
}

// ---- RCCL through dlopen: the handful of entry points the exchange needs (NCCL C API, stable ABI)
namespace {
typedef int (*fn_comm_init_all)(void **comms, int ndev, const int *devlist);
typedef int (*fn_comm_destroy)(void *comm);
typedef int (*fn_group)(void);
typedef int (*fn_broadcast)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t s);
typedef int (*fn_allgather)(const void *send, void *recv, size_t sendcount, int dtype, void *comm, hipStream_t s);
typedef const char *(*fn_err)(int);
constexpr int NCCL_FLOAT32 = 7;
enum { F_INIT, F_DESTROY, F_GSTART, F_GEND, F_BCAST, F_ERR, F_ALLGATHER };
}  // namespace

RcclGather::~RcclGather() {
    if (fn_[F_DESTROY])
        for (void *c : comms_) ((fn_comm_destroy)fn_[F_DESTROY])(c);
    if (lib_) dlclose(lib_);
}

bool RcclGather::init(const std::vector<int> &devices, std::string &err) {
    if (ready()) return true;
    for (size_t i = 0; i < devices.size(); ++i)
        for (size_t j = 0; j < i; ++j)
            if (devices[i] == devices[j]) { err = "RCCL communicator: device " + std::to_string(devices[i]) + " is listed twice"; return false; }
    auto fail = [&]() {                                       // nothing half-initialised stays behind (a later call starts over)
        comms_.clear();
        if (lib_) { dlclose(lib_); lib_ = nullptr; }
        for (auto &f : fn_) f = nullptr;
        return false;
    };
    // The RCCL that belongs to the HIP runtime this process runs on: the one next to the loaded libamdhip64.  A process can
    // hold two (a PyTorch wheel brings its own ROCm libraries; "librccl.so" by name then resolves to whichever was loaded
    // first, and a librccl of another ROCm release on this runtime fails in ncclCommInitAll with "unhandled cuda error").
    std::vector<std::string> candidates;
    Dl_info hip_lib;
    if (dladdr((const void *)&hipGetDeviceCount, &hip_lib) && hip_lib.dli_fname) {
        const std::string path = hip_lib.dli_fname;
        const size_t slash = path.rfind('/');
        if (slash != std::string::npos) {
            candidates.push_back(path.substr(0, slash) + "/librccl.so");
            candidates.push_back(path.substr(0, slash) + "/librccl.so.1");
        }
    }
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) candidates.push_back(name);
    for (const std::string &name : candidates) {
        lib_ = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (lib_) break;
    }
    if (!lib_) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
    const char *names[] = {"ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclBroadcast", "ncclGetErrorString", "ncclAllGather"};
    for (int i = 0; i < 7; ++i) {
        fn_[i] = dlsym(lib_, names[i]);
        if (!fn_[i]) { err = std::string("librccl.so lacks ") + names[i]; return fail(); }
    }
    devices_ = devices;
    comms_.assign(devices.size(), nullptr);
    const int rc = ((fn_comm_init_all)fn_[F_INIT])(comms_.data(), (int)devices.size(), devices.data());
    if (rc != 0) {
        err = std::string("ncclCommInitAll: ") + ((fn_err)fn_[F_ERR])(rc);
        return fail();
    }
    return true;
}

bool RcclGather::equal_shards(const std::vector<int> &bounds) {
    const int n = (int)bounds.size() - 1;
    if (n < 1 || bounds[1] - bounds[0] <= 0) return false;
    for (int r = 1; r < n; ++r)
        if (bounds[r + 1] - bounds[r] != bounds[1] - bounds[0]) return false;
    return true;
}

// rank d's part of the exchange, on the calling thread's current device (the caller has selected devices_[d]).  Equal shards
// (every BASELINE config: fixed-length batches cut by token count): ONE ncclAllGather — rank r's shard lands at row
// bounds[r] = r * count of every device's matrix.  Unequal shards: a broadcast per shard, grouped (non-roots pass their own
// destination as the unused send buffer: a null pointer does not survive NCCL_CHECK_POINTERS).
int RcclGather::issue(int d, const float *src, float *dst, const std::vector<int> &bounds, int H, hipStream_t stream, bool grouped) {
    const int n = (int)comms_.size();
    if (equal_shards(bounds))
        return ((fn_allgather)fn_[F_ALLGATHER])(src, dst, (size_t)(bounds[1] - bounds[0]) * H, NCCL_FLOAT32, comms_[d], stream);
    int rc = grouped ? ((fn_group)fn_[F_GSTART])() : 0;
    for (int root = 0; rc == 0 && root < n; ++root) {
        const size_t count = (size_t)(bounds[root + 1] - bounds[root]) * H;
        if (!count) continue;
        float *slot = dst + (size_t)bounds[root] * H;
        rc = ((fn_broadcast)fn_[F_BCAST])(d == root ? src : slot, slot, count, NCCL_FLOAT32, root, comms_[d], stream);
    }
    if (grouped) {
        const int rc2 = ((fn_group)fn_[F_GEND])();
        if (rc == 0) rc = rc2;
    }
    return rc;
}

bool RcclGather::all_gather(float *const *src, float *const *dst, const std::vector<int> &bounds, int H, hipStream_t *streams,
                            std::string &err) {
    const int n = (int)comms_.size();
    if ((int)bounds.size() != n + 1) { err = "RcclGather: shard count does not match the communicator"; return false; }
    // (what can fail outside RCCL fails before the group is opened: a group closed over some ranks' calls only leaves those
    // ranks in a collective their peers never join)
    for (int d = 0; d < n; ++d)
        if (hipSetDevice(devices_[d]) != hipSuccess) { err = "RCCL exchange: hipSetDevice failed"; return false; }
    // one host thread drives every device's communicator: the calls of the n ranks form ONE group
    int rc = ((fn_group)fn_[F_GSTART])();
    for (int d = 0; rc == 0 && d < n; ++d) {
        (void)hipSetDevice(devices_[d]);
        rc = issue(d, src[d], dst[d], bounds, H, streams[d], false);
    }
    const int rc2 = ((fn_group)fn_[F_GEND])();
    if (rc == 0) rc = rc2;
    if (rc != 0) { err = std::string("RCCL exchange: ") + ((fn_err)fn_[F_ERR])(rc); return false; }
    return true;
}

bool RcclGather::exchange_on(int d, const float *src, float *dst, const std::vector<int> &bounds, int H, hipStream_t stream,
                             std::string &err) {
    const int n = (int)comms_.size();
    if ((int)bounds.size() != n + 1 || d < 0 || d >= n) { err = "RcclGather: shard count does not match the communicator"; return false; }
    if (hipSetDevice(devices_[d]) != hipSuccess) { err = "RCCL exchange: hipSetDevice failed"; return false; }
    const int rc = issue(d, src, dst, bounds, H, stream, true);
    if (rc != 0) { err = std::string("RCCL exchange: ") + ((fn_err)fn_[F_ERR])(rc); return false; }
    return true;
}

}  // namespace bert_hip
