// multi_device.cpp — see multi_device.h.
#include "multi_device.h"

#include <dlfcn.h>

#include <cmath>
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

int dispatch_shards(const std::vector<int> &bounds, const std::function<int(int, int, int)> &eval) {
    const int n_shards = (int)bounds.size() - 1;
    std::vector<int> rc((size_t)n_shards, 0);
    std::vector<std::thread> pool;
    for (int r = 1; r < n_shards; ++r)
        if (bounds[r + 1] > bounds[r]) pool.emplace_back([&, r] { rc[r] = eval(r, bounds[r], bounds[r + 1]); });
    if (n_shards > 0 && bounds[1] > bounds[0]) rc[0] = eval(0, bounds[0], bounds[1]);
    for (auto &t : pool) t.join();
    for (int r = 0; r < n_shards; ++r)
        if (rc[r]) return rc[r];
    return 0;
}

// ---- RCCL through dlopen: the handful of entry points the exchange needs (NCCL C API, stable ABI)
namespace {
typedef int (*fn_comm_init_all)(void **comms, int ndev, const int *devlist);
typedef int (*fn_comm_destroy)(void *comm);
typedef int (*fn_group)(void);
typedef int (*fn_broadcast)(const void *send, void *recv, size_t count, int dtype, int root, void *comm, hipStream_t s);
typedef const char *(*fn_err)(int);
constexpr int NCCL_FLOAT32 = 7;
enum { F_INIT, F_DESTROY, F_GSTART, F_GEND, F_BCAST, F_ERR };
}  // namespace

RcclGather::~RcclGather() {
    if (fn_[F_DESTROY])
        for (void *c : comms_) ((fn_comm_destroy)fn_[F_DESTROY])(c);
    if (lib_) dlclose(lib_);
}

bool RcclGather::init(const std::vector<int> &devices, std::string &err) {
    if (ready()) return true;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        lib_ = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (lib_) break;
    }
    if (!lib_) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
    const char *names[] = {"ncclCommInitAll", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclBroadcast", "ncclGetErrorString"};
    for (int i = 0; i < 6; ++i) {
        fn_[i] = dlsym(lib_, names[i]);
        if (!fn_[i]) { err = std::string("librccl.so lacks ") + names[i]; return false; }
    }
    devices_ = devices;
    comms_.assign(devices.size(), nullptr);
    const int rc = ((fn_comm_init_all)fn_[F_INIT])(comms_.data(), (int)devices.size(), devices.data());
    if (rc != 0) {
        err = std::string("ncclCommInitAll: ") + ((fn_err)fn_[F_ERR])(rc);
        comms_.clear();
        return false;
    }
    return true;
}

bool RcclGather::all_gather(float *const *src, float *const *dst, const std::vector<int> &bounds, int H, hipStream_t *streams,
                            std::string &err) {
    const int n = (int)comms_.size();
    if ((int)bounds.size() != n + 1) { err = "RcclGather: shard count does not match the communicator"; return false; }
    int rc = ((fn_group)fn_[F_GSTART])();
    for (int root = 0; rc == 0 && root < n; ++root) {
        const size_t count = (size_t)(bounds[root + 1] - bounds[root]) * H;
        if (!count) continue;
        for (int d = 0; rc == 0 && d < n; ++d) {
            (void)hipSetDevice(devices_[d]);
            rc = ((fn_broadcast)fn_[F_BCAST])(d == root ? src[root] : nullptr, dst[d] + (size_t)bounds[root] * H, count, NCCL_FLOAT32,
                                             root, comms_[d], streams[d]);
        }
    }
    const int rc2 = ((fn_group)fn_[F_GEND])();
    if (rc == 0) rc = rc2;
    if (rc != 0) { err = std::string("RCCL exchange: ") + ((fn_err)fn_[F_ERR])(rc); return false; }
    return true;
}

}  // namespace bert_hip
