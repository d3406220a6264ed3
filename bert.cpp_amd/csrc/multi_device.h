// multi_device.h — one bert_ctx over several GPUs of a node: the sentences of a call are independent
// (the reference evaluates them in a sequential loop, bert.cpp:750), so a call is cut into contiguous shards with
// near-equal token counts, one per device; every device holds a replica of the weights and evaluates its shard on its
// own host thread and stream.  Results go straight to the caller's host rows (bert.h API), or stay on the devices and
// are exchanged by ONE RCCL step so that every device holds the whole [n_sentences][n_embd] matrix (bert_hip.h API).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace bert_hip {

// bounds[r] .. bounds[r+1]: sentences of shard r (n_shards + 1 entries, bounds[0] = 0, bounds[n_shards] = n_sentences).
// Shards are contiguous, in order, balanced by TOKEN count; a shard may be empty.  Same rule as bert.cpp_amd/dist.py.
void shard_bounds(const int32_t *cu_seqlens, int n_sentences, int n_shards, std::vector<int> &bounds);

// The host threads of a multi-device context: worker r - 1 serves shard r (device r) for the life of the context, shard 0
// runs on the calling thread.  Threads are created ONCE (a forward pass takes under a millisecond: creating and joining
// threads per call is a visible fraction of an 8-GPU step); a worker that cannot be started is not an error, its shard runs on
// the caller.  No exception leaves a worker: it becomes a non-zero result and a message.
class ShardWorkers {
public:
    explicit ShardWorkers(int n_workers);
    ~ShardWorkers();
    ShardWorkers(const ShardWorkers &) = delete;
    ShardWorkers &operator=(const ShardWorkers &) = delete;
    int n_threads() const { return (int)threads_.size(); }
    // eval(shard, first, last) for every non-empty shard.  Returns 0, or the first (lowest shard) non-zero result; *err
    // receives the message of an exception thrown by eval (result -9).  Calls are serialised by the caller (a bert_ctx is not
    // thread-safe, like the reference's).
    int run(const std::vector<int> &bounds, const std::function<int(int, int, int)> &eval, std::string *err = nullptr);
    // threads this process has created for shard work so far (test hook: a thousand calls must not create a thousand threads)
    static long threads_created();

private:
    struct Impl;
    Impl *impl_;
    std::vector<void *> threads_;       // (std::thread objects live in Impl; this only counts them)
};

// RCCL (librccl.so, loaded on first use: libbert.so has no link-time dependency on it), one communicator per device.
class RcclGather {
public:
    ~RcclGather();
    bool init(const std::vector<int> &devices, std::string &err);
    bool ready() const { return !comms_.empty(); }
    // src[r]: shard r on device r ((bounds[r+1] - bounds[r]) * H floats); dst[d]: [n_sentences][H] on device d.  The path's
    // ONE exchange step, on the devices' streams, not synchronised: one ncclAllGather per device when the shards are equal
    // (every fixed-length batch), a grouped broadcast per shard otherwise.
    // all_gather: every device's call issued by the calling thread (one group); exchange_on: device d's call only, to be
    // issued by the host thread that serves device d (ShardWorkers) — the n threads' calls meet inside RCCL.
    bool all_gather(float *const *src, float *const *dst, const std::vector<int> &bounds, int H, hipStream_t *streams,
                    std::string &err);
    bool exchange_on(int d, const float *src, float *dst, const std::vector<int> &bounds, int H, hipStream_t stream, std::string &err);
    static bool equal_shards(const std::vector<int> &bounds);

private:
    int issue(int d, const float *src, float *dst, const std::vector<int> &bounds, int H, hipStream_t stream, bool grouped);
    void *lib_ = nullptr;
    std::vector<void *> comms_;
    std::vector<int> devices_;
    void *fn_[8] = {};
};

}  // namespace bert_hip
