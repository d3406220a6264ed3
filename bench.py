#!/usr/bin/env python3
"""bench.py — sentences/sec of the bert_eval hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (bert_hip_eval_packed_device: embeddings -> L encoder
layers -> mean-pool + L2) over one batch of synthetic token ids that is already resident in HBM;
with --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) each rank evaluates its own
shard of sentences (weights replicated, no data-path collective inside the forward pass) and the step
ends with ONE RCCL all-gather of the final embeddings over xGMI, as BASELINE.json's north_star
describes.  value = sentences all ranks processed / max-over-ranks wall time.

    python bench.py --gpus 1 --steps 100 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --inproc          # ONE process, N GPUs inside libbert.so (BERT_HIP_DEVICES, RCCL gather)

Prints ONE JSON line on rank 0 (contract in the task description).  The timed region of K steps is
repeated (--repeat, default 5): `value` / `ms_per_step` are the MEDIAN region, `regions` lists them all
(the boxes of the pool and the chip's power management spread single regions by several percent).
Extra objects:
  roofline     — dominant kernel, algorithmic FLOPs per launch / HIP-event launch duration vs the
                 dense f16 MFMA peak (2.5 PFLOP/s); traffic = HBM bytes per launch from the committed
                 rocprofv3 PMC pass (profiles/traffic.json);
  host_api     — the same batch through the host-to-host API (bert_hip_eval_packed = what
                 bert_eval_batch runs after packing its pointer arrays: pinned staging, H2D ids, forward,
                 D2H embeddings, blocking): SURVEY.md §8(d)'s metric as the reference's callers see it;
  cpu_baseline — the CPU oracle in ggml-faithful mode (kind "port": the reference itself cannot be
                 built, its arithmetic lives in the un-vendored ggml submodule) timed on this box's
                 host cores over a bounded sample of the same sentences;
  also         — the other single-GPU BASELINE configs (2 as written and with the engine default, 3) and a
                 real-text-like mixed-length batch (ids in HBM, and host to host), each with its own roofline /
                 cosine / cpu sample.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("BERT_HIP_QUIET", "1")

from bert_cpp_amd import dist as bdist  # noqa: E402
from bert_cpp_amd import ggml_file as gf  # noqa: E402
from bert_cpp_amd import pybert  # noqa: E402

MFMA_PEAK_F16 = 2.5e15      # dense, /opt/skills/guides/MI355X_MICROARCH.md §Matrix cores
# What the matrix cores sustain on RANDOM f16 operands under the board's power limit: v_mfma_f32_32x32x16_f16 back to back from
# registers, nothing else running (tools/ubench/mfma_power.hip, profiles/r2_mfma_power.txt: 1644 TFLOP/s; 2465 on zeros).
MFMA_RATE_RANDOM_F16 = 1.644e15

# BASELINE.json configs[1..4]; configs[0] is the CPU-only plumbing case (= the cpu_baseline leg).
CONFIGS = {
    1: dict(name="all-MiniLM-L6-v2 f16, batch=256 seq_len=128", dims="minilm-l6", ftype="f16", batch=256, seq_len=128),
    # configs[2] says "(fused dequant-GEMM)": measured literally with BERT_HIP_Q4=fused (4-bit planes in HBM, dequant in the
    # GEMM tile load); the engine's default expands q4 matrices to f16 once at load (same values) and is reported
    # next to it as config2_expanded
    2: dict(name="all-MiniLM-L6-v2 q4_0, batch=1024 seq_len=128 (fused dequant-GEMM)", dims="minilm-l6", ftype="q4_0", batch=1024,
            seq_len=128, env={"BERT_HIP_Q4": "fused"}),
    22: dict(name="all-MiniLM-L6-v2 q4_0, batch=1024 seq_len=128 (q4 matrices expanded to f16 at load: engine default)",
             dims="minilm-l6", ftype="q4_0", batch=1024, seq_len=128, key="config2_expanded"),
    3: dict(name="bert-base-uncased q4_1, batch=512 seq_len=512", dims="bert-base", ftype="q4_1", batch=512, seq_len=512),
    # configs[3] / [4] with the weights 4-bit in HBM (north_star: "q4 block dequant fused into the GEMM tile load"): gemm256's q4 loader
    33: dict(name="bert-base-uncased q4_1, batch=512 seq_len=512 (4-bit planes in HBM, dequantised in gemm256's tile load)", dims="bert-base",
             ftype="q4_1", batch=512, seq_len=512, env={"BERT_HIP_Q4": "fused"}, key="config3_fused"),
    4: dict(name="mpnet-base dims (BERT arch) q4_0, seq_len=128, 8192-sentence steps", dims="mpnet-dims", ftype="q4_0",
            batch=8192, seq_len=128),
    42: dict(name="mpnet-base dims (BERT arch) q4_0, seq_len=128, 8192-sentence steps (4-bit planes in HBM, dequantised in gemm256's tile load)",
             dims="mpnet-dims", ftype="q4_0", batch=8192, seq_len=128, env={"BERT_HIP_Q4": "fused"}, key="config4_fused"),
    # BASELINE configs[4] as SURVEY.md §8(d) writes it ("Config 5"): 1,000,000 sentences over 8 GPUs = 125,000 per GPU.  ONE GPU's
    # share through ONE host-to-host call (ids in host memory -> embeddings in host memory: 61 chunks through the two-slot staging
    # pipeline, 64 MB in, 384 MB out), cosine on a fixed 256-sentence sample; 45: the same call through the gather entry point
    # (results stay in HBM, the path's RCCL exchange step runs on a 1-rank communicator)
    44: dict(name="mpnet-base dims (BERT arch) q4_0: one GPU's share of 1M sentences, ONE host-to-host call of 125,000 x 128 tokens",
             dims="mpnet-dims", ftype="q4_0", batch=125000, seq_len=128, key="config4_share", host_step=True, share=True),
    45: dict(name="mpnet-base dims (BERT arch) q4_0: one GPU's share of 1M sentences, ONE bert_hip_eval_packed_gather call of 125,000 x 128 tokens "
                  "(device-resident result + RCCL exchange step, 1 rank)", dims="mpnet-dims", ftype="q4_0", batch=125000, seq_len=128,
             key="config4_share_gather", gather_step=True, share=True, seed_id=44),
    # not a BASELINE config: sentence lengths like real text (reference examples/sample_client_texts.txt: ~22 words per line)
    # 5: inputs resident in HBM like every other config (the engine packs the sentences into the 128-slot windows of the fused
    # attention kernel with a kernel of its own); 55: the same batch host to host through bert_hip_eval_packed
    5: dict(name="all-MiniLM-L6-v2 f16, 16384 sentences of mixed length (log-normal, mean ~25 tokens, 3..128)", dims="minilm-l6",
            ftype="f16", batch=16384, seq_len=None, key="mixed_len"),
    55: dict(name="all-MiniLM-L6-v2 f16, 16384 sentences of mixed length (log-normal, mean ~25 tokens, 3..128), host API", dims="minilm-l6",
             ftype="f16", batch=16384, seq_len=None, key="mixed_len_host_api", host_step=True),
    # the same batch with the opt-in 8-slot places of the attention windows (BERT_HIP_WINDOW_SLOTS=8: fewer windows; a sentence's last
    # bits then depend on its place in its window, which is why it is not the default — DESIGN.md section 3)
    58: dict(name="all-MiniLM-L6-v2 f16, 16384 sentences of mixed length (mean ~25 tokens), BERT_HIP_WINDOW_SLOTS=8 (opt-in)", dims="minilm-l6",
             ftype="f16", batch=16384, seq_len=None, key="mixed_len_slots8", option=("window_slots", "8"), seed_id=5),
}


def flops_per_sentence(hp, n):
    H, I, L = hp.n_embd, hp.n_intermediate, hp.n_layer
    return L * (n * (8 * H * H + 4 * H * I) + 4 * n * n * H) + 2 * n * H


def config_inputs(cfg, cfg_id, hp, rank):
    """(flat ids, cu_seqlens, max_len): seeded synthetic ids of SURVEY.md §8d."""
    B, N = cfg["batch"], cfg["seq_len"]
    if N is not None:
        ids = gf.synthetic_token_ids(B, N, hp.n_vocab, seed=1234 + (cfg.get("seed_id", cfg_id) % 20) + 1000 * rank)
        return ids.reshape(-1), (np.arange(B + 1) * N).astype(np.int32), N
    rng = np.random.default_rng(5 + rank)
    lens = np.clip(np.round(rng.lognormal(np.log(21.0), 0.55, B)), 3, 128).astype(np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = rng.integers(1000, hp.n_vocab, size=int(cu[-1])).astype(np.int32)
    ids[cu[:-1]] = 101
    ids[cu[1:] - 1] = 102
    return ids, cu, int(lens.max())


def load_model(cfg, path):
    saved = {k: os.environ.get(k) for k in cfg.get("env", {})}
    os.environ.update(cfg.get("env", {}))                 # engine options are read when the model is loaded
    try:
        return pybert.BertModel(path)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def timed_regions(step, steps, warmup, repeat, sync, barrier=None, reduce_max=None, warm_step=None):
    """W untimed steps, then `repeat` regions of exactly `steps` steps, each bracketed by barrier + synchronize on both
    sides; per region the max over ranks.  Returns the list of region times (s)."""
    for _ in range(warmup):
        (warm_step or step)()
    out = []
    for _ in range(repeat):
        sync()
        if barrier:
            barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        if barrier:
            barrier()
        sync()
        dt = time.perf_counter() - t0
        out.append(reduce_max(dt) if reduce_max else dt)
    return out


def run_config(cfg_id, args, rank, world, device, dist, torch, tmpdir, steps=None, warmup=None, repeat=None):
    cfg = CONFIGS[cfg_id]
    hp = gf.MODEL_DIMS[cfg["dims"]]
    path = os.path.join(tmpdir, f"{cfg['dims']}_{cfg['ftype']}_rank{rank}.bin")
    if not os.path.exists(path):
        gf.make_synthetic_model(path, cfg["dims"], cfg["ftype"], seed=0)
    model = load_model(cfg, path)
    if cfg.get("option"):
        model.set_option(*cfg["option"])
    B, H = cfg["batch"], hp.n_embd
    flat, cu, max_len = config_inputs(cfg, cfg_id, hp, rank)
    T = int(cu[-1])
    host_side = cfg.get("host_step") or cfg.get("gather_step")       # ids start in host memory: the engine stages them chunk by chunk
    d_tokens = d_cu = d_out = None
    if not host_side:
        d_tokens = torch.from_numpy(flat).to(device)
        d_cu = torch.from_numpy(cu).to(device)
        d_out = torch.empty((B, H), dtype=torch.float32, device=device)
        model.reserve(T, B)
    stream = torch.cuda.current_stream(device)
    counts = [B] * world
    # N > 1: two result buffers and two gathered matrices — the exchange of step k (RCCL's own stream) runs under the forward
    # pass of step k + 1, which writes the other pair; a buffer is reused only after the exchange that read it has finished
    # ("per super-batch, overlapped with the next super-batch's compute").  Everything is finished inside
    # the timed region: it ends with a device-wide synchronize.
    d_outs = [d_out, torch.empty_like(d_out)] if world > 1 and d_out is not None else [d_out]
    d_alls = [torch.empty((world * B, H), dtype=torch.float32, device=device) for _ in range(2)] if world > 1 else None
    works = [None, None]
    turn = [0]

    h_out = np.empty((B, H), dtype=np.float32) if cfg.get("host_step") else None
    if cfg.get("gather_step"):
        model.set_option("test_rccl_single", "1")             # one device: the exchange step still runs (1-rank communicator)
    gathered = {}

    def step():
        if cfg.get("host_step"):
            model.eval_packed(flat, cu, out=h_out)            # (the caller's rows, as bert_eval_batch's `float **batch_embeddings`)
            return
        if cfg.get("gather_step"):
            gathered["ptr"] = model.eval_packed_gather(flat, cu)[0]
            return
        if world == 1:
            model.eval_packed_device(d_tokens.data_ptr(), d_cu.data_ptr(), B, T, max_len, d_out.data_ptr(), stream.cuda_stream)
            return
        i = turn[0] & 1
        turn[0] += 1
        if works[i] is not None:
            works[i].wait()                                   # (orders the launch stream behind the exchange that read d_outs[i])
        model.eval_packed_device(d_tokens.data_ptr(), d_cu.data_ptr(), B, T, max_len, d_outs[i].data_ptr(), stream.cuda_stream)
        # RCCL over xGMI: the path's one exchange step (bert.cpp_amd/dist.py), [world*B, H] on every rank
        works[i] = bdist.gather_embeddings(d_outs[i], counts, out=d_alls[i], async_op=True)[1]

    def reduce_max(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    warm_step = None
    if cfg.get("share"):
        # (a 3 s call as warm-up would double the entry's time: the same entry point on a 4096-sentence prefix — two chunks —
        # sizes the workspace and both staging slots and brings the clocks up)
        nw = min(B, 4096)
        h_warm = np.empty((nw, H), dtype=np.float32)
        if cfg.get("gather_step"):
            warm_step = lambda: model.eval_packed_gather(flat[:cu[nw]], cu[:nw + 1])
        else:
            warm_step = lambda: model.eval_packed(flat[:cu[nw]], cu[:nw + 1], out=h_warm)
    steps = steps or args.steps
    regions = timed_regions(step, steps, warmup if warmup is not None else args.warmup, repeat or args.repeat,
                            lambda: torch.cuda.synchronize(device), dist.barrier if world > 1 else None, reduce_max, warm_step)
    dt = float(np.median(regions))
    for w in works:
        if w is not None:
            w.wait()
    out = torch.from_numpy(h_out) if h_out is not None else d_outs[(turn[0] - 1) & 1 if world > 1 and turn[0] else 0]
    if cfg.get("gather_step"):
        # the gathered matrix lives in the context's device buffer: fetch a copy through torch (hipMemcpy D2H)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        host = np.empty((B, H), dtype=np.float32)
        torch.cuda.synchronize(device)
        assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(gathered["ptr"]), ctypes.c_size_t(host.nbytes), 2) == 0
        out = torch.from_numpy(host)
    res = dict(cfg=cfg, cfg_id=cfg_id, hp=hp, model=model, path=path, flat=flat, cu=cu, out=out, steps=steps, regions=regions, world=world,
               value=world * B * steps / dt, ms_per_step=1e3 * dt / steps, step=step, tokens=T, max_len=max_len)
    return res


def committed_traffic(cfg_key, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC pass of this same command
    (profiles/traffic.json, written by tools/pmc_traffic.py: FETCH_SIZE doubled as the gfx950 note of
    MI355X_MICROARCH.md prescribes, + WRITE_SIZE).  Counters cannot be read from inside the process."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        d = t.get(cfg_key.replace("_host_api", ""), {}).get(kernel)      # (the host-API entry runs the same kernels on the same batch)
        return None if d is None else d["bytes"]
    except (OSError, ValueError, KeyError):
        return None


# kernels that work IN PLACE (x -> x): a repeat of one of their launches runs on its own output
IN_PLACE_KERNELS = {"model_kernel", "layer_tail", "layernorm", "skinny_layernorm"}


def kernel_roofline(res, torch, device, steps=5, sync=None, groups=3):
    """The dominant kernel's roofline entry, measured live on the launch stream with HIP events.
    (1) One pass over `steps` steps with an event pair per launch: which kernel dominates, launches per step, the per-kernel
        breakdown `kernel_ms_per_step_timed_alone_upper_bound`.  A launch timed alone reads LONG — a sub-millisecond kernel that lives on cache-resident
        weights by 5-10 % (model_kernel: 825-866 us timed alone, 780 us in rocprofv3's trace of the same steps) — so these are
        upper bounds and not what the roofline uses.
    (2) `avg_launch_us`.  Replay groups (the engine's "profile_replay" option): K back-to-back repeats of ONE launch of a kernel
        between ONE event pair, median of `groups` groups — the pair's cost is spread over K launches.
        - dominant kernel NOT in place (the GEMMs, attention, qkv_attention2): its own replay groups;
        - dominant kernel IN PLACE (model_kernel, layer_tail: x -> x): repeats would run on their own output, and activations
          pushed through the encoder again and again converge — such operands draw less power and the repeats run 7 % fast
          (measured: 720 us against 780 us in the step).  Its time is the STEP's time (the timed region's ms_per_step: the
          GPU is never idle in it) minus the replay-group times of the step's OTHER kernels (embedding, pooling, window
          kernels: all replay-safe), divided by its launches per step.
        Either way launches_per_step x avg_launch_us cannot exceed ms_per_step; bench.py fails loudly if it does.
    rocprofv3 --kernel-trace --stats of this same command is committed under profiles/ together with the line bench.py printed
    UNDER the profiler: compare like with like — a profiled process runs 2-4 % slower (clocks)."""
    model = res["model"]
    sync = sync or (lambda: torch.cuda.synchronize(device))
    model.profile(True)
    for _ in range(steps):
        res["step"]()
    sync()
    rep = {k: v for k, v in model.profile_report().items() if not k.startswith("family:")}
    if not rep:
        model.profile(False)
        return None, rep
    name, st = max(rep.items(), key=lambda kv: kv[1]["total_ms"])
    pair_avg_s = st["total_ms"] / st["launches"] * 1e-3
    per_step = {k: v["launches"] / steps for k, v in rep.items()}
    flops = st["flops_per_launch"]
    total_ms = sum(v["total_ms"] for v in rep.values())

    def replay_avg(kernel, alone_s):
        K = int(min(50, max(5, 30e-3 / max(alone_s, 1e-6))))
        model.set_option("profile_replay", f"{kernel}:{K}")
        samples = []
        for _ in range(max(1, groups)):
            res["step"]()
            sync()
            r2 = model.profile_report().get(kernel)
            if r2 and r2["launches"]:
                samples.append(r2["total_ms"] / r2["launches"] * 1e-3)
        model.set_option("profile_replay", "")
        return (float(np.median(samples)) if samples else alone_s), K, len(samples)

    out = res.get("out")
    saved = None if out is None else out.clone()
    if name in IN_PLACE_KERNELS:
        others = {k: replay_avg(k, rep[k]["total_ms"] / rep[k]["launches"] * 1e-3)[0] for k in rep if k != name}
        others_ms = sum(per_step[k] * t * 1e3 for k, t in others.items())
        avg_s = (res["ms_per_step"] - others_ms) * 1e-3 / per_step[name]
        timing = (f"in-place kernel: step time ({res['ms_per_step']:.4f} ms) minus the replay-group times of the step's other kernels "
                  f"({', '.join(f'{k} {per_step[k]:g} x {t * 1e6:.1f} us' for k, t in sorted(others.items()))}), over {per_step[name]:g} launches per step")
        if res["cfg"].get("host_step"):
            timing += " — a host-paced step (staging, copies): the GPU idles in it, so this is an UPPER bound of the kernel's time (the device-resident entry of the same batch has the kernel's own)"
        if res.get("world", 1) > 1:
            timing += " — the RCCL all-gather of a step's embeddings runs under the next step's forward pass; what is not hidden of it this difference books on the kernel: an UPPER bound of its time"
    else:
        avg_s, K, n = replay_avg(name, pair_avg_s)
        timing = f"{K} back-to-back launches between one HIP event pair on the launch stream, median of {n} such groups"
    model.profile(False)
    if saved is not None:
        out.copy_(saved)
    error = None
    if per_step[name] * avg_s * 1e3 > res["ms_per_step"] * 1.005 or avg_s <= 0:
        # (a noisy replay group must not cost the line: report the inconsistency and fall back to the launch timed alone,
        # capped by the step — an UPPER bound of the kernel's time, i.e. a lower bound of `achieved`)
        error = (f"inconsistent: {per_step[name]:g} launches x {avg_s * 1e6:.1f} us against ms_per_step {res['ms_per_step']:.4f}; "
                 f"fell back to min(timed alone, step / launches)")
        print(f"bench.py: roofline {name}: {error}", file=sys.stderr)
        # (clamped to something positive: a replay group that reported nothing must not turn into a division by zero and take the line)
        step_bound_s = res["ms_per_step"] * 1e-3 / per_step[name]
        avg_s = max(min(pair_avg_s if pair_avg_s > 0 else step_bound_s, step_bound_s), 1e-9)
    achieved = flops / avg_s
    key = res["cfg"].get("key", f"config{res['cfg_id']}")
    roof = {"bound": "mfma", "kernel": name, "achieved": achieved / 1e12, "peak": MFMA_PEAK_F16 / 1e12,
            "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_F16, "traffic": committed_traffic(key, name),
            "mfma_rate_under_power_limit": MFMA_RATE_RANDOM_F16 / 1e12, "frac_of_that": achieved / MFMA_RATE_RANDOM_F16,
            "avg_launch_us": avg_s * 1e6, "flops_per_launch": flops, "launches_per_step": per_step[name],
            "timing": timing, "avg_launch_us_timed_alone": pair_avg_s * 1e6,
            "step_share": per_step[name] * avg_s * 1e3 / res["ms_per_step"],
            "kernel_time_share": st["total_ms"] / total_ms if total_ms else None,
            "derived": name in IN_PLACE_KERNELS}
    if error:
        roof["error"] = error
        roof["frac_is_a_bound"] = True          # (frac comes from an upper bound of the kernel's time: a LOWER bound of the real fraction, not a measurement)
    breakdown = {k: round(v["total_ms"] / steps, 4) for k, v in sorted(rep.items())}
    return roof, breakdown


def host_api_rate(res, calls=None):
    """Host-to-host: ids in host memory -> embeddings in host memory through bert_hip_eval_packed (blocking): SURVEY.md
    §8(d)'s metric as the reference's callers see it (bert_eval_batch = this after packing its pointer arrays)."""
    m, flat, cu = res["model"], res["flat"], res["cu"]
    B = len(cu) - 1
    calls = calls or max(5, min(200, int(0.5 / max(res["ms_per_step"] * 1e-3, 1e-4))))
    out = np.empty((B, res["hp"].n_embd), dtype=np.float32)       # the caller's rows (bert.h: `float **batch_embeddings`), reused like a C caller's
    for _ in range(2):
        m.eval_packed(flat, cu, out=out)
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        m.eval_packed(flat, cu, out=out)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": B / med, "unit": "sentences/s", "ms_per_call": 1e3 * med, "p05": B / float(np.percentile(ts, 95)),
            "p95": B / float(np.percentile(ts, 5)), "min": B / float(np.max(ts)), "max": B / float(np.min(ts)), "calls": calls,
            "entry": "bert_hip_eval_packed (host ids -> host embeddings: pinned staging, one H2D copy, forward, rows written into pinned host memory, blocking)"}, out


def eval_batch_api_rate(res, calls=None):
    """The literal SURVEY.md §8(d) entry point: bert_eval_batch (reference bert.cpp:730-749) with an array of per-sentence
    host pointers in and an array of per-sentence row pointers out, each sentence's ids in an allocation of its own.  The
    pointer arrays are built once outside the timed calls, as a C caller's are."""
    import ctypes as C
    m, flat, cu = res["model"], res["flat"], res["cu"]
    B, H = len(cu) - 1, res["hp"].n_embd
    calls = calls or max(5, min(200, int(0.5 / max(res["ms_per_step"] * 1e-3, 1e-4))))
    i32p, f32p = C.POINTER(C.c_int32), C.POINTER(C.c_float)
    sents = [np.array(flat[cu[i]:cu[i + 1]], dtype=np.int32) for i in range(B)]              # (copies: B separate allocations)
    lens = np.diff(cu).astype(np.int32)
    out = np.full((B, H), np.nan, dtype=np.float32)
    tok_ptrs = (i32p * B)(*[a.ctypes.data_as(i32p) for a in sents])
    out_ptrs = (f32p * B)(*[C.cast(out[i].ctypes.data, f32p) for i in range(B)])
    call = lambda: m.lib.bert_eval_batch(m.ctx, 6, B, tok_ptrs, lens.ctypes.data_as(i32p), out_ptrs)
    for _ in range(2):
        call()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": B / med, "unit": "sentences/s", "ms_per_call": 1e3 * med, "calls": calls,
            "entry": "bert_eval_batch (n_threads 6, B per-sentence token pointers, B row pointers), blocking"}, out


def latency_b1(tmpdir, calls=200):
    """One sentence per call, host to host (bert_hip_eval_packed with n_sentences = 1: what bert_encode / the reference's
    server loop do per request, reference bert.cpp:943-950, examples/server.cpp:98-114): median microseconds of `calls`
    calls, and the per-kernel HIP-event times of the same call."""
    out = {}
    for ftype in ("f16", "q4_0"):
        path = os.path.join(tmpdir, f"minilm-l6_{ftype}_rank0.bin")
        if not os.path.exists(path):
            gf.make_synthetic_model(path, "minilm-l6", ftype, seed=0)
        m = pybert.BertModel(path)
        hp = gf.MODEL_DIMS["minilm-l6"]
        for n in (128, 25):
            ids = gf.synthetic_token_ids(1, n, hp.n_vocab, seed=77).reshape(-1)
            cu = np.array([0, n], dtype=np.int32)
            for _ in range(10):
                m.eval_packed(ids, cu)
            ts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                m.eval_packed(ids, cu)
                ts.append(time.perf_counter() - t0)
            m.profile(True)
            for _ in range(5):
                m.eval_packed(ids, cu)
            rep = m.profile_report()
            m.profile(False)
            ts = np.asarray(ts) * 1e6
            out[f"{ftype}_n{n}"] = {"median_us": float(np.median(ts)), "p10_us": float(np.percentile(ts, 10)), "p90_us": float(np.percentile(ts, 90)),
                                    "calls": calls, "launches": int(sum(v["launches"] for k, v in rep.items() if not k.startswith("family:")) // 5),
                                    "kernel_us": {k: round(1e3 * v["total_ms"] / 5, 1) for k, v in sorted(rep.items()) if not k.startswith("family:")}}
        if ftype == "f16":
            # the small batches of a polling server (reference examples/server.cpp answers one client at a time; this library's
            # server evaluates a poll round as one call): 8 / 16 sentences of ~25 tokens, 4 of 128 — calls of up to 768 tokens
            # take the latency route too (same bits as the batch route)
            rng = np.random.default_rng(8)
            for B, n in ((8, 25), (16, 25), (4, 128)):
                lens = np.full(B, n) if n == 128 else np.clip(np.round(rng.lognormal(np.log(21.0), 0.55, B)), 3, 128).astype(np.int32)
                cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
                ids = rng.integers(1000, hp.n_vocab, size=int(cu[-1])).astype(np.int32)
                buf = np.empty((B, hp.n_embd), dtype=np.float32)
                for _ in range(10):
                    m.eval_packed(ids, cu, out=buf)
                ts = []
                for _ in range(calls):
                    t0 = time.perf_counter()
                    m.eval_packed(ids, cu, out=buf)
                    ts.append(time.perf_counter() - t0)
                ts = np.asarray(ts) * 1e6
                out[f"{ftype}_b{B}_n{n}"] = {"median_us": float(np.median(ts)), "p10_us": float(np.percentile(ts, 10)), "p90_us": float(np.percentile(ts, 90)),
                                            "calls": calls, "sentences": B, "tokens": int(cu[-1])}
        m.close()
    out["entry"] = "bert_hip_eval_packed, n_sentences = 1 (and small batches b<B>_n<tokens>), all-MiniLM-L6-v2 dims (host ids -> host embeddings, blocking)"
    return out


def encode_batch_rate(tmpdir, n_texts=32768, calls=5):
    """Strings in, embeddings out: bert_encode_batch (what the reference's ctypes callers use: sample_dylib.py:50-58,
    run_mteb.py:57-72) on English-like text of ~22 words per line (the length of the reference's sample_client_texts.txt lines)
    with a WordPiece vocabulary built from the same text; the host tokenizes and packs group g + 1 on `threads` threads while group g is
    on the GPU (groups of 2048, 4096, 8192, then 16384 texts).  texts/s, median of `calls` calls; the pointer arrays are built once outside the timed calls."""
    import collections
    import ctypes as C
    import random
    import re

    rng = random.Random(1)
    syll = ["ta", "re", "mo", "in", "ul", "es", "ka", "do", "vi", "ne", "or", "shi", "pla", "con", "ter", "ing", "ed", "ly", "un", "pre"]
    words = ["".join(rng.choice(syll) for _ in range(rng.randint(1, 4))) for _ in range(6000)]
    lines = [" ".join(rng.choice(words) for _ in range(rng.randint(5, 40))) + rng.choice([".", "?", "!", ""]) for _ in range(3000)]
    freq = collections.Counter(w for l in lines for w in re.findall(r"[a-z0-9]+", l.lower()))
    hp = gf.MODEL_DIMS["minilm-l6"]
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += [chr(c) for c in range(33, 127)] + ["##" + chr(c) for c in range(97, 123)] + ["##" + str(d) for d in range(10)]
    vocab += [w for w, _ in freq.most_common(12000)] + ["##" + x for x in ("s", "ed", "ing", "ly", "er", "es", "tion", "al", "ment", "ness")]
    vocab = list(dict.fromkeys(vocab))
    vocab += [f"[unused{i}]" for i in range(len(vocab), hp.n_vocab)]
    path = os.path.join(tmpdir, "minilm_text_vocab.bin")
    gf.write_model(path, hp, gf.synthetic_weights(hp, 0, "sensitive"), gf.FTYPE_BY_NAME["f16"], vocab=[v.encode("utf-8") for v in vocab[:hp.n_vocab]])
    m = pybert.BertModel(path)
    texts = [lines[i % len(lines)].encode("utf-8") for i in range(n_texts)]
    out = np.empty((n_texts, hp.n_embd), dtype=np.float32)
    out_ptrs = (C.POINTER(C.c_float) * n_texts)(*[out[i].ctypes.data_as(C.POINTER(C.c_float)) for i in range(n_texts)])
    txt = (C.c_char_p * n_texts)(*texts)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    n_tok = sum(len(t) for t in m.tokenize_batch(texts[:2000], threads)) / 2000.0
    ts = []
    for i in range(calls + 1):
        t0 = time.perf_counter()
        m.lib.bert_encode_batch(m.ctx, threads, 16, n_texts, txt, out_ptrs)
        if i:
            ts.append(time.perf_counter() - t0)
    ok = bool(np.isfinite(out).all() and abs(float(np.linalg.norm(out[0])) - 1) < 1e-3)
    same = bool(np.array_equal(out[0], m.encode(texts[0].decode())))
    m.close()
    med = float(np.median(ts))
    return {"value": n_texts / med, "unit": "texts/s", "ms_per_call": 1e3 * med, "n_texts": n_texts, "mean_tokens_per_text": n_tok,
            "host_threads": threads, "finite_unit_norm": ok, "row0_equals_bert_encode": same,
            "entry": "bert_encode_batch (host strings -> tokenizer on host threads, pipelined with the GPU -> host embeddings), all-MiniLM-L6-v2 dims f16"}


def cpu_baseline_and_cosine(res, budget_s=12.0, max_sent=4096, gpu=None, sample=None):
    """Oracle (ggml-faithful mode) on the host cores over a bounded sample of the same sentences: by default the step's
    sentences in order until the time budget is spent; `sample`: exactly these sentence indices."""
    from oracle import oracle as orc

    o = orc.Oracle(res["path"])
    # intra-op threading over the sentence's token rows (like ggml's n_threads): more threads
    # than ~32 only add barrier cost, and the box's logical-CPU count can exceed its cgroup quota
    cores = int(os.environ.get("ORACLE_THREADS", min(orc.usable_cores(), 32)))
    gpu = res["out"].cpu().numpy() if gpu is None else gpu
    flat, cu = res["flat"], res["cu"]
    B = len(cu) - 1
    sent = lambda i: flat[cu[i]:cu[i + 1]]
    o.eval(sent(0), orc.MODE_GGML, cores)            # warm-up (tables, page-in)
    n, t0, coss = 0, time.perf_counter(), []
    while (n < len(sample)) if sample is not None else (n < max_sent and (time.perf_counter() - t0 < budget_s or n < 2)):
        i = int(sample[n]) if sample is not None else n % B      # bounded sample: cycle through the step's sentences
        ref = o.eval(sent(i), orc.MODE_GGML, cores)
        if sample is not None or n < B:
            coss.append(float(gpu[i] @ ref / (np.linalg.norm(gpu[i]) * np.linalg.norm(ref))))
        n += 1
    dt = time.perf_counter() - t0
    what = (f"a fixed random sample of {n} of the step's {B} sentences (numpy default_rng(256))" if sample is not None
            else f"{n} single-sentence evaluations drawn from the step's sentences")
    base = {"value": n / dt, "unit": "sentences/s", "cores": cores, "kind": "port",
            "sample": f"{what} (mean length {res['tokens'] / B:.0f}), oracle ggml-faithful mode, "
                      f"OpenMP {cores} threads (usable cores {orc.usable_cores()}, logical {os.cpu_count()}), {dt:.1f} s"}
    return base, float(np.mean(coss)), float(np.min(coss)), n


def cpu_torch_fp32(hp, seq_len, budget_s=6.0):
    """The second, independent CPU number of SURVEY.md §8(d): a HuggingFace `BertModel` of the same dimensions in fp32 on torch's
    CPU backend (oneDNN GEMMs, all usable cores), batches of 16 sentences of `seq_len` tokens — a tuned CPU library's rate beside the
    restated ggml-semantics path's.  None when transformers is not importable."""
    try:
        import torch
        import transformers
    except ImportError:
        return None
    from oracle import oracle as orc
    cores = int(min(orc.usable_cores(), 64))
    torch.set_num_threads(cores)
    cfg = transformers.BertConfig(vocab_size=hp.n_vocab, hidden_size=hp.n_embd, num_hidden_layers=hp.n_layer, num_attention_heads=hp.n_head,
                                  intermediate_size=hp.n_intermediate, max_position_embeddings=hp.n_max_tokens, hidden_act="gelu_new",
                                  layer_norm_eps=1e-5)
    model = transformers.BertModel(cfg, add_pooling_layer=False).eval()
    ids = torch.from_numpy(gf.synthetic_token_ids(16, seq_len, hp.n_vocab, seed=3).astype(np.int64))
    with torch.no_grad():
        model(input_ids=ids)                                   # warm-up
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s or n == 0:
            h = model(input_ids=ids).last_hidden_state
            e = h.mean(dim=1)
            e = e / e.norm(dim=1, keepdim=True)
            n += ids.shape[0]
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "sentences/s", "cores": cores, "kind": "independent",
            "sample": f"{n} sentences of {seq_len} tokens in batches of 16 through transformers.BertModel (fp32, torch CPU, {cores} threads), {dt:.1f} s"}


SHARE_ROWS = {}       # config4_share: the sampled rows of the host-to-host call, compared with the gather entry point's


def report(res, world, torch, device, args, prof_steps, cpu_budget, replay_groups=3, headline=False):
    cfg, hp = res["cfg"], res["hp"]
    B = cfg["batch"]
    fl = [flops_per_sentence(hp, int(n)) for n in np.diff(res["cu"])] if cfg["seq_len"] is None else None
    fps = float(np.mean(fl)) if fl else flops_per_sentence(hp, cfg["seq_len"])
    r = sorted(world * B * res["steps"] / np.asarray(res["regions"]))
    e = {"workload": cfg["name"], "value": res["value"], "unit": "sentences/s", "ms_per_step": res["ms_per_step"],
         "regions": {"n": len(r), "steps_each": res["steps"], "median": float(np.median(r)), "min": float(r[0]), "max": float(r[-1])},
         "path_gflop_per_sentence": fps / 1e9, "path_mfma_frac": res["value"] * fps / (world * MFMA_PEAK_F16)}
    if cfg.get("share"):
        # one GPU's share of the 1M-sentence config: the step IS the host-to-host (or gather) call.  (Its kernels are config4's:
        # no separate roofline pass — a pass costs one more 3 s call.)
        sample = np.sort(np.random.default_rng(256).choice(B, size=48, replace=False))
        rows = res["out"].numpy()[sample].copy()
        e["entry"] = ("bert_hip_eval_packed_gather (host ids -> [B, H] matrix resident in HBM + the RCCL exchange step on a 1-rank communicator)"
                      if cfg.get("gather_step") else "bert_hip_eval_packed (host ids -> host embeddings, blocking; 61 chunks of 2048 sentences + one of 72)")
        e["chunks"] = int(-(-res["tokens"] // 262144))
        e["bytes_in"], e["bytes_out"] = int(res["flat"].nbytes), int(B * hp.n_embd * 4)
        if cfg.get("gather_step"):
            if "rows" in SHARE_ROWS:
                e["sample_rows_equal_host_call"] = bool(np.array_equal(rows, SHARE_ROWS["rows"]))
            return e
        SHARE_ROWS["rows"] = rows
        if not args.no_cpu_baseline:
            base, mc, mn, _ = cpu_baseline_and_cosine(res, gpu=res["out"].numpy(), sample=sample)
            e.update(cpu_baseline=base, mean_cosine_vs_cpu=mc, min_cosine_vs_cpu=mn, speedup_vs_cpu=res["value"] / base["value"])
        return e
    roof, bd = kernel_roofline(res, torch, device, steps=prof_steps, groups=replay_groups)
    e["roofline"] = roof
    e["kernel_ms_per_step_timed_alone_upper_bound"] = bd
    if world == 1:
        e["host_api"], host_out = host_api_rate(res)
        if headline:
            e["eval_batch_api"], api_out = eval_batch_api_rate(res)
            e["eval_batch_api"]["vs_eval_packed"] = e["eval_batch_api"]["value"] / e["host_api"]["value"]
            e["eval_batch_api"]["rows_equal_eval_packed"] = bool(np.array_equal(api_out, host_out))
        if not args.no_cpu_baseline:
            base, mc, mn, _ = cpu_baseline_and_cosine(res, budget_s=cpu_budget, gpu=host_out)
            e.update(cpu_baseline=base, mean_cosine_vs_cpu=mc, min_cosine_vs_cpu=mn, speedup_vs_cpu=res["value"] / base["value"])
    return e


# ---- the ONE line: compact (the driver parses it; round 4's 26 kB line was not parseable).  Everything else -> bench_detail.json
LINE_LIMIT = 6144


def sig(x, n=6):
    """n significant digits (floats), untouched otherwise."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    return x


def compact_roofline(r):
    if not r:
        return None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "flops_per_launch", "launches_per_step",
            "frac_of_that", "derived", "error")
    out = {k: sig(r[k]) for k in keep if k in r}
    if "frac_of_that" in out:
        out["frac_of_power_limited_rate"] = out.pop("frac_of_that")
    return out


def compact_cpu(b):
    if not b:
        return None
    return {"value": sig(b["value"]), "unit": b["unit"], "cores": b["cores"], "kind": b["kind"], "sample": b["sample"][:160]}


def brief(e):
    """<= 12 numbers per `also` entry."""
    if "error" in e:
        return {"error": str(e["error"])[:120]}
    r = e.get("roofline") or {}
    fields = {"sentences_per_s": e.get("value"), "ms_per_step": e.get("ms_per_step"), "path_mfma_frac": e.get("path_mfma_frac"),
              "kernel": r.get("kernel"), "kernel_frac": r.get("frac"), "kernel_avg_us": r.get("avg_launch_us"), "traffic": r.get("traffic"),
              "host_to_host": (e.get("host_api") or {}).get("value"), "mean_cosine": e.get("mean_cosine_vs_cpu"),
              "min_cosine": e.get("min_cosine_vs_cpu"), "cpu_sentences_per_s": (e.get("cpu_baseline") or {}).get("value"),
              "sample_rows_equal_host_call": e.get("sample_rows_equal_host_call")}
    return {k: sig(v, 8 if "cosine" in k else 5) for k, v in fields.items() if v is not None}


def compact_line(contract, e, extras, detail_path=None):
    """contract: the contract keys + config; e: the headline entry's full report; extras: {key: full report} of `also`."""
    line = dict(contract)
    if "regions" in e:
        line["regions"] = {k: sig(v) for k, v in e["regions"].items()}
    for k in ("path_gflop_per_sentence", "path_mfma_frac", "mean_cosine_vs_cpu", "min_cosine_vs_cpu", "speedup_vs_cpu"):
        if k in e:
            line[k] = sig(e[k], 8 if "cosine" in k else 6)
    line["roofline"] = compact_roofline(e.get("roofline"))
    # (an event pair around ONE launch reads 5-10 % long for sub-millisecond kernels: the breakdown says which kernels a step is made
    # of, each entry an UPPER bound — model_kernel's can exceed ms_per_step; the roofline's kernel time is measured differently)
    if e.get("kernel_ms_per_step_timed_alone_upper_bound"):
        line["kernel_ms_per_step_timed_alone_upper_bound"] = e["kernel_ms_per_step_timed_alone_upper_bound"]
    line["cpu_baseline"] = compact_cpu(e.get("cpu_baseline"))
    if isinstance(e.get("cpu_torch_fp32"), dict) and "value" in e["cpu_torch_fp32"]:
        line["cpu_torch_fp32"] = {"value": sig(e["cpu_torch_fp32"]["value"]), "unit": "sentences/s", "cores": e["cpu_torch_fp32"]["cores"], "kind": "independent"}
    if "host_api" in e:
        h = e["host_api"]
        line["host_to_host"] = {"value": sig(h["value"]), "unit": "sentences/s", "ms_per_step": sig(h["ms_per_call"]), "p05": sig(h["p05"]),
                                "p95": sig(h["p95"]), "calls": h["calls"]}
        line["device_resident"] = {"value": sig(e["value"]), "ms_per_step": sig(e["ms_per_step"])}
    if "eval_batch_api" in e:
        a = e["eval_batch_api"]
        line["eval_batch_api"] = {"value": sig(a["value"]), "unit": "sentences/s", "ms_per_step": sig(a["ms_per_call"]),
                                  "vs_eval_packed": sig(a["vs_eval_packed"], 4), "rows_equal_eval_packed": a["rows_equal_eval_packed"]}
    if extras:
        also = {}
        for k, v in extras.items():
            if "error" in v:
                also[k] = {"error": str(v["error"])[:120]}
            elif k == "latency_b1":
                also[k] = {kk: sig(vv["median_us"], 4) for kk, vv in v.items() if isinstance(vv, dict)}
                also[k]["unit"] = "us per call, host to host, one sentence"
            elif k == "encode_batch_text":
                also[k] = {"texts_per_s": sig(v["value"], 5), "mean_tokens_per_text": sig(v["mean_tokens_per_text"], 3)}
            else:
                also[k] = brief(v)
                # (north_star's target config has two answers: as written — 4-bit planes in HBM — and the engine default, which expands
                # q4 matrices to f16 at load: a reader of either entry is pointed at the other)
                twin = {"config2": ("default", "config2_expanded"), "config2_expanded": ("as_written", "config2"),
                        "config3_fused": ("default", "config3"), "config4_fused": ("default", "config4")}.get(k)
                if twin and twin[1] in extras:
                    also[k][twin[0]] = twin[1]
        line["also"] = also
    if detail_path:
        line["detail"] = detail_path
    # last resort: the line must stay parseable by the driver whatever else happens
    for drop in ("kernel_ms_per_step_timed_alone_upper_bound", "regions", "device_resident", "cpu_torch_fp32"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) >= LINE_LIMIT and "also" in line:
        line["also"] = {k: {kk: (vv[:40] if isinstance(vv, str) else vv) for kk, vv in v.items() if kk in ("sentences_per_s", "kernel_frac", "error")}
                        for k, v in line["also"].items()}
    if len(json.dumps(line)) >= LINE_LIMIT and "also" in line:
        line["also"] = {"dropped": len(line["also"]), "see": "detail"}
    return line


def write_detail(obj, name="bench_detail.json"):
    """The full per-config detail (what round 4 printed in the line), beside the line: cwd first, the temp dir if the cwd is read-only."""
    for d in (os.getcwd(), tempfile.gettempdir()):
        try:
            path = os.path.join(d, name)
            with open(path, "w") as f:
                json.dump(obj, f, default=str)
            return path
        except OSError:
            continue
    return None


def run_inproc(args):
    """ONE process drives N GPUs through libbert.so's own multi-device layer (BERT_HIP_DEVICES): token-balanced shards, one
    persistent host thread + stream per device, RCCL all-gather of the embeddings (bert_hip_eval_packed_gather).  Host-resident
    ids.  The line carries the same objects as the torchrun line: regions, roofline (device 0's kernels), cpu_baseline."""
    n = args.gpus
    os.environ["BERT_HIP_DEVICES"] = ",".join(str(d) for d in range(n))
    cfg = CONFIGS[args.config]
    hp = gf.MODEL_DIMS[cfg["dims"]]
    with tempfile.TemporaryDirectory(prefix="bert_bench_") as tmpdir:
        path = os.path.join(tmpdir, "m.bin")
        gf.make_synthetic_model(path, cfg["dims"], cfg["ftype"], seed=0)
        m = load_model(cfg, path)
        ids, cu, _ = config_inputs(dict(cfg, batch=cfg["batch"] * n), args.config, hp, 0)
        B = len(cu) - 1
        step = lambda: m.eval_packed_gather(ids, cu)
        regions = timed_regions(step, args.steps, args.warmup, args.repeat, lambda: None)
        dt = float(np.median(regions))
        r = sorted(B * args.steps / np.asarray(regions))
        fps = flops_per_sentence(hp, cfg["seq_len"] or 25)
        # per-kernel HIP-event times of device 0 (every device runs the same launches on its shard)
        res = dict(cfg=dict(cfg, host_step=True), cfg_id=args.config, hp=hp, model=m, path=path, flat=ids, cu=cu, step=step, tokens=int(cu[-1]),
                   value=B * args.steps / dt, ms_per_step=1e3 * dt / args.steps, steps=args.steps, regions=regions)
        roof, bd = kernel_roofline(res, None, None, steps=3, sync=lambda: None)
        contract = {
            "metric": "sentences/sec (seq_len=%s)" % cfg["seq_len"], "value": B * args.steps / dt, "unit": "sentences/s", "n_gpus": m.n_devices(),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic (seeded random weights in bert.cpp file format, random token ids)",
            "config": {"workload": cfg["name"], "per_gpu_batch": cfg["batch"], "global_batch": B, "seq_len": cfg["seq_len"],
                       "weights": cfg["ftype"], "parallelism": f"dp{n} in one process (libbert.so: engine + thread + stream per device, "
                                                                "RCCL all-gather per step; ids start in HOST memory)"}}
        e = {"regions": {"n": len(r), "steps_each": args.steps, "median": float(np.median(r)), "min": float(r[0]), "max": float(r[-1])},
             "path_gflop_per_sentence": fps / 1e9, "path_mfma_frac": B * args.steps / dt * fps / (n * MFMA_PEAK_F16),
             "roofline": roof, "kernel_ms_per_step_timed_alone_upper_bound": bd}
        if not args.no_cpu_baseline:
            host = m.eval_packed(ids[:int(cu[min(B, 64)])], cu[:min(B, 64) + 1])
            res["out"] = None
            base, mc, mn, _ = cpu_baseline_and_cosine(dict(res, cu=cu[:min(B, 64) + 1], tokens=int(cu[min(B, 64)])), budget_s=10.0, gpu=host)
            e.update(cpu_baseline=base, mean_cosine_vs_cpu=mc, min_cosine_vs_cpu=mn, speedup_vs_cpu=contract["value"] / base["value"])
        line = compact_line(contract, e, {}, write_detail(dict(contract, **e), "bench_detail_inproc.json"))
        emit_line(line)
        m.close()


def emit_line(line):
    """The ONE JSON line, as the LAST line of stdout: whatever native libraries have printed through C stdio so far (RCCL writes
    a version banner at its first communicator: the in-process gather entry) sits in C's buffer and would otherwise come out
    at exit, behind Python's own buffer."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=5, help="timed regions of --steps steps each; value = the median region")
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument("--also", type=int, nargs="*", default=None,
                    help="extra configs reported under 'also' (default at N=1 with config 1: 2, 22, 3, 33, 4, 42, 44, 45, 5, 55, 58)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inproc", action="store_true", help="one process, --gpus N devices inside libbert.so")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.inproc and world == 1:
        return run_inproc(args)

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # BERT_BENCH_SHARED_GPU=1 (a validation aid for a box with fewer GPUs than ranks, never a measurement): the ranks share the
    # visible GPUs round robin and exchange over gloo — RCCL refuses two ranks on one device — so that the N > 1 code path of this
    # file (barriers, max over ranks, the gather per step, the rank-0 line) can be run end to end on the 1-GPU box
    shared = os.environ.get("BERT_BENCH_SHARED_GPU", "") not in ("", "0") and world > 1
    if shared:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # one rank = one GPU: the rank's context lives on its own device only
    os.environ["BERT_HIP_DEVICES"] = str(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    with tempfile.TemporaryDirectory(prefix="bert_bench_") as tmpdir:
        t_start = time.perf_counter()
        clock = lambda what: print(f"[bench] {what}: {time.perf_counter() - t_start:.1f} s since start", file=sys.stderr) if rank == 0 else None
        res = run_config(args.config, args, rank, world, device, dist, torch, tmpdir)
        contract = e = None
        if rank == 0:
            cfg = res["cfg"]
            prof_steps = int(min(20, max(2, 0.4 / (res["ms_per_step"] * 1e-3))))        # (about 0.4 s of profiled steps)
            e = report(res, world, torch, device, args, prof_steps=prof_steps, cpu_budget=10.0, replay_groups=3, headline=True)
            if world == 1 and not args.no_cpu_baseline and cfg["seq_len"]:
                try:
                    e["cpu_torch_fp32"] = cpu_torch_fp32(res["hp"], cfg["seq_len"], budget_s=4.0)
                except Exception as ex:
                    e["cpu_torch_fp32"] = {"error": f"{type(ex).__name__}: {ex}"}
            contract = {
                "metric": "sentences/sec (seq_len=%s)" % cfg["seq_len"], "value": res["value"], "unit": "sentences/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
                "data": "synthetic (seeded random weights in bert.cpp file format, random token ids)",
                "config": {"workload": cfg["name"], "per_gpu_batch": cfg["batch"], "global_batch": cfg["batch"] * world,
                           "seq_len": cfg["seq_len"], "weights": cfg["ftype"],
                           "parallelism": f"dp{world} (replicated weights, sharded sentences"
                                          + (", RCCL all-gather of embeddings per step, under the next step's forward pass)" if world > 1 else ")")},
            }
            if shared:
                contract["validation_only"] = "BERT_BENCH_SHARED_GPU: the ranks share the box's GPU(s) and exchange over gloo — not a scaling measurement"
            if "host_api" in e:
                # SURVEY.md §8(d) quotes the metric host to host; the bench contract keeps `value` on HBM-resident inputs
                # ("the PCIe-inclusive rate ... is never `value`"), so the host-to-host rate travels beside it, in `config` too
                contract["config"]["host_to_host_sentences_per_s"] = sig(e["host_api"]["value"])
            clock("headline config reported")
        else:
            kernel_roofline(res, torch, device, steps=int(min(20, max(2, 0.4 / (res["ms_per_step"] * 1e-3)))), groups=3)      # (the same passes as rank 0's report: the steps hold collectives)
        res["model"].close()
        also = args.also if args.also is not None else ([2, 22, 3, 33, 4, 42, 44, 45, 5, 55, 58] if world == 1 and args.config == 1 else [])
        extras = {}
        for cid in also:
            big = CONFIGS[cid]["dims"] in ("bert-base", "mpnet-dims")
            share = bool(CONFIGS[cid].get("share"))                   # (a step is one 125,000-sentence call: about three seconds)
            key = CONFIGS[cid].get("key", f"config{cid}")
            try:
                # share: ONE timed region of one call; the warm-up (workspace, staging, clocks) is the same call on a 4096-sentence prefix
                r2 = run_config(cid, args, rank, world, device, dist, torch, tmpdir, steps=1 if share else 3 if big else max(10, args.steps // 4),
                                warmup=1 if big else 5, repeat=1 if share else 3)
                if rank == 0:
                    extras[key] = report(r2, world, torch, device, args, prof_steps=2 if big else 3, cpu_budget=2.5 if big else 2.0)
                else:
                    kernel_roofline(r2, torch, device, steps=2 if big else 3)
                if CONFIGS[cid].get("option", ("", ""))[0] == "window_slots":
                    r2["model"].set_option("window_slots", "16")          # (the setting is process-wide)
                r2["model"].close()
            except (Exception, SystemExit) as ex:                      # an auxiliary entry must never cost the headline line (its error is in the line and on stderr)
                if world > 1:
                    raise
                extras[key] = {"workload": CONFIGS[cid]["name"], "error": f"{type(ex).__name__}: {ex}"}
                print(f"bench.py: entry {key} failed: {ex}", file=sys.stderr)
                if CONFIGS[cid].get("option", ("", ""))[0] == "window_slots":        # (process-wide: back to the default whatever happened)
                    try:
                        m2 = pybert.BertModel(os.path.join(tmpdir, f"minilm-l6_f16_rank{rank}.bin"))
                        m2.set_option("window_slots", "16")
                        m2.close()
                    except Exception:
                        pass
            clock(key)
        if rank == 0:
            if world == 1 and args.config == 1 and args.also is None:
                for k, fn in (("latency_b1", latency_b1), ("encode_batch_text", encode_batch_rate)):
                    try:
                        extras[k] = fn(tmpdir)
                    except Exception as ex:
                        extras[k] = {"error": f"{type(ex).__name__}: {ex}"}
                        print(f"bench.py: entry {k} failed: {ex}", file=sys.stderr)
                    clock(k)
            strip = lambda d: {k: v for k, v in d.items() if k not in ("cfg", "hp", "model", "step", "flat", "cu", "out")}
            detail = write_detail(dict(contract, **strip(e), also={k: strip(v) for k, v in extras.items()}))
            emit_line(compact_line(contract, dict(e, value=res["value"], ms_per_step=res["ms_per_step"]), extras, detail))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
