#!/usr/bin/env python3
# model_kernel.hip on ragged windows against the two-kernels-per-layer route: equal bits (host windows, device-built windows,
# one sentence per window), device-resident rate on the bench's mixed-length batch
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np, torch
import bench
from bert_cpp_amd import ggml_file as gf, pybert
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream(dev)

def device_eval(m, flat, cu, max_len, H):
    B, T = len(cu) - 1, int(cu[-1])
    t = torch.from_numpy(flat).to(dev); c = torch.from_numpy(cu).to(dev)
    out = torch.empty((B, H), dtype=torch.float32, device=dev)
    m.eval_packed_device(t.data_ptr(), c.data_ptr(), B, T, max_len, out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()

gf.MODEL_DIMS["h256"] = gf.BertHParams(1000, 128, 256, 1024, 8, 3)
for dims, ftype in (("minilm-l6", "f16"), ("minilm-l12", "q4_1"), ("h256", "f16")):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.bin"); hp = gf.make_synthetic_model(p, dims, ftype, seed=0)
        m = pybert.BertModel(p)
        rng = np.random.default_rng(5)
        cases = {"mixed 300": rng.integers(1, 129, size=300), "ones 70": np.ones(70, dtype=np.int64), "16s": np.full(40, 16),
                 "17s": np.full(33, 17), "two": np.array([128, 3]), "max 64": rng.integers(40, 65, size=50),
                 "max 32 equal": np.full(64, 32), "one 5": np.array([5]), "127s": np.full(9, 127), "big": rng.integers(3, 129, size=3000)}
        for name, lens in cases.items():
            lens = lens.astype(np.int64)
            cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            flat = rng.integers(0, hp.n_vocab, size=int(cu[-1])).astype(np.int32)
            ml = int(lens.max())
            m.set_option("latency", "0")
            m.set_option("one_launch", "0"); ref = m.eval_packed(flat, cu); refd = device_eval(m, flat, cu, ml, hp.n_embd)
            m.set_option("one_launch", "1"); got = m.eval_packed(flat, cu)
            m.profile(True); gotd = device_eval(m, flat, cu, ml, hp.n_embd); names = sorted(m.profile_report()); m.profile(False)
            gotd128 = device_eval(m, flat, cu, 128, hp.n_embd)
            print(dims, ftype, name, "host", bool(np.array_equal(ref, got)), "device", bool(np.array_equal(refd, gotd)), bool(np.array_equal(ref, gotd)),
                  bool(np.array_equal(ref, gotd128)), "finite", bool(np.isfinite(got).all()), names, flush=True)
        m.close()

cfg = bench.CONFIGS[5]
hp = gf.MODEL_DIMS[cfg["dims"]]
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.bin"); gf.make_synthetic_model(p, cfg["dims"], cfg["ftype"], seed=0)
    m = pybert.BertModel(p)
    flat, cu, max_len = bench.config_inputs(cfg, 5, hp, 0)
    B, T = len(cu) - 1, int(cu[-1])
    t = torch.from_numpy(flat).to(dev); c = torch.from_numpy(cu).to(dev)
    out = torch.empty((B, hp.n_embd), dtype=torch.float32, device=dev)
    m.reserve(T, B)
    res = {}
    for mode in ("0", "1", "0", "1"):
        m.set_option("one_launch", mode)
        for _ in range(5): m.eval_packed_device(t.data_ptr(), c.data_ptr(), B, T, max_len, out.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): m.eval_packed_device(t.data_ptr(), c.data_ptr(), B, T, max_len, out.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[mode] = out.cpu().numpy()
        print("mixed", B, "sentences", T, "tokens, one_launch", mode, f"{B * 30 / dt:,.0f} sent/s", flush=True)
        t0 = time.perf_counter()
        for _ in range(10): m.eval_packed(flat, cu)
        dt = time.perf_counter() - t0
        print("   host to host", f"{B * 10 / dt:,.0f} sent/s", flush=True)
    print("mixed equal bits", bool(np.array_equal(res["0"], res["1"])))
    m.profile(True)
    for _ in range(3): m.eval_packed_device(t.data_ptr(), c.data_ptr(), B, T, max_len, out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    print({k: round(v["total_ms"] / v["launches"], 4) for k, v in m.profile_report().items()})
    m.close()
