#!/usr/bin/env python3
"""Quick A/B probes for kernel work, one gpurun call for several libraries (BERT_HIP_LIB=bert.cpp_amd/libbert_<variant>.so, built by
tools/variant.sh):
    probe.py rate [regions]          device-resident sentences/s of configs[1]'s shape (256 x 128, f16), 300-step regions
                                     (PROBE_OPTIONS=key=value,...: bert_hip_set_option calls after the load)
    probe.py kernels <config> [...]  per-kernel HIP-event milliseconds per step of one bench config (bench.py's CONFIGS ids)
    probe.py latency [n_tokens] [calls]   one sentence per call through bert_hip_eval_packed: median microseconds, host to host
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np  # noqa: E402

from bert_cpp_amd import ggml_file as gf, pybert  # noqa: E402

LIB = os.environ.get("BERT_HIP_LIB", "default").split("libbert_")[-1]


def rate(regions=3):
    import torch
    dev = torch.device("cuda", 0)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.bin")
        hp = gf.make_synthetic_model(p, "minilm-l6", "f16", seed=0)
        m = pybert.BertModel(p)
        for kv in filter(None, os.environ.get("PROBE_OPTIONS", "").split(",")):      # e.g. PROBE_OPTIONS=one_launch=0
            m.set_option(*kv.split("="))
        B = 256
        ids = gf.synthetic_token_ids(B, 128, hp.n_vocab, seed=1235)
        t = torch.from_numpy(ids.reshape(-1).copy()).to(dev)
        cu = torch.from_numpy((np.arange(B + 1) * 128).astype(np.int32)).to(dev)
        out = torch.empty((B, hp.n_embd), dtype=torch.float32, device=dev)
        m.reserve(B * 128, B)
        s = torch.cuda.current_stream(dev)
        step = lambda: m.eval_packed_device(t.data_ptr(), cu.data_ptr(), B, B * 128, 128, out.data_ptr(), s.cuda_stream)
        rates = []
        for _ in range(regions):
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                step()
            torch.cuda.synchronize()
            rates.append(B * 300 / (time.perf_counter() - t0))
        print(LIB, " ".join(f"{r / 1e3:.1f}k" for r in rates), "checksum", float(out.double().sum()), flush=True)
        m.close()


def kernels(cfg):
    import bench
    import torch

    class A:
        steps = int(os.environ.get("STEPS", "5")); warmup = 2; repeat = int(os.environ.get("REPEAT", "3")); also = False; config = 1; gpus = 1
        inproc = False; no_cpu_baseline = True
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    with tempfile.TemporaryDirectory() as d:
        r = bench.run_config(cfg, A, 0, 1, dev, None, torch, d)
        _, bd = bench.kernel_roofline(r, torch, dev, steps=5)
        print(f"{LIB} cfg{cfg} {r['value']:.0f} sent/s {r['ms_per_step']:.3f} ms/step", json.dumps(bd), flush=True)
        r["model"].close()


def latency(n=128, calls=200):
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "m.bin")
        hp = gf.make_synthetic_model(path, "minilm-l6", "f16", seed=0)
        m = pybert.BertModel(path)
        ids = gf.synthetic_token_ids(1, n, hp.n_vocab, seed=77).reshape(-1)
        cu = np.array([0, n], dtype=np.int32)
        for _ in range(10):
            m.eval_packed(ids, cu)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            m.eval_packed(ids, cu)
            ts.append(time.perf_counter() - t0)
        print(f"{LIB} n={n}: median {1e6 * np.median(ts):.1f} us over {calls} calls", flush=True)
        m.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "rate"
    args = [int(a) for a in sys.argv[2:]]
    if what == "rate":
        rate(*args[:1])
    elif what == "kernels":
        for c in args or [1]:
            kernels(c)
    elif what == "latency":
        latency(*args[:2])
    else:
        raise SystemExit(__doc__)
