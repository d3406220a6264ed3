#!/usr/bin/env python3
"""Hypothesis probe: two half-batches on two streams, out of phase by a fraction of a kernel, against one full batch on one stream
(layer_tail's HBM-burst prologue / epilogue under the other lane's compute).  Two contexts = two workspaces.
usage: dual_lane_probe.py [lag_us ...]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np, torch
from bert_cpp_amd import ggml_file as gf, pybert

B, N, STEPS = 256, 128, 200
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.bin"); hp = gf.make_synthetic_model(p, "minilm-l6", "f16", seed=0)
    ids = gf.synthetic_token_ids(B, N, hp.n_vocab, seed=1235)
    H = hp.n_embd

    def lane(b0, b1):
        m = pybert.BertModel(p)
        t = torch.from_numpy(ids[b0:b1].reshape(-1).copy()).to(dev)
        cu = torch.from_numpy((np.arange(b1 - b0 + 1) * N).astype(np.int32)).to(dev)
        out = torch.empty((b1 - b0, H), dtype=torch.float32, device=dev)
        m.reserve((b1 - b0) * N, b1 - b0)
        s = torch.cuda.Stream(dev)
        return dict(m=m, t=t, cu=cu, out=out, s=s, nb=b1 - b0)

    def run(l):
        l["m"].eval_packed_device(l["t"].data_ptr(), l["cu"].data_ptr(), l["nb"], l["nb"] * N, N, l["out"].data_ptr(), l["s"].cuda_stream)

    full = lane(0, B)
    for _ in range(20): run(full)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS): run(full)
    torch.cuda.synchronize()
    base = B * STEPS / (time.perf_counter() - t0)
    print(f"one lane, {B} sentences per step: {base:,.0f} sent/s")
    ref = full["out"].cpu().numpy().copy()

    for L, join, skew_us in ((2, False, 0), (2, True, 0), (2, True, 10), (2, True, 25), (4, True, 0)):
        lanes = [lane(k * B // L, (k + 1) * B // L) for k in range(L)]
        main = torch.cuda.Stream(dev)
        def step():
            if join:                       # what an engine-internal split would have to do: fork from / join into the caller's stream
                e0 = torch.cuda.Event(); e0.record(main)
                for k, l in enumerate(lanes):
                    l["s"].wait_event(e0)
                    if skew_us and k:
                        with torch.cuda.stream(l["s"]): torch.cuda._sleep(int(skew_us * 2100 * k))
                    run(l)
                    e = torch.cuda.Event(); e.record(l["s"]); main.wait_event(e)
            else:
                for l in lanes: run(l)
        for _ in range(10): step()
        torch.cuda.synchronize()
        best = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(STEPS): step()
            torch.cuda.synchronize()
            best = max(best, B * STEPS / (time.perf_counter() - t0))
        got = np.concatenate([l["out"].cpu().numpy() for l in lanes])
        print(f"{L} lanes of {B // L}, join per step {join}, skew {skew_us} us: {best:,.0f} sent/s ({best / base:.3f}x)  equal bits: {np.array_equal(got, ref)}")
        for l in lanes: l["m"].close()
