#!/usr/bin/env python3
"""Stress test of model_kernel.hip on ragged windows: random sub-batches of a pool of sentences through the one-launch route
(one_launch=2, host-built and device-built windows) must reproduce, bit for bit and every time, the pool's embeddings from
the two-launch route.  usage: stress_one_launch.py [iterations]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np, torch
from bert_cpp_amd import ggml_file as gf, pybert

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream(dev)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.bin"); hp = gf.make_synthetic_model(p, "minilm-l6", "f16", seed=0)
    m = pybert.BertModel(p)
    m.set_option("latency", "0")
    rng = np.random.default_rng(1)
    lens = np.concatenate([rng.integers(1, 129, size=300), np.full(60, 128), rng.integers(120, 129, size=40)])
    pool = [rng.integers(0, hp.n_vocab, size=int(n)).astype(np.int32) for n in lens]
    m.set_option("one_launch", "0")
    ref = m.eval_batch(pool)
    m.set_option("one_launch", "2")
    bad = 0
    for it in range(iters):
        k = int(rng.integers(1, 400))
        idx = rng.choice(len(pool), size=k, replace=True)
        sents = [pool[i] for i in idx]
        if it % 2 == 0:
            out = m.eval_batch(sents)
        else:
            flat = np.concatenate(sents); cu = np.concatenate([[0], np.cumsum([len(x) for x in sents])]).astype(np.int32)
            t = torch.from_numpy(flat).to(dev); c = torch.from_numpy(cu).to(dev)
            o = torch.empty((k, hp.n_embd), dtype=torch.float32, device=dev)
            m.eval_packed_device(t.data_ptr(), c.data_ptr(), k, int(cu[-1]), 128, o.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize()
            out = o.cpu().numpy()
        for j, i in enumerate(idx):
            if not np.array_equal(out[j], ref[i]):
                bad += 1
                if bad <= 10:
                    print(f"iter {it}: sentence {i} (len {lens[i]}) at {j} of {k} differs, max abs {np.abs(out[j] - ref[i]).max():.3e}")
    print(f"{iters} batches through model_kernel, {bad} mismatches")
    sys.exit(1 if bad else 0)
