#!/usr/bin/env python3
# model_kernel.hip against the two-kernels-per-layer route: equal bits, device-resident rate
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np, torch
from bert_cpp_amd import ggml_file as gf, pybert
dev = torch.device("cuda", 0)
for dims, ftype in (("minilm-l6", "f16"), ("minilm-l6", "q4_0")):
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.bin"); hp = gf.make_synthetic_model(p, dims, ftype, seed=0)
        m = pybert.BertModel(p)
        for B in (2, 5, 256, 1024):
            ids = gf.synthetic_token_ids(B, 128, hp.n_vocab, seed=3 + B)
            flat = ids.reshape(-1).copy(); cu = (np.arange(B + 1) * 128).astype(np.int32)
            m.set_option("one_launch", "0"); ref = m.eval_packed(flat, cu)
            m.set_option("one_launch", "1"); got = m.eval_packed(flat, cu)
            m.profile(True); m.eval_packed(flat, cu); names = sorted(m.profile_report()); m.profile(False)
            print(dims, ftype, "B", B, "equal bits", bool(np.array_equal(ref, got)), "max diff", float(np.abs(ref - got).max()), names, flush=True)
        B = 256
        ids = gf.synthetic_token_ids(B, 128, hp.n_vocab, seed=1235)
        t = torch.from_numpy(ids.reshape(-1).copy()).to(dev); cu = torch.from_numpy((np.arange(B + 1) * 128).astype(np.int32)).to(dev)
        out = torch.empty((B, hp.n_embd), dtype=torch.float32, device=dev)
        m.reserve(B * 128, B)
        s = torch.cuda.current_stream(dev)
        for mode in ("0", "1", "0", "1"):
            m.set_option("one_launch", mode)
            for _ in range(20): m.eval_packed_device(t.data_ptr(), cu.data_ptr(), B, B * 128, 128, out.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(400): m.eval_packed_device(t.data_ptr(), cu.data_ptr(), B, B * 128, 128, out.data_ptr(), s.cuda_stream)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(dims, ftype, "one_launch", mode, f"{B * 400 / dt:,.0f} sent/s", flush=True)
        m.close()
