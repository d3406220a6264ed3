#!/usr/bin/env python3
# device-resident rate of configs[1]'s shape (256 x 128, f16) for the library named by BERT_HIP_LIB: A/B runs of kernel variants
# (tools/variant.sh) in one gpurun call.  usage: rate_probe.py [regions]
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BERT_HIP_QUIET", "1")
import numpy as np, torch
from bert_cpp_amd import ggml_file as gf, pybert
dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "m.bin"); hp = gf.make_synthetic_model(p, "minilm-l6", "f16", seed=0)
    m = pybert.BertModel(p)
    B = 256
    ids = gf.synthetic_token_ids(B, 128, hp.n_vocab, seed=1235)
    t = torch.from_numpy(ids.reshape(-1).copy()).to(dev); cu = torch.from_numpy((np.arange(B + 1) * 128).astype(np.int32)).to(dev)
    out = torch.empty((B, hp.n_embd), dtype=torch.float32, device=dev)
    m.reserve(B * 128, B)
    s = torch.cuda.current_stream(dev)
    rates = []
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        for _ in range(30): m.eval_packed_device(t.data_ptr(), cu.data_ptr(), B, B * 128, 128, out.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): m.eval_packed_device(t.data_ptr(), cu.data_ptr(), B, B * 128, 128, out.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize(); rates.append(B * 300 / (time.perf_counter() - t0))
    print(os.environ.get("BERT_HIP_LIB", "default").split("libbert_")[-1], " ".join(f"{r / 1e3:.1f}k" for r in rates), "checksum", float(out.double().sum()), flush=True)
    m.close()
