import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bert_cpp_amd import pybert
for n_head in (4, 12):
    for lens in ([128,128,128], [96,97,48]):
        d_head, H = 32, 32 * n_head
        rng = np.random.default_rng(sum(lens) + n_head)
        cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        T = int(cu[-1])
        x = rng.normal(0, 1, (T, H)).astype(np.float16)
        W = (rng.normal(0, 1, (3 * H, H)) / np.sqrt(H)).astype(np.float16)
        W[:H] *= 1.7
        W[:, : H // 2] *= 1.3
        bias = rng.normal(0, 0.3, 3 * H).astype(np.float32)
        r = {m: pybert.test_qkv_attention(x, cu, n_head, d_head, W.view(np.uint8), 1, bias, m) for m in (0, 1, 2, 3)}
        for a in (0, 1, 2, 3):
            for b in range(a + 1, 4):
                neq = np.argwhere(r[a].view(np.uint16) != r[b].view(np.uint16))
                print(n_head, lens, "modes", a, b, "differ at", len(neq), neq[:4].tolist(),
                      [(float(r[a][i, j]), float(r[b][i, j])) for i, j in neq[:3]])
