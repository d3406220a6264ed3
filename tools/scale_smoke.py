#!/usr/bin/env python3
"""One arm of tools/scale_smoke.sh: evaluate a FIXED batch on N GPUs and print a digest of the [B, H] embedding matrix.

    python tools/scale_smoke.py inproc N              one process, N devices inside libbert.so (BERT_HIP_DEVICES=0..N-1):
                                                      bert_hip_eval_packed (shards write the caller's host rows) AND
                                                      bert_hip_eval_packed_gather (RCCL all-gather; every device's copy checked)
    torchrun ... tools/scale_smoke.py torchrun        one process per GPU (bert.cpp_amd/dist.py): shard, evaluate, all_gather

The batch: 4096 sentences of 128 tokens (equal shards: the ncclAllGather form of the exchange) followed by 3001 sentences of
3..128 tokens (token-balanced unequal shards: the grouped-broadcast form), MiniLM-L6 dims f16, seed-fixed.  Per-sentence bits do
not depend on the number of GPUs: every arm must print the same two digests as N = 1."""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BERT_HIP_QUIET", "1")

from bert_cpp_amd import ggml_file as gf  # noqa: E402


def batches(hp):
    fixed = gf.synthetic_token_ids(4096, 128, hp.n_vocab, seed=11)
    rng = np.random.default_rng(12)
    lens = rng.integers(3, 129, size=3001)
    ragged = [rng.integers(1000, hp.n_vocab, size=int(n)).astype(np.int32) for n in lens]
    return [list(fixed), ragged]


def pack(sents):
    cu = np.concatenate([[0], np.cumsum([len(s) for s in sents])]).astype(np.int32)
    return np.concatenate(sents).astype(np.int32), cu


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()[:16]


def main():
    mode = sys.argv[1]
    hp = gf.MODEL_DIMS["minilm-l6"]
    with tempfile.TemporaryDirectory() as d:
        if mode == "inproc":
            n = int(sys.argv[2])
            os.environ["BERT_HIP_DEVICES"] = ",".join(str(i) for i in range(n))
            from bert_cpp_amd import pybert
            import ctypes
            path = os.path.join(d, "m.bin")
            gf.make_synthetic_model(path, "minilm-l6", "f16", seed=0)
            m = pybert.BertModel(path)
            assert m.n_devices() == n, (m.n_devices(), n)
            hip = ctypes.CDLL("libamdhip64.so")
            out = []
            for sents in batches(hp):
                flat, cu = pack(sents)
                host = m.eval_packed(flat, cu)
                ptrs = m.eval_packed_gather(flat, cu)
                for dev, p in enumerate(ptrs):              # every device holds the whole matrix after the exchange
                    got = np.empty_like(host)
                    assert hip.hipSetDevice(dev) == 0 and hip.hipDeviceSynchronize() == 0
                    assert hip.hipMemcpy(ctypes.c_void_p(got.ctypes.data), ctypes.c_void_p(p), ctypes.c_size_t(got.nbytes), 2) == 0
                    assert np.array_equal(got, host), f"device {dev}: gathered matrix differs from the host rows"
                out.append(digest(host))
            print(f"scale_smoke inproc n={n} devices_in_context={m.n_devices()} (one RCCL rank per device, communicator made at load) digests {' '.join(out)}")
        else:
            import torch
            import torch.distributed as dist
            from bert_cpp_amd import dist as bdist
            from bert_cpp_amd import pybert
            rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
            # BERT_BENCH_SHARED_GPU=1 (validation on a box with fewer GPUs than ranks, as in bench.py): the ranks share the GPUs round
            # robin and exchange over gloo on host tensors — RCCL refuses two ranks on one device
            shared = os.environ.get("BERT_BENCH_SHARED_GPU", "") not in ("", "0")
            if shared:
                local %= torch.cuda.device_count()
            torch.cuda.set_device(local)
            os.environ["BERT_HIP_DEVICES"] = str(local)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if shared:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            path = os.path.join(d, f"m{rank}.bin")
            gf.make_synthetic_model(path, "minilm-l6", "f16", seed=0)
            m = pybert.BertModel(path)
            out = []
            for sents in batches(hp):
                def ev(ss):
                    rows = m.eval_packed(*pack(ss))
                    if os.environ.get("SCALE_SMOKE_CORRUPT") and rank == dist.get_world_size() - 1 and len(rows):
                        rows = rows.copy(); rows[0, 0] += 1.0       # (the harness's own check: a wrong row must fail the script)
                    return rows
                emb = bdist.encode_sharded(ev, sents, device="cpu" if shared else torch.device("cuda", local))
                out.append(digest(emb.cpu().numpy()))
            if rank == 0:
                print(f"scale_smoke torchrun n={dist.get_world_size()} ranks_in_group={dist.get_world_size()} backend={dist.get_backend()} digests {' '.join(out)}")
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
