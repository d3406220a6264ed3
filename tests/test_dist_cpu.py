"""Multi-rank path on CPU: world_size-2 (and 3) gloo process groups exercise the sharding and the
single all-gather of embeddings (bert.cpp_amd/dist.py); the per-rank evaluator here is the CPU
oracle standing in for the rank's GPU context."""
import os
import socket

import numpy as np
import pytest

from bert_cpp_amd import dist as bdist
from bert_cpp_amd import ggml_file as gf


def test_shard_bounds_properties():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 2, 7, 100):
            lens = rng.integers(1, 129, size=n).tolist()
            b = bdist.shard_bounds(lens, world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(s <= e for s, e in b)
            if n >= 4 * world:
                tok = [sum(lens[s:e]) for s, e in b]
                assert max(tok) - min(tok) <= 2 * 128
    # equal lengths -> equal counts
    assert bdist.shard_bounds([128] * 1024, 8) == [(i * 128, (i + 1) * 128) for i in range(8)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, model_path, n_sent, seed, out_dir):
    import torch
    import torch.distributed as dist

    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["ORACLE_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = orc.Oracle(model_path)
        rng = np.random.default_rng(seed)              # same sentences on every rank
        sents = [rng.integers(0, 256, size=int(n)).astype(np.int32) for n in rng.integers(1, 40, size=n_sent)]
        emb = bdist.encode_sharded(lambda ss: o.eval_batch(ss, orc.MODE_PLAIN, 2), sents)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), emb.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "ref.npy"), o.eval_batch(sents, orc.MODE_PLAIN, 2) if sents else np.zeros((0, o.n_embd)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_sent", [(2, 9), (2, 1), (3, 2), (8, 37)])        # (8: the node's rank count)
def test_sharded_encode_matches_single_process(tmp_path, world, n_sent):
    import torch.multiprocessing as mp

    model_path = str(tmp_path / "tiny.bin")
    gf.make_synthetic_model(model_path, "tiny", "f32", seed=5)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, model_path, n_sent, 42, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "ref.npy")
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == ref.shape
        assert np.array_equal(got, ref)            # same code per sentence wherever it runs: bit-exact


def _overlap_worker(rank, world, port, out_dir):
    """bench.py's N > 1 step: two result buffers in turn, the exchange of step k left running under step k + 1
    (gather_embeddings(async_op=True)); a buffer is rewritten only after the exchange that read it was waited for."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, H, steps = 5, 7, 6
        locals_ = [torch.empty((B, H)), torch.empty((B, H))]
        alls = [torch.empty((world * B, H)), torch.empty((world * B, H))]
        works = [None, None]
        seen = []
        for k in range(steps):
            i = k & 1
            if works[i] is not None:
                works[i].wait()
                seen.append(alls[i].clone())
            locals_[i].copy_(torch.full((B, H), float(100 * k + rank)) + torch.arange(B)[:, None])
            out, works[i] = bdist.gather_embeddings(locals_[i], [B] * world, out=alls[i], async_op=True)
            assert out is alls[i]
        for k in (steps - 2, steps - 1):
            works[k & 1].wait()
            seen.append(alls[k & 1].clone())
        torch.save(torch.stack(seen), os.path.join(out_dir, f"overlap{rank}.pt"))
        # unequal shards cannot be exchanged without the wait
        with pytest.raises(AssertionError):
            bdist.gather_embeddings(torch.zeros((rank + 1, H)), [r + 1 for r in range(world)], async_op=True)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gather_left_running_under_the_next_step(tmp_path, world):
    import torch
    import torch.multiprocessing as mp

    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    B, H, steps = 5, 7, 6
    want = torch.stack([torch.cat([torch.full((B, H), float(100 * k + r)) + torch.arange(B)[:, None] for r in range(world)]) for k in range(steps)])
    for r in range(world):
        assert torch.equal(torch.load(tmp_path / f"overlap{r}.pt"), want), r
