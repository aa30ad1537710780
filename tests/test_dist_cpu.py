"""Multi-GPU path rehearsed on CPU with gloo, world_size 2: env sharding and the summary all-reduce."""

import os
import socket

import numpy as np
import torch.multiprocessing as mp

from metta_amd.dist import shard_range, shard_seeds, summary_to_metrics


def test_shard_ranges_cover_everything():
    for total, world in [(4096, 8), (10, 3), (7, 2), (32768, 8)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and b > a
    s0, s1 = shard_seeds(100, 10, 0, 2), shard_seeds(100, 10, 1, 2)
    assert np.concatenate([s0, s1]).tolist() == list(range(100, 110))  # env i -> seed base + i (PUF/vector.py:90)


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from metta_amd.dist import all_reduce_summary

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local = np.array([rank + 1.0, 100.0 * (rank + 1), 2.5 * (rank + 1), 7.0, 0.0], dtype=np.float64)
    out = all_reduce_summary(local)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out.tolist()))


def test_summary_all_reduce_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    assert res[0] == res[1] == [3.0, 300.0, 7.5, 14.0, 0.0]
    m = summary_to_metrics(np.array(res[0]), ["tokens_written", "tokens_dropped"], agents_per_env=24)
    assert m["episodes"] == 3 and m["episode_length"] == 100 and abs(m["game/tokens_written"] - 14 / 3) < 1e-12
    assert "game/tokens_dropped" not in m


def _norm_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from metta_amd.advantage import normalize_advantage_distributed

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.from_numpy(np.random.default_rng(7).standard_normal((6, 10)).astype(np.float32) * 3 + 1)
    out = normalize_advantage_distributed(full[rank * 3 : rank * 3 + 3])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out.numpy().tolist()))


def test_normalize_advantage_distributed_gloo_world2():
    """metta/rl/advantage.py:122-150 over two ranks = the global (biased-variance) normalisation of the union."""
    import torch

    from metta_amd.advantage import normalize_advantage_distributed

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_norm_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    full = np.random.default_rng(7).standard_normal((6, 10)).astype(np.float32) * 3 + 1
    want = (full - full.mean()) / (np.sqrt(max(full.var(), 1e-8)) + 1e-8)
    got = np.concatenate([np.array(res[0]), np.array(res[1])])
    assert got.shape == (6, 10) and np.allclose(got, want, rtol=1e-4, atol=1e-5)
    # single process: torch's unbiased std, and the switch
    t = torch.from_numpy(full)
    one = normalize_advantage_distributed(t)
    assert torch.allclose(one, (t - t.mean()) / (t.std() + 1e-8))
    assert normalize_advantage_distributed(t, norm_adv=False) is t
