"""Multi-GPU path rehearsed on CPU with gloo, world_size 2: env sharding and the summary all-reduce."""

import os
import socket

import numpy as np
import pytest
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


def test_bench_sharding_world_2_and_8_equals_one_big_handle_on_the_oracle():
    """`bench.py --gpus N` gives rank r the global env ids [r*E, (r+1)*E): the ids are disjoint and complete, and the
    environments a rank steps are the same ones (seed, map, trajectory) as that block of a single N*E-env handle."""
    from metta_amd.builder import make_arena
    from metta_amd.compile import compile_game
    from metta_amd.dist import bench_shard
    from oracle.mg_oracle import make_oracle_env
    from tests.helpers import load_map_pool

    pool = load_map_pool()
    for world, E in [(2, 4096), (8, 4096), (8, 3)]:
        seeds, maps = [], []
        for r in range(world):
            s, m = bench_shard(r, E, len(pool))
            assert len(s) == len(m) == E
            seeds += s
            maps += m
        assert seeds == list(range(world * E)), "global env ids: disjoint, complete, in rank order"
        assert maps == [g % len(pool) for g in range(world * E)]
        assert bench_shard(0, world * E, len(pool)) == (seeds, maps), "N shards of E == one handle of N*E envs"

    # the oracle steps the shards of 2 and of 8 ranks and one big handle through the same actions per global env id
    cfg = make_arena(24)
    comp = compile_game(cfg, init_obs_passes=2)
    E, T, A = 2, 12, 24
    for world in (2, 8):
        total = world * E
        acts = np.random.default_rng(world).integers(0, 5, size=(T, total, A)).astype(np.int32)
        big_seeds, big_maps = bench_shard(0, total, len(pool))
        big = [make_oracle_env(comp, comp.encode_map(pool[m]), s) for s, m in zip(big_seeds, big_maps)]
        for t in range(T):
            for g, o in enumerate(big):
                o.step(acts[t, g])
        for r in range(world):
            s, m = bench_shard(r, E, len(pool))
            for e in range(E):
                g = r * E + e
                o = make_oracle_env(comp, comp.encode_map(pool[m[e]]), s[e])
                for t in range(T):
                    o.step(acts[t, g])
                assert np.array_equal(o.obs, big[g].obs) and np.array_equal(o.rewards, big[g].rewards)


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


def _shard_summary(rank, world, E, T, workload="arena"):
    """[envs, env-steps, sum of rewards] of rank's bench shard stepped on the oracle with per-global-id actions: the
    configuration, pool and action range `bench.py --workload <workload>` uses (`bench_workload`; combat at a size the
    oracle steps in seconds, the rollout workload shards exactly like arena)."""
    from metta_amd.compile import compile_game
    from metta_amd.dist import bench_shard, bench_workload
    from oracle.mg_oracle import make_oracle_env
    from tests.helpers import load_map_pool

    wl = bench_workload(workload, small=True)
    pool = wl["maps"] or load_map_pool()
    comp = compile_game(wl["cfg"], init_obs_passes=2)
    acts = np.random.default_rng(5).integers(0, wl["num_actions"], size=(T, 4 * E, wl["agents"])).astype(np.int32)  # by GLOBAL env id
    seeds, maps = bench_shard(rank, E * (1 if world > 1 else 2), len(pool))
    out = np.zeros(3)
    for e, (s, m) in enumerate(zip(seeds, maps)):
        g = rank * E + e
        o = make_oracle_env(comp, comp.encode_map(pool[m]), s)
        for t in range(T):
            o.step(acts[t, g])
            out[2] += float(np.asarray(o.rewards, dtype=np.float64).sum())
        out[0] += 1
        out[1] += T
    return out


def _shard_worker(rank, world, port, q, workload="arena"):
    import torch.distributed as dist

    from metta_amd.dist import all_reduce_summary

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = all_reduce_summary(_shard_summary(rank, world, 3, 40, workload))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out.tolist()))


@pytest.mark.parametrize("workload", ["arena", "combat", "rollout"])
def test_two_ranks_of_E_envs_reduce_to_one_handle_of_2E_envs_gloo(workload):
    """World 2 over gloo: each rank steps its bench shard on the oracle, the summary all-reduce equals the summary of
    one process holding all 2E environments (no data-path collective) -- for every workload `bench.py`
    shards (configs[1], [3], [4])."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q, workload)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=180) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    one = _shard_summary(0, 1, 3, 40, workload)
    assert res[0] == res[1] and res[0][:2] == [6.0, 240.0] == one[:2].tolist()
    assert abs(res[0][2] - one[2]) < 1e-9  # f64 sum of f32 rewards, associated per rank


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
