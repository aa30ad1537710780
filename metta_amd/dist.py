"""Multi-GPU: environments shard across ranks as independent instances; the only collective is the all-reduce of
the finished-episode summary vector (pattern of `metta/rl/advantage.py:127-145`).

One process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm) for device tensors or
"gloo" for the CPU rehearsal used by the tests.  The vector is < 1 KB, so the all-reduce is latency bound.
"""

from __future__ import annotations

import numpy as np


def shard_range(num_envs_total: int, rank: int, world: int) -> tuple[int, int]:
    """Env index range [lo, hi) of `rank`: contiguous blocks, remainders to the low ranks."""
    base, rem = divmod(num_envs_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(base_seed: int, num_envs_total: int, rank: int, world: int) -> np.ndarray:
    """pufferlib convention env i -> seed base + i (PUF/vector.py:90), restricted to this rank's shard."""
    lo, hi = shard_range(num_envs_total, rank, world)
    return ((np.arange(lo, hi, dtype=np.int64) + base_seed) & 0xFFFFFFFF).astype(np.uint32)


def bench_shard(rank: int, envs_per_rank: int, pool_len: int, base_seed: int = 0) -> tuple[list[int], list[int]]:
    """Weak-scaling shard of `bench.py`: rank r owns the global env ids [r*E, (r+1)*E); env g gets seed base + g
    (PUF/vector.py:90) and map g mod pool_len, so N ranks x E envs are exactly one N*E-env handle cut into blocks."""
    lo = rank * envs_per_rank
    ids = range(lo, lo + envs_per_rank)
    return [(base_seed + g) & 0xFFFFFFFF for g in ids], [g % pool_len for g in ids]


def bench_workload(name: str, small: bool = False) -> dict:
    """The workloads `bench.py --workload` shards over ranks (BASELINE configs[1], [3], [4]); `small` = the same
    configuration family at a size the CPU oracle steps in seconds (tests/test_dist_cpu.py).  `maps` is None where the
    caller loads the committed pool of reference-generated arena maps."""
    from .builder import make_arena, make_combat_stress

    if name in ("arena", "rollout"):
        return {
            "name": name, "cfg": make_arena(24), "maps": None, "agents": 24, "envs": 4096, "num_actions": 5,
            # (rollout: one group -- the policy's kernels serialise on the GPU anyway, and two half-size groups double
            # the host's launch work per step: 1.03 against 0.67 ms/step)
            "groups": 4 if name == "arena" else 1,
            "describe": "arena envs x 24 agents per GPU (BASELINE configs[1])" if name == "arena"
            else "arena envs x 24 agents per GPU + LSTM policy forward and sampling in bf16 (BASELINE configs[4])",
        }  # fmt: skip
    if name == "combat":
        agents, side = (64, 32) if small else (1024, 256)
        cfg = make_combat_stress(agents, side)
        return {
            "name": name, "cfg": cfg, "maps": [cfg.game.map_builder.build(np.random.default_rng(i)) for i in range(4)],
            "agents": agents, "envs": 256, "num_actions": 8, "groups": 1,  # (one environment per CU: 64 leave 3/4 of the GPU idle)
            "describe": f"combat-stress envs x {agents} agents on {side}x{side} maps per GPU (BASELINE configs[3])",
        }  # fmt: skip
    raise ValueError(f"unknown workload {name!r}: arena, combat or rollout")


def all_reduce_summary(summary: np.ndarray, device=None) -> np.ndarray:
    """Sum the per-rank summary vector [episodes, sum(len), sum(reward), game stat sums...] over all ranks."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return np.asarray(summary, dtype=np.float64)
    t = torch.as_tensor(np.asarray(summary, dtype=np.float64))
    if dist.get_backend() == "nccl":
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def summary_to_metrics(summary: np.ndarray, game_slot_names: list[str], agents_per_env: int) -> dict:
    """Means per finished episode, keyed like the reference's rollout stats (`metta/rl/stats.py:18-49`)."""
    n = float(summary[0])
    if n <= 0:
        return {"episodes": 0.0}
    out = {
        "episodes": n,
        "episode_length": float(summary[1]) / n,
        "episode_reward_per_agent": float(summary[2]) / (n * agents_per_env),
    }
    for i, name in enumerate(game_slot_names):
        if summary[3 + i] != 0:
            out[f"game/{name}"] = float(summary[3 + i]) / n
    return out
