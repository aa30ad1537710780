#!/usr/bin/env python3
"""Throughput of BASELINE configs[2] and [3] (informational; bench.py stays on configs[1]).

  python3 tools/bench_configs.py navigation   # 8192 envs x 4 agents, 64 maps of mixed sizes (60..120)^2
  python3 tools/bench_configs.py combat [steps] [envs]  # 64 (default) envs x 1024 agents on 256 x 256 maps (grid tiles in HBM)
  python3 tools/bench_configs.py rollout      # configs[4] on one GPU: 4096 arena envs x 24 agents + LSTM policy (bf16)
"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from metta_amd.batch import MettaGridBatch  # noqa: E402
from metta_amd.builder import make_combat_stress, make_navigation  # noqa: E402
from metta_amd.mapgen import RandomMapBuilder  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "navigation"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rs = np.random.RandomState(0)
    if which == "rollout":
        return rollout_bench(steps)
    if which == "navigation":
        A, E = 4, 8192
        cfg = make_navigation(A)
        maps = []
        for i in range(64):
            h, w = int(rs.randint(60, 121)), int(rs.randint(60, 121))
            mb = RandomMapBuilder(width=w, height=h, border_width=1, agents=A,
                                  objects={"wall": (h * w) // 10, "assembler": int(rs.randint(1, 11))})
            maps.append(mb.build(np.random.default_rng(i)))
        n_act = 5
    else:
        A, E = 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 64
        cfg = make_combat_stress(A, 256)
        maps = [cfg.game.map_builder.build(np.random.default_rng(i)) for i in range(4)]
        n_act = 8
    b = MettaGridBatch(cfg, E, maps, device=0, init_obs_passes=2, seeds=list(range(E)), map_ids=[e % len(maps) for e in range(E)])
    b.set_auto_reset(True, 1)
    table = torch.randint(0, n_act, (16, E * A), device="cuda", dtype=torch.int32)
    for t in range(10):
        b.step(table[t % 16])
    torch.cuda.synchronize()
    b.set_profiling(True)
    t0 = time.perf_counter()
    for t in range(steps):
        b.step(table[t % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = b.profile_read(4096)
    print(json.dumps({"workload": which, "envs": E, "agents_per_env": A, "steps": steps, "ms_per_step": dt / steps * 1e3,
                      "kernel_ms_mean": float(ms.mean()), "agent_steps_per_s": E * A * steps / dt}))
    b.close()


def rollout_bench(steps):
    from metta_amd.builder import make_arena
    from metta_amd.rollout import LSTMPolicy, rollout
    from metta_amd.vecenv import MettaGridDeviceVecEnv

    z = np.load(Path(__file__).resolve().parents[1] / "tests" / "golden" / "arena24_map_pool.npz", allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    names = np.array(meta["cell_names"])
    pool = [names[m] for m in z["maps"]]
    cfg = make_arena(24)
    env = MettaGridDeviceVecEnv(cfg, 4096, maps=pool, device=0)
    env.stats_every = 64
    env.async_reset(0)
    pol = LSTMPolicy(200, env.single_action_space.n).to("cuda").to(torch.bfloat16)
    rollout(env, pol, 5)
    n, dt, _, infos = rollout(env, pol, steps)
    print(json.dumps({"workload": "rollout (env step + LSTM policy forward + sampling, bf16)", "envs": 4096,
                      "agents_per_env": 24, "steps": steps, "ms_per_step": dt / steps * 1e3,
                      "agent_steps_per_s": n / dt, "episode_infos": len(infos)}))
    env.close()


if __name__ == "__main__":
    main()
