#!/usr/bin/env python3
"""Throughput of BASELINE configs[2] and [3] (informational; bench.py stays on configs[1]).

  python3 tools/bench_configs.py navigation   # 8192 envs x 4 agents, 64 maps of mixed sizes (60..120)^2
  python3 tools/bench_configs.py combat [steps] [envs]  # 64 (default) envs x 1024 agents on 256 x 256 maps (grid tiles in HBM)
  python3 tools/bench_configs.py mapgen       # 4096 arena envs, 8192 device-generated slots: cost of one pool refresh
  python3 tools/bench_configs.py curriculum   # 4096 arena envs under CurriculumDeviceVecEnv: three tasks, 500-step episodes
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
    if which == "mapgen":
        return mapgen_bench()
    if which == "curriculum":
        return curriculum_bench()
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
    # MG_GROUPS=G: the environments as G independent groups on G streams, like bench.py (default: one handle)
    import os
    G = int(os.environ.get("MG_GROUPS", "1"))
    n = E // G
    bs = [MettaGridBatch(cfg, n, maps, device=0, init_obs_passes=2, seeds=list(range(q * n, (q + 1) * n)),
                         map_ids=[e % len(maps) for e in range(q * n, (q + 1) * n)]) for q in range(G)]  # fmt: skip
    for b in bs:
        b.set_auto_reset(True, 1)
    table = torch.randint(0, n_act, (16, E * A), device="cuda", dtype=torch.int32)
    tabs = [table[:, q * n * A : (q + 1) * n * A].contiguous() for q in range(G)]
    sts = [torch.cuda.Stream() for _ in range(G)] if G > 1 else [torch.cuda.current_stream()]
    ptrs = [[tb[r].data_ptr() for r in range(16)] for tb in tabs]

    def run(k):
        for t in range(k):
            for q, b in enumerate(bs):
                b.step_raw(ptrs[q][t % 16], sts[q].cuda_stream)

    torch.cuda.synchronize()
    run(10)
    torch.cuda.synchronize()
    for b in bs:
        b.set_profiling(True)
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = np.concatenate([b.profile_read(4096) for b in bs])
    print(json.dumps({"workload": which, "envs": E, "agents_per_env": A, "groups": G, "steps": steps, "ms_per_step": dt / steps * 1e3,
                      "kernel_ms_mean": float(ms.mean()), "agent_steps_per_s": E * A * steps / dt}))
    for b in bs:
        b.close()


def mapgen_bench():
    from metta_amd.builder import make_arena
    from metta_amd.mapgen import GeneratedMaps

    cfg = make_arena(24)
    E = 4096
    t0 = time.perf_counter()
    b = MettaGridBatch(cfg, E, GeneratedMaps(cfg.game.map_builder.device_spec(), 2 * E, 0), device=0, seeds=list(range(E)))
    torch.cuda.synchronize()
    t_create = time.perf_counter() - t0
    b.set_auto_reset(True, E)
    table = torch.randint(0, 5, (16, E * 24), device="cuda", dtype=torch.int32)
    acts = table[0]
    out = {"workload": "device map pool, arena 62x62, 4096 envs, 8192 slots", "create_s": t_create}
    for name, force in (("refresh_all_8192_slots_ms", True), ("refresh_nothing_to_do_ms", False)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            b.refresh_maps(force=force)
        ev1.record()
        torch.cuda.synchronize()
        out[name] = ev0.elapsed_time(ev1) / 5
    b.reset(seeds=list(range(E)))
    # steady state: short episodes so that slots turn over; a refresh every 50 steps
    b.end_episode()
    b.step(acts)
    b.step(acts)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    b.refresh_maps()
    ev1.record()
    torch.cuda.synchronize()
    out["refresh_4096_used_idle_slots_ms"] = ev0.elapsed_time(ev1)
    for t in range(20):
        b.step(table[t % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(1000):
        b.step(table[t % 16])
        if t % 100 == 99:
            b.refresh_maps()
    torch.cuda.synchronize()
    out["ms_per_step_with_a_refresh_every_100_steps"] = (time.perf_counter() - t0)  # 1000 steps: s == ms per step
    b.check_errors()
    print(json.dumps(out))
    b.close()


def curriculum_bench():
    import copy

    from metta_amd.builder import make_arena
    from metta_amd.vecenv import CurriculumDeviceVecEnv

    class Task:
        def __init__(self, i, cfg):
            self._task_id, self.cfg, self.n = i, cfg, 0

        def get_env_cfg(self):
            return self.cfg

        def complete(self, score):
            self.n += 1

    class Cur:
        def __init__(self):
            base = make_arena(24)
            base.game.max_steps = 500
            self.tasks = []
            for i, (w, ms) in enumerate(((1.0, 500), (0.5, 400), (0.1, 450))):
                c = copy.deepcopy(base)
                c.game.max_steps = ms
                c.game.agent.rewards.inventory = {"heart": w, "ore_red": 0.1 * i}
                self.tasks.append(Task(i, c))
            self.k = 0

        def get_task(self):
            self.k += 1
            return self.tasks[self.k % 3]

        def update_task_performance(self, task_id, score):
            pass

        def stats(self):
            return {"n": float(self.k)}

    E = 4096
    cur = Cur()
    t0 = time.perf_counter()
    env = CurriculumDeviceVecEnv(cur, E, device=0, seed=0, device_maps=True, early_reset=True)
    env.async_reset()
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    table = torch.randint(0, 5, (16, E * 24), device="cuda", dtype=torch.int32)
    for t in range(50):
        env.recv()
        env.send(table[t % 16])
    torch.cuda.synchronize()
    done0 = env.tasks_completed
    t0 = time.perf_counter()
    steps = 1000
    for t in range(steps):
        env.recv()
        env.send(table[t % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "curriculum: 4096 arena envs, 3 tasks, 400-500 step episodes, device maps, async infos",
                      "setup_s": t_setup, "ms_per_step": dt / steps * 1e3, "agent_steps_per_s": E * 24 * steps / dt,
                      "tasks_completed": env.tasks_completed - done0, "host_syncs": env.host_syncs}))
    env.batch.check_errors()
    env.close()


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
