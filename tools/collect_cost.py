#!/usr/bin/env python3
"""Where the host time of a collection of finished episodes goes (rollout workload, stats_every = 64)."""
import cProfile
import json
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from metta_amd.builder import make_arena  # noqa: E402
from metta_amd.rollout import LSTMPolicy, rollout  # noqa: E402
from metta_amd.vecenv import MettaGridDeviceVecEnv  # noqa: E402

z = np.load(Path(__file__).resolve().parents[1] / "tests" / "golden" / "arena24_map_pool.npz", allow_pickle=False)
names = np.array(json.loads(bytes(z["meta"]).decode())["cell_names"])
pool = [names[m] for m in z["maps"]]
early = "--no-early" not in sys.argv
env = MettaGridDeviceVecEnv(make_arena(24), 4096, maps=pool, device=0, early_reset=early)
env.stats_every = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
env.async_reset(0)
pol = LSTMPolicy(200, env.single_action_space.n).to("cuda").to(torch.bfloat16)
rollout(env, pol, 5)
tot = {"n": 0, "t": 0.0}
orig = env._collect_finished_impl


def timed(*a):
    t0 = time.perf_counter()
    orig(*a)
    tot["n"] += 1
    tot["t"] += time.perf_counter() - t0


env._collect_finished_impl = timed
b = env.batch
for name in ("finished_ring", "finished_batch"):
    def wrap(f, name=name):
        def g(*a, **k):
            t0 = time.perf_counter()
            r = f(*a, **k)
            tot[name] = tot.get(name, 0.0) + time.perf_counter() - t0
            tot[name + "_n"] = tot.get(name + "_n", 0) + (len(r[0]) if name == "finished_ring" else len(a[0]))
            return r
        return g
    setattr(b, name, wrap(getattr(b, name)))
pr = cProfile.Profile()
pr.enable()
n, dt, _, infos = rollout(env, pol, 150)
pr.disable()
print(json.dumps({"ms_per_step": dt / 150 * 1e3, "collections": tot["n"], "collect_ms_total": tot["t"] * 1e3, "infos": len(infos), "detail": {k: v for k, v in tot.items() if k not in ("n", "t")}, "dropped": env.episodes_dropped,
                  "early_reset": early, "stats_every": env.stats_every}))
tr = env.__dict__.get("_trace", [])
print([(a, round((t - tr[0][1]) * 1e3, 2)) for a, t in tr])
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
env.close()
