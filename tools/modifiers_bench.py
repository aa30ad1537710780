#!/usr/bin/env python3
"""Arena with inventory-limit modifiers (the fused kernel's build with the nested limit enforcement), 4096 environments
as G groups: MG_FUSED_O7=0 forces the 8-waves build, 1 the 6-waves build.   python3 tools/modifiers_bench.py [G]"""
import json, os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from metta_amd.batch import MettaGridBatch
from metta_amd.builder import make_arena
from metta_amd.config import ResourceLimitsConfig
z = np.load(Path(__file__).resolve().parents[1] / "tests/golden/arena24_map_pool.npz")
meta = json.loads(bytes(z["meta"]).decode()); names = np.array(meta["cell_names"]); pool = [names[m] for m in z["maps"]]
cfg = make_arena(24)
cfg.game.agent.inventory.limits["ore"] = ResourceLimitsConfig(limit=2, resources=["ore_red"], modifiers={"battery_red": 2})
E, G = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = E // G
bs = [MettaGridBatch(cfg, n, pool, seeds=list(range(q * n, (q + 1) * n)), map_ids=[e % 64 for e in range(q * n, (q + 1) * n)]) for q in range(G)]
tabs = [torch.randint(0, 5, (16, n * 24), device="cuda", dtype=torch.int32) for _ in range(G)]
sts = [torch.cuda.Stream() for _ in range(G)]
step = MettaGridBatch.group_stepper(bs, [[tb[r].data_ptr() for r in range(16)] for tb in tabs], [s.cuda_stream for s in sts])
for t in range(50): step(t % 16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(300): step(t % 16)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
print(json.dumps({"workload": "arena + limit modifiers", "envs": E, "groups": G, "kernel": bs[0].step_path(), "ms_per_step": dt * 1e3, "agent_steps_per_s": E * 24 / dt}))
