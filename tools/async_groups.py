#!/usr/bin/env python3
"""Arena workload stepped as G independent groups on G HIP streams (the reference's async_factor double buffering,
training_environment.py:45-46: a group's step k+1 is ordered after its own step k only), against one group.
  python3 tools/async_groups.py [total_envs] [groups ...]"""
import json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from metta_amd.batch import MettaGridBatch
from metta_amd.builder import make_arena
z = np.load(Path(__file__).resolve().parents[1] / "tests/golden/arena24_map_pool.npz")
meta = json.loads(bytes(z["meta"]).decode()); names = np.array(meta["cell_names"]); pool = [names[m] for m in z["maps"]]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for G in [int(x) for x in sys.argv[2:]] or [1, 2, 4]:
    n = E // G
    bs = [MettaGridBatch(make_arena(24), n, pool, seeds=list(range(g * n, (g + 1) * n)), map_ids=[e % 64 for e in range(n)]) for g in range(G)]
    for b in bs:
        b.set_early_reset(np.random.RandomState(0).randint(1, 1001, size=n).astype(np.uint32))
    tabs = [torch.randint(0, 5, (64, n * 24), device="cuda", dtype=torch.int32) for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    torch.cuda.synchronize()
    def run(steps):
        for t in range(steps):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    bs[g].step(tabs[g][t % 64])
    run(100)
    torch.cuda.synchronize()
    for off in ([0] if G == 1 else [0, 1]):
        if off:  # start the groups out of phase: group g waits g/G of a step
            for g in range(1, G):
                with torch.cuda.stream(streams[g]):
                    torch.cuda._sleep(int(2.1e9 * 65e-6 * g / G))
        t0 = time.perf_counter(); run(600); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 600
        print(json.dumps({"envs": E, "groups": G, "offset": off, "us_per_step_all_groups": dt * 1e6,
                          "agent_steps_per_s": E * 24 / dt}), flush=True)
    for b in bs: b.close()
