#!/usr/bin/env python3
"""Step time of the arena workload against the number of environments (latency floor vs throughput slope).
  python3 tools/sweep.py [E ...]      (MG_DIAG / MG_NO_FUSED are read from the environment by the library)"""
import json, os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from metta_amd.batch import MettaGridBatch
from metta_amd.builder import make_arena
z = np.load(Path(__file__).resolve().parents[1] / "tests/golden/arena24_map_pool.npz")
meta = json.loads(bytes(z["meta"]).decode()); names = np.array(meta["cell_names"]); pool = [names[m] for m in z["maps"]]
Es = [int(x) for x in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192]
for E in Es:
    b = MettaGridBatch(make_arena(24), E, pool, seeds=list(range(E)), map_ids=[e % 64 for e in range(E)])
    b.set_early_reset(np.random.RandomState(0).randint(1, 1001, size=E).astype(np.uint32))
    tab = torch.randint(0, 5, (64, E * 24), device="cuda", dtype=torch.int32)
    for t in range(100): b.step(tab[t % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(400): b.step(tab[t % 64])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 400
    print(json.dumps({"E": E, "diag": os.environ.get("MG_DIAG", "0"), "us_per_step": dt * 1e6, "path": b.step_path()[:40]}), flush=True)
    b.close()
