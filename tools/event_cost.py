"""Cost of the per-step HIP event pair (mg_set_profiling) on the wall clock of the bench loop: same loop with and without."""
import json, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from metta_amd.batch import MettaGridBatch
from metta_amd.builder import make_arena

E, A = 4096, 24
cfg = make_arena(A); pool = bench._load_pool()
b = MettaGridBatch(cfg, E, pool, device=0, init_obs_passes=2, seeds=list(range(E)), map_ids=[e % len(pool) for e in range(E)])
b.set_auto_reset(True, 1)
b.set_early_reset(np.random.RandomState(0).randint(1, cfg.game.max_steps + 1, size=E).astype(np.uint32))
table = torch.randint(0, 5, (64, E * A), device="cuda", dtype=torch.int32)
def run(k):
    for t in range(k): b.step(table[t % 64])
run(50); torch.cuda.synchronize()
for mode in (False, True, False, True):
    b.set_profiling(mode)
    t0 = time.perf_counter(); run(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if mode: b.profile_read(4096)
    print(json.dumps({"profiling": mode, "ms_per_step": dt / 400 * 1e3}))
b.close()
