"""Phase stamps of the general (BIG) act and observation kernels on the combat workload."""
import ctypes as ct
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from metta_amd import native  # noqa: E402
from metta_amd.batch import MettaGridBatch  # noqa: E402
from metta_amd.builder import make_combat_stress  # noqa: E402

A, E = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = make_combat_stress(A, 256)
maps = [cfg.game.map_builder.build(np.random.default_rng(i)) for i in range(4)]
b = MettaGridBatch(cfg, E, maps, device=0, init_obs_passes=2, seeds=list(range(E)), map_ids=[e % 4 for e in range(E)])
b.set_auto_reset(True, 1)
tab = torch.randint(0, 8, (16, E * A), device="cuda", dtype=torch.int32)
for t in range(30):
    b.step(tab[t % 16])
torch.cuda.synchronize()
print(b.step_path())
native.check(b.L.mg_debug_stamps(b._h, None, 1))
b.step(tab[0])
torch.cuda.synchronize()
out = np.zeros((E, 16), np.uint64)
native.check(b.L.mg_debug_stamps(b._h, out.ctypes.data_as(ct.POINTER(ct.c_uint64)), 1))
out = out.astype(np.int64)
act = out[:, 7:14]
d = np.diff(act, axis=1)
print("act kernel phases (stage | overlay+rng | shuffle | action rounds | regen+damage | write-back), medians:", np.median(d, axis=0), "total", np.median(act[:, -1] - act[:, 0]))
print("first action round:", np.median(out[:, 15] - out[:, 10]), "later rounds together:", np.median(out[:, 11] - out[:, 15]))
print("action rounds [min, median, max]:", out[:, 14].min(), np.median(out[:, 14]), out[:, 14].max())
obs = out[:, 0:5]
d = np.diff(obs, axis=1)
print("obs kernel phases (stage | overlay | tokens | encode), medians:", np.median(d, axis=0), "total to encode end", np.median(obs[:, -1] - obs[:, 0]))
