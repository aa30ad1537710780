"""Phase stamps of the fused kernel on the navigation workload (8192 envs x 4 agents, one handle)."""
import ctypes as ct
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from metta_amd import native  # noqa: E402
from metta_amd.batch import MettaGridBatch  # noqa: E402
from metta_amd.builder import make_navigation  # noqa: E402
from metta_amd.mapgen import RandomMapBuilder  # noqa: E402

rs = np.random.RandomState(0)
A, E = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cfg = make_navigation(A)
maps = []
for i in range(64):
    h, w = int(rs.randint(60, 121)), int(rs.randint(60, 121))
    mb = RandomMapBuilder(width=w, height=h, border_width=1, agents=A, objects={"wall": (h * w) // 10, "assembler": int(rs.randint(1, 11))})
    maps.append(mb.build(np.random.default_rng(i)))
b = MettaGridBatch(cfg, E, maps, device=0, init_obs_passes=2, seeds=list(range(E)), map_ids=[e % 64 for e in range(E)])
b.set_auto_reset(True, 1)
tab = torch.randint(0, 5, (16, E * A), device="cuda", dtype=torch.int32)
for t in range(50):
    b.step(tab[t % 16])
torch.cuda.synchronize()
print(b.step_path())
native.check(b.L.mg_debug_stamps(b._h, None, 1))
b.step(tab[0])
torch.cuda.synchronize()
out = np.zeros((E, 16), np.uint64)
native.check(b.L.mg_debug_stamps(b._h, out.ctypes.data_as(ct.POINTER(ct.c_uint64)), 1))
ok = out[:, 0] != 0
out = out[ok].astype(np.int64)
t0 = out[:, 0].min()
print("blocks", int(ok.sum()), "kernel span", int(out[:, 6].max() - t0))
print("start offsets [1,50,90,99,100]", np.percentile(out[:, 0] - t0, [1, 50, 90, 99, 100]))
print("end offsets   [1,50,90,99,100]", np.percentile(out[:, 6] - t0, [1, 50, 90, 99, 100]))
d = np.diff(out[:, :7], axis=1)
print("phase medians P0 P1 P2 tok enc tail:", np.median(d, axis=0), "means", d.mean(axis=0).round(0))
print("per-block total [10,50,90,99,100]", np.percentile(out[:, 6] - out[:, 0], [10, 50, 90, 99, 100]))
p0a = out[:, 11] - out[:, 0]; p0c = out[:, 12] - out[:, 11]; p0d = out[:, 1] - out[:, 12]
print("P0 split medians a/bc/d:", np.median(p0a), np.median(p0c), np.median(p0d))
