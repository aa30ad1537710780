import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from metta_amd.batch import MettaGridBatch
from metta_amd.builder import make_arena
z = np.load("tests/golden/arena24_map_pool.npz"); meta = json.loads(bytes(z["meta"]).decode()); names = np.array(meta["cell_names"]); pool = [names[m] for m in z["maps"]]
for G in (1, 2, 4, 8):
    n = 8
    bs = [MettaGridBatch(make_arena(24), n, pool, seeds=list(range(n)), map_ids=[0]*n) for _ in range(G)]
    tabs = [torch.randint(0, 5, (16, n*24), device="cuda", dtype=torch.int32) for _ in range(G)]
    sts = [torch.cuda.Stream() for _ in range(G)]
    step = MettaGridBatch.group_stepper(bs, [[tb[r].data_ptr() for r in range(16)] for tb in tabs], [s.cuda_stream for s in sts])
    for t in range(200): step(t % 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(3000): step(t % 16)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"G={G}: enqueue {1e6*(t1-t0)/3000:.1f} us/step ({1e6*(t1-t0)/3000/G:.1f} per launch), with drain {1e6*(t2-t0)/3000:.1f}")
    for b in bs: b.close()
