import sys, time, json
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from metta_amd.batch import MettaGridBatch
from metta_amd.builder import make_arena
z=np.load('/root/repo/tests/golden/arena24_map_pool.npz'); meta=json.loads(bytes(z['meta']).decode()); names=np.array(meta['cell_names']); pool=[names[m] for m in z['maps']]
for E in (64, 4096):
    b=MettaGridBatch(make_arena(24),E,pool,seeds=list(range(E)),map_ids=[e%64 for e in range(E)])
    tab=torch.randint(0,5,(64,E*24),device='cuda',dtype=torch.int32)
    for t in range(50): b.step(tab[t%64])
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for t in range(1000): b.step(tab[t%64])
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(E,'enqueue us/step',(t1-t0)*1e3,'total us/step',(t2-t0)*1e3)
    b.close()
