import sys, json, ctypes as ct
sys.path.insert(0,''+str(__import__("pathlib").Path(__file__).resolve().parents[1])+'')
import numpy as np, torch
from metta_amd.batch import MettaGridBatch
from metta_amd.builder import make_arena
from metta_amd import native
z=np.load(''+str(__import__("pathlib").Path(__file__).resolve().parents[1])+'/tests/golden/arena24_map_pool.npz'); meta=json.loads(bytes(z['meta']).decode()); names=np.array(meta['cell_names']); pool=[names[m] for m in z['maps']]
E=4096
b=MettaGridBatch(make_arena(24),E,pool,seeds=list(range(E)),map_ids=[e%64 for e in range(E)])
tab=torch.randint(0,5,(64,E*24),device='cuda',dtype=torch.int32)
for t in range(50): b.step(tab[t%64])
torch.cuda.synchronize()
native.check(b.L.mg_debug_stamps(b._h, None, 1))
b.step(tab[0]); torch.cuda.synchronize()
out=np.zeros((E,16),np.uint64)
native.check(b.L.mg_debug_stamps(b._h, out.ctypes.data_as(ct.POINTER(ct.c_uint64)), 1))
ok = out[:,0] != 0
print("blocks with stamps:", int(ok.sum()), "of", E)
out = out[ok]
E = int(ok.sum())
t0=int(out[:,0].min())
st = out[:,0].astype(np.int64)-t0; en = out[:,6].astype(np.int64)-t0
print("start offset percentiles [1,10,50,90,99,100]:", np.percentile(st,[1,10,50,90,99,100]))
print("end offset percentiles   [1,10,50,90,99,100]:", np.percentile(en,[1,10,50,90,99,100]))
print("per-block total percentiles [10,50,90,99,100]:", np.percentile(en-st,[10,50,90,99,100]))
print("start offsets (first/second wave of workgroups): median", np.median(out[:,0].astype(np.int64)-t0), "p90", np.percentile(out[:,0].astype(np.int64)-t0,90))
print("uses/no-uses split of stage 2->3:", np.percentile(np.diff(out[:,2:4].astype(np.int64),axis=1),[10,50,90,99]))
d=np.diff(out[:,:7].astype(np.int64),axis=1)
print("phase medians (stamp units):", np.median(d,axis=0), "mean", d.mean(axis=0))
print("total per block median", np.median(out[:,6].astype(np.int64)-out[:,0].astype(np.int64)))
print("kernel span", int(out[:,6].max()-out[:,0].min()))
cnt=out[:,8].astype(np.int64); rdm=out[:,9].astype(np.int64); okc=out[:,10].astype(np.int64)
p2=(out[:,3].astype(np.int64)-out[:,2].astype(np.int64))
# blocks without a use leave slots 8.. from an earlier step: only trust blocks whose P2 was long
has = p2 > 2000
print("blocks with uses:", int(has.sum()), " uses/block mean", cnt[has].mean(), " successes/block mean", okc[has].mean())
for c in range(1,6):
    for r in range(0,3):
        for k in range(0,3):
            sel = has & (cnt==c) & (rdm==r) & (okc==k)
            if sel.sum() > 20: print(f"count {c} rounds {r+1} ok {k}: n={int(sel.sum())} P2 median {np.median(p2[sel]):.0f} p90 {np.percentile(p2[sel],90):.0f}")

tot = out[:,6].astype(np.int64)-out[:,0].astype(np.int64)
idx = np.argsort(-tot)[:12]
print("slowest blocks: total, phases P0 P1 P2 tok enc tail, uses count / ok")
for i in idx:
    print(int(tot[i]), d[i].tolist(), int(cnt[i]) if has[i] else 0, int(okc[i]) if has[i] else 0)
p1 = d[:,1]
print("P1 median: no queued use", np.median(p1[~has]), " p10", np.percentile(p1[~has],10), "| count1", np.median(p1[has&(cnt==1)]), "| count2", np.median(p1[has&(cnt==2)]), "| count3", np.median(p1[has&(cnt==3)]))
print("P0 percentiles [10,50,90,99]:", np.percentile(d[:,0],[10,50,90,99]), " enc:", np.percentile(d[:,4],[10,50,90,99]), " tok:", np.percentile(d[:,3],[10,50,90,99]))
print("P1 percentiles [50,90,99,100]: no queued use", np.percentile(p1[~has],[50,90,99,100]), " with", np.percentile(p1[has],[50,90,99,100]))
big = p1 > 30000
print("blocks with P1 > 30k:", int(big.sum()), " of which with queued uses", int((big&has).sum()), " P0 median of those", np.median(d[big,0]) if big.any() else 0, " count hist", np.bincount(cnt[big&has].astype(int))[:8] if (big&has).any() else 0)

p0a = out[:,11].astype(np.int64)-out[:,0].astype(np.int64); p0c = out[:,12].astype(np.int64)-out[:,11].astype(np.int64); p0d = out[:,1].astype(np.int64)-out[:,12].astype(np.int64)
print("P0 split medians: a (loads issued, env words in)", np.median(p0a), " b+c (restart test, map, rng window, records to LDS)", np.median(p0c), " d (LDS init, barrier)", np.median(p0d))
