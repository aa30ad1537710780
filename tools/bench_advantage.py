"""Advantage kernel timing: HIP kernel against the HBM roofline (20 B per element), with the
pure-torch horizon loop the reference runs on ROCm (metta/rl/mps.py:5-41, restated here) and the C oracle beside it.

usage: python tools/bench_advantage.py [rows horizon]...   (default: the trainer's 8192 x 256 and 65536 x 256)
"""

import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from metta_amd.advantage import compute_puff_advantage  # noqa: E402
from oracle.advantage_oracle import make_case, oracle_advantage  # noqa: E402


def torch_loop(values, rewards, dones, w, rho_clip, c_clip, gamma, lam):
    B, T = rewards.shape
    adv = torch.zeros_like(values)
    rho, c = torch.clamp(w, max=rho_clip), torch.clamp(w, max=c_clip)
    nn = 1.0 - dones[:, 1:]
    delta = rho[:, :-1] * (rewards[:, 1:] + gamma * values[:, 1:] * nn - values[:, :-1])
    gl = gamma * lam
    last = torch.zeros(B, device=values.device)
    for t in range(T - 2, -1, -1):
        last = delta[:, t] + gl * c[:, t] * last * nn[:, t]
        adv[:, t] = last
    return adv


def main():
    args = [int(x) for x in sys.argv[1:]]
    shapes = list(zip(args[::2], args[1::2])) or [(8192, 256), (65536, 256)]
    for B, T in shapes:
        ins = make_case(1, B, T)
        ts = [torch.from_numpy(x).cuda() for x in ins]
        adv = torch.zeros_like(ts[0])
        hp = (0.977, 0.916, 1.0, 1.0)
        for _ in range(5):
            compute_puff_advantage(*ts, adv, *hp)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 50
        ev[0].record()
        for _ in range(n):
            compute_puff_advantage(*ts, adv, *hp)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / n
        torch_loop(*ts, 1.0, 1.0, hp[0], hp[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref = torch_loop(*ts, 1.0, 1.0, hp[0], hp[1])
        torch.cuda.synchronize()
        loop_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        ora = oracle_advantage(*ins, *hp)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        exact = bool(np.array_equal(adv.cpu().numpy().view(np.uint32), ora.view(np.uint32)))
        close = bool(torch.allclose(adv, ref, rtol=1e-4, atol=1e-5))
        gbs = B * T * 20 / (ms * 1e-3) / 1e9
        print(json.dumps({"rows": B, "horizon": T, "hip_ms": round(ms, 4), "algorithmic_GBps": round(gbs, 1),
                          "frac_of_8TBps": round(gbs / 8000, 3), "torch_loop_ms": round(loop_ms, 2),
                          "oracle_1core_ms": round(cpu_ms, 2), "bit_exact_vs_oracle": exact, "close_to_torch_loop": close}))


if __name__ == "__main__":
    main()
