"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/advantage_cases.npz from the reference itself (this container only).

Two reference outputs per case:
  adv_ref_<i>  the reference's compiled CPU extension `torch.ops.pufferlib.compute_puff_advantage`
               (oracle/_ref/pufferlib_C.so = packages/pufferlib-core/src/pufferlib/extensions/pufferlib.cpp built by
               `make -C oracle ref_puff`): the bit-exact target;
  adv_mps_<i>  `metta.rl.mps.advantage` (metta/rl/mps.py:5-41, the pure-torch loop the reference runs on ROCm),
               loaded from /root/reference by file path: same recurrence, different rounding (gamma*lambda folded
               in double, no per-step double subtraction): the tolerance target.

Usage: python oracle/gen_golden_advantage.py
"""

import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle.advantage_oracle import make_case, reference_advantage  # noqa: E402

CASES = [  # (seed, B, T, kind, gamma, lambda, rho_clip, c_clip)
    (0, 5, 16, "rollout", 0.99, 0.95, 1.0, 1.0),
    (1, 33, 64, "rollout", 0.977, 0.916, 1.0, 1.0),  # the trainer's defaults (metta/rl/trainer_config.py)
    (2, 17, 67, "rollout", 0.99, 0.9, 0.8, 1.3),  # horizon not a multiple of 4 or of the tile
    (3, 40, 256, "rollout", 0.977, 0.916, 1.0, 1.0),  # bptt_horizon 256
    (4, 9, 130, "edge", 0.95, 0.8, 2.0, 0.5),
    (5, 3, 2, "rollout", 0.99, 0.95, 1.0, 1.0),
    (6, 4, 1, "rollout", 0.99, 0.95, 1.0, 1.0),
]


def main():
    spec = importlib.util.spec_from_file_location("ref_mps", "/root/reference/metta/rl/mps.py")
    mps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mps)
    out = {"cases": np.array(CASES, dtype=object).astype(str)}
    for i, (seed, B, T, kind, gamma, lam, rho, c) in enumerate(CASES):
        v, r, d, w = make_case(seed, B, T, kind)
        out[f"values_{i}"], out[f"rewards_{i}"], out[f"dones_{i}"], out[f"importance_{i}"] = v, r, d, w
        out[f"adv_ref_{i}"] = reference_advantage(v, r, d, w, gamma, lam, rho, c)
        with np.errstate(all="ignore"):
            a = mps.advantage(*[torch.from_numpy(x) for x in (v, r, d, w)], rho, c, gamma, lam, torch.device("cpu"))
        out[f"adv_mps_{i}"] = a.numpy()
    path = ROOT / "tests" / "golden" / "advantage_cases.npz"
    np.savez_compressed(path, **out)
    print("wrote", path, path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
