"""TEST INFRASTRUCTURE ONLY -- ctypes face of oracle/advantage_oracle.c (CPU restatement of the reference's
`puff_advantage`, packages/pufferlib-core/src/pufferlib/extensions/pufferlib.cpp:28-72) and a loader for the
reference's own compiled CPU extension (oracle/_ref/pufferlib_C.so, `make -C oracle ref_puff`).

Never imported by the product path (metta_amd/): only tests/, __graft_entry__.smoke() and bench tools use it.
"""

from __future__ import annotations

import ctypes as ct
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None
REF_SO = _HERE / "_ref" / "pufferlib_C.so"


def _lib():
    global _LIB
    if _LIB is None:
        path = _HERE / "libadvantage_oracle.so"
        if not path.exists():
            raise ImportError(f"{path} missing: run `make -C oracle cpu`")
        L = ct.CDLL(str(path))
        fp = ct.POINTER(ct.c_float)
        L.oracle_puff_advantage.restype = None
        L.oracle_puff_advantage.argtypes = [fp, fp, fp, fp, fp, ct.c_float, ct.c_float, ct.c_float, ct.c_float, ct.c_int, ct.c_int]
        _LIB = L
    return _LIB


def oracle_advantage(values, rewards, dones, importance, gamma, lam, rho_clip=1.0, c_clip=1.0, advantages=None):
    """numpy f32 [B, T] in, numpy f32 [B, T] out (advantages[:, T-1] keeps the input's value, zero by default)."""
    arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (values, rewards, dones, importance)]
    B, T = arrs[0].shape
    for x in arrs:
        assert x.shape == (B, T)
    out = np.zeros((B, T), np.float32) if advantages is None else np.array(advantages, dtype=np.float32, order="C")
    fp = ct.POINTER(ct.c_float)
    _lib().oracle_puff_advantage(
        *[x.ctypes.data_as(fp) for x in arrs], out.ctypes.data_as(fp), gamma, lam, rho_clip, c_clip, B, T
    )
    return out


_REF_LOADED = False


def reference_available() -> bool:
    return REF_SO.exists()


def reference_advantage(values, rewards, dones, importance, gamma, lam, rho_clip=1.0, c_clip=1.0):
    """The reference's own compiled `compute_puff_advantage` (CPU dispatch) on numpy inputs."""
    import torch

    global _REF_LOADED
    if not _REF_LOADED:
        torch.ops.load_library(str(REF_SO))
        _REF_LOADED = True
    ts = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) for x in (values, rewards, dones, importance)]
    adv = torch.zeros_like(ts[0])
    torch.ops.pufferlib.compute_puff_advantage(*ts, adv, gamma, lam, rho_clip, c_clip)
    return adv.numpy()


def make_case(seed: int, B: int, T: int, kind: str = "rollout"):
    """Seeded synthetic inputs shaped like the trainer's (metta/rl/training/core.py: values, rewards, dones, ratio)."""
    rng = np.random.default_rng(seed)
    values = rng.standard_normal((B, T)).astype(np.float32)
    rewards = (rng.standard_normal((B, T)) * (rng.random((B, T)) < 0.1)).astype(np.float32)
    dones = (rng.random((B, T)) < 0.02).astype(np.float32)
    importance = np.exp(rng.standard_normal((B, T)) * 0.3).astype(np.float32)
    if kind == "edge":  # non-binary dones, huge / tiny / special importance weights
        dones = rng.random((B, T)).astype(np.float32)
        importance[rng.random((B, T)) < 0.05] = np.float32(np.inf)
        importance[rng.random((B, T)) < 0.05] = np.float32(np.nan)
        importance[rng.random((B, T)) < 0.05] = np.float32(1e-30)
        values[rng.random((B, T)) < 0.02] = np.float32(1e30)
    return values, rewards, dones, importance
