"""Minimal `Box` / `Discrete` spaces (the reference uses gymnasium's, `PY/policy/policy_env_interface.py:109-112`).
gymnasium is used when importable; these stand-ins keep the vecenv importable without it."""

from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box, Discrete, MultiDiscrete  # type: ignore
except Exception:  # noqa: BLE001

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low = np.full(shape, low, dtype=dtype)
            self.high = np.full(shape, high, dtype=dtype)
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

        def __repr__(self) -> str:
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete:  # type: ignore[no-redef]
        def __init__(self, n: int, start: int = 0):
            self.n = int(n)
            self.start = int(start)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def contains(self, x) -> bool:
            return self.start <= int(x) < self.start + self.n

        def __repr__(self) -> str:
            return f"Discrete({self.n})"

    class MultiDiscrete:  # type: ignore[no-redef]
        def __init__(self, nvec):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape
            self.dtype = np.dtype(np.int64)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            return x.shape == self.shape and bool((x >= 0).all() and (x < self.nvec).all())


def joint_space(space, n: int):
    """pufferlib.spaces.joint_space: the space of `n` agents' values."""
    if isinstance(space, Discrete):
        return MultiDiscrete([space.n] * n)
    return Box(low=float(space.low.min()), high=float(space.high.max()), shape=(n, *space.shape), dtype=space.dtype)
