"""Shared helpers for the parity tests (fixtures, fuzz configs, comparisons)."""

from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden"


def digest(a: np.ndarray) -> int:
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(), "little")


def load_fixture(name: str):
    z = np.load(GOLD / f"{name}.npz", allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def fixture_grid(z, meta) -> np.ndarray:
    names = np.array(meta["cell_names"])
    return names[z["map_codes"]]


def load_map_pool():
    z = np.load(GOLD / "arena24_map_pool.npz", allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    names = np.array(meta["cell_names"])
    return [names[m] for m in z["maps"]]


def fixture_actions(meta, num_agents: int) -> np.ndarray:
    spec = meta["action_spec"]
    seed = meta["seed"]
    T = spec["steps"]
    if spec["kind"] == "RandomState.randint":
        return np.random.RandomState(seed).randint(spec["low"], spec["high"], size=(T, num_agents)).astype(np.int32)
    if spec["kind"] == "nav_mixed":
        a = np.random.RandomState(seed).randint(0, spec["n_actions"], size=(T, num_agents)).astype(np.int32)
        a[np.random.RandomState(seed + 1).rand(T, num_agents) < 0.7] %= 5
        return a
    raise ValueError(spec["kind"])


def stats_equal(a: dict, b: dict, ignore_invalid_variants: bool = True) -> str | None:
    """Compare two get_episode_stats() dicts; returns a description of the first difference or None."""

    def norm(d):
        if not ignore_invalid_variants:
            return d
        return {k: v for k, v in d.items() if not k.startswith("action.invalid_index.")}

    if a["game"] != b["game"]:
        ks = set(a["game"]) | set(b["game"])
        return "game: " + str({k: (a["game"].get(k), b["game"].get(k)) for k in ks if a["game"].get(k) != b["game"].get(k)})
    for i, (x, y) in enumerate(zip(a["agent"], b["agent"])):
        x, y = norm(x), norm(y)
        if x != y:
            ks = set(x) | set(y)
            return f"agent {i}: " + str({k: (x.get(k), y.get(k)) for k in ks if x.get(k) != y.get(k)})
    return None
