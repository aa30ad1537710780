"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build container only).

    make -C oracle ref && python oracle/gen_golden.py

Imports the reference's Python package from /root/reference (config presets, map generation, the real
`convert_to_cpp_game_config`) on top of the reference's C++ core compiled by oracle/Makefile, runs seeded
episodes and records inputs + expected outputs.  Nothing of the reference is copied: a fixture holds a map
(cell names as a small string table + uint8 codes), the seed, the action stream's generator parameters, and
expected outputs (per-step 64-bit digests of the observation buffer, rewards, terminal flags, final
observation, final agent state and `get_episode_stats()`).

Harness-side import shims (ordinary ImportErrors on Python 3.10, not denials -- SURVEY.md §8c):
`enum.StrEnum`, `typing.Self/override`, stub `boto3/botocore` modules, and `mettagrid.mettagrid_c` bound to
oracle/_ref's build.
"""

from __future__ import annotations

import enum
import hashlib
import json
import sys
import types
import typing
from pathlib import Path

import numpy as np
import typing_extensions

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"


def install_reference_shims() -> None:
    if not hasattr(enum, "StrEnum"):

        class StrEnum(str, enum.Enum):
            def __str__(self):
                return str(self.value)

            @staticmethod
            def _generate_next_value_(name, start, count, last_values):
                return name.lower()

        enum.StrEnum = StrEnum
    for n in ("Self", "override", "Never", "assert_never", "Unpack", "NotRequired", "Required"):
        if not hasattr(typing, n):
            setattr(typing, n, getattr(typing_extensions, n))

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, n):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    for name in ("boto3", "botocore", "botocore.exceptions", "botocore.config", "botocore.client"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(
                ClientError=Exception, NoCredentialsError=Exception, Config=_Any, client=_Any(), BaseClient=_Any,
                __getattr__=lambda n: _Any(),
            )  # fmt: skip
            sys.modules[name] = m
    from oracle.ref_bridge import load_reference_module

    sys.modules["mettagrid.mettagrid_c"] = load_reference_module()
    sys.path.insert(0, "/root/reference/packages/mettagrid/python/src")


def digest(a: np.ndarray) -> int:
    return int.from_bytes(hashlib.blake2b(np.ascontiguousarray(a).tobytes(), digest_size=8).digest(), "little")


def encode_names(grid: np.ndarray) -> tuple[list[str], np.ndarray]:
    names = sorted(set(str(x) for x in np.unique(grid)))
    idx = {n: i for i, n in enumerate(names)}
    codes = np.vectorize(lambda s: idx[str(s)], otypes=[np.uint8])(grid)
    return names, codes


def run_reference(ref_cfg, grid, seed: int, actions: np.ndarray, snapshot_steps=()):
    """Drive the reference through its own `Simulation`-equivalent path (ctor + set_buffers)."""
    from mettagrid.config.mettagrid_c_config import convert_to_cpp_game_config
    from mettagrid.mettagrid_c import MettaGrid

    c_cfg = convert_to_cpp_game_config(ref_cfg.game.model_dump())
    env = MettaGrid(c_cfg, grid.tolist(), seed)
    A, T = ref_cfg.game.num_agents, ref_cfg.game.obs.num_tokens
    obs = np.zeros((A, T, 3), np.uint8)
    term = np.zeros(A, bool)
    trunc = np.zeros(A, bool)
    rew = np.zeros(A, np.float32)
    act = np.zeros(A, np.int32)
    env.set_buffers(obs, term, trunc, rew, act)
    out = {
        "obs0": obs.copy(),
        "obs_digest": np.zeros(len(actions), np.uint64),
        "rew_digest": np.zeros(len(actions), np.uint64),
        "succ_digest": np.zeros(len(actions), np.uint64),
        "rew_sum": np.zeros(len(actions), np.float64),
        "done_step": -1,
    }
    snaps = {}
    for t in range(len(actions)):
        act[:] = actions[t]
        env.step()
        out["obs_digest"][t] = digest(obs)
        out["rew_digest"][t] = digest(rew)
        out["succ_digest"][t] = digest(np.array(env.action_success(), dtype=np.uint8))
        out["rew_sum"][t] = float(rew.astype(np.float64).sum())
        if t + 1 in snapshot_steps:
            snaps[t + 1] = obs.copy()
        if (term.all() or trunc.all()) and out["done_step"] < 0:
            out["done_step"] = t + 1
    out["obs_final"] = obs.copy()
    out["episode_rewards"] = np.array(env.get_episode_rewards(), dtype=np.float32)
    out["stats"] = env.get_episode_stats()
    go = env.grid_objects()
    agents = sorted((o for o in go.values() if "agent_id" in o), key=lambda o: o["agent_id"])
    out["agent_rc"] = np.array([[o["r"], o["c"]] for o in agents], np.int32)
    out["agent_inv"] = [[[int(k), int(v)] for k, v in o["inventory"].items()] for o in agents]
    out["snaps"] = snaps
    return out, env


def save_fixture(name: str, grid: np.ndarray, seed: int, action_spec: dict, out: dict, extra: dict | None = None) -> None:
    names, codes = encode_names(grid)
    meta = {
        "seed": seed,
        "cell_names": names,
        "action_spec": action_spec,
        "done_step": int(out["done_step"]),
        "stats": out["stats"],
        "agent_inv": out["agent_inv"],
        "snap_steps": sorted(out["snaps"].keys()),
    }
    if extra:
        meta.update(extra)
    arrays = dict(
        map_codes=codes,
        obs0=out["obs0"],
        obs_final=out["obs_final"],
        obs_digest=out["obs_digest"],
        rew_digest=out["rew_digest"],
        succ_digest=out["succ_digest"],
        rew_sum=out["rew_sum"],
        episode_rewards=out["episode_rewards"],
        agent_rc=out["agent_rc"],
    )
    for s, o in out["snaps"].items():
        arrays[f"snap_{s}"] = o
    GOLD.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLD / f"{name}.npz", meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"wrote {name}.npz ({(GOLD / f'{name}.npz').stat().st_size} bytes)")


def main() -> None:
    install_reference_shims()
    from mettagrid.builder.envs import make_arena, make_navigation

    # --- config 1: arena-24, one full 10 000-step episode incl. termination, seeds {0, 1, 42} ----
    for seed in (0, 1, 42):
        cfg = make_arena(24)
        cfg.game.map_builder.seed = 1000 + seed
        grid = cfg.game.map_builder.create().build_for_num_agents(24).grid
        T = 10000 if seed == 0 else 2500
        actions = np.random.RandomState(seed).randint(0, 5, size=(T, 24)).astype(np.int32)
        out, _ = run_reference(cfg, grid, seed, actions, snapshot_steps=(1, 100, 1000))
        save_fixture(f"arena24_seed{seed}", grid, seed, {"kind": "RandomState.randint", "low": 0, "high": 5, "steps": T}, out)

    # --- map pool for config 2: 64 reference-generated arena maps (inputs only) -----------------------------
    maps = []
    names_all = None
    for i in range(64):
        cfg = make_arena(24)
        cfg.game.map_builder.seed = 5000 + i
        grid = cfg.game.map_builder.create().build_for_num_agents(24).grid
        names, codes = encode_names(grid)
        if names_all is None:
            names_all = names
        assert names == names_all
        maps.append(codes)
    GOLD.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(
        GOLD / "arena24_map_pool.npz",
        maps=np.stack(maps),
        meta=np.frombuffer(json.dumps({"cell_names": names_all}).encode(), dtype=np.uint8),
    )
    print("wrote arena24_map_pool.npz")

    # --- config 3 flavour: navigation (157 actions incl. change_vibe), sparse random map -----------------------
    for seed in (3,):
        cfg = make_navigation(4)
        mb = cfg.game.map_builder
        mb.width, mb.height, mb.border_width, mb.seed = 40, 32, 1, 77
        mb.objects = {"assembler": 6, "wall": 60}
        cfg.game.max_steps = 500
        grid = mb.create().build().grid
        T = 600
        nA = len(cfg.game.actions.actions())
        actions = np.random.RandomState(seed).randint(0, nA, size=(T, 4)).astype(np.int32)
        actions[np.random.RandomState(seed + 1).rand(T, 4) < 0.7] %= 5
        out, _ = run_reference(cfg, grid, seed, actions, snapshot_steps=(1, 250))
        save_fixture(
            f"navigation4_seed{seed}", grid, seed,
            {"kind": "nav_mixed", "n_actions": nA, "steps": T}, out, extra={"max_steps": 500, "map": [40, 32, 1]},
        )  # fmt: skip


if __name__ == "__main__":
    main()
