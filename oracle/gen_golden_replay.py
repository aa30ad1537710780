"""Replay fixture from the reference's own replay writer (run in the build container only).

    make -C oracle ref && python oracle/gen_golden_replay.py

Drives the reference's `Simulation` with its `InMemoryReplayWriter` handler
(python/src/mettagrid/simulator/replay_log_writer.py:21-46,93-226) over one short arena episode and stores the replay
tree it builds (`EpisodeReplay.get_replay_data()`, without the embedded `mg_config` dump, which is the reference's
schema) together with the map, seed and action stream that produced it.
"""

from __future__ import annotations

import json
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle.gen_golden import GOLD, encode_names, install_reference_shims  # noqa: E402
from oracle.gen_golden_infos import jsonable  # noqa: E402


def main() -> None:
    install_reference_shims()
    from mettagrid.builder.envs import make_arena
    from mettagrid.simulator.replay_log_writer import InMemoryReplayWriter
    from mettagrid.simulator.simulator import Simulation

    cfg = make_arena(24)
    steps, seed = 48, 21
    cfg.game.max_steps = steps
    cfg.game.map_builder.seed = 777
    writer = InMemoryReplayWriter()
    sim = Simulation(cfg, seed=seed, event_handlers=[writer])
    A = cfg.game.num_agents
    grid = cfg.game.map_builder.create().build_for_num_agents(A).grid
    nA = len(sim.action_names)
    rs = np.random.RandomState(seed)
    actions = []
    while not sim.is_done():
        acts = np.where(rs.rand(A) < 0.8, rs.randint(0, 5, size=A), rs.randint(0, nA, size=A)).astype(np.int32)
        sim._c_sim.actions()[:] = acts
        actions.append(acts.tolist())
        sim.step()
    replay = writer.get_completed_replays()[0].get_replay_data()
    tree = jsonable({k: v for k, v in replay.items() if k != "mg_config"})
    names, codes = encode_names(grid)

    def key_paths(x, prefix=""):  # the shape of the dump: every key path, list positions folded
        if isinstance(x, dict):
            out = set()
            for k, v in x.items():
                out |= key_paths(v, f"{prefix}/{k}")
            return out or {prefix}
        if isinstance(x, list):
            out = set()
            for v in x:
                out |= key_paths(v, prefix + "[]")
            return out or {prefix}
        return {prefix}

    mgc = jsonable(replay["mg_config"])
    game = mgc["game"]
    actions_cfg = json.loads(json.dumps(game["actions"]))
    # (the vibe table's display metadata -- symbol, category -- is not part of the fixture: names only)
    actions_cfg["change_vibe"]["vibes"] = [v["name"] for v in actions_cfg["change_vibe"]["vibes"]]
    out = {"seed": seed, "max_steps": steps, "cell_names": names, "map_shape": list(codes.shape),
           "map_codes": codes.reshape(-1).tolist(), "actions": actions, "replay": tree,
           # of the embedded `mg_config` (sim.config.model_dump(mode="json"), replay_log_writer.py:118): its key paths
           # outside the map builder, and the scalar values of a few sections a replay viewer reads
           "mg_config_key_paths": sorted(p for p in key_paths(mgc) if not p.startswith("/game/map_builder/")),
           "mg_config_has_map_builder": "map_builder" in game,
           "mg_config_values": {"label": mgc.get("label"), "num_agents": game["num_agents"], "max_steps": game["max_steps"],
                                "resource_names": game["resource_names"], "obs": game["obs"],
                                "actions": actions_cfg}}
    data = zlib.compress(json.dumps(out).encode())
    (GOLD / "replay_arena24.json.z").write_bytes(data)
    print("wrote replay_arena24.json.z", len(data), "bytes;", len(tree["objects"]), "objects,", len(actions), "steps")


if __name__ == "__main__":
    main()
