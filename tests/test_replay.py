"""Replay writer: the replay tree of a device environment equals the one the reference's
`InMemoryReplayWriter` built for the same map, seed and actions (tests/golden/replay_arena24.json.z, written by
oracle/gen_golden_replay.py with the reference's Simulation; its `mg_config` dump is not part of the fixture)."""

import json
import zlib

import numpy as np
import pytest

from metta_amd.replay import _merge, format_grid_object
from tests.helpers import GOLD


def test_change_lists_follow_the_reference_rules():
    rec: dict = {}
    _merge(rec, 0, {"a": 1, "b": [1, 2]})
    _merge(rec, 1, {"a": 1, "b": [1, 3], "c": 5})  # unchanged / changed / appears late
    _merge(rec, 2, {"a": 2, "b": [1, 3]})  # c vanishes -> zero entry
    _merge(rec, 3, {"a": 2, "b": [1, 3]})
    assert rec == {"a": [[0, 1], [2, 2]], "b": [[0, [1, 2]], [1, [1, 3]]], "c": [[0, 0], [1, 5], [2, 0]]}
    wall = format_grid_object({"id": 1, "type_name": "wall", "location": (3, 4)}, [], [], [], [])
    assert wall == {"id": 1, "type_name": "wall", "location": [3, 4], "orientation": 0, "inventory": [],
                    "inventory_max": 0, "color": 0}  # fmt: skip


@pytest.mark.gpu
def test_replay_equals_the_reference_writers(tmp_path):
    from metta_amd.batch import MettaGridBatch
    from metta_amd.builder import make_arena
    from metta_amd.replay import DeviceReplayWriter

    fx = json.loads(zlib.decompress((GOLD / "replay_arena24.json.z").read_bytes()))
    cfg = make_arena(24)
    cfg.game.max_steps = fx["max_steps"]
    grid = np.array(fx["cell_names"])[np.array(fx["map_codes"]).reshape(fx["map_shape"])]
    b = MettaGridBatch(cfg, 2, [grid], device=0, init_obs_passes=1, seeds=[fx["seed"], 5], map_ids=[0, 0])
    b.set_auto_reset(False)
    w = DeviceReplayWriter(b, env=0, replay_dir=str(tmp_path))
    rs = np.random.RandomState(0)
    for acts in fx["actions"]:
        both = np.concatenate([np.asarray(acts, np.int32), rs.randint(0, 5, 24).astype(np.int32)])
        b.step(both)
        w.on_step()
    (rep,) = w.get_completed_replays()
    got = json.loads(json.dumps({k: v for k, v in rep.get_replay_data().items() if k != "mg_config"}))
    want = fx["replay"]
    assert set(got) == set(want)
    for k in want:
        if k != "objects":
            assert got[k] == want[k], k
    assert len(got["objects"]) == len(want["objects"])
    for g, x in zip(got["objects"], want["objects"]):
        assert g == x, f"object {x.get('id')}: " + str({k: (g.get(k), x.get(k)) for k in set(g) | set(x) if g.get(k) != x.get(k)})
    # the file form: zlib-compressed JSON, as the reference writes `<uuid>.json.z`
    (path,) = w.get_written_replay_paths()
    disk = json.loads(zlib.decompress(open(path, "rb").read()))
    assert disk["version"] == 3 and len(disk["objects"]) == len(want["objects"]) and "mg_config" in disk
    b.close()


def test_mg_config_is_the_full_dump_of_the_reference():
    """`mg_config` of a replay = `sim.config.model_dump(mode="json")` in full (replay_log_writer.py:118), map builder
    included.  Against the reference's own dump (fixture: its key paths outside the map builder and the values of the
    sections a viewer reads): same tree shape, same values; the vibe table's display metadata (symbol, category) is not
    reproduced by this package -- names and order are."""
    from metta_amd.builder import make_arena
    from metta_amd.replay import _dump

    fx = json.loads(zlib.decompress((GOLD / "replay_arena24.json.z").read_bytes()))
    cfg = make_arena(24)
    cfg.game.max_steps = fx["max_steps"]
    d = _dump(cfg)

    def key_paths(x, prefix=""):
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

    assert ("map_builder" in d["game"]) == fx["mg_config_has_map_builder"]
    assert sorted(p for p in key_paths(d) if not p.startswith("/game/map_builder/")) == fx["mg_config_key_paths"]
    want = fx["mg_config_values"]
    assert d["label"] == want["label"]
    for k in ("num_agents", "max_steps", "resource_names", "obs"):
        assert d["game"][k] == want[k], k
    acts = json.loads(json.dumps(d["game"]["actions"]))
    acts["change_vibe"]["vibes"] = [v["name"] for v in acts["change_vibe"]["vibes"]]
    assert acts == want["actions"]


@pytest.mark.gpu
def test_recording_that_begins_mid_episode_still_holds_the_walls():
    """the static objects are logged with the first step a replay SEES, not only with step 0 of the episode."""
    from metta_amd.batch import MettaGridBatch
    from metta_amd.builder import make_arena
    from metta_amd.replay import DeviceReplayWriter
    from tests.helpers import load_map_pool

    cfg = make_arena(24)
    cfg.game.max_steps = 30
    b = MettaGridBatch(cfg, 1, load_map_pool()[:1], device=0, seeds=[3], map_ids=[0])
    b.set_auto_reset(False)
    rs = np.random.RandomState(1)
    for _ in range(7):
        b.step(rs.randint(0, 5, 24).astype(np.int32))
    w = DeviceReplayWriter(b, env=0)
    for _ in range(23):
        b.step(rs.randint(0, 5, 24).astype(np.int32))
        w.on_step()
    (rep,) = w.get_completed_replays()
    types = {o["type_name"] if not isinstance(o["type_name"], list) else o["type_name"][-1][1] for o in rep.get_replay_data()["objects"]}
    assert "wall" in types and "agent" in types
    b.close()
