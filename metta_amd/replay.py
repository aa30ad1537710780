"""Replay logs of device environments in the reference's format (version 3), readable by MettaScope.

Mirrors `python/src/mettagrid/simulator/replay_log_writer.py` (`EpisodeReplay`, `InMemoryReplayWriter`,
`ReplayLogWriter`) and `util/grid_object_formatter.py`: one record per grid object, every field either a constant or
a `[[step, value], ...]` change list, zlib-compressed JSON.  The source of the per-step object state is
`MettaGridBatch.grid_objects(env)` -- the device state read back through `mg_agent_state` / `mg_assembler_state` /
`mg_chest_state` -- so recording an episode costs one host synchronisation per recorded step: it is an evaluation
tool, as in the reference, not part of the training loop.
"""

from __future__ import annotations

import gzip
import json
import logging
import uuid
import zlib
from pathlib import Path
from typing import Any, Optional

import numpy as np

STATIC_OBJECT_TYPES = frozenset({"wall"})  # recorded once (replay_log_writer.py:97)


def _pairs(d: dict) -> list:
    return [[int(k), int(v)] for k, v in d.items()]


def format_grid_object(obj: dict, actions, action_success, rewards, total_rewards) -> dict:
    """One object's fields for one step (grid_object_formatter.py:8-151)."""
    out: dict[str, Any] = {
        "id": obj["id"],
        "type_name": obj["type_name"],
        "location": list(obj["location"]),
        "orientation": obj.get("orientation", 0),
        "inventory": _pairs(obj.get("inventory", {})),
        "inventory_max": obj.get("inventory_max", 0),
        "color": obj.get("color", 0),
    }
    if "collective_id" in obj:
        out["collective_id"] = obj["collective_id"]
    if "agent_id" in obj:
        i = obj["agent_id"]
        out.update(
            is_agent=True,
            agent_id=i,
            vision_size=11,
            action_id=int(actions[i]),
            action_param=0,
            action_success=bool(action_success[i]),
            current_reward=float(rewards[i]),
            total_reward=float(total_rewards[i]),
            freeze_remaining=obj.get("freeze_remaining", 0),
            is_frozen=obj.get("is_frozen", False),
            freeze_duration=obj.get("freeze_duration", 0),
            group_id=obj["group_id"],
            vibe_id=obj.get("vibe", 0),
            vibe=obj.get("vibe", 0),
        )
    elif "protocols" in obj:
        out.update(
            cooldown_remaining=obj.get("cooldown_remaining", 0),
            cooldown_duration=obj.get("cooldown_duration", 0),
            is_clipped=obj.get("is_clipped", False),
            is_clip_immune=obj.get("is_clip_immune", False),
            uses_count=obj.get("uses_count", 0),
            max_uses=obj.get("max_uses", 0),
            allow_partial_usage=obj.get("allow_partial_usage", False),
            protocols=[
                {
                    "minAgents": p.get("min_agents", 0),
                    "vibes": list(p.get("vibes", [])),
                    "inputs": _pairs(p.get("inputs", {})),
                    "outputs": _pairs(p.get("outputs", {})),
                    "cooldown": p.get("cooldown", 0),
                }
                for p in obj["protocols"]
            ],
        )
    return out


class EpisodeReplay:
    """The replay tree of one episode of one environment (replay_log_writer.py:93-226)."""

    def __init__(self, batch, env: int, mg_config: Optional[dict] = None):
        self.batch, self.env = batch, int(env)
        c = batch.compiled
        self.step = 0
        self.objects: list[dict] = []
        self.total_rewards = np.zeros(batch.num_agents)
        self._index: dict[int, int] = {}
        self._logged = False  # the first logged step records the static objects too, wherever in the episode it falls
        self._compress, self.content_type = zlib.compress, "application/x-compress"
        h, w = batch.env_map_dims(batch.env_info(self.env)["map_id"])
        self.replay_data = {
            "version": 3,
            "action_names": list(c.action_names),
            "item_names": list(c.resource_names),
            "type_names": list(c.object_type_names),
            "map_size": [w, h],
            "num_agents": batch.num_agents,
            "max_steps": int(c.max_steps),
            "mg_config": mg_config if mg_config is not None else _dump(batch.cfg),
            "objects": self.objects,
        }

    def set_compression(self, compression: str) -> None:
        if compression == "zlib":
            self._compress, self.content_type = zlib.compress, "application/x-compress"
        elif compression == "gzip":
            self._compress, self.content_type = gzip.compress, "application/gzip"
        else:
            raise ValueError(f"unknown compression {compression!r}, try 'zlib' or 'gzip'")

    def log_step(self, current_step: int, actions, rewards, action_success) -> None:
        """After step `current_step` of the episode: merge every object's fields into its change lists."""
        self.total_rewards += np.asarray(rewards, dtype=np.float64)
        objs = self.batch.grid_objects(self.env, ignore_types=() if not self._logged else tuple(STATIC_OBJECT_TYPES))
        self._logged = True
        for oid, obj in objs.items():
            idx = self._index.get(oid)
            if idx is None:
                idx = self._index[oid] = len(self.objects)
                self.objects.append({})
            _merge(self.objects[idx], self.step, format_grid_object(obj, actions, action_success, rewards, self.total_rewards))
        self.step += 1
        if current_step != self.step:
            raise ValueError(f"Writing multiple steps at once: step {current_step} != Replay step {self.step}.")

    def get_replay_data(self) -> dict:
        self.replay_data["max_steps"] = self.step
        for obj in self.objects:  # a field that never changed is stored as its value
            for key, changes in list(obj.items()):
                if isinstance(changes, list) and len(changes) == 1 and isinstance(changes[0], (list, tuple)) and len(changes[0]) == 2:
                    obj[key] = changes[0][1]
        return self.replay_data

    def write_replay(self, path) -> None:
        Path(path).write_bytes(self._compress(json.dumps(self.get_replay_data()).encode("utf-8")))


def _merge(record: dict, step: int, update: dict) -> None:
    """replay_log_writer.py:166-182."""
    for key, value in update.items():
        if key not in record:
            record[key] = [[step, value]] if step == 0 else [[0, 0], [step, value]]
        elif record[key][-1][1] != value:
            record[key].append([step, value])
    for key in record:
        if key not in update and record[key][-1][1] != 0:
            record[key].append([step, 0])


def _dump(cfg) -> dict:
    """`sim.config.model_dump(mode="json")` in full, as the reference stores it (replay_log_writer.py:118).  A map builder
    that does not serialise (a builder object holding arrays, a device pool) is left out with a warning; anything else that
    fails to serialise is an error, not an empty tree."""
    try:
        return cfg.model_dump(mode="json")
    except Exception as ex:
        logging.getLogger("metta_amd.replay").warning("replay mg_config: game.map_builder left out (%s: %s)", type(ex).__name__, ex)
        return cfg.model_dump(mode="json", exclude={"game": {"map_builder"}})


class DeviceReplayWriter:
    """Records episodes of ONE environment of a batch; the counterpart of `InMemoryReplayWriter` / `ReplayLogWriter`
    (replay_log_writer.py:21-91).  Call `on_step()` after every `batch.step()`; episodes are delimited by the
    environment's own step counter (auto-reset included).  With `replay_dir` every finished episode is written to
    `<replay_dir>/<uuid>.json.z`."""

    def __init__(self, batch, env: int = 0, replay_dir: Optional[str] = None):
        self.batch, self.env, self.replay_dir = batch, int(env), replay_dir
        self._replay: Optional[EpisodeReplay] = None
        self._completed: list[EpisodeReplay] = []
        self._paths: list[str] = []

    def on_step(self, actions=None) -> None:
        """`actions`: what was passed to `batch.step` for this environment's agents, when it was a device tensor read
        in place (the batch's own action buffer is then not written); default: the action buffer."""
        b, e, A = self.batch, self.env, self.batch.num_agents
        step = b.current_step(e)
        if self._replay is None or step == 1:  # the environment restarted on this step (or recording begins)
            if self._replay is not None and self._replay.step > 0:
                self._finish()
            self._replay = EpisodeReplay(b, e)
            self._replay.step = step - 1  # recording may begin mid-episode
        rows = slice(e * A, (e + 1) * A)
        acts = b.actions[rows].cpu().numpy() if actions is None else np.asarray(actions).reshape(-1)
        self._replay.log_step(step, acts, b.rewards[rows].cpu().numpy(), b.action_success[rows].cpu().numpy())
        if bool(b.terminals[rows].all().item()) or bool(b.truncations[rows].all().item()):
            self._finish()

    def _finish(self) -> None:
        rep, self._replay = self._replay, None
        self._completed.append(rep)
        if self.replay_dir is not None:
            path = str(Path(self.replay_dir) / f"{uuid.uuid4()}.json.z")
            rep.write_replay(path)
            self._paths.append(path)

    def get_completed_replays(self) -> list[EpisodeReplay]:
        return self._completed

    def get_written_replay_paths(self) -> list[str]:
        return list(self._paths)
