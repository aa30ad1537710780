"""GameConfig -> packed int32 config blob (+ host-side name tables).

Restates `convert_to_cpp_game_config` (reference: `PY/config/mettagrid_c_config.py:30-577`) but, instead of
building pybind11 objects, writes the flat stream parsed by `csrc/mg_blob.h`.  Every reference container
whose *iteration order* is observable is emitted already ordered, using the libstdc++
emulation in `metta_amd/umap.py`:

* maps that reach C++ through pybind11 (`input_resources`, `consumed_resources`, `initial_inventory`,
  `stat_rewards`, ...) -> `order_from_pydict`;
* maps filled by `operator[]` inside the reference (`Inventory::_limits`, the scaled partial-usage protocol,
  `Move::_handlers`) -> `order_from_inserts`.
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Any

import numpy as np

from metta_amd import stats as stat_names
from metta_amd.config import (
    AgentConfig,
    AssemblerConfig,
    ChestConfig,
    GameConfig,
    MettaGridConfig,
    WallConfig,
)
from metta_amd.id_map import FIXED_FEATURES, encoder_num_inventory_tokens
from metta_amd.umap import order_from_inserts, order_from_pydict, std_hash_string

MG_BLOB_MAGIC = 0x4D474231
MG_BLOB_VERSION = 5

ACT_NOOP, ACT_MOVE, ACT_CHANGE_VIBE = 0, 1, 2
CELL_WALL, CELL_AGENT, CELL_ASSEMBLER, CELL_CHEST = 1, 2, 3, 4
VH_ATTACK, VH_TRANSFER = 1, 2

ORIENTATION = {
    "north": 0, "south": 1, "west": 2, "east": 3,
    "northwest": 4, "northeast": 5, "southwest": 6, "southeast": 7,
}  # fmt: skip
TEAM_NAMES = {0: "red", 1: "blue", 2: "green", 3: "yellow", 4: "purple", 5: "orange"}
EMPTY_CELL_NAMES = ("empty", ".", " ")


def _f32_bits(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


class _Writer:
    def __init__(self) -> None:
        self.w: list[int] = []

    def i(self, *vals: int) -> None:
        for v in vals:
            v = int(v)
            if v >= 1 << 31:
                v -= 1 << 32
            self.w.append(v)

    def f(self, x: float) -> None:
        self.w.append(_f32_bits(x))

    def list(self, xs) -> None:
        xs = list(xs)
        self.i(len(xs), *xs)

    def kv(self, pairs) -> None:
        pairs = list(pairs)
        self.i(len(pairs))
        for k, v in pairs:
            self.i(k, v)


def _ordered_items(d: dict[int, int]) -> list[tuple[int, int]]:
    """(key, value) pairs of a pybind11-converted unordered_map<uint8/int, T> in C++ iteration order."""
    return [(k, d[k]) for k in order_from_pydict(list(d.keys()))]


@dataclass
class CompiledGame:
    blob: np.ndarray
    num_agents: int
    num_tokens: int
    obs_width: int
    obs_height: int
    max_steps: int
    resource_names: list[str]
    action_names: list[str]
    cell_names: list[str]  # cell_names[code - 1]
    cell_kinds: list[int]
    cell_group: list[int]  # group index for agent cells, else -1
    cell_type_names: list[str]  # reference `type_name` of the object a cell code creates ("agent" or the objects key)
    group_names: list[str]
    tag_names: list[str]
    object_type_names: list[str]
    feature_names: dict[int, str]
    agent_slot_names: list[list[str]]  # per group
    game_slot_names: list[str]
    extra: dict[str, Any] = field(default_factory=dict)
    assembler_protocol_listing: dict = field(default_factory=dict)  # assembler object name -> `protocols` of grid_objects
    unclip_protocol_listing: list = field(default_factory=list)  # the clipper's unclipping protocols, same form

    @property
    def cell_code(self) -> dict[str, int]:
        return {n: i + 1 for i, n in enumerate(self.cell_names)}

    def encode_map(self, grid) -> np.ndarray:
        """Cell names -> uint8 cell codes (0 = empty).  Unknown names raise, as BIND/mettagrid_c.cpp:191-193."""
        arr = np.asarray(grid)
        out = np.zeros(arr.shape, dtype=np.uint8)
        code = self.cell_code
        for name in np.unique(arr):
            s = str(name)
            if s in EMPTY_CELL_NAMES:
                continue
            if s not in code:
                raise RuntimeError(f"Unknown object type: {s}")
            out[arr == name] = code[s]
        return out

    def decode_map(self, codes: np.ndarray) -> list[list[str]]:
        names = ["empty"] + self.cell_names
        return [[names[int(v)] for v in row] for row in np.asarray(codes)]


def _inv_cfg(w: _Writer, limit_defs: list[tuple[list[int], int, dict[int, int]]], initial: dict[int, int]) -> None:
    w.i(len(limit_defs))
    insert_seq: list[int] = []
    owner: dict[int, int] = {}
    for li, (res, base, mods) in enumerate(limit_defs):
        w.i(base)
        w.list(res)
        w.kv(_ordered_items(mods))
        for r in res:
            insert_seq.append(r)
            owner[r] = li  # `_limits[resource] = limit` (inventory.cpp:19-21): last writer wins
    w.kv([(r, owner[r]) for r in order_from_inserts(insert_seq)])
    w.kv(_ordered_items(initial))


def compile_game(cfg: GameConfig | MettaGridConfig, init_obs_passes: int = 2) -> CompiledGame:
    """`init_obs_passes`: 2 reproduces the Puffer/Simulation path, where observations (and the `tokens_*` game
    stats) are computed once by the constructor and once more by `set_buffers`; 1 reproduces a
    bare `MettaGrid(cfg, map, seed)`."""
    game = cfg.game if isinstance(cfg, MettaGridConfig) else cfg
    if game.clipper is not None and not game.clipper.unclipping_protocols:
        raise RuntimeError("Clipper config provided but unclipping_protocols is empty")  # mettagrid_c.cpp:145-147
    # collectives (mettagrid_c_config.py:544-575; BIND/mettagrid_c.cpp:95-128): only what the step path can observe --
    # the statistics `Agent::compute_stat_rewards` reads through `getCollective()` (agent.cpp:115-117)
    coll_id = {c.name: i for i, c in enumerate(game.collectives)}

    resource_names = list(game.resource_names)
    rid = {n: i for i, n in enumerate(resource_names)}
    R = len(resource_names)
    vibes = game.actions.change_vibe.vibes
    vibe_names = [v.name for v in vibes]
    vid = {n: i for i, n in enumerate(vibe_names)}

    type_names_sorted = sorted(game.objects.keys())
    type_id = {n: i + 1 for i, n in enumerate(type_names_sorted)}

    agents = list(game.agents)
    if not agents:
        base = game.agent.model_dump()
        base["team_id"] = 0
        agents = [AgentConfig(**base) for _ in range(game.num_agents)]
    default_limit = game.agent.inventory.default_limit

    all_tags: set[str] = set()
    for o in game.objects.values():
        all_tags.update(o.tags)
    for a in agents:
        all_tags.update(a.tags)
    tag_names = sorted(all_tags)
    if len(tag_names) > 256:
        raise ValueError(f"Too many unique tags ({len(tag_names)}). Maximum supported is 256 due to uint8 limit.")
    tag_id = {t: i for i, t in enumerate(tag_names)}

    team_groups: dict[int, list[AgentConfig]] = {}
    for a in agents:
        team_groups.setdefault(a.team_id, []).append(a)
    for team, members in team_groups.items():
        first = set(members[0].tags)
        for m in members[1:]:
            if set(m.tags) != first:
                raise ValueError(f"All agents in team {team} must have identical tags.")
    group_names = [TEAM_NAMES.get(t, f"team_{t}") for t in team_groups]
    G = len(group_names)

    # ---- cell table (the keys of objects_cpp_params, mettagrid_c_config.py:234-242,255,307,357) ----------
    cell_names: list[str] = []
    cell_kinds: list[int] = []
    cell_idx: list[int] = []
    cell_group: list[int] = []
    cell_type_names: list[str] = []

    def add_cell(name: str, kind: int, idx: int, group: int = -1, type_name: str = "agent") -> None:
        if name in cell_names:  # later registration overwrites (dict semantics)
            k = cell_names.index(name)
            cell_kinds[k], cell_idx[k], cell_group[k], cell_type_names[k] = kind, idx, group, type_name
            return
        cell_names.append(name)
        cell_kinds.append(kind)
        cell_idx.append(idx)
        cell_group.append(group)
        cell_type_names.append(type_name)

    for gi, team in enumerate(team_groups):
        add_cell("agent." + group_names[gi], CELL_AGENT, gi, gi)
        add_cell(f"agent.team_{team}", CELL_AGENT, gi, gi)
        if team == 0:
            add_cell("agent.default", CELL_AGENT, gi, gi)
            add_cell("agent.agent", CELL_AGENT, gi, gi)
    walls: list[WallConfig] = []
    assemblers: list[AssemblerConfig] = []
    chests: list[ChestConfig] = []
    obj_keys: list[str] = []
    for key, o in game.objects.items():
        obj_keys.append(key)
        if isinstance(o, WallConfig):
            add_cell(o.map_name or key, CELL_WALL, len(walls), type_name=key)
            walls.append(o)
        elif isinstance(o, AssemblerConfig):
            add_cell(o.map_name or key, CELL_ASSEMBLER, len(assemblers), type_name=key)
            assemblers.append(o)
        elif isinstance(o, ChestConfig):
            add_cell(o.map_name or key, CELL_CHEST, len(chests), type_name=key)
            chests.append(o)
        else:
            raise ValueError(f"Unknown object type: {key}")
    key_of = {id(o): k for k, o in game.objects.items()}
    if len(cell_names) > 255:
        raise ValueError("too many distinct map cell names (uint8 cell codes)")

    actions = game.actions
    action_names = [a.name for a in actions.actions()]
    n_actions = len(action_names)

    feats = game.id_map().features()
    fid = {f.name: f.id for f in feats}
    if max(fid.values()) > 254:
        raise ValueError("too many observation features for uint8 ids")

    agent_slot_names = [stat_names.agent_slot_names(resource_names, group_names, gn, n_actions) for gn in group_names]
    game_slot_names = stat_names.game_slot_names(resource_names, cell_names)

    w = _Writer()
    w.i(MG_BLOB_MAGIC, MG_BLOB_VERSION)
    go = game.global_obs
    w.i(
        R, game.num_agents, game.obs.width, game.obs.height, game.obs.num_tokens, game.obs.token_value_base,
        game.max_steps, int(game.episode_truncates),
        int(go.episode_completion_pct), int(go.last_action), int(go.last_reward), int(go.compass), int(go.goal_obs),
        int(game.protocol_details_obs), game.inventory_regen_interval, init_obs_passes, len(vibe_names), G,
    )  # fmt: skip
    w.i(*[fid[n] for n, _ in FIXED_FEATURES])
    n_inv = encoder_num_inventory_tokens(65535, game.obs.token_value_base)
    w.i(n_inv)
    inv_fids: list[int] = []
    for r in resource_names:
        inv_fids.append(fid[f"inv:{r}"])
        for p in range(1, n_inv):
            key = f"inv:{r}:p{p}"
            if key not in fid:
                raise RuntimeError(f"Inventory power feature '{key}' not found in feature_ids")
            inv_fids.append(fid[key])
    w.list(inv_fids)
    if game.protocol_details_obs:
        w.list([fid[f"protocol_input:{r}"] for r in resource_names])
        w.list([fid[f"protocol_output:{r}"] for r in resource_names])
    else:
        w.list([])
        w.list([])

    # ---- actions (BIND/mettagrid_c.cpp:291-352) --------------------------------------------------------
    # The C++ core always creates noop and every allowed move direction, whatever `enabled` says
    # (noop.hpp:13-17, move.hpp:37-47); `enabled=False` only empties their resource maps.
    table: list[tuple[int, int]] = [(ACT_NOOP, 0)]
    for d in actions.move.allowed_directions:
        table.append((ACT_MOVE, ORIENTATION[d]))
    n_vibe_actions = len(vibes) if actions.change_vibe.enabled else 0
    for i in range(n_vibe_actions):
        table.append((ACT_CHANGE_VIBE, i))
    w.i(len(table))
    for k, a in table:
        w.i(k, a)

    def handler_res(ac) -> tuple[dict[int, int], dict[int, int]]:
        if not ac.enabled:
            return {}, {}
        missing = [r for r in ac.consumed_resources if r not in rid]
        if missing:
            raise ValueError(
                f"Action '{ac.action_handler}' has consumed_resources {missing} that are not in resource_names."
            )
        consumed = {rid[k]: int(v) for k, v in ac.consumed_resources.items()}
        src = ac.required_resources or ac.consumed_resources
        required = {rid[k]: int(v) for k, v in src.items()}
        for item, amt in consumed.items():
            if item not in required:
                raise RuntimeError(f"Consumed resource item {item} not found in required resources")
            if required[item] < amt:
                raise RuntimeError("Required resources must be >= consumed resources.")
        return required, consumed

    for ac in (actions.noop, actions.move, actions.change_vibe):
        req, con = handler_res(ac)
        w.kv(_ordered_items(req))
        w.kv(_ordered_items(con))
    w.i(1)  # max priority: Attack::priority == 1 always (attack.hpp:79), so the priority loop runs twice

    # ---- vibe-triggered handlers (move.hpp:49-71) -----------------------------------------------------
    atk = actions.attack
    for v in atk.vibes:
        if v not in vid:
            raise ValueError(f"Unknown vibe name '{v}' in attack.vibes")
    atk_vibes = [vid[v] for v in atk.vibes]
    xfer = actions.transfer
    xfer_map: dict[int, Any] = {}
    for vt in xfer.vibe_transfers:
        if vt.vibe not in vid:
            raise ValueError(f"Unknown vibe name '{vt.vibe}' in transfer.vibe_transfers")
        if vid[vt.vibe] in xfer_map:
            raise ValueError(f"Duplicate vibe name '{vt.vibe}' in transfer.vibe_transfers")
        xfer_map[vid[vt.vibe]] = vt
    vibe_handler = [0] * 256
    for name in order_from_inserts(["attack", "transfer"], std_hash_string):
        if name == "attack":
            for v in atk_vibes:
                vibe_handler[v] = VH_ATTACK
        else:
            for v in xfer_map:
                vibe_handler[v] = VH_TRANSFER
    w.list(vibe_handler)

    _, atk_consumed = handler_res(atk)
    w.kv(_ordered_items(atk_consumed))
    w.kv(_ordered_items({rid[k]: v for k, v in atk.defense_resources.items()}))
    w.kv(_ordered_items({rid[k]: v for k, v in atk.armor_resources.items()}))
    w.kv(_ordered_items({rid[k]: v for k, v in atk.weapon_resources.items()}))
    w.kv(_ordered_items({rid[k]: v for k, v in atk.success.actor_inv_delta.items()}))
    w.kv(_ordered_items({rid[k]: v for k, v in atk.success.target_inv_delta.items()}))
    w.list([rid[n] for n in atk.success.loot])
    w.i(atk.success.freeze)
    bonus = [0] * 256
    for v, b in atk.vibe_bonus.items():
        if v not in vid:
            raise ValueError(f"Unknown vibe name '{v}' in attack.vibe_bonus")
        bonus[vid[v]] = int(b)
    w.list(bonus)
    vibe_res = [-1] * 256
    for i, vn in enumerate(vibe_names[:256]):
        if i > 0 and vn in rid:
            vibe_res[i] = rid[vn]
    w.list(vibe_res)

    w.kv(_ordered_items({rid[k]: int(v) for k, v in xfer.required_resources.items()}))
    w.i(len(xfer_map))
    for v, vt in xfer_map.items():
        w.i(v)
        w.kv(_ordered_items({rid[k]: d for k, d in vt.actor.items()}))
        w.kv(_ordered_items({rid[k]: d for k, d in vt.target.items()}))

    # ---- agent groups (mettagrid_c_config.py:98-242) ---------------------------------------------------
    w.i(G)
    for gi, (team, members) in enumerate(team_groups.items()):
        first = members[0]
        inv = first.inventory
        w.i(team, gi, first.freeze_duration, first.initial_vibe)
        w.list([tag_id[t] for t in first.tags])
        limit_defs: list[tuple[list[int], int, dict[int, int]]] = []
        configured: set[str] = set()
        for lim in inv.limits.values():
            limit_defs.append(
                ([rid[n] for n in lim.resources], lim.limit, {rid[n]: b for n, b in lim.modifiers.items() if n in rid})
            )
            configured.update(lim.resources)
        for n in resource_names:
            if n not in configured:
                limit_defs.append(([rid[n]], default_limit, {}))
        _inv_cfg(w, limit_defs, {rid[k]: v for k, v in inv.initial.items()})

        stat_rewards: dict[str, float] = dict(first.rewards.stats)
        stat_max: dict[str, float] = dict(first.rewards.stats_max)
        for k, v in first.rewards.inventory.items():
            assert k in rid, f"Inventory reward {k} not in resource_names"
            assert k + ".amount" not in stat_rewards, f"Stat reward {k}.amount already exists"
            stat_rewards[k + ".amount"] = v
        for k, v in first.rewards.inventory_max.items():
            assert k in rid, f"Inventory reward max {k} not in resource_names"
            assert k + ".amount" not in stat_max, f"Stat reward max {k}.amount already exists"
            stat_max[k + ".amount"] = v
        order = order_from_pydict(list(stat_rewards.keys()), std_hash_string)
        aslot = {n: i for i, n in enumerate(agent_slot_names[gi]) if n}
        gslot = {n: i for i, n in enumerate(game_slot_names)}
        w.i(len(order))
        for name in order:
            w.i(aslot.get(name, -1), gslot.get(name, -1))
            w.f(stat_rewards[name])
            w.i(1 if name in stat_max else 0)
            w.f(stat_max.get(name, 0.0))
            # what a collective's StatsTracker can hold under this name (collective.hpp:46-52,101-108)
            if name == "aligned.agent":
                w.i(1, 0)
            elif name.startswith("collective.") and name.endswith(".deposited") and name[11:-10] in rid:
                w.i(2, rid[name[11:-10]])
            else:
                w.i(0, 0)
        prefix = "collective:"  # membership by tag, in tag order; the last existing one stays (alignable.cpp:5-16)
        w.list([coll_id[t[len(prefix):]] for t in first.tags if t.startswith(prefix) and t[len(prefix):] in coll_id])
        goal: list[int] = []
        seen: set[str] = set()
        for name in order:  # BIND/mettagrid_c.cpp:371-392
            dot = name.find(".")
            if dot < 0:
                continue
            res = name[:dot]
            if res not in seen and res in rid:
                goal.append(fid[f"inv:{res}"])
                seen.add(res)
        w.list(goal)
        regen = inv.regen_amounts
        w.i(len(regen))
        for vname, amounts in regen.items():
            w.i(vid[vname])
            w.kv(_ordered_items({rid[k]: a for k, a in amounts.items()}))
        w.list([rid[n] for n in first.diversity_tracked_resources if n in rid])
        dmg = first.damage
        w.kv(_ordered_items({rid[k]: v for k, v in dmg.threshold.items()}) if dmg else [])
        w.kv(_ordered_items({rid[k]: v for k, v in dmg.resources.items()}) if dmg else [])

    # ---- cells, walls, assemblers, chests --------------------------------------------------------------
    w.i(len(cell_names) + 1)
    w.i(0, 0)
    for k, i in zip(cell_kinds, cell_idx):
        w.i(k, i)
    w.i(len(walls))
    for o in walls:
        w.i(type_id[key_of[id(o)]], o.vibe)
        w.list([tag_id[t] for t in o.tags])
    assembler_protocol_listing: dict[str, list] = {}
    w.i(len(assemblers))
    for o in assemblers:
        w.i(type_id[key_of[id(o)]], o.vibe, int(o.allow_partial_usage), o.max_uses, int(o.clip_immune),
            int(o.start_clipped), o.chest_search_distance)  # fmt: skip
        w.list([tag_id[t] for t in o.tags])
        protos = []
        seen_pm: list[tuple[list[int], int]] = []
        for pc in reversed(o.protocols):
            for v in pc.vibes:
                if v not in vid:
                    raise ValueError(f"Unknown vibe name '{v}' in assembler '{key_of[id(o)]}' protocol")
            vids = sorted(vid[v] for v in pc.vibes)
            if (vids, pc.min_agents) in seen_pm:
                raise ValueError(
                    f"Protocol with vibes {pc.vibes} and min_agents {pc.min_agents} already exists in {key_of[id(o)]}"
                )
            seen_pm.append((vids, pc.min_agents))
            ins = _ordered_items({rid[k]: int(v) for k, v in pc.input_resources.items()})
            outs = _ordered_items({rid[k]: int(v) for k, v in pc.output_resources.items()})
            protos.append((pc.min_agents, pc.cooldown, vids, ins, outs))
        w.i(len(protos))
        keymap: dict[int, list[int]] = {}
        for pi, (min_agents, cooldown, vids, ins, outs) in enumerate(protos):
            w.i(min_agents, cooldown)
            w.list(vids)
            w.kv(ins)
            w.kv(outs)
            w.list(order_from_inserts([k for k, _ in ins]))
            w.list(order_from_inserts([k for k, _ in outs]))
            key = 0
            for v in vids:
                key = ((key << 8) | v) & ((1 << 64) - 1)
            keymap.setdefault(key, []).append(pi)
        w.i(len(keymap))
        for key, plist in keymap.items():
            plist = sorted(plist, key=lambda pi: -protos[pi][0])
            w.i(key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)
            w.list(plist)
        # the listing `MettaGrid::grid_objects` gives (mettagrid_c.cpp:864-893): the protocol map's iteration order
        # (assembler.hpp:334-349: operator[] inserts in list order, identity hash on the 64-bit key), each key's
        # protocols by descending min_agents
        listing = []
        for key in order_from_inserts(list(keymap.keys())):
            for pi in sorted(keymap[key], key=lambda pi: -protos[pi][0]):
                min_agents, cooldown, vids, ins, outs = protos[pi]
                listing.append({"inputs": dict(ins), "outputs": dict(outs), "cooldown": cooldown,
                                "min_agents": min_agents, "vibes": list(vids)})  # fmt: skip
        assembler_protocol_listing[key_of[id(o)]] = listing
    w.i(len(chests))
    for o in chests:
        w.i(type_id[key_of[id(o)]], o.vibe)
        w.list([tag_id[t] for t in o.tags])
        limit_defs = []
        for lim in o.inventory.limits.values():
            res = [rid[n] for n in lim.resources if n in rid]
            if res:
                limit_defs.append((res, lim.limit, {rid[n]: b for n, b in lim.modifiers.items() if n in rid}))
        _inv_cfg(w, limit_defs, {rid[k]: v for k, v in o.inventory.initial.items()})
        vt_map: dict[int, dict[int, int]] = {}
        for vname, deltas in o.vibe_transfers.items():
            if vname not in vid:
                raise ValueError(f"Unknown vibe name '{vname}' in chest '{key_of[id(o)]}' vibe_transfers")
            vt_map[vid[vname]] = {rid[k]: d for k, d in deltas.items()}
        w.i(len(vt_map))
        for v, deltas in vt_map.items():
            w.i(v)
            w.kv(_ordered_items(deltas))

    w.i(len(game.collectives))
    for cc in game.collectives:
        w.kv(_ordered_items({rid[k]: int(v) for k, v in cc.inventory.initial.items() if k in rid}))
    # clipper (mettagrid_c_config.py:506-539; systems/clipper.hpp): the unclipping protocols in list order (NOT
    # reversed, unlike an assembler's), each with the key `build_protocol_map` files it under (assembler.hpp:326-349)
    clip = game.clipper
    unclip_protocol_listing: list = []
    w.i(0 if clip is None else 1)
    if clip is not None:
        w.i(clip.length_scale, clip.scaled_cutoff_distance, clip.clip_period, len(clip.unclipping_protocols))
        for pc in clip.unclipping_protocols:
            for v in pc.vibes:
                if v not in vid:
                    raise ValueError(f"Unknown vibe name '{v}' in clipper unclipping_protocols")
            vids = sorted(vid[v] for v in pc.vibes)
            ins = _ordered_items({rid[k]: int(v) for k, v in pc.input_resources.items()})
            outs = _ordered_items({rid[k]: int(v) for k, v in pc.output_resources.items()})
            w.i(pc.min_agents, pc.cooldown)
            w.list(vids)
            w.kv(ins)
            w.kv(outs)
            w.list(order_from_inserts([k for k, _ in ins]))
            w.list(order_from_inserts([k for k, _ in outs]))
            key = 0
            for v in vids:
                key = ((key << 8) | v) & ((1 << 64) - 1)
            w.i(key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)
            unclip_protocol_listing.append({"inputs": dict(ins), "outputs": dict(outs), "cooldown": pc.cooldown,
                                            "min_agents": pc.min_agents, "vibes": list(vids)})  # fmt: skip
        # (An assembler that starts clipped draws its unclipping protocol from the environment's seed when the Clipper
        # is built, clipper.hpp:107-110,186-193.  When the protocols differ in shape the `tokens_*` statistics of the
        # initial observations differ per environment: the device recounts them at every restart instead of taking
        # them from the per-map template, mg_recount_initial_tokens.)
    w.i(len(agent_slot_names[0]) if agent_slot_names else 0, len(game_slot_names))
    w.i(MG_BLOB_MAGIC)

    # mettagrid_c.cpp:163-177: sized by the number of cell entries (agent cells included), grown to the largest type id
    object_type_names = [""] * max(len(type_names_sorted) + 1, len(cell_names))
    object_type_names[0] = "agent"
    for n, i in type_id.items():
        object_type_names[i] = n

    return CompiledGame(
        blob=np.asarray(w.w, dtype=np.int32),
        num_agents=game.num_agents,
        num_tokens=game.obs.num_tokens,
        obs_width=game.obs.width,
        obs_height=game.obs.height,
        max_steps=game.max_steps,
        resource_names=resource_names,
        action_names=action_names,
        cell_names=cell_names,
        cell_kinds=cell_kinds,
        cell_group=cell_group,
        cell_type_names=cell_type_names,
        group_names=group_names,
        tag_names=tag_names,
        object_type_names=object_type_names,
        assembler_protocol_listing=assembler_protocol_listing,
        unclip_protocol_listing=unclip_protocol_listing,
        feature_names={f.id: f.name for f in feats},
        agent_slot_names=agent_slot_names,
        game_slot_names=game_slot_names,
        extra={"n_action_table": len(table), "vibe_names": vibe_names},
    )
