"""Bridge to the reference's own compiled core (`oracle/_ref/mettagrid_c*.so`).   *** TEST INFRASTRUCTURE ***

`oracle/Makefile ref` compiles the reference's C++ sources (where they lie under /root/reference) into
`oracle/_ref/`; that file travels to the GPU box with the tree.  The reference's *Python* package does not
travel, so this module rebuilds the pybind11 config objects from a `metta_amd.config.GameConfig` directly --
the same field-by-field mapping as the reference's `convert_to_cpp_game_config`
(`packages/mettagrid/python/src/mettagrid/config/mettagrid_c_config.py:30-577`).  `oracle/gen_golden_features.py`
cross-checks this mapping against the real converter, step by step, on twelve feature-dense configurations
(attack, transfer, chests, damage, regeneration, partial usage, ...) when it writes tests/golden/feature_*.npz.

Used by tests (HIP / oracle vs the real reference) and by bench.py's `cpu_baseline` leg (kind "reference").
"""

from __future__ import annotations

import importlib.util
import sys
from pathlib import Path
from typing import Any

import numpy as np

_REF_DIR = Path(__file__).resolve().parent / "_ref"
_mod = None


def load_reference_module():
    """Import oracle/_ref/mettagrid_c*.so (raises FileNotFoundError if it was never built)."""
    global _mod
    if _mod is not None:
        return _mod
    cands = sorted(_REF_DIR.glob("mettagrid_c*.so"))
    if not cands:
        raise FileNotFoundError(f"no reference build under {_REF_DIR}; run `make -C oracle ref` where /root/reference exists")
    spec = importlib.util.spec_from_file_location("mettagrid_c", cands[0])
    mod = importlib.util.module_from_spec(spec)
    sys.modules.setdefault("mettagrid_c", mod)
    spec.loader.exec_module(mod)
    _mod = mod
    return mod


def have_reference() -> bool:
    try:
        load_reference_module()
        return True
    except Exception:
        return False


def to_ref_game_config(game) -> Any:
    """metta_amd.config.GameConfig -> mettagrid_c.GameConfig (pybind11 object of the reference)."""
    from metta_amd.config import AgentConfig, AssemblerConfig, ChestConfig, WallConfig

    C = load_reference_module()
    resource_names = list(game.resource_names)
    rid = {n: i for i, n in enumerate(resource_names)}
    vibe_names = [v.name for v in game.actions.change_vibe.vibes]
    vid = {n: i for i, n in enumerate(vibe_names)}
    type_id = {n: i + 1 for i, n in enumerate(sorted(game.objects.keys()))}

    agents = list(game.agents)
    if not agents:
        base = game.agent.model_dump()
        base["team_id"] = 0
        agents = [AgentConfig(**base) for _ in range(game.num_agents)]
    default_limit = game.agent.inventory.default_limit

    tags: set[str] = set()
    for o in game.objects.values():
        tags.update(o.tags)
    for a in agents:
        tags.update(a.tags)
    tag_id = {t: i for i, t in enumerate(sorted(tags))}

    teams: dict[int, list] = {}
    for a in agents:
        teams.setdefault(a.team_id, []).append(a)
    team_names = {0: "red", 1: "blue", 2: "green", 3: "yellow", 4: "purple", 5: "orange"}

    objects: dict[str, Any] = {}
    for team, members in teams.items():
        first = members[0]
        inv = first.inventory
        stat_rewards = dict(first.rewards.stats)
        stat_max = dict(first.rewards.stats_max)
        for k, v in first.rewards.inventory.items():
            stat_rewards[k + ".amount"] = v
        for k, v in first.rewards.inventory_max.items():
            stat_max[k + ".amount"] = v
        limit_defs = []
        configured: set[str] = set()
        for lim in inv.limits.values():
            limit_defs.append(
                C.LimitDef([rid[n] for n in lim.resources], lim.limit, {rid[n]: b for n, b in lim.modifiers.items() if n in rid})
            )
            configured.update(lim.resources)
        for n in resource_names:
            if n not in configured:
                limit_defs.append(C.LimitDef([rid[n]], default_limit))
        inv_cfg = C.InventoryConfig()
        inv_cfg.limit_defs = limit_defs
        dmg = C.DamageConfig()
        if first.damage is not None:
            dmg.threshold = {rid[k]: v for k, v in first.damage.threshold.items()}
            dmg.resources = {rid[k]: v for k, v in first.damage.resources.items()}
        gname = team_names.get(team, f"team_{team}")
        ac = C.AgentConfig(
            type_id=0,
            type_name="agent",
            group_id=team,
            group_name=gname,
            freeze_duration=first.freeze_duration,
            initial_vibe=first.initial_vibe,
            inventory_config=inv_cfg,
            stat_rewards=stat_rewards,
            stat_reward_max=stat_max,
            initial_inventory={rid[k]: v for k, v in inv.initial.items()},
            inventory_regen_amounts={
                vid[vn]: {rid[k]: a for k, a in amts.items()} for vn, amts in inv.regen_amounts.items()
            },
            diversity_tracked_resources=[rid[n] for n in first.diversity_tracked_resources if n in rid],
            damage_config=dmg,
        )
        ac.tag_ids = [tag_id[t] for t in first.tags]
        objects["agent." + gname] = ac
        objects[f"agent.team_{team}"] = ac
        if team == 0:
            objects["agent.default"] = ac
            objects["agent.agent"] = ac

    for key, o in game.objects.items():
        tag_ids = [tag_id[t] for t in o.tags]
        if isinstance(o, WallConfig):
            wc = C.WallConfig(type_id=type_id[key], type_name=key, initial_vibe=o.vibe)
            wc.tag_ids = tag_ids
            objects[o.map_name or key] = wc
        elif isinstance(o, AssemblerConfig):
            protos = []
            for pc in reversed(o.protocols):
                p = C.Protocol()
                p.min_agents = pc.min_agents
                p.vibes = sorted(vid[v] for v in pc.vibes)
                ins: dict[int, int] = {}
                for k, v in pc.input_resources.items():
                    ins[int(rid[k])] = int(v)
                outs: dict[int, int] = {}
                for k, v in pc.output_resources.items():
                    outs[int(rid[k])] = int(v)
                p.input_resources = ins
                p.output_resources = outs
                p.cooldown = pc.cooldown
                protos.append(p)
            ac = C.AssemblerConfig(type_id=type_id[key], type_name=key, initial_vibe=o.vibe)
            ac.tag_ids = tag_ids
            ac.protocols = protos
            ac.allow_partial_usage = o.allow_partial_usage
            ac.max_uses = o.max_uses
            ac.clip_immune = o.clip_immune
            ac.start_clipped = o.start_clipped
            ac.chest_search_distance = o.chest_search_distance
            objects[o.map_name or key] = ac
        elif isinstance(o, ChestConfig):
            limit_defs = []
            for lim in o.inventory.limits.values():
                res = [rid[n] for n in lim.resources if n in rid]
                if res:
                    limit_defs.append(C.LimitDef(res, lim.limit, {rid[n]: b for n, b in lim.modifiers.items() if n in rid}))
            inv_cfg = C.InventoryConfig()
            inv_cfg.limit_defs = limit_defs
            cc = C.ChestConfig(type_id=type_id[key], type_name=key, initial_vibe=o.vibe)
            cc.vibe_transfers = {vid[vn]: {rid[r]: d for r, d in ds.items()} for vn, ds in o.vibe_transfers.items()}
            cc.initial_inventory = {rid[r]: a for r, a in o.inventory.initial.items()}
            cc.inventory_config = inv_cfg
            cc.tag_ids = tag_ids
            objects[o.map_name or key] = cc
        else:
            raise ValueError(f"Unknown object type: {key}")

    def res_pair(ac) -> dict[str, dict[int, int]]:
        if not ac.enabled:
            return {"consumed_resources": {}, "required_resources": {}}
        consumed = {rid[k]: int(v) for k, v in ac.consumed_resources.items()}
        src = ac.required_resources or ac.consumed_resources
        return {"consumed_resources": consumed, "required_resources": {rid[k]: int(v) for k, v in src.items()}}

    A = game.actions
    actions: dict[str, Any] = {}
    actions["noop"] = C.ActionConfig(**res_pair(A.noop))
    actions["move"] = C.MoveActionConfig(allowed_directions=list(A.move.allowed_directions), **res_pair(A.move))
    atk = A.attack
    actions["attack"] = C.AttackActionConfig(
        defense_resources={rid[k]: v for k, v in atk.defense_resources.items()},
        armor_resources={rid[k]: v for k, v in atk.armor_resources.items()},
        weapon_resources={rid[k]: v for k, v in atk.weapon_resources.items()},
        success=C.AttackOutcome(
            {rid[k]: v for k, v in atk.success.actor_inv_delta.items()},
            {rid[k]: v for k, v in atk.success.target_inv_delta.items()},
            [rid[n] for n in atk.success.loot],
            atk.success.freeze,
        ),
        enabled=atk.enabled,
        vibes=[vid[v] for v in atk.vibes],
        vibe_bonus={vid[v]: b for v, b in atk.vibe_bonus.items()},
        **res_pair(atk),
    )
    xf = A.transfer
    actions["transfer"] = C.TransferActionConfig(
        required_resources={rid[k]: int(v) for k, v in xf.required_resources.items()},
        vibe_transfers={
            vid[vt.vibe]: C.VibeTransferEffect({rid[k]: v for k, v in vt.target.items()}, {rid[k]: v for k, v in vt.actor.items()})
            for vt in xf.vibe_transfers
        },
        enabled=xf.enabled,
    )
    actions["change_vibe"] = C.ChangeVibeActionConfig(
        number_of_vibes=len(A.change_vibe.vibes) if A.change_vibe.enabled else 0, **res_pair(A.change_vibe)
    )

    go = game.global_obs
    feature_ids = {f.name: f.id for f in game.id_map().features()}
    return C.GameConfig(
        num_agents=game.num_agents,
        max_steps=game.max_steps,
        episode_truncates=game.episode_truncates,
        obs_width=game.obs.width,
        obs_height=game.obs.height,
        resource_names=resource_names,
        vibe_names=vibe_names,
        num_observation_tokens=game.obs.num_tokens,
        global_obs=C.GlobalObsConfig(
            episode_completion_pct=go.episode_completion_pct,
            last_action=go.last_action,
            last_reward=go.last_reward,
            compass=go.compass,
            goal_obs=go.goal_obs,
        ),
        feature_ids=feature_ids,
        actions=actions,
        objects=objects,
        tag_id_map={i: t for t, i in tag_id.items()},
        collectives=_collectives(C, game, rid),
        clipper=_clipper(C, game, rid, vid),
        protocol_details_obs=game.protocol_details_obs,
        inventory_regen_interval=game.inventory_regen_interval,
        token_value_base=game.obs.token_value_base,
    )


def _clipper(C, game, rid, vid):
    """mettagrid_c_config.py:506-539 (list order kept; vibes sorted; maps filled item by item)."""
    clip = game.clipper
    if clip is None:
        return None
    protos = []
    for pc in clip.unclipping_protocols:
        p = C.Protocol()
        p.min_agents = pc.min_agents
        p.vibes = sorted(vid[v] for v in pc.vibes)
        ins: dict[int, int] = {}
        for k, v in pc.input_resources.items():
            ins[int(rid[k])] = int(v)
        outs: dict[int, int] = {}
        for k, v in pc.output_resources.items():
            outs[int(rid[k])] = int(v)
        p.input_resources = ins
        p.output_resources = outs
        p.cooldown = pc.cooldown
        protos.append(p)
    cc = C.ClipperConfig()
    cc.unclipping_protocols = protos
    cc.length_scale = clip.length_scale
    cc.scaled_cutoff_distance = clip.scaled_cutoff_distance
    cc.clip_period = clip.clip_period
    return cc


def _collectives(C, game, rid) -> dict:
    """mettagrid_c_config.py:544-575."""
    out = {}
    for cc in game.collectives:
        inv = C.InventoryConfig()
        defs = []
        for lim in cc.inventory.limits.values():
            ids = [rid[n] for n in lim.resources if n in rid]
            if ids:
                defs.append(C.LimitDef(ids, lim.limit, {rid[n]: b for n, b in lim.modifiers.items() if n in rid}))
        inv.limit_defs = defs
        c = C.CollectiveConfig(cc.name)
        c.inventory_config = inv
        c.initial_inventory = {rid[k]: int(v) for k, v in cc.inventory.initial.items() if k in rid}
        out[cc.name] = c
    return out


class RefEnv:
    """One reference `MettaGrid` with caller-owned numpy buffers (the Puffer path: ctor + set_buffers)."""

    def __init__(self, game, grid_names, seed: int, with_buffers: bool = True):
        C = load_reference_module()
        self.C = C
        self.cfg = to_ref_game_config(game)
        grid = grid_names.tolist() if isinstance(grid_names, np.ndarray) else grid_names
        self.env = C.MettaGrid(self.cfg, grid, int(seed))
        A, T = game.num_agents, game.obs.num_tokens
        if with_buffers:
            self.obs = np.zeros((A, T, 3), dtype=np.uint8)
            self.terminals = np.zeros(A, dtype=bool)
            self.truncations = np.zeros(A, dtype=bool)
            self.rewards = np.zeros(A, dtype=np.float32)
            self.actions = np.zeros(A, dtype=np.int32)
            self.env.set_buffers(self.obs, self.terminals, self.truncations, self.rewards, self.actions)
        else:
            self.obs = self.env.observations()
            self.terminals = self.env.terminals()
            self.truncations = self.env.truncations()
            self.rewards = self.env.rewards()
            self.actions = self.env.actions()

    def step(self, actions) -> None:
        self.actions[:] = actions
        self.env.step()
