"""Feature-family fixtures through the REAL `convert_to_cpp_game_config` (run in the build container only).

    make -C oracle ref && python oracle/gen_golden_features.py

tests/golden/arena24_* and navigation4_* pin the config conversion for the arena / navigation presets only.  This
script closes the gap for every other feature of the step path: for a list of seeds, `tests/fuzz_configs.rand_cfg`
authors a feature-dense configuration in this repository's schema (vibe-triggered attack with defence / armor /
weapon / loot / freeze, swaps, transfer, chests, damage, regeneration, partial usage, max_uses, multi-vibe protocols,
limit modifiers, compass / goal tokens, odd windows, small token budgets, token bases).  Its dump is validated by the
REFERENCE's pydantic `GameConfig` (so a field this schema names or defaults differently is an error here), converted
by the reference's own `convert_to_cpp_game_config` (packages/mettagrid/python/src/mettagrid/config/
mettagrid_c_config.py:30-577) and stepped by the reference's compiled core.  Recorded: map, seed, action stream and
the expected per-step digests / final stats, and the configuration itself (this repository's schema, JSON).  The tests
load that configuration and push it through metta_amd.compile + oracle + HIP: a wrong default or field mapping in compile.py (or in oracle/ref_bridge.py,
whose hand-written mapping is cross-checked here against the real converter, step by step) shows up as a digest
mismatch.
"""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle.gen_golden import GOLD, digest, encode_names, install_reference_shims  # noqa: E402

SEEDS = [101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117,
         119, 127, 130, 133, 141, 143, 145, 146, 155, 164, 401, 403, 407]
# + a Clipper.  The reference picks the next assembler to clip by walking std::set<Assembler*> (clipper.hpp:24-26,
# 207-229): POINTER order, so two reference instances in one process can disagree (seen here: seed 119 diverges at
# step 4 between two instances built back to back).  Each of these fixtures is therefore produced by the FIRST
# MettaGrid of a fresh process (`--only SEED`), where glibc hands out the assemblers at ascending addresses, i.e. in
# creation (row-major) order -- the order oracle/mg_oracle.cpp restates; no second instance runs beside it.
# (seeds picked for what happens in their 200 steps: 20-60 clip events and as many unclips, stations that start clipped,
# clip-immune stations, 1-3 unclipping protocols, automatic length scale, clip_period 0 and 1)
CLIPPER_SEEDS = {119, 127, 130, 133, 141, 143, 145, 146, 155, 164, 401, 403, 407}
# stations that start clipped under unclipping protocols of DIFFERENT shapes, protocol tokens in the observations: the
# `tokens_*` statistics of the constructor's passes depend on the environment's seed (which protocol each station drew)
ANY_SHAPE_SEEDS = {401, 403, 407}
WIDE_SEEDS = {116: 24, 117: 32}  # resource lists beyond 16: one team starts with every resource (map growth 13 -> 29 -> 59)
COLLECTIVE_SEEDS = {113, 114, 115}  # + collectives: membership by tag, stat rewards on the collective's statistics
STEPS = 200


def features_of(game) -> dict:
    asm = [o for o in game.objects.values() if type(o).__name__ == "AssemblerConfig"]
    return {
        "attack": bool(game.actions.attack.enabled),
        "transfer": bool(game.actions.transfer.enabled),
        "chest": any(type(o).__name__ == "ChestConfig" for o in game.objects.values()),
        "damage": any(a.damage is not None for a in game.agents),
        "regen": game.inventory_regen_interval > 0 and any(a.inventory.regen_amounts for a in game.agents),
        "partial_usage": any(a.allow_partial_usage for a in asm),
        "max_uses": any(a.max_uses > 0 for a in asm),
        "multi_vibe_protocol": any(len(p.vibes) > 1 for a in asm for p in a.protocols),
        "limit_modifiers": any(l.modifiers for a in game.agents for l in a.inventory.limits.values()),
        "chest_search": any(a.chest_search_distance > 0 for a in asm),
        "move_costs": bool(game.actions.move.consumed_resources),
        "collectives": bool(game.collectives) and any("aligned.agent" in a.rewards.stats for a in game.agents),
        "wide_resources": len(game.resource_names) > 16,
        "clipper": game.clipper is not None,
        "clipper_start_clipped": game.clipper is not None and any(a.start_clipped and not a.clip_immune for a in asm),
        "token_base": int(game.obs.token_value_base),
    }


def main(only: int | None = None) -> None:
    install_reference_shims()
    from mettagrid.config.mettagrid_c_config import convert_to_cpp_game_config
    from mettagrid.config.mettagrid_config import GameConfig as RefGameConfig
    from mettagrid.mettagrid_c import MettaGrid

    from metta_amd.compile import compile_game
    from oracle.ref_bridge import RefEnv
    from tests.fuzz_configs import fuzz_actions, rand_cfg

    covered: dict[str, int] = {}
    for seed in SEEDS if only is None else [only]:
        if seed in CLIPPER_SEEDS and only is None:
            import subprocess

            subprocess.run([sys.executable, __file__, "--only", str(seed)], check=True)
            covered["clipper"] = covered.get("clipper", 0) + 1
            covered["clipper_start_clipped"] = covered.get("clipper_start_clipped", 0) + 1  # (asserted by --only runs below)
            continue
        cfg = rand_cfg(np.random.RandomState(seed), device_safe=True, collectives=seed in COLLECTIVE_SEEDS,
                       num_resources=WIDE_SEEDS.get(seed, 0), clipper=seed in CLIPPER_SEEDS,
                       start_clipped_any_shape=seed in ANY_SHAPE_SEEDS)
        game = cfg.game
        if seed in ANY_SHAPE_SEEDS:
            game.protocol_details_obs = True
        if seed in WIDE_SEEDS:
            game.agents[0].inventory.initial = {r: int(1 + (i * 7) % 11) for i, r in enumerate(game.resource_names)}
            game.agents[0].inventory.default_limit = 65535
        grid = np.asarray(game.map_builder.build(np.random.default_rng(seed)))
        dump = game.model_dump(mode="json", exclude={"map_builder"})
        ref_game = RefGameConfig.model_validate(dump)  # the reference's schema accepts (and re-defaults) our dump
        c_cfg = convert_to_cpp_game_config(ref_game.model_dump())
        A, T = game.num_agents, game.obs.num_tokens
        env = MettaGrid(c_cfg, grid.tolist(), seed)
        obs = np.zeros((A, T, 3), np.uint8)
        term, trunc = np.zeros(A, bool), np.zeros(A, bool)
        rew, act = np.zeros(A, np.float32), np.zeros(A, np.int32)
        env.set_buffers(obs, term, trunc, rew, act)
        solo = seed in CLIPPER_SEEDS
        # the hand-written mapping of oracle/ref_bridge.py, side by side (not beside a Clipper: see CLIPPER_SEEDS)
        bridge = None if solo else RefEnv(game, grid, seed)
        assert solo or np.array_equal(obs, bridge.obs), f"seed {seed}: ref_bridge differs from the real converter at step 0"
        obs0 = obs.copy()
        nA = len(compile_game(cfg, init_obs_passes=2).action_names)
        rs = np.random.RandomState(seed + 7)
        actions = np.stack([fuzz_actions(rs, game, nA) for _ in range(STEPS)])
        od, rd, sd = (np.zeros(STEPS, np.uint64) for _ in range(3))
        for t in range(STEPS):
            act[:] = actions[t]
            env.step()
            if not solo:
                bridge.step(actions[t])
                assert np.array_equal(obs, bridge.obs) and np.array_equal(rew, bridge.rewards), (
                    f"seed {seed}: ref_bridge differs from the real converter at step {t + 1}"
                )
            od[t], rd[t] = digest(obs), digest(rew)
            sd[t] = digest(np.array(env.action_success(), dtype=np.uint8))
        stats = env.get_episode_stats()
        assert solo or stats == bridge.env.get_episode_stats(), f"seed {seed}: stats differ between the two conversions"
        names, codes = encode_names(grid)
        feats = features_of(game)
        for k, v in feats.items():
            if v and k != "token_base":
                covered[k] = covered.get(k, 0) + 1
        meta = {"seed": seed, "cell_names": names, "steps": STEPS, "stats": stats, "features": feats, "game": dump,
                "terminated": bool(term.all()), "truncated": bool(trunc.all())}
        np.savez_compressed(
            GOLD / f"feature_{seed}.npz", meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), map_codes=codes,
            actions=actions.astype(np.int32), obs0=obs0 if solo else bridge_obs0(game, grid, seed), obs_final=obs.copy(),
            obs_digest=od,
            rew_digest=rd, succ_digest=sd, episode_rewards=np.array(env.get_episode_rewards(), dtype=np.float32),
        )
        print(f"wrote feature_{seed}.npz", {k: v for k, v in feats.items() if v})
    if only is not None:
        return
    print("feature coverage over the fixture set:", covered)
    need = ["attack", "transfer", "chest", "damage", "regen", "partial_usage", "max_uses", "multi_vibe_protocol",
            "limit_modifiers", "chest_search", "collectives", "wide_resources", "clipper", "clipper_start_clipped"]
    missing = [k for k in need if not covered.get(k)]
    assert not missing, f"no fixture exercises {missing}: extend SEEDS"


def bridge_obs0(game, grid, seed):
    from oracle.ref_bridge import RefEnv

    return RefEnv(game, grid, seed).obs.copy()


if __name__ == "__main__":
    main(int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "--only" else None)
