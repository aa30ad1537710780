"""GPU parity: the HIP step path (through the C-ABI) against the CPU oracle, the golden fixtures emitted by the
reference, and -- when oracle/_ref travelled with the tree -- the reference's own compiled core.

Bar (BASELINE.json north_star): observations, grid state, inventories bit-exact; rewards equal (they are sums of
exactly representable f32 values here, so we require equality, which is stricter than the stated 1e-6 relative).
"""

import numpy as np
import pytest

from metta_amd.builder import make_arena, make_combat_stress, make_navigation
from metta_amd.compile import compile_game
from tests.fuzz_configs import fuzz_actions, rand_cfg
from tests.helpers import digest, fixture_actions, fixture_grid, load_fixture, load_map_pool, stats_equal

pytestmark = pytest.mark.gpu


def _batch(cfg, maps, num_envs, seeds, map_ids, passes=2):
    from metta_amd.batch import MettaGridBatch

    return MettaGridBatch(cfg, num_envs, maps, device=0, init_obs_passes=passes, seeds=seeds, map_ids=map_ids)


def _oracles(cfg, maps, seeds, map_ids, passes=2):
    from oracle.mg_oracle import make_oracle_env

    comp = compile_game(cfg, init_obs_passes=passes)
    return comp, [make_oracle_env(comp, comp.encode_map(maps[m]), int(s)) for s, m in zip(seeds, map_ids)]


def _compare_step(b, oras, t, A):
    E = len(oras)
    obs = b.observations.cpu().numpy().reshape(E, A, -1, 3)
    rew = b.rewards.cpu().numpy().reshape(E, A)
    term = b.terminals.cpu().numpy().reshape(E, A)
    trunc = b.truncations.cpu().numpy().reshape(E, A)
    succ = b.action_success.cpu().numpy().reshape(E, A)
    for e, o in enumerate(oras):
        if not np.array_equal(obs[e], o.obs):
            bad = np.argwhere(obs[e] != o.obs)
            a = bad[0][0]
            raise AssertionError(
                f"step {t} env {e}: obs differ at {bad[:4].tolist()}\n hip={obs[e][a][:24].tolist()}\n ora={o.obs[a][:24].tolist()}"
            )
        assert np.array_equal(rew[e], o.rewards), f"step {t} env {e}: rewards {rew[e]} vs {o.rewards}"
        assert np.array_equal(term[e], o.terminals.astype(bool)), f"step {t} env {e}: terminals"
        assert np.array_equal(trunc[e], o.truncations.astype(bool)), f"step {t} env {e}: truncations"
        assert np.array_equal(succ[e], o.action_success.astype(bool)), f"step {t} env {e}: action_success"


def _compare_state(b, oras, t):
    for e, o in enumerate(oras):
        hs, os_ = b.agent_state(e), o.agent_state()
        assert np.array_equal(hs, os_), f"step {t} env {e}: agent state\n{hs}\n{os_}"
        assert np.array_equal(b.assembler_state(e), o.assembler_state()), f"step {t} env {e}: assembler state"
        assert np.array_equal(b.chest_state(e), o.chest_state()), f"step {t} env {e}: chest state\n{b.chest_state(e)}\n{o.chest_state()}"
        d = stats_equal(b.get_episode_stats(e), o.get_episode_stats(), ignore_invalid_variants=False)
        assert d is None, f"step {t} env {e}: stats differ: {d}"
        assert np.array_equal(
            b.episode_rewards.cpu().numpy().reshape(len(oras), -1)[e], o.episode_rewards
        ), f"step {t} env {e}: episode rewards"
        assert b.current_step(e) == o.current_step


@pytest.mark.parametrize(
    "path_env",
    [{}, {"MG_DIAG": "8"}, {"MG_FORCE_GENERAL": "1"}, {"MG_FUSED_O7": "0"}, {"MG_FUSED_O7": "0", "MG_DIAG": "8"}, {"MG_NO_FUSED": "1"},
     {"MG_FUSED_W1": "1"}, {"MG_FUSED_W1": "1", "MG_DIAG": "8"}, {"MG_FUSED_W1": "8"}],
    ids=["fast", "fast-serial-redraw", "general", "fused-8-waves-build", "fused-8-waves-build-serial-redraw", "three-launch",
         "fused-one-wave-build", "fused-one-wave-build-serial-redraw", "fused-one-wave-8-waves-build"],
)
def test_arena_vs_oracle(path_env, monkeypatch):
    """The same comparison through each act-kernel path: the wave-parallel fast kernel (+ the assembler-use
    kernel), its serial shuffle redraw (taken on a Lemire rejection, about once in 1e6 env-steps; MG_DIAG bit 8
    forces it), and the general sequential kernel.  The variables are read by mg_create."""
    for k, v in path_env.items():
        monkeypatch.setenv(k, v)
    cfg = make_arena(24)
    maps = load_map_pool()[:6]
    E, A = 12, 24
    seeds = [0, 1, 42, 7, 1000, 2**31 + 5, 3, 4, 5, 6, 8, 9]
    map_ids = [e % len(maps) for e in range(E)]
    b = _batch(cfg, maps, E, seeds, map_ids)
    comp, oras = _oracles(cfg, maps, seeds, map_ids)
    _compare_step(b, oras, 0, A)
    _compare_state(b, oras, 0)
    rs = np.random.RandomState(123)
    for t in range(1, 401):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        b.step(acts.reshape(-1))
        for e, o in enumerate(oras):
            o.step(acts[e])
        _compare_step(b, oras, t, A)
        if t % 100 == 0:
            _compare_state(b, oras, t)
    b.close()


@pytest.mark.parametrize("seed", list(range(48)))
def test_fuzz_vs_oracle(seed):
    rs = np.random.RandomState(seed)
    cfg = rand_cfg(rs, device_safe=True, collectives=seed >= 40)  # 40..47: + collectives and rewards on their statistics
    game = cfg.game
    passes = int(rs.randint(1, 3))
    maps = [game.map_builder.build(np.random.default_rng(1000 * seed + i)) for i in range(3)]
    E, A = 5, game.num_agents
    seeds = [seed * 10 + e for e in range(E)]
    map_ids = [e % 3 for e in range(E)]
    b = _batch(cfg, maps, E, seeds, map_ids, passes)
    b.set_auto_reset(False)
    comp, oras = _oracles(cfg, maps, seeds, map_ids, passes)
    nA = len(comp.action_names)
    _compare_step(b, oras, 0, A)
    for t in range(1, 161):
        acts = np.stack([fuzz_actions(rs, game, nA) for _ in range(E)])
        if rs.randint(30) == 0:
            e, ag = int(rs.randint(E)), int(rs.randint(A))
            inv = {int(rs.randint(len(game.resource_names))): int(rs.randint(0, 400)) for _ in range(rs.randint(0, 4))}
            b.set_inventory(e, ag, inv)
            oras[e].set_inventory(ag, inv)
        b.step(acts.reshape(-1))
        for e, o in enumerate(oras):
            o.step(acts[e])
        _compare_step(b, oras, t, A)
        if t % 40 == 0:
            _compare_state(b, oras, t)
    b.close()


@pytest.mark.parametrize("name", ["arena24_seed0", "arena24_seed1", "arena24_seed42"])
def test_golden_arena(name):
    """Fixture emitted by the reference (oracle/gen_golden.py): per-step digests of obs / rewards / success.
    `arena24_seed0` is one full 10 000-step episode: the f32 game stats `tokens_free_space` / `tokens_written` pass
    2^24 (after ~4.6 k / ~15 k steps, SURVEY hard part 3), so their final values only match if the device adds them
    in the reference's order with f32 rounding; the episode ends by itself at max_steps (mettagrid_c.cpp:667-677)."""
    z, meta = load_fixture(name)
    cfg = make_arena(24)
    grid = fixture_grid(z, meta)
    b = _batch(cfg, [grid], 1, [meta["seed"]], [0])
    assert np.array_equal(b.observations.cpu().numpy(), z["obs0"])
    acts = fixture_actions(meta, 24)
    snaps = set(meta["snap_steps"])
    for t in range(len(acts)):
        b.step(acts[t])
        if t % 10 == 0 or (t + 1) in snaps or t == len(acts) - 1:
            obs = b.observations.cpu().numpy()
            assert digest(obs) == int(z["obs_digest"][t]), f"{name}: obs digest differs at step {t + 1}"
            assert digest(b.rewards.cpu().numpy()) == int(z["rew_digest"][t]), f"{name}: reward digest at step {t + 1}"
            assert digest(b.action_success.cpu().numpy().astype(np.uint8)) == int(z["succ_digest"][t])
            if (t + 1) in snaps:
                assert np.array_equal(obs, z[f"snap_{t + 1}"])
    assert np.array_equal(b.observations.cpu().numpy(), z["obs_final"])
    assert np.array_equal(b.episode_rewards.cpu().numpy(), z["episode_rewards"])
    st = b.agent_state(0)
    assert np.array_equal(st[:, :2], z["agent_rc"])
    for a, inv in enumerate(meta["agent_inv"]):
        got = [[int(st[a, 9 + 2 * k]), int(st[a, 10 + 2 * k])] for k in range(st[a, 8])]
        assert got == inv, f"agent {a} inventory order/amounts"
    d = stats_equal(b.get_episode_stats(0), meta["stats"])
    assert d is None, d
    if name == "arena24_seed0":
        assert len(acts) == 10000 and bool(b.terminals.all().item()) and not bool(b.truncations.any().item())
        assert meta["stats"]["game"]["tokens_free_space"] > 2**24  # the rounding regime was really entered
    b.close()


@pytest.mark.parametrize("seed,num_resources", [(s, r) for r in (17, 24, 32) for s in range(200 + r, 204 + r)])
def test_fuzz_wide_resource_lists_vs_oracle(seed, num_resources):
    """17..32 resources: the wide-inventory build (libmettagrid_hip_r32.so, 160-byte agent records).  One team starts
    with every resource, so the inventory's unordered_map emulation grows through 13 -> 29 -> 59 buckets."""
    from metta_amd import native

    rs = np.random.RandomState(seed)
    cfg = rand_cfg(rs, device_safe=True, num_resources=num_resources, collectives=bool(seed % 2))
    game = cfg.game
    full = game.agents[0]
    full.inventory.initial = {r: int(1 + (i * 7) % 11) for i, r in enumerate(game.resource_names)}
    full.inventory.default_limit = 65535
    maps = [game.map_builder.build(np.random.default_rng(77 * seed + i)) for i in range(2)]
    E, A = 3, game.num_agents
    seeds = [seed * 10 + e for e in range(E)]
    map_ids = [e % 2 for e in range(E)]
    b = _batch(cfg, maps, E, seeds, map_ids)
    assert b.L is native.load(wide=True) and b.L.mg_max_resources() == 32
    b.set_auto_reset(False)
    comp, oras = _oracles(cfg, maps, seeds, map_ids)
    nA = len(comp.action_names)
    _compare_step(b, oras, 0, A)
    for t in range(1, 121):
        acts = np.stack([fuzz_actions(rs, game, nA) for _ in range(E)])
        if rs.randint(20) == 0:
            e, ag = int(rs.randint(E)), int(rs.randint(A))
            inv = {int(rs.randint(num_resources)): int(rs.randint(0, 400)) for _ in range(rs.randint(0, 24))}
            b.set_inventory(e, ag, inv)
            oras[e].set_inventory(ag, inv)
        b.step(acts.reshape(-1))
        for e, o in enumerate(oras):
            o.step(acts[e])
        _compare_step(b, oras, t, A)
        if t % 40 == 0:
            _compare_state(b, oras, t)
    b.close()


@pytest.mark.parametrize("seed", list(range(300, 316)) + list(range(400, 412)))
def test_fuzz_clipper_vs_oracle(seed):
    """A Clipper (systems/clipper.hpp) on top of the fuzzed configurations: clip events, stations that start clipped,
    unclipping by use, clip-immune stations, clip_period 0, automatic length scale.  Short episodes with auto-reset:
    every restart rebuilds the environment's clipper from its own seed (which unclipping protocol a station that starts
    clipped offers is drawn there).  Infection weights and the length scale are compared with the oracle's.
    Seeds from 400: stations start clipped under unclipping protocols of DIFFERENT shapes with protocol tokens in the
    observations and two constructor passes -- the `tokens_*` statistics of an episode's start then depend on the
    environment's seed, and a restart recounts them (`mg_recount_initial_tokens`; refused until round 3); the statistics
    are compared right after every restart."""
    from oracle.mg_oracle import make_oracle_env

    rs = np.random.RandomState(seed)
    any_shape = seed >= 400
    cfg = rand_cfg(rs, device_safe=True, clipper=True, collectives=bool(seed % 4 == 0), start_clipped_any_shape=any_shape)
    compile_game(cfg, init_obs_passes=2)
    game = cfg.game
    game.max_steps = int(rs.choice([20, 35, 50]))
    passes = int(rs.randint(1, 3))
    if any_shape:
        game.protocol_details_obs = True
        passes = 2
        if not any(getattr(o, "start_clipped", False) and not getattr(o, "clip_immune", False) for o in game.objects.values()):
            next(o for o in game.objects.values() if hasattr(o, "start_clipped")).start_clipped = True
    maps = [game.map_builder.build(np.random.default_rng(1000 * seed + i)) for i in range(3)]
    E, A = 4, game.num_agents
    seeds = [seed * 10 + e for e in range(E)]
    cur_map = [e % 3 for e in range(E)]
    b = _batch(cfg, maps, E, seeds, cur_map, passes)
    b.set_auto_reset(True, 1)
    comp, oras = _oracles(cfg, maps, seeds, cur_map, passes)
    nA = len(comp.action_names)
    _compare_step(b, oras, 0, A)

    def compare_clipper(t):
        for e, o in enumerate(oras):
            ls, w = o.clipper_state()
            dls, dw = b.clipper_state(e)
            assert dls == ls, f"step {t} env {e}: length scale"
            assert np.array_equal(dw, w), f"step {t} env {e}: infection weights"

    compare_clipper(0)
    _compare_state(b, oras, 0)
    for t in range(1, 141):
        acts = np.stack([fuzz_actions(rs, game, nA) for _ in range(E)])
        restarted = False
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + 1) % 3
                oras[e] = make_oracle_env(comp, comp.encode_map(maps[cur_map[e]]), seeds[e])
                restarted = True
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
        if t % 10 == 0:
            compare_clipper(t)
        if t % 35 == 0 or (any_shape and restarted):
            _compare_state(b, oras, t)
    b.check_errors()
    b.close()


def test_arena_with_limit_modifiers_runs_the_fused_kernel_with_them():
    """Inventory-limit modifiers (inventory.cpp:87-158: a held resource raises a limit, losing it drops the excess) on an
    otherwise wave-parallel configuration: the fused kernel's build WITH the nested limit enforcement
    (`mg_step_fused_kernel`; arena and navigation themselves run the build without it)."""
    from metta_amd.config import ResourceLimitsConfig

    cfg = make_arena(24)
    ag = cfg.game.agent
    ag.inventory.limits["ore"] = ResourceLimitsConfig(limit=2, resources=["ore_red"], modifiers={"battery_red": 2})
    ag.inventory.limits["bat"] = ResourceLimitsConfig(limit=3, resources=["battery_red"], modifiers={"heart": 1})
    ag.inventory.initial = {"ore_red": 2, "battery_red": 1}
    maps = load_map_pool()[:4]
    E, A = 8, 24
    seeds = list(range(200, 200 + E))
    map_ids = [e % len(maps) for e in range(E)]
    b = _batch(cfg, maps, E, seeds, map_ids)
    assert b.step_path() == "mg_step_fused_o6_kernel"  # (8 environments: the larger register budget)
    comp, oras = _oracles(cfg, maps, seeds, map_ids)
    _compare_step(b, oras, 0, A)
    rs = np.random.RandomState(9)
    for t in range(1, 301):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        b.step(acts.reshape(-1))
        for e, o in enumerate(oras):
            o.step(acts[e])
        _compare_step(b, oras, t, A)
        if t % 100 == 0:
            _compare_state(b, oras, t)
    st = b.agent_state(0)
    assert (st[:, 8] > 0).any()
    b.check_errors()
    b.close()


def test_more_resources_than_the_wide_build_holds_is_an_error():
    cfg = rand_cfg(np.random.RandomState(1), device_safe=True, num_resources=33)
    maps = [cfg.game.map_builder.build(np.random.default_rng(1))]
    with pytest.raises(RuntimeError, match="33 resources, this build of the library holds 16"):
        _batch(cfg, maps, 1, [1], [0])


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111, 112, 113, 114, 115, 116, 117,
                                  119, 127, 130, 133, 141, 143, 145, 146, 155, 164, 401, 403, 407])  # (from 119 on: with a Clipper; 40x: start-clipped stations under protocols of different shapes)
def test_golden_feature_families(seed):
    """tests/golden/feature_<seed>.npz: configurations with attack / armor / loot / freeze, transfer, chests, damage,
    regeneration, partial usage ... converted by the reference's REAL `convert_to_cpp_game_config` and stepped by its
    core (oracle/gen_golden_features.py).  The HIP path, fed by metta_amd.compile, must reproduce the digests."""
    from tests.test_oracle_golden import load_feature_fixture

    z, meta, cfg, grid = load_feature_fixture(seed)
    A = cfg.game.num_agents
    b = _batch(cfg, [grid], 1, [seed], [0])
    b.set_auto_reset(False)
    assert np.array_equal(b.observations.cpu().numpy(), z["obs0"])
    acts = z["actions"]
    for t in range(len(acts)):
        b.step(acts[t])
        assert digest(b.observations.cpu().numpy()) == int(z["obs_digest"][t]), f"seed {seed}: obs digest at step {t + 1}"
        assert digest(b.rewards.cpu().numpy()) == int(z["rew_digest"][t]), f"seed {seed}: reward digest at step {t + 1}"
        assert digest(b.action_success.cpu().numpy().astype(np.uint8)) == int(z["succ_digest"][t])
    assert np.array_equal(b.observations.cpu().numpy(), z["obs_final"])
    assert np.array_equal(b.episode_rewards.cpu().numpy(), z["episode_rewards"])
    d = stats_equal(b.get_episode_stats(0), meta["stats"])
    assert d is None, d
    b.check_errors()
    assert A == len(z["episode_rewards"])
    b.close()


def test_golden_navigation():
    z, meta = load_fixture("navigation4_seed3")
    cfg = make_navigation(4)
    cfg.game.max_steps = meta["max_steps"]
    grid = fixture_grid(z, meta)
    b = _batch(cfg, [grid], 1, [meta["seed"]], [0])
    b.set_auto_reset(False)
    assert np.array_equal(b.observations.cpu().numpy(), z["obs0"])
    acts = fixture_actions(meta, 4)
    for t in range(len(acts)):
        b.step(acts[t])
        assert digest(b.observations.cpu().numpy()) == int(z["obs_digest"][t]), f"obs digest at step {t + 1}"
        assert digest(b.rewards.cpu().numpy()) == int(z["rew_digest"][t])
        done = bool(b.terminals.all().item() or b.truncations.all().item())
        assert done == (t + 1 >= meta["done_step"])
    assert np.array_equal(b.observations.cpu().numpy(), z["obs_final"])
    d = stats_equal(b.get_episode_stats(0), meta["stats"])
    assert d is None, d
    b.close()


def test_auto_reset_matches_puffer_semantics():
    """`MettaGridPufferEnv.step`: when all agents are done, the NEXT step first builds a new simulation (same seed,
    next map) and steps it (mettagrid_puffer_env.py:158-171)."""
    from oracle.mg_oracle import make_oracle_env

    cfg = make_arena(24)
    cfg.game.max_steps = 25
    maps = load_map_pool()[:4]
    E, A = 3, 24
    seeds, map_ids = [11, 12, 13], [0, 1, 2]
    b = _batch(cfg, maps, E, seeds, map_ids)
    b.set_auto_reset(True, 1)
    comp, oras = _oracles(cfg, maps, seeds, map_ids)
    cur_map = list(map_ids)
    rs = np.random.RandomState(5)
    episodes = 0
    for t in range(1, 80):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + 1) % len(maps)
                oras[e] = make_oracle_env(comp, comp.encode_map(maps[cur_map[e]]), seeds[e])
                episodes += 1
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
    _compare_state(b, oras, 79)
    assert episodes == 9
    s = b.finished_summary(clear=True)
    assert s[0] == 9 and s[1] == 9 * 25
    b.close()


@pytest.mark.parametrize("path_env", [{}, {"MG_FUSED_W1": "8"}, {"MG_FUSED_W1": "0"}], ids=["one-wave", "one-wave-8-waves-build", "two-waves"])
def test_mixed_map_sizes_vs_oracle(path_env, monkeypatch):
    """BASELINE config 3 (navigation suite): one pool holding maps of different sizes.  Bounds, the compass token
    (towards the centre of the map being played) and auto-reset across sizes against one oracle per (env, map) --
    through every build of the fused kernel a navigation handle can land on (the library picks by environment count)."""
    from metta_amd.mapgen import RandomMapBuilder
    from oracle.mg_oracle import make_oracle_env

    for k, v in path_env.items():
        monkeypatch.setenv(k, v)

    cfg = make_navigation(4)
    cfg.game.global_obs.compass = True
    cfg.game.max_steps = 40
    shapes = [(12, 17), (31, 20), (9, 9), (44, 72), (20, 88), (15, 14)]
    maps = [
        RandomMapBuilder(width=w, height=h, border_width=1, agents=4, objects={"wall": (h * w) // 9, "assembler": 3}).build(
            np.random.default_rng(100 + i)
        )
        for i, (h, w) in enumerate(shapes)
    ]
    E, A = 9, 4
    seeds = [50 + e for e in range(E)]
    cur_map = [e % len(maps) for e in range(E)]
    b = _batch(cfg, maps, E, seeds, list(cur_map))
    b.set_auto_reset(True, 1)
    want = {"": "mg_step_fused_nomod_w1_kernel", "8": "mg_step_fused_nomod_w18_kernel", "0": "mg_step_fused_nomod_o7_kernel"}
    assert b.step_path() == want[path_env.get("MG_FUSED_W1", "")], b.step_path()
    assert b.env_map_dims(3) == (44, 72) and b.env_map_dims(2) == (9, 9)
    comp = compile_game(cfg, init_obs_passes=2)
    oras = [make_oracle_env(comp, comp.encode_map(maps[m]), s) for s, m in zip(seeds, cur_map)]
    nA = len(comp.action_names)
    _compare_step(b, oras, 0, A)
    rs = np.random.RandomState(11)
    for t in range(1, 130):
        acts = np.where(rs.rand(E, A) < 0.85, rs.randint(0, 5, size=(E, A)), rs.randint(0, nA, size=(E, A))).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + 1) % len(maps)
                oras[e] = make_oracle_env(comp, comp.encode_map(maps[cur_map[e]]), seeds[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
    _compare_state(b, oras, 129)
    b.close()


@pytest.mark.parametrize(
    "agents,size,envs,steps",
    [(1024, 256, 2, 60), (256, 48, 3, 120), (32, 300, 2, 60)],
    ids=["256x256x1024-hbm-grid", "48x48x256-dense", "300x300x32-hbm-grid"],
)
def test_combat_stress_vs_oracle(agents, size, envs, steps):
    """BASELINE config 4: large-map combat (attack under the weapon vibe, armor defence with vibe bonus, freeze, loot,
    swaps with frozen agents, assembler cooldowns).  At 256 x 256 x 1024 agents and at 300 x 300 the map tile does not
    fit LDS: the kernels keep each environment's grid (and, for observations, the agent records) in HBM."""
    cfg = make_combat_stress(agents, size)
    cfg.game.max_steps = 50  # the episode end and the auto-reset (template restart) are part of the run
    maps = [cfg.game.map_builder.build(np.random.default_rng(7 + i)) for i in range(2)]
    seeds = [900 + e for e in range(envs)]
    cur_map = [e % 2 for e in range(envs)]
    b = _batch(cfg, maps, envs, seeds, list(cur_map))
    comp, oras = _oracles(cfg, maps, seeds, cur_map)
    from oracle.mg_oracle import make_oracle_env

    nA = len(comp.action_names)
    _compare_step(b, oras, 0, agents)
    rs = np.random.RandomState(21)
    for t in range(1, steps + 1):
        acts = np.where(rs.rand(envs, agents) < 0.7, rs.randint(1, 5, size=(envs, agents)), rs.randint(0, nA, size=(envs, agents)))
        acts = acts.astype(np.int32)
        for e in range(envs):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur_map[e] = (cur_map[e] + 1) % 2
                oras[e] = make_oracle_env(comp, comp.encode_map(maps[cur_map[e]]), seeds[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, agents)
        if t % 40 == 0:
            _compare_state(b, oras, t)
    _compare_state(b, oras, steps)
    b.close()


def test_shuffle_rejections_are_redrawn_in_parallel():
    """std::shuffle draws with Lemire's rejection (uniform_int_dist.h:258-269): a rejected draw takes another RNG output
    and every later draw moves one output on.  With 1024 agents about 4 % of the environment-steps hold one; the general
    action kernel settles them with a block scan (k_act_body.h) instead of sending the workgroup through the serial
    shuffle.  Here: 8 environments x 16 steps must contain rejections (seen as steps that consume more than A/2 outputs
    -- the shuffle is this configuration's only consumer), and every step equals the oracle."""
    from oracle.mg_oracle import make_oracle_env  # noqa: F401

    agents, envs, steps = 1024, 8, 16
    cfg = make_combat_stress(agents, 256)
    maps = [cfg.game.map_builder.build(np.random.default_rng(70))]
    seeds = [4000 + e for e in range(envs)]
    b = _batch(cfg, maps, envs, seeds, [0] * envs)
    comp, oras = _oracles(cfg, maps, seeds, [0] * envs)
    nA = len(comp.action_names)
    rs = np.random.RandomState(5)
    D = agents // 2
    idx = b.env_table()[:, 1].astype(np.int64)
    extra = 0
    for t in range(1, steps + 1):
        acts = np.where(rs.rand(envs, agents) < 0.7, rs.randint(1, 5, size=(envs, agents)), rs.randint(0, nA, size=(envs, agents)))
        acts = acts.astype(np.int32)
        for e in range(envs):
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, agents)
        now = b.env_table()[:, 1].astype(np.int64)
        used = (now - idx) % 624
        idx = now
        assert (used >= D).all() and (used <= D + 3).all(), used
        extra += int((used - D).sum())
    assert extra >= 2, "no rejected draw in this run: choose other seeds"
    _compare_state(b, oras, steps)
    b.check_errors()
    b.close()


def test_vs_reference_core():
    """HIP against the reference's own compiled C++ core (oracle/_ref), when it travelled with the tree."""
    from oracle.ref_bridge import RefEnv, have_reference

    if not have_reference():
        pytest.skip("oracle/_ref not built")
    cfg = make_arena(24)
    maps = load_map_pool()[:2]
    seeds, map_ids = [5, 6], [0, 1]
    b = _batch(cfg, maps, 2, seeds, map_ids)
    refs = [RefEnv(cfg.game, maps[m], s) for s, m in zip(seeds, map_ids)]
    rs = np.random.RandomState(9)
    for t in range(300):
        acts = rs.randint(0, 5, size=(2, 24)).astype(np.int32)
        b.step(acts.reshape(-1))
        obs = b.observations.cpu().numpy().reshape(2, 24, 200, 3)
        rew = b.rewards.cpu().numpy().reshape(2, 24)
        for e, r in enumerate(refs):
            r.step(acts[e])
            assert np.array_equal(obs[e], r.obs), f"step {t} env {e}"
            assert np.array_equal(rew[e], r.rewards)
    for e, r in enumerate(refs):
        d = stats_equal(b.get_episode_stats(e), r.env.get_episode_stats())
        assert d is None, d
        # grid_objects (mettagrid_c.cpp:754-918): every key our host side reproduces, for every object
        ours, theirs = b.grid_objects(e), r.env.grid_objects()
        assert set(ours) == set(int(k) for k in theirs)
        for oid, od in ours.items():
            td = theirs[oid]
            for k, v in od.items():
                tv = td[k]
                tv = {int(a): int(q) for a, q in tv.items()} if isinstance(tv, dict) else tv
                assert tv == v, f"env {e} object {oid} key {k}: {v} vs {tv}"
        sub = b.grid_objects(e, 10, 30, 5, 40, ignore_types=["wall"])
        ref_sub = r.env.grid_objects(10, 30, 5, 40, ["wall"])
        assert set(sub) == set(int(k) for k in ref_sub)
    b.close()


def test_full_size_properties():
    """BASELINE config 2 size (4096 envs x 24 agents): size-independent properties.
    (1) determinism: two batches with the same seeds/maps/actions give identical buffers;
    (2) replication: envs that share seed+map+actions are identical row for row;
    (3) a sampled subset matches the oracle after K steps."""
    import torch

    cfg = make_arena(24)
    maps = load_map_pool()
    E, A, P = 4096, 24, len(maps)
    seeds = [e % 512 for e in range(E)]  # env e and e+512k are replicas (512 % 64 == 0 -> same map too)
    map_ids = [e % P for e in range(E)]
    b1 = _batch(cfg, maps, E, seeds, map_ids)
    b2 = _batch(cfg, maps, E, seeds, map_ids)
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        # more environments than one round of the two-wave build holds (14 per CU), at most one round of the one-wave
        # build (16 per CU by LDS): the library steps them all at once with the latter
        assert b1.step_path() == "mg_step_fused_nomod_w1_kernel", b1.step_path()
    g = torch.Generator(device="cpu").manual_seed(0)
    K = 64
    base_actions = torch.randint(0, 5, (K, 512, A), generator=g, dtype=torch.int32)
    for t in range(K):
        acts = base_actions[t].repeat(E // 512, 1).reshape(-1).cuda()
        b1.step(acts)
        b2.step(acts)
    o1 = b1.observations.reshape(E // 512, 512, A, 200, 3)
    assert torch.equal(b1.observations, b2.observations) and torch.equal(b1.rewards, b2.rewards)
    assert torch.equal(b1.episode_rewards, b2.episode_rewards)
    assert bool((o1 == o1[0:1]).all().item()), "replica envs diverged"
    comp, oras = _oracles(cfg, maps, seeds[:6], map_ids[:6])
    for t in range(K):
        for e, o in enumerate(oras):
            o.step(base_actions[t, e].numpy())
    obs = b1.observations.cpu().numpy().reshape(E, A, 200, 3)
    for e, o in enumerate(oras):
        assert np.array_equal(obs[e], o.obs)
        assert np.array_equal(obs[e + 512 * 3], o.obs)
    b1.close()
    b2.close()


@pytest.mark.parametrize("path_env", [{}, {"MG_NO_FUSED": "1"}, {"MG_FORCE_GENERAL": "1"}], ids=["fused", "fast", "general"])
def test_invalid_action_ids_keep_their_keys(path_env, monkeypatch):
    """`action.invalid_index.<id>` for ids far from the action table (mettagrid_c.cpp:565-570): the (id, count) pairs at
    the end of the agent's stat row, on every step path, against the oracle (which
    tests/test_oracle_golden.py::test_invalid_action_ids_make_the_reference_keys pins to the reference's core)."""
    from metta_amd.batch import MettaGridBatch
    from oracle.mg_oracle import make_oracle_env

    for k, v in path_env.items():
        monkeypatch.setenv(k, v)
    cfg = make_arena(24)
    cfg.game.max_steps = 45
    comp = compile_game(cfg, init_obs_passes=2)
    maps = load_map_pool()[:2]
    E, A = 2, 24
    b = MettaGridBatch(cfg, E, maps, device=0, seeds=[3, 4], map_ids=[0, 1])
    b.set_auto_reset(True, 1)
    oras = [make_oracle_env(comp, comp.encode_map(maps[m]), s) for m, s in ((0, 3), (1, 4))]
    wild = [1000, -7, 2**31 - 1, -(2**31), 5, 11, -1, 77, 78]
    rs = np.random.RandomState(2)
    for t in range(1, 44):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        mask = rs.rand(E, A) < 0.15
        acts[mask] = np.asarray(wild, dtype=np.int64)[rs.randint(len(wild), size=int(mask.sum()))].astype(np.int32)
        for e in range(E):
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
    for e in range(E):
        d = stats_equal(b.get_episode_stats(e), oras[e].get_episode_stats(), ignore_invalid_variants=False)
        assert d is None, d
        keys = set().union(*[set(a) for a in b.get_episode_stats(e)["agent"]])
        assert "action.invalid_index.1000" in keys and "action.invalid_index.-2147483648" in keys
    b.check_errors()
    b.close()
