"""Per-slot configurations and the curriculum vecenv (reference `CurriculumEnv`,
metta/cogworks/curriculum/curriculum_env.py:96-142): an environment takes up a new MettaGridConfig at every episode
end.  Parity: every episode of every environment equals the oracle built from THAT episode's (configuration, map)."""

import copy

import numpy as np
import pytest

from metta_amd.builder import make_arena
from metta_amd.compile import compile_game
from metta_amd.config import ResourceLimitsConfig
from tests.helpers import load_map_pool
from tests.test_gpu_parity import _compare_state, _compare_step

pytestmark = pytest.mark.gpu


def _tasks():
    """What the reference's arena curriculum buckets over (recipes/prod/arena_basic_easy_shaped.py:60-80) and a few
    more: reward weights and caps, attack cost, episode length, initial inventories, limits, protocol cooldowns."""
    base = make_arena(24)
    base.game.max_steps = 40
    t1 = copy.deepcopy(base)
    t1.game.max_steps = 25
    t1.game.agent.rewards.inventory = {"heart": 0.5, "ore_red": 0.1, "battery_red": 0.9}
    t1.game.agent.rewards.inventory_max = {"ore_red": 2, "battery_red": 1}
    t1.game.actions.attack.consumed_resources = {"laser": 100}
    t2 = copy.deepcopy(base)
    t2.game.max_steps = 33
    t2.game.agent.inventory.initial = {"ore_red": 3, "battery_red": 2, "laser": 1}
    t2.game.agent.inventory.default_limit = 7
    t2.game.agent.rewards.inventory = {"heart": 1, "laser": 0.25}
    for name in ("mine_red", "generator_red"):
        for p in t2.game.objects[name].protocols:
            p.cooldown = 3
    return [base, t1, t2]


@pytest.mark.parametrize("path_env", [{}, {"MG_NO_FUSED": "1"}, {"MG_FORCE_GENERAL": "1"}], ids=["fused", "fast", "general"])
def test_environments_change_configuration_between_episodes(path_env, monkeypatch):
    from metta_amd.batch import MettaGridBatch
    from oracle.mg_oracle import make_oracle_env

    for k, v in path_env.items():
        monkeypatch.setenv(k, v)
    tasks = _tasks()
    comps = [compile_game(t, init_obs_passes=2) for t in tasks]
    maps = load_map_pool()[:3]
    E, A, P = 3, 24, 3
    seeds = [31, 32, 33]
    b = MettaGridBatch(tasks[0], E, maps, device=0, config_slots=P)
    for s in range(1, P):
        b.write_config_slot(s, tasks[s])
    b.assign_map_configs([0, 1, 2], [0, 1, 2])  # map slot m plays task m
    b.reset(seeds=seeds, map_ids=[0, 1, 2])
    b.set_auto_reset(True, 1)
    cur = [0, 1, 2]
    oras = [make_oracle_env(comps[m], comps[m].encode_map(maps[m]), s) for s, m in zip(seeds, cur)]
    _compare_step(b, oras, 0, A)
    rs = np.random.RandomState(8)
    switches = 0
    for t in range(1, 150):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur[e] = (cur[e] + 1) % P
                m = cur[e]
                oras[e] = make_oracle_env(comps[m], comps[m].encode_map(maps[m]), seeds[e])
                switches += 1
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
        if t % 30 == 0:
            _compare_state(b, oras, t)
    assert switches >= 10
    # a task rewritten in place while no environment plays it: slot 0 <- task 2's configuration with another reward
    idle = [s for s in range(P) if s not in cur]
    if idle:
        s = idle[0]
        t3 = copy.deepcopy(tasks[2])
        t3.game.agent.rewards.inventory = {"heart": 2}
        t3.game.max_steps = 12
        b.write_config_slot(s, t3)
        comps[s] = compile_game(t3, init_obs_passes=2)
    for t in range(150, 260):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur[e] = (cur[e] + 1) % P
                m = cur[e]
                oras[e] = make_oracle_env(comps[m], comps[m].encode_map(maps[m]), seeds[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
    _compare_state(b, oras, 259)
    b.check_errors()
    b.close()


def test_task_outside_the_fast_class_moves_the_handle_to_the_general_kernels():
    from metta_amd.batch import MettaGridBatch
    from oracle.mg_oracle import make_oracle_env

    tasks = _tasks()
    slow = copy.deepcopy(tasks[0])
    slow.game.max_steps = 20
    slow.game.agent.inventory.initial = {"ore_red": 5}
    slow.game.actions.move.consumed_resources = {"ore_red": 1}  # a handler with costs: not the wave-parallel path
    slow.game.agent.inventory.limits["ore"] = ResourceLimitsConfig(limit=9, resources=["ore_red"], modifiers={"battery_red": 2})
    maps = load_map_pool()[:2]
    E, A = 2, 24
    b = MettaGridBatch(tasks[0], E, maps, device=0, config_slots=2)
    assert "fused" in b.step_path()
    b.write_config_slot(1, slow)
    assert "fused" not in b.step_path()
    b.assign_map_configs([0, 1], [0, 1])
    b.reset(seeds=[5, 6], map_ids=[0, 1])
    b.set_auto_reset(True, 1)
    comps = [compile_game(tasks[0], init_obs_passes=2), compile_game(slow, init_obs_passes=2)]
    cur = [0, 1]
    oras = [make_oracle_env(comps[m], comps[m].encode_map(maps[m]), s) for s, m in zip([5, 6], cur)]
    rs = np.random.RandomState(1)
    for t in range(1, 90):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                cur[e] = (cur[e] + 1) % 2
                oras[e] = make_oracle_env(comps[cur[e]], comps[cur[e]].encode_map(maps[cur[e]]), [5, 6][e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
    _compare_state(b, oras, 89)
    b.close()


def test_incompatible_tasks_are_refused():
    from metta_amd.batch import MettaGridBatch
    from metta_amd.builder import make_navigation

    base = _tasks()[0]
    maps = load_map_pool()[:2]
    b = MettaGridBatch(base, 1, maps, device=0, config_slots=2, config_headroom_words=8)
    with pytest.raises(RuntimeError, match="differs from the handle's in"):
        b.write_config_slot(1, make_navigation(24))
    with pytest.raises(RuntimeError, match="slot out of range"):
        b.write_config_slot(2, base)
    fat = copy.deepcopy(base)
    fat.game.agent.rewards.stats = {f"{r}.gained": 0.1 for r in ("ore_red", "battery_red", "heart", "laser", "armor", "blueprint")}
    with pytest.raises(RuntimeError, match="exceeds the slot"):
        b.write_config_slot(1, fat)
    b.close()
    plain = MettaGridBatch(base, 1, maps, device=0)
    with pytest.raises(RuntimeError, match="mg_set_config_slots first"):
        plain.write_config_slot(0, base)
    plain.close()


class _Task:
    def __init__(self, task_id, cfg):
        self._task_id, self._cfg, self.completions = task_id, cfg, []

    def get_env_cfg(self):
        return self._cfg

    def complete(self, score):
        self.completions.append(score)


class _Curriculum:
    """Round-robin over a fixed task list, recording what the vecenv reports (the reference's Curriculum protocol)."""

    def __init__(self, cfgs):
        self.tasks = [_Task(i, c) for i, c in enumerate(cfgs)]
        self.n, self.updates = 0, []

    def get_task(self):
        t = self.tasks[self.n % len(self.tasks)]
        self.n += 1
        return t

    def update_task_performance(self, task_id, score):
        self.updates.append((task_id, score))

    def stats(self):
        return {"num_completed": float(len(self.updates))}


def test_curriculum_vecenv_completes_tasks_with_their_episode_rewards():
    import torch

    from metta_amd.vecenv import CurriculumDeviceVecEnv
    from oracle.mg_oracle import make_oracle_env

    tasks = _tasks()
    cur = _Curriculum(tasks)
    E, A = 4, 24
    maps = load_map_pool()[:5]
    env = CurriculumDeviceVecEnv(cur, E, maps=maps, device=0, seed=70)
    env.stats_every = 1
    env.async_reset()
    # constructor drew task 0; reset drew 2E tasks round-robin: slot s plays task (1 + s) % 3 on map s % 5
    slot_task = [(1 + s) % 3 for s in range(2 * E)]
    comps = [compile_game(t, init_obs_passes=2) for t in tasks]
    slot_of = list(range(E))
    oras = [make_oracle_env(comps[slot_task[s]], comps[0].encode_map(maps[s % 5]), 70 + e) for e, s in enumerate(slot_of)]
    next_draw = 1 + 2 * E
    g = torch.Generator().manual_seed(4)
    infos, expected = [], []
    for t in range(1, 140):
        o, r, d, tr, ta, info, ids, mask = env.recv()
        infos += info
        obs = o.cpu().numpy().reshape(E, A, -1, 3)
        for e in range(E):
            assert np.array_equal(obs[e], oras[e].obs), f"step {t} env {e}"
        acts = torch.randint(0, 5, (E, A), generator=g, dtype=torch.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                s = slot_of[e]
                expected.append((slot_task[s], float(np.mean(oras[e].episode_rewards))))
                slot_task[s] = next_draw % 3  # the vecenv refills the slot that has just been left
                next_draw += 1
                s = slot_of[e] = (s + E) % (2 * E)
                oras[e] = make_oracle_env(comps[slot_task[s]], comps[0].encode_map(maps[s % 5]), 70 + e)
            oras[e].step(acts[e].numpy())
        env.send(acts.reshape(-1).cuda())
    assert len(expected) >= 12
    # with stats_every = 1 finished episodes are reported in step order, environments ascending: same as `expected`
    assert [u[0] for u in cur.updates] == [x[0] for x in expected[: len(cur.updates)]]
    assert np.allclose([u[1] for u in cur.updates], [x[1] for x in expected[: len(cur.updates)]], rtol=0, atol=1e-6)
    assert env.tasks_completed == len(cur.updates) >= len(expected) - E
    assert sum(len(t.completions) for t in cur.tasks) == len(cur.updates)
    assert any("env_curriculum/num_completed" in i for i in infos)
    ep = [i for i in infos if "attributes" in i]
    assert {i["attributes"]["max_steps"] for i in ep} == {40, 25, 33}
    env.close()


def test_curriculum_vecenv_on_a_device_generated_pool():
    """Tasks and maps both live in the pool: 2E device-drawn map slots, each with its own configuration slot; a refresh
    redraws idle maps and rebuilds their templates with the configuration the slot plays now."""
    import torch

    from metta_amd.vecenv import CurriculumDeviceVecEnv
    from oracle.mg_oracle import make_oracle_env

    tasks = _tasks()
    cur = _Curriculum(tasks)
    E, A = 3, 24
    env = CurriculumDeviceVecEnv(cur, E, device=0, seed=40, device_maps=True, map_seed=3, map_refresh_every=7)
    env.stats_every = 1
    env.async_reset()
    slot_task = [(1 + s) % 3 for s in range(2 * E)]
    comps = [compile_game(t, init_obs_passes=2) for t in tasks]
    slot_of = list(range(E))
    b = env.batch
    oras = [make_oracle_env(comps[slot_task[s]], b.get_map(s), 40 + e) for e, s in enumerate(slot_of)]
    next_draw = 1 + 2 * E
    g = torch.Generator().manual_seed(9)
    maps_seen = [set() for _ in range(E)]
    for t in range(1, 150):
        o, r, d, tr, ta, info, ids, mask = env.recv()
        obs = o.cpu().numpy().reshape(E, A, -1, 3)
        rew = r.cpu().numpy().reshape(E, A)
        for e in range(E):
            assert np.array_equal(obs[e], oras[e].obs), f"step {t} env {e}"
            assert np.array_equal(rew[e], oras[e].rewards if t > 1 else np.zeros(A, np.float32)), f"step {t} env {e}: rewards"
        acts = torch.randint(0, 5, (E, A), generator=g, dtype=torch.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                s = slot_of[e]
                slot_task[s] = next_draw % 3
                next_draw += 1
                s = slot_of[e] = (s + E) % (2 * E)
                m = b.get_map(s)
                maps_seen[e].add(m.tobytes())
                oras[e] = make_oracle_env(comps[slot_task[s]], m, 40 + e)
            oras[e].step(acts[e].numpy())
        env.send(acts.reshape(-1).cuda())
    assert len(cur.updates) >= 9 and all(len(s) >= 3 for s in maps_seen)
    b.check_errors()
    env.close()


def test_curriculum_vecenv_builds_each_task_its_own_map():
    """`task_maps=True`: a curriculum that buckets over map parameters (here the object counts of the arena builder).
    Every episode plays the map its task's builder made (`mg_write_map_slot` into the idle slot) under the task's
    configuration; bit-exact against the oracle built from that pair."""
    import torch

    from metta_amd.vecenv import CurriculumDeviceVecEnv
    from oracle.mg_oracle import make_oracle_env

    tasks = _tasks()
    mines = [tasks[0].game.map_builder.objects["mine_red"], 1, 3]
    assert mines[0] > 3
    for t, n in zip(tasks, mines):
        t.game.map_builder.objects = dict(t.game.map_builder.objects, mine_red=n)
    instances = tasks[0].game.map_builder.device_spec().rooms.__len__()
    cur = _Curriculum(tasks)
    E, A = 3, 24
    env = CurriculumDeviceVecEnv(cur, E, device=0, seed=11, map_seed=6, task_maps=True)
    env.stats_every = 1
    env.async_reset()
    slot_task = [(1 + s) % 3 for s in range(2 * E)]
    comps = [compile_game(t, init_obs_passes=2) for t in tasks]
    mine_code = comps[0].cell_code["mine_red"]
    slot_of = list(range(E))
    b = env.batch

    def start(e, s):
        m = np.array(b.map_pool[s])
        assert int((m == mine_code).sum()) == mines[slot_task[s]] * instances, "the slot holds its task's map"
        return make_oracle_env(comps[slot_task[s]], m, 11 + e), m.tobytes()

    oras, seen = [], [set() for _ in range(E)]
    for e, s in enumerate(slot_of):
        o, key = start(e, s)
        oras.append(o)
        seen[e].add(key)
    next_draw = 1 + 2 * E
    g = torch.Generator().manual_seed(2)
    for t in range(1, 130):
        o, r, d, tr, ta, info, ids, mask = env.recv()
        obs = o.cpu().numpy().reshape(E, A, -1, 3)
        for e in range(E):
            assert np.array_equal(obs[e], oras[e].obs), f"step {t} env {e}"
        acts = torch.randint(0, 5, (E, A), generator=g, dtype=torch.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                s = slot_of[e]
                slot_task[s] = next_draw % 3
                next_draw += 1
                s = slot_of[e] = (s + E) % (2 * E)
                oras[e], key = start(e, s)
                seen[e].add(key)
            oras[e].step(acts[e].numpy())
        env.send(acts.reshape(-1).cuda())
    assert len(cur.updates) >= 9 and all(len(s) >= 3 for s in seen), "every episode had a map of its own"
    b.check_errors()
    env.close()


def test_task_with_limit_modifiers_keeps_the_modifier_build_across_new_pools():
    """a task with inventory-limit modifiers written into a configuration slot must keep the handle off the
    modifier-free fused kernels, also when the pool is (re)built afterwards -- `mg_set_maps` used to recompute the flag
    from the handle's own configuration only.  Parity of the modifier task against the oracle after a second
    `set_maps`, and the step path names the build with modifiers."""
    from metta_amd.batch import MettaGridBatch
    from oracle.mg_oracle import make_oracle_env

    tasks = _tasks()
    mod = copy.deepcopy(tasks[0])
    mod.game.max_steps = 30
    mod.game.agent.inventory.initial = {"ore_red": 6, "battery_red": 1}
    mod.game.agent.inventory.limits["ore"] = ResourceLimitsConfig(limit=4, resources=["ore_red"], modifiers={"battery_red": 3})
    maps = load_map_pool()[:2]
    E, A = 2, 24
    b = MettaGridBatch(tasks[0], E, maps, device=0, config_slots=2)
    assert "nomod" in b.step_path()
    b.write_config_slot(1, mod)
    assert "fused" in b.step_path() and "nomod" not in b.step_path()
    b.set_maps(maps)  # a new pool on the live handle: the slot (and its task) survives it
    assert "fused" in b.step_path() and "nomod" not in b.step_path(), b.step_path()
    b.assign_map_configs([0, 1], [0, 1])
    b.reset(seeds=[11, 12], map_ids=[0, 1])
    b.set_auto_reset(True, 0)  # every environment stays on its map (and task)
    comps = [compile_game(tasks[0], init_obs_passes=2), compile_game(mod, init_obs_passes=2)]
    oras = [make_oracle_env(comps[m], comps[m].encode_map(maps[m]), s) for m, s in ((0, 11), (1, 12))]
    _compare_step(b, oras, 0, A)
    rs = np.random.RandomState(3)
    for t in range(1, 70):
        acts = rs.randint(0, 5, size=(E, A)).astype(np.int32)
        for e in range(E):
            if oras[e].terminals.all() or oras[e].truncations.all():
                oras[e] = make_oracle_env(comps[e], comps[e].encode_map(maps[e]), (11, 12)[e])
            oras[e].step(acts[e])
        b.step(acts.reshape(-1))
        _compare_step(b, oras, t, A)
    _compare_state(b, oras, 69)
    b.check_errors()
    b.close()


def test_lag0_curriculum_equals_the_reference_curriculum_env():
    """tests/golden/curriculum_env.json (oracle/gen_golden_curriculum.py): the reference's own `CurriculumEnv`
    (curriculum_env.py:10-142) over its `Simulator` + `StatsTracker`, three environments of a serial vecenv sharing one
    curriculum whose choices depend on the order of everything it is told.  `lag=0` must reproduce the curriculum's call
    log entry for entry (`get` / `complete` / `update`, task ids exact, rewards to 1e-6), play the same task in every
    episode of every environment, and end every episode on the same step with the same rewards and observations."""
    import json
    from pathlib import Path

    from metta_amd.vecenv import CurriculumDeviceVecEnv
    from oracle.gen_golden import digest
    from oracle.gen_golden_curriculum import StubCurriculum, apply_recipe

    fx = json.loads((Path(__file__).resolve().parent / "golden" / "curriculum_env.json").read_text())
    E, A, steps = fx["num_envs"], fx["num_agents"], fx["steps"]
    cfgs = [apply_recipe(make_arena(A), tuple(r)) for r in fx["recipes"]]
    grid = np.array(fx["cell_names"])[np.array(fx["map_codes"], np.uint8).reshape(fx["map_shape"])]
    cur = StubCurriculum(cfgs, fx["threshold"])
    env = CurriculumDeviceVecEnv(cur, E, maps=[grid], device=0, seed=fx["seed"], lag=0)
    env.async_reset(fx["seed"])
    assert cur.draws == fx["draws_before_stepping"]
    assert [env._slot_task[e]._task_id for e in range(E)] == fx["first_playing"]
    first = [next(ep for ep in fx["episodes"] if ep["env"] == e) for e in range(E)]
    assert any(ep["played"] != ep["task"] for ep in first), "the fixture holds the reference's first-episode quirk"
    actions = np.array(fx["actions"], np.int32).reshape(steps, E * A)
    want = {(ep["env"], ep["step"]): ep for ep in fx["episodes"]}
    infos, seen = list(env.recv()[5]), 0
    for t in range(steps):
        env.send(actions[t])
        o, r, d, tr, ta, info, ids, mask = env.recv()
        infos += info
        done = (d.view(E, A).all(dim=1) | tr.view(E, A).all(dim=1)).cpu().numpy()
        obs = o.cpu().numpy().reshape(E, A, -1, 3)
        for e in range(E):
            assert bool(done[e]) == ((e, t) in want), f"step {t} env {e}: episode end"
            if done[e]:
                assert digest(obs[e]) == want[(e, t)]["obs_digest"], f"step {t} env {e}: last observations"
                seen += 1
    assert seen == len(fx["episodes"]) >= 40
    assert len(cur.log) == len(fx["log"])
    for i, (a, b) in enumerate(zip(cur.log, fx["log"])):
        assert a[:2] == b[:2], f"call {i}: {a} vs {b}"
        if len(b) == 3:
            assert a[2] == pytest.approx(b[2], rel=0, abs=1e-6), f"call {i}: {a} vs {b}"
    eps = [i for i in infos if "attributes" in i]
    assert len(eps) == len(fx["episodes"])
    for got, ep in zip(eps, fx["episodes"]):  # reported in the reference's order: by step, environments ascending
        assert got["attributes"]["steps"] == ep["steps"] and got["attributes"]["seed"] == fx["seed"] + ep["env"]
        assert got["attributes"]["max_steps"] == fx["recipes"][ep["played"]][0]
        # (`env_curriculum/*` rides on an info every 50 steps -- of each wrapper there, of the vecenv here)
        assert set(got) >= {k for k in ep["info_keys"] if not k.startswith("env_curriculum/")}
        assert got["agent"].get("action.move.success", 0.0) == pytest.approx(ep["agent_move_success"], rel=1e-6)
        assert got["per_label_rewards"][cfgs[0].label] == pytest.approx(ep["mean_reward"], rel=0, abs=1e-6)
    env.close()
