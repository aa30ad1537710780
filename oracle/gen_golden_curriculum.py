"""Curriculum fixture from the reference's own `CurriculumEnv` (run in the build container only).

    make -C oracle ref && python oracle/gen_golden_curriculum.py

What is the reference's: `CurriculumEnv` (metta/cogworks/curriculum/curriculum_env.py:10-142: the order `complete` ->
`update_task_performance` -> `get_task` -> `set_mg_config` at an episode end, the draws in the constructor and in
`reset`), `Simulator` / `Simulation` / `StatsTracker` and the compiled core (`oracle/_ref`).  What is restated here, in a
few lines each, because pufferlib and gymnasium are absent from this image:

  * `_PufferSide`: the part of `MettaGridPufferEnv` the wrapper talks to (mettagrid_puffer_env.py:119-183:
    `set_mg_config`, `get_episode_rewards`, `reset(seed)`, `step(actions)` with the new simulation built by the step
    after the one that ended the episode), over the reference's `Simulator.new_simulation(cfg, seed, buffers)`;
  * the loop of pufferlib's `Serial` backend (vector.py:52-175): environments made in order by `make_env_func`
    (metta/rl/vecenv.py:23-49: one `get_task()` per environment for the constructor's configuration), `reset(seed)`
    gives environment i the seed `seed + i`, every step visits the environments in order;
  * `pufferlib.PufferEnv` as an empty base class (the wrapper never calls its constructor, curriculum_env.py:28-29).

The curriculum is a stub whose choices depend on everything it has been told so far (how many tasks were drawn, how many
were completed, whether the last reported reward was above a threshold), so that a wrapper which reports or draws in
another order produces another sequence of tasks.

A quirk the fixture pins: `reset` is not among the names the wrapper keeps for itself (curriculum_env.py:164-177), so a
vecenv's `reset` reaches the wrapped environment directly; no task is drawn there, the first episode of an environment
plays the configuration drawn by `make_env_func` and is reported to the task drawn by the wrapper's constructor
(`played` != `task` in the first episode of every environment).

Recorded: the task recipes, the map (the builder is seeded: every episode plays the same grid), the action stream, the
curriculum's call log (`get` / `complete` / `update` with task ids and rewards), and per finished episode (env, task id,
steps, mean reward, per-agent rewards, a digest of its last observations).
"""

from __future__ import annotations

import json
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle.gen_golden import GOLD, digest, encode_names, install_reference_shims  # noqa: E402

E, A, STEPS, SEED, THRESHOLD = 3, 6, 140, 500, 0.3

# (max_steps, reward per successful move, reward per ore_red held, initial ore_red)
RECIPES = [(7, 0.1, 0.0, 0), (10, 0.05, 0.1, 2), (13, 0.2, 0.0, 0), (9, 0.0, 0.25, 1), (11, 0.15, 0.05, 3)]


def apply_recipe(cfg, recipe):
    """The same few fields on the reference's MettaGridConfig and on metta_amd's mirror of it."""
    max_steps, w_move, w_ore, ore0 = recipe
    cfg.game.max_steps = max_steps
    cfg.game.agent.rewards.stats = {"action.move.success": w_move} if w_move else {}
    cfg.game.agent.rewards.inventory = {"heart": 1, "ore_red": w_ore} if w_ore else {"heart": 1}
    cfg.game.agent.inventory.initial = {"ore_red": ore0} if ore0 else {}
    return cfg


class StubTask:
    def __init__(self, cur, task_id, cfg):
        self._cur, self._task_id, self._cfg = cur, task_id, cfg

    def get_env_cfg(self):
        return self._cfg

    def complete(self, score):
        self._cur.log.append(["complete", self._task_id, float(score)])


class StubCurriculum:
    """Deterministic, and sensitive to the order of the wrapper's calls (see the module docstring)."""

    def __init__(self, cfgs, threshold=THRESHOLD):
        self.cfgs, self.threshold = cfgs, threshold
        self.draws = self.updates = self.above = 0
        self.log = []

    def get_task(self):
        k = (self.draws + 3 * self.updates + 2 * self.above) % len(self.cfgs)
        self.draws += 1
        self.log.append(["get", k])
        return StubTask(self, k, self.cfgs[k])

    def update_task_performance(self, task_id, score):
        self.updates += 1
        self.above += int(float(score) > self.threshold)
        self.log.append(["update", int(task_id), float(score)])

    def stats(self):
        return {"num_completed": float(self.updates)}


def main() -> None:
    install_reference_shims()
    puf = types.ModuleType("pufferlib")
    puf.PufferEnv = type("PufferEnv", (), {})
    sys.modules["pufferlib"] = puf
    sys.path.insert(0, "/root/reference")
    from metta.cogworks.curriculum.curriculum_env import CurriculumEnv
    from mettagrid.builder.envs import make_arena
    from mettagrid.envs.stats_tracker import StatsTracker
    from mettagrid.simulator.simulator import Buffers, Simulator
    from mettagrid.util.stats_writer import NoopStatsWriter

    def task_cfg(recipe):
        cfg = apply_recipe(make_arena(A), recipe)
        cfg.game.map_builder.seed = 4242
        return cfg

    cfgs = [task_cfg(r) for r in RECIPES]
    grid = cfgs[0].game.map_builder.create().build_for_num_agents(A).grid
    T = cfgs[0].game.obs.num_tokens

    class _PufferSide:
        def __init__(self, cfg, seed=0):
            self._simulator = Simulator()
            self._simulator.add_event_handler(StatsTracker(NoopStatsWriter()))
            self._cfg, self._seed = cfg, seed
            self._buffers = Buffers(
                observations=np.zeros((A, T, 3), np.uint8), terminals=np.zeros(A, bool), truncations=np.zeros(A, bool),
                rewards=np.zeros(A, np.float32), masks=np.ones(A, bool), actions=np.zeros(A, np.int32),
                teacher_actions=np.zeros(A, np.int32),
            )  # fmt: skip
            self._sim = None
            self._new_sim()

        def _new_sim(self):
            if self._sim is not None:
                self._sim.close()
            self._sim = self._simulator.new_simulation(self._cfg, self._seed, buffers=self._buffers)

        def set_mg_config(self, cfg):
            self._cfg = cfg

        def get_episode_rewards(self):
            return self._sim.episode_rewards

        def reset(self, seed=None):
            if seed is not None:
                self._seed = seed
            self._new_sim()
            return self._buffers.observations, {}

        def step(self, actions):
            if self._sim._c_sim.terminals().all() or self._sim._c_sim.truncations().all():
                self._new_sim()
            np.copyto(self._buffers.actions, np.asarray(actions, np.int32))
            self._sim.step()
            b = self._buffers
            return b.observations, b.rewards, b.terminals, b.truncations, self._sim._context.get("infos", {})

    cur = StubCurriculum(cfgs)
    envs = []
    for _ in range(E):
        first = cur.get_task().get_env_cfg()  # metta/rl/vecenv.py:45
        envs.append(CurriculumEnv(_PufferSide(first), cur))  # (draws another: curriculum_env.py:32)
    playing = [None] * E
    for i, env in enumerate(envs):
        # NOT a third draw: the wrapper's `__getattribute__` (curriculum_env.py:158-186) hands every name outside its
        # list to the wrapped environment, and "reset" is not on the list -- so the wrapper's own `reset` (:65-94) never
        # runs, the first episode plays the configuration of the FIRST draw (vecenv.py:45), and its result is reported
        # to the task of the SECOND (curriculum_env.py:32)
        env.reset(seed=SEED + i)
        playing[i] = env._current_task._task_id
    draws_before_stepping = cur.draws
    rs = np.random.RandomState(SEED)
    actions = rs.randint(0, 5, size=(STEPS, E, A)).astype(np.int32)
    episodes = []
    first_playing = list(playing)
    for t in range(STEPS):
        for i, env in enumerate(envs):
            obs, rew, term, trunc, infos = env.step(actions[t, i])
            if term.all() or trunc.all():
                side = env._env
                er = np.array(side.get_episode_rewards(), np.float32)
                episodes.append({
                    "env": i, "step": t, "task": playing[i], "steps": int(side._sim.current_step),
                    "played": next(k for k, c in enumerate(cfgs) if c is side._sim.config),
                    "mean_reward": float(er.mean()), "rewards": [float(x) for x in er],
                    "obs_digest": digest(obs), "info_keys": sorted(infos.keys()),
                    "agent_move_success": float(infos["agent"].get("action.move.success", 0.0)),
                })  # fmt: skip
                playing[i] = env._current_task._task_id  # what the wrapper has just drawn for the next episode
    margins = [abs(e["mean_reward"] - THRESHOLD) for e in episodes]
    assert min(margins) > 1e-3, "a reward sits on the stub's threshold: choose another threshold"
    names, codes = encode_names(grid)
    out = {
        "num_envs": E, "num_agents": A, "steps": STEPS, "seed": SEED, "threshold": THRESHOLD, "recipes": RECIPES,
        "cell_names": names, "map_shape": list(codes.shape), "map_codes": codes.reshape(-1).tolist(),
        "actions": actions.reshape(-1).tolist(), "first_playing": first_playing,
        "draws_before_stepping": draws_before_stepping, "log": cur.log, "episodes": episodes,
    }  # fmt: skip
    (GOLD / "curriculum_env.json").write_text(json.dumps(out))
    print("episodes", len(episodes), "draws", cur.draws, "above", cur.above, "min margin", min(margins))
    print("tasks played", sorted({e["task"] for e in episodes}), "size", (GOLD / "curriculum_env.json").stat().st_size)


if __name__ == "__main__":
    main()
