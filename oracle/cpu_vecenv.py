"""A multiprocess CPU vecenv around the reference core.   *** TEST INFRASTRUCTURE (bench.py's `cpu_baseline` leg) ***

The CPU baseline BASELINE.json names is "the reference's multiprocess CPU vecenv": pufferlib's `Multiprocessing`
backend (`packages/pufferlib-core/src/pufferlib/vector.py`).  The reference's Python does not travel to the GPU box,
so this file restates the mechanism in ~100 lines of our own, around the compiled reference core
(`oracle/ref_bridge.RefEnv`; the oracle port when `oracle/_ref` is absent):

* one worker process per core, each stepping `envs_per_worker` environments one after the other
  (`Serial.send`, vector.py:138-161);
* observations / rewards / terminals / truncations / actions of ALL workers in `multiprocessing.RawArray` shared
  memory, shaped `[workers, envs_per_worker * agents, ...]`; every environment adopts its slice as its buffers
  (zero copy: `MettaGrid.set_buffers`), so a step writes straight into shared memory (vector.py:203-214,341-370);
* one `uint8` semaphore per worker that both sides busy-poll: the driver writes the actions and sets STEP, the worker
  steps and sets MAIN (vector.py:222-251,415-511); a worker that has waited for 0.5 s starts sleeping 10 ms per poll,
  as the reference's does;
* the synchronous form (`batch_size == num_envs`, `workers_per_batch == num_workers`): `recv` returns when every
  worker is back.

What is left out: info dicts over pipes, episode restarts (no environment reaches `max_steps` inside the bounded
sample), `CurriculumEnv` and the Python wrapper layers around `MettaGrid.step` (`Simulation.step`, event handlers) --
all of which cost the reference more host time, so this is still on the fast side of the real thing.
"""

from __future__ import annotations

import multiprocessing as mp
import time
from multiprocessing import RawArray

import numpy as np

STEP, RESET, CLOSE, MAIN = 0, 1, 2, 3  # semaphore states (vector.py:27-31 uses the same four + INFO)


def _views(shm, workers: int, per_worker: int, T: int):
    shape = (workers, per_worker)
    return dict(
        observations=np.ndarray((*shape, T, 3), dtype=np.uint8, buffer=shm["observations"]),
        rewards=np.ndarray(shape, dtype=np.float32, buffer=shm["rewards"]),
        terminals=np.ndarray(shape, dtype=bool, buffer=shm["terminals"]),
        truncations=np.ndarray(shape, dtype=bool, buffer=shm["truncateds"]),
        actions=np.ndarray(shape, dtype=np.int32, buffer=shm["actions"]),
        semaphores=np.ndarray(workers, dtype=np.uint8, buffer=shm["semaphores"]),
    )


def _worker(kind, make_cfg, pool, worker, workers, n_envs, A, T, seed0, shm):
    buf = _views(shm, workers, n_envs * A, T)
    sem = buf["semaphores"]
    cfg = make_cfg()
    envs = []
    if kind == "reference":
        from oracle.ref_bridge import RefEnv

        for i in range(n_envs):
            g = worker * n_envs + i
            sl = slice(i * A, (i + 1) * A)
            e = RefEnv(cfg.game, pool[g % len(pool)], seed0 + g, with_buffers=False)
            e.obs, e.terminals, e.truncations = buf["observations"][worker, sl], buf["terminals"][worker, sl], buf["truncations"][worker, sl]
            e.rewards, e.actions = buf["rewards"][worker, sl], buf["actions"][worker, sl]
            e.env.set_buffers(e.obs, e.terminals, e.truncations, e.rewards, e.actions)  # zero copy into shared memory
            envs.append(e)

        def step_all():
            for e in envs:
                e.env.step()  # reads its slice of the shared action array, writes its slices of the others
    else:
        from metta_amd.compile import compile_game
        from oracle.mg_oracle import make_oracle_env

        comp = compile_game(cfg, init_obs_passes=2)
        for i in range(n_envs):
            g = worker * n_envs + i
            envs.append(make_oracle_env(comp, comp.encode_map(pool[g % len(pool)]), seed0 + g))

        def step_all():  # (the port owns its buffers: one copy per environment into shared memory)
            for i, e in enumerate(envs):
                sl = slice(i * A, (i + 1) * A)
                e.step(buf["actions"][worker, sl])
                buf["observations"][worker, sl] = e.obs.reshape(A, T, 3)
                buf["rewards"][worker, sl] = e.rewards

    sem[worker] = MAIN  # ready
    idle_since = time.time()
    while True:
        s = sem[worker]
        if s >= MAIN:
            if time.time() - idle_since > 0.5:
                time.sleep(0.01)
            continue
        if s == CLOSE:
            break
        step_all()
        sem[worker] = MAIN
        idle_since = time.time()


class CpuVecEnv:
    def __init__(self, kind: str, make_cfg, pool, workers: int, envs_per_worker: int, agents: int, tokens: int, seed: int = 0):
        self.workers, self.per_worker, self.A, self.T = workers, envs_per_worker * agents, agents, tokens
        n = workers * self.per_worker
        self.shm = dict(
            observations=RawArray("B", n * tokens * 3),
            actions=RawArray("i", n),
            rewards=RawArray("f", n),
            terminals=RawArray("b", n),
            truncateds=RawArray("b", n),
            semaphores=RawArray("B", workers),
        )
        self.buf = _views(self.shm, workers, self.per_worker, tokens)
        ctx = mp.get_context("fork")
        self.procs = []
        for w in range(workers):
            self.buf["semaphores"][w] = MAIN + 1  # "building": the worker lowers it to MAIN when its environments exist
            p = ctx.Process(target=_worker, args=(kind, make_cfg, pool, w, workers, envs_per_worker, agents, tokens, seed, self.shm), daemon=True)
            p.start()
            self.procs.append(p)
        t0 = time.time()
        while (self.buf["semaphores"] != MAIN).any():
            if time.time() - t0 > 120 or any(not p.is_alive() for p in self.procs):
                self.close()
                raise RuntimeError("cpu vecenv: a worker failed to start")
            time.sleep(0.005)

    def send(self, actions: np.ndarray) -> None:
        self.buf["actions"][:] = actions.reshape(self.workers, self.per_worker)
        self.buf["semaphores"][:] = STEP

    def recv(self):
        sem = self.buf["semaphores"]
        while True:  # busy-poll, like the driver side of vector.py:415-455
            if (sem >= MAIN).all():
                break
        b = self.buf
        return b["observations"], b["rewards"], b["terminals"], b["truncations"]

    def close(self) -> None:
        self.buf["semaphores"][:] = CLOSE
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
