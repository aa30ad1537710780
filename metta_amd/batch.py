"""`MettaGridBatch` -- E environments of one game configuration resident on one MI355X.

Host-side counterpart of the reference's `MettaGrid` C++ object (`BIND/mettagrid_c.cpp:998-1093`), batched:
the same accessors (`observations()/rewards()/terminals()/truncations()/actions()/masks()`,
`get_episode_rewards()`, `get_episode_stats()`, `action_success()`, `set_inventory()`, `current_step`,
`map_width/map_height`), but every buffer is a torch ROCm tensor aliasing the library's HBM buffers
(zero-copy, `__cuda_array_interface__`), shaped `[E*A, ...]` like pufferlib's shared buffers
(`PUF/pufferlib.py:18-45`).
"""

from __future__ import annotations

import ctypes as ct
from typing import Optional, Sequence

import numpy as np

from metta_amd import native
from metta_amd.compile import CompiledGame, compile_game
from metta_amd.stats_decode import decode_episode_stats
from metta_amd.umap import order_from_pydict


class _HandleOwner:
    """Owns the native handle.  Tensors aliasing the library's HBM hold a reference to this object -- not to the
    `MettaGridBatch` -- so there is no reference cycle through torch's storage deleter, and the handle is destroyed
    when the batch AND every tensor it handed out are gone (or at once by `MettaGridBatch.close()`)."""

    def __init__(self, lib, handle):
        self.lib, self.h = lib, handle

    def destroy(self) -> None:
        if self.h is not None and self.h:
            self.lib.mg_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class _DevArray:
    """Minimal `__cuda_array_interface__` carrier so torch can alias a raw device pointer."""

    def __init__(self, ptr: int, shape: tuple, typestr: str, owner):
        self.__cuda_array_interface__ = {
            "shape": shape,
            "typestr": typestr,
            "data": (int(ptr), False),
            "version": 3,
            "strides": None,
        }
        self._owner = owner  # the _HandleOwner: keeps the HBM alive while tensors exist


_ALIASED = ("observations", "rewards", "terminals", "truncations", "actions", "masks", "episode_rewards", "action_success")


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(ct.POINTER(t))


class _DevicePool:
    """`map_pool` of a device-generated pool: maps are read back from HBM on demand (they change as slots refresh)."""

    def __init__(self, batch, num_maps: int):
        self._batch, self._n = batch, int(num_maps)

    def __len__(self) -> int:
        return self._n

    def __getitem__(self, i: int) -> np.ndarray:
        return self._batch.get_map(int(i) % self._n)


class MettaGridBatch:
    def __init__(
        self,
        cfg,
        num_envs: int,
        maps: Sequence,
        device: int = 0,
        init_obs_passes: int = 2,
        seeds: Optional[Sequence[int]] = None,
        map_ids: Optional[Sequence[int]] = None,
        compiled: Optional[CompiledGame] = None,
        config_slots: int = 0,
        config_headroom_words: int = -1,
    ):
        import torch

        self._torch = torch
        self.cfg = cfg
        self.compiled = compiled or compile_game(cfg, init_obs_passes=init_obs_passes)
        # inventories of up to 16 resources: libmettagrid_hip.so; 17..32: the wide-inventory build of the same sources
        self.L = native.load_for(len(self.compiled.resource_names))
        self.num_envs = int(num_envs)
        self.num_agents = self.compiled.num_agents
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device_index)
        blob = np.ascontiguousarray(self.compiled.blob, dtype=np.int32)
        self._h = ct.c_void_p()
        self._check(self.L.mg_create(_p(blob, ct.c_int32), blob.size, self.num_envs, self.device_index, ct.byref(self._h)))
        self._owner = _HandleOwner(self.L, self._h)
        self.config_slots = int(config_slots)
        self._init_obs_passes = int(init_obs_passes)  # (tasks are compiled like the handle's own configuration)
        if self.config_slots > 0:  # per-slot configurations (curriculum tasks): before the pool, which is sized for them
            self._check(self.L.mg_set_config_slots(self._h, self.config_slots, int(config_headroom_words)))
        self.set_maps(maps)
        b = native.MgBuffers()
        self._check(self.L.mg_get_buffers(self._h, ct.byref(b)))
        N, T = self.num_envs * self.num_agents, self.compiled.num_tokens
        self.num_actions = int(b.num_actions)

        def alias(ptr, shape, typestr, dtype):
            t = torch.as_tensor(_DevArray(ptr, shape, typestr, self._owner), device=self.device)
            assert t.dtype == dtype and t.data_ptr() == ptr
            return t

        self.observations = alias(b.observations, (N, T, 3), "|u1", torch.uint8)
        self.rewards = alias(b.rewards, (N,), "<f4", torch.float32)
        self.terminals = alias(b.terminals, (N,), "|b1", torch.bool)
        self.truncations = alias(b.truncations, (N,), "|b1", torch.bool)
        self.actions = alias(b.actions, (N,), "<i4", torch.int32)
        self.masks = alias(b.masks, (N,), "|b1", torch.bool)
        self.episode_rewards = alias(b.episode_rewards, (N,), "<f4", torch.float32)
        self.action_success = alias(b.action_success, (N,), "|b1", torch.bool)
        if seeds is not None or map_ids is not None:
            self.reset(seeds=seeds, map_ids=map_ids)

    def _check(self, rc: int) -> None:
        native.check(rc, self.L)

    # ---- maps / reset / step ---------------------------------------------------------------------------
    def set_maps(self, maps: Sequence) -> None:
        """`maps`: sequence of 2-D arrays, either uint8 cell codes or cell-name grids (reference `GameMap.grid`), or a
        `metta_amd.mapgen.GeneratedMaps` (the pool is then drawn on the device, see `set_generated_maps`)."""
        from .mapgen import GeneratedMaps

        if isinstance(maps, GeneratedMaps):
            return self.set_generated_maps(maps.spec, maps.num_maps, maps.seed)
        self.generated = False
        enc = []
        for m in maps:
            arr = np.asarray(m)
            enc.append(arr.astype(np.uint8) if arr.dtype.kind in "ui" else self.compiled.encode_map(arr))
        enc = [np.ascontiguousarray(e, dtype=np.uint8) for e in enc]
        if any(e.ndim != 2 for e in enc):
            raise RuntimeError("maps must be 2-D")
        self.map_pool = enc
        self.map_max_steps = None
        self.map_dims = np.array([e.shape for e in enc], dtype=np.int32)  # [P][2] = (height, width)
        # the reference's read-only map_height / map_width; for a mixed pool: the largest (see map_dims, env_map_dims)
        self.map_height, self.map_width = int(self.map_dims[:, 0].max()), int(self.map_dims[:, 1].max())
        if len({e.shape for e in enc}) == 1:
            pool = np.ascontiguousarray(np.stack(enc), dtype=np.uint8)
            self._check(self.L.mg_set_maps(self._h, _p(pool, ct.c_uint8), pool.shape[0], pool.shape[1], pool.shape[2]))
        else:  # maps of different sizes (navigation suite): back to back, with their dimensions
            flat = np.ascontiguousarray(np.concatenate([e.reshape(-1) for e in enc]), dtype=np.uint8)
            hs = np.ascontiguousarray(self.map_dims[:, 0])
            ws = np.ascontiguousarray(self.map_dims[:, 1])
            self._check(
                self.L.mg_set_maps_ragged(self._h, _p(flat, ct.c_uint8), len(enc), _p(hs, ct.c_int32), _p(ws, ct.c_int32))
            )

    def set_generated_maps(self, spec, num_maps: int, seed: int = 0) -> None:
        """A pool of `num_maps` slots drawn on the device from `spec` (`metta_amd.mapgen.DeviceMapSpec`): the
        reference's `RandomMapBuilder` / MapGen `Random` scene distribution (random_map.py:29-96, scenes/random.py:19-69)
        without a host copy.  `refresh_maps()` redraws used, idle slots."""
        e = spec.encode(self.compiled)
        self._check(
            self.L.mg_set_generated_maps(
                self._h, e["height"], e["width"], e["fill_code"], len(e["rooms"]), _p(e["rooms"], ct.c_int32),
                _p(e["room_sym_begin"], ct.c_int32), _p(e["sym_code"], ct.c_int32), _p(e["sym_count"], ct.c_int32),
                int(num_maps), int(seed) & 0xFFFFFFFFFFFFFFFF,
            )
        )  # fmt: skip
        self.generated = True
        self.map_max_steps = None
        self.map_dims = np.tile(np.array([[e["height"], e["width"]]], dtype=np.int32), (int(num_maps), 1))
        self.map_height, self.map_width = e["height"], e["width"]
        self.map_pool = _DevicePool(self, num_maps)

    def refresh_maps(self, force: bool = False) -> None:
        """Redraw, on the launch stream and without a host sync, every slot that an episode has used since it was
        drawn and that no environment is on now (`force`: every slot; the environments then need a `reset`).  With
        2E slots and `set_auto_reset(True, map_stride=E)` each environment alternates between two slots, and a refresh
        once per episode gives every episode a new map, like the reference's per-episode `map_builder.build()`
        (simulator.py:264-267)."""
        self._check(self.L.mg_refresh_maps(self._h, int(bool(force)), self._stream()))

    def get_map(self, map_id: int) -> np.ndarray:
        """uint8 cell codes of slot `map_id` of a device-generated pool (host copy; synchronises the stream)."""
        out = np.zeros((self.map_height, self.map_width), np.uint8)
        self._check(self.L.mg_get_map(self._h, int(map_id), _p(out, ct.c_uint8), self._stream()))
        return out

    def write_map_slot(self, slot: int, grid) -> None:
        """Replace map slot `slot` by `grid` (cell names or codes, the pool's size): the reference's per-episode
        `map_builder.build()` (simulator.py:264-267) for maps built on the host.  Write the slot an environment enters
        next, not one an episode is on."""
        arr = np.asarray(grid)
        enc = arr.astype(np.uint8) if arr.dtype.kind in "ui" else self.compiled.encode_map(arr)
        enc = np.ascontiguousarray(enc, dtype=np.uint8)
        if enc.shape != (self.map_height, self.map_width):
            raise RuntimeError(f"the map is {enc.shape}, the pool's slots are {(self.map_height, self.map_width)}")
        self._check(self.L.mg_write_map_slot(self._h, int(slot), _p(enc, ct.c_uint8), self._stream()))
        if not getattr(self, "generated", False):
            self.map_pool[int(slot)] = enc

    def write_config_slot(self, slot: int, cfg) -> None:
        """Compile `cfg` (a MettaGridConfig, or a CompiledGame) into configuration slot `slot`: the device side of
        `CurriculumEnv`'s `env.set_mg_config(task.get_env_cfg())` (curriculum_env.py:111-126).  Raises RuntimeError when
        the task differs from this handle's configuration in what sizes buffers or observations."""
        comp = cfg if isinstance(cfg, CompiledGame) else compile_game(cfg, init_obs_passes=self._init_obs_passes)
        if comp.agent_slot_names != self.compiled.agent_slot_names or comp.game_slot_names != self.compiled.game_slot_names:
            raise RuntimeError("mg_write_config_slot: the task's configuration differs from the handle's in the statistics layout")
        blob = np.ascontiguousarray(comp.blob, dtype=np.int32)
        self._check(self.L.mg_write_config_slot(self._h, int(slot), _p(blob, ct.c_int32), blob.size, self._stream()))

    def register_config(self, cfg) -> int:
        """Compile `cfg` (a MettaGridConfig, or a CompiledGame) once into the handle's device-side registry
        (`mg_register_config`); the id goes to `write_config_slots`.  Same checks as `write_config_slot`."""
        comp = cfg if isinstance(cfg, CompiledGame) else compile_game(cfg, init_obs_passes=self._init_obs_passes)
        if comp.agent_slot_names != self.compiled.agent_slot_names or comp.game_slot_names != self.compiled.game_slot_names:
            raise RuntimeError("mg_register_config: the task's configuration differs from the handle's in the statistics layout")
        blob = np.ascontiguousarray(comp.blob, dtype=np.int32)
        cid = ct.c_int32()
        self._check(self.L.mg_register_config(self._h, _p(blob, ct.c_int32), blob.size, ct.byref(cid)))
        return int(cid.value)

    def write_config_slots(self, slots: Sequence[int], config_ids: Sequence[int]) -> None:
        """Registered configurations into many slots at once: one copy kernel + one template rebuild, stream-ordered."""
        sl = np.ascontiguousarray(slots, dtype=np.int32)
        ids = np.ascontiguousarray(config_ids, dtype=np.int32)
        if sl.shape != ids.shape or sl.ndim != 1:
            raise RuntimeError("slots and config_ids must be 1-D and of equal length")
        if len(sl):
            self._check(self.L.mg_write_config_slots(self._h, len(sl), _p(sl, ct.c_int32), _p(ids, ct.c_int32), self._stream()))

    def assign_map_configs(self, map_ids: Sequence[int], cfg_slots: Sequence[int]) -> None:
        """Map slot map_ids[i] plays configuration slot cfg_slots[i] from its next episode start on."""
        m = np.ascontiguousarray(map_ids, dtype=np.int32)
        c = np.ascontiguousarray(cfg_slots, dtype=np.int32)
        if m.shape != c.shape or m.ndim != 1:
            raise RuntimeError("map_ids and cfg_slots must be 1-D and of equal length")
        self._check(self.L.mg_assign_map_configs(self._h, len(m), _p(m, ct.c_int32), _p(c, ct.c_int32), self._stream()))

    def set_map_max_steps(self, max_steps: Optional[Sequence[int]]) -> None:
        """Episode length per map of the pool (0 = the configuration's `max_steps`); None clears it.  The reference's
        evaluation suites give each map its own `game.max_steps` (recipes/experiment/navigation.py:32-59)."""
        if max_steps is None:
            self._check(self.L.mg_set_map_max_steps(self._h, None))
            self.map_max_steps = None
            return
        ms = np.ascontiguousarray(max_steps, dtype=np.int32)
        if ms.shape != (len(self.map_pool),):
            raise RuntimeError("max_steps must have one entry per map of the pool")
        self._check(self.L.mg_set_map_max_steps(self._h, _p(ms, ct.c_int32)))
        self.map_max_steps = ms

    def env_map_dims(self, map_id: int) -> tuple[int, int]:
        """(height, width) of map `map_id` of the pool (reference: `MettaGrid.map_height / map_width`)."""
        h, w = ct.c_int32(), ct.c_int32()
        self._check(self.L.mg_map_dims(self._h, int(map_id), ct.byref(h), ct.byref(w)))
        return int(h.value), int(w.value)

    def _stream(self) -> ct.c_void_p:
        return ct.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, seeds=None, map_ids=None, env_ids=None) -> None:
        """Restart environments (all by default).  Env i defaults to seed i and map i mod P."""
        n = self.num_envs if env_ids is None else len(env_ids)
        ids = None if env_ids is None else np.ascontiguousarray(env_ids, dtype=np.int32)
        base = np.arange(self.num_envs) if env_ids is None else np.asarray(env_ids)
        s = np.ascontiguousarray(base if seeds is None else seeds, dtype=np.uint32)
        m = np.ascontiguousarray(base % len(self.map_pool) if map_ids is None else map_ids, dtype=np.int32)
        if s.shape != (n,) or m.shape != (n,):
            raise RuntimeError("seeds / map_ids must have one entry per environment being reset")
        self._check(
            self.L.mg_reset(self._h, None if ids is None else _p(ids, ct.c_int32), n, _p(s, ct.c_uint32), _p(m, ct.c_int32), self._stream())
        )

    def step(self, actions=None) -> None:
        """One tick of every environment.  `actions`: None (use the action buffer as is), a contiguous int32 ROCm
        tensor (read in place, zero-copy: `mg_step_actions`), or anything else (copied into the action buffer)."""
        torch = self._torch
        if actions is not None:
            if (
                isinstance(actions, torch.Tensor)
                and actions.is_cuda
                and actions.dtype == torch.int32
                and actions.is_contiguous()
                and actions.device == self.device
            ):
                if actions.numel() != self.actions.numel():
                    raise RuntimeError("actions has the wrong shape")
                self._check(self.L.mg_step_actions(self._h, ct.c_void_p(actions.data_ptr()), self._stream()))
                return
            t = torch.as_tensor(actions)
            if t.numel() != self.actions.numel():
                raise RuntimeError("actions has the wrong shape")
            # np.copyto(buffers.actions, actions, casting="safe") in MettaGridPufferEnv.step (mettagrid_puffer_env.py:
            # 166-169) raises TypeError for anything that does not cast safely to int32
            if t.dtype not in (torch.int32, torch.int16, torch.int8, torch.uint8):
                raise TypeError(f"Cannot cast actions from {t.dtype} to int32 according to the rule 'safe'")
            self.actions.copy_(t.reshape(-1).to(dtype=torch.int32), non_blocking=True)
        self._check(self.L.mg_step(self._h, self._stream()))

    def step_raw(self, actions_ptr: int, stream: int) -> None:
        """`mg_step_actions` with a device pointer to E*A int32 actions and a HIP stream handle, both as integers: for
        launch loops that have validated their tensors once (several handles stepped round-robin on their own streams
        leave about ten microseconds of host time per launch)."""
        if self.L.mg_step_actions(self._h, actions_ptr, stream):
            self._check(1)

    @staticmethod
    def group_stepper(batches, action_rows, streams):
        """`mg_step_groups` for a fixed set of handles: returns `step(row)` that steps batches[q] with the device
        actions at action_rows[q][row] (integer pointers) on streams[q] (HIP stream handles), all in one library call."""
        L, n = batches[0].L, len(batches)
        hs = (ct.c_void_p * n)(*[ct.cast(b._h, ct.c_void_p).value for b in batches])
        sts = (ct.c_void_p * n)(*[int(s) for s in streams])
        rows = [(ct.c_void_p * n)(*[int(action_rows[q][r]) for q in range(n)]) for r in range(len(action_rows[0]))]
        check = batches[0]._check

        def step(row: int) -> None:
            if L.mg_step_groups(n, hs, rows[row], sts):
                check(1)

        return step

    def step_path(self) -> str:
        """Names of the kernels one step launches for this configuration (diagnostic)."""
        return self.L.mg_step_path(self._h).decode()

    def set_auto_reset(self, enabled: bool, map_stride: int = 1) -> None:
        self._check(self.L.mg_set_auto_reset(self._h, int(enabled), int(map_stride)))
        self.map_stride = int(map_stride)

    def set_early_reset(self, steps: Optional[Sequence[int]]) -> None:
        if steps is None:
            self._check(self.L.mg_set_early_reset(self._h, None))
        else:
            s = np.ascontiguousarray(steps, dtype=np.uint32)
            assert s.shape == (self.num_envs,)
            self._check(self.L.mg_set_early_reset(self._h, _p(s, ct.c_uint32)))

    def end_episode(self, env: int = -1) -> None:
        self._check(self.L.mg_end_episode(self._h, int(env), self._stream()))

    def set_profiling(self, enabled, stride: int = 1) -> None:
        """False/0 off, True/1 time whole steps, 2 time the observation kernel only (HIP events on the launch stream).
        `stride`: record every stride-th step only (the event pair costs about 7 % of a 90 us step)."""
        self._check(self.L.mg_set_profiling_stride(self._h, int(stride)))
        self._check(self.L.mg_set_profiling(self._h, int(enabled)))

    def profile_read(self, max_steps: int = 4096) -> np.ndarray:
        """Per-step kernel durations (ms) recorded since profiling was enabled / last read."""
        ms = np.zeros(max_steps, np.float32)
        n = ct.c_int()
        self._check(self.L.mg_profile_read(self._h, _p(ms, ct.c_float), max_steps, ct.byref(n)))
        return ms[: n.value].copy()

    # ---- inspection ------------------------------------------------------------------------------------------
    def env_info(self, env: int) -> dict:
        step, mid, seed, eps = ct.c_uint32(), ct.c_int32(), ct.c_uint32(), ct.c_uint32()
        self._check(
            self.L.mg_env_info(self._h, env, ct.byref(step), ct.byref(mid), ct.byref(seed), ct.byref(eps), self._stream())
        )
        return {"current_step": step.value, "map_id": mid.value, "seed": seed.value, "episodes": eps.value}

    def env_table(self) -> np.ndarray:
        """u32 [E, 8]: current_step, rng index, seed, map_id, episodes, early_reset, done, error."""
        out = np.zeros((self.num_envs, 8), np.uint32)
        self._check(self.L.mg_env_table(self._h, _p(out, ct.c_uint32), self._stream()))
        return out

    def finished_rewards(self, env: int):
        r = np.zeros(self.num_agents, np.float32)
        steps = ct.c_uint32()
        self._check(self.L.mg_finished_rewards(self._h, env, _p(r, ct.c_float), ct.byref(steps), self._stream()))
        return r, int(steps.value)

    def finished_ring_capacity(self) -> int:
        return int(self.L.mg_finished_ring_capacity(self._h))

    def finished_ring(self, from_count: int, to_count: int, stream=None):
        """(env ids, map slots) of the episodes that finished between two readings of the finished-episode counter, in the
        order they ended (`mg_finished_ring`; no stream synchronisation)."""
        n = (int(to_count) - int(from_count)) & 0xFFFFFFFF
        envs, maps = np.zeros(n, np.int32), np.zeros(n, np.int32)
        if n:
            self._check(self.L.mg_finished_ring(self._h, int(from_count) & 0xFFFFFFFF, int(to_count) & 0xFFFFFFFF,
                                                _p(envs, ct.c_int32), _p(maps, ct.c_int32),
                                                self._stream() if stream is None else ct.c_void_p(int(stream))))  # fmt: skip
        return envs, maps

    def finished_batch(self, env_ids: Sequence[int], stream=None):
        """The last finished episode of several environments in one device round trip: (agent_values [n, A, SA],
        agent_touched [n, A, SA] bool, game_values [n, SG], game_touched [n, SG] bool, episode_rewards [n, A], steps [n])."""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        n, A = len(ids), self.num_agents
        Sa, Sg = len(self.compiled.agent_slot_names[0]), len(self.compiled.game_slot_names)
        av, at = np.zeros((n, A, Sa), np.float32), np.zeros((n, A, Sa), np.uint8)
        gv, gt = np.zeros((n, Sg), np.float32), np.zeros((n, Sg), np.uint8)
        rew, steps = np.zeros((n, A), np.float32), np.zeros(n, np.uint32)
        if n:
            self._check(
                self.L.mg_finished_batch(
                    self._h, n, _p(ids, ct.c_int32), _p(av, ct.c_float), _p(at, ct.c_uint8), _p(gv, ct.c_float),
                    _p(gt, ct.c_uint8), _p(rew, ct.c_float), _p(steps, ct.c_uint32),
                    self._stream() if stream is None else ct.c_void_p(int(stream)),
                )
            )  # fmt: skip
        return av, at.astype(bool), gv, gt.astype(bool), rew, steps

    def map_agent_groups(self, map_id: int) -> list[int]:
        """Group index of every agent of map slot `map_id`, in agent-id order (the row-major order of the agent cells)."""
        flat = self.map_pool[map_id].reshape(-1)
        c = self.compiled
        return [c.cell_group[int(v) - 1] for v in flat if v != 0 and c.cell_kinds[int(v) - 1] == 2]

    def current_step(self, env: int) -> int:
        return self.env_info(env)["current_step"]

    def raw_stats(self, env: int, last_finished: bool = False):
        A, Sa, Sg = self.num_agents, len(self.compiled.agent_slot_names[0]), len(self.compiled.game_slot_names)
        av, at = np.zeros((A, Sa), np.float32), np.zeros((A, Sa), np.uint8)
        gv, gt = np.zeros(Sg, np.float32), np.zeros(Sg, np.uint8)
        self._check(
            self.L.mg_episode_stats(
                self._h, env, int(last_finished), _p(av, ct.c_float), _p(at, ct.c_uint8), _p(gv, ct.c_float),
                _p(gt, ct.c_uint8), self._stream(),
            )
        )  # fmt: skip
        return av, at, gv, gt

    def agent_groups(self, env: int, last_finished: bool = False) -> list[int]:
        mid = self.env_info(env)["map_id"]
        if last_finished:  # the finished episode played the previous map of the rotation
            mid = (mid - getattr(self, "map_stride", 1)) % len(self.map_pool)
        flat = self.map_pool[mid].reshape(-1)
        c = self.compiled
        return [c.cell_group[int(v) - 1] for v in flat if v != 0 and c.cell_kinds[int(v) - 1] == 2]

    def get_episode_stats(self, env: int, last_finished: bool = False) -> dict:
        """`MettaGrid::get_episode_stats` (BIND/mettagrid_c.cpp:936-953) of one environment."""
        av, at, gv, gt = self.raw_stats(env, last_finished)
        return decode_episode_stats(self.compiled, av, at, gv, gt, self.agent_groups(env, last_finished))

    def agent_state(self, env: int) -> np.ndarray:
        R = len(self.compiled.resource_names)
        out = np.zeros((self.num_agents, 9 + 2 * R), np.int32)
        self._check(self.L.mg_agent_state(self._h, env, _p(out, ct.c_int32), self._stream()))
        return out

    def assembler_state(self, env: int) -> np.ndarray:
        n = self.L.mg_num_assemblers(self._h, env)
        if n < 0:
            raise RuntimeError("mg_num_assemblers failed")
        out = np.zeros((max(n, 1), 8), np.int32)
        if n:
            self._check(self.L.mg_assembler_state(self._h, env, _p(out, ct.c_int32), self._stream()))
        return out[:n]

    def clipper_state(self, env: int):
        """(length scale in use, infection weight of every assembler of `env`): the Clipper's `length_scale` and
        `assembler_infection_weight` (clipper.hpp:19-30)."""
        n = self.L.mg_num_assemblers(self._h, env)
        if n < 0:
            raise RuntimeError("mg_num_assemblers failed")
        w = np.zeros(max(n, 1), np.uint32)
        ls = ct.c_int32()
        self._check(self.L.mg_clipper_state(self._h, env, _p(w, ct.c_uint32), ct.byref(ls), self._stream()))
        return int(ls.value), w[:n]

    def chest_state(self, env: int) -> np.ndarray:
        """[n][5 + 2R]: r, c, chest config index, object id, n_items, (item, amount) in inventory order."""
        n = ct.c_int32()
        self._check(self.L.mg_chest_state(self._h, env, None, 0, ct.byref(n), self._stream()))
        R = len(self.compiled.resource_names)
        out = np.zeros((max(n.value, 1), 5 + 2 * R), np.int32)
        if n.value:
            self._check(self.L.mg_chest_state(self._h, env, _p(out, ct.c_int32), n.value, ct.byref(n), self._stream()))
        return out[: n.value]

    def grid_objects(self, env: int, min_row: int = -1, max_row: int = -1, min_col: int = -1, max_col: int = -1,
                     ignore_types: Sequence[str] = ()) -> dict:
        """`MettaGrid::grid_objects` (BIND/mettagrid_c.cpp:754-918) of one environment: {object id: dict}.

        Holds the identity, position, inventory and the agent / assembler / chest state fields of the reference's
        dicts, and an assembler's `protocols` listing in the reference's order.  Not reproduced: the injected
        per-feature observation values (`tag`, `protocol_input:*`, ...) and `current_protocol_*` (they restate the
        configuration and the observation tokens)."""
        c = self.compiled
        game = self.cfg.game if hasattr(self.cfg, "game") else None
        mid = self.env_info(env)["map_id"]
        grid = self.map_pool[mid]
        ast = self.agent_state(env)
        asm = self.assembler_state(env)
        chs = self.chest_state(env)
        rew = np.zeros(self.num_agents, np.float32)
        self._check(self.L.mg_agent_stat_reward(self._h, env, _p(rew, ct.c_float), self._stream()))
        step = self.current_step(env)
        groups = self.agent_groups(env)
        use_bounds = min_row >= 0 and max_row >= 0 and min_col >= 0 and max_col >= 0
        out: dict[int, dict] = {}
        oid = 0
        n_agent = n_asm = n_chest = 0
        obj_cfg = {k: o for k, o in game.objects.items()} if game is not None else {}
        agent_cfgs = game.agents if (game is not None and game.agents) else None
        for r in range(grid.shape[0]):
            for col in range(grid.shape[1]):
                code = int(grid[r, col])
                if code == 0:
                    continue
                oid += 1
                kind = c.cell_kinds[code - 1]
                tname = c.cell_type_names[code - 1]
                d: dict = {"id": oid, "type_name": tname}
                rr, cc = r, col
                if kind == 2:  # agent
                    i = n_agent
                    n_agent += 1
                    st = ast[i]
                    rr, cc = int(st[0]), int(st[1])
                    acfg = agent_cfgs[i] if agent_cfgs else (game.agent if game is not None else None)
                    d["inventory"] = {int(st[9 + 2 * k]): int(st[10 + 2 * k]) for k in range(int(st[8]))}
                    d.update(
                        group_id=int(acfg.team_id) if acfg is not None else groups[i],
                        group_name=c.group_names[groups[i]],
                        is_frozen=bool(st[4] != 0),
                        freeze_remaining=int(st[4]),
                        freeze_duration=int(acfg.freeze_duration) if acfg is not None else 0,
                        vibe=int(st[5]),
                        agent_id=i,
                        current_stat_reward=float(rew[i]),
                        steps_without_motion=int(st[6]),
                    )
                elif kind == 3:  # assembler
                    a = asm[n_asm]
                    n_asm += 1
                    ocfg = obj_cfg.get(tname)
                    d.update(
                        cooldown_remaining=max(0, int(np.uint32(a[5])) - step),
                        cooldown_duration=int(np.uint32(a[6])),
                        is_clipped=bool(a[4]),
                        is_clip_immune=bool(getattr(ocfg, "clip_immune", False)),
                        uses_count=int(np.uint32(a[7])),
                        max_uses=int(getattr(ocfg, "max_uses", 0)),
                        allow_partial_usage=bool(getattr(ocfg, "allow_partial_usage", False)),
                        # (a clipped station lists its unclipping protocols instead, mettagrid_c.cpp:872-873)
                        protocols=[dict(p) for p in (c.assembler_protocol_listing or {}).get(c.cell_names[code - 1], [])]
                        if int(a[4]) == 0
                        else [dict(c.unclip_protocol_listing[int(a[4]) - 2])] if int(a[4]) >= 2 else [],
                    )
                elif kind == 4:  # chest
                    ch = chs[n_chest]
                    n_chest += 1
                    d["inventory"] = {int(ch[5 + 2 * k]): int(ch[6 + 2 * k]) for k in range(int(ch[4]))}
                d["location"] = (cc, rr)
                d["r"], d["c"] = rr, cc
                if tname in ignore_types:
                    continue
                if use_bounds and (rr < min_row or rr >= max_row or cc < min_col or cc >= max_col):
                    continue
                out[oid] = d
        return out

    def set_inventory(self, env: int, agent: int, inventory: dict[int, int]) -> None:
        keys = order_from_pydict(list(inventory.keys()))
        items = np.asarray(keys, dtype=np.int32)
        amts = np.asarray([inventory[k] for k in keys], dtype=np.int32)
        self._check(
            self.L.mg_set_inventory(self._h, env, agent, len(keys), _p(items, ct.c_int32), _p(amts, ct.c_int32), self._stream())
        )

    def finished_summary(self, clear: bool = True) -> np.ndarray:
        out = np.zeros(3 + len(self.compiled.game_slot_names), np.float64)
        self._check(self.L.mg_finished_summary(self._h, _p(out, ct.c_double), int(clear), self._stream()))
        return out

    def dlpack(self, name: str):
        """DLPack capsule of one of the aliased buffers (`observations`, `rewards`, `terminals`, `truncations`,
        `actions`, `masks`, `episode_rewards`, `action_success`): the zero-copy hand-over to any DLPack consumer.
        The torch tensors of this object are themselves DLPack producers (`torch.from_dlpack(batch.rewards)`)."""
        from torch.utils.dlpack import to_dlpack

        t = getattr(self, name)
        if not isinstance(t, self._torch.Tensor):
            raise KeyError(name)
        return to_dlpack(t)

    def summary_tensor(self):
        """The finished-episode accumulator vector as a torch float64 tensor aliasing HBM (see finished_summary):
        `s = t.clone(); t.zero_(); dist.all_reduce(s)` keeps the multi-GPU exchange on the stream."""
        ptr, n = ct.POINTER(ct.c_double)(), ct.c_int32()
        self._check(self.L.mg_summary_buffer(self._h, ct.byref(ptr), ct.byref(n)))
        addr = ct.cast(ptr, ct.c_void_p).value
        return self._torch.as_tensor(_DevArray(addr, (int(n.value),), "<f8", self._owner), device=self.device)

    def status_tensor(self):
        """u32[4] in HBM as a torch int32 tensor: [0] OR of every environment's error bits (sticky), [1] episodes
        finished since creation.  Reading it needs no per-environment copy; `check_errors` is the raising form."""
        ptr, n = ct.POINTER(ct.c_uint32)(), ct.c_int32()
        self._check(self.L.mg_status_buffer(self._h, ct.byref(ptr), ct.byref(n)))
        addr = ct.cast(ptr, ct.c_void_p).value
        return self._torch.as_tensor(_DevArray(addr, (int(n.value),), "<i4", self._owner), device=self.device)

    def check_errors(self) -> None:
        """Raise RuntimeError if any environment flagged an internal error (the reference's core throws out of
        `step()`; here the flag is raised on the device and surfaces at the caller's next synchronisation point)."""
        self._check(self.L.mg_check_errors(self._h, self._stream()))

    def close(self) -> None:
        """Free the handle's HBM now.  Every tensor this object handed out (`observations`, `rewards`, ...,
        `summary_tensor()`, DLPack capsules, whatever `recv()` returned) aliases that memory and is INVALID afterwards
        -- clone what must outlive the environment before closing.  The object's own aliases are dropped here.
        Without `close()` the memory is released when the batch and all of those tensors have been collected."""
        owner = getattr(self, "_owner", None)
        if owner is not None and owner.h is not None:
            for name in _ALIASED:
                setattr(self, name, None)
            owner.destroy()
        self._h = ct.c_void_p()
