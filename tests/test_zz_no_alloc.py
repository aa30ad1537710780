"""mg_step_noalloc: the allocation-free variant of mg_step / mg_step_actions (include/mettagrid_hip.h).

Called through the C-ABI and checked, bit for bit, against the existing entry point on twin handles (same config,
maps, seeds, actions): two runs in a row, the second with a smaller input (fewer environments), so that whatever the
first run left in the GPU's static scratch array is still there when the second starts.  Covers the three step paths:
the fused one-launch step (arena: its hand-overs live in LDS, the scratch stays idle and the entry point must still
agree), the three-launch fast path (`MG_NO_FUSED=1`: mode words + pending assembler uses + executed actions go through
the scratch) and the general path (combat: executed actions).  Collected last (the round's end runs pytest -x).
"""

import ctypes as ct

import numpy as np
import pytest

from metta_amd.builder import make_arena, make_combat_stress
from tests.helpers import load_map_pool

pytestmark = pytest.mark.gpu


def _twin_run(cfg, maps, num_envs, steps, n_actions, seed):
    import torch

    from metta_amd import native
    from metta_amd.batch import MettaGridBatch

    seeds = [seed + e for e in range(num_envs)]
    map_ids = [e % len(maps) for e in range(num_envs)]
    ref = MettaGridBatch(cfg, num_envs, maps, device=0, seeds=seeds, map_ids=map_ids)
    new = MettaGridBatch(cfg, num_envs, maps, device=0, seeds=seeds, map_ids=map_ids)
    L = new.L
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = new.num_agents
    for t in range(steps):
        acts = torch.randint(0, n_actions, (num_envs * A,), generator=g, dtype=torch.int32).cuda()
        ref.step(acts)  # mg_step_actions
        if t % 2 == 0:  # device action tensor read in place
            native.check(L.mg_step_noalloc(new._h, ct.c_void_p(acts.data_ptr()), new._stream()))
        else:  # NULL = the handle's own action buffer
            new.actions.copy_(acts)
            native.check(L.mg_step_noalloc(new._h, None, new._stream()))
        torch.cuda.synchronize()
        for name in ("observations", "rewards", "terminals", "truncations", "action_success", "episode_rewards"):
            assert torch.equal(getattr(ref, name), getattr(new, name)), f"step {t + 1}: {name} differs"
    for e in range(0, num_envs, max(1, num_envs // 4)):
        assert np.array_equal(ref.agent_state(e), new.agent_state(e))
        assert np.array_equal(ref.assembler_state(e), new.assembler_state(e))
        assert ref.get_episode_stats(e) == new.get_episode_stats(e)
    new.check_errors()
    ref.close()
    new.close()


def test_noalloc_step_matches_mg_step_fast_path():
    cfg = make_arena(24)
    cfg.game.max_steps = 40  # episode turnover (template restarts) inside the run
    maps = load_map_pool()[:8]
    _twin_run(cfg, maps, 96, 100, 5, seed=5)
    _twin_run(cfg, maps, 7, 100, 5, seed=9)  # smaller input, right after the larger one


def test_noalloc_step_matches_mg_step_three_launch_fast_path(monkeypatch):
    monkeypatch.setenv("MG_NO_FUSED", "1")  # read by mg_set_maps: act_fast -> uses -> obs, hand-overs in the scratch
    cfg = make_arena(24)
    cfg.game.max_steps = 40
    maps = load_map_pool()[:8]
    _twin_run(cfg, maps, 64, 80, 5, seed=15)
    _twin_run(cfg, maps, 5, 80, 5, seed=19)


def test_noalloc_step_matches_mg_step_general_path():
    cfg = make_combat_stress(256, 48)
    cfg.game.max_steps = 30
    maps = [cfg.game.map_builder.build(np.random.default_rng(3 + i)) for i in range(2)]
    from metta_amd.compile import compile_game

    nA = len(compile_game(cfg, init_obs_passes=2).action_names)
    _twin_run(cfg, maps, 6, 50, nA, seed=21)
    _twin_run(cfg, maps, 2, 50, nA, seed=22)


def test_noalloc_step_rejects_sizes_beyond_the_static_scratch():
    from metta_amd import native
    from metta_amd.batch import MettaGridBatch

    cfg = make_arena(24)
    maps = load_map_pool()[:2]
    E = 131072 // 24 + 1  # one environment too many for the 131072-agent scratch
    b = MettaGridBatch(cfg, E, maps, device=0, seeds=list(range(E)), map_ids=[e % 2 for e in range(E)])
    rc = b.L.mg_step_noalloc(b._h, None, b._stream())
    assert rc != 0 and "131072" in b.L.mg_last_error().decode()
    native.check(b.L.mg_step(b._h, b._stream()))  # the existing entry point is unaffected
    b.close()
