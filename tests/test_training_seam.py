"""The rollout seam: batch sizing, the segmented experience store, make_vecenv / async groups, and the device-resident
rollout phase (BASELINE configs[4])."""

import numpy as np
import pytest
import torch

from metta_amd.training import DeviceExperience, calculate_batch_sizes, experience_spec


def test_calculate_batch_sizes_known_answers():
    """metta/rl/training/batch.py:6-21, evaluated by hand for the trainer's defaults and the edge cases."""
    assert calculate_batch_sizes(4096, 24, 1, 2) == (170, 170, 340)  # TrainingEnvironmentConfig defaults, arena
    assert calculate_batch_sizes(4096, 24, 8, 2) == (170, 168, 336)  # rounded down to a multiple of the workers
    assert calculate_batch_sizes(10, 24, 4, 1) == (4, 4, 4)  # at least max(2, num_workers) environments
    assert calculate_batch_sizes(4096, 4, 1, 1) == (1024, 1024, 1024)  # navigation, serial
    assert calculate_batch_sizes(1, 24, 1, 3) == (2, 2, 6)


def test_experience_store_segments_and_row_rotation():
    """metta/rl/training/experience.py:41-58,114-141: one row slot per agent, a full row retires and the agent takes
    the next free slot; `ready_for_training` once every segment has been filled."""
    A, H = 6, 4
    ex = DeviceExperience(total_agents=A, batch_size=12 * H, bptt_horizon=H, spec={"x": ((2,), torch.float32), "a": ((), torch.int32)}, device="cpu")
    assert ex.segments == 12 and not ex.ready_for_training
    halves = [slice(0, 3), slice(3, 6)]
    step = 0
    while not ex.ready_for_training:
        for sl in halves:
            n = sl.stop - sl.start
            base = torch.arange(sl.start, sl.stop, dtype=torch.float32)
            ex.store({"x": torch.stack([base, torch.full((n,), float(step))], 1), "a": torch.full((n,), step, dtype=torch.int32)}, sl)
        step += 1
    assert step == 2 * H and ex.full_rows == 12
    # rows 0..5: the agents' first H steps; rows 6..11: their next H steps, in retirement order (slice 0 first)
    for row in range(12):
        agent = row % 6
        assert torch.equal(ex.buffer["x"][row, :, 0], torch.full((H,), float(agent)))
        first = 0 if row < 6 else H
        assert ex.buffer["a"][row].tolist() == list(range(first, first + H))
    ex.reset_for_rollout()
    assert ex.full_rows == 0 and ex.row_slot_ids.tolist() == [0, 1, 2, 3, 4, 5]
    with pytest.raises(ValueError, match="too small for 6 agents"):
        DeviceExperience(total_agents=6, batch_size=5 * H, bptt_horizon=H, spec={}, device="cpu")


def test_make_vecenv_argument_checks():
    from metta_amd.builder import make_arena
    from metta_amd.vecenv import make_vecenv

    with pytest.raises(ValueError, match="num_envs must be at least 1"):
        make_vecenv(make_arena(24), "device", num_envs=0)
    with pytest.raises(ValueError, match="Invalid --vector"):
        make_vecenv(make_arena(24), "multiprocessing", num_envs=2)


class _SyncCounter:
    """Counts host reads of device memory while active."""

    def __enter__(self):
        self.count = 0
        self._orig = {}
        counter = self

        def wrap(name):
            orig = getattr(torch.Tensor, name)
            self._orig[name] = orig

            def f(t, *a, **k):
                if t.is_cuda:
                    counter.count += 1
                return orig(t, *a, **k)

            setattr(torch.Tensor, name, f)

        for name in ("item", "cpu", "tolist", "numpy", "__bool__", "__int__", "__float__"):
            wrap(name)
        self._sync = torch.cuda.synchronize

        def sync(*a, **k):
            counter.count += 1
            return self._sync(*a, **k)

        torch.cuda.synchronize = sync
        return self

    def __exit__(self, *exc):
        for name, orig in self._orig.items():
            setattr(torch.Tensor, name, orig)
        torch.cuda.synchronize = self._sync


@pytest.mark.gpu
def test_device_rollout_phase_4096_envs_against_oracle():
    """BASELINE configs[4] on one GPU: 4096 arena envs x 24 agents in two async groups feed an LSTM policy; the rollout
    fills a [98304, 8] experience store without a single host read of device memory, the store feeds the advantage
    kernel, and the stored observations / rewards of sampled environments equal the oracle's when it replays the
    actions the policy sampled."""
    from metta_amd.builder import make_arena
    from metta_amd.compile import compile_game
    from metta_amd.rollout import LSTMPolicy
    from metta_amd.training import compute_rollout_advantages, rollout_phase
    from metta_amd.vecenv import AsyncDeviceVecEnv, DeviceTrainingEnvironment
    from oracle.mg_oracle import make_oracle_env
    from tests.helpers import load_map_pool

    cfg = make_arena(24)
    maps = load_map_pool()
    E, A, H = 4096, 24, 8
    env = DeviceTrainingEnvironment(cfg, num_envs=E, device=0, seed=100, async_factor=2, maps=maps, early_reset=False)
    assert isinstance(env.vecenv, AsyncDeviceVecEnv) and env.batch_info.batch_size == 2048 and env.batch_info.num_envs == E
    total_agents = env.total_parallel_agents
    assert total_agents == E * A
    ex = DeviceExperience(total_agents, total_agents * H, H, experience_spec(200), device="cuda")
    torch.manual_seed(0)
    policy = LSTMPolicy(200, 5).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    seen = []
    torch.cuda.synchronize()
    with _SyncCounter() as sc:
        res = rollout_phase(env, policy, ex, generator=g, on_step=lambda td, sl: seen.append(sl))
        n_reads = sc.count
    assert n_reads == 0, f"{n_reads} host reads of device memory inside the rollout loop"
    torch.cuda.synchronize()
    assert ex.ready_for_training and res.agent_steps == total_agents * H
    assert seen[:2] == [slice(0, 2048 * A), slice(2048 * A, 4096 * A)] and len(seen) == 2 * H
    adv = compute_rollout_advantages(ex)
    assert adv.shape == (total_agents, H) and bool(torch.isfinite(adv).all().item())
    # replay the sampled actions of a few environments of both groups through the oracle
    comp = compile_game(cfg, init_obs_passes=2)
    obs = ex.buffer["env_obs"]
    for e in (0, 1, 777, 2047, 2048, 3000, 4095):
        grp, local = divmod(e, 2048)
        o = make_oracle_env(comp, comp.encode_map(maps[local % len(maps)]), 100 + e)
        rows = slice(e * A, (e + 1) * A)
        acts = ex.buffer["actions"][rows].cpu().numpy()  # [A, H]
        for t in range(H):
            assert np.array_equal(obs[rows, t].cpu().numpy(), o.obs), f"env {e} step {t}: stored observation"
            assert np.array_equal(ex.buffer["rewards"][rows, t].cpu().numpy(), o.rewards if t else np.zeros(A, np.float32))
            o.step(acts[:, t])
    env.vecenv.flush_infos()
    env.close()


@pytest.mark.gpu
def test_training_environment_reference_sizing_and_async_slices():
    """`get_observations` serves the async groups as contiguous agent-id slices (training_environment.py:253-270) and the
    default sizing is the reference's (4096 agents per forward pass, async_factor 2 -> 170 envs per group, 340 envs)."""
    from metta_amd.builder import make_arena
    from metta_amd.vecenv import DeviceTrainingEnvironment
    from tests.helpers import load_map_pool

    env = DeviceTrainingEnvironment(make_arena(24), device=0, seed=3, async_factor=2, maps=load_map_pool()[:4])
    bi = env.batch_info
    assert (bi.target_batch_size, bi.batch_size, bi.num_envs) == (170, 170, 340)
    slices = []
    for _ in range(4):
        o, r, d, t, ta, info, sl, mask, n = env.get_observations()
        assert o.shape == (170 * 24, 200, 3) and n == 170 * 24 and o.is_cuda
        slices.append(sl)
        env.send_actions(torch.zeros(n, dtype=torch.int32, device="cuda"))
    assert slices == [slice(0, 4080), slice(4080, 8160), slice(0, 4080), slice(4080, 8160)]
    env.close()
