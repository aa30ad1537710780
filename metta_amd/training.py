"""Device-resident rollout phase: the seam between the vectorised environment and a PPO trainer.

Mirrors, for the rollout path only:

* `calculate_batch_sizes` -- metta/rl/training/batch.py:6-21 (same signature and arithmetic);
* `DeviceExperience` -- the segmented experience store of metta/rl/training/experience.py:17-160: `[segments,
  bptt_horizon]` buffers, one row slot per agent, `store(data, env_id: slice)` writes one time step of the agents
  `env_id` names at `(row_slot_ids[env_id], t_in_row)`, rows that fill up are retired and the agents get fresh slots
  (`_reset_completed_episodes`), `ready_for_training` once `full_rows >= segments`.  Plain tensors instead of a
  TensorDict (tensordict / torchrl are not part of this path); the host keeps the per-slice time index as Python
  integers exactly like the reference's `t_in_row_cpu`, so storing never reads the device;
* `rollout_phase` -- `CoreTrainingLoop.rollout_phase` (metta/rl/training/core.py:67-189) with the two lines that
  move data across PCIe removed: the policy reads the observation tensor the environment wrote in HBM
  (`td["env_obs"] = o.to(device)` at :104 is a no-op alias) and `env.send_actions` takes the action tensor itself
  (`td["actions"].cpu().numpy()` at :180 is gone).  Nothing in the loop reads device memory from the host.
* `compute_rollout_advantages` -- metta/rl/advantage.py:77-119 on the filled store (mg_compute_advantage).
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Optional, Tuple

import torch
from torch import Tensor


def calculate_batch_sizes(
    forward_pass_minibatch_target_size: int,
    num_agents: int,
    num_workers: int,
    async_factor: int,
) -> Tuple[int, int, int]:
    """(target_batch_size, batch_size, num_envs), all counted in environments (batch.py:6-21)."""
    target_batch_size = forward_pass_minibatch_target_size // num_agents
    min_required = max(2, num_workers)  # pufferlib needs a batch of at least 2 environments
    if target_batch_size < min_required:
        target_batch_size = min_required
    batch_size = (target_batch_size // num_workers) * num_workers
    num_envs = batch_size * async_factor
    return target_batch_size, batch_size, num_envs


class DeviceExperience:
    """Segmented experience store in HBM (experience.py:17-160).  `spec`: name -> (trailing shape, dtype)."""

    def __init__(self, total_agents: int, batch_size: int, bptt_horizon: int, spec: dict, device):
        self.total_agents = int(total_agents)
        self.batch_size = int(batch_size)
        self.bptt_horizon = int(bptt_horizon)
        self.device = torch.device(device)
        self.segments = self.batch_size // self.bptt_horizon
        if self.total_agents > self.segments:
            raise ValueError(
                f"batch_size ({batch_size}) is too small for {total_agents} agents.\n"
                f"Segments = batch_size // bptt_horizon = {batch_size} // {bptt_horizon} = {self.segments}\n"
                f"But we need segments >= total_agents ({total_agents}).\n"
                f"Please set trainer.batch_size >= {total_agents * bptt_horizon} in your configuration."
            )
        self.buffer: dict[str, Tensor] = {
            k: torch.zeros((self.segments, self.bptt_horizon, *shape), dtype=dtype, device=self.device)
            for k, (shape, dtype) in spec.items()
        }
        self._range_tensor = torch.arange(self.total_agents, device=self.device, dtype=torch.int64)
        self.row_slot_ids = self._range_tensor % self.segments
        self.t_in_row = torch.zeros(self.total_agents, device=self.device, dtype=torch.int64)
        self._t_in_row_host: dict[int, int] = {}  # slice start -> time index (the reference's t_in_row_cpu)
        self.free_idx = self.total_agents % self.segments
        self.full_rows = 0

    @property
    def ready_for_training(self) -> bool:
        return self.full_rows >= self.segments

    def store(self, data: dict, env_id: slice) -> None:
        assert isinstance(env_id, slice), f"env_id expected to be a slice for segmented storage, got {type(env_id).__name__}"
        t = self._t_in_row_host.get(env_id.start, 0)
        rows = self.row_slot_ids[env_id]
        for k, buf in self.buffer.items():
            if k in data:
                buf[rows, t] = data[k].to(buf.dtype)
        self.t_in_row[env_id] += 1
        self._t_in_row_host[env_id.start] = t + 1
        if t + 1 >= self.bptt_horizon:
            self._reset_completed_episodes(env_id)

    def _reset_completed_episodes(self, env_id: slice) -> None:
        num_full = env_id.stop - env_id.start
        self.row_slot_ids[env_id] = (self.free_idx + self._range_tensor[:num_full]) % self.segments
        self.t_in_row[env_id] = 0
        self._t_in_row_host[env_id.start] = 0
        self.free_idx = (self.free_idx + num_full) % self.segments
        self.full_rows += num_full

    def reset_for_rollout(self) -> None:
        self.full_rows = 0
        self.free_idx = self.total_agents % self.segments
        self.row_slot_ids = self._range_tensor % self.segments
        self.t_in_row.zero_()
        self._t_in_row_host.clear()


@dataclass
class RolloutResult:
    raw_infos: list = field(default_factory=list)
    agent_steps: int = 0
    training_env_id: Optional[slice] = None


def experience_spec(num_tokens: int, token_dim: int = 3) -> dict:
    """The keys PPO's rollout stores (metta/rl/loss/ppo.py `get_experience_spec` + the policy's env_obs)."""
    f32 = torch.float32
    return {
        "env_obs": ((num_tokens, token_dim), torch.uint8),
        "actions": ((), torch.int32),
        "act_log_prob": ((), f32),
        "values": ((), f32),
        "rewards": ((), f32),
        "dones": ((), f32),
        "truncateds": ((), f32),
    }


@torch.no_grad()
def rollout_phase(
    env,
    policy,
    experience: DeviceExperience,
    policy_state: Optional[dict] = None,
    generator=None,
    on_step: Optional[Callable[[dict, slice], None]] = None,
) -> RolloutResult:
    """Collect experience until the store is full.  `env`: a `DeviceTrainingEnvironment`; `policy(obs_u8, state) ->
    (logits, values, state)` with a recurrent state per agent (kept per env slice in `policy_state`)."""
    experience.reset_for_rollout()
    result = RolloutResult()
    state_by_slice = policy_state if policy_state is not None else {}
    while not experience.ready_for_training:
        o, r, d, t, ta, info, training_env_id, mask, num_steps = env.get_observations()
        key = training_env_id.start
        n = training_env_id.stop - training_env_id.start
        st = state_by_slice.get(key)
        if st is None:
            st = policy.initial_state(n, o.device, next(policy.parameters()).dtype)
        done = (d | t).to(st[0].dtype).unsqueeze(1)
        st = (st[0] * (1 - done), st[1] * (1 - done))  # an episode that ended starts from a fresh recurrent state
        logits, values, st = policy(o, st)
        state_by_slice[key] = st
        logp_all = torch.log_softmax(logits.float(), dim=-1)
        actions = torch.multinomial(logp_all.exp(), 1, generator=generator).squeeze(1)
        td = {
            "env_obs": o,
            "actions": actions.to(torch.int32),
            "act_log_prob": logp_all.gather(1, actions.unsqueeze(1)).squeeze(1),
            "values": values.float(),
            "rewards": r,
            "dones": d.to(torch.float32),
            "truncateds": t.to(torch.float32),
        }
        experience.store(td, training_env_id)
        if on_step is not None:
            on_step(td, training_env_id)
        env.send_actions(td["actions"])  # the device tensor itself: no .cpu().numpy()
        if info:
            result.raw_infos.extend(info)
        result.agent_steps += num_steps
        result.training_env_id = training_env_id
    return result


def compute_rollout_advantages(experience: DeviceExperience, gamma: float = 0.977, gae_lambda: float = 0.891477,
                               rho_clip: float = 1.0, c_clip: float = 1.0) -> Tensor:
    """`compute_advantage` (metta/rl/advantage.py:77-119) over the filled store; defaults = trainer_config.py's PPO."""
    from metta_amd.advantage import compute_advantage

    b = experience.buffer
    adv = torch.zeros_like(b["values"])
    ratio = torch.ones_like(b["values"])
    return compute_advantage(b["values"], b["rewards"], b["dones"], ratio, adv, gamma, gae_lambda, experience.device, rho_clip, c_clip)
