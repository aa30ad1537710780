"""Device-resident rollout loop with a recurrent policy (BASELINE config 5).

The reference's `CoreTrainingLoop.rollout_phase` (`metta/rl/training/core.py:67-189`) copies observations host ->
device and actions device -> host every iteration.  Here observations, rewards and actions never leave HBM: the
policy reads `vecenv.observations` in place and `send` hands the action tensor's device pointer to the step
kernels (`mg_step_actions`).

`LSTMPolicy` is the mettagrid-package reference policy (`PY/policy/lstm.py:19-36`): flatten [T*3] uint8 tokens ->
Linear(128) -> Linear(128) -> LSTM(128) -> action logits / value.
"""

from __future__ import annotations

import time

import torch
import torch.nn as nn


class LSTMPolicy(nn.Module):
    def __init__(self, num_tokens: int, num_actions: int, hidden: int = 128):
        super().__init__()
        self.enc = nn.Sequential(nn.Linear(num_tokens * 3, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU())
        self.rnn = nn.LSTMCell(hidden, hidden)
        self.pi = nn.Linear(hidden, num_actions)
        self.v = nn.Linear(hidden, 1)
        self.hidden = hidden

    def initial_state(self, n: int, device, dtype=torch.float32):
        z = torch.zeros(n, self.hidden, device=device, dtype=dtype)
        return z, z.clone()

    def forward(self, obs_u8: torch.Tensor, state):
        x = obs_u8.reshape(obs_u8.shape[0], -1).to(self.enc[0].weight.dtype) * (1.0 / 255.0)
        h, c = self.rnn(self.enc(x), state)
        return self.pi(h), self.v(h).squeeze(-1), (h, c)


@torch.no_grad()
def rollout(vecenv, policy: LSTMPolicy, steps: int, state=None, generator=None):
    """`steps` iterations of recv -> policy -> send, all on the device.  Returns (agent_steps, seconds, state, infos)."""
    dev = vecenv.observations.device
    n = vecenv.num_agents
    if state is None:
        state = policy.initial_state(n, dev, next(policy.parameters()).dtype)
    infos_all = []
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        o, r, d, t, ta, infos, ids, mask = vecenv.recv()
        logits, value, state = policy(o, state)
        done = (d | t).to(state[0].dtype).unsqueeze(1)
        state = (state[0] * (1 - done), state[1] * (1 - done))  # reset recurrent state where an episode ended
        actions = torch.multinomial(torch.softmax(logits.float(), dim=-1), 1, generator=generator).squeeze(1).to(torch.int32)
        vecenv.send(actions)
        if infos:
            infos_all.extend(infos)
    torch.cuda.synchronize(dev)
    return n * steps, time.perf_counter() - t0, state, infos_all
