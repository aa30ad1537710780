"""V-trace / GAE advantages on the device: the drop-in for the reference's `compute_advantage`
(metta/rl/advantage.py:77-119) whose ROCm branch (`:94-97`) is a Python loop over the horizon (metta/rl/mps.py).

Same signature, argument meaning and checks (`vtrace_check_cuda`, pufferlib.cu:22-38: 2-D, equal shapes, one device,
float32; non-contiguous inputs are made contiguous).  The work is one HIP kernel launch on torch's current stream
(`mg_compute_advantage`, metta_amd/csrc/advantage.hip); there is no CPU path.
"""

from __future__ import annotations

import torch
from torch import Tensor

from metta_amd import native


def compute_puff_advantage(
    values: Tensor,
    rewards: Tensor,
    dones: Tensor,
    importance: Tensor,
    advantages: Tensor,
    gamma: float,
    gae_lambda: float,
    rho_clip: float,
    c_clip: float,
) -> None:
    """`torch.ops.pufferlib.compute_puff_advantage` (in place on `advantages`, [segments, horizon] f32, on the GPU)."""
    tensors = {"values": values, "rewards": rewards, "dones": dones, "importance": importance, "advantages": advantages}
    if values.dim() != 2:
        raise RuntimeError("Tensor must be 2D")
    num_rows, horizon = values.shape
    device = values.device
    for name, t in tensors.items():
        if t.dim() != 2:
            raise RuntimeError("Tensor must be 2D")
        if t.device != device:
            raise RuntimeError("All tensors must be on same device")
        if t.size(0) != num_rows:
            raise RuntimeError("First dimension must match num_steps")
        if t.size(1) != horizon:
            raise RuntimeError("Second dimension must match horizon")
        if t.dtype != torch.float32:
            raise RuntimeError("All tensors must be float32")
    if device.type != "cuda":
        raise RuntimeError("All tensors must be on GPU")
    if not advantages.is_contiguous():
        raise RuntimeError("advantages must be contiguous (it is written in place)")
    ins = [t if t.is_contiguous() else t.contiguous() for t in (values, rewards, dones, importance)]
    L = native.load()
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        native.check(
            L.mg_compute_advantage(
                *[t.data_ptr() for t in ins],
                advantages.data_ptr(),
                float(gamma),
                float(gae_lambda),
                float(rho_clip),
                float(c_clip),
                int(num_rows),
                int(horizon),
                stream,
            )
        )


def compute_advantage(
    values: Tensor,
    rewards: Tensor,
    dones: Tensor,
    importance_sampling_ratio: Tensor,
    advantages: Tensor,
    gamma: float,
    gae_lambda: float,
    device: torch.device,
    vtrace_rho_clip: float = 1.0,
    vtrace_c_clip: float = 1.0,
) -> Tensor:
    """metta/rl/advantage.py:77-119 with the custom kernel on ROCm; returns `advantages` (moved to `device`)."""
    device = torch.device(device)
    values, rewards, dones, importance_sampling_ratio, advantages = (
        t.to(device) for t in (values, rewards, dones, importance_sampling_ratio, advantages)
    )
    compute_puff_advantage(
        values, rewards, dones, importance_sampling_ratio, advantages, gamma, gae_lambda, vtrace_rho_clip, vtrace_c_clip
    )
    return advantages


def normalize_advantage_distributed(adv: Tensor, norm_adv: bool = True) -> Tensor:
    """metta/rl/advantage.py:122-150: normalise advantages over ALL ranks' elements with one all-reduce of
    [sum, sum of squares, count] (RCCL on ROCm, `backend="nccl"`); single process: `(adv - mean) / (std + 1e-8)`
    with torch's unbiased std, as the reference does.  Shape is preserved."""
    if not norm_adv:
        return adv
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        flat = adv.reshape(-1)
        stats = torch.stack(
            [flat.sum(), (flat * flat).sum(), torch.tensor(float(flat.numel()), dtype=adv.dtype, device=adv.device)]
        )
        torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM)
        mean = stats[0] / stats[2]
        var = stats[1] / stats[2] - mean * mean
        std = torch.sqrt(var.clamp(min=1e-8))
        return (adv - mean) / (std + 1e-8)
    return (adv - adv.mean()) / (adv.std() + 1e-8)
