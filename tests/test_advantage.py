"""Advantage kernel: oracle pinned by the reference, HIP path bit-exact against the oracle.

CPU tests: oracle/advantage_oracle.c against tests/golden/advantage_cases.npz (outputs of the reference's compiled
CPU extension and of metta/rl/mps.py, written by oracle/gen_golden_advantage.py) and, where oracle/_ref is built,
against the reference extension directly on fresh inputs.  GPU tests go through the C-ABI (mg_compute_advantage).

Tolerances: bit-exact (u32 views, NaN payloads included) against the reference's C++ function; rtol 1e-4 / atol 1e-5
against the reference's pure-torch loop, which rounds differently (gamma*lambda folded in double).
"""

import numpy as np
import pytest

from oracle.advantage_oracle import make_case, oracle_advantage, reference_advantage, reference_available


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


def _golden(golden_dir):
    g = np.load(golden_dir / "advantage_cases.npz")
    for i, c in enumerate(g["cases"]):
        seed, B, T, kind, gamma, lam, rho, cc = c
        ins = [g[f"{k}_{i}"] for k in ("values", "rewards", "dones", "importance")]
        yield i, ins, (float(gamma), float(lam), float(rho), float(cc)), g[f"adv_ref_{i}"], g[f"adv_mps_{i}"]


def test_oracle_matches_reference_goldens(golden_dir):
    n = 0
    for i, ins, hp, ref, mps in _golden(golden_dir):
        out = oracle_advantage(*ins, *hp)
        assert _bits_equal(out, ref), f"case {i}: oracle differs from the reference's compiled extension"
        fin = np.isfinite(ref) & np.isfinite(mps)
        assert np.allclose(out[fin], mps[fin], rtol=1e-4, atol=1e-5), f"case {i}: oracle vs metta/rl/mps.py"
        n += 1
    assert n == 7


def test_oracle_matches_reference_extension_fresh_inputs():
    if not reference_available():
        pytest.skip("oracle/_ref/pufferlib_C.so not built (make -C oracle ref_puff where /root/reference exists)")
    for seed, (B, T) in enumerate([(1, 2), (7, 5), (64, 64), (100, 129), (16, 512)]):
        for kind in ("rollout", "edge"):
            ins = make_case(100 + seed, B, T, kind)
            hp = (0.977, 0.916, 1.0, 1.0) if seed % 2 == 0 else (0.9, 0.99, 0.7, 1.5)
            assert _bits_equal(oracle_advantage(*ins, *hp), reference_advantage(*ins, *hp)), (B, T, kind)


def test_last_column_is_left_alone():
    ins = make_case(3, 6, 9)
    init = np.full((6, 9), 7.5, np.float32)
    out = oracle_advantage(*ins, 0.99, 0.95, advantages=init)
    assert np.all(out[:, -1] == 7.5) and not np.any(out[:, :-1] == 7.5)


def test_host_checks_mirror_vtrace_check():
    import torch

    from metta_amd.advantage import compute_puff_advantage

    z = torch.zeros(4, 8)
    with pytest.raises(RuntimeError, match="2D"):
        compute_puff_advantage(z[0], z, z, z, z, 0.99, 0.95, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="horizon"):
        compute_puff_advantage(z, z, torch.zeros(4, 7), z, z, 0.99, 0.95, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="num_steps"):
        compute_puff_advantage(z, z, torch.zeros(3, 8), z, z, 0.99, 0.95, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="float32"):
        compute_puff_advantage(z, z, z.double(), z, z, 0.99, 0.95, 1.0, 1.0)
    with pytest.raises(RuntimeError, match="GPU"):  # no CPU path in the product
        compute_puff_advantage(z, z, z, z, z, 0.99, 0.95, 1.0, 1.0)


# ---------------------------------------------------------------------------------------------------------------
# GPU: through the C-ABI
# ---------------------------------------------------------------------------------------------------------------
def _hip(ins, hp, init=None):
    import torch

    from metta_amd.advantage import compute_puff_advantage

    ts = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in ins]
    adv = torch.zeros_like(ts[0]) if init is None else torch.from_numpy(init.copy()).cuda()
    compute_puff_advantage(*ts, adv, *hp)
    torch.cuda.synchronize()
    return adv.cpu().numpy()


@pytest.mark.gpu
def test_hip_matches_goldens(golden_dir):
    for i, ins, hp, ref, mps in _golden(golden_dir):
        out = _hip(ins, hp)
        assert _bits_equal(out, ref), f"case {i}: HIP differs from the reference's compiled extension"


@pytest.mark.gpu
def test_hip_matches_oracle_shapes():
    # vector path (horizon % 4 == 0), scalar path, ragged rows (not a multiple of 16), ragged tiles (not of 64),
    # horizon 1 / 2 / 3, one row; rollout-like and edge inputs (non-binary dones, inf / nan / denormal weights)
    shapes = [(1, 1), (1, 2), (3, 3), (5, 4), (16, 64), (17, 64), (15, 68), (33, 63), (40, 65), (64, 128), (100, 129),
              (7, 256), (250, 260), (1000, 32), (31, 1024)]
    for s, (B, T) in enumerate(shapes):
        for kind in ("rollout", "edge"):
            ins = make_case(200 + s, B, T, kind)
            hp = (0.977, 0.916, 1.0, 1.0) if s % 2 == 0 else (0.9, 0.99, 0.7, 1.5)
            assert _bits_equal(_hip(ins, hp), oracle_advantage(*ins, *hp)), (B, T, kind)


@pytest.mark.gpu
def test_hip_leaves_last_column_and_handles_views():
    import torch

    from metta_amd.advantage import compute_advantage

    ins = make_case(5, 37, 100)
    init = np.full((37, 100), -3.25, np.float32)
    out = _hip(ins, (0.99, 0.95, 1.0, 1.0), init=init)
    assert _bits_equal(out, oracle_advantage(*ins, 0.99, 0.95, advantages=init))
    assert np.all(out[:, -1] == -3.25)
    # unaligned base pointer (scalar path although horizon % 4 == 0) and a non-contiguous input
    flat = [torch.zeros(37 * 100 + 1).cuda() for _ in range(5)]
    views = [f[1:].view(37, 100) for f in flat]
    for v, x in zip(views[:4], ins):
        v.copy_(torch.from_numpy(x))
    res = compute_advantage(views[0], views[1], views[2], views[3], views[4], 0.99, 0.95, torch.device("cuda"))
    assert _bits_equal(res.cpu().numpy(), oracle_advantage(*ins, 0.99, 0.95))
    vt = torch.from_numpy(np.ascontiguousarray(ins[0].T)).cuda().t()  # same values, column-major storage
    assert not vt.is_contiguous()
    adv = torch.zeros(37, 100).cuda()
    res = compute_advantage(vt, *[torch.from_numpy(x).cuda() for x in ins[1:]], adv, 0.99, 0.95, torch.device("cuda"))
    assert _bits_equal(res.cpu().numpy(), oracle_advantage(*ins, 0.99, 0.95))


@pytest.mark.gpu
def test_hip_full_size_trainer_batch():
    # the trainer's default batch: 2 097 152 steps / bptt_horizon 256 = 8192 segments (metta/rl/trainer_config.py:86-89)
    ins = make_case(9, 8192, 256)
    hp = (0.977, 0.916, 1.0, 1.0)
    out = _hip(ins, hp)
    assert _bits_equal(out, oracle_advantage(*ins, *hp))
    # size-independent property: rows are independent -- any permutation of the rows permutes the result
    perm = np.random.default_rng(0).permutation(8192)
    assert _bits_equal(_hip([x[perm] for x in ins], hp), out[perm])
