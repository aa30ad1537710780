"""BASELINE configs[2]: the reference's navigation evaluation suite (tests/golden/navigation_suite.npz, written by
oracle/gen_golden_navsuite.py from the reference's own maps, MapGen, config classes, converter and core): 43 maps of
different sizes, each with its own episode length, 4 agents per map."""

import json
from pathlib import Path

import numpy as np
import pytest

from metta_amd.builder import make_navigation
from metta_amd.compile import compile_game
from tests.helpers import digest, stats_equal

GOLD = Path(__file__).resolve().parent / "golden"


def load_suite():
    z = np.load(GOLD / "navigation_suite.npz", allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    names = np.array(meta["cell_names"])
    maps, off = [], 0
    for h, w in z["dims"]:
        maps.append(names[z["maps"][off : off + h * w].reshape(h, w)])
        off += h * w
    return z, meta, maps


def nav_cfg(max_steps: int):
    cfg = make_navigation(4)
    cfg.game.max_steps = int(max_steps)
    cfg.game.agent.rewards.inventory["heart"] = 0.333  # make_nav_eval_env (recipes/experiment/navigation.py:26-29)
    return cfg


def test_oracle_plays_every_suite_map_like_the_reference():
    from oracle.mg_oracle import make_oracle_env

    z, meta, maps = load_suite()
    assert len(maps) == 43 and len({m.shape for m in maps}) > 30  # a genuinely ragged pool
    for k, ent in enumerate(meta["entries"]):
        ms = int(z["max_steps"][k])
        comp = compile_game(nav_cfg(ms), init_obs_passes=2)
        o = make_oracle_env(comp, comp.encode_map(maps[k]), ent["seed"])
        assert digest(o.obs) == ent["obs0_digest"], ent["name"]
        acts = np.random.RandomState(ent["seed"]).randint(0, 5, size=(ms, 4)).astype(np.int32)
        od = z[f"od_{ent['name']}"]
        for t in range(ms):
            o.step(acts[t])
            assert digest(o.obs) == int(od[t]), f"{ent['name']}: obs digest at step {t + 1}"
        assert bool(o.terminals.all()) == ent["terminated"] and bool(o.truncations.all()) == ent["truncated"]
        assert np.allclose(o.episode_rewards, ent["episode_rewards"], rtol=1e-6, atol=0)
        d = stats_equal(o.get_episode_stats(), ent["stats"])
        assert d is None, f"{ent['name']}: {d}"


@pytest.mark.gpu
def test_navigation_suite_8192_envs():
    """8192 environments over the 43-map ragged pool with per-map episode lengths: environment e plays map e % 43 with
    that map's seed and action stream (so environments sharing a map are replicas).  Checked per map, at the middle
    and at the end of its episode, against the reference's digests; final stats and episode rewards at the episode's
    own last step; replicas identical."""
    import torch

    from metta_amd.batch import MettaGridBatch

    z, meta, maps = load_suite()
    P, E, A = len(maps), 8192, 4
    ms = z["max_steps"].astype(np.int64)
    seeds = [meta["entries"][e % P]["seed"] for e in range(E)]
    b = MettaGridBatch(nav_cfg(1000), E, maps, device=0, init_obs_passes=2)
    b.set_map_max_steps(ms)
    b.reset(seeds=seeds, map_ids=[e % P for e in range(E)])
    b.set_auto_reset(False)
    assert "fused" in b.step_path()
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert b.step_path() == "mg_step_fused_nomod_w18_kernel"  # 8192 > one round of 28 per CU, = one round of 32
    Tmax = int(ms.max())
    acts = np.zeros((Tmax, P, A), np.int32)
    for k, ent in enumerate(meta["entries"]):
        acts[: ms[k], k] = np.random.RandomState(ent["seed"]).randint(0, 5, size=(ms[k], A))
    acts_dev = torch.from_numpy(acts).cuda()
    map_of_env = (torch.arange(E, device="cuda") % P).long()
    obs0 = b.observations.view(E, A, 200, 3)
    for k, ent in enumerate(meta["entries"]):
        assert digest(obs0[k].cpu().numpy()) == ent["obs0_digest"], ent["name"]
    check_at = {}
    for k in range(P):
        check_at.setdefault(int(ms[k]) // 2 + 1, []).append((k, "mid"))
        check_at.setdefault(int(ms[k]), []).append((k, "end"))
    for t in range(1, Tmax + 1):
        b.step(acts_dev[t - 1][map_of_env].reshape(-1).contiguous())
        for k, which in check_at.get(t, ()):
            ent = meta["entries"][k]
            od = z[f"od_{ent['name']}"]
            o = b.observations.view(E, A, 200, 3)
            assert digest(o[k].cpu().numpy()) == int(od[t - 1]), f"{ent['name']}: obs digest at step {t} ({which})"
            assert torch.equal(o[k], o[k + P * 100]), f"{ent['name']}: replica environments diverged"
            if which == "end":
                term = b.terminals.view(E, A)[k].cpu().numpy()
                trunc = b.truncations.view(E, A)[k].cpu().numpy()
                assert bool(term.all()) == ent["terminated"] and bool(trunc.all()) == ent["truncated"], ent["name"]
                er = b.episode_rewards.view(E, A)[k].cpu().numpy()
                assert np.allclose(er, ent["episode_rewards"], rtol=1e-6, atol=0), ent["name"]
                d = stats_equal(b.get_episode_stats(k), ent["stats"])
                assert d is None, f"{ent['name']}: {d}"
            elif t > 1:
                assert not bool(b.terminals.view(E, A)[k].any().item()), f"{ent['name']}: ended early"
    b.check_errors()
    b.close()
