"""Host-side pieces of the device vecenv that need no GPU: the lazily decoded statistics sections of an episode's
infos and the timer sections (`StatsTracker._add_timing_info`, PY/envs/stats_tracker.py:109-140)."""

import copy
import json
import pickle
import time

from metta_amd.vecenv import _LazyStats, _Timers


def _lazy(calls):
    def fill():
        calls.append(1)
        return {"a": 1.0, "b": 2.5}

    return _LazyStats(fill)


def test_lazy_stats_behave_as_the_dict_they_become():
    for use in (
        lambda d: d["a"] == 1.0, lambda d: len(d) == 2, lambda d: "b" in d, lambda d: list(d) == ["a", "b"],
        lambda d: d == {"a": 1.0, "b": 2.5}, lambda d: d.get("b") == 2.5 and d.get("z", 7) == 7,
        lambda d: dict(d) == {"a": 1.0, "b": 2.5}, lambda d: {**d} == {"a": 1.0, "b": 2.5},
        lambda d: sorted(d.items()) == [("a", 1.0), ("b", 2.5)], lambda d: sorted(d.values()) == [1.0, 2.5],
        lambda d: json.loads(json.dumps(d)) == {"a": 1.0, "b": 2.5}, lambda d: copy.deepcopy(d) == {"a": 1.0, "b": 2.5},
        lambda d: pickle.loads(pickle.dumps(d)) == {"a": 1.0, "b": 2.5}, lambda d: (d | {"c": 3}) == {"a": 1.0, "b": 2.5, "c": 3},
        lambda d: ({"c": 3} | d) == {"c": 3, "a": 1.0, "b": 2.5}, lambda d: bool(d), lambda d: d.copy() == {"a": 1.0, "b": 2.5},
        lambda d: repr(d) == repr({"a": 1.0, "b": 2.5}), lambda d: sum(d[k] for k in d) == 3.5,
    ):  # fmt: skip
        calls = []
        d = _lazy(calls)
        assert calls == [], "nothing is decoded before the section is looked at"
        assert use(d)
        assert use(d)
        assert calls == [1], "decoded exactly once"


def test_lazy_stats_keep_what_a_consumer_writes():
    calls = []
    d = _lazy(calls)
    d["a"] = 9.0  # (set before anything was read: the decode must not overwrite it)
    assert d == {"a": 9.0, "b": 2.5} and calls == [1]
    d = _lazy([])
    d.update(z=1)
    assert d.pop("a") == 1.0 and d.setdefault("q", 4) == 4 and d == {"b": 2.5, "z": 1, "q": 4}
    d = _lazy([])
    d |= {"b": 0}
    assert d == {"a": 1.0, "b": 0}
    d = _lazy(calls := [])
    d.clear()
    assert d == {} and calls == []


def test_timer_sections_follow_the_reference_formulas():
    """Key sets and arithmetic of stats_tracker.py:109-140 as the reference actually behaves: it pops the idle timer
    under the name "thread_idle" while the timer is called "sim.thread_idle", so the idle time stays among the per-timer
    entries and `frac/thread_idle` is 0 (the fixture tests/golden/episode_infos.json holds the reference's key sets)."""
    tm = _Timers()
    tm.start("sim.thread_idle")
    time.sleep(0.002)
    tm.stop("sim.thread_idle")
    tm.start("sim.step")
    tm.start("sim.step.c_sim")
    time.sleep(0.003)
    tm.stop("sim.step.c_sim")
    tm.stop("sim.step")
    s = tm.sections()
    per, cum = s["timing_per_epoch"], s["timing_cumulative"]
    ops = {k[len("msec/"):] for k in per if k.startswith("msec/")}
    assert {"sim.step", "sim.step.c_sim", "global", "sim.thread_idle"} <= ops
    assert {k for k in per if k.startswith("active_frac/")} == {"active_frac/" + o for o in ops}
    assert {k for k in cum if k.startswith("active_frac/")} == {"active_frac/" + o for o in ops if o != "global"}
    assert per["frac/thread_idle"] == 0.0 and cum["frac/thread_idle"] == 0.0
    assert per["msec/sim.step.c_sim"] >= 2.5 and per["msec/sim.step"] >= per["msec/sim.step.c_sim"]
    assert per["msec/sim.thread_idle"] >= 1.5
    lap_sum = sum(per["msec/" + o] for o in ops)
    assert abs(per["active_frac/sim.step"] - per["msec/sim.step"] / lap_sum) < 1e-9
    assert 0 < cum["active_frac/sim.step"] <= 1
    # a second lap starts from zero; the cumulative figures keep their history
    s2 = tm.sections()
    assert s2["timing_per_epoch"]["msec/sim.step"] == 0.0
    assert s2["timing_cumulative"]["active_frac/sim.step"] > 0
