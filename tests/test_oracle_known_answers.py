"""Known-answer tests that pin the oracle's restatements of libstdc++ and of reference helpers.

* std::shuffle + std::mt19937 known answers: SURVEY.md Appendix A (verified there against g++ 11.4).
* ObservationPattern == offsets sorted by (manhattan, dr, dc): reference tests/test_observations.cpp:16-39,84-104.
* HasInventory::shared_update: the 11 known-answer cases of reference tests/test_mettagrid.cpp:1408-1663.
* unordered_map iteration order examples: SURVEY.md §8a-I / Appendix B (oracle-verified there).
"""

import ctypes as ct

import numpy as np
import pytest

from metta_amd.umap import order_from_inserts, order_from_pydict, std_hash_string
from oracle import mg_oracle


def _shuffle(seed, n, rounds):
    L = mg_oracle.lib()
    out = np.zeros((rounds, n), np.int32)
    nxt = ct.c_uint32()
    L.mgo_test_shuffle(seed, n, rounds, out.ctypes.data_as(ct.POINTER(ct.c_int32)), ct.byref(nxt))
    return out, nxt.value


def test_shuffle_known_answers_n5():
    assert _shuffle(0, 5, 1)[0][0].tolist() == [2, 4, 3, 0, 1]
    assert _shuffle(42, 5, 1)[0][0].tolist() == [4, 0, 2, 3, 1]


def test_shuffle_known_answers_n24():
    o, nxt = _shuffle(0, 24, 3)
    assert o[0].tolist() == [0, 13, 1, 17, 9, 11, 4, 16, 22, 10, 3, 18, 21, 20, 7, 8, 19, 5, 15, 12, 23, 14, 6, 2]
    assert nxt == 3342157822
    o, nxt = _shuffle(42, 24, 3)
    assert o[0].tolist() == [1, 6, 22, 20, 5, 9, 21, 23, 18, 19, 13, 17, 2, 3, 10, 15, 7, 8, 14, 4, 0, 11, 16, 12]
    assert nxt == 1855189739


def test_shuffle_known_answers_n1024_fnv():
    def fnv(arr):
        h = 1469598103934665603
        for v in arr.reshape(-1).tolist():
            h ^= v
            h = (h * 1099511628211) & ((1 << 64) - 1)
        return h

    assert fnv(_shuffle(0, 1024, 3)[0]) == 2556913099609935663
    assert fnv(_shuffle(42, 1024, 3)[0]) == 6470427229661035971


def test_mt19937_matches_numpy_legacy_seeding():
    # std::mt19937(seed) is init_genrand(seed): same raw stream as numpy's RandomState(seed)
    for seed in (0, 1, 42, 2**31 + 5):
        _, nxt = _shuffle(seed, 0, 0)
        rs = np.random.RandomState(seed)
        first = int(rs.randint(0, 2**32, dtype=np.uint64))  # first tempered 32-bit output
        assert nxt == first


@pytest.mark.parametrize("hw", [(3, 9), (7, 3), (5, 5), (1, 1), (1, 5), (5, 1), (11, 11), (3, 5)])
def test_observation_pattern_matches_sorted_offsets(hw):
    h, w = hw
    L = mg_oracle.lib()
    out = np.zeros((h * w + 8, 2), np.int32)
    n = L.mgo_test_pattern(h, w, out.ctypes.data_as(ct.POINTER(ct.c_int32)))
    got = [tuple(x) for x in out[:n].tolist()]
    exp = [(dr, dc) for dr in range(-(h // 2), h // 2 + 1) for dc in range(-(w // 2), w // 2 + 1)]
    exp.sort(key=lambda o: (abs(o[0]) + abs(o[1]), o[0], o[1]))
    assert n == h * w and got == exp
    assert len(set(got)) == h * w


SHARED_UPDATE_CASES = [
    # (start amounts, limits, delta) -> (consumed, end amounts)       reference test name
    (([0, 0, 0], [100, 100, 100], 30), (30, [10, 10, 10])),  # PositiveDelta_EvenDistribution
    (([0, 0, 0], [100, 100, 100], 31), (31, [11, 10, 10])),  # PositiveDelta_UnevenDistribution
    (([5, 0, 0], [10, 10, 10], 30), (25, [10, 10, 10])),  # PositiveDelta_WithLimits
    (([20, 20, 20], [100, 100, 100], -30), (-30, [10, 10, 10])),  # NegativeDelta_EvenDistribution
    (([5, 20, 20], [100, 100, 100], -30), (-30, [0, 7, 8])),  # NegativeDelta_InsufficientResources
    (([20, 20, 20], [100, 100, 100], -31), (-31, [9, 10, 10])),  # NegativeDelta_UnevenDistribution
    (([], [], 10), (0, [])),  # EmptyInventoriesList
    (([0], [100], 25), (25, [25])),  # SingleInventory
    (([10, 10], [10, 10], 20), (0, [10, 10])),  # AllInventoriesAtLimit
    (([0, 0, 0], [10, 20, 30], 45), (45, [10, 18, 17])),  # MixedLimits
]


@pytest.mark.parametrize("case", SHARED_UPDATE_CASES)
def test_shared_update_known_answers(case):
    (amounts, limits, delta), (consumed, end) = case
    L = mg_oracle.lib()
    n = len(amounts)
    a = np.asarray(amounts + [0], np.int32)
    lim = np.asarray(limits + [0], np.int32)
    out = np.zeros(n + 1, np.int32)
    p = lambda x: x.ctypes.data_as(ct.POINTER(ct.c_int32))  # noqa: E731
    got = L.mgo_test_shared_update(n, p(a), p(lim), delta, p(out))
    assert got == consumed
    assert out[:n].tolist() == end


def test_unordered_map_order_examples():
    # set_inventory({8:5, 6:1, 1:1}) shows items in order 8, 6, 1; {0,3,8} in order 0, 3, 8:
    # the token order is the *inventory's* (default-constructed map) order after inserting in the converted
    # map's iteration order.
    assert order_from_inserts(order_from_pydict([8, 6, 1])) == [8, 6, 1]
    assert order_from_inserts(order_from_pydict([0, 3, 8])) == [0, 3, 8]
    # growth 13 -> 29 buckets reorders
    keys = list(range(14))
    assert order_from_inserts(keys) != list(reversed(keys))
    assert sorted(order_from_inserts(keys)) == keys
    # string hash is deterministic and spreads
    assert std_hash_string("heart.amount") != std_hash_string("heart.amounu")
    assert order_from_pydict(["a", "b"], std_hash_string) in (["a", "b"], ["b", "a"])


def test_unordered_map_order_vs_reference_when_available():
    from oracle.ref_bridge import have_reference, load_reference_module

    if not have_reference():
        pytest.skip("oracle/_ref not built")
    C = load_reference_module()
    rs = np.random.RandomState(0)
    for _ in range(400):
        n = int(rs.randint(0, 24))
        keys = [int(k) for k in rs.choice(60, size=n, replace=False)]
        p = C.Protocol()
        p.input_resources = {k: 1 for k in keys}
        assert list(p.input_resources.keys()) == order_from_pydict(keys)
    words = ["heart.amount", "ore_red.gained", "action.move.success", "laser.lost", "x", "a.b", "tokens_written"]
    for _ in range(200):
        ks = [words[i] for i in rs.choice(len(words), size=int(rs.randint(0, 7)), replace=False)]
        a = C.AgentConfig(type_id=0, type_name="agent", group_id=0, group_name="red", stat_rewards={k: 1.0 for k in ks})
        assert list(a.stat_rewards.keys()) == order_from_pydict(ks, std_hash_string)
