"""Emulation of libstdc++-11 `std::unordered_map` *iteration order*.

The reference's results depend on it: the order of inventory tokens in observations
(`SRC/objects/agent.cpp:201-205`), of protocol input/output tokens (`CPP/objects/assembler.hpp:553-564`),
the order in which a protocol's resources are consumed/produced, and the f32 summation order of
`stat_rewards` (`SRC/objects/agent.cpp:110-124`).  SURVEY.md Appendix B gives the verified rules
(`/usr/include/c++/11/bits/hashtable.h:1888-1912,2380-2412`, `hashtable_policy.h:576-598`):

* a node inserted into an empty bucket goes to the FRONT of the global list, otherwise directly before the
  first node of its bucket;
* growth: first insert into a default-constructed map -> 13 buckets, then 29, 59, ...; rehash re-inserts the
  nodes front-to-back with the same rule;
* a map converted from a Python dict by pybind11 is `clear(); reserve(len); emplace(k) in dict order`.

`hash<uint8_t/int>` is the identity; `hash<std::string>` is `_Hash_bytes` (64-bit Murmur-style,
seed 0xc70f6907, libstdc++ `hash_bytes.cc`), restated in `std_hash_string`.
"""

from __future__ import annotations

from typing import Callable, Hashable, Iterable

_M64 = (1 << 64) - 1

# libstdc++ __prime_list (prefix; ample for maps keyed by uint8 ids or a handful of strings)
_PRIMES = [
    2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113,
    127, 137, 139, 149, 157, 167, 179, 193, 199, 211, 227, 241, 257, 277, 293, 313, 337, 359, 383, 409, 439, 467,
    503, 541, 577, 619, 661, 709, 761, 823, 887, 953, 1031, 1109, 1193, 1289, 1381, 1493, 1613, 1741, 1879, 2029,
]  # fmt: skip
_FAST_BKT = [2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13]


def std_hash_string(s: str | bytes) -> int:
    """libstdc++ `std::hash<std::string>` on x86-64: `_Hash_bytes(ptr, len, 0xc70f6907)`."""
    b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
    mul = ((0xC6A4A793 << 32) + 0x5BD1E995) & _M64
    n = len(b)
    h = (0xC70F6907 ^ ((n * mul) & _M64)) & _M64

    def mix(v: int) -> int:
        return v ^ (v >> 47)

    aligned = n & ~7
    for i in range(0, aligned, 8):
        k = int.from_bytes(b[i : i + 8], "little")
        data = (mix((k * mul) & _M64) * mul) & _M64
        h ^= data
        h = (h * mul) & _M64
    if n & 7:
        data = int.from_bytes(b[aligned:], "little")
        h ^= data
        h = (h * mul) & _M64
    h = (mix(h) * mul) & _M64
    h = mix(h)
    return h


def _identity(k) -> int:
    return int(k)


class UMap:
    """Keys only (values never influence order).  `order` is the iteration order, front = begin()."""

    def __init__(self, hash_fn: Callable[[Hashable], int] = _identity):
        self.hash_fn = hash_fn
        self.order: list = []
        self.nb = 1
        self.next_resize = 0

    # -- rehash policy ---------------------------------------------------------------------------------
    def _next_bkt(self, n: int) -> int:
        if n < len(_FAST_BKT):
            if n == 0:
                return 1
            r = _FAST_BKT[n]
            self.next_resize = r
            return r
        for p in _PRIMES[6:]:
            if p >= n:
                self.next_resize = p
                return p
        raise ValueError("unordered_map emulation: table too large")

    def _insert_node(self, k) -> None:
        b = self.hash_fn(k) % self.nb
        for i, x in enumerate(self.order):
            if self.hash_fn(x) % self.nb == b:
                self.order.insert(i, k)
                return
        self.order.insert(0, k)

    def _rehash(self, nb: int) -> None:
        old, self.order, self.nb = self.order, [], nb
        for k in old:
            self._insert_node(k)

    # -- public operations -----------------------------------------------------------------------------
    def reserve(self, n: int) -> None:
        saved = self.next_resize
        want = max(len(self.order) + 1, n)  # max(_M_bkt_for_elements(size + 1), n)
        want = self._next_bkt(want)
        if want != self.nb:
            self._rehash(want)
        else:
            self.next_resize = saved

    def emplace(self, k) -> bool:
        if k in self.order:
            return False
        n_elt = len(self.order)
        if n_elt + 1 > self.next_resize:
            min_bkts = max(n_elt + 1, 0 if self.next_resize else 11)
            if min_bkts >= self.nb:
                self._rehash(self._next_bkt(max(min_bkts + 1, self.nb * 2)))
            else:
                self.next_resize = self.nb
        self._insert_node(k)
        return True

    def erase(self, k) -> None:
        if k in self.order:
            self.order.remove(k)


def order_from_pydict(keys: Iterable, hash_fn: Callable[[Hashable], int] = _identity) -> list:
    """Iteration order of a C++ map argument converted by pybind11 from a dict with these keys."""
    keys = list(keys)
    m = UMap(hash_fn)
    m.reserve(len(keys))
    for k in keys:
        m.emplace(k)
    return list(m.order)


def order_from_inserts(keys: Iterable, hash_fn: Callable[[Hashable], int] = _identity) -> list:
    """Iteration order of a default-constructed map after `m[k] = ...` for each key in turn."""
    m = UMap(hash_fn)
    for k in keys:
        m.emplace(k)
    return list(m.order)
