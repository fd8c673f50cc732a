"""
``IntSet`` -- CPython's ``set`` of non-negative ints with the table kept natively (``csrc/pyset.cpp`` in libs3topo.so).

The reference keeps its cell bookkeeping in Python sets and numbers new cells in the iteration order of those sets
(s_cube.py:531-555, 601-621, 865-902).  ``IntSet`` goes through the same table states as the interpreter's set does for
the same sequence of operations (slot-for-slot, ``tests/test_pyset.py``), so iterating it yields the same order, but a
whole batch of ids moves in one call and a 10^7-entry set costs 8 bytes per slot instead of a Python object graph.
Only the operations ``SamplingTree`` uses are offered.
"""
import ctypes as C

import numpy as np

from . import _lib


def _check(rc):
    if rc == -1:
        raise MemoryError("IntSet: out of host memory")
    if rc == -2:
        raise ValueError("IntSet holds non-negative integers only")


def _ids(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class IntSet:
    __slots__ = ("_lib", "_h")

    def __init__(self, items=None):
        self._lib = _lib.topo_lib()
        self._h = C.c_void_p(self._lib.s3set_create())
        if not self._h.value:
            raise MemoryError("IntSet: out of host memory")
        if items is not None:
            self.update(items)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.s3set_destroy(h)
            self._h = None

    # -- construction / bulk updates (each mirrors one CPython code path) ---------------------------------------
    def update(self, items):
        """``s.update(x)``: ``x`` an IntSet (set_merge), a ``range`` with step 1, or a sequence / array of ids (one
        insertion per element, in order)"""
        if isinstance(items, IntSet):
            _check(self._lib.s3set_update_set(self._h, items._h))
        elif isinstance(items, range) and items.step == 1:
            _check(self._lib.s3set_update_range(self._h, items.start, items.stop))
        else:
            a = _ids(items if isinstance(items, np.ndarray) else list(items))
            _check(self._lib.s3set_update_ids(self._h, a.ctypes.data_as(C.c_void_p), len(a)))
        return self

    def update_flagged(self, ids, flags):
        """``s.update(i for i, f in zip(ids, flags) if f and i)`` (s_cube.py:709: id 0 and None are filtered out)"""
        a, f = _ids(ids), np.ascontiguousarray(flags, dtype=np.uint8)
        if len(a) != len(f):
            raise ValueError("ids and flags differ in length")
        _check(self._lib.s3set_update_flagged(self._h, a.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), len(a)))
        return self

    def __ior__(self, other):
        return self.update(other)

    def __isub__(self, other):
        """``s -= t`` (set_difference_update_internal), ``t`` an IntSet"""
        if not isinstance(other, IntSet):
            other = IntSet(other)
        _check(self._lib.s3set_difference_update(self._h, other._h))
        return self

    def copy(self):
        return IntSet(self)

    def add(self, key):
        _check(self._lib.s3set_add(self._h, int(key)))

    def discard(self, key):
        self._lib.s3set_discard(self._h, int(key))

    # -- queries ----------------------------------------------------------------------------------------------
    def __len__(self):
        return self._lib.s3set_len(self._h)

    def __bool__(self):
        return len(self) > 0

    def __contains__(self, key):
        return bool(self._lib.s3set_contains(self._h, int(key)))

    def to_array(self):
        """the elements in iteration order, int64"""
        out = np.empty(len(self), dtype=np.int64)
        self._lib.s3set_to_array(self._h, out.ctypes.data_as(C.c_void_p))
        return out

    def __iter__(self):
        return iter(self.to_array().tolist())

    def __eq__(self, other):
        if isinstance(other, IntSet):
            return len(self) == len(other) and np.array_equal(np.sort(self.to_array()), np.sort(other.to_array()))
        if isinstance(other, (set, frozenset)):
            return set(self) == other
        return NotImplemented

    __hash__ = None

    def issubset(self, other):
        return all(i in other for i in self)

    def table(self):
        """(mask, fill, slots): the raw table, -1 = unused, -2 = dummy (tests)"""
        mask = self._lib.s3set_mask(self._h)
        ptr = self._lib.s3set_table(self._h)
        buf = (C.c_int64 * (mask + 1)).from_address(ptr)
        return mask, self._lib.s3set_fill(self._h), np.frombuffer(buf, dtype=np.int64).copy()

    def __repr__(self):
        return f"IntSet({self.to_array().tolist()!r})"
