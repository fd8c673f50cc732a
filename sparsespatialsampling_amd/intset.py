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
    """``deferred=True``: ``update(RangeSet)`` and ``difference_update_ids`` are queued for the set's native worker thread
    (applied in the order issued; every other operation on the set, and its use as the argument of another set's update,
    waits for them inside the library).  ``SamplingTree`` keeps its leaf set this way: merging a batch of new cells into
    a table of 10^5 .. 10^7 entries is cache misses and the occasional rebuild, and nothing the refine loop does next --
    the geometry kernels, the download of their flags -- needs the result.  A failure of a deferred update (out of memory)
    is raised by the next operation that waits.  The two deferred operations DECLARE their effect on the length -- the ids
    of a ``RangeSet`` are new to the set, the ids of ``difference_update_ids`` are distinct members of it (new cells /
    the leaves just refined / new cells just found invalid) -- so ``len()`` answers from that bookkeeping without waiting
    and the worker may run several iterations behind; when the queue is drained the real length is compared with the
    declared one and a mismatch raises."""
    __slots__ = ("_lib", "_h", "_deferred", "_declared", "_order")

    def __init__(self, items=None, deferred=False):
        self._lib = _lib.topo_lib()
        self._deferred = bool(deferred)
        self._declared = None              # length after the queued updates, while any are outstanding
        self._order = None                 # to_array() of the current state (read-only), dropped by every update
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
        self._order = None
        if isinstance(items, RangeSet) and self._deferred:
            self._declared = len(self) + len(items)
            _check(self._lib.s3set_update_rangeset_async(self._h, items.start, items.stop))
            return self
        self._settle()
        if isinstance(items, IntSet):
            items._settle()
            _check(self._lib.s3set_update_set(self._h, items._h))
        elif isinstance(items, RangeSet):
            _check(self._lib.s3set_update_rangeset(self._h, items.start, items.stop))
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
        self._settle()
        self._order = None
        _check(self._lib.s3set_update_flagged(self._h, a.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p), len(a)))
        return self

    def __ior__(self, other):
        return self.update(other)

    def __isub__(self, other):
        """``s -= t`` (set_difference_update_internal), ``t`` an IntSet"""
        if isinstance(other, RangeSet):
            other = other.materialise()
        if not isinstance(other, IntSet):
            other = IntSet(other)
        self._settle()
        other._settle()
        self._order = None
        _check(self._lib.s3set_difference_update(self._h, other._h))
        return self

    def difference_update_ids(self, ids):
        """``s -= set(ids)`` for distinct ids without building that set: discards do not depend on their order, and the
        rebuild rule of set_difference_update_internal looks at the table once, at the end"""
        a = _ids(ids)
        self._order = None
        if self._deferred:
            self._declared = len(self) - len(a)
            _check(self._lib.s3set_difference_update_ids_async(self._h, a.ctypes.data_as(C.c_void_p), len(a)))   # (copies the ids)
        else:
            _check(self._lib.s3set_difference_update_ids(self._h, a.ctypes.data_as(C.c_void_p), len(a)))
        return self

    def _settle(self):
        """wait for the queued updates and check what they declared"""
        if self._declared is None:
            return
        declared, self._declared = self._declared, None
        _check(self._lib.s3set_wait(self._h))
        if self._lib.s3set_len(self._h) != declared:
            raise RuntimeError(f"IntSet: the deferred updates left {self._lib.s3set_len(self._h)} elements, {declared} were "
                               f"declared (ids that were not new / not members)")

    def copy(self):
        return IntSet(self)

    def add(self, key):
        self._settle()
        self._order = None
        _check(self._lib.s3set_add(self._h, int(key)))

    def discard(self, key):
        self._settle()
        self._order = None
        self._lib.s3set_discard(self._h, int(key))

    # -- queries ----------------------------------------------------------------------------------------------
    def __len__(self):
        if self._declared is not None:
            return self._declared
        return self._lib.s3set_len(self._h)

    def __bool__(self):
        return len(self) > 0

    def __contains__(self, key):
        self._settle()
        return bool(self._lib.s3set_contains(self._h, int(key)))

    def to_array(self):
        """the elements in iteration order, int64 (read-only: the array is kept until the set changes -- the tree asks for the
        order of its 10^7 leaves three times in a row at the end of a refine, each a scan of a 256 MB table)"""
        self._settle()
        if self._order is None:
            out = np.empty(len(self), dtype=np.int64)
            self._lib.s3set_to_array(self._h, out.ctypes.data_as(C.c_void_p))
            out.setflags(write=False)
            self._order = out
        return self._order

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
        self._settle()
        mask = self._lib.s3set_mask(self._h)
        ptr = self._lib.s3set_table(self._h)
        buf = (C.c_int64 * (mask + 1)).from_address(ptr)
        return mask, self._lib.s3set_fill(self._h), np.frombuffer(buf, dtype=np.int64).copy()

    def __repr__(self):
        return f"IntSet({self.to_array().tolist()!r})"



class RangeSet:
    """``s = set(); s.update(range(start, stop))`` without the table: consecutive ids never collide, id ``i`` sits in slot
    ``i & mask`` of a table whose size follows from the rebuilds of ``stop - start`` single insertions, so the iteration order
    is the part of the range behind the last multiple of the table size followed by the part before it.  The tree's
    ``all_children`` of a batch (s_cube.py:531-555, 879-900) is such a set: it is iterated (geometry check) and merged
    into the leaf set, never modified.  ``IntSet.update(RangeSet)`` replays CPython's set_merge against the virtual table;
    anything else goes through ``materialise()``."""
    __slots__ = ("start", "stop", "_set")

    def __init__(self, start, stop):
        if start < 0:
            raise ValueError("IntSet holds non-negative integers only")
        self.start, self.stop, self._set = int(start), max(int(stop), int(start)), None

    def __len__(self):
        return self.stop - self.start

    def __bool__(self):
        return self.stop > self.start

    def __contains__(self, key):
        return self.start <= int(key) < self.stop

    @property
    def mask(self):
        return _lib.topo_lib().s3set_range_mask(len(self))

    def to_array(self):
        wrap = (self.start | self.mask) + 1
        if wrap < self.stop:
            return np.concatenate([np.arange(wrap, self.stop, dtype=np.int64), np.arange(self.start, wrap, dtype=np.int64)])
        return np.arange(self.start, self.stop, dtype=np.int64)

    def __iter__(self):
        return iter(self.to_array().tolist())

    def materialise(self):
        if self._set is None:
            self._set = IntSet(range(self.start, self.stop))
        return self._set

    def __eq__(self, other):
        return self.materialise() == other

    __hash__ = None

    def __repr__(self):
        return f"RangeSet({self.start}, {self.stop})"
