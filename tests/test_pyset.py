"""
The native restatement of CPython's set (csrc/pyset.cpp, ``IntSet``) against the interpreter's own set objects, slot by
slot: the PySetObject of a real set is read through ctypes (layout: /usr/include/python3.x/setobject.h) after every
operation of random traces.  Iteration order follows from the table, so equal tables mean equal cell numbering in
``SamplingTree`` (reference s_cube.py:531-555, 601-621, 865-902).
"""
import ctypes as C
import random

import numpy as np
import pytest

from sparsespatialsampling_amd.intset import IntSet


class _Entry(C.Structure):
    _fields_ = [("key", C.c_void_p), ("hash", C.c_ssize_t)]


class _PySetObject(C.Structure):
    _fields_ = [("ob_refcnt", C.c_ssize_t), ("ob_type", C.c_void_p), ("fill", C.c_ssize_t), ("used", C.c_ssize_t),
                ("mask", C.c_ssize_t), ("table", C.POINTER(_Entry))]


def real_table(s):
    """(mask, fill, slots) of a real set of non-negative ints: -1 unused, -2 dummy, else the key (== its hash)"""
    o = _PySetObject.from_address(id(s))
    raw = np.frombuffer((C.c_int64 * (2 * (o.mask + 1))).from_address(C.addressof(o.table.contents)), dtype=np.int64)
    key, hsh = raw[0::2], raw[1::2]
    slots = np.where(key == 0, -1, np.where(hsh == -1, -2, hsh))
    return o.mask, o.fill, slots


def same(py, nat):
    m, f, t = real_table(py)
    nm, nf, nt = nat.table()
    return m == nm and f == nf and np.array_equal(t, nt) and list(py) == nat.to_array().tolist() and len(py) == len(nat)


def test_real_table_reader_sees_hash_order():
    s = set([5000, 37, 1200, 64, 129, 9, 4097])
    assert list(s) == [64, 129, 4097, 37, 5000, 9, 1200]            # SURVEY.md section 7, hard part 1
    _, _, t = real_table(s)
    assert [int(v) for v in t if v >= 0] == list(s)


@pytest.mark.parametrize("seed", range(12))
def test_random_traces_slot_for_slot(seed):
    rng = random.Random(seed)
    py, nat = set(), IntSet()
    hi = rng.choice([40, 300, 5000, 200_000])
    for step in range(rng.choice([300, 1500])):
        op = rng.random()
        if op < 0.30:
            x = rng.randrange(hi)
            py.add(x); nat.add(x)
        elif op < 0.45:
            x = rng.randrange(hi)
            py.discard(x); nat.discard(x)
        elif op < 0.60:
            xs = [rng.randrange(hi) for _ in range(rng.randrange(0, 400))]
            py.update(xs); nat.update(xs)
        elif op < 0.70:
            a = rng.randrange(hi); b = a + rng.randrange(0, 3000)
            py.update(range(a, b)); nat.update(range(a, b))
        elif op < 0.82:
            xs = [rng.randrange(hi) for _ in range(rng.randrange(0, 600))]
            o_py, o_nat = set(), IntSet()
            o_py.update(xs); o_nat.update(xs)
            assert same(o_py, o_nat)
            py.update(o_py); nat.update(o_nat)
        elif op < 0.95:
            pool = list(py)
            xs = rng.sample(pool, min(len(pool), rng.randrange(0, 500))) + [rng.randrange(hi) for _ in range(rng.randrange(0, 50))]
            o_py, o_nat = set(), IntSet()
            o_py.update(xs); o_nat.update(xs)
            py -= o_py; nat -= o_nat
        else:
            py, nat = set(py), nat.copy()
        assert same(py, nat), (seed, step)
        if rng.random() < 0.1:
            x = rng.randrange(hi)
            assert (x in py) == (x in nat)


def test_refine_like_trace_large():
    """the set traffic of one SamplingTree run (uniform levels, then batches of parents replaced by 8 children, some
    children removed again), 10^5-10^6 ids: tables stay identical through every growth / rebuild step"""
    rng = np.random.default_rng(0)
    leaf_py, leaf_nat = set(), IntSet()
    leaf_py.add(0); leaf_nat.add(0)
    n_cells = 1
    for it in range(40):
        order = list(leaf_py)
        assert order == leaf_nat.to_array().tolist()
        if it >= 5:                                        # adaptive: a few hundred / thousand parents by "gain"
            pick = rng.permutation(len(order))[: min(4000, max(1, len(order) // 20))]
            sel = [order[i] for i in pick]
            tr_py, tr_nat = set(), IntSet()
            tr_py.update(sel); tr_nat.update(np.asarray(sel))
            order = list(tr_py)
            assert order == tr_nat.to_array().tolist()
        first, n_new = n_cells, 8 * len(order)
        par_py, par_nat = set(), IntSet()
        par_py.update(order); par_nat.update(order)
        ch_py, ch_nat = set(), IntSet()
        ch_py.update(range(first, first + n_new)); ch_nat.update(range(first, first + n_new))
        leaf_py -= par_py; leaf_nat -= par_nat
        leaf_py.update(ch_py); leaf_nat.update(ch_nat)
        n_cells += n_new
        new_py, new_nat = set(range(first, first + n_new)), IntSet(range(first, first + n_new))
        ids = np.fromiter(new_py, dtype=np.int64)
        assert np.array_equal(ids, new_nat.to_array())
        flags = rng.random(len(ids)) < 0.03
        bad_py = set(i for i in ids[flags].tolist() if i)
        bad_nat = IntSet().update_flagged(ids, flags)
        assert same(bad_py, bad_nat)
        leaf_py -= bad_py; leaf_nat -= bad_nat
        assert same(leaf_py, leaf_nat), it
    assert len(leaf_py) > 500_000


def test_errors_and_views():
    s = IntSet([3, 1, 2])
    assert 2 in s and 7 not in s and -1 not in s and len(s) == 3 and s == {1, 2, 3}
    with pytest.raises(ValueError):
        s.add(-4)
    with pytest.raises(ValueError):
        s.update([1, -2])
    t = s.copy()
    t -= s
    assert len(t) == 0 and not t and s.issubset(IntSet(range(0, 10)))



@pytest.mark.parametrize("n", [0, 1, 4, 5, 6, 19, 20, 77, 1229, 13107, 13108, 50000, 52429, 52430, 209715, 209716, 400000])
def test_range_sets_without_their_table(n):
    """set(); s.update(range(a, a + n)) in closed form (IntSet.update(range) on a new set, RangeSet): table, mask, order"""
    from sparsespatialsampling_amd.intset import RangeSet
    for a in (0, 1, 7, 4096, 123457, 1 << 22):
        py = set()
        py.update(range(a, a + n))
        nat = IntSet()
        nat.update(range(a, a + n))
        assert same(py, nat), (a, n)
        r = RangeSet(a, a + n)
        m, _, _ = real_table(py)
        assert (n == 0 or r.mask == m) and r.to_array().tolist() == list(py) and len(r) == n
        assert (a in r) == (a in py) and (a + n in r) == (a + n in py)


@pytest.mark.parametrize("deferred", [False, True])
@pytest.mark.parametrize("seed", range(8))
def test_refine_loop_trace_with_virtual_sets(seed, deferred):
    """the refine loop's bookkeeping (leaf -= all_parents; leaf |= all_children; leaf -= invalid) with all_parents replaced
    by its id array (IntSet.difference_update_ids) and all_children by a RangeSet, against real Python sets slot by slot;
    deferred: the two bulk updates applied by the set's worker thread"""
    from sparsespatialsampling_amd.intset import RangeSet
    rng = np.random.default_rng(seed)
    py, nat = set(), IntSet(deferred=deferred)
    py.add(0); nat.add(0)
    first, nch = 1, (4 if seed % 2 else 8)
    for it in range(30):
        leaves = np.fromiter(py, dtype=np.int64, count=len(py))
        if it < 3:
            picked = leaves
        else:
            picked = rng.choice(leaves, min(max(1, len(leaves) // 6), int(rng.choice([40, 900, 7000]))), replace=False)
        to_refine_py, to_refine_nat = set(), IntSet()
        to_refine_py.update(picked.tolist()); to_refine_nat.update(picked)
        order = to_refine_nat.to_array()
        assert order.tolist() == list(to_refine_py)
        n_new = nch * len(order)
        all_parents, all_children = set(), set()
        all_parents.update(order.tolist())
        all_children.update(range(first, first + n_new))
        py -= all_parents
        py.update(all_children)
        children = RangeSet(first, first + n_new)
        assert children.to_array().tolist() == list(all_children)
        nat.difference_update_ids(order)
        nat.update(children)
        assert same(py, nat), (seed, it)
        new_ids = children.to_array()
        bad = new_ids[rng.random(n_new) < (0.15 if it % 4 == 0 else 0.0)]
        bad_py = set(i for i in bad.tolist() if i)
        py -= bad_py
        nat -= IntSet().update_flagged(new_ids, np.isin(new_ids, bad))
        assert same(py, nat), (seed, it)
        first += n_new
    other = IntSet()
    other.update(RangeSet(10, 500))                 # empty target: the slot-wise copy / clean insertion paths of set_merge
    ref = set()
    ref.update(set(range(10, 500)))
    assert same(ref, other)
    nat -= RangeSet(first - 5, first)
    py -= set(range(first - 5, first))
    assert same(py, nat)


def test_deferred_sets_end_their_thread_and_serve_as_sources():
    import threading
    from sparsespatialsampling_amd.intset import RangeSet
    before = threading.active_count()                  # (native threads are not counted: this only checks nothing leaks here)
    for _ in range(50):
        s = IntSet(deferred=True)
        s.update(RangeSet(5, 70000))
        s.difference_update_ids(np.arange(100, 60000))
        t = IntSet(s)                                  # a set with queued updates as the source of another set's update
        ref = set()
        ref.update(set(range(5, 70000)))
        ref -= set(range(100, 60000))
        assert same(ref, s) and sorted(t.to_array().tolist()) == sorted(ref)
        assert (99 in s) and (100 not in s) and len(s) == len(ref)
        del s, t
    assert threading.active_count() == before


def test_deferred_updates_declare_their_effect_on_the_length():
    """len() of a deferred set answers from the declared effect of the queued updates; when the queue is drained a declaration
    that did not hold (ids that were not new / not members) raises"""
    from sparsespatialsampling_amd.intset import RangeSet
    s = IntSet(deferred=True)
    s.update(RangeSet(0, 1000))
    s.difference_update_ids(np.arange(10, 20))
    assert len(s) == 990 and s.to_array().size == 990
    s.update(RangeSet(500, 600))                       # not new
    assert len(s) == 1090
    with pytest.raises(RuntimeError):
        s.to_array()
    assert len(s) == 990                               # the set itself is what CPython's would be
    s.difference_update_ids(np.arange(10, 20))         # not members
    with pytest.raises(RuntimeError):
        5 in s
    assert len(s) == 990
