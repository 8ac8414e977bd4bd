#!/usr/bin/env python3
"""Derives the comparator network of the 5x5 median kernel and writes it as C++ statements.

Wires: w[5 r + k], r = rank inside window column k after that column has been sorted (the
kernel sorts image columns once and shares them between the windows that hold them).  What is
searched here is what follows: a sort of every row across the five columns and a selection of
the median among the 13 positions that can still hold it, pruned greedily to the comparators
the one output needs.  Every candidate network is checked on all 2^25 zero-one inputs
(bit-parallel: one 2^25-bit vector per wire, a comparator is an AND and an OR), which by the
zero-one principle proves it for all inputs.  Output: imageanalysis_amd/csrc/iak_median25_net.inc
usage: python tools/make_median25_net.py   (about a minute, 400 MB)"""
import os
import random

import numpy as np

N = 25
WORDS = (1 << N) // 64
S5 = [(0, 1), (3, 4), (2, 4), (2, 3), (1, 4), (0, 3), (0, 2), (1, 3), (1, 2)]  # sorts 5 wires


def input_wire(i):
    if i < 6:
        pat = 0
        for b in range(64):
            if (b >> i) & 1:
                pat |= 1 << b
        return np.full(WORDS, pat, dtype=np.uint64)
    idx = np.arange(WORDS, dtype=np.uint64)
    return np.where((idx >> np.uint64(i - 6)) & np.uint64(1) == 1, np.uint64(0xFFFFFFFFFFFFFFFF), np.uint64(0))


def majority():
    out = np.zeros(WORDS, dtype=np.uint64)
    chunk = 1 << 20
    for s in range(0, 1 << N, chunk):
        x = np.arange(s, s + chunk, dtype=np.uint32)
        pc = np.zeros(chunk, dtype=np.uint8)
        for _ in range(N):
            pc += (x & 1).astype(np.uint8)
            x >>= 1
        bits = (pc >= 13).reshape(-1, 64)
        w = np.zeros(chunk // 64, dtype=np.uint64)
        for b in range(64):
            w |= bits[:, b].astype(np.uint64) << np.uint64(b)
        out[s // 64:(s + chunk) // 64] = w
    return out


def batcher(n):
    p = 1
    while p < n:
        p *= 2
    ces = []

    def merge(lo, hi, r):
        step = r * 2
        if step < hi - lo:
            merge(lo, hi, step)
            merge(lo + r, hi, step)
            for i in range(lo + r, hi - r, step):
                ces.append((i, i + r))
        else:
            ces.append((lo, lo + r))

    def sort(lo, hi):
        if hi - lo >= 1:
            mid = lo + (hi - lo) // 2
            sort(lo, mid)
            sort(mid + 1, hi)
            merge(lo, hi, 1)

    sort(0, p - 1)
    return [(a, b) for a, b in ces if a < n and b < n]


def main():
    for x in range(32):  # S5 sorts
        v = [(x >> i) & 1 for i in range(5)]
        for a, b in S5:
            if v[a] > v[b]:
                v[a], v[b] = v[b], v[a]
        assert v == sorted(v)
    idx = lambda r, k: r * 5 + k
    truth = majority()
    start = [input_wire(i) for i in range(N)]
    for k in range(5):  # the shared column sorts
        for a, b in S5:
            lo, hi = start[idx(a, k)] & start[idx(b, k)], start[idx(a, k)] | start[idx(b, k)]
            start[idx(a, k)], start[idx(b, k)] = lo, hi
    # positions (r, k) that can hold the median once rows and columns are sorted: at most 12 elements
    # are known to be below and at most 12 above
    cand = [(r, k) for r in range(5) for k in range(5) if (r + 1) * (k + 1) - 1 <= 12 and (5 - r) * (5 - k) - 1 <= 12]
    assert len(cand) == 13
    cw = [idx(r, k) for r, k in cand]
    out = idx(2, 2)
    assert cw[6] == out

    def evaluate(net):
        v = list(start)
        for a, b in net:
            v[a], v[b] = v[a] & v[b], v[a] | v[b]
        return v[out]

    def needed(net):
        need, ops, kept = {out}, 0, []
        for a, b in reversed(net):
            na, nb = a in need, b in need
            if na or nb:
                ops += int(na) + int(nb)
                kept.append((a, b, na, nb))
                need.update((a, b))
        return ops, kept[::-1]

    def prune(net, seeds):
        best, best_ops = net, needed(net)[0]
        for seed in seeds:
            order = list(range(len(net)))
            if seed:
                random.Random(seed).shuffle(order)
            else:
                order.reverse()
            gone = set()
            for i in order:
                if np.array_equal(evaluate([c for j, c in enumerate(net) if j != i and j not in gone]), truth):
                    gone.add(i)
            cur = [c for j, c in enumerate(net) if j not in gone]
            if needed(cur)[0] < best_ops:
                best, best_ops = cur, needed(cur)[0]
        return best

    rows = [(idx(r, a), idx(r, b)) for r in range(5) for a, b in S5]
    select = [(cw[a], cw[b]) for a, b in batcher(13)]
    assert np.array_equal(evaluate(rows + select), truth)
    net = prune(rows + prune_select(rows, select, evaluate, truth), range(4))
    assert np.array_equal(evaluate(net), truth)
    ops, kept = needed(net)
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "imageanalysis_amd", "csrc", "iak_median25_net.inc")
    with open(path, "w") as f:
        f.write("// Written by tools/make_median25_net.py: do not edit.  %d comparators, %d min/max operations;\n" % (len(kept), ops))
        f.write("// checked on all 2^25 zero-one inputs.  w[5 r + k]: rank r of window column k; the median ends in w[12].\n")
        for a, b, na, nb in kept:
            if na and nb:
                f.write("M25_CE(%d, %d)\n" % (a, b))
            elif na:
                f.write("M25_MIN(%d, %d)\n" % (a, b))
            else:
                f.write("M25_MAX(%d, %d)\n" % (a, b))
    print("comparators", len(kept), "operations", ops, "->", os.path.normpath(path))


def prune_select(rows, select, evaluate, truth):
    """the selection stage pruned first, on its own (the search over the whole network is then shorter)"""
    best = select
    for seed in range(6):
        order = list(range(len(select)))
        if seed:
            random.Random(seed).shuffle(order)
        else:
            order.reverse()
        gone = set()
        for i in order:
            if np.array_equal(evaluate(rows + [c for j, c in enumerate(select) if j != i and j not in gone]), truth):
                gone.add(i)
        cur = [c for j, c in enumerate(select) if j not in gone]
        if len(cur) < len(best):
            best = cur
    return best


if __name__ == "__main__":
    main()
