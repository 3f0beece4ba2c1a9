"""Generates cuda-flow2d_amd/csrc/median7_pair_network.inc: a comparator program that takes EIGHT sorted groups of seven
values (the sorted 7-tuples of eight consecutive image rows) and produces the medians of groups 0..6 and of groups 1..7,
i.e. the 7x7 medians of two vertically adjacent pixels (the window-7 sibling of tools/gen_median_network.py).

Construction: Batcher odd-even merges.  The six shared groups are merged once (1+2, 3+4, 5+6, then (1..4), then all 42), the
sorted 42 are merged with group 0 for the first output and (as a copy) with group 7 for the second.  Pruning: comparators
whose outputs cannot reach either median are dropped (backward liveness), then every remaining comparator is removed on
trial and the program re-verified.  Verification is exhaustive over the 0-1 inputs with sorted groups (8^8 = 16 777 216
cases, evaluated bit-parallel: min = AND, max = OR on packed bits): by the 0-1 principle for selection (thresholding commutes
with min / max and keeps sorted groups sorted) that proves the program for all inputs.  A random-float check runs on top.

usage: python tools/gen_median7_network.py        (a few minutes)
"""
import os
import random

import numpy as np

R = 7
GROUPS = R + 1
SHARED = (R - 1) * R      # 42
COPY_BASE = GROUPS * R    # 56
WIRES = COPY_BASE + SHARED
MEDIAN = (R * R) // 2     # 24


def oe_merge(a, b):
    """Batcher odd-even merge of two sorted wire lists -> (comparators, output wire order)."""
    if not a:
        return [], list(b)
    if not b:
        return [], list(a)
    if len(a) == 1 and len(b) == 1:
        return [(a[0], b[0])], [a[0], b[0]]
    c1, v = oe_merge(a[0::2], b[0::2])
    c2, w = oe_merge(a[1::2], b[1::2])
    comps = c1 + c2
    out = [v[0]]
    i = 0
    while i < len(w) or i + 1 < len(v):
        if i < len(w) and i + 1 < len(v):
            comps.append((w[i], v[i + 1]))
            out += [w[i], v[i + 1]]
        elif i < len(w):
            out.append(w[i])
        else:
            out.append(v[i + 1])
        i += 1
    return comps, out


def build():
    g = lambda k: [k * R + i for i in range(R)]
    prog = []

    def merge(a, b):
        c, out = oe_merge(a, b)
        prog.extend(("x", p, q) for p, q in c)
        return out

    s12, s34, s56 = merge(g(1), g(2)), merge(g(3), g(4)), merge(g(5), g(6))
    shared = merge(merge(s12, s34), s56)
    copy = [COPY_BASE + i for i in range(len(shared))]
    prog.extend(("c", shared[i], copy[i]) for i in range(len(shared)))
    out_a = merge(shared, g(0))
    out_b = merge(copy, g(GROUPS - 1))
    return prog, out_a[MEDIAN], out_b[MEDIAN]


def inputs01():
    """bit-packed 0-1 inputs: case c = (k_0 .. k_7) in base 8, group g holds k_g zeros followed by 7 - k_g ones"""
    n = (R + 1) ** GROUPS
    c = np.arange(n, dtype=np.int64)
    wires = [None] * WIRES
    zeros = []
    for g in range(GROUPS):
        k = (c // (R + 1) ** g) % (R + 1)
        zeros.append(k)
        for j in range(R):
            wires[g * R + j] = np.packbits(j >= k)
    blank = np.zeros_like(wires[0])
    for i in range(COPY_BASE, WIRES):
        wires[i] = blank
    want_a = np.packbits(sum(zeros[0:R]) <= MEDIAN)       # element 24 of the ascending 49 is a one iff at most 24 zeros
    want_b = np.packbits(sum(zeros[1:R + 1]) <= MEDIAN)
    return wires, want_a, want_b


def run01(prog, wires):
    v = list(wires)
    for kind, a, b in prog:
        if kind == "c":
            v[b] = v[a]
        else:
            v[a], v[b] = v[a] & v[b], v[a] | v[b]
    return v


def ok(prog, wa, wb, tests):
    wires, want_a, want_b = tests
    v = run01(prog, wires)
    return np.array_equal(v[wa], want_a) and np.array_equal(v[wb], want_b)


def prune_dead(prog, outs):
    live = set(outs)
    keep = []
    for op in reversed(prog):
        kind, a, b = op
        if kind == "c":
            if b in live:
                keep.append(op)
                live.discard(b)
                live.add(a)
        elif a in live or b in live:
            keep.append(op)
            live.add(a)
            live.add(b)
    return keep[::-1]


def run(prog, v):
    v = list(v)
    for kind, a, b in prog:
        if kind == "c":
            v[b] = v[a]
        elif v[a] > v[b]:
            v[a], v[b] = v[b], v[a]
    return v


def main():
    prog, wa, wb = build()
    tests = inputs01()
    assert ok(prog, wa, wb, tests)
    full = sum(1 for o in prog if o[0] == "x")
    prog = prune_dead(prog, [wa, wb])
    print("merge tree %d comparators, %d after the liveness pass" % (full, sum(1 for o in prog if o[0] == "x")), flush=True)
    i = len(prog) - 1
    while i >= 0:
        if prog[i][0] == "x":
            trial = prog[:i] + prog[i + 1:]
            if ok(trial, wa, wb, tests):
                prog = trial
        i -= 1
    prog = prune_dead(prog, [wa, wb])
    assert ok(prog, wa, wb, tests)
    rng = random.Random(7)
    for _ in range(5000):
        groups = [sorted(rng.choice([rng.random(), float(rng.randint(0, 3))]) for _ in range(R)) for _ in range(GROUPS)]
        v = [x for grp in groups for x in grp] + [0.0] * SHARED
        r = run(prog, v)
        assert r[wa] == sorted(v[0:R * R])[MEDIAN] and r[wb] == sorted(v[R:R * R + R])[MEDIAN]
    n = sum(1 for o in prog if o[0] == "x")
    print("%d comparators for two medians (merge tree before pruning: %d)" % (n, full))
    body = ", ".join("{%d, %d, %d}" % (1 if k == "c" else 0, a, b) for k, a, b in prog)
    out = [
        "// Generated by tools/gen_median7_network.py -- do not edit.",
        "// Program on %d wires: wires 7g..7g+6 = ascending 7-tuple of row g (g = 0..7), wires %d.. = scratch." % (WIRES, COPY_BASE),
        "// {0, a, b}: (wire a, wire b) <- (min, max);  {1, a, b}: wire b <- wire a.",
        "// Afterwards wire kMedian7PairOutA holds the median of rows 0..6, wire kMedian7PairOutB that of rows 1..7.",
        "constexpr int kMedian7PairWires = %d;" % WIRES,
        "constexpr int kMedian7PairOps = %d;  // %d comparators" % (len(prog), n),
        "constexpr int kMedian7PairOutA = %d, kMedian7PairOutB = %d;" % (wa, wb),
        "constexpr MedianPairOp kMedian7PairProgram[kMedian7PairOps] = {%s};" % body,
    ]
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow2d_amd", "csrc",
                       "median7_pair_network.inc")
    open(dst, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
