"""Generates cuda-flow2d_amd/csrc/median7_pair_network.inc: a comparator program that takes EIGHT sorted groups of seven
values (the sorted 7-tuples of eight consecutive image rows) and produces the medians of groups 0..6 and of groups 1..7,
i.e. the 7x7 medians of two vertically adjacent pixels (the window-7 sibling of tools/gen_median_network.py).

Construction: Batcher odd-even merges.  The six shared groups are merged once (1+2, 3+4, 5+6, then (1..4), then all 42), the
sorted 42 are merged with group 0 for the first output and (as a copy) with group 7 for the second.  Pruning: comparators
whose outputs cannot reach either median are dropped (backward liveness), then every remaining comparator is removed on
trial and the program re-verified.  Verification is exhaustive over the 0-1 inputs with sorted groups (8^8 = 16 777 216
cases, evaluated bit-parallel: min = AND, max = OR on packed bits): by the 0-1 principle for selection (thresholding commutes
with min / max and keeps sorted groups sorted) that proves the program for all inputs.  A random-float check runs on top.

Round 6: the comparator program is then lowered to one-result instructions (v_min / v_max / v_min3 / v_max3 / v_med3) and shortened by
tools/median_select3.py under the same exhaustive check, like the window-5 program.

usage: python tools/gen_median7_network.py        (about half an hour; MEDIAN7_STEPS=n bounds the shortening walk)
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


SORT7 = [(0, 6), (2, 3), (4, 5), (0, 2), (1, 4), (3, 6), (0, 1), (2, 5), (3, 4), (1, 2), (4, 6), (2, 3), (4, 5), (1, 2), (3, 4), (5, 6)]


def case_vectors():
    """all 0-1 inputs with sorted groups as uint64 words, in a fixed random order (a prefix of the bits is a fair sample)"""
    n = (R + 1) ** GROUPS
    c = np.random.default_rng(1).permutation(n).astype(np.int64)
    zeros = [(c // (R + 1) ** g) % (R + 1) for g in range(GROUPS)]
    pack = lambda b: np.packbits(b, bitorder="little").view(np.uint64)  # (n is a multiple of 64)
    inv = [pack(j >= zeros[g]) for g in range(GROUPS) for j in range(R)]
    want = [pack(sum(zeros[0:R]) <= MEDIAN), pack(sum(zeros[1:R + 1]) <= MEDIAN)]
    return inv, want


def main():
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import median_select3 as S

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
    n = sum(1 for o in prog if o[0] == "x")
    print("%d comparators for two medians (merge tree before pruning: %d)" % (n, full), flush=True)
    del tests
    # comparators -> one-result instructions (min / max / min3 / max3 / med3), shortened (tools/median_select3.py; candidates from a
    # node's fan-in cone of depth 4: the program is four times the window-5 one and its value vectors are 2 MB each)
    inv, want = case_vectors()
    steps = int(os.environ.get("MEDIAN7_STEPS", "12000"))
    ops, outs = S.lower(prog, WIRES, COPY_BASE, inv, [wa, wb], want, steps=steps, seed=7, cone_depth=4, log=lambda m: print(m, flush=True))
    sort_ops, sort_outs = S.sorter(SORT7, R, steps=4000, seed=1, log=lambda m: print("sort7:", m, flush=True))
    rng = random.Random(7)
    for _ in range(5000):
        groups = [sorted(rng.choice([rng.random(), float(rng.randint(0, 3))]) for _ in range(R)) for _ in range(GROUPS)]
        v = [x for grp in groups for x in grp]
        r = S.evaluate(ops, COPY_BASE, v)
        assert r[outs[0]] == sorted(v[0:R * R])[MEDIAN] and r[outs[1]] == sorted(v[R:R * R + R])[MEDIAN]
        t = [rng.choice([rng.random(), float(rng.randint(0, 2))]) for _ in range(R)]
        r = S.evaluate(sort_ops, R, t)
        assert [r[o] for o in sort_outs] == sorted(t)
    out = [
        "// Generated by tools/gen_median7_network.py (comparator networks lowered by tools/median_select3.py) -- do not edit.",
        "// One-result selection programs, {kind, a, b, c} as in median5_pair_network.inc.  Pair program: inputs 7g..7g+6 = ascending",
        "// 7-tuple of row g (g = 0..7); node kMedian7PairOutA = median of rows 0..6, node kMedian7PairOutB = median of rows 1..7.",
        "// %d instructions (%s) for the %d comparators of the merge network; verified over the 8^8 sorted 0-1 inputs." % (len(ops), S.histogram(ops), n),
        "constexpr int kMedian7PairInputs = %d;" % COPY_BASE,
        "constexpr int kMedian7PairOps = %d;" % len(ops),
        "constexpr int kMedian7PairOutA = %d, kMedian7PairOutB = %d;" % (outs[0], outs[1]),
        "constexpr SelectOp kMedian7PairProgram[kMedian7PairOps] = {%s};" % S.emit(ops, ""),
        "// Sorter of a row's seven values (x-3 .. x+3): %d instructions (%s) for a 16-comparator network; node kSort7Out[k] = k-th smallest." % (len(sort_ops), S.histogram(sort_ops)),
        "constexpr int kSort7Ops = %d;" % len(sort_ops),
        "constexpr int kSort7Out[7] = {%s};" % ", ".join(str(o) for o in sort_outs),
        "constexpr SelectOp kSort7Program[kSort7Ops] = {%s};" % S.emit(sort_ops, ""),
    ]
    dst = os.environ.get("MEDIAN7_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow2d_amd", "csrc",
                                                        "median7_pair_network.inc")
    open(dst, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
