"""Generates cuda-flow2d_amd/csrc/median5_pair_network.inc: a comparator program that takes SIX sorted groups of
five values (the sorted 5-tuples of six consecutive image rows) and produces the medians of groups 0..4 and of
groups 1..5, i.e. the 5x5 medians of two vertically adjacent pixels.

Construction: Batcher odd-even merges.  The four shared groups are merged once (1+2, 3+4, then both), the
sorted 20 are then merged with group 0 for the first output and (as a copy) with group 5 for the second.
Pruning: comparators whose outputs cannot reach either median are dropped (backward liveness), then every
remaining comparator is removed on trial and the program re-verified.  Verification is exhaustive over the 0-1
inputs with sorted groups (6^6 = 46656): by the 0-1 principle for selection (thresholding commutes with
min/max and keeps sorted groups sorted) that proves the program for all inputs.  A random-float check runs
on top before the file is written.

Round 6: the comparator program is then lowered to one-result instructions -- v_min / v_max and the three-input v_min3 / v_max3 /
v_med3 -- and shortened by tools/median_select3.py (semantic rewrites verified on the same exhaustive case set): 81 comparators =
123 live two-input instructions -> 78, and the five-value sorter of a row 18 -> 13.

usage: python tools/gen_median_network.py        (takes a few minutes)
"""
import itertools
import os
import random

R = 5
COPY_BASE = 6 * R  # wires 30..49: the copy of the shared sorted 20


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


def run(prog, v):
    v = list(v)
    for kind, a, b in prog:
        if kind == "c":
            v[b] = v[a]
        elif v[a] > v[b]:
            v[a], v[b] = v[b], v[a]
    return v


def build():
    g = lambda k: [k * R + i for i in range(R)]
    prog = []
    c, s12 = oe_merge(g(1), g(2))
    prog += [("x", a, b) for a, b in c]
    c, s34 = oe_merge(g(3), g(4))
    prog += [("x", a, b) for a, b in c]
    c, shared = oe_merge(s12, s34)
    prog += [("x", a, b) for a, b in c]
    copy = [COPY_BASE + i for i in range(len(shared))]
    prog += [("c", shared[i], copy[i]) for i in range(len(shared))]
    c, out_a = oe_merge(shared, g(0))
    prog += [("x", a, b) for a, b in c]
    c, out_b = oe_merge(copy, g(5))
    prog += [("x", a, b) for a, b in c]
    return prog, out_a[12], out_b[12]


def tests01():
    tests = []
    for ones in itertools.product(range(R + 1), repeat=6):
        v = [0] * (COPY_BASE + 20)
        for g, k in enumerate(ones):
            for i in range(R - k, R):
                v[g * R + i] = 1
        want_a = 1 if 25 - sum(ones[0:5]) <= 12 else 0
        want_b = 1 if 25 - sum(ones[1:6]) <= 12 else 0
        tests.append((v, want_a, want_b))
    return tests


def ok(prog, wa, wb, tests):
    for v, want_a, want_b in tests:
        r = run(prog, v)
        if r[wa] != want_a or r[wb] != want_b:
            return False
    return True


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


SORT5 = [(0, 1), (3, 4), (2, 4), (2, 3), (0, 3), (0, 2), (1, 4), (1, 3), (1, 2)]  # 9 comparators


def case_vectors():
    """all 0-1 inputs with sorted groups, in a fixed random order (a prefix of the bits is a fair sample: median_select3's filter)"""
    import numpy as np

    import median_select3 as S

    cases = np.array(list(itertools.product(range(R + 1), repeat=6)))
    cases = cases[np.random.default_rng(1).permutation(len(cases))]
    inv = [S.pack_cases(i >= R - cases[:, g]) for g in range(6) for i in range(R)]
    want = [S.pack_cases(25 - cases[:, 0:5].sum(axis=1) <= 12), S.pack_cases(25 - cases[:, 1:6].sum(axis=1) <= 12)]
    return inv, want


def main():
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import median_select3 as S

    prog, wa, wb = build()
    tests = tests01()
    assert ok(prog, wa, wb, tests)
    full = sum(1 for o in prog if o[0] == "x")
    prog = prune_dead(prog, [wa, wb])
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
    # comparators -> one-result instructions (min / max / min3 / max3 / med3), shortened (tools/median_select3.py)
    inv, want = case_vectors()
    ops, outs = S.lower(prog, COPY_BASE + 20, 6 * R, inv, [wa, wb], want, steps=9000, seed=15, log=lambda m: print(m, flush=True))
    sort_ops, sort_outs = S.sorter(SORT5, R, steps=3000, seed=1, log=lambda m: print("sort5:", m, flush=True))
    rng = random.Random(5)
    for _ in range(20000):
        groups = [sorted(rng.choice([rng.random(), float(rng.randint(0, 3))]) for _ in range(R)) for _ in range(6)]
        v = [x for grp in groups for x in grp]
        r = S.evaluate(ops, 6 * R, v)
        assert r[outs[0]] == sorted(v[0:25])[12] and r[outs[1]] == sorted(v[5:30])[12]
        t = [rng.choice([rng.random(), float(rng.randint(0, 2))]) for _ in range(R)]
        r = S.evaluate(sort_ops, R, t)
        assert [r[o] for o in sort_outs] == sorted(t)
    out = [
        "// Generated by tools/gen_median_network.py (comparator networks lowered by tools/median_select3.py) -- do not edit.",
        "// One-result selection programs: {kind, a, b, c} defines node (inputs + index) = kind(node a, node b, node c);",
        "// kind 0 min, 1 max (two inputs), 2 min3, 3 max3, 4 med3.  Verified over all 0-1 inputs with sorted groups.",
        "// Pair program: inputs 5g..5g+4 = ascending 5-tuple of row g (g = 0..5); node kMedianPairOutA = median of rows 0..4,",
        "// node kMedianPairOutB = median of rows 1..5.  %d instructions (%s) for the %d comparators of the merge network." % (len(ops), S.histogram(ops), n),
        "constexpr int kMedianPairInputs = %d;" % (6 * R),
        "constexpr int kMedianPairOps = %d;" % len(ops),
        "constexpr int kMedianPairOutA = %d, kMedianPairOutB = %d;" % (outs[0], outs[1]),
        "constexpr SelectOp kMedianPairProgram[kMedianPairOps] = {%s};" % S.emit(ops, ""),
        "// Sorter of a row's five values (x-2 .. x+2): %d instructions (%s) for a 9-comparator network; node kSort5Out[k] = k-th smallest." % (len(sort_ops), S.histogram(sort_ops)),
        "constexpr int kSort5Ops = %d;" % len(sort_ops),
        "constexpr int kSort5Out[5] = {%s};" % ", ".join(str(o) for o in sort_outs),
        "constexpr SelectOp kSort5Program[kSort5Ops] = {%s};" % S.emit(sort_ops, ""),
    ]
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-flow2d_amd", "csrc",
                       "median5_pair_network.inc")
    open(dst, "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
