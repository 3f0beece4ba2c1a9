"""Lowers a verified comparator program (tools/gen_median_network.py, tools/gen_median7_network.py) to a program of ONE-RESULT
selection instructions of gfx950 -- v_min_f32 / v_max_f32 and the three-input v_min3_f32 / v_max3_f32 / v_med3_f32 -- and shortens it.

A comparator is two instructions (min and max); after dead-code elimination many comparators keep one half only, chains of one-sided
comparators are three-input minima / maxima, and wherever the order of two operands is KNOWN -- the inputs are sorted tuples, and a
merge network carries a lot of order along -- `min(max(x, a), b)` is the median of three.  Instead of pattern rules the pass is
semantic: every node of the program carries its value on ALL the 0-1 inputs with sorted groups (one big integer, a bit per case:
min = AND, max = OR, med3 = majority), and a node may be replaced by any one-instruction function of up to three earlier nodes
that has the same bits; a seeded random walk over such rewrites (equal-length ones included) keeps the shortest program it meets.  By the 0-1 principle for selection (thresholding commutes with min, max and the median of
three, and keeps sorted groups sorted) equality on those cases is equality on all inputs the kernel can present.  The result is
re-verified from scratch (fresh evaluation of the emitted program against the wanted outputs) before it is written, and with random
floats on top.

Not a reference restatement (the reference sorts each window by insertion, src/kernels/median_2d.cu:52-63): selection is order-free
for windows without NaN and -0, and those go through exact_median() in csrc/median.hip as before."""
import itertools

MIN, MAX, MIN3, MAX3, MED3 = 0, 1, 2, 3, 4
NAMES = {MIN: "min", MAX: "max", MIN3: "min3", MAX3: "max3", MED3: "med3"}


def ev(kind, a, b, c=None):
    if kind == MIN:
        return a & b
    if kind == MAX:
        return a | b
    if kind == MIN3:
        return a & b & c
    if kind == MAX3:
        return a | b | c
    return (a & b) | (a & c) | (b & c)


class Dag:
    """nodes 0 .. n_in-1 are the inputs; ops[i] = (kind, (operands...)) defines node n_in + i"""

    def __init__(self, n_in, in_values):
        self.n_in = n_in
        self.ops = []
        self.val = list(in_values)

    def add(self, kind, args):
        self.ops.append((kind, tuple(args)))
        self.val.append(ev(kind, *[self.val[a] for a in args]))
        return self.n_in + len(self.ops) - 1

    def args(self, n):
        return self.ops[n - self.n_in][1] if n >= self.n_in else ()

    def live(self, outs):
        seen = set()
        stack = list(outs)
        while stack:
            n = stack.pop()
            if n in seen:
                continue
            seen.add(n)
            stack.extend(self.args(n))
        return seen

    def cost(self, outs):
        return sum(1 for n in self.live(outs) if n >= self.n_in)


def from_comparators(prog, n_wires, n_in, in_values, outs):
    """SSA form of a comparator program ('x', a, b): (a, b) <- (min, max); ('c', a, b): b <- a"""
    d = Dag(n_in, in_values)
    cur = list(range(n_in)) + [None] * (n_wires - n_in)
    for kind, a, b in prog:
        if kind == "c":
            cur[b] = cur[a]
        else:
            lo, hi = d.add(MIN, (cur[a], cur[b])), d.add(MAX, (cur[a], cur[b]))
            cur[a], cur[b] = lo, hi
    return d, [cur[o] for o in outs]


def cone(d, n, depth):
    """ancestors of n within `depth` levels (n itself excluded)"""
    out, frontier = set(), {n}
    for _ in range(depth):
        nxt = set()
        for m in frontier:
            nxt.update(d.args(m))
        out |= nxt
        frontier = nxt
    return out


def candidates(d, outs, n, cone_depth=0, filter_words=4):
    """every one-instruction function of up to three live nodes defined before n (cone_depth > 0: of n's fan-in cone of that depth only --
    large programs) that equals node n on all cases; d.val are numpy uint64 arrays"""
    import numpy as np

    live = d.live(outs)
    pool = sorted(m for m in (cone(d, n, cone_depth) if cone_depth else live) if m < n and m in live)
    if len(pool) < 2:
        return []
    pool = np.array(pool, dtype=np.int64)
    P = np.stack([d.val[m][:filter_words] for m in pool])
    t = d.val[n][:filter_words]
    cands = []
    i2, j2 = np.triu_indices(len(pool), 1)
    A, B = P[i2], P[j2]
    for kind, V in ((MIN, A & B), (MAX, A | B)):
        for k in np.nonzero((V == t).all(axis=1))[0]:
            cands.append((kind, (int(pool[i2[k]]), int(pool[j2[k]]))))
    m = len(pool)
    for a in range(m - 2):
        rest = np.arange(a + 1, m)
        bi, ci = np.triu_indices(len(rest), 1)
        b, c = rest[bi], rest[ci]
        Va, Vb, Vc = P[a], P[b], P[c]
        for kind, V in ((MIN3, Va & Vb & Vc), (MAX3, Va | Vb | Vc), (MED3, (Va & Vb) | (Va & Vc) | (Vb & Vc))):
            for k in np.nonzero((V == t).all(axis=1))[0]:
                cands.append((kind, (int(pool[a]), int(pool[b[k]]), int(pool[c[k]]))))
    want = d.val[n]
    return [c for c in cands if np.array_equal(ev(c[0], *[d.val[x] for x in c[1]]), want)]


def anneal(d, outs, steps, seed, accept_equal=0.5, cone_depth=0, log=None):
    """Random walk over equal-value rewrites: pick a live node, replace it by another one-instruction function of earlier nodes with the
    same value when the live count does not grow (equal counts with probability accept_equal: the plateau moves are what lets the
    greedy result drop from ~92 to ~78 instructions for the window-5 pair program).  Keeps the best program seen; deterministic for a seed."""
    import random

    rnd = random.Random(seed)
    base = d.cost(outs)
    best, best_ops = base, list(d.ops)
    for step in range(steps):
        live = sorted(x for x in d.live(outs) if x >= d.n_in)
        n = rnd.choice(live)
        old = d.ops[n - d.n_in]
        cs = [c for c in candidates(d, outs, n, cone_depth) if c != old]
        rnd.shuffle(cs)
        pick, pc = None, None
        for c in cs[:40]:
            d.ops[n - d.n_in] = c
            cost = d.cost(outs)
            d.ops[n - d.n_in] = old
            if cost < base or (cost == base and rnd.random() < accept_equal):
                if pick is None or cost < pc:
                    pick, pc = c, cost
        if pick:
            d.ops[n - d.n_in] = pick
            base = pc
            if base < best:
                best, best_ops = base, list(d.ops)
                if log:
                    log("step %d: %d instructions" % (step, best))
    d.ops = best_ops
    return best


def pack_cases(bits):
    """bool array over the cases -> uint64 words (little-endian bit order, zero padded)"""
    import numpy as np

    bits = np.concatenate([bits, np.zeros((-len(bits)) % 64, bool)])
    return np.packbits(bits, bitorder="little").view(np.uint64)


def lower(prog, n_wires, n_in, in_values, outs, want, steps, seed, cone_depth=0, log=None):
    """comparator program -> shortened one-result program: (ops, output nodes), checked on a fresh evaluation against `want`"""
    import numpy as np

    d, o = from_comparators(prog, n_wires, n_in, in_values, outs)
    before = d.cost(o)
    anneal(d, o, steps, seed, cone_depth=cone_depth, log=log)
    ops, oo = compact(d, o)
    v = evaluate(ops, n_in, in_values)
    for node, w in zip(oo, want):
        assert np.array_equal(v[node], w), "the shortened program differs from the comparator program"
    if log:
        log("%d two-input instructions -> %d (%s)" % (before, len(ops), histogram(ops)))
    return ops, oo


def sorter(comparators, n, steps, seed, log=None):
    """a sorting network on n unordered inputs -> shortened program whose outputs are the n order statistics, ascending"""
    import itertools

    import numpy as np

    cases = np.array(list(itertools.product((0, 1), repeat=n)))
    inv = [pack_cases(cases[:, i] == 1) for i in range(n)]
    want = [pack_cases(cases.sum(axis=1) >= n - k) for k in range(n)]
    return lower([("x", a, b) for a, b in comparators], n, n, inv, list(range(n)), want, steps, seed, log=log)


def compact(d, outs):
    """live ops only, renumbered in order: (ops, outs) with inputs 0 .. n_in-1"""
    live = d.live(outs)
    remap = {i: i for i in range(d.n_in)}
    ops = []
    for i, (kind, args) in enumerate(d.ops):
        n = d.n_in + i
        if n in live:
            remap[n] = d.n_in + len(ops)
            ops.append((kind, tuple(remap[a] for a in args)))
    return ops, [remap[o] for o in outs]


def evaluate(ops, n_in, in_values):
    """fresh evaluation (bit vectors or numbers): the list of all node values"""
    v = list(in_values)
    for kind, args in ops:
        x = [v[a] for a in args]
        if isinstance(x[0], float):
            if kind == MIN:
                v.append(min(x))
            elif kind == MAX:
                v.append(max(x))
            elif kind == MIN3:
                v.append(min(x))
            elif kind == MAX3:
                v.append(max(x))
            else:
                v.append(sorted(x)[1])
        else:
            v.append(ev(kind, *x))
    return v


def histogram(ops):
    h = {}
    for kind, _ in ops:
        h[NAMES[kind]] = h.get(NAMES[kind], 0) + 1
    return ", ".join("%d %s" % (h[k], k) for k in ("min", "max", "min3", "max3", "med3") if k in h)


def emit(ops, prefix):
    """C++ table: {kind, a, b, c} per instruction (c = a for the two-input kinds)"""
    return ", ".join("{%d, %d, %d, %d}" % (k, a[0], a[1], a[2] if len(a) > 2 else a[0]) for k, a in ops)
