"""Executable specification (plain Python integers mod r) of the SUBTREE sharding of a HyperKZG opening:
what each rank holds, what it computes locally and the few field elements that cross ranks.  tests/test_subtree_model.py checks every
step against the same step on the global polynomial; jolt_amd/csrc/hyperkzg.hip (jolt_host_hyperkzg_open_subtree) mirrors it.

Ownership.  G = 2^gamma ranks, coefficient index i in [0, 2^ell).  Read i in binary: the gamma bits below its leading one name the
owner.  Rank g's terms, in index order, are its COMPACT array:
    slot 0       <-> index g                      (the "crown": indices below G, one per rank)
    slot c >= 1  <-> index insert(c, g) = c with the gamma bits of g inserted below c's leading one
so that (i) every prefix [0, n) of the indices is a prefix of every rank's compact array -- one compact SRS and one set of window
tables per rank serve all levels; (ii) LowToHigh folding is local: insert(2c, g) = 2 insert(c, g), insert(2c+1, g) = 2 insert(c, g) + 1
for c >= 1 -- a rank's subtree folds into itself, level after level; only the crown (indices < G) pairs values of different ranks.
"""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def insert(c, g, gamma):
    if c == 0:
        return g
    lp = c.bit_length() - 1
    return (1 << (lp + gamma)) | (g << lp) | (c - (1 << lp))


def slot_of(i, gamma):
    """global index -> (owner, compact slot)"""
    G = 1 << gamma
    if i < G:
        return i, 0
    lp = i.bit_length() - 1 - gamma
    return (i >> lp) & (G - 1), (1 << lp) | (i & ((1 << lp) - 1))


def owned(n, g, gamma):
    """how many of the indices [0, n) rank g owns = the length of its compact prefix"""
    lo, hi = 0, n + 1  # insert is increasing in c
    while lo < hi:
        mid = (lo + hi) // 2
        if insert(mid, g, gamma) < n:
            lo = mid + 1
        else:
            hi = mid
    return lo


def to_compact(poly, g, gamma):
    return [poly[insert(c, g, gamma)] for c in range(owned(len(poly), g, gamma))]


def fold_levels_global(poly, point):
    """HyperKZGScheme fold_polynomials (scheme.rs:88-114): level i from level i-1 with point[ell - i]"""
    ell = len(point)
    levels = [list(poly)]
    for i in range(1, ell):
        x, prev = point[ell - i], levels[-1]
        levels.append([(prev[2 * y] + x * (prev[2 * y + 1] - prev[2 * y])) % R for y in range(len(prev) // 2)])
    return levels


def fold_levels_sharded(compacts, point, gamma):
    """compacts[g] = rank g's compact level-0 array.  Returns levels[g][k] = rank g's compact array of level k (all ell levels).
    Local part: the plain LowToHigh fold of the compact array (slot 0 of the result is garbage).  Exchange: every rank publishes slot 1
    of each level (its subtree's root, index G + g) and slot 0 of level 0; the crowns of all levels follow from those on every rank."""
    G, ell = 1 << gamma, len(point)
    lam = ell - gamma
    levels = []
    for g in range(G):
        mine = [list(compacts[g])]
        for i in range(1, lam):  # levels whose compact array has >= 2 slots
            x, prev = point[ell - i], mine[-1]
            mine.append([(prev[2 * c] + x * (prev[2 * c + 1] - prev[2 * c])) % R for c in range(len(prev) // 2)])
        levels.append(mine)
    # the exchange: roots[k][g] = levels[g][k][1] for k < lam, crown0[g] = levels[g][0][0]
    roots = [[levels[g][k][1] for g in range(G)] for k in range(lam)]
    crown = [levels[g][0][0] for g in range(G)]
    crowns = [crown]
    for k in range(1, ell):
        x = point[ell - k]
        prev = crowns[-1] + (roots[k - 1] if k - 1 < lam else [])  # indices [0, 2G) of level k-1 (or all of it once it is <= G long)
        crowns.append([(prev[2 * y] + x * (prev[2 * y + 1] - prev[2 * y])) % R for y in range(len(prev) // 2)])
    for g in range(G):
        for k in range(1, ell):
            if k < lam:
                levels[g][k][0] = crowns[k][g]
            else:  # levels of at most G coefficients: crown only
                levels[g].append([crowns[k][g]] if g < len(crowns[k]) else [])
    return levels


def evaluate_sharded(level_compacts, u, gamma):
    """P(u) = sum_i P[i] u^i from the ranks' compact arrays: slot 0 weighs u^g, segment [2^L, 2^(L+1)) of the compact array is a
    plain Horner sum times u^(2^(L+gamma) + g 2^L)"""
    total = 0
    for g, a in enumerate(level_compacts):
        if not a:
            continue
        part = a[0] * pow(u, g, R)
        L = 0
        while (1 << L) < len(a):
            seg = a[1 << L: 1 << (L + 1)]
            h = 0
            for x in reversed(seg):
                h = (h * u + x) % R
            part += h * pow(u, (1 << (L + gamma)) + (g << L), R)
            L += 1
        total += part
    return total % R


def witness_global(f, u):
    """compute_witness_polynomial (kzg.rs:34-46): h[k] = s[k+1], s[k] = f[k] + u s[k+1]"""
    s = [0] * (len(f) + 1)
    for k in range(len(f) - 1, -1, -1):
        s[k] = (f[k] + u * s[k + 1]) % R
    return s[1:len(f)]


def witness_sharded(f_compacts, u, gamma, n):
    """f_compacts[g]: compact arrays of f (n coefficients).  Every rank's segments are contiguous index ranges; a segment [a, b) needs
    the carry E(b) = sum_{i >= b} f[i] u^(i - b) from the segments above it.  Exchange: every rank publishes the Horner sum of each of
    its segments (and its crown value); the chain of carries is then evaluated identically on every rank.  Returns the compact arrays
    of h (n - 1 coefficients: the owner of index n - 1 holds one slot less)."""
    G = 1 << gamma
    segs = []  # (start index, length, rank, first slot, Horner sum)
    for g, a in enumerate(f_compacts):
        if a:
            segs.append((insert(0, g, gamma), 1, g, 0, a[0] % R))
        L = 0
        while (1 << L) < len(a):
            seg = a[1 << L: 1 << (L + 1)]
            h = 0
            for x in reversed(seg):
                h = (h * u + x) % R
            segs.append((insert(1 << L, g, gamma), len(seg), g, 1 << L, h))
            L += 1
    segs.sort()
    carry_in = {}
    e = 0  # E(n) = 0
    for start, length, g, slot, h in reversed(segs):
        carry_in[(g, slot)] = e
        e = (h + pow(u, length, R) * e) % R
    out = []
    for g, a in enumerate(f_compacts):
        h = [0] * len(a)
        if a:
            h[0] = carry_in[(g, 0)]  # h[g] = s[g + 1] = E(g + 1)
        L = 0
        while (1 << L) < len(a):
            lo, hi = 1 << L, min(1 << (L + 1), len(a))
            acc = carry_in[(g, lo)]
            h[hi - 1] = acc  # the segment's top entry is the carry itself
            for c in range(hi - 1, lo, -1):
                acc = (a[c] + u * acc) % R
                h[c - 1] = acc
            L += 1
        out.append(h[:owned(n - 1, g, gamma)])
    return out
