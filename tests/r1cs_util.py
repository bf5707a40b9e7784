"""Test helper: seeded random satisfiable R1CS instances (pure Python ints)."""
import numpy as np
from oracle import pyoracle as o

R = o.R_MOD

def random_r1cs(seed, n_inputs, n_vars, n_cons, max_terms=4, small_frac=0.7):
    """Returns (R1CS, z) with z a (n_vars,4) uint64 array.  Variables 1..n_vars (0 = ONE).  Coefficients are a mix of
    +-1 / small / full-width, like the circuits' (mostly +-1 and powers of two)."""
    g = o.SplitMix64(seed)
    def coeff():
        s = g.next() % 100
        if s < 40: return 1
        if s < 60: return R - 1
        if s < int(100 * small_frac) + 20: return 1 << (g.next() % 64)
        return g.field()
    z = [1] + [0] * n_vars
    defined = 0
    rows = [[], [], []]
    def rand_lc(limit):
        k = 1 + g.next() % max_terms; terms = {}
        for _ in range(k):
            terms[g.next() % (limit + 1)] = coeff()
        return sorted(terms.items())
    def ev(lc): return sum(c * z[i] for i, c in lc) % R
    for i in range(n_cons):
        if defined < n_vars and (i < n_vars):
            # inputs and early variables: some get free values, others are defined by the constraint
            if defined < n_inputs or g.next() % 4 == 0:
                defined += 1; z[defined] = g.field() if g.next() % 2 else g.next() % 2
            a = rand_lc(defined); b = rand_lc(defined); val = ev(a) * ev(b) % R
            if defined < n_vars:
                defined += 1; z[defined] = val; c = [(defined, 1)]
            else:
                c = [(0, val)]
        else:
            while defined < n_vars:
                defined += 1; z[defined] = g.field()
            a = rand_lc(n_vars); b = rand_lc(n_vars); val = ev(a) * ev(b) % R
            j = 1 + g.next() % n_vars
            c = [(j, val * pow(z[j], R - 2, R) % R)] if z[j] else [(0, val)]
        for m, lc in enumerate((a, b, c)): rows[m].append(lc)
    while defined < n_vars:
        defined += 1; z[defined] = g.field()
    rp, col, co = [], [], []
    for m in range(3):
        ptr = [0]; cc = []; vv = []
        for lc in rows[m]:
            for i, c in lc: cc.append(i); vv.append(c)
            ptr.append(len(cc))
        rp.append(np.array(ptr, dtype=np.uint32)); col.append(np.array(cc, dtype=np.uint32)); co.append(o.to_arr(vv) if vv else np.zeros((0, 4), dtype=np.uint64))
    cs = o.R1CS(n_inputs, n_vars, n_cons, rp, col, co)
    return cs, o.to_arr(z[1:])
