#!/usr/bin/env python3
"""Generates field29_gfx950.inc: Fq arithmetic on NINE 29-bit limbs (Montgomery radix R' = 2^261) for the inner loop of the H query's bucket accumulation.

Why: measured on MI355X (tools/valu_probe.hip), v_addc_co_u32 / v_subb_co_u32 issue at the rate of v_mad_u64_u32 (32 T lane-ops/s against 61 T for v_add_u32): the 128
carry additions of a product on 8 x 32-bit limbs (field_mul_gfx950.inc) cost as much as its 128 multiplications, and every modular difference is 25 such instructions.
With 29-bit limbs a column of the product — at most 9 + 9 terms below 2^60 — fits a 64-bit accumulator with NO carry tracking: 162 v_mad_u64_u32 and no v_addc per product
(squaring: 45 + 81), and sums / differences are nine independent 32-bit operations (limbs may run over; `norm` brings them back with one parallel carry step).

Representation: value v (mod p) is kept as ANY integer x = v * 2^261 (mod p), 0 <= x < 2^261, in limbs l[0..8], x = sum l[i] 2^(29 i).  "Normalized" = limbs 0..7 below
2^29 + 8 (the top limb holds the rest).  Products take normalized operands (one of them may have limbs up to 2^31.4) and return exactly-29-bit limbs and a value below
(a b) / 2^261 + p.  A difference a - b is computed as a + K_c - b with K_c = c p written with every low limb in [3 * 2^29 + 64, 4 * 2^29) ("borrow adjusted"), so that
no limb goes negative when b's limbs are below 3 * (2^29 + 8) and b's value is below c p.  The value bounds of the mixed addition's intermediate results are checked
below by interval arithmetic (BOUNDS); they stay under 11 p < 2^258.
"""
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
B = 29; M = (1 << B) - 1; NL = 9; RP = 1 << (B * NL)   # R' = 2^261
def limbs29(x, n=NL): return [(x >> (B * i)) & M for i in range(n - 1)] + [x >> (B * (n - 1))]
def adjusted(c):
    """c*p with limbs 0..7 in [3*2^29 + 64, 4*2^29): borrow 3 or 4 units from the limb above"""
    l = limbs29(c * Q); lo = 3 * (1 << B) + 64
    for i in range(NL - 1):
        n = 3 if l[i] + 3 * (1 << B) >= lo else 4
        l[i] += n << B; l[i + 1] -= n
        assert lo <= l[i] < 4 << B, (c, i, l[i])
    assert l[NL - 1] > 0 and sum(v << (B * i) for i, v in enumerate(l)) == c * Q
    return l
def adjusted_light(c):
    """c*p with limbs 0..7 in [2^29 + 64, 2 * 2^29 + 64): a + KL - t has no negative limb for t with exact 29-bit limbs (a product's result) and value below c p"""
    l = limbs29(c * Q); lo = (1 << B) + 64
    for i in range(NL - 1):
        n = 1 if l[i] + (1 << B) >= lo else 2
        l[i] += n << B; l[i + 1] -= n
        assert lo <= l[i] < lo + (1 << B), (c, i, l[i])
    assert l[NL - 1] > 0 and sum(v << (B * i) for i, v in enumerate(l)) == c * Q
    return l
INV = (-pow(Q, -1, 1 << B)) % (1 << B)
P29 = limbs29(Q); ONE29 = limbs29(RP % Q); RCONV = limbs29((1 << 256) % Q)
KS = {c: adjusted(c) for c in (2, 4, 6, 12, 18)}   # (12 p and 18 p: the Fq2 mixed addition of the G2 witness MSM, msm.cuh: XYZZ2_29)

# ---- bounds of the mixed addition (values in units of p; RATIO = 2^261 / p) ------------------------------------------------------------------------
RATIO = RP / Q
def prod(a, b): return a * b / RATIO + 1.0
def check_bounds():
    bX, bY, bZZ, bZZZ = 5.5, 3.6, 1.1, 1.1                      # the invariant: X < 5.5 p, Y < 3.6 p, ZZ, ZZZ < 1.1 p  (it holds for the lifted point: x < p, y' < 2p, one = 2^261 mod p < p)
    U2 = prod(1.0, bZZ); S2 = prod(2.0, bZZZ)                   # y' = K_2 - y < 2p
    cP, cR, cX, cT, cY = 6, 4, 4, 6, 2
    assert cP >= bX and cR >= bY
    bP = U2 + cP; bR = S2 + cR; PP = prod(bP, bP); PPP = prod(bP, PP); Qv = prod(bX, PP); RR = prod(bR, bR)
    assert cX >= PPP + 2 * Qv + 0.01; nX = RR + cX
    assert cT >= nX + 0.01; bT = Qv + cT
    A = prod(bR, bT); Bv = prod(bY, PPP); assert cY >= Bv + 0.01; nY = A + cY
    nZZ = prod(bZZ, PP); nZZZ = prod(bZZZ, PPP)
    assert nX <= bX and nY <= bY and nZZ <= bZZ and nZZZ <= bZZZ, (nX, nY, nZZ, nZZZ)
    top = max(bP, bR, bT, nX, nY) ; assert top * Q < 1 << 258
    # column bound of a product: 9 terms (limb < 2^31.4) x (limb < 2^29 + 8), 9 terms m*p below 2^58, carry below 2^35
    assert 9 * int(2 ** 31.4) * ((1 << 29) + 8) + 9 * (1 << 58) + (1 << 35) < 1 << 64
    # top limb of a subtrahend b < c p never exceeds K_c's top limb:  (c p) >> 232 minus at most 4
    for c, need in ((cP, bX), (cR, bY), (cX, PPP + 2 * Qv), (cT, nX), (cY, Bv), (2, 1.0)):
        assert KS[c][NL - 1] >= int(need * Q) >> (B * (NL - 1)), (c, need)
    return dict(cP=cP, cR=cR, cX=cX, cT=cT, cY=cY, bounds=dict(X=nX, Y=nY, ZZ=nZZ, ZZZ=nZZZ, P=bP, R=bR, T=bT))
BOUNDS = check_bounds()
def check_bounds_dual_y():
    """round 5: the mixed addition's Y3 = R (Q - X3) - Y PPP as ONE dual product mul2(R, Q - X3, K_4 - Y, PPP) (msm.cuh: XYZZ29::madd_tail_pp): the negated Y needs
    Y < 4 p (the invariant: 3.6 p) with normalized limbs, all four operands are normalized (the column bound of mul2 with 18 + 9 terms below 2^58 + ...), and the
    result — a product's, below ((R)(Q - X3) + (4)(PPP)) / 2^261 + p — stays within the invariant with room to spare"""
    b = BOUNDS["bounds"]; PPP = prod(b["P"], prod(b["P"], b["P"])); nY = (b["R"] * b["T"] + 4.0 * PPP) / RATIO + 1.0
    assert 3.6 + 0.01 <= 4 and KS[4][NL - 1] >= int(3.6 * Q) >> (B * (NL - 1)) and nY <= 3.6, nY
    assert 18 * ((1 << 29) + 8) ** 2 + 9 * (1 << 58) + (1 << 35) < 1 << 64
    return nY
BOUNDS_DUAL_Y = check_bounds_dual_y()
def check_bounds_ntt_stages():
    """the transform tiles (ntt.cuh: ntt29_lds_pass) normalize after every second butterfly stage: a normalized limb (below 2^29 + 8) that goes through two un-normalized
    differences u + KL_2 - t (each adds at most 2^30 + 64, sub_product) must still be a legal wide operand of a product (below 2^31.4, the column bound in check_bounds),
    and fit 32 bits with room for the carry step"""
    wide = int(2 ** 31.4); limb = (1 << 29) + 8
    for stage in range(2): limb += (1 << 30) + 64
    assert limb < wide, (limb, wide)
    assert limb + (1 << 30) + 64 >= wide            # (and a third stage would not fit: the rule in ntt29_lds_pass is as lazy as it can be)
check_bounds_ntt_stages()
def check_bounds_add():
    """the general addition of two accumulators (k_hacc_combine29): add-2008-s on operands within the invariant above stays within it"""
    bX, bY, bZ = 5.5, 3.6, 1.1
    U = prod(bX, bZ); S = prod(bY, bZ); cP = cR = 2; assert cP >= U + 0.01 and cR >= S + 0.01
    bP = U + cP; bR = S + cR; PP = prod(bP, bP); PPP = prod(bP, PP); Qv = prod(U, PP); RR = prod(bR, bR)
    cX = 4; assert cX >= PPP + 2 * Qv + 0.01; nX = RR + cX; cT = 6; assert cT >= nX + 0.01; bT = Qv + cT
    A = prod(bR, bT); Bv = prod(S, PPP); cY = 2; assert cY >= Bv + 0.01; nY = A + cY
    nZ = prod(prod(bZ, bZ), max(PP, PPP))
    assert nX <= bX and nY <= bY and nZ <= bZ, (nX, nY, nZ)
    for c, need in ((cP, U), (cR, S), (cX, PPP + 2 * Qv), (cT, nX), (cY, Bv)): assert KS[c][NL - 1] >= int(need * Q) >> (B * (NL - 1)), (c, need)
check_bounds_add()
def check_bounds_add_quad():
    """the quad-cooperative form of the same addition (htail29.cuh: quad29_add): identical formulas and constants, except that S1 enters its product with PPP as
    S2 + (2p + S1 - S2), i.e. with a value below S + cR instead of S — the product's result must still fit the constant of the difference that follows (cY), and the
    operand's limbs (a product's result plus a normalized difference: below 2^29 + 2^29 + 8) must stay below the 2^31 a product tolerates in ONE operand"""
    bX, bY, bZ = 5.5, 3.6, 1.1
    U = prod(bX, bZ); S = prod(bY, bZ); cP = cR = 2; bP = U + cP; bR = S + cR; PP = prod(bP, bP); PPP = prod(bP, PP); Qv = prod(U, PP); RR = prod(bR, bR)
    cX = 4; nX = RR + cX; cT = 6; bT = Qv + cT; A = prod(bR, bT); Bv = prod(S + cR, PPP); cY = 2
    assert cY >= Bv + 0.01 and A + cY <= bY, (Bv, A)
    assert KS[cY][NL - 1] >= int(Bv * Q) >> (B * (NL - 1))
    assert (1 << B) + (1 << B) + 8 < (1 << 31)
check_bounds_add_quad()
def check_bounds_dbl():
    """the doubling of an affine point (msm.cuh: xyzz29_dbl_affine, mdbl-2008-s-1): x < p, y < 2 p (K_2 - y); the result keeps to the invariant of the mixed addition"""
    x, y = 1.0, 2.0; U = 2 * y; V = prod(U, U); W = prod(U, V); S = prod(x, V); M = 3 * prod(x, x)
    cX = 4; assert cX >= 2 * S + 0.01; nX = prod(M, M) + cX; cT = 6; assert cT >= nX + 0.01; bT = S + cT
    A = prod(M, bT); Bv = prod(W, y); cY = 2; assert cY >= Bv + 0.01; nY = A + cY
    assert nX <= 5.5 and nY <= 3.6 and V <= 1.1 and W <= 1.1, (nX, nY, V, W)
    assert 3 * ((1 << B) + 8) < 1 << 31                                                 # 3 x^2 and 2 y before their carry step: limbs below 2^31
    for c, need in ((cX, 2 * S), (cT, nX), (cY, Bv)): assert KS[c][NL - 1] >= int(need * Q) >> (B * (NL - 1)), (c, need)
check_bounds_dbl()
def check_bounds_g2():
    """the mixed addition over Fq2 on 29-bit limbs (msm.cuh: XYZZ2_29::madd — the G2 half of the B query): Karatsuba products (three Fq products: v0 = a0 b0, v1 = a1 b1,
    v2 = (a0 + a1)(b0 + b1); c0 = v0 + K_2 - v1, c1 = v2 + K_4 - (v0 + v1)), complex squarings (c0 = (a0 + a1)(a0 + K_12 - a1), c1 = 2 a0 a1), differences with the
    constants below, and ONE Barrett step (value below 4.1 p) on each component of X3 and Y3.  Returns the invariant it proves: X, Y < 4.1, ZZ / ZZZ components below bZ."""
    def mul2(a, b):
        v0 = prod(a[0], b[0]); v1 = prod(a[1], b[1]); v2 = prod(a[0] + a[1], b[0] + b[1])
        assert v1 + 0.01 <= 2 and v0 + v1 + 0.01 <= 4, (v0, v1)                       # the constants K_2 / K_4 of the two differences
        return (v0 + 2, v2 + 4)
    def sqr2(a, c):
        assert a[1] + 0.01 <= c; m = prod(a[0], a[1]); return (prod(a[0] + a[1], a[0] + c), 2 * m)
    def sub2(a, b, c):
        assert b[0] + 0.01 <= c[0] and b[1] + 0.01 <= c[1], (b, c)
        for ci, bi in zip(c, b): assert KS[ci][NL - 1] >= int(bi * Q) >> (B * (NL - 1)), (ci, bi)   # the subtrahend's top limb never exceeds K_c's
        return (a[0] + c[0], a[1] + c[1])
    bXY = (4.1, 4.1); bZ = (3.2, 5.7)                                                 # the invariant (ZZ = ZZZ = one for the lifted point: below p)
    px = (1.0, 1.0); py = (2.0, 2.0)                                                  # y' = K_2 - y, normalized
    U2 = mul2(px, bZ); S2 = mul2(py, bZ); P = sub2(U2, bXY, (6, 6)); R = sub2(S2, bXY, (6, 6))
    PP = sqr2(P, 12); PPP = mul2(P, PP); Qv = mul2(bXY, PP); RR = sqr2(R, 12)
    s = (PPP[0] + 2 * Qv[0], PPP[1] + 2 * Qv[1]); X3 = sub2(RR, s, (12, 18)); assert max(X3) < LIN_MAX_UNITS
    T = sub2(Qv, bXY, (6, 6))                                                         # X3 after its Barrett step: below 4.1
    A = mul2(R, T); Bv = mul2(bXY, PPP); Y3 = sub2(A, Bv, (4, 6)); assert max(Y3) < LIN_MAX_UNITS
    ZZ3 = mul2(bZ, PP); ZZZ3 = mul2(bZ, PPP)
    assert max(ZZ3[0], ZZZ3[0]) <= bZ[0] and max(ZZ3[1], ZZZ3[1]) <= bZ[1], (ZZ3, ZZZ3)
    # a product's column: both operands may be sums of two normalized values (Karatsuba), limbs below 2^30 + 16
    assert 9 * ((1 << 30) + 16) ** 2 + 9 * (1 << 58) + (1 << 35) < 1 << 64
    return dict(X3=X3, Y3=Y3, ZZ=ZZ3, ZZZ=ZZZ3, P=P, R=R, T=T)

def check_bounds_oct():
    """the general addition over Fq2 with the point spread over EIGHT lanes (oct29.cuh: oct29_add — fold and tail of the G2 witness MSM, round 5): the formulas of
    add-2008-s in the four rounds of quad29_add, every Fq2 product as two dual products mul2 (component 0: a0 b0 + (K_6 - a1) b1, component 1: a0 b1 + a1 b0), every
    difference with the constants of the G1 form.  Operands: sums left by the lanes (XYZZ2_29::madd: X, Y < 4.1 p after their Barrett step, ZZ / ZZZ components below
    5.7 p) or earlier results of this addition.  Proves: the invariant X < 5.5, Y < 4.1, ZZ, ZZZ < 5.7 per component is kept, every negated operand stays below 6 p
    (K_6), every subtrahend below its constant, and the columns of mul2 below 2^64 with the limb sizes the kernel feeds it."""
    bX, bY, bZ = 5.5, 4.1, 5.7; NEG = 6.0
    def m2(a, b):
        assert a <= NEG - 0.01, a                                                     # (a's partner component is negated with K_6)
        return max(prod(a, b) + prod(NEG, b) - 1.0, 2 * prod(a, b) - 1.0)              # (x y + z w) / 2^261 + p  in units of p
    U = m2(bX, bZ); S = m2(bY, bZ); cP = cR = 2; assert cP >= U + 0.01 and cR >= S + 0.01
    bP = U + cP; bR = S + cR; PP = m2(bP, bP); RR = m2(bR, bR); Z12 = m2(bZ, bZ)
    PPP = m2(bP, PP); Qv = m2(U, PP); ZZ3 = m2(Z12, PP)
    cX = 4; assert cX >= PPP + 2 * Qv + 0.01; nX = RR + cX; cT = 6; assert cT >= nX + 0.01; bT = Qv + cT
    S1r = S + cR                                                                      # S1 rebuilt as S2 + (K_2 + S1 - S2), like the quad form
    ZZZ3 = m2(Z12, PPP); A = m2(bR, bT); Bv = m2(S1r, PPP); cY = 2; assert cY >= Bv + 0.01; nY = A + cY
    assert nX <= bX and nY <= bY and max(ZZ3, ZZZ3) <= bZ, (nX, nY, ZZ3, ZZZ3)
    for c, need in ((cP, U), (cR, S), (cX, PPP + 2 * Qv), (cT, nX), (cY, Bv), (6, NEG - 0.01)): assert KS[c][NL - 1] >= int(need * Q) >> (B * (NL - 1)), (c, need)
    # columns of mul2: 9 terms x y and 9 terms z w, x and z with limbs up to 2^30 + 8 (S1 rebuilt: a product's result plus a normalized difference), y and w normalized
    assert 18 * ((1 << 30) + 8) * ((1 << 29) + 8) + 9 * (1 << 58) + (1 << 35) < 1 << 64
    # the negation K_6 - v takes limbs below 3 * (2^29 + 8)
    assert (1 << 30) + 8 < 3 * ((1 << 29) + 8)
    return dict(X=nX, Y=nY, ZZ=ZZ3, ZZZ=ZZZ3, P=bP, R=bR, T=bT)
BOUNDS_OCT = check_bounds_oct()

HEADER = ["// GENERATED by gen_field29.py - do not edit.  Fq and Fr on nine 29-bit limbs (R' = 2^261): Fq29 for k_hacc_runs29 (msm.cuh) and the verifier's schedule",
          "// (pairing.cuh), Fr29 for the transforms (ntt.cuh); device compilation only.",
          "// value bounds of the mixed addition, in units of p (interval arithmetic in the generator): " + ", ".join("%s < %.2f" % kv for kv in BOUNDS["bounds"].items())]
EMIT_BARRETT = False
def gen_struct():
    """the struct for the modulus the globals (P29, INV, ONE29, RCONV, KS) currently describe, under the name Fq29 (the caller renames)"""
    global out, cols
    out = [
           "struct Fq29 {",
           "  uint32_t l[9];",
           "  static constexpr uint32_t MASK = 0x%xu, INV = 0x%xu;   // INV = -p^-1 mod 2^29" % (M, INV)]
    def arr(name, v, comment=""): out.append("  static constexpr uint32_t %s[9] = {%s};%s" % (name, ", ".join("0x%xu" % x for x in v), ("   // " + comment) if comment else ""))
    arr("P29", P29, "p"); arr("ONE", ONE29, "2^261 mod p: the field's one"); arr("RCONV", RCONV, "2^256 mod p as a plain integer: a Montgomery product with it turns v 2^261 into v 2^256 (the 8 x 32-bit form)")
    for c, v in KS.items(): arr("K%d" % c, v, "%d p, low limbs in [3 * 2^29 + 64, 4 * 2^29)" % c)
    if EMIT_BARRETT: arr("NP", NP, "2^264 - p (top limb: 32 bits)")
    arr("KL2", adjusted_light(2), "2 p, low limbs in [2^29 + 64, 2 * 2^29 + 64)"); assert adjusted_light(2)[NL - 1] >= int(1.5 * Q) >> (B * (NL - 1))   # a subtrahend below 1.5 p never exceeds KL2's top limb
    def mads(pairs, const_b, first):
        """one asm statement: acc (+)= sum of the pairs' products; no carries: the column stays below 2^64"""
        txt = " ".join('"v_mad_u64_u32 %%0, vcc, %%%d, %%%d, %s\\n\\t"' % (1 + 2 * i, 2 + 2 * i, "0" if first and i == 0 else "%0") for i in range(len(pairs)))
        ins = ", ".join('"v"(%s), "%s"(%s)' % (x, "s" if const_b else "v", y) for x, y in pairs)
        out.append("    asm(%s : \"%s\"(acc) : %s : \"vcc\");" % (txt, "=&v" if first else "+v", ins))
    def product(sig, columns, prologue):
        out.append("  static __device__ __forceinline__ Fq29 %s {" % sig)
        out.append("    uint64_t acc; uint32_t m0, m1, m2, m3, m4, m5, m6, m7, m8; Fq29 r;")
        out.extend(prologue)
        for k in range(17):
            first = k == 0
            for ch in range(0, len(columns[k]), 9): mads(columns[k][ch:ch + 9], False, first and ch == 0)   # (an asm statement takes 30 operands: nine pairs a time)
            if k < 9:
                mp = [("m%d" % i, "P29[%d]" % (k - i)) for i in range(0, k)]
                if mp: mads(mp, True, False)
                out.append("    m%d = ((uint32_t)acc * INV) & MASK;" % k)
                mads([("m%d" % k, "P29[0]")], True, False)
                out.append("    acc >>= 29;")
            else:
                mads([("m%d" % i, "P29[%d]" % (k - i)) for i in range(k - 8, 9)], True, False)
                out.append("    r.l[%d] = (uint32_t)acc & MASK; acc >>= 29;" % (k - 9))
        out.append("    r.l[8] = (uint32_t)acc;")
        out.append("    return r;")
        out.append("  }")
    product("mul(const Fq29 &a, const Fq29 &b)", [[("a.l[%d]" % i, "b.l[%d]" % (k - i)) for i in range(max(0, k - 8), min(k, 8) + 1)] for k in range(17)], [])
    # a b + c d with ONE reduction (the two halves of an Fq2 product's component: a0 b0 + (-a1) b1, a0 b1 + a1 b0): 162 + 81 multiply-adds instead of 2 x 162.  Columns
    # of up to 18 + 9 terms: every operand normalized (limbs below 2^29 + 8) except a and c, whose limbs may reach 2^30 + 8 (check_bounds_oct)
    product("mul2(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d)",
            [[("a.l[%d]" % i, "b.l[%d]" % (k - i)) for i in range(max(0, k - 8), min(k, 8) + 1)] + [("c.l[%d]" % i, "d.l[%d]" % (k - i)) for i in range(max(0, k - 8), min(k, 8) + 1)] for k in range(17)], [])
    cols = [[] for _ in range(17)]
    for i in range(9):
        cols[2 * i].append(("a.l[%d]" % i, "a.l[%d]" % i))
        for j in range(i + 1, 9): cols[i + j].append(("a.l[%d]" % i, "d%d" % j))
    assert sum(len(c) for c in cols) == 45
    product("sqr(const Fq29 &a)", cols, ["    const uint32_t " + ", ".join("d%d = a.l[%d] << 1" % (j, j) for j in range(1, 9)) + ";   // a normalized: 2 a_j < 2^30 + 16"])
    out.append("""  // one parallel carry step: limbs 0..7 back below 2^29 + 8 (inputs below 2^32)
      __device__ __forceinline__ Fq29 norm() const { Fq29 r; r.l[0] = l[0] & MASK;
    #pragma unroll
        for (int i = 1; i < 8; i++) r.l[i] = (l[i] & MASK) + (l[i - 1] >> 29);
        r.l[8] = l[8] + (l[7] >> 29); return r; }
""" + ("""      // a - b (mod p) as a + K_C - b, normalized.  b: limbs below 3 * (2^29 + 8), value below C p
      template <int C> static __device__ __forceinline__ Fq29 sub(const Fq29 &a, const Fq29 &b) { Fq29 d;
    #pragma unroll
        for (int i = 0; i < 9; i++) d.l[i] = a.l[i] + (C == 2 ? K2[i] : C == 4 ? K4[i] : C == 6 ? K6[i] : C == 12 ? K12[i] : K18[i]) - b.l[i];
        return d.norm(); }
      // neg ? -a : a for a canonical a (limbs below 2^29): K_2 - a is left as it is (limbs below 2^31, fine as ONE operand of a product)
      static __device__ __forceinline__ Fq29 cond_neg(const Fq29 &a, bool neg) { Fq29 r;
    #pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = neg ? K2[i] - a.l[i] : a.l[i];
        return r; }
""" if KS else "") + """      // a - t (mod p) as a + KL_2 - t for a product's result t (exact 29-bit limbs, value below 2 p); limbs of the difference: a's + 2^30 + 64 at most, NOT normalized
      static __device__ __forceinline__ Fq29 sub_product(const Fq29 &a, const Fq29 &t) { Fq29 d;
    #pragma unroll
        for (int i = 0; i < 9; i++) d.l[i] = a.l[i] + KL2[i] - t.l[i];
        return d; }
      // KL_2 - t: minus a product's result (limbs below 2^30 + 64: fine as ONE operand of a product)
      static __device__ __forceinline__ Fq29 neg_product(const Fq29 &t) { Fq29 d;
    #pragma unroll
        for (int i = 0; i < 9; i++) d.l[i] = KL2[i] - t.l[i];
        return d; }
""" + ("""      // V - q p with q = max(estimate of V / p from the top limb - 1, 0), as V + q (2^264 - p) with the top limb modulo 2^32 (the q multiples of 2^264 fall out): limbs below
      // 2^31 and a value below 1200 p in, normalized limbs and a value below 4.1 p out (the model and its check on integers: barrett / lin_check in the generator)
      __device__ __forceinline__ Fq29 barrett() const {
        uint32_t q = (uint32_t)(((uint64_t)l[8] * %dull) >> 32) >> %d; q = q ? q - 1 : 0;
        uint64_t acc[9]; Fq29 r, t;
    #pragma unroll
        for (int i = 0; i < 9; i++) acc[i] = (uint64_t)l[i] + (uint64_t)q * NP[i];
        t.l[0] = (uint32_t)acc[0] & MASK;
    #pragma unroll
        for (int i = 1; i < 8; i++) t.l[i] = ((uint32_t)acc[i] & MASK) + (uint32_t)(acc[i - 1] >> 29);
        t.l[8] = (uint32_t)acc[8] + (uint32_t)(acc[7] >> 29);
        r.l[0] = t.l[0] & MASK;
    #pragma unroll
        for (int i = 1; i < 8; i++) r.l[i] = (t.l[i] & MASK) + (t.l[i - 1] >> 29);
        r.l[8] = t.l[8] + (t.l[7] >> 29); return r; }
""" % (MU, MU_SHIFT - 32) if EMIT_BARRETT else "") + """      // limb-wise sum, NOT normalized
      static __device__ __forceinline__ Fq29 add_raw(const Fq29 &a, const Fq29 &b) { Fq29 d;
    #pragma unroll
        for (int i = 0; i < 9; i++) d.l[i] = a.l[i] + b.l[i];
        return d; }
      static __device__ __forceinline__ Fq29 one() { Fq29 r;
    #pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = ONE[i];
        return r; }
      // the 8 x 32-bit words of a canonical value (below 2^256) <-> its limbs
      static __device__ __forceinline__ Fq29 unpack(const uint32_t (&w)[8]) { Fq29 r;
    #pragma unroll
        for (int i = 0; i < 9; i++) { const int bit = 29 * i, j = bit >> 5, s = bit & 31; uint32_t v = w[j] >> s; if (s > 3 && j + 1 < 8) v |= w[j + 1] << (32 - s); r.l[i] = i < 8 ? v & MASK : v; }
        return r; }
      // the 8 x 32-bit words of a value with exact 29-bit limbs below 2^256
      __device__ __forceinline__ void pack_words(uint32_t (&w)[8]) const {
    #pragma unroll
        for (int j = 0; j < 8; j++) { const int bit = 32 * j, i = bit / 29, s = bit - 29 * i; uint32_t v = l[i] >> s; v |= l[i + 1] << (29 - s); if (29 - s + 29 < 32 && i + 2 < 9) v |= l[i + 2] << (58 - s); w[j] = v; } }
      // Montgomery product with 2^256 mod p: x = v 2^261 becomes v 2^256 (mod p), below 2p, as 8 x 32-bit words — the lazy domain of field.cuh
      __device__ __forceinline__ void to_words(uint32_t (&w)[8]) const { Fq29 c;
    #pragma unroll
        for (int i = 0; i < 9; i++) c.l[i] = RCONV[i];
        const Fq29 t = mul(*this, c);   // exact 29-bit limbs, value below 2p < 2^255
    #pragma unroll
        for (int j = 0; j < 8; j++) { const int bit = 32 * j, i = bit / 29, s = bit - 29 * i; uint32_t v = t.l[i] >> s; v |= t.l[i + 1] << (29 - s); if (29 - s + 29 < 32 && i + 2 < 9) v |= t.l[i + 2] << (58 - s); w[j] = v; } }
    };""")
    
    return out
# ---- self-check: the column lists above, evaluated on integers, against big-number arithmetic -------------------------------------------------------
def model_product(columns, env):
    """mirrors product(): columns[k] = pairs of operand names looked up in env (limb values); returns the nine result limbs"""
    acc = 0; m = []; r = []
    for k in range(17):
        for x, y in columns[k]: acc += env[x] * env[y]; assert acc < 1 << 64
        if k < 9:
            for i in range(k): acc += m[i] * P29[k - i]
            m.append(((acc & 0xffffffff) * INV) & M); acc += m[k] * P29[0]; assert acc < 1 << 64 and acc & M == 0; acc >>= B
        else:
            for i in range(k - 8, 9): acc += m[i] * P29[k - i]
            assert acc < 1 << 64; r.append(acc & M); acc >>= B
    r.append(acc); return r
def val(l): return sum(v << (B * i) for i, v in enumerate(l))
def self_check():
    import random; rnd = random.Random(29); RPinv = pow(RP, -1, Q)
    mulcols = [[("a%d" % i, "b%d" % (k - i)) for i in range(max(0, k - 8), min(k, 8) + 1)] for k in range(17)]
    sqrcols = [[(x.replace("a.l[", "a").replace("]", ""), y.replace("a.l[", "a").replace("]", "")) for x, y in c] for c in cols]
    mul2cols = [[("a%d" % i, "b%d" % (k - i)) for i in range(max(0, k - 8), min(k, 8) + 1)] + [("c%d" % i, "d%d" % (k - i)) for i in range(max(0, k - 8), min(k, 8) + 1)] for k in range(17)]
    def norm(l): return [l[0] & M] + [(l[i] & M) + (l[i - 1] >> B) for i in range(1, 8)] + [l[8] + (l[7] >> B)]
    for it in range(2000):
        # a: an un-normalized difference (limbs up to 2^31.4, value below 8p), b: normalized
        bv = rnd.randrange(0, 8 * Q); b = norm(limbs29(bv)); av = rnd.randrange(0, 2 * Q); a = [x + y - z for x, y, z in zip(limbs29(av), KS[6] if KS else adjusted_light(2), limbs29(rnd.randrange(0, (5 if KS else 2) * Q)))]
        if it < 50: a = [min(int(2 ** 31.4), (1 << 32) - 1)] * 8 + [a[8]]; b = [(1 << 29) + 7] * 8 + [b[8]]      # the column bound at its worst
        assert all(0 <= x < 1 << 32 for x in a)
        env = {"a%d" % i: a[i] for i in range(9)}; env.update({"b%d" % i: b[i] for i in range(9)})
        r = model_product(mulcols, env); assert all(x <= M for x in r[:8]) and val(r) % Q == val(a) * val(b) * RPinv % Q and val(r) < val(a) * val(b) // RP + Q + 1
        env = {"a%d" % i: b[i] for i in range(9)}; env.update({"d%d" % i: (b[i] << 1) & 0xffffffff for i in range(1, 9)})
        r = model_product(sqrcols, env); assert val(r) % Q == val(b) * val(b) * RPinv % Q
        # the dual product a b + c d: a, c raw sums of two normalized values (limbs up to 2^30 + 8), b, d normalized; at its worst in the first iterations
        wide = lambda: [x + y for x, y in zip(norm(limbs29(rnd.randrange(0, 6 * Q))), norm(limbs29(rnd.randrange(0, 6 * Q))))]
        a2, c2 = wide(), wide(); b2, d2 = norm(limbs29(rnd.randrange(0, 8 * Q))), norm(limbs29(rnd.randrange(0, 8 * Q)))
        if it < 50: a2 = [(1 << 30) + 8] * 8 + [a2[8]]; c2 = [(1 << 30) + 8] * 8 + [c2[8]]; b2 = [(1 << 29) + 7] * 8 + [b2[8]]; d2 = [(1 << 29) + 7] * 8 + [d2[8]]
        env = {}
        for nm, v in (("a", a2), ("b", b2), ("c", c2), ("d", d2)): env.update({"%s%d" % (nm, i): v[i] for i in range(9)})
        r = model_product(mul2cols, env); tot = val(a2) * val(b2) + val(c2) * val(d2); assert all(x <= M for x in r[:8]) and val(r) % Q == tot * RPinv % Q and val(r) < tot // RP + Q + 1
        # difference + normalization
        if not KS:   # the light difference: a + KL_2 - t for a product's result t, then one carry step
            tv = rnd.randrange(0, 2 * Q); d = [x + k - y for x, k, y in zip(b, adjusted_light(2), limbs29(tv))]; assert all(0 <= x < 1 << 32 for x in d); n = norm(d); assert val(n) == val(b) + 2 * Q - tv and all(x < (1 << 29) + 8 for x in n[:8])
            continue
        c = rnd.choice([2, 4, 6]); sv = rnd.randrange(0, c * Q); sl = limbs29(sv)
        if c == 4: s1, s2 = rnd.randrange(0, Q + Q // 8), rnd.randrange(0, Q + Q // 8); sl = [x + 2 * y for x, y in zip(limbs29(s1), limbs29(s2))]; sv = s1 + 2 * s2      # PPP + 2Q, limb-wise
        d = [x + k - y for x, k, y in zip(b, KS[c], sl)]; assert all(0 <= x < 1 << 32 for x in d), d
        n = norm(d); assert val(n) == val(b) + c * Q - sv and all(x < (1 << 29) + 8 for x in n[:8])
        # words <-> limbs
        w = rnd.randrange(0, 1 << 256); ws = [(w >> (32 * j)) & 0xffffffff for j in range(8)]; u = []
        for i in range(9):
            bit = 29 * i; j = bit >> 5; sft = bit & 31; v = ws[j] >> sft
            if sft > 3 and j + 1 < 8: v |= (ws[j + 1] << (32 - sft)) & 0xffffffff
            u.append(v & M if i < 8 else v)
        assert val(u) == w
        t = limbs29(rnd.randrange(0, 2 * Q)); back = []
        for j in range(8):
            bit = 32 * j; i = bit // 29; sft = bit - 29 * i; v = t[i] >> sft; v |= (t[i + 1] << (29 - sft)) & 0xffffffff
            if 29 - sft + 29 < 32 and i + 2 < 9: v |= (t[i + 2] << (58 - sft)) & 0xffffffff
            back.append(v & 0xffffffff)
        assert sum(x << (32 * j) for j, x in enumerate(back)) == val(t)
gen_struct(); self_check()

# ---- plain constants + the linear-combination pipeline of the verifier's schedule (verify_sched.hpp, k_verify_sched29 in pairing.cuh) -------------------------
# LIN: v = sum c_t * y_t over 64-bit limb accumulators (y = x or K_6 - x), one parallel carry step to 32-bit limbs, a tree of limb-wise sums, then a Barrett-like
# step: q ~ floor(V / p) from the top limb, V + q * (2^264 - p) with the top limb kept modulo 2^32 (the multiples of 2^264 fall out), two carry steps.  Result: limbs
# below 2^29 + 2, value below 4.1 p.  Checked here on integers against big-number arithmetic, at the bounds the schedule builder enforces (sum |c| * 6 p < 1200 p).
TOP = B * (NL - 1)                                   # 232: the top limb holds bits 232 .. 263
P8 = Q >> TOP; NP = limbs29((1 << 264) - Q); assert NP[8] < 1 << 32
MU_SHIFT = 53; MU = (1 << MU_SHIFT) // (P8 + 1); assert MU < 1 << 32
LIN_MAX_UNITS = 1200                                 # V < 1200 p: the top limb stays below 2^32
assert (LIN_MAX_UNITS * Q) >> TOP < 1 << 32
def norm64(acc, wrap_top=False):
    l = [acc[0] & M] + [(acc[i] & M) + (acc[i - 1] >> B) for i in range(1, 8)] + [acc[8] + (acc[7] >> B)]
    if wrap_top: l[8] &= 0xffffffff
    return l
def norm32(l, wrap_top=False):
    r = [l[0] & M] + [(l[i] & M) + (l[i - 1] >> B) for i in range(1, 8)] + [l[8] + (l[7] >> B)]
    if wrap_top: r[8] &= 0xffffffff
    return r
def barrett(l):
    """l: limbs below 2^31, value V below LIN_MAX_UNITS p.  Returns normalized limbs of V - q p, q = max(q_est - 1, 0)"""
    assert all(0 <= x < 1 << 32 for x in l)
    q = ((l[8] * MU) >> 32) >> (MU_SHIFT - 32); q = q - 1 if q else 0
    acc = [l[i] + q * NP[i] for i in range(9)]; assert all(a < 1 << 64 for a in acc)
    r = norm32(norm64(acc, True), True); return r, q
def lin_check():
    import random; rnd = random.Random(261)
    for it in range(4000):
        nt = rnd.randrange(1, 25); units = 0; terms = []
        for t in range(nt):
            c = rnd.choice([1, 1, 1, 2, 3, 9, 18, 32, 81]); 
            if (units + c) * 6 >= LIN_MAX_UNITS - 6: break
            units += c; xv = rnd.randrange(0, int(4.1 * Q)) if it % 7 else int(4.1 * Q) - 1 - rnd.randrange(0, 1000); x = limbs29(xv)
            if it % 3 == 0: x = [min(v + rnd.randrange(0, 3), (1 << 29) + 1) for v in x[:8]] + [x[8]]; xv = val(x)      # lazily normalized operand
            terms.append((c, rnd.random() < 0.4, x, xv))
        if it % 50 == 0: terms = [(1, False, limbs29(5), 5), (1, True, limbs29(5), 5)]                                     # a difference that is exactly a multiple of p
        # eight lanes, three terms each (the kernel's layout)
        lanes = [[0] * 9 for _ in range(8)]; V = 0
        for k, (c, neg, x, xv) in enumerate(terms):
            y = [KS[6][i] - x[i] for i in range(9)] if neg else x; assert all(0 <= v < 1 << 31 for v in y[:8]), y
            V += c * (6 * Q - xv if neg else xv)
            for i in range(9): lanes[k % 8][i] += c * y[i]
        part = [norm64(a) for a in lanes]; assert all(v < (1 << 29) + (1 << 10) for pl in part for v in pl[:8])
        s4 = [[sum(part[j][i] for j in range(g, g + 4)) for i in range(9)] for g in (0, 4)]; assert all(v < 1 << 32 for s_ in s4 for v in s_)
        s4 = [norm32(x) for x in s4]; tot = [s4[0][i] + s4[1][i] for i in range(9)]; assert val(tot) == V and all(v < 1 << 31 for v in tot[:8]) and tot[8] < 1 << 32
        r, q = barrett(tot); assert val(r) == V - q * Q, (val(r), V, q); assert all(v < (1 << 29) + 2 for v in r[:8]) and val(r) < 4.1 * Q and (q == 0 or val(r) >= Q)
        # multiples of p are recognised after exact carry propagation
        f = list(r)
        for i in range(8): f[i + 1] += f[i] >> B; f[i] &= M
        assert (V % Q == 0) == any(f == limbs29(k * Q) for k in range(5))
lin_check()
BOUNDS_G2 = check_bounds_g2()
prm = ["// GENERATED by gen_field29.py - do not edit.  Constants of Fq on nine 29-bit limbs, for host and device code (verify_sched.hpp).", "#pragma once", "#include <cstdint>", "namespace zk { namespace p29 {"]
def carr(name, v, comment=""): prm.append("constexpr uint32_t %s[9] = {%s};%s" % (name, ", ".join("0x%xu" % x for x in v), ("   // " + comment) if comment else ""))
prm.append("constexpr uint32_t MASK = 0x%xu, INV = 0x%xu, MU = 0x%xu, MU_SHIFT = %d, LIN_MAX_UNITS = %d;   // MU = floor(2^53 / ((p >> 232) + 1)); a linear combination's value stays below LIN_MAX_UNITS p" % (M, INV, MU, MU_SHIFT, LIN_MAX_UNITS))
carr("P", P29, "p"); carr("K6", KS[6], "6 p, low limbs in [3 * 2^29 + 64, 4 * 2^29): K6 - x has no negative limb for x below 6 p with limbs below 3 * (2^29 + 8)"); carr("NP", NP, "2^264 - p (top limb: 32 bits)")
prm.append("constexpr uint32_t KP[5][9] = {%s};   // k p, k = 0 .. 4, exact limbs" % ", ".join("{" + ", ".join("0x%xu" % x for x in limbs29(k * Q)) + "}" for k in range(5)))
prm.append("} }")
open(__file__.replace("gen_field29.py", "field29_params.h"), "w").write("\n".join(prm) + "\n")
EMIT_BARRETT = True; FQ_LINES = gen_struct(); EMIT_BARRETT = False
# ---- the same arithmetic for Fr (the transforms): constants of the scalar field, same column schedule, same self-check --------------------------------------------
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_SAVE = Q; KS_SAVE = KS; Q = R_MOD; INV = (-pow(Q, -1, 1 << B)) % (1 << B); P29 = limbs29(Q); ONE29 = limbs29(RP % Q); RCONV = limbs29((1 << 256) % Q); KS = {}   # (no borrow-adjusted 2p, 4p, 6p with limbs in [3 * 2^29 + 64, 4 * 2^29) exist for r = 1 mod 2^28; the transforms only subtract products)
FR_LINES = [l.replace("Fq29", "Fr29") for l in gen_struct()]; self_check()
open(__file__.replace("gen_field29.py", "field29_gfx950.inc"), "w").write("\n".join(HEADER + FQ_LINES + FR_LINES) + "\n")
print("bounds:", BOUNDS); print("bounds of the oct addition:", BOUNDS_OCT)
