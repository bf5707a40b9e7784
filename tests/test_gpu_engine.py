"""GPU parity: the HIP kernels (through libzkgpu.so's C-ABI) against the CPU oracle on the same seeded inputs, plus
size-independent properties at larger sizes.  Integer work: every comparison is bit-exact."""
import numpy as np
import pytest
from oracle import pyoracle as o
from blockmaze_amd import engine as e
from r1cs_util import random_r1cs

pytestmark = pytest.mark.gpu

def rand_field_arr(seed, n):
    g = o.SplitMix64(seed); return o.to_arr([g.field() for _ in range(n)])

@pytest.mark.parametrize("field", [o.FR, o.FQ])
def test_field_ops_match_oracle(field):
    n = 300; a = rand_field_arr(11 + field, n); b = rand_field_arr(23 + field, n)
    mod = o.R_MOD if field == o.FR else o.Q_MOD
    edge = o.to_arr([0, 1, mod - 1, mod - 2, 2, (1 << 253) - 1, mod >> 1, 3]); a[:8] = edge; b[:8] = edge[::-1]
    for op in ("mul", "add", "sub"):
        assert o.from_arr(e.field_op(field, op, a, b)) == o.field_op(field, op, o.from_arr(a), o.from_arr(b)), op
    for op in ("sqr", "neg"):
        assert o.from_arr(e.field_op(field, op, a)) == o.field_op(field, op, o.from_arr(a)), op
    nz = a[8:40]; assert o.from_arr(e.field_op(field, "inv", nz)) == o.field_op(field, "inv", o.from_arr(nz))
    # the lazy domain [0, 2p) of the bucket accumulation (field.cuh): products without the final subtraction on operands pushed towards 2p, differences that add 2p after a
    # borrow, the masked negation — all normalized before they come back
    for lazy, op in (("mul_lazy", "mul"), ("sub_lazy", "sub")):
        assert o.from_arr(e.field_op(field, lazy, a, b)) == o.field_op(field, op, o.from_arr(a), o.from_arr(b)), lazy
    assert o.from_arr(e.field_op(field, "sqr_lazy", a)) == o.field_op(field, "sqr", o.from_arr(a))
    assert o.from_arr(e.field_op(field, "neg_masked", a)) == o.field_op(field, "neg", o.from_arr(a))

def test_fq2_ops_match_oracle():
    n = 64; a = rand_field_arr(5, 2 * n).reshape(n, 8); b = rand_field_arr(6, 2 * n).reshape(n, 8)
    mul = e.fq2_op("mul", a, b); sqr = e.fq2_op("sqr", a); inv = e.fq2_op("inv", a)
    for i in range(n):
        x = tuple(o.from_arr(a[i])); y = tuple(o.from_arr(b[i]))
        assert tuple(o.from_arr(mul[i])) == o.fq2_op("mul", x, y) and tuple(o.from_arr(sqr[i])) == o.fq2_op("sqr", x) and tuple(o.from_arr(inv[i])) == o.fq2_op("inv", x)

def test_g1_group_law_matches_oracle():
    n = 40; g = o.SplitMix64(77); P = o.g1_consecutive(g.field(), n); Q = o.g1_consecutive(g.field(), n)
    Q[3] = P[3]                                   # doubling through the addition formulas
    Q[4] = o.g1_arr([o.g1_op("neg", o.g1_from(P[4])[0])])[0]   # P + (-P) = infinity
    P[5] = 0                                      # infinity + Q
    Q[6] = 0                                      # P + infinity
    add = e.group_op(1, "add", P, Q); madd = e.group_op(1, "madd", P, Q); dbl = e.group_op(1, "dbl", P)
    ks = [int(g.next() & 0xFFFF) for _ in range(n)]; ks[0] = 0; ks[1] = 1; mul = e.group_op(1, "mul_small", P, ks)
    for i in range(n):
        p, q = o.g1_from(P[i])[0], o.g1_from(Q[i])[0]; exp = o.g1_op("add", p, q)
        assert o.g1_from(add[i])[0] == exp and o.g1_from(madd[i])[0] == exp, i
        assert o.g1_from(dbl[i])[0] == o.g1_op("dbl", p) and o.g1_from(mul[i])[0] == o.g1_op("mul", p, k=ks[i]), i

def test_g2_group_law_matches_oracle():
    n = 16; g = o.SplitMix64(78); P = o.g2_consecutive(g.field(), n); Q = o.g2_consecutive(g.field(), n); Q[3] = P[3]; P[5] = 0
    add = e.group_op(2, "add", P, Q); madd = e.group_op(2, "madd", P, Q); dbl = e.group_op(2, "dbl", P)
    for i in range(n):
        p, q = o.g2_from(P[i])[0], o.g2_from(Q[i])[0]; exp = o.g2_op("add", p, q)
        assert o.g2_from(add[i])[0] == exp and o.g2_from(madd[i])[0] == exp and o.g2_from(dbl[i])[0] == o.g2_op("dbl", p), i

def witness_like_scalars(seed, n):
    """the reference witness mix: ~51% zero, ~46% one, rest small / full width"""
    g = o.SplitMix64(seed); z = []
    for _ in range(n):
        sel = g.next() % 1000
        z.append(0 if sel < 509 else 1 if sel < 967 else (g.next() & 0xFFFFFFFF) if sel < 993 else (g.next()) if sel < 999 else g.field())
    return o.to_arr(z)

@pytest.mark.parametrize("n,c", [(1, 0), (2, 7), (100, 8), (1000, 0), (4096, 10), (5000, 13)])
def test_msm_g1_matches_oracle(n, c):
    g = o.SplitMix64(1000 + n); P = o.g1_consecutive(g.field(), n); K = rand_field_arr(2000 + n, n)
    if n > 50: P[7] = 0; K[9] = 0; K[10] = o.to_arr([o.R_MOD - 1])[0]; P[11] = P[12]
    assert o.g1_from(e.msm(1, P, K, c))[0] == o.msm_g1(P, K)
    Z = witness_like_scalars(3000 + n, n)
    assert o.g1_from(e.msm(1, P, Z, c, filter_ones=True))[0] == o.msm_g1(P, Z, mixed=True)
    assert o.g1_from(e.msm(1, P, Z, c, filter_ones=False))[0] == o.msm_g1(P, Z, mixed=True)      # same sum without the 0/1 fast path

@pytest.mark.parametrize("n,c", [(1, 0), (33, 7), (1000, 9)])
def test_msm_g2_matches_oracle(n, c):
    g = o.SplitMix64(1100 + n); P = o.g2_consecutive(g.field(), n); K = rand_field_arr(2100 + n, n); Z = witness_like_scalars(3100 + n, n)
    assert o.g2_from(e.msm(2, P, K, c))[0] == o.msm_g2(P, K)
    assert o.g2_from(e.msm(2, P, Z, c, filter_ones=True))[0] == o.msm_g2(P, Z, mixed=True)

@pytest.mark.parametrize("group", [1, 2])
def test_witness_msm_fast_path_edge_cases(group):
    """the three-launch witness path (k_wsort / k_wacc_* / k_wtail, window 8 = 128 buckets) on the shapes that stress it: nothing but ones, nothing but zeros, no
    ones at all, one value repeated (every digit lands in the same bucket of its window: lanes are dealt by fill, and past the region size the general path must take
    over), a single non-trivial scalar, and ordinary witness-like vectors before and after on the same resident object"""
    n = 3000 if group == 1 else 600; g = o.SplitMix64(500 + group); P = (o.g1_consecutive if group == 1 else o.g2_consecutive)(g.field(), n)
    ref = (lambda K: o.msm_g1(P, K, mixed=True)) if group == 1 else (lambda K: o.msm_g2(P, K, mixed=True)); dec = o.g1_from if group == 1 else o.g2_from
    m = e.ResidentMsm(group, P, 8, filter_ones=True)
    def check(K): m.set_scalars(K); assert dec(m.run())[0] == ref(K)
    check(witness_like_scalars(77, n))
    check(o.to_arr([1] * n)); check(o.to_arr([0] * n)); check(o.to_arr([2 + (i % 251) for i in range(n)]))
    check(o.to_arr([0x0101010101010101] * n))                                                      # eight digits equal to 1: one bucket holds 8 n entries
    one = [0] * n; one[n // 2] = o.R_MOD - 1; check(o.to_arr(one))
    check(o.to_arr([o.R_MOD - 1 - i for i in range(n)]))                                           # full-width values only: 32 digits each
    check(witness_like_scalars(78, n)); m.close()

def test_msm_one_pass_sort_and_its_overflow_fallback():
    """uniform-scalar hint (the H query): slots of fixed capacity per bucket; a scalar vector that piles everything into a few buckets overflows them and must
    come out right through the two-pass fallback, and the object must keep working afterwards"""
    n = 6000; g = o.SplitMix64(77); P = o.g1_consecutive(g.field(), n); m = e.ResidentMsm(1, P, 10, filter_ones=2)
    K = rand_field_arr(4242, n); m.set_scalars(K); assert o.g1_from(m.run())[0] == o.msm_g1(P, K)
    same = o.to_arr([0x1234567 << 40 | 3] * n); m.set_scalars(same); assert o.g1_from(m.run())[0] == o.msm_g1(P, same)          # every digit equal: overflow
    K2 = rand_field_arr(4243, n); m.set_scalars(K2); assert o.g1_from(m.run())[0] == o.msm_g1(P, K2)
    m.close()

def test_h_path_degenerate_additions_fall_back_and_stay_exact():
    """the H query's 29-bit run accumulation (k_hacc_runs29, filter_ones=2) uses INCOMPLETE mixed additions: an operand equal to +-the accumulator leaves ZZ = 0 (mod p),
    which k_hacc_combine29 must notice and send to the general path (complete formulas: alt_bn128_g1.cpp:139-195 '+' with its doubling fallback, multiexp.tcc:165-282).
    Inputs a key could legally hold: repeated points with equal scalars (doubling inside a run), P / -P pairs with equal scalars (cancellation), every point equal, and a
    mixture of all three; each must give the oracle's sum and the resident object must keep working on ordinary scalars afterwards."""
    n = 6000; g = o.SplitMix64(4104); base = o.g1_consecutive(g.field(), n); K = rand_field_arr(4105, n)
    def neg(pt):                                                                                  # (x, -y): q - y on the stored words, Montgomery form or not
        r = pt.copy(); r[4:] = o.limbs((o.Q_MOD - o.from_arr(pt)[1]) % o.Q_MOD); return r
    def run(P, S):
        m = e.ResidentMsm(1, P, 10, filter_ones=2); m.set_scalars(S); got = o.g1_from(m.run())[0]; assert got == o.msm_g1(P, S)
        S2 = rand_field_arr(4106, len(P)); m.set_scalars(S2); assert o.g1_from(m.run())[0] == o.msm_g1(P, S2); m.close()        # still usable, still exact
    # (a) pairs of equal points with equal scalars: the same digit in every window, so both land in the same bucket
    P = base.copy(); S = K.copy(); P[1::2] = P[0::2]; S[1::2] = S[0::2]; run(P, S)
    # (b) P, -P with equal scalars: every bucket cancels to the point at infinity (the whole sum is the group's zero)
    P = base.copy(); S = K.copy()
    for i in range(0, 400, 2): P[i + 1] = neg(P[i]); S[i + 1] = S[i]
    P = P[:400]; S = S[:400]; m = e.ResidentMsm(1, P, 10, filter_ones=2); m.set_scalars(S); assert o.g1_from(m.run())[0] is None and o.msm_g1(P, S) is None; m.close()
    # (c) every point equal, random scalars: every bucket collision is a doubling
    P = np.repeat(base[:1], 3000, axis=0); S = K[:3000]; run(P, S)
    # (d) a mixture inside an otherwise ordinary query
    P = base.copy(); S = K.copy(); P[100:200] = P[99]; S[100:150] = S[99]
    for i in range(1000, 1100, 2): P[i + 1] = neg(P[i]); S[i + 1] = S[i]
    run(P, S)

@pytest.mark.parametrize("group", [1, 2], ids=["G1", "G2"])
def test_witness_path_degenerate_points_fall_back_and_stay_exact(group):
    """the fused witness path (window 8, filter_ones) on points a key could legally hold but the incomplete additions of its folds and tails (htail29.cuh quad29_add,
    oct29.cuh oct29_add — and the G2 lanes) cannot add: every point EQUAL (every fold addition is P + P: ZZ = 0 mod p, flagged, the general path repeats the MSM with
    complete formulas), P / -P pairs (sums that are the point at infinity: tracked as flags, exact), repeated points inside an ordinary query — with the scalars a witness
    has (mostly 0 / 1, some small values).  The oracle's sum every time, and the resident object keeps working"""
    n = 1200 if group == 1 else 480; g = o.SplitMix64(900 + group); base = (o.g1_consecutive if group == 1 else o.g2_consecutive)(g.field(), n); half = base.shape[1] // 2
    ref = (lambda P, K: o.msm_g1(P, K, mixed=True)) if group == 1 else (lambda P, K: o.msm_g2(P, K, mixed=True)); dec = o.g1_from if group == 1 else o.g2_from
    def neg(pt):
        r = pt.copy(); ys = o.from_arr(pt[half:].reshape(-1, 4)); r[half:] = o.to_arr([(o.Q_MOD - y) % o.Q_MOD for y in ys]).reshape(-1); return r
    def run(P, K):
        m = e.ResidentMsm(group, P, 8, filter_ones=True); m.set_scalars(K); assert dec(m.run())[0] == ref(P, K)
        K2 = witness_like_scalars(31, len(P)); m.set_scalars(K2); assert dec(m.run())[0] == ref(P, K2); m.close()
    ones = o.to_arr([1] * n); small = o.to_arr([5] * n); wl = witness_like_scalars(30 + group, n)
    same = np.repeat(base[:1], n, axis=0); run(same, ones); run(same, small); run(same, wl)                  # every point equal
    P = base.copy()
    for i in range(0, n, 2): P[i + 1] = neg(P[i])
    m = e.ResidentMsm(group, P, 8, filter_ones=True); m.set_scalars(ones); assert dec(m.run())[0] is None and ref(P, ones) is None; m.close()   # P, -P, all ones: zero
    run(P, small); run(P, wl)
    P = base.copy(); P[100:200] = P[99]; P[300:340] = P[7]                                                  # repeated points inside an ordinary query
    for i in range(400, 440, 2): P[i + 1] = neg(P[i])
    run(P, ones); run(P, wl)

def test_msm_degenerate_inputs():
    assert o.g1_from(e.msm(1, np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64)))[0] is None         # empty
    P = o.g1_consecutive(5, 64); assert o.g1_from(e.msm(1, P, np.zeros((64, 4), np.uint64), filter_ones=True))[0] is None   # all-zero scalars
    ones = o.to_arr([1] * 64); assert o.g1_from(e.msm(1, P, ones, filter_ones=True))[0] == o.msm_g1(P, ones, mixed=True)     # all ones
    same = np.repeat(P[:1], 64, axis=0); K = rand_field_arr(9, 64)                                                           # all points equal: every bucket hit doubles
    assert o.g1_from(e.msm(1, same, K, 7))[0] == o.g1_op("mul", o.g1_from(P[0])[0], k=sum(o.from_arr(K)) % o.R_MOD)

def test_msm_linearity_at_scale():
    """2^18-point H-query-sized MSM: MSM(k) + MSM(k') == MSM(k + k') and MSM over bases (b0+i)G == (sum k_i (b0+i)) G"""
    n = 1 << 18; b0 = 12345; P = o.g1_consecutive(b0, n); rng = np.random.default_rng(5)
    K1 = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); K2 = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); K1[:, 3] >>= 2; K2[:, 3] >>= 2
    def to_ints(K): return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in K]
    k1, k2 = to_ints(K1), to_ints(K2); m = e.ResidentMsm(1, P, 0, False)
    def run(K): m.set_scalars(K); return o.g1_from(m.run())[0]
    r1, r2 = run(K1), run(K2); G = o.g1_gen()
    s1 = sum(k * (b0 + i) for i, k in enumerate(k1)) % o.R_MOD; assert r1 == o.g1_op("mul", G, k=s1)
    K3 = o.to_arr([(a + b) % o.R_MOD for a, b in zip(k1, k2)]); assert run(K3) == o.g1_op("add", r1, r2)
    m.close()

@pytest.mark.parametrize("m", [2, 4, 16, 24, 48, 64, 80, 1024, 1536, 2048, 4096, 5120, 1 << 13, (1 << 14) + (1 << 12), 1 << 16])
def test_domain_transforms_match_oracle(m):
    assert e.domain_size(m) == o.domain_size(m); a = rand_field_arr(40 + m, o.domain_size(m))
    for op in ("fft", "ifft", "cosetfft", "icosetfft"):
        assert np.array_equal(e.domain_transform(m, op, a), o.domain_op(op, m, a)), (m, op)

@pytest.mark.parametrize("m", [1 << 18, (1 << 17) + (1 << 16)])
def test_domain_roundtrip_at_scale(m):
    """send-sized (2^18) and mint-sized (196,608, step) domains: iFFT(FFT(a)) == a, icosetFFT(cosetFFT(a)) == a"""
    a = np.random.default_rng(m).integers(0, 1 << 62, size=(m, 4), dtype=np.uint64); a[:, 3] >>= 2
    assert np.array_equal(e.domain_transform(m, "ifft", e.domain_transform(m, "fft", a)), a)
    assert np.array_equal(e.domain_transform(m, "icosetfft", e.domain_transform(m, "cosetfft", a)), a)

def test_domain_beyond_two_pass_tiles():
    """2^23 points take the stage-per-launch path: FFT of the unit impulse at index 1 is the table of powers of the root of unity, and the round trip is exact"""
    m = 1 << 23; a = np.zeros((m, 4), dtype=np.uint64); a[1, 0] = 1; f = e.domain_transform(m, "fft", a)
    w = pow(pow(5, (o.R_MOD - 1) >> 28, o.R_MOD), 1 << 5, o.R_MOD)      # alt_bn128_init.cpp:112-116: Fr::root_of_unity = 5^((r-1)/2^28), squared down to order 2^23
    for k in (0, 1, 2, 12345, m // 2, m - 1): assert o.from_arr(f[k:k + 1])[0] == pow(w, k, o.R_MOD), k
    assert np.array_equal(e.domain_transform(m, "ifft", f), a)

@pytest.mark.parametrize("seed,ni,nv,nc", [(7, 3, 40, 60), (8, 4, 30, 40), (21, 5, 600, 1019), (22, 2, 900, 1400)])
def test_witness_map_matches_oracle(seed, ni, nv, nc):
    cs, z = random_r1cs(seed, ni, nv, nc); dev = e.R1cs(ni, nv, nc, cs.rowptr, cs.col, cs.coeff)
    assert np.array_equal(dev.witness_map(z), o.witness_map(cs, z))
    z2 = z.copy(); z2[nv - 1, 0] ^= 1
    if not o.r1cs_is_satisfied(cs, z2):
        with pytest.raises(e.ZkGpuError): dev.witness_map(z2)
    dev.close()
