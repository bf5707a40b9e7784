"""Which constraints of the four circuits are REFERENCE-COMPILED, and which are read off the reference's constructors (round 5).

The reference's circuit compositions (src/{mint,send,deposit,redeem}/circuit/gadget.tcc) cannot be compiled in this image — circuit/utils.tcc needs BOOST_FOREACH —, but
every gadget they are made of can, and was: tests/golden holds canonical R1CS hashes and witnesses of libsnark's multipacking_gadget (the public-input unpacker), of
BlockMaze's less_comparison_gadget (comparison.tcc compiled for real), of the CMTA / CMTS / PRF / CRH hashers composed from libsnark's compression gadget exactly like
commitment.tcc composes them, and of libsnark's merkle_tree_check_read_gadget (oracle/ref_harness.cpp, oracle/make_golden.py).  test_circuits_cpu.py compares this
engine's stand-alone blocks with those dumps.  HERE every such block is located INSIDE the full circuits: the rows [pos, pos + n) of the exported mint / redeem / send /
deposit constraint system must equal the reference-compiled block's rows term by term under an explicit variable substitution sigma (block variable -> circuit variable,
the wiring read off the constructor: which note field feeds which hasher, where the block's own variables start).  What is left — booleanity rows of the note fields,
`equal`, `ZERO = 0` — is asserted row by row, and the test computes the share of each circuit that is reference-compiled: above 95 % for all four.

Sources: send/circuit/gadget.tcc:80-225, mint/circuit/gadget.tcc:71-193, redeem/circuit/gadget.tcc:70-175, deposit/circuit/gadget.tcc:88-234, note.tcc, commitment.tcc,
less_cmp.tcc / add_cmp.tcc / sub_cmp.tcc, merkle.tcc."""
import json, os
import numpy as np
import pytest
from oracle import pyoracle as o
from blockmaze_amd import engine as e
from test_circuits_cpu import canonical_hash, _bool_var

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
J = lambda n: json.load(open(os.path.join(GOLD, n)))
SHA_V, SHA_C = 24792, 27280                                                            # one compression gadget: variables / constraints

@pytest.fixture(scope="module")
def blocks(tmp_path_factory):
    """the stand-alone blocks as this engine builds them, each checked against the reference-compiled dump's canonical hash on the spot"""
    t = tmp_path_factory.mktemp("blocks"); out = {}
    def load(name, kind, arg, gold):
        p = str(t / (name + ".bin")); e.circuit_export(kind, p, arg); cs = o.R1CS.load(p); assert canonical_hash(cs) == gold["canonical_r1cs_sha256"] and cs.n_cons == gold["constraints"], name; out[name] = cs
    up = J("unpacker_gadget.json"); hb = J("hash_blocks.json")
    for nbits in (832, 1024, 1440): load("unpacker%d" % nbits, "unpacker", nbits, up["bits%d" % nbits])
    load("lesscmp", "lesscmp", 8, J("lesscmp_gadget.json")); load("cmta", "cmta", 8, J("cmta_gadget.json")); load("merkle8", "merkle", 8, J("merkle_gadget.json")["depth8"])
    for k in ("cmts", "prf", "crh"): load(k, k, 8, hb[k])
    return out

def sigma_of(blk, *ranges):
    """substitution array: (first block variable, count, first circuit variable) triples; ONE stays ONE; anything else is unmapped (-1)"""
    s = np.full(blk.n_vars + 1, -1, dtype=np.int64); s[0] = 0
    for b0, n, c0 in ranges: s[b0:b0 + n] = np.arange(c0, c0 + n)
    return s
def _sorted_terms(cs, m, r0, n, sigma=None):
    a0, a1 = int(cs.rowptr[m][r0]), int(cs.rowptr[m][r0 + n]); rows = np.repeat(np.arange(n), np.diff(cs.rowptr[m][r0:r0 + n + 1].astype(np.int64))); col = cs.col[m][a0:a1].astype(np.int64)
    if sigma is not None: col = sigma[col]
    order = np.lexsort((col, rows)); return rows[order], col[order], cs.coeff[m][a0:a1][order]
def differing_rows(full, pos, blk, r0, n, sigma):
    """block rows [r0, r0 + n) under sigma against circuit rows [pos, pos + n): the (block-relative) rows that differ in any of the three matrices"""
    bad = set()
    for m in range(3):
        la = np.diff(blk.rowptr[m][r0:r0 + n + 1].astype(np.int64)); lb = np.diff(full.rowptr[m][pos:pos + n + 1].astype(np.int64)); bad |= set(np.nonzero(la != lb)[0].tolist())
        if not np.array_equal(la, lb): continue
        ra, ca, va = _sorted_terms(blk, m, r0, n, sigma); rb, cb, vb = _sorted_terms(full, m, pos, n); diff = (ca != cb) | (va != vb).any(axis=1) | (ca < 0); bad |= set(ra[diff].tolist())
    return sorted(bad)
def row(cs, i):
    """constraint i as three {variable: coefficient} dicts"""
    return [{int(cs.col[m][k]): o.from_arr(cs.coeff[m][k:k + 1])[0] for k in range(int(cs.rowptr[m][i]), int(cs.rowptr[m][i + 1]))} for m in range(3)]

class Walk:
    """walks a circuit's constraints in emission order, counting what was matched against a reference-compiled block and what was read off"""
    def __init__(self, cs, blocks): self.cs, self.blocks, self.pos, self.compiled, self.readoff, self.log = cs, blocks, 0, 0, 0, []
    def block(self, name, r0, n, sigma, what, tolerate=()):
        blk = self.blocks[name]; n = blk.n_cons - r0 if n is None else n; bad = differing_rows(self.cs, self.pos, blk, r0, n, sigma); assert bad == list(tolerate), (what, self.pos, bad[:8])
        self.log.append((self.pos, n, "compiled", what)); self.pos += n; self.compiled += n - len(tolerate); self.readoff += len(tolerate)
    def bools(self, vars_, what):
        vars_ = list(vars_); got = [_bool_var(self.cs, self.pos + k) for k in range(len(vars_))]; assert got == vars_, (what, self.pos); self.log.append((self.pos, len(vars_), "read-off", what)); self.pos += len(vars_); self.readoff += len(vars_)
    def exact(self, A, B, C, what):
        assert row(self.cs, self.pos) == [A, B, C], (what, self.pos, row(self.cs, self.pos)); self.log.append((self.pos, 1, "read-off", what)); self.pos += 1; self.readoff += 1
    def hasher(self, name, sigma_in, inter_first, what):
        """a CMTA / CMTS / PRF hasher (intermediate digest's booleanity, two compressions) or CRH (one compression): all rows of the block but its first (ZERO = 0, which the circuits emit once, elsewhere)"""
        blk = self.blocks[name]; n_in = blk.n_vars - (SHA_V if name == "crh" else 2 * SHA_V + 256) - 1 - 256; first_own = 1 + 1 + n_in + 256            # ZERO, inputs, output digest, then the block's own variables
        s = sigma_of(blk, *sigma_in, (first_own, blk.n_vars + 1 - first_own, inter_first)); self.block(name, 1, None, s, what)
    def done(self, n_cons, min_share):
        assert self.pos == self.cs.n_cons == n_cons and self.compiled + self.readoff == n_cons; share = self.compiled / n_cons; assert share >= min_share, share; return share
R = lambda a, n: range(a, a + n)
MINUS1 = o.R_MOD - 1

def test_send_constraints_are_reference_compiled_blocks(blocks, tmp_path):
    """send/circuit/gadget.tcc:196-225 over the variables of SURVEY.md Appendix C"""
    p = str(tmp_path / "c.bin"); e.circuit_export("send", p); cs = o.R1CS.load(p); w = Walk(cs, blocks); ZERO = 1030
    cmtA_old, sn_old, cmtS, cmtA = 6, 262, 518, 774; value_old, r_old, value_s, pk_recv, pk_sender, r_s, value, sn, r, sk = 1031, 1095, 1351, 1415, 1575, 1735, 1991, 2055, 2311, 2567
    w.block("unpacker1024", 0, None, sigma_of(blocks["unpacker1024"], (1, 1029, 1)), "unpacker")                                                       # :198
    note = list(R(value_old, 64)) + list(R(value_s, 64)) + list(R(sn_old, 256)) + list(R(r_old, 256)) + list(R(pk_recv, 160)) + list(R(r_s, 256))
    w.bools(note, "lessCMP: note fields")                                                                                                              # note.tcc:40-62
    w.block("lesscmp", 128, None, sigma_of(blocks["lesscmp"], (129, 1, 2823), (130, 1, 2824), (131, 67, 2825)), "less_comparison_gadget")              # less_cmp.tcc:29-33 -> comparison.tcc
    w.bools(note + list(R(value, 64)) + list(R(sn, 256)) + list(R(r, 256)) + list(R(sk, 256)) + list(R(pk_sender, 160)), "noteSUB: note fields")
    w.exact({0: 1}, {2892: 1, 2893: MINUS1}, {2894: 1}, "1 * (value_old - value_s) = value")                                                           # note.tcc:130-132
    w.exact({0: 1}, {ZERO: 1}, {}, "ZERO = 0")                                                                                                         # gadget.tcc:205
    w.bools(R(r_s, 256), "r_s"); w.hasher("crh", [(1, 1, ZERO), (2, 160, pk_sender), (162, 256, r), (418, 256, r_s)], 2895, "CRH(pk_sender, r) -> r_s")
    w.bools(R(sn, 256), "sn"); w.hasher("prf", [(1, 1, ZERO), (2, 256, sk), (258, 256, r), (514, 256, sn)], 27687, "PRF(sk, r) -> sn")
    w.bools(list(R(sn_old, 256)) + list(R(cmtA_old, 256)), "sn_old, cmtA_old"); w.hasher("cmta", [(1, 1, ZERO), (2, 64, value_old), (66, 256, sn_old), (322, 256, r_old), (578, 256, cmtA_old)], 77527, "CMTA_old")
    w.bools(R(cmtS, 256), "cmtS"); w.hasher("cmts", [(1, 1, ZERO), (2, 64, value_s), (66, 160, pk_recv), (226, 256, r_s), (482, 256, sn_old), (738, 256, cmtS)], 127367, "CMTS")
    w.bools(R(cmtA, 256), "cmtA"); w.hasher("cmta", [(1, 1, ZERO), (2, 64, value), (66, 256, sn), (322, 256, r), (578, 256, cmtA)], 177207, "CMTA")
    share = w.done(252286, 0.95); print("send: %.2f %% of the constraints are reference-compiled blocks" % (100 * share))

@pytest.mark.parametrize("kind", ["mint", "redeem"])
def test_mint_redeem_constraints_are_reference_compiled_blocks(kind, blocks, tmp_path):
    """mint/circuit/gadget.tcc:165-193 + note.tcc:44-72 + add_cmp.tcc:23-29; redeem/circuit/gadget.tcc:151-175 + note.tcc:48-79 + sub_cmp.tcc:29-37"""
    redeem = kind == "redeem"; p = str(tmp_path / "c.bin"); e.circuit_export(kind, p); cs = o.R1CS.load(p); w = Walk(cs, blocks); ZERO = 837
    cmtA_old, sn_old, cmtA, value_s, value, value_old, sk, r, r_old, sn = 5, 261, 517, 773, 838, 902, 966, 1222, 1478, 1734; value_packed, value_old_packed, value_s_packed = 1990, 1991, 1992
    h0 = 2060 if redeem else 1993; step = 256 + 2 * SHA_V
    w.block("unpacker832", 0, None, sigma_of(blocks["unpacker832"], (1, 836, 1)), "unpacker")
    w.bools(list(R(value_old, 64)) + list(R(value_s, 64)) + list(R(value, 64)) + list(R(sk, 256)) + list(R(r, 256)) + list(R(r_old, 256)), "note fields")
    if redeem:
        w.bools(list(R(sn, 256)) + list(R(sn_old, 256)), "note: sn, sn_old")
        w.exact({0: 1}, {value_old_packed: 1, value_s_packed: MINUS1}, {value_packed: 1}, "1 * (value_old - value_s) = value")                        # sub_cmp.tcc:33-34
        w.block("lesscmp", 128, None, sigma_of(blocks["lesscmp"], (129, 1, value_old_packed), (130, 1, value_s_packed), (131, 67, 1993)), "less_comparison_gadget")
    else: w.exact({0: 1}, {value_old_packed: 1, value_s_packed: 1}, {value_packed: 1}, "1 * (value_old + value_s) = value")                             # add_cmp.tcc:27-28
    w.exact({0: 1}, {ZERO: 1}, {}, "ZERO = 0")
    w.bools(R(sn, 256), "sn"); w.hasher("prf", [(1, 1, ZERO), (2, 256, sk), (258, 256, r), (514, 256, sn)], h0, "PRF(sk, r) -> sn")
    w.bools(list(R(sn_old, 256)) + list(R(cmtA_old, 256)), "sn_old, cmtA_old"); w.hasher("cmta", [(1, 1, ZERO), (2, 64, value_old), (66, 256, sn_old), (322, 256, r_old), (578, 256, cmtA_old)], h0 + step, "CMTA_old")
    w.bools(R(cmtA, 256), "cmtA"); w.hasher("cmta", [(1, 1, ZERO), (2, 64, value), (66, 256, sn), (322, 256, r), (578, 256, cmtA)], h0 + 2 * step, "CMTA")
    share = w.done(167853 if redeem else 167270, 0.95); print("%s: %.2f %% of the constraints are reference-compiled blocks" % (kind, 100 * share))

def test_deposit_constraints_are_reference_compiled_blocks(blocks, tmp_path):
    """deposit/circuit/gadget.tcc:200-234 + note.tcc:64-103 + merkle.tcc:36-52"""
    p = str(tmp_path / "c.bin"); e.circuit_export("deposit", p); cs = o.R1CS.load(p); w = Walk(cs, blocks); value_enforce, ZERO = 1447, 1448
    rt, pk_recv, cmtB_old, sn_old, cmtB, sn_s = 7, 263, 423, 679, 935, 1191
    value_s, r_s, sn_A_old, cmtS, value_old, r_old, value, sn, r, sk = 1449, 1513, 1769, 2025, 2281, 2345, 2601, 2665, 2921, 3177; value_s_packed, value_old_packed, value_packed = 3433, 3434, 3435
    step = 256 + 2 * SHA_V; prf_sn, prf_sn_s, cmts_i, old_i, new_i = [3436 + k * step for k in range(5)]; positions = 3436 + 5 * step
    w.block("unpacker1440", 0, None, sigma_of(blocks["unpacker1440"], (1, 1446, 1)), "unpacker")
    w.bools(list(R(value_s, 64)) + list(R(value_old, 64)) + list(R(value, 64)), "note values")
    w.exact({0: 1}, {value_old_packed: 1, value_s_packed: 1}, {value_packed: 1}, "1 * (value_old + value_s) = value")
    w.bools(list(R(pk_recv, 160)) + list(R(r_s, 256)) + list(R(sn_A_old, 256)) + list(R(sn_old, 256)) + list(R(r_old, 256)) + list(R(sn, 256)) + list(R(r, 256)) + list(R(sk, 256)), "note digests")
    w.exact({0: 1}, {ZERO: 1}, {}, "ZERO = 0")
    w.bools(R(sn_s, 256), "sn_s"); w.hasher("prf", [(1, 1, ZERO), (2, 256, sk), (258, 256, r_s), (514, 256, sn_s)], prf_sn_s, "PRF(sk, r_s) -> sn_s")
    w.bools(R(sn, 256), "sn"); w.hasher("prf", [(1, 1, ZERO), (2, 256, sk), (258, 256, r), (514, 256, sn)], prf_sn, "PRF(sk, r) -> sn")
    w.bools(list(R(sn_old, 256)) + list(R(cmtS, 256)), "sn_old, cmtS"); w.hasher("cmts", [(1, 1, ZERO), (2, 64, value_s), (66, 160, pk_recv), (226, 256, r_s), (482, 256, sn_A_old), (738, 256, cmtS)], cmts_i, "CMTS")
    w.bools(R(cmtB_old, 256), "cmtB_old"); w.hasher("cmta", [(1, 1, ZERO), (2, 64, value_old), (66, 256, sn_old), (322, 256, r_old), (578, 256, cmtB_old)], old_i, "CMTB_old")
    w.bools(R(cmtB, 256), "cmtB"); w.hasher("cmta", [(1, 1, ZERO), (2, 64, value), (66, 256, sn), (322, 256, r), (578, 256, cmtB)], new_i, "CMTB")
    w.bools(list(R(rt, 256)) + [value_enforce] + list(R(positions, 8)), "rt, value_enforce, positions")
    # libsnark's merkle_tree_check_read_gadget: address bits = positions, leaf = cmtS, root = rt, the gadget's own variables follow the positions.  The dump was made with
    # read_successful = ONE (libsnark's self-test); deposit passes value_enforce (gadget.tcc:250): the rows of the final bit_vector_copy_gadget that carry it differ in that
    # one variable — checked below, counted as read off
    mk = blocks["merkle8"]; s = sigma_of(mk, (1, 8, positions), (9, 256, cmtS), (265, 256, rt), (521, mk.n_vars - 520, positions + 8)); pos0 = w.pos
    bad = differing_rows(cs, pos0, mk, 0, mk.n_cons, s); assert 0 < len(bad) <= 4 and min(bad) >= mk.n_cons - 8, bad
    for i in bad:
        got, ref = row(cs, pos0 + i), row(mk, i); ref = [{int(s[v]): c for v, c in d.items()} for d in ref]
        moved = [(m, c) for m in range(3) for v, c in ref[m].items() if v == 0 and got[m].get(value_enforce) == c and 0 not in got[m]]; assert len(moved) == 1, (i, got, ref)
        m, c = moved[0]; fixed = dict(ref[m]); del fixed[0]; fixed[value_enforce] = c; assert got[m] == fixed and all(got[k] == ref[k] for k in range(3) if k != m), (i, got, ref)
    w.block("merkle8", 0, None, s, "merkle_tree_check_read_gadget", tolerate=bad)
    share = w.done(503863, 0.95); print("deposit: %.2f %% of the constraints are reference-compiled blocks" % (100 * share))

# ---- one more number the compiled reference printed for all four circuits: libsnark's own estimate of the proving key's size ----------------------------------------
@pytest.mark.parametrize("kind,size_mb,n_cons,n_vars,n_inputs,m", [("mint", 25.35, 167270, 151512, 4, 196608), ("redeem", 25.36, 167853, 151579, 4, 196608), ("send", 36.94, 252286, 227046, 5, 262144), ("deposit", 74.39, 503863, 457127, 6, 524288)])
def test_proving_key_size_is_the_reference_s_estimate(kind, size_mb, n_cons, n_vars, n_inputs, m, tmp_path):
    """BASELINE.md §2.1, column "PK size (libsnark estimate)": r1cs_gg_ppzksnark_proving_key::size_in_bits() printed by the compiled reference's self-tests
    (r1cs_gg_ppzksnark.tcc:54-58 / print_size: G1 255 bits, G2 509, a B-query entry 64 + 509 + 255 — sparse_vector.tcc size_in_bits).  Besides the counts it depends on
    the number of variables with a non-zero B polynomial, i.e. on WHICH variables occur in the (A/B-swapped, r1cs.tcc:182-231) B matrix: 136,316 for send.
    Two decimals of a megabyte pin that number to +-50 for the three circuits whose layout is read off the constructors"""
    p = str(tmp_path / "c.bin"); e.circuit_export(kind, p); cs = o.R1CS.load(p); assert (cs.n_cons, cs.n_vars, cs.n_inputs, cs.domain_m) == (n_cons, n_vars, n_inputs, m)
    if cs.swap_ab_beneficial(): cs = cs.swapped()
    nB = len(np.unique(cs.col[1])); bits = 3 * 255 + 2 * 509 + (n_vars + 1) * 255 + nB * (64 + 509 + 255) + (m - 1) * 255 + (n_vars - n_inputs) * 255
    if kind == "send": assert nB == 136316
    assert round(bits / 8 / 1e6, 2) == size_mb, (nB, bits / 8 / 1e6)
