#!/usr/bin/env python3
"""Equal points inside a proving key's queries (A, B1, L): groups of indices that hold the same affine point (infinity apart).  python tools/dup_points.py [send|mint|redeem|deposit]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from blockmaze_amd import engine as e
from oracle import pyoracle as o
import numpy as np
kind = sys.argv[1] if len(sys.argv) > 1 else "send"; tmp = tempfile.mkdtemp(); pkp, vkp = os.path.join(tmp, "pk.txt"), os.path.join(tmp, "vk.txt"); e.keygen(kind, pkp, vkp, seed=99); os.environ["ZK_KEY_CACHE"] = "0"
pk, cs = o.parse_pk(pkp)
def dups(name, pts, index=None):
    pts = np.ascontiguousarray(pts); v = pts.view([("p", pts.dtype, pts.shape[1])]).ravel(); nz = (pts != 0).any(axis=1); _, inv, cnt = np.unique(v, return_inverse=True, return_counts=True)
    groups = {}
    for i in np.nonzero(nz & (cnt[inv] > 1))[0]: groups.setdefault(int(inv[i]), []).append(int(index[i]) if index is not None else int(i))
    print("%s %s: %d points, %d at infinity, %d groups of equal points%s" % (kind, name, len(pts), int((~nz).sum()), len(groups), (": " + str(sorted(groups.values())[:6])) if groups else ""))

def same_x(name, pts, index=None):
    """groups of indices whose points share the x coordinate: equal points and pairs P, -P (the incomplete additions of the fast paths meet both the same way)"""
    pts = np.ascontiguousarray(pts); h = pts.shape[1] // 2; xs = np.ascontiguousarray(pts[:, :h]); v = xs.view([("p", xs.dtype, h)]).ravel(); nz = (pts != 0).any(axis=1)
    _, inv, cnt = np.unique(v, return_inverse=True, return_counts=True); groups = {}
    for i in np.nonzero(nz & (cnt[inv] > 1))[0]: groups.setdefault(int(inv[i]), []).append(int(i))
    opposite = [sorted((int(index[i]) if index is not None else i) for i in g) for g in groups.values() if len({pts[i].tobytes() for i in g}) > 1]
    print("%s %s: %d groups share an x coordinate, %d of them hold a point AND its negative%s" % (kind, name, len(groups), len(opposite), (": " + str(sorted(opposite)[:6])) if opposite else ""))
dups("A", pk.A); dups("B_g1", pk.B_g1, pk.B_idx); dups("L (variables after the inputs)", pk.L)
same_x("A", pk.A); same_x("B_g1", pk.B_g1, pk.B_idx); same_x("L (variables after the inputs)", pk.L)
