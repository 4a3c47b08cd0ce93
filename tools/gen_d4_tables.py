#!/usr/bin/env python3
"""Generates fractal-image-compression_amd/csrc/fic_d4_tables.h: the group-Fourier tables that let the VALU sweep get the
covariances of all 8 isometry copies of a range block from ONE set of n + n/2 integer products instead of 8 n.

The 8 isometries form the dihedral group D4 acting on the B x B pixel positions.  On every orbit of positions the 8 inner
products  dot_k = sum_pos r[s_k(pos)] * d[pos]  are a correlation on the group, which the irreducible representations of D4
(four 1-dimensional ones and the 2-dimensional one) diagonalise:  16 * dot_k = sum_acc C[k][acc] * T[acc],  T[acc] = sum over
the accumulator's slots of  U[slot] * V[slot],  where U (range side) and V (domain side) are signed sums of <= 8 pixels.
Everything is integer and exact.  This script builds the change of basis numerically from the kernel's own isometry
definition (iso_source / iso_inverse of fic_devfn.h), solves for the products that are needed, groups them into 8
accumulators, checks the closed form the kernel uses (4 bases +/- 4 E terms) against brute force on random blocks, and
writes the tables."""
import itertools
import os
import sys
from fractions import Fraction

import numpy as np


def iso_source(k, B, x, y):
    m = B - 1
    sx, sy = [(x, y), (y, m - x), (m - x, m - y), (m - y, x), (m - x, y), (x, m - y), (y, x), (m - y, m - x)][k]
    return sx + sy * B


def iso_inverse(k):
    return {1: 3, 3: 1}.get(k, k)


def perm(k, B):
    """s_k: copy_k[pos] = r[s_k(pos)]  (k_range_copies in fic_prep.hip)."""
    ki = iso_inverse(k)
    return np.array([iso_source(ki, B, pos % B, pos // B) for pos in range(B * B)])


def linear_part(k, B):
    """2x2 integer matrix of s_k acting on centred coordinates (the 2-dimensional irrep)."""
    c = (B - 1) / 2.0
    s = perm(k, B)
    def img(x, y):
        p = s[x + y * B]
        return np.array([p % B - c, p // B - c])
    # solve from two positions
    p1, p2 = np.array([0 - c, 0 - c]), np.array([1 - c, 0 - c])
    q1, q2 = img(0, 0), img(1, 0)
    ex = q2 - q1                      # image of unit x
    p3, q3 = np.array([0 - c, 1 - c]), img(0, 1)
    ey = q3 - q1
    L = np.stack([ex, ey], axis=1)
    assert np.allclose(L @ p1, q1) and np.allclose(L @ p3, q3)
    return np.rint(L).astype(int)


def build(B):
    n = B * B
    S = [perm(k, B) for k in range(8)]
    L = [linear_part(k, B) for k in range(8)]
    chi = {
        "A1": [1] * 8,
        "A2": [int(round(np.linalg.det(L[k]))) for k in range(8)],
        "B1": [1 if L[k][0, 1] == 0 else -1 for k in range(8)],
    }
    chi["B2"] = [chi["A2"][k] * chi["B1"][k] for k in range(8)]
    # orbits
    seen, orbits = set(), []
    for p in range(n):
        if p in seen:
            continue
        orb = sorted({int(S[k][p]) for k in range(8)})
        seen.update(orb)
        orbits.append(orb)
    # candidate coefficient functionals (rows over positions), orbit by orbit, dropping dependent ones
    rows, kinds = [], []
    for oi, orb in enumerate(orbits):
        p0 = orb[0]
        cand = []
        for name in ("A1", "A2", "B1", "B2"):
            v = np.zeros(n, int)
            for k in range(8):
                v[S[k][p0]] += chi[name][k]
            cand.append((name, v))
        for a in range(2):
            for b in range(2):
                v = np.zeros(n, int)
                for k in range(8):
                    v[S[k][p0]] += L[k][a, b]
                cand.append((f"E{a}{b}", v))
        sub = []
        for name, v in cand:
            if not v.any():
                continue
            test = np.array([r for r in sub] + [v], float)
            if np.linalg.matrix_rank(test) == len(sub) + 1:
                sub.append(v)
                rows.append(v)
                kinds.append((oi, name))
        assert len(sub) == len(orb), (orb, len(sub))
    Q = np.array(rows, float)
    assert Q.shape == (n, n) and np.linalg.matrix_rank(Q) == n
    Qi = np.linalg.inv(Q)
    # G^k = Q^-T P_k^T Q^-1 with (P_k r)[pos] = r[S_k[pos]]:  dot_k = (P_k r) . d = u^T G^k v,  u = Q r, v = Q d
    G = []
    for k in range(8):
        P = np.zeros((n, n))
        P[np.arange(n), S[k]] = 1.0
        Gk = Qi.T @ P.T @ Qi
        G16 = np.rint(Gk * 16 * 8)            # denominators divide 128 at most; reduce below
        assert np.allclose(G16, Gk * 128, atol=1e-6)
        G.append(G16.astype(int))
    G = np.array(G)                            # [8][n][n], = 128 * G^k
    nz = np.argwhere(np.any(G != 0, axis=0))
    prods = []
    for a, b in nz:
        c = G[:, a, b]
        g = np.gcd.reduce(np.abs(c))
        prods.append((int(a), int(b), tuple((c // g).tolist()), int(g)))
    # every product's scale g must make 16*dot integer: 128*G = g*pattern -> weight on the domain side = g/8 (check integrality)
    for a, b, pat, g in prods:
        assert g % 8 == 0, (a, b, g)
    # group by pattern up to sign
    acc_of, acc_pat, slots = {}, [], []
    for a, b, pat, g in prods:
        key, sgn = pat, 1
        neg = tuple(-x for x in pat)
        if neg in acc_of:
            key, sgn = neg, -1
        if key not in acc_of:
            acc_of[key] = len(acc_pat)
            acc_pat.append(key)
        slots.append((acc_of[key], a, b, sgn * (g // 8)))
    return dict(B=B, n=n, Q=Q.astype(int), kinds=kinds, acc_pat=acc_pat, slots=slots, S=S, chi=chi, L=L)


def order_accumulators(t):
    """Canonical order [A1, A2, B1, B2, E0, E1, E2, E3] and the closed form 16*dot_k = base_i(k) + sigma_k * e_j(k)."""
    pats = [np.array(p) for p in t["acc_pat"]]
    assert len(pats) == 8, len(pats)
    one_d = [i for i, p in enumerate(pats) if np.all(p != 0)]
    e_acc = [i for i, p in enumerate(pats) if np.any(p == 0)]
    assert len(one_d) == 4 and len(e_acc) == 4
    def find(name):
        target = np.array(t["chi"][name])
        for i in one_d:
            if np.array_equal(pats[i], target):
                return i, 1
            if np.array_equal(pats[i], -target):
                return i, -1
        raise AssertionError(name)
    order, flip = [], []
    for name in ("A1", "A2", "B1", "B2"):
        i, s = find(name)
        order.append(i); flip.append(s)
    # E accumulators: pair them into (p, q) with the same support so that p+q and p-q are the terms
    supp = {}
    for i in e_acc:
        supp.setdefault(tuple((pats[i] != 0).tolist()), []).append(i)
    assert len(supp) == 2 and all(len(v) == 2 for v in supp.values()), supp
    for key in sorted(supp, reverse=True):
        order += supp[key]; flip += [1, 1]
    remap = {old: new for new, old in enumerate(order)}
    C = np.array([pats[old] * flip[new] for new, old in enumerate(order)]).T        # [8 k][8 acc]
    slots = [(remap[acc], a, b, w * flip[remap[acc]]) for acc, a, b, w in t["slots"]]
    # closed form: T0..T3 -> bases b0 = T0+T1+T2+T3, b1 = T0+T1-T2-T3, b2 = T0-T1+T2-T3, b3 = T0-T1-T2+T3
    base_sign = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, 1, -1], [1, -1, -1, 1]])
    e_form = np.array([[1, 1, 0, 0], [1, -1, 0, 0], [0, 0, 1, 1], [0, 0, 1, -1]])  # e0 = T4+T5, e1 = T4-T5, e2 = T6+T7, e3 = T6-T7
    form = []
    for k in range(8):
        hit = None
        for bi in range(4):
            if not np.array_equal(C[k, :4], base_sign[bi]):
                continue
            for ej in range(4):
                for sg in (1, -1):
                    if np.array_equal(C[k, 4:], sg * e_form[ej]):
                        hit = (bi, ej, sg)
        assert hit is not None, (k, C[k])
        form.append(hit)
    t = dict(t)
    t.update(C=C, slots=slots, form=form)
    return t


def pack(t):
    """Slots as dot2 pairs, each accumulator padded to an even slot count; the pairs are emitted ROUND-ROBIN over the
    accumulators (pair i of every accumulator before pair i+1 of any), so that consecutive v_dot2c in word order are
    independent and the kernel can consume a record front to back.  U/V functionals per slot."""
    Q = t["Q"]
    by_acc = [[] for _ in range(8)]
    for acc, a, b, w in t["slots"]:
        by_acc[acc].append((a, b, w))
    counts = []
    for acc in range(8):
        by_acc[acc].sort()
        if len(by_acc[acc]) % 2:
            by_acc[acc].append((-1, -1, 0))
        counts.append(len(by_acc[acc]) // 2)
    packed = []
    for i in range(max(counts)):
        for acc in range(8):
            if i < counts[acc]:
                packed += [(acc,) + by_acc[acc][2 * i], (acc,) + by_acc[acc][2 * i + 1]]
    U = np.zeros((len(packed), t["n"]), int)
    V = np.zeros((len(packed), t["n"]), int)
    for s, (acc, a, b, w) in enumerate(packed):
        if a >= 0:
            U[s] = Q[a]
            V[s] = Q[b] * w
    return packed, counts, U, V


def verify(t, packed, counts, U, V, trials=200):
    rng = np.random.default_rng(1)
    n, B = t["n"], t["B"]
    acc_of_slot = np.array([p[0] for p in packed])
    for _ in range(trials):
        r = rng.integers(0, 256, n)
        d = rng.integers(0, 256, n)
        u, v = U @ r, V @ d
        assert np.abs(u).max() < 32768 and np.abs(v).max() < 32768
        T = np.array([np.sum(u[acc_of_slot == a] * v[acc_of_slot == a]) for a in range(8)])
        b = [T[0] + T[1] + T[2] + T[3], T[0] + T[1] - T[2] - T[3], T[0] - T[1] + T[2] - T[3], T[0] - T[1] - T[2] + T[3]]
        e = [T[4] + T[5], T[4] - T[5], T[6] + T[7], T[6] - T[7]]
        for k in range(8):
            want = 16 * int(np.dot(r[t["S"][k]], d))
            bi, ej, sg = t["form"][k]
            assert b[bi] + sg * e[ej] == want, (k, b[bi] + sg * e[ej], want)
    # worst-case magnitudes: all-255 / all-0 patterns per functional
    umax = np.abs(U).sum(axis=1).max() * 255
    vmax = np.abs(V).sum(axis=1).max() * 255
    return int(umax), int(vmax)


def emit(tables, path):
    out = ["// fic_d4_tables.h -- GENERATED by tools/gen_d4_tables.py; do not edit.",
           "// Group-Fourier tables of the 8 isometries (D4) for k_sweep_d4: per block size the slot functionals U (range side)",
           "// and V (domain side) as (position, weight) lists, the accumulator of every dot2 pair, and the closed form",
           "//   16 * dot_k = base[form[k].b] + form[k].s * e[form[k].e],   base/e built from the 8 accumulators T0..T7:",
           "//   b0 = T0+T1+T2+T3  b1 = T0+T1-T2-T3  b2 = T0-T1+T2-T3  b3 = T0-T1-T2+T3   e0 = T4+T5  e1 = T4-T5  e2 = T6+T7  e3 = T6-T7.",
           "#pragma once", "#include <stdint.h>", ""]
    for t, packed, counts, U, V, umax, vmax in tables:
        B = t["B"]
        ns = len(packed)
        maxterms = max(int((U != 0).sum(axis=1).max()), int((V != 0).sum(axis=1).max()))
        out.append(f"// B = {B}: {ns} slots = {ns // 2} dot2 per (range, domain) pair instead of {8 * t['n'] // 4} dot4; |U| <= {umax}, |V| <= {vmax} (i16)")
        out.append(f"#define FIC_D4_B{B}_SLOTS {ns}")
        out.append(f"#define FIC_D4_B{B}_TERMS {maxterms}")
        accs = [packed[2 * w][0] for w in range(ns // 2)]
        assert all(packed[2 * w][0] == packed[2 * w + 1][0] for w in range(ns // 2))
        out.append(f"static constexpr int fic_d4_b{B}_acc[{ns // 2}] = {{{', '.join(str(c) for c in accs)}}};   // accumulator of dot2 pair w (round-robin order)")
        def table(name, M):
            rows = []
            for s in range(ns):
                idx = np.nonzero(M[s])[0]
                ent = [f"{{{int(p)}, {int(M[s][p])}}}" for p in idx] + ["{0, 0}"] * (maxterms - len(idx))
                rows.append("    {" + ", ".join(ent) + "}")
            out.append(f"static constexpr int16_t {name}[{ns}][{maxterms}][2] = {{\n" + ",\n".join(rows) + "\n};")
        table(f"fic_d4_b{B}_U", U)
        table(f"fic_d4_b{B}_V", V)
        out.append(f"static constexpr int fic_d4_b{B}_form[8][3] = {{" + ", ".join(f"{{{b}, {e}, {s}}}" for b, e, s in t["form"]) + "};   // k -> {base, e term, sign}")
        out.append("")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tables = []
    for B in (8,):          # B = 16 verifies too (384 slots) but its 192-dword record does not fit the SGPR file: not emitted
        t = order_accumulators(build(B))
        packed, counts, U, V = pack(t)
        umax, vmax = verify(t, packed, counts, U, V)
        print(f"B={B}: {len(packed)} slots, pairs per accumulator {counts}, |U|<={umax}, |V|<={vmax}, form {t['form']}")
        tables.append((t, packed, counts, U, V, umax, vmax))
    emit(tables, os.path.join(root, "fractal-image-compression_amd", "csrc", "fic_d4_tables.h"))
