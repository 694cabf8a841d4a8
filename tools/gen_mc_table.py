"""Generates include/mmf_mc_table.h: the marching-cubes triangle table used by the mesh-topology kernels and the CPU oracle.

The table is derived, not copied: for each of the 256 inside/outside corner patterns the cut edges of every cube face are
joined into segments (two cut edges: one segment; four cut edges -- the ambiguous face -- each INSIDE corner of the face is
cut off on its own), the segments chain into closed loops over the cube surface, and every loop is fan-triangulated with
its normal pointing from inside (negative distance) to outside.  The face rule depends only on the face's own corner
signs, so two cubes sharing a face agree on its segments and the surface is watertight.  Conventions:
  corner c = dx*4 + dy*2 + dz, bit c of the pattern set iff distance(corner) < 0;
  cube edge e = a*4 + s1*2 + s2: along axis a, starting at the corner with offset s1 on axis (a+1)%3 and s2 on axis (a+2)%3.
Running the script also self-checks the table on sampled implicit surfaces (closed, manifold, consistently oriented).
"""
import itertools
import os
import sys

import numpy as np


def corner_id(d):
    return d[0] * 4 + d[1] * 2 + d[2]


def edge_ends(e):
    a, s1, s2 = e // 4, (e // 2) & 1, e & 1
    p = [0, 0, 0]
    p[(a + 1) % 3] = s1
    p[(a + 2) % 3] = s2
    q = list(p)
    q[a] = 1
    return tuple(p), tuple(q)


EDGE_OF = {}
for e in range(12):
    p, q = edge_ends(e)
    EDGE_OF[(p, q)] = e
    EDGE_OF[(q, p)] = e


def faces():
    out = []
    for f in range(3):
        for side in (0, 1):
            u, v = (f + 1) % 3, (f + 2) % 3
            cyc = []
            for (du, dv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[f], p[u], p[v] = side, du, dv
                cyc.append(tuple(p))
            out.append(cyc)
    return out


FACES = faces()


def build(pattern):
    inside = lambda p: (pattern >> corner_id(p)) & 1
    cut = [e for e in range(12) if inside(edge_ends(e)[0]) != inside(edge_ends(e)[1])]
    nbr = {e: [] for e in cut}
    for cyc in FACES:
        fe = [EDGE_OF[(cyc[i], cyc[(i + 1) % 4])] for i in range(4)]  # edge i joins corner i and i+1
        active = [i for i in range(4) if fe[i] in nbr]
        if len(active) == 2:
            a, b = fe[active[0]], fe[active[1]]
            nbr[a].append(b)
            nbr[b].append(a)
        elif len(active) == 4:
            for i in range(4):  # corner i sits between face edges i-1 and i
                if inside(cyc[i]):
                    a, b = fe[(i - 1) % 4], fe[i]
                    nbr[a].append(b)
                    nbr[b].append(a)
    assert all(len(v) == 2 for v in nbr.values()), (pattern, nbr)
    loops, seen = [], set()
    for e in cut:
        if e in seen:
            continue
        loop, prev, cur = [e], None, e
        seen.add(e)
        while True:
            nxt = [x for x in nbr[cur] if x != prev] if prev is not None else [nbr[cur][0]]
            if prev is not None and nbr[cur][0] == nbr[cur][1]:
                nxt = [nbr[cur][0]]
            n = nxt[0]
            if n == loop[0]:
                break
            loop.append(n)
            seen.add(n)
            prev, cur = cur, n
        loops.append(loop)
    tris = []
    for loop in loops:
        assert len(loop) >= 3, (pattern, loop)
        mid = [(np.array(edge_ends(e)[0], float) + np.array(edge_ends(e)[1], float)) / 2 for e in loop]
        normal = np.zeros(3)
        for i in range(len(loop)):
            normal += np.cross(mid[i], mid[(i + 1) % len(loop)])
        g = np.zeros(3)
        for e in loop:
            p, q = edge_ends(e)
            g += (np.array(q, float) - np.array(p, float)) * (1 if inside(p) else -1)  # inside -> outside
        if normal.dot(g) < 0:
            loop = loop[::-1]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    return tris


TABLE = [build(p) for p in range(256)]
MAXT = max(len(t) for t in TABLE)


def mesh_grid(D):
    """Triangles (global lattice-edge ids) of a scalar grid D[nx,ny,nz] with the table."""
    nx, ny, nz = D.shape
    tris = []
    for o in itertools.product(range(nx - 1), range(ny - 1), range(nz - 1)):
        pat = 0
        for d in itertools.product((0, 1), repeat=3):
            if D[o[0] + d[0], o[1] + d[1], o[2] + d[2]] < 0:
                pat |= 1 << corner_id(d)
        for t in TABLE[pat]:
            ids = []
            for e in t:
                p, _ = edge_ends(e)
                q = (o[0] + p[0], o[1] + p[1], o[2] + p[2])
                ids.append(((q[0] * ny + q[1]) * nz + q[2]) * 3 + e // 4)
            tris.append(tuple(ids))
    return tris


def self_check():
    rng = np.random.default_rng(0)
    n = 14
    g = np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).astype(float)
    fields = {
        "sphere": np.linalg.norm(g - 6.3, axis=-1) - 4.1,
        "two blobs": np.minimum(np.linalg.norm(g - np.array([4.2, 4.4, 4.1]), axis=-1) - 2.6, np.linalg.norm(g - np.array([9.1, 8.7, 9.3]), axis=-1) - 3.2),
        "torus": np.sqrt((np.sqrt((g[..., 0] - 6.5) ** 2 + (g[..., 1] - 6.5) ** 2) - 4.0) ** 2 + (g[..., 2] - 6.5) ** 2) - 1.6,
        "noise": None,
    }
    noise = rng.standard_normal((n, n, n))
    noise[0], noise[-1], noise[:, 0], noise[:, -1], noise[:, :, 0], noise[:, :, -1] = 1, 1, 1, 1, 1, 1  # closed: outside on the border
    fields["noise"] = noise
    for name, D in fields.items():
        tris = mesh_grid(D)
        directed = {}
        for t in tris:
            assert len(set(t)) == 3, (name, t)
            for i in range(3):
                k = (t[i], t[(i + 1) % 3])
                directed[k] = directed.get(k, 0) + 1
        for (a, b), c in directed.items():
            # closed + consistently oriented: every directed edge is matched by its reverse, equally often (pinched
            # vertices of ambiguous patterns may stack two sheets on one lattice edge: count 2 both ways)
            assert directed.get((b, a), 0) == c, (name, (a, b), c, directed.get((b, a), 0))
        V = len({v for t in tris for v in t})
        E = len({tuple(sorted(k)) for k in directed})
        print(f"  {name:10s} V={V} E={E} F={len(tris)} euler={V - E + len(tris)}")
        if name == "sphere":
            assert V - E + len(tris) == 2
        if name == "torus":
            assert V - E + len(tris) == 0


def write_header(path):
    with open(path, "w") as f:
        f.write("/* mmf_mc_table.h -- GENERATED by tools/gen_mc_table.py (marching-cubes triangles per corner pattern); do not edit.\n"
                " * corner c = dx*4+dy*2+dz, bit c set iff distance < 0; cube edge e = a*4 + s1*2 + s2 (axis a, offsets on axes\n"
                " * (a+1)%3, (a+2)%3); triangles are wound so that normals point from negative to positive distance.\n"
                " * Define MMF_MC_QUAL before including (e.g. `static const` in C, `static __device__ const` in HIP). */\n")
        f.write("#ifndef MMF_MC_TABLE_H\n#define MMF_MC_TABLE_H\n#ifndef MMF_MC_QUAL\n#define MMF_MC_QUAL static const\n#endif\n")
        f.write(f"#define MMF_MC_MAX_TRIS {MAXT}\n")
        f.write("MMF_MC_QUAL unsigned char mmf_mc_num_tris[256] = {\n")
        for r in range(0, 256, 32):
            f.write("    " + ", ".join(str(len(TABLE[p])) for p in range(r, r + 32)) + ",\n")
        f.write("};\n")
        f.write(f"MMF_MC_QUAL signed char mmf_mc_tris[256][{3 * MAXT}] = {{\n")
        for p in range(256):
            flat = [e for t in TABLE[p] for e in t]
            flat += [-1] * (3 * MAXT - len(flat))
            f.write("    {" + ", ".join(f"{e:2d}" for e in flat) + "},\n")
        f.write("};\n#endif\n")


if __name__ == "__main__":
    print(f"max triangles per cube: {MAXT}; total triangles over the 256 patterns: {sum(len(t) for t in TABLE)}")
    self_check()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mmf_mc_table.h")
    write_header(sys.argv[1] if len(sys.argv) > 1 else out)
    print("wrote", sys.argv[1] if len(sys.argv) > 1 else out)
