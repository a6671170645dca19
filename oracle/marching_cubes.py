"""numpy restatement of the marching-cubes scene-mesh extraction — TEST INFRASTRUCTURE (oracle).

The reference calls skimage.measure.marching_cubes(tsdf_vol, level=0) (utils.py:225-229,232-241).  skimage is not
vendored and not installed here, and its default (Lewiner) case tables cannot be read, so this restates the classic
construction the HIP kernels implement (csrc/marching_cubes.hip): one vertex per sign-changing grid edge at the linear
zero crossing; faces from a 256-case table generated from ONE rule (segments on cube faces, inside corners cut off on
ambiguous faces, loops fanned).  PARITY UNPINNED against skimage for the face list; the mesh is pinned by its
properties (watertight, on the level set) in tests/test_marching_cubes.py.  Written independently of the C++ generator
(different data structures) so that comparing the two tables is a real check.
"""
import numpy as np

F32 = np.float32


def corner_xyz(i):
    return (i & 1, (i >> 1) & 1, i >> 2)


def edge_of(c0, c1):
    """id of the cube edge joining corners c0, c1: axis * 4 + (b + 2 c), (b, c) = coordinates on the other two axes
    in ascending axis order"""
    a, b = corner_xyz(c0), corner_xyz(c1)
    axis = [k for k in range(3) if a[k] != b[k]]
    assert len(axis) == 1
    axis = axis[0]
    others = [k for k in range(3) if k != axis]
    return axis * 4 + a[others[0]] + 2 * a[others[1]]


def build_table():
    """-> int8[256,16] edge triples, -1 terminated"""
    table = np.full((256, 16), -1, np.int8)
    faces = []
    for f in range(3):
        for s in range(2):
            u, w = (f + 1) % 3, (f + 2) % 3
            ring = []
            for cu, cw in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[f], p[u], p[w] = s, cu, cw
                ring.append(p[0] + 2 * p[1] + 4 * p[2])
            faces.append(ring)
    most = 0
    for cs in range(256):
        inside = [(cs >> k) & 1 for k in range(8)]
        links = {}
        for ring in faces:
            edges = [edge_of(ring[k], ring[(k + 1) % 4]) for k in range(4)]
            cut = [inside[ring[k]] != inside[ring[(k + 1) % 4]] for k in range(4)]
            if sum(cut) == 2:
                a, b = [edges[k] for k in range(4) if cut[k]]
                links.setdefault(a, []).append(b)
                links.setdefault(b, []).append(a)
            elif sum(cut) == 4:
                for k in range(4):
                    if inside[ring[k]]:
                        a, b = edges[(k + 3) % 4], edges[k]
                        links.setdefault(a, []).append(b)
                        links.setdefault(b, []).append(a)
        seen, tris = set(), []
        for e0 in range(12):
            if e0 in seen or e0 not in links:
                continue
            loop, prev, cur = [], None, e0
            while True:
                loop.append(cur)
                seen.add(cur)
                nx = links[cur][0] if links[cur][0] != prev else links[cur][1]
                prev, cur = cur, nx
                if cur == e0:
                    break
            for k in range(1, len(loop) - 1):
                tris += [loop[0], loop[k], loop[k + 1]]
        most = max(most, len(tris) // 3)
        assert len(tris) <= 15, (cs, len(tris))
        table[cs, :len(tris)] = tris
    return table, most


def marching_cubes(vol, level=0.0, table=None):
    """-> verts f32[N,3] (voxel coordinates), faces int32[M,3] (oriented along the field gradient), in the raster order
    of the HIP kernels.  Pure Python loops: small volumes only."""
    vol = np.asarray(vol, F32)
    if table is None:
        table = build_table()[0]
    dx, dy, dz = vol.shape
    inside = vol < F32(level)
    vid, verts = {}, []
    for x in range(dx):
        for y in range(dy):
            for z in range(dz):
                for axis in range(3):
                    q = [x, y, z]
                    q[axis] += 1
                    if q[0] >= dx or q[1] >= dy or q[2] >= dz or inside[x, y, z] == inside[tuple(q)]:
                        continue
                    v0, v1 = vol[x, y, z], vol[tuple(q)]
                    t = F32(F32(level) - v0) / F32(v1 - v0)
                    p = [F32(x), F32(y), F32(z)]
                    p[axis] = F32(p[axis] + t)
                    vid[(x, y, z, axis)] = len(verts)
                    verts.append(p)
    verts = np.array(verts, F32).reshape(-1, 3)
    faces = []
    for x in range(dx - 1):
        for y in range(dy - 1):
            for z in range(dz - 1):
                val = [vol[x + (k & 1), y + ((k >> 1) & 1), z + (k >> 2)] for k in range(8)]
                cs = sum((1 << k) for k in range(8) if val[k] < F32(level))
                row = table[cs]
                if row[0] < 0:
                    continue
                g = np.array([F32(0.25) * ((val[1] + val[3] + val[5] + val[7]) - (val[0] + val[2] + val[4] + val[6])),
                              F32(0.25) * ((val[2] + val[3] + val[6] + val[7]) - (val[0] + val[1] + val[4] + val[5])),
                              F32(0.25) * ((val[4] + val[5] + val[6] + val[7]) - (val[0] + val[1] + val[2] + val[3]))], F32)
                for t in range(5):
                    if row[3 * t] < 0:
                        break
                    tri = []
                    for e in row[3 * t: 3 * t + 3]:
                        axis, b, c = int(e) >> 2, int(e) & 1, (int(e) >> 1) & 1
                        q = [x, y, z]
                        others = [k for k in range(3) if k != axis]
                        q[others[0]] += b
                        q[others[1]] += c
                        tri.append(vid[(q[0], q[1], q[2], axis)])
                    a, b_, c_ = (verts[i] for i in tri)
                    u, w = (b_ - a).astype(F32), (c_ - a).astype(F32)
                    nrm = np.array([u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]], F32)
                    if F32(nrm[0] * g[0] + nrm[1] * g[1] + nrm[2] * g[2]) < 0:
                        tri[1], tri[2] = tri[2], tri[1]
                    faces.append(tri)
    return verts, np.array(faces, np.int32).reshape(-1, 3)
