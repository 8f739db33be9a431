"""Minimal OBJ reader / procedural meshes for tests and the benchmark scene (the C ABI has its own C++ OBJ/STL loader:
Testbed::load_mesh, reference src/testbed_geometry_training.cu:2786-2866)."""
import numpy as np


def load_obj(path):
    """Returns float32 (n_tris, 3, 3); polygons are fan-triangulated."""
    verts, tris = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    tris.append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(verts, np.float32)
    return v[np.asarray(tris, np.int64)].astype(np.float32)


def save_obj(path, tris):
    tris = np.asarray(tris, np.float32).reshape(-1, 3, 3)
    with open(path, "w") as f:
        for t in tris:
            for v in t:
                f.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        for i in range(tris.shape[0]):
            f.write(f"f {3 * i + 1} {3 * i + 2} {3 * i + 3}\n")


def icosphere(subdiv=3, radius=1.0):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    vv = np.asarray(v, np.float32) * radius
    return vv[np.asarray(f, np.int64)].astype(np.float32)


def torus(n_major=48, n_minor=24, R=1.0, r=0.35):
    u = np.linspace(0, 2 * np.pi, n_major, endpoint=False)
    w = np.linspace(0, 2 * np.pi, n_minor, endpoint=False)
    U, W = np.meshgrid(u, w, indexing="ij")
    P = np.stack([(R + r * np.cos(W)) * np.cos(U), r * np.sin(W), (R + r * np.cos(W)) * np.sin(U)], -1)
    tris = []
    for i in range(n_major):
        for j in range(n_minor):
            a, b = P[i, j], P[(i + 1) % n_major, j]
            c, d = P[(i + 1) % n_major, (j + 1) % n_minor], P[i, (j + 1) % n_minor]
            tris += [[a, b, c], [a, c, d]]
    return np.asarray(tris, np.float32)
