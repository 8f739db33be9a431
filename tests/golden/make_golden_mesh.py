"""Generates tests/golden/mesh_{bunny,armadillo}_v1.npz: the reference's own meshes as data (data/geometry/objs/bunny.obj,
armadillo.obj: vertices + faces, not the file's text), seeded rays, and what the CPU oracle's mesh path answers for them --
BVH hit records (hit position, face normal) of mesh_raytrace_kernel (src/geometry_bvh.cu:646-676) over the BVH4 of
triangle_bvh.cu:425-508, and a small shaded frame with its sun shadow rays (render_geometry_mesh,
src/testbed_geometry_training.cu:2202-2320). SURVEY section 8(c), fixture (4).

The reference ships no expected outputs for this path (PARITY UNPINNED, oracle/orc_common.h): the vectors pin the oracle
against drift and give the GPU box a committed target. Needs /root/reference (this container only):
    python tests/golden/make_golden_mesh.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
OBJS = "/root/reference/data/geometry/objs"
CENTER = (0.0, 0.0, 0.0)
W, H = 96, 54
SHADE = dict(sun_dir=(0.3, 0.8, 0.5), roughness=0.4, metallic=0.1, sheen=0.2, clearcoat=0.3, clearcoat_gloss=0.6, subsurface=0.1,
             basecolor=(0.8, 0.6, 0.4), ambientcolor=(0.1, 0.1, 0.15))


def read_obj_indexed(path):
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, np.float32), np.asarray(faces, np.int32)


def rays(n, seed):
    """origins on a sphere of radius 1.6 around the normalised mesh (unit cube at CENTER), aimed at points inside it;
    a fifth of them aimed anywhere (misses, grazing hits)"""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c = np.asarray(CENTER) + 0.5
    o = c + 1.6 * d
    target = c + rng.uniform(-0.45, 0.45, (n, 3))
    target[::5] = c + rng.uniform(-2.0, 2.0, (n // 5 + (1 if n % 5 else 0), 3))[: target[::5].shape[0]]
    dirs = target - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    return o.astype(np.float32), dirs.astype(np.float32)


def main():
    import oracle as orc

    scene = importlib.import_module(PKG + ".scene")
    o = orc.Oracle()
    for name, n_rays in (("bunny", 8192), ("armadillo", 4096)):
        verts, faces = read_obj_indexed(os.path.join(OBJS, name + ".obj"))
        tris = verts[faces]
        h = o.mesh_scene([(tris, CENTER)])
        ro, rd = rays(n_rays, 17)
        hp, hn = o.trace_mesh(h, ro, rd)
        mat = scene.orbit_camera(35.0, 20.0, 7.0)
        focal = scene.focal_from_fov_x(W, 0.45)
        fb, db = o.render_mesh(h, o.make_camera(mat, W, H, focal), o.make_mesh_opts(**SHADE))
        lo, hi = o.mesh_scene_aabb(h)
        o.mesh_scene_destroy(h)
        hit = ~np.all(hn == rd, axis=1)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mesh_%s_v1.npz" % name)
        np.savez_compressed(path, verts=verts, faces=faces, ray_o=ro, ray_d=rd, hit_pos=hp, hit_normal=hn, scene_aabb=np.concatenate([lo, hi]),
                            cam_matrix=mat.astype(np.float32), focal=np.asarray(focal, np.float32), frame=fb.astype(np.float32), depth=db.astype(np.float32))
        print("wrote", path, os.path.getsize(path), "bytes;", tris.shape[0], "triangles,", int(hit.sum()), "of", n_rays, "rays hit,",
              "lit pixels", int((fb[..., 3] > 0).sum()))


if __name__ == "__main__":
    main()
