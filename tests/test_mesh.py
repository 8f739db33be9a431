"""Geometry mode: BVH4 build invariants (CPU), and GPU parity of the mesh trace / shadow / BRDF pass and of the
depth-composited hybrid frame against the oracle."""
import os

import numpy as np
import pytest

from conftest import pkg, psnr

REF_BUNNY = "/root/reference/data/geometry/objs/bunny.obj"


def _meshes():
    mi = pkg("meshio")
    return [(mi.icosphere(3), (0.0, 0.0, 0.0)), (mi.torus(40, 20), (0.9, 0.1, 0.4)), (mi.icosphere(1), (-0.8, 0.5, -0.3))]


def _check_bvh(nodes, tris, n_leaf=8):
    """Every triangle is in exactly one leaf, leaves hold <= 8 triangles, node boxes bound their subtree."""
    seen = np.zeros(tris.shape[0], np.int32)

    def tri_bounds(lo, hi):
        pts = np.concatenate([tris["a"][lo:hi], tris["b"][lo:hi], tris["c"][lo:hi]])
        return pts.min(0), pts.max(0)

    def visit(i):
        nd = nodes[i]
        if nd["left_idx"] < 0:
            lo, hi = -nd["left_idx"] - 1, -nd["right_idx"] - 1
            assert 0 <= hi - lo <= n_leaf
            seen[lo:hi] += 1
            if hi > lo:
                bmin, bmax = tri_bounds(lo, hi)
                assert np.all(nd["bmin"] <= bmin) and np.all(nd["bmax"] >= bmax)
                return bmin, bmax
            return None
        assert nd["right_idx"] - nd["left_idx"] >= 4
        boxes = [visit(c) for c in range(nd["left_idx"], nd["left_idx"] + 4)]
        boxes = [b for b in boxes if b is not None]
        bmin = np.min([b[0] for b in boxes], 0)
        bmax = np.max([b[1] for b in boxes], 0)
        assert np.all(nd["bmin"] <= bmin) and np.all(nd["bmax"] >= bmax)
        return bmin, bmax

    visit(0)
    assert np.all(seen == 1)


def test_bvh_build_invariants_host(native):
    ctx = native.Context(-1)
    for tris, c in _meshes():
        ctx.add_mesh(tris, c)
    assert ctx.n_meshes() == 3
    for m in range(3):
        nodes, tris = ctx.mesh_bvh(m)
        info = ctx.mesh_info(m)
        assert info["n_tris"] == _meshes()[m][0].shape[0] and nodes.dtype.itemsize == 32 and tris.dtype.itemsize == 36
        _check_bvh(nodes, tris)
        # load_mesh normalisation: unit cube around `center` (inflated by 0.5 % of the diagonal, so slightly inside)
        lo, hi = info["aabb"]
        c = np.asarray(_meshes()[m][1], np.float32)
        assert np.all(lo >= c - 1e-6) and np.all(hi <= c + 1 + 1e-6) and (hi - lo).max() > 0.95
    lo, hi = ctx.mesh_info(-1)["aabb"]
    all_lo = np.min([ctx.mesh_info(m)["aabb"][0] for m in range(3)], 0)
    assert np.allclose(lo, all_lo - 4.0) and ctx.mesh_info(-1)["n_tris"] == sum(t.shape[0] for t, _ in _meshes())
    ctx.clear_meshes()
    assert ctx.n_meshes() == 0
    with pytest.raises(RuntimeError, match="no triangles"):
        ctx.add_mesh(np.zeros((0, 3, 3), np.float32))
    ctx.close()


def test_obj_stl_scene_loading_host(tmp_path, native):
    mi = pkg("meshio")
    tris = mi.icosphere(1)
    mi.save_obj(str(tmp_path / "ico.obj"), tris)
    # binary STL of the same mesh
    with open(tmp_path / "ico.stl", "wb") as f:
        f.write(b"\0" * 80 + np.uint32(tris.shape[0]).tobytes())
        for t in tris:
            f.write(np.zeros(3, np.float32).tobytes() + t.astype(np.float32).tobytes() + b"\0\0")
    # quad face -> two triangles, negative indices, vt/vn suffixes
    (tmp_path / "quad.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nf 1/1 2/1 3/1 4/1\nf -4//1 -3//1 -2//1\n")
    (tmp_path / "scene.json").write_text('{"geometry": [{"center": [0,0,0], "path": "ico.obj", "type": "Mesh"},'
                                         ' {"center": [1.5,0,0], "path": "ico.stl", "type": "Mesh"}, {"center": [0,2,0], "path": "quad.obj", "type": "Mesh"}]}')
    ctx = native.Context(-1)
    ctx.load_scene(str(tmp_path / "scene.json"))
    assert ctx.n_meshes() == 3
    assert ctx.mesh_info(0)["n_tris"] == tris.shape[0] == ctx.mesh_info(1)["n_tris"] and ctx.mesh_info(2)["n_tris"] == 3
    n0, t0 = ctx.mesh_bvh(0)
    n1, t1 = ctx.mesh_bvh(1)
    assert np.allclose(np.sort(t0["a"], 0), np.sort(t1["a"] - np.float32([1.5, 0, 0]), 0), atol=1e-5)
    with pytest.raises(RuntimeError, match="obj or binary .stl"):
        ctx.load_mesh_file(str(tmp_path / "scene.json"))
    (tmp_path / "bad.json").write_text('{"geometry": [{"center": [0,0,0], "path": "ico.obj", "type": "Volume"}]}')
    with pytest.raises(RuntimeError, match="'Mesh' or 'Nerf'"):
        ctx.load_scene(str(tmp_path / "bad.json"))
    ctx.close()


@pytest.mark.skipif(not os.path.exists(REF_BUNNY), reason="reference mount not present")
def test_reference_bunny_loads_host(native):
    ctx = native.Context(-1)
    ctx.load_mesh_file(REF_BUNNY)
    assert ctx.mesh_info(0)["n_tris"] == 4968  # bunny 4968 triangles
    nodes, tris = ctx.mesh_bvh(0)
    _check_bvh(nodes, tris)
    ctx.close()


def test_oracle_bvh_traversal_matches_brute_force(oracle):
    """The oracle's BVH4 traversal returns the brute-force nearest hit (pins the oracle's mesh path)."""
    mi = pkg("meshio")
    h = oracle.mesh_scene([(mi.icosphere(2), (0.0, 0.0, 0.0))])
    rng = np.random.default_rng(4)
    o = rng.uniform(-1, 2, (2000, 3)).astype(np.float32)
    d = rng.normal(size=(2000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p, n = oracle.trace_mesh(h, o, d)
    t = np.linalg.norm(p - o, axis=1)
    hit = ~np.all(n == d, axis=1)
    # analytic: the normalised icosphere is inscribed in a sphere of radius ~0.5/1.01 around (0.5,0.5,0.5)
    oc = o - 0.5
    b = (oc * d).sum(1)
    disc = b * b - ((oc * oc).sum(1) - 0.4951 ** 2)
    analytic_hit = (disc > 0) & ((-b + np.sqrt(np.maximum(disc, 0))) > 0)
    assert (hit == analytic_hit).mean() > 0.97  # faceted sphere vs smooth sphere differ only at the silhouette
    assert hit.sum() > 100 and np.all(t[hit] < 100.0)
    assert np.allclose(np.linalg.norm(n[hit], axis=1), 1.0, atol=1e-5)
    oracle.mesh_scene_destroy(h)


# ----------------------------------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
def test_mesh_trace_parity(gpu_ctx, oracle):
    gpu_ctx.clear_meshes()
    for tris, c in _meshes():
        gpu_ctx.add_mesh(tris, c)
    h = oracle.mesh_scene(_meshes())
    lo, hi = oracle.mesh_scene_aabb(h)
    glo, ghi = gpu_ctx.mesh_info(-1)["aabb"]
    assert np.array_equal(lo, glo) and np.array_equal(hi, ghi)
    rng = np.random.default_rng(12)
    n = 50000
    o = rng.uniform(-2, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gp, gd = gpu_ctx.trace_mesh_rays(o, d)
    rp, rd = oracle.trace_mesh(h, o, d)
    hit = ~np.all(rd == d, axis=1)
    assert hit.sum() > 1000
    # different (but both valid) BVH partitions only matter on exact ties between triangles
    same = np.all(gp == rp, axis=1) & np.all(gd == rd, axis=1)
    assert same.mean() > 0.999
    assert np.abs(gp - rp)[np.isfinite(rp).all(1)].max() < 1e-4
    oracle.mesh_scene_destroy(h)
    gpu_ctx.clear_meshes()


@pytest.mark.gpu
def test_mesh_render_parity(gpu_ctx, oracle, native, scene_mod):
    gpu_ctx.clear_meshes()
    for tris, c in _meshes():
        gpu_ctx.add_mesh(tris, c)
    h = oracle.mesh_scene(_meshes())
    kw = dict(sun_dir=(0.3, 0.8, 0.5), roughness=0.35, metallic=0.2, sheen=0.3, clearcoat=0.5, clearcoat_gloss=0.7, subsurface=0.2,
              basecolor=(0.9, 0.5, 0.3), ambientcolor=(0.1, 0.1, 0.15))
    gpu_ctx.set_geometry_opts(**kw)
    w, hgt = 200, 112
    mat = scene_mod.orbit_camera(35.0, 20.0, 7.0)
    focal = scene_mod.focal_from_fov_x(w, 0.9)
    img, depth = gpu_ctx.render(native.make_camera(mat, w, hgt, focal), native.make_opts(testbed_mode=native.MODE_GEOMETRY, background=(0, 0, 0, 0)), want_depth=True)
    fb, db = oracle.render_mesh(h, oracle.make_camera(mat, w, hgt, focal), oracle.make_mesh_opts(**kw))
    lit = fb[..., :3].sum(-1) > 0
    assert lit.mean() > 0.05 and (fb[..., 3] == 1).mean() > 0.5
    assert np.array_equal(img[..., 3], fb[..., 3])
    assert (np.abs(img - fb).max(-1) < 1e-4).mean() > 0.999  # silhouette / edge ties only
    assert psnr(img[..., :3], fb[..., :3]) > 55.0
    assert (np.abs(depth - db) < 1e-4).mean() > 0.999
    oracle.mesh_scene_destroy(h)
    gpu_ctx.clear_meshes()
    gpu_ctx.set_geometry_opts()


@pytest.mark.gpu
def test_hybrid_frame_parity(gpu_ctx, oracle, native, scene_mod, scene_unit):
    """Geometry mode: mesh pass, then the NeRF pass clipped to the scene AABB and depth-tested against the mesh
    (shade_kernel_nerf_geometry)."""
    gpu_ctx.set_model(scene_unit)
    gpu_ctx.clear_meshes()
    mi = pkg("meshio")
    meshes = [(mi.icosphere(3), (0.55, -0.1, 0.0)), (mi.torus(32, 16), (-0.3, 0.25, 0.4))]
    for tris, c in meshes:
        gpu_ctx.add_mesh(tris, c)
    h = oracle.mesh_scene(meshes)
    w, hgt = 160, 90
    mat = scene_mod.orbit_camera(60.0, 25.0, 5.5)
    focal = scene_mod.focal_from_fov_x(w, 0.8)
    img, depth = gpu_ctx.render(native.make_camera(mat, w, hgt, focal), native.make_opts(testbed_mode=native.MODE_GEOMETRY), want_depth=True)
    st = gpu_ctx.render_stats()
    ocam = oracle.make_camera(mat, w, hgt, focal)
    fb, db = oracle.render_mesh(h, ocam)
    sc = dict(scene_unit)
    lo, hi = oracle.mesh_scene_aabb(h)
    sc["render_aabb"] = (tuple(lo.tolist()), tuple(hi.tolist()))
    m = oracle.make_model(sc)
    fb2, db2, ost = oracle.render_nerf(m, ocam, oracle.make_opts(depth_test=True), frame_buffer=fb, depth_buffer=db)
    ref = oracle.tonemap(oracle.accumulate(fb2.reshape(-1, 4), np.zeros((w * hgt, 4), np.float32), 0)).reshape(hgt, w, 4)
    assert ost["n_rays_hit"] > 500 and abs(int(st["n_rays_hit"]) - int(ost["n_rays_hit"])) <= 3
    # both contributions are present and they interact (some NeRF rays are hidden by the mesh)
    nerf_only, _, _ = oracle.render_nerf(m, ocam)
    assert (np.abs(fb2 - fb).sum(-1) > 0).mean() > 0.05 and (np.abs(fb2[..., :3] - nerf_only[..., :3]).sum(-1) > 1e-3).mean() > 0.02
    assert psnr(img[..., :3], ref[..., :3]) > 48.0
    assert (np.abs(img - ref).max(-1) < 1e-2).mean() > 0.998
    oracle.release(m)
    oracle.mesh_scene_destroy(h)
    gpu_ctx.clear_meshes()
