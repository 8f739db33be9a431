"""How far does the accumulation width of the MLPs move an image? The reference's tcnn FullyFusedMLP sums in fp16 accumulator
fragments (nerf_network.h:120,130); the MI355X build sums in fp32 MFMA accumulators. Neither can be compared with the reference
directly (PARITY UNPINNED), so both are measured against the float64 network on the same model and camera:

  ideal     oracle, mlp_accumulate = "ideal"     (float64 interpolation, encodings and MLPs; no intermediate rounding)
  exact     oracle, mlp_accumulate = "exact"     (fp16 activations, every dot product exact: the oracle's default)
  fp16_k16  oracle, mlp_accumulate = "fp16_k16"  (running sum rounded to fp16 after every 16-wide K block: tcnn's fragments)
  hip       libngp_hip.so on the GPU (fp16 activations, fp32 MFMA accumulation; block_jumps = 0: the oracle's sample sets) -- only with a GPU

usage: python tools/mlp_bracket.py [W H]   -> one JSON line per scene (PSNR in dB on the sRGB-free linear frame, clipped to [0, 1],
as scripts/run.py:252-258 computes it). PSNR(hip, fp16_k16) is the bound on "within 0.1 dB of the CUDA reference" that can be stated."""
import importlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
native, synthetic, scene = (importlib.import_module(PKG + "." + m) for m in ("native", "synthetic", "scene"))
import oracle as orc


def psnr(a, b):
    mse = float(((np.clip(a[..., :3], 0, 1) - np.clip(b[..., :3], 0, 1)) ** 2).mean())
    return 99.0 if mse == 0 else -10.0 * np.log10(mse)


def bracket(o, ctx, sc, mat, w, h):
    """PSNR matrix of one view; ctx = None leaves the GPU out"""
    focal = scene.focal_from_fov_x(w, 0.6911)
    frames = {}
    for mode in ("ideal", "exact", "fp16_k16"):
        s2 = dict(sc)
        s2["mlp_accumulate"] = mode
        m = o.make_model(s2)
        fb, _, st = o.render_nerf(m, o.make_camera(mat, w, h, focal))
        frames[mode] = o.tonemap(o.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
        o.release(m)
    if ctx is not None:
        ctx.set_model(sc)
        # block_jumps = 0: the march takes the reference's one-voxel steps, so the HIP frame has the oracle's sample sets and what is
        # left is arithmetic (with jumps, ~7e-6 of the rays gain or lose a boundary sample, which is all a 75 dB comparison sees)
        ctx.set_schedule(64, 4, 32, 1, 1, 4, 0)
        frames["hip"] = ctx.render(native.make_camera(mat, w, h, focal), native.make_opts())
        ctx.set_schedule(64, 4, 32, 1, 1, 4, 1)
    out = {"samples": int(st["n_samples"]), "hit": int(st["n_rays_hit"])}
    for a, b in (("exact", "ideal"), ("fp16_k16", "ideal"), ("hip", "ideal"), ("fp16_k16", "exact"), ("hip", "exact"), ("hip", "fp16_k16")):
        if a in frames and b in frames:
            out[f"psnr_{a}_vs_{b}"] = round(psnr(frames[a], frames[b]), 2)
            out[f"max_abs_{a}_vs_{b}"] = round(float(np.abs(frames[a] - frames[b]).max()), 5)
            d = np.abs(frames[a][..., :3] - frames[b][..., :3])
            d = d[frames[b][..., 3] > 0.01]  # pixels the object covers
            out[f"q50_{a}_vs_{b}"] = float(np.quantile(d, 0.5))  # (robust against the rays whose sample set differs)
            out[f"q99_{a}_vs_{b}"] = float(np.quantile(d, 0.99))
    return out


def with_bitfield(o, sc):
    bf, mean = o.density_grid_to_bitfield(np.asarray(sc["density_grid"], np.float16).astype(np.float32), sc["max_cascade"])
    sc["density_grid_bitfield"] = bf
    return sc


if __name__ == "__main__":
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 144)
    o = orc.Oracle()
    ctx = None
    try:
        import torch

        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
            ctx = native.Context(0)
    except Exception:
        ctx = None
    for name, sc, view in (("bench model (Lego-shaped, T = 2^19)", synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19), (45.0, 30.0, 4.03)),
                           ("garden-shaped (aabb 16, upstream per_level_scale)", synthetic.make_scene(aabb_scale=16, seed=5, log2_hashmap_size=19, pls_rule="upstream"), (40.0, 25.0, 4.03))):
        r = bracket(o, ctx, with_bitfield(o, sc), scene.orbit_camera(*view), w, h)
        r["scene"] = name
        r["resolution"] = [w, h]
        print(json.dumps(r), flush=True)
