#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of NeRF inference at 1920x1080 (BASELINE.json metric).

A step is one frame (1 spp) of the fixed 1080p orbit camera over the Lego-shaped scene (configs[1]: HashGrid
L8/F4/T19 + 64-wide MLPs; synthetic weights and occupancy, there is no pretrained snapshot in the reference mount).
With N GPUs the frame's 8x8-pixel camera tiles are dealt round-robin to the ranks (one process per GPU, launched by
torch.distributed.run) and one RCCL gather per frame brings every rank's tile-packed rgba+depth to rank 0, which scatters the
tiles into the image: total work is fixed, so this is strong scaling. Inputs (model, camera) are resident in HBM before the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = "surface-irradiance-estimation-from-neural-radiance-fields_amd"
sys.path.insert(0, ROOT)

WIDTH, HEIGHT = 1920, 1080
FOV_X = 0.6911  # Blender Lego camera_angle_x
AZIMUTHS = [0.0, 45.0, 90.0, 135.0, 180.0, 225.0, 270.0, 315.0]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_TFLOPS = 2500.0  # dense f16 MFMA (MI355X_MICROARCH.md)
BYTES_PER_SAMPLE = 512.0  # 8 levels x 8 corners x 8 B of hash-grid gathers
BYTES_PER_RAY = 80.0  # payload/rgba/depth once + frame-buffer scatter
FLOP_PER_SAMPLE = 20480.0  # both MLPs
# What binds the fused kernel is the CU's texture-address / vector-L1 path, which is paced by LANE-loads, not bytes
# (profiles/r2_*_pmc_l1.json; tools/micro/gather_probe.hip, output profiles/r2_gather_probe.txt: a CU sustains 1.92-1.93 scattered
# lane-loads per clock when every one hits its L1 -- 8 KB table -- whether a lane asks for 4, 8 or 16 bytes). The kernel's gathers are 8-byte lane-loads, 64 per sample = the 512
# algorithmic bytes, so that ceiling, in the same unit as the algorithmic figure, is
GATHER_PEAK_GBS = 1.92 * 8.0 * 256 * 2.4  # lane-loads/clk/CU x 8 B x 256 CUs x 2.4 GHz = 9437 GB/s
GATHER_PEAK_LANE_LOADS = 1.92  # per clock and CU (profiles/r2_gather_probe.txt, 8 KB table: every load an L1 hit)
# ngp_set_schedule: refill_min, skip_steps, go_min, max_stall, k_busy, k_drain, block_jumps, share (defaults of csrc/ngp_host.h)
DEFAULT_SCHEDULE = (64, 4, 32, 1, 1, 4, 1, 1)
EXACT_MARCH = (64, 4, 32, 1, 1, 4, 0, 1)  # block_jumps = 0: empty space is walked voxel by voxel like the reference -- its exact sample sets


def pkg(sub):
    return importlib.import_module(PKG + "." + sub)


def cpu_baseline(scene_dict, scene_mod, gpu_ctx=None, native=None):
    """The oracle (kind "port": the reference has no CPU path, scripts/run.py:25 hard-imports the CUDA module) on a
    bounded sample of the same workload: ONE frame of the same camera and model at 1920x1080, ~30 s on 16 cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc

    out_dir = os.path.join(ROOT, "gpurun_out", "oracle_native")
    os.makedirs(out_dir, exist_ok=True)
    try:
        lib = orc.build(native=True, out_dir=out_dir)
    except Exception:
        lib = None
    o = orc.Oracle(lib)
    sc = dict(scene_dict)
    grid = np.asarray(sc["density_grid"], np.float16).astype(np.float32)
    sc["density_grid_bitfield"], _ = o.density_grid_to_bitfield(grid, sc["max_cascade"])
    m = o.make_model(sc)
    w, h = WIDTH, HEIGHT
    cam = o.make_camera(scene_mod.orbit_camera(AZIMUTHS[1]), w, h, scene_mod.focal_from_fov_x(w, FOV_X))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, int(os.environ.get("NGP_BENCH_CPU_THREADS", "16")))  # a one-GPU box's CPU share is 16 cores
    t0 = time.perf_counter()
    fb, _, st = o.render_nerf(m, cam, o.make_opts(n_threads=cores))
    dt = time.perf_counter() - t0
    o.release(m)
    out = {"value": round(w * h / dt / 1e6, 5), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": f"one frame (azimuth {AZIMUTHS[1]:.0f}) of the same model+camera at {w}x{h}, {st['n_samples']} samples, {dt:.1f} s, OpenMP over rays"}
    if gpu_ctx is not None:
        # the metric's "PSNR vs reference": the HIP frame of that camera against the frame the oracle just rendered (the
        # oracle stands in for the reference, PARITY UNPINNED: oracle/orc_common.h); the oracle is the checker here, no more
        try:
            ref = o.tonemap(o.accumulate(fb.reshape(-1, 4), np.zeros((w * h, 4), np.float32), 0)).reshape(h, w, 4)
            got = gpu_ctx.render(native.make_camera(scene_mod.orbit_camera(AZIMUTHS[1]), w, h, scene_mod.focal_from_fov_x(w, FOV_X)))
            mse = float(np.mean((np.clip(got[..., :3], 0, 1) - np.clip(ref[..., :3], 0, 1)) ** 2))
            out["psnr_vs_oracle_db"] = round(-10.0 * math.log10(max(mse, 1e-12)), 2)
            out["max_abs_diff_vs_oracle"] = round(float(np.abs(got - ref).max()), 6)
            out["psnr_is"] = "PSNR vs oracle (parity unpinned): the reference ships no snapshot, test or vector for this path, the oracle restates it"
            # the same frame with block_jumps = 0 (the march takes the reference's one-voxel steps): the oracle's sample sets
            gpu_ctx.set_schedule(*EXACT_MARCH)
            exact = gpu_ctx.render(native.make_camera(scene_mod.orbit_camera(AZIMUTHS[1]), w, h, scene_mod.focal_from_fov_x(w, FOV_X)))
            gpu_ctx.set_schedule(*DEFAULT_SCHEDULE)
            mse_e = float(np.mean((np.clip(exact[..., :3], 0, 1) - np.clip(ref[..., :3], 0, 1)) ** 2))
            out["exact_march_psnr_vs_oracle_db"] = round(-10.0 * math.log10(max(mse_e, 1e-12)), 2)
            out["exact_march_max_abs_diff_vs_oracle"] = round(float(np.abs(exact - ref).max()), 6)
        except Exception as e:
            out["psnr_vs_oracle_db"] = None
            out["psnr_error"] = str(e)[:160]
    return out


def latest_profile(pattern):
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.basename(files[-1])
    except Exception:
        return None, None


def pmc_evidence():
    """Counter evidence of the fused kernel from the committed rocprofv3 --pmc passes of this same command (counters cannot be
    read from inside the process, hence the files; tools/profile_round.sh and tools/pmc_l1.sh produce them):
    HBM-side bytes per launch (FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950), issue-slot and matrix-pipe utilisation, texture-address busy fraction, L1 / L2 hit rates."""
    out = {}
    hbm, name = latest_profile("r*_pmc_hbm.json")
    if hbm:
        out["traffic"] = int(hbm["traffic_bytes_per_launch_corrected"])
        out["traffic_source"] = "profiles/" + name
    comp, name = latest_profile("r*_pmc_compute.json")
    if comp:
        out["valu_issue_util"] = round(comp["valu_issue_utilisation"], 3)
        out["mfma_pipe_util"] = round(comp["mfma_pipe_utilisation"], 3)
        out["issue_source"] = "profiles/" + name
    l1, name = latest_profile("r*_pmc_l1.json")
    if l1:
        cyc = l1["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        out["ta_busy_avg"] = round(l1["TA_BUSY_avr"] / cyc, 3)
        out["ta_busy_max"] = round(l1["TA_BUSY_max"] / cyc, 3)
        out["l1_hit_rate"] = round(1.0 - l1["TCP_TCC_READ_REQ_sum"] / l1["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)
        out["l2_hit_rate"] = round(l1["TCC_HIT_sum"] / (l1["TCC_HIT_sum"] + l1["TCC_MISS_sum"]), 3)
        # (flat loads in round 1, buffer loads since: the latter count in TA_TOTAL_WAVEFRONTS only)
        lane_loads = max(l1["TA_FLAT_READ_WAVEFRONTS_sum"], l1.get("TA_TOTAL_WAVEFRONTS_sum", 0.0)) * 64.0
        out["lane_loads_per_clk_per_cu"] = round(lane_loads / 256.0 / cyc, 3)
        out["l1_source"] = "profiles/" + name
    return out


def frequency_probe(native, scene_mod):
    """SURVEY 8 a-19, reported beside the headline (never part of `value`): the same 1080p camera over the same occupancy with
    the configs/nerf/frequency.json network (Frequency encodings, MLPs 256 x 7 + 256 x 1; 434176 MACs per sample, 42x base.json's).
    This path is bound by the matrix pipe: its rate is quoted against the dense f16 MFMA peak."""
    try:
        synthetic = pkg("synthetic")
        sc = synthetic.make_scene(aabb_scale=1, seed=1234, cfg=scene_mod.frequency_network_config())
        fctx = native.Context(0)
        fctx.set_model(sc)
        cam = native.make_camera(scene_mod.orbit_camera(AZIMUTHS[1]), WIDTH, HEIGHT, scene_mod.focal_from_fov_x(WIDTH, FOV_X))
        opts = native.make_opts()
        fctx.render_pinned(cam, opts)
        ms = []
        for _ in range(3):
            fctx.render_pinned(cam, opts)
            ms.append(fctx.render_stats()["kernel_device_ms"])
        st = fctx.render_stats()
        fctx.close()
        t = min(ms)
        flops = st["n_samples"] * 2.0 * (421888 + 12288)
        return {"ms_per_frame": round(t, 3), "mrays_s": round(WIDTH * HEIGHT / t / 1e3, 2), "gsamples_s": round(st["n_samples"] / t / 1e6, 3),
                "samples_per_frame": int(st["n_samples"]), "achieved_tflops": round(flops / t / 1e9, 1), "mfma_frac": round(flops / t / 1e9 / MFMA_PEAK_TFLOPS, 3),
                "config": "configs/nerf/frequency.json architecture, synthetic weights, %dx%d, kernel time on the device clock" % (WIDTH, HEIGHT)}
    except Exception as e:  # the headline line must come out whatever happens here
        return {"error": str(e)[:200]}


def training_probe(native, scene_mod, gt_ctx):
    """SURVEY 8 f-2, reported beside the headline (never part of `value`): the training step at the reference's batch of
    2^18 samples on views rendered from the bench scene. Failures are reported, not raised."""
    try:
        res, views, warm, steps, batch = 256, 16, 300, 200, 1 << 18
        focal = scene_mod.focal_from_fov_x(res, FOV_X)
        opts = native.make_opts(background=(0.0, 0.0, 0.0, 0.0))
        mats = [scene_mod.orbit_camera(360.0 * k / views, 15.0 + 40.0 * ((k * 7) % views) / views) for k in range(views)]
        imgs = [gt_ctx.render(native.make_camera(m, res, res, focal), opts) for m in mats]
        path = scene_mod.write_transforms(os.path.join("/tmp", "bench_train_%d.json" % os.getpid()), mats, res, res, FOV_X)
        tctx = native.Context(0)
        tctx.load_training_data(path)
        os.remove(path)
        for i, im in enumerate(imgs):
            tctx.set_training_image(i, im)
        tctx.reset_network(log2_hashmap_size=19, seed=1337)
        tctx.train(warm, batch)
        t0 = time.perf_counter()
        loss = tctx.train(steps, batch)
        dt = time.perf_counter() - t0
        st = tctx.training_state()
        test = scene_mod.orbit_camera(77.0, 33.0)
        got = tctx.render(native.make_camera(test, res, res, focal), opts)
        ref = gt_ctx.render(native.make_camera(test, res, res, focal), opts)
        mse = float(((got[..., :3] - ref[..., :3]) ** 2).mean())
        tctx.close()
        return {"steps_per_s": round(steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "batch_samples": batch, "samples_per_s": round(steps * st["measured_batch_size"] / dt),
                "steps_total": st["training_step"], "loss": loss, "heldout_psnr_db": round(-10.0 * math.log10(max(mse, 1e-12)), 2),
                "config": "fresh base.json network (HashGrid T2^19), %d views %dx%d rendered from the bench scene" % (views, res, res)}
    except Exception as e:  # the headline line must come out whatever happens here
        return {"error": str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-training-probe", action="store_true")
    ap.add_argument("--no-extra-probes", action="store_true", help="skip the exact-march and sharded-share measurements (counter passes: every launch of the run is then a full frame of the default schedule)")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("NGP_BENCH_INFLIGHT", "0")),
                    help="frames in flight (streams / buffer sets); 0 = 2 on one or two GPUs, 6 beyond (a rank's share of a frame is short: tools/shard_probe.py)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the renderer has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # == local_rank on a real run; lets a 1-GPU box rehearse N > 1 (gloo)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("NGP_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # one process per node links the library (build.py renames a finished file into place); the others wait, then only load it
    if local_rank == 0:
        pkg("build").build()
    if world > 1:
        dist.barrier()
    native, synthetic, scene_mod, parallel = pkg("native"), pkg("synthetic"), pkg("scene"), pkg("parallel")
    sc = synthetic.make_scene(aabb_scale=1, seed=1234, log2_hashmap_size=19)  # identical on every rank (seeded)
    ctx = native.Context(dev_index)
    ctx.set_model(sc)  # replica per GPU; a broadcast is not needed because every rank generates the same bytes

    w, h = args.width, args.height
    focal = scene_mod.focal_from_fov_x(w, FOV_X)
    cams = [native.make_camera(scene_mod.orbit_camera(az), w, h, focal) for az in AZIMUTHS]
    # Frames in flight: frame i is rendered (and, for N > 1, gathered) on stream i % k into buffer set i % k, so
    # the drain of one frame's persistent kernel and its all_gather overlap the next frame's render. Every launch,
    # copy and collective of a step is enqueued on that step's stream; fence() joins both.
    n_inflight = args.inflight if args.inflight > 0 else (2 if world <= 2 else 6)
    streams = [torch.cuda.Stream(dev) for _ in range(n_inflight)]
    if world > 1:
        # tile-packed output: the fused kernel writes this rank's tiles in the layout the all_gather moves
        opts = native.make_opts(shard_index=rank, shard_count=world, packed_output=True)
        gatherers = [parallel.PackedFrameGather(w, h, world, dev) for _ in streams]
        outs = [g.buffers() for g in gatherers]
    else:
        opts = native.make_opts()
        outs = [(torch.zeros((h, w, 4), dtype=torch.float32, device=dev), torch.zeros((h, w), dtype=torch.float32, device=dev)) for _ in streams]

    def step(i):
        b = i % len(streams)
        rgba, depth = outs[b]
        with torch.cuda.stream(streams[b]):
            ctx.render_device(cams[i % len(cams)], opts, rgba.data_ptr(), depth.data_ptr(), streams[b].cuda_stream)
            if world > 1:
                return gatherers[b].gather(dst=0, rank=rank)  # rank 0 ends the step holding the full frame
        return rgba, depth

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- outside the timed region (N = 1): one frame at a time, for the kernel's own duration and the latency of a single frame
    # (in the timed region two launches overlap in their drain, so a per-launch HIP-event time there includes the wait for CUs),
    # and the host-buffer entry point ngp_render (what pyngp's Testbed.render returns: render + device-to-host copy)
    timed_hist = ctx.render_history(min(args.steps, 256))
    solo = None
    if world == 1 and args.steps > 0:
        n_solo = min(args.steps, 16)
        rgba, depth = outs[0]
        for i in range(2):
            ctx.render_device(cams[i % len(cams)], opts, rgba.data_ptr(), depth.data_ptr(), streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(n_solo):
            ctx.render_device(cams[i % len(cams)], opts, rgba.data_ptr(), depth.data_ptr(), streams[0].cuda_stream)
            torch.cuda.synchronize(dev)
        solo_dt = (time.perf_counter() - t1) / n_solo
        sh = ctx.render_history(n_solo)
        n_host = min(args.steps, 8)
        ctx.render(cams[0], opts)
        t1 = time.perf_counter()
        for i in range(n_host):
            ctx.render(cams[i % len(cams)], opts)
        pageable_dt = (time.perf_counter() - t1) / n_host
        ctx.render_pinned(cams[0], opts)
        t1 = time.perf_counter()
        for i in range(n_host):
            ctx.render_pinned(cams[i % len(cams)], opts)  # a fresh pooled page-locked array per frame, as pyngp's Testbed.render returns
        host_dt = (time.perf_counter() - t1) / n_host
        solo = {"frame_s": solo_dt, "kernel_ms": float(np.mean([x["kernel_ms"] for x in sh])), "kernel_device_ms": float(np.mean([x["kernel_device_ms"] for x in sh])),
                "n_samples": float(np.mean([x["n_samples"] for x in sh])), "n_rays": float(np.mean([x["n_rays"] for x in sh])), "host_s": host_dt, "pageable_s": pageable_dt, "n": n_solo}

    def in_flight_ms(render_opts, n_frames, k_streams, out_bufs):
        """ms per frame of `n_frames` frames issued over `k_streams` streams (throughput with frames overlapped, image left in HBM)"""
        ss = [torch.cuda.Stream(dev) for _ in range(k_streams)]
        for i in range(2 * k_streams):
            ctx.render_device(cams[i % len(cams)], render_opts, out_bufs[i % k_streams][0].data_ptr(), out_bufs[i % k_streams][1].data_ptr(), ss[i % k_streams].cuda_stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(n_frames):
            ctx.render_device(cams[i % len(cams)], render_opts, out_bufs[i % k_streams][0].data_ptr(), out_bufs[i % k_streams][1].data_ptr(), ss[i % k_streams].cuda_stream)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t1) / n_frames * 1e3

    exact_ms = None
    shares = None
    if world == 1 and args.steps > 0 and not args.no_extra_probes:
        # the benchmarked schedule leaves empty 4^3 / 16^3 blocks in one step; with block_jumps = 0 the march is the reference's, sample for sample
        ctx.set_schedule(*EXACT_MARCH)
        exact_ms = in_flight_ms(opts, min(args.steps, 16), len(streams), outs)
        ctx.set_schedule(*DEFAULT_SCHEDULE)
        # scaling readiness on ONE GPU: rank 0's share of the frame cut N ways (tile-packed output, what a rank of an N-GPU job renders),
        # frames overlapped as bench.py --gpus N overlaps them, and one share at a time; no gather (34 / 136 us on its own link)
        shares = {}
        for n_sh in (2, 4, 8):
            o_sh = native.make_opts(shard_index=0, shard_count=n_sh, packed_output=True)
            k_sh = 2 if n_sh <= 2 else 6
            bufs = [(torch.zeros((h, w, 4), dtype=torch.float32, device=dev), torch.zeros((h, w), dtype=torch.float32, device=dev)) for _ in range(k_sh)]
            shares[str(n_sh)] = {"in_flight_ms": round(in_flight_ms(o_sh, 48, k_sh, bufs), 4), "frames_in_flight": k_sh, "one_at_a_time_ms": round(in_flight_ms(o_sh, 16, 1, bufs[:1]), 4)}

    # outside the timed region: the frame the ranks assembled must be the frame one GPU renders alone
    gather_diff = None
    if world > 1 and rank == 0:
        last = args.steps - 1
        img_g, depth_g = gatherers[last % len(streams)].img.view(h, w, 4), gatherers[last % len(streams)].depth.view(h, w)
        solo_rgba = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        solo_depth = torch.zeros((h, w), dtype=torch.float32, device=dev)
        if args.steps > 0:
            ctx.render_device(cams[last % len(cams)], native.make_opts(), solo_rgba.data_ptr(), solo_depth.data_ptr(), streams[0].cuda_stream)
            torch.cuda.synchronize(dev)
            gather_diff = max(float((img_g - solo_rgba).abs().max().item()), float((depth_g - solo_depth).abs().max().item()))
    hist = timed_hist
    local = np.array([[s["n_rays"], s["n_rays_hit"], s["n_samples"], s["kernel_ms"], s["frame_ms"], s["kernel_device_ms"]] for s in hist], np.float64)
    if world > 1:
        tl = torch.tensor(local, dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(allr, tl)
        per_rank = torch.stack(allr).cpu().numpy()
    else:
        per_rank = local[None]

    if rank == 0:
        n_rays = w * h
        ms = dt / args.steps * 1e3
        value = n_rays / (dt / args.steps) / 1e6
        samples_per_frame = per_rank[:, :, 2].sum(0).mean()
        hits_per_frame = per_rank[:, :, 1].sum(0).mean()
        # dominant kernel: render_nerf_fused on rank 0 (one launch per step). Its duration for the roofline is the HIP-event time
        # of launches issued one at a time (N = 1); with N > 1 the device-clock duration of the timed region's launches stands in.
        k_ms_timed = float(per_rank[0, :, 5].mean())  # first wave in .. last wave out on the chip's 100 MHz clock, timed region
        if solo:
            k_ms, k_rays, k_samples = solo["kernel_ms"], solo["n_rays"], solo["n_samples"]
        else:
            k_ms, k_rays, k_samples = k_ms_timed, float(per_rank[0, :, 0].mean()), float(per_rank[0, :, 2].mean())
        algo_bytes = k_samples * BYTES_PER_SAMPLE + k_rays * BYTES_PER_RAY
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        ev = pmc_evidence()
        mfma_tflops = k_samples * FLOP_PER_SAMPLE / (k_ms * 1e-3) / 1e12
        # which unit binds, by the counters of the committed profile passes of this command (fractions of each unit's ceiling):
        # "ta" = texture-address / vector-L1 path (lane-loads per clock and CU against the measured all-hit ceiling), "valu" = issue
        # slots of the vector ALU, "hbm" = HBM-side bytes (FETCH_SIZE x 2 + WRITE_SIZE) against 8 TB/s, "mfma" = matrix pipe
        util = {}
        if ev.get("lane_loads_per_clk_per_cu") is not None:
            util["ta"] = ev["lane_loads_per_clk_per_cu"] / GATHER_PEAK_LANE_LOADS
        if ev.get("valu_issue_util") is not None:
            util["valu"] = ev["valu_issue_util"]
        if ev.get("mfma_pipe_util") is not None:
            util["mfma"] = ev["mfma_pipe_util"]
        if ev.get("traffic"):
            util["hbm"] = ev["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        bound = max(util, key=util.get) if util else "ta"
        roof = {
            "bound": bound,
            "unit_utilisation_from_counters": {k: round(v, 3) for k, v in util.items()},
            "achieved": round(achieved, 2),
            "peak": round(GATHER_PEAK_GBS, 1),
            "unit": "GB/s",
            "frac": round(achieved / GATHER_PEAK_GBS, 5),
            "traffic": ev.get("traffic"),
            "kernel": "render_nerf_fused",
            "kernel_ms": round(k_ms, 4),
            "kernel_ms_how": ("HIP events around launches issued one at a time (%d frames after the timed region)" % solo["n"]) if solo else "device clock, timed region (launches of consecutive frames overlap)",
            "kernel_ms_timed_region_device_clock": round(k_ms_timed, 4),
            "algorithmic_bytes_per_launch": int(algo_bytes),
            "peak_is": "`achieved` / `peak` / `frac` price the algorithmic gather bytes against the measured L1 gather ceiling for 8-byte lane-loads (tools/micro/gather_probe.hip, profiles/r2_gather_probe.txt): 1.92 lane-loads/clk/CU x 8 B x 256 CUs x 2.4 GHz; `bound` names the unit the counters show closest to ITS ceiling",
            "hbm_algorithmic_frac": round(achieved / HBM_PEAK_GBS, 5),  # SURVEY 8(d)'s figure: algorithmic bytes against the 8 TB/s HBM peak
            "hbm_traffic_frac": round(ev["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ev.get("traffic") else None,
            "mfma_tflops": round(mfma_tflops, 3),
            "mfma_frac": round(mfma_tflops / MFMA_PEAK_TFLOPS, 5),
            "counters": ev,
        }
        out = {
            "metric": "Mrays/s @1080p NeRF inference (Lego snapshot); PSNR vs reference",
            "value": round(value, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": "Lego-shaped synthetic scene (HashGrid L8 F4 T2^19 b=2.0, density MLP 32-64-16, rgb MLP 32-64-64-16, aabb_scale 1), "
                            f"{w}x{h} pinhole camera fov_x 0.6911 rad, 8 orbit azimuths, 1 spp, render_mode Shade, min_transmittance 0.01",
                "rays_per_step": n_rays,
                "samples_per_hit_ray": round(samples_per_frame / max(hits_per_frame, 1.0), 2),
                "hit_fraction": round(hits_per_frame / n_rays, 4),
                "samples_per_step": int(samples_per_frame),
                "tile_sharding": f"8x8 tiles round-robin over {world} rank(s)" + (", gather of tile-packed rgba+depth to rank 0 per frame (RCCL send/recv)" if world > 1 else ""),
                "frames_in_flight": len(streams),
            },
            "roofline": roof,
        }
        if solo:
            out["single_frame_ms"] = round(solo["frame_s"] * 1e3, 4)          # one frame at a time, image left in HBM
            out["single_frame_mrays"] = round(n_rays / solo["frame_s"] / 1e6, 2)
            out["with_host_copy_ms"] = round(solo["host_s"] * 1e3, 4)         # ngp_render into a page-locked, device-mapped host array: the kernels write it over the link themselves (PCIe-inclusive; never `value`)
            out["with_host_copy_mrays"] = round(n_rays / solo["host_s"] / 1e6, 2)
            out["with_host_copy_pageable_mrays"] = round(n_rays / solo["pageable_s"] / 1e6, 2)  # into ordinary (pageable) memory
        if exact_ms is not None:
            out["exact_march_ms"] = round(exact_ms, 4)  # block_jumps = 0: the reference's march, the oracle's sample sets (cpu_baseline.exact_march_max_abs_diff_vs_oracle)
            out["exact_march_mrays"] = round(n_rays / exact_ms / 1e3, 2)
        if shares is not None:
            out["sharded_share_ms"] = shares
            out["sharded_share_speedup_bound"] = {k: round(ms / v["in_flight_ms"], 2) for k, v in shares.items()}  # this run's ms_per_step over a rank's share: compute side only
        if gather_diff is not None:
            out["gathered_frame_max_abs_diff_vs_single_gpu"] = gather_diff  # rank 0's check of the assembled frame, outside the timed region
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, scene_mod, ctx, native)
        if world == 1 and not args.no_training_probe:
            out["training"] = training_probe(native, scene_mod, ctx)
            out["frequency_json"] = frequency_probe(native, scene_mod)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
