#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box) -- the evidence bench.py's roofline block cites:
#   1. default bench run                         -> gpurun_out/<tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of it    -> gpurun_out/<tag>_kernel_stats.csv
#   3. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section) -> gpurun_out/<tag>_pmc_hbm.json
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done"
rm -rf gpurun_out/prof_$tag && mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --no-training-probe --no-extra-probes > gpurun_out/prof_$tag/bench.log 2>&1
cp gpurun_out/prof_$tag/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
echo "kernel trace done"
bash tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE > gpurun_out/${tag}_pmc.txt
bash tools/pmc_pass.sh ${tag}_write WRITE_SIZE >> gpurun_out/${tag}_pmc.txt
bash tools/pmc_pass.sh ${tag}_compute SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE >> gpurun_out/${tag}_pmc.txt
cat gpurun_out/${tag}_pmc.txt
python3 - "$tag" <<'PY'
import json, re, sys
tag = sys.argv[1]
vals = {}
for line in open(f"gpurun_out/{tag}_pmc.txt"):
    m = re.match(r"\S+ (\w+) mean=([0-9.e+]+) n=(\d+)", line)
    if m:
        vals[m.group(1)] = (float(m.group(2)), int(m.group(3)))
fetch_kb, nf = vals["FETCH_SIZE"]
write_kb, nw = vals["WRITE_SIZE"]
out = {
    "FETCH_SIZE_KB_per_launch": fetch_kb, "FETCH_SIZE_launches": nf,
    "WRITE_SIZE_KB_per_launch": write_kb, "WRITE_SIZE_launches": nw,
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --steps 3 --warmup 1), kernel render_nerf_fused_unit_plain, 1080p. "
            "Counters are in KB; gfx950 reports half of the fetched bytes (MI355X_MICROARCH.md), so read bytes = 2 x FETCH_SIZE. Infinity-Cache hits are included.",
    "traffic_bytes_per_launch_corrected": int((2.0 * fetch_kb + write_kb) * 1024.0),
}
json.dump(out, open(f"gpurun_out/{tag}_pmc_hbm.json", "w"), indent=1)
print(out)
# compute-side utilisation of the same launch: 256 CUs x 4 SIMDs; a wave64 VALU instruction occupies its SIMD for 4 cycles,
# v_mfma_f32_16x16x32_f16 for 16 (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA); GRBM_GUI_ACTIVE is summed over the 8 XCDs
if "SQ_INSTS_VALU" in vals:
    cyc = vals["GRBM_GUI_ACTIVE"][0] / 8.0
    simds = 1024.0
    comp = {k: vals[k][0] for k in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")}
    comp["kernel_cycles_per_xcd"] = cyc
    comp["valu_issue_utilisation"] = comp["SQ_INSTS_VALU"] * 4.0 / simds / cyc
    comp["mfma_pipe_utilisation"] = comp["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / cyc
    comp["note"] = "per launch of render_nerf_fused_unit_plain at 1080p (bench.py --steps 3 --warmup 1, kernels serialised by the counter collection)"
    json.dump(comp, open(f"gpurun_out/{tag}_pmc_compute.json", "w"), indent=1)
    print(comp)
PY
