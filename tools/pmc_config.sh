#!/bin/bash
# usage: tools/pmc_config.sh <tag> <config 2|4|5|1>   (on the GPU box) -- counter passes of the fused render kernel on one BASELINE configuration:
# issue mix, texture path, L1 / L2 hit rates, HBM-side bytes (FETCH_SIZE and WRITE_SIZE in passes of their own, MI355X_MICROARCH.md).
# Result: gpurun_out/<tag>_pmc_config<cfg>.json. Every pass is its own short run, the program directly after `--`.
tag=$1; cfg=$2
export TMPDIR=/tmp
out=gpurun_out/${tag}_pmc_config${cfg}.txt
: > $out
pass() {
	name=$1; shift
	d=gpurun_out/pmcc_${tag}_${cfg}_$name
	rm -rf $d && mkdir -p $d
	if rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python3 tools/render_config.py $cfg 4 > $d/run.log 2>&1; then
		python3 - "$d" "$name" >> $out <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for fn in glob.glob(f'{d}/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        if 'render_nerf_fused' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            agg['kernel'] = [r['Kernel_Name'].split('(')[0]]
for k, v in agg.items():
    if k == 'kernel': print(f"{name} kernel {v[0]}")
    else: print(f"{name} {k} mean={sum(v)/len(v):.6g} n={len(v)}")
PY
	else
		echo "pass $name FAILED"; tail -3 $d/run.log
	fi
}
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass ta1 TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
pass ta2 TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cat $out
python3 - "$tag" "$cfg" <<'PY'
import json, re, sys
tag, cfg = sys.argv[1], sys.argv[2]
v = {}
for line in open(f"gpurun_out/{tag}_pmc_config{cfg}.txt"):
    m = re.match(r"(\S+) (\w+) mean=([0-9.e+-]+) n=(\d+)", line)
    if m: v[m.group(2)] = float(m.group(3))
    m = re.match(r"\S+ kernel (\S+)", line)
    if m: v["kernel"] = m.group(1)
if "GRBM_GUI_ACTIVE" in v:
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    d = {"config": cfg, "kernel_cycles_per_xcd": cyc}
    if "SQ_INSTS_VALU" in v:
        d["valu_issue_utilisation"] = v["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc
        d["mfma_pipe_utilisation"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc
    if "TA_BUSY_avr" in v: d["ta_busy_avg"], d["ta_busy_max"] = v["TA_BUSY_avr"] / cyc, v["TA_BUSY_max"] / cyc
    if "TA_TOTAL_WAVEFRONTS_sum" in v: d["lane_loads_per_clk_per_cu"] = v["TA_TOTAL_WAVEFRONTS_sum"] * 64.0 / 256.0 / cyc
    if "TCP_TCC_READ_REQ_sum" in v: d["l1_hit_rate"] = 1.0 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if "TCC_HIT_sum" in v: d["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        d["traffic_bytes_per_launch_corrected"] = int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0)  # gfx950: FETCH_SIZE counts half of the fetched bytes (MI355X_MICROARCH.md)
        d["hbm_side_GBps_at_2p4GHz"] = d["traffic_bytes_per_launch_corrected"] / (cyc / 2.4e9) / 1e9
    v["derived"] = d
v["note"] = "per launch of the fused render kernel on tools/render_config.py <cfg> (one frame at a time; launches serialised by the counter collection); one rocprofv3 --pmc pass per line group of tools/pmc_config.sh; *_sum = summed over the chip's instances"
json.dump(v, open(f"gpurun_out/{tag}_pmc_config{cfg}.json", "w"), indent=1)
print(json.dumps(v.get("derived")))
PY
