#!/bin/bash
# usage: tools/pmc_l1.sh <tag>   (on the GPU box) -- texture-address / vector-L1 / L2 counters of the fused render kernel,
# a few counters per rocprofv3 --pmc pass (asking for many TA counters at once is rejected with error 38, "exceeds the
# capabilities of the hardware"); every pass is its own short bench run, the program directly after `--`.
# Result: gpurun_out/<tag>_pmc_l1.json (copied into profiles/ by hand).
tag=$1
export TMPDIR=/tmp
out=gpurun_out/${tag}_pmc_l1.txt
: > $out
pass() {
	name=$1; shift
	d=gpurun_out/pmc_${tag}_$name
	rm -rf $d && mkdir -p $d
	if rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-training-probe --no-extra-probes --inflight 1 > $d/bench.log 2>&1; then
		python3 - "$d" "$name" >> $out <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for fn in glob.glob(f'{d}/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        if 'render_nerf_fused' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"{name} {k} mean={sum(v)/len(v):.6g} n={len(v)}")
PY
		echo "pass $name ok"
	else
		echo "pass $name FAILED (see $d/bench.log)"
		tail -3 $d/bench.log
	fi
}
pass ta1 TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
pass ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass ta3 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass tcp3 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
cat $out
python3 - "$tag" <<'PY'
import json, re, sys
tag = sys.argv[1]
vals = {}
for line in open(f"gpurun_out/{tag}_pmc_l1.txt"):
    m = re.match(r"(\S+) (\w+) mean=([0-9.e+-]+) n=(\d+)", line)
    if m:
        vals[m.group(2)] = float(m.group(3))
vals["note"] = ("per launch of render_nerf_fused_unit at 1080p (bench.py --steps 3 --warmup 1 --inflight 1; launches serialised by the counter collection); "
                "one rocprofv3 --pmc pass per line group of tools/pmc_l1.sh; *_sum = summed over the chip's instances")
json.dump(vals, open(f"gpurun_out/{tag}_pmc_l1.json", "w"), indent=1)
print(json.dumps(vals))
PY
