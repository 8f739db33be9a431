#!/bin/bash
# usage: tools/pmc_quick.sh <tag>  -- four short --pmc passes (issue mix, texture path, L1/L2) of the fused render kernel: A/B evidence for kernel edits
tag=$1
export TMPDIR=/tmp
out=gpurun_out/${tag}_pmc_quick.txt
: > $out
pass() {
	name=$1; shift
	d=gpurun_out/pmcq_${tag}_$name
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
	else
		echo "pass $name FAILED"; tail -3 $d/bench.log
	fi
}
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass ta TA_BUSY_avr TA_BUSY_max TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES
cat $out
