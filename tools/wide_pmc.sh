#!/bin/bash
# usage: tools/wide_pmc.sh <tag> -- kernel time and a few --pmc passes of the frequency.json network kernel (tools/wide_net_probe.py)
tag=$1
export TMPDIR=/tmp
out=gpurun_out/${tag}_wide_pmc.txt
: > $out
d=gpurun_out/widepmc_${tag}_trace
rm -rf $d && mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/wide_net_probe.py > $d/run.log 2>&1 && find $d -name "*kernel_stats.csv" | head -1 | xargs head -4 >> $out
pass() {
	name=$1; shift
	d=gpurun_out/widepmc_${tag}_$name
	rm -rf $d && mkdir -p $d
	if rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -- python3 tools/wide_net_probe.py > $d/run.log 2>&1; then
		python3 - "$d" "$name" >> $out <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for fn in glob.glob(f'{d}/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        if 'network_inference_wide' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"{name} {k} mean={sum(v)/len(v):.6g} n={len(v)}")
PY
	else
		echo "pass $name FAILED"; tail -3 $d/run.log
	fi
}
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cat $out
