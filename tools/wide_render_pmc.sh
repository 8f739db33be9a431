#!/bin/bash
# usage: tools/wide_render_pmc.sh <tag> -- instruction mix of the frequency.json RENDER kernel (tools/wide_probe.py), one --pmc pass
tag=$1
export TMPDIR=/tmp
d=gpurun_out/widerpmc_${tag}
rm -rf $d && mkdir -p $d
N=2 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -- python3 tools/wide_probe.py > $d/run.log 2>&1
python3 - "$d" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(list)
for fn in glob.glob(f'{d}/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        if 'render_nerf_wide' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"{k} mean={sum(v)/len(v):.6g} n={len(v)}")
PY
