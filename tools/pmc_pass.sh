#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counter list>"  -- one rocprofv3 --pmc pass over a short bench run (kernel trace only)
set -e
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$tag
rocprofv3 --pmc $@ --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-training-probe --no-extra-probes > gpurun_out/pmc_$tag/bench.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/pmc_{tag}/*/*counter_collection.csv')
agg = collections.defaultdict(list)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'render_nerf_fused' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"{tag} {k} mean={sum(v)/len(v):.6g} n={len(v)}")
PY
