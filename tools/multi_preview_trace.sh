#!/bin/bash
# usage (GPU box): tools/multi_preview_trace.sh <tag> -- HIP API statistics of tools/multi_preview.py with 10 and with 40 loop iterations: the calls whose
# count does not grow with the iterations are set-up; hipDeviceSynchronize must be among them.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 10 40; do
  d=gpurun_out/hiptrace_${tag}_$n
  rm -rf $d && mkdir -p $d
  rocprofv3 --hip-trace --stats --output-format csv -d $d -- python3 tools/multi_preview.py $n > $d/run.log 2>&1
  echo "== $n iterations"; grep "LOOP END" $d/run.log
  python3 - "$d" <<'PY'
import csv, glob, sys
for fn in glob.glob(sys.argv[1] + "/*/*hip_api_stats.csv"):
    for r in csv.DictReader(open(fn)):
        if any(k in r["Name"] for k in ("Synchronize", "MemcpyPeer", "StreamWaitEvent", "hipMemcpy")):
            print(f'  {r["Name"]:32s} calls {r["Calls"]:>6s}  total {float(r["TotalDurationNs"]) / 1e6:9.2f} ms')
PY
done
