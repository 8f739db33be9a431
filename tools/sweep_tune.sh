#!/bin/bash
# diagnostic: sweep the scheduling knobs (NGP_TUNE=refill_min,skip_steps,go_min,max_stall) on the benchmark frame
for t in "$@"; do
  echo -n "tune $t: "
  NGP_TUNE=$t timeout -k 10 120 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s', d['roofline']['kernel_ms'], 'ms')"
done
