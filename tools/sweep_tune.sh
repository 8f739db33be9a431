#!/bin/bash
# diagnostic: sweep the scheduling knobs (NGP_TUNE=refill_min,skip_steps,go_min,max_stall,k_busy,k_drain,block_jumps,guided) on the benchmark frame
for t in "$@"; do
  echo -n "tune $t: "
  NGP_TUNE=$t timeout -k 10 120 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-training-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s in flight |', d.get('single_frame_mrays'), 'one at a time | kernel', d['roofline']['kernel_ms'], 'ms')"
done
