#!/bin/bash
# usage: tools/tune_ab.sh "t0,t1,..." ...   -- bench N=1 and the 8-way shard probe for each NGP_TUNE setting
for t in "$@"; do
  export NGP_TUNE=$t
  v=$(timeout -k 10 120 python bench.py --steps 16 --warmup 4 --no-cpu-baseline | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['roofline']['kernel_ms'])")
  s=$(timeout -k 10 200 python tools/shard_probe.py 2>&1 | grep "N=[248]:" | sed 's/.*rank \([0-9.]*\) ms.*/\1/' | tr '\n' ' ')
  echo "tune $t : N=1 $v | slowest rank ms at N=2,4,8: $s"
done
