#!/bin/bash
# sweep schedule knobs on small frames / shares in flight: tools/tune_small.sh "64,4,32,1,1,8,1,0" "32,4,32,1,1,8,1,0" ...
for t in "$@"; do
  NGP_TUNE=$t K=${K:-6} python tools/small_frame_rate.py 2>&1 | grep "in flight"
done
