#!/bin/bash
# usage: tools/profile_extra.sh <tag>  (on the GPU box) -- (1) kernel trace of bench.py with ONE frame in flight: the launch durations
# bench.py's roofline block quotes (with two frames in flight, the default, concurrent launches stretch each other);
# (2) kernel trace of the frequency.json renderer (tools/wide_probe.py)
tag=$1
export TMPDIR=/tmp
d=gpurun_out/prof_${tag}_inflight1
rm -rf $d && mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --inflight 1 --no-cpu-baseline --no-training-probe --no-extra-probes > $d/bench.log 2>&1 && cp $d/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats_inflight1.csv
d=gpurun_out/prof_${tag}_wide
rm -rf $d && mkdir -p $d
N=8 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/wide_probe.py > $d/run.log 2>&1 && cp $d/*/*kernel_stats.csv gpurun_out/${tag}_wide_kernel_stats.csv
tail -3 $d/run.log
head -3 gpurun_out/${tag}_kernel_stats_inflight1.csv gpurun_out/${tag}_wide_kernel_stats.csv
