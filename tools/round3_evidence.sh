#!/bin/bash
# usage (on the GPU box): tools/round3_evidence.sh <tag> <part 1|2>  -- everything profiles/<tag>_* is copied from
tag=$1; part=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$part" = "1" ]; then
  timeout -k 10 500 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_pytest.txt 2>&1; tail -n 2 gpurun_out/${tag}_pytest.txt
  timeout -k 10 600 bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1; tail -n 2 gpurun_out/${tag}_profile_round.log
  timeout -k 10 300 bash tools/pmc_l1.sh $tag > gpurun_out/${tag}_pmc_l1.log 2>&1; tail -n 1 gpurun_out/${tag}_pmc_l1.log
else
  timeout -k 10 300 bash tools/profile_extra.sh $tag > gpurun_out/${tag}_profile_extra.log 2>&1
  timeout -k 10 300 python tools/run_configs.py > gpurun_out/${tag}_configs.jsonl 2> gpurun_out/${tag}_configs.err
  timeout -k 10 200 tools/pmc_config.sh $tag 5 > gpurun_out/${tag}_pmc5.log 2>&1; tail -n 1 gpurun_out/${tag}_pmc5.log
  timeout -k 10 200 tools/pmc_config.sh $tag 4 > gpurun_out/${tag}_pmc4.log 2>&1; tail -n 1 gpurun_out/${tag}_pmc4.log
  (NGP_BENCH_INFLIGHT=2 NGP_SHARD_ALL_RANKS=0 timeout -k 10 120 python tools/shard_probe.py; NGP_BENCH_INFLIGHT=6 timeout -k 10 200 python tools/shard_probe.py; NGP_BENCH_INFLIGHT=1 NGP_SHARD_ALL_RANKS=0 timeout -k 10 120 python tools/shard_probe.py) > gpurun_out/${tag}_shard_probe.txt 2>&1
  (K=6 timeout -k 10 120 python tools/small_frame_rate.py; K=2 timeout -k 10 120 python tools/small_frame_rate.py; timeout -k 10 120 python tools/latency_probe.py 4) > gpurun_out/${tag}_small_frame.txt 2>&1
  timeout -k 10 100 python tools/wave_trace.py 256 256 1 1 > gpurun_out/${tag}_trace_256_sections.txt 2>&1
  timeout -k 10 100 python tools/wave_trace.py 256 256 1 2 > gpurun_out/${tag}_trace_256_network.txt 2>&1
  timeout -k 10 100 python tools/wave_trace.py 1920 1080 8 1 > gpurun_out/${tag}_trace_share8_sections.txt 2>&1
  timeout -k 10 100 python tools/wave_trace.py 1920 1080 1 1 > gpurun_out/${tag}_trace_1080p_sections.txt 2>&1
  tail -n 4 gpurun_out/${tag}_shard_probe.txt gpurun_out/${tag}_small_frame.txt
fi
