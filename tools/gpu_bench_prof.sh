#!/bin/bash
# bench.py once plain, then under rocprofv3 --kernel-trace --stats twice: the headline + ICP alone (--no-extras: the
# 1M x 1M launches are the only ones of their kernels in that table) and with every extra; summaries under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 python3 bench.py "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
rm -rf gpurun_out/prof_bench gpurun_out/prof_bench_headline
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench_headline -- python3 bench.py --no-cpu --no-extras "$@" > gpurun_out/prof_bench_headline.log 2>&1 || { tail -20 gpurun_out/prof_bench_headline.log; exit 1; }
python3 tools/stats_summary.py gpurun_out/prof_bench_headline
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --no-cpu "$@" > gpurun_out/prof_bench.log 2>&1 || { tail -20 gpurun_out/prof_bench.log; exit 1; }
python3 tools/stats_summary.py gpurun_out/prof_bench
