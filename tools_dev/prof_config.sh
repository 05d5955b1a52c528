#!/bin/bash
# gpurun -- 'bash tools_dev/prof_config.sh <config> <tag> [extra bench flags]': kernel-trace stats of one bench config -> gpurun_out/<tag>/
set -e
cfg=${1:-none}; tag=${2:-prof_$cfg}; shift 2 || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o r --output-format csv -- python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-side-stream --traffic file --no-other-configs "$@" > $out/bench_stats.log 2>&1
python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/bench_result.json 2> $out/bench_result.err
tail -1 $out/bench_result.json | cut -c1-300
