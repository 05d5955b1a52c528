#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools_dev/collect_profiles.sh <tag>'): kernel-trace stats of the default bench, then the two
# PMC passes for HBM traffic.  Everything lands under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards.
set -e
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o r --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/stats_single -o r --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-stream > $out/bench_stats_single.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc/fetch -o r --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc/write -o r --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/pmc_write.log 2>&1
python tools_dev/summarize_pmc.py $out/pmc lrkd $out/pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $out/bench_result.json 2> $out/bench_result.err
tail -1 $out/bench_result.json | cut -c1-400
