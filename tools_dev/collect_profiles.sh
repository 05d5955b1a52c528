#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools_dev/collect_profiles.sh <tag>'): kernel-trace stats of the default bench (two streams and
# single stream) and of the student-only config, then the PMC passes -- HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA
# utilisation / wave-cycle breakdown (SQ counters, their own passes).  Everything lands under gpurun_out/<tag>/; copy the summaries into
# profiles/ afterwards.  rocprofv3 gets the program itself after `--` (no env / bash -c hop).
set -e
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/stats -o r --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic file --no-other-configs > $out/bench_stats.log 2>&1
echo "stats (two streams) done"
rocprofv3 --kernel-trace --stats -d $out/stats_single -o r --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-stream --traffic file --no-other-configs > $out/bench_stats_single.log 2>&1
echo "stats (single stream) done"
rocprofv3 --kernel-trace --stats -d $out/stats_none -o r --output-format csv -- python bench.py --config none --steps 10 --warmup 3 --no-cpu-baseline --traffic file --no-other-configs > $out/bench_stats_none.log 2>&1
echo "stats (student only) done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc/fetch -o r --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-stream --traffic file --no-other-configs > $out/pmc_fetch.log 2>&1
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc/write -o r --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-stream --traffic file --no-other-configs > $out/pmc_write.log 2>&1
echo "pmc write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $out/pmc/mfma -o r --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-stream --traffic file --no-other-configs > $out/pmc_mfma.log 2>&1
echo "pmc mfma done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/pmc/waves -o r --output-format csv -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-side-stream --traffic file --no-other-configs > $out/pmc_waves.log 2>&1
echo "pmc waves done"
python tools_dev/summarize_pmc.py $out/pmc lrkd $out/pmc_traffic.json > $out/pmc_traffic.log
python tools_dev/summarize_sq.py $out/pmc $out/pmc_sq.json > $out/pmc_sq.log
python bench.py --steps 50 --warmup 5 > $out/bench_result.json 2> $out/bench_result.err
python bench.py --config none --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_none_result.json 2> $out/bench_none_result.err
tail -1 $out/bench_result.json | cut -c1-300
