#!/bin/bash
# PMC passes over tools_dev/gemm_bench.py <shapes> (gpurun -- 'bash tools_dev/gemm_pmc.sh <tag> <shape>...'): wave-cycle breakdown, LDS, texture path.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_TD_BUSY_sum TCC_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d $out/p$i -o r --output-format csv -- python tools_dev/gemm_bench.py "$@" > $out/p$i.log 2>&1 || echo "pass $i failed: $ctrs"
done
python - "$out" <<'P'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm_\w+<[^>]*>|gemm_\w+)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()): print("    %-36s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
P
