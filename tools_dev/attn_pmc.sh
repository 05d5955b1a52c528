#!/bin/bash
# PMC passes over tools_dev/attn_bench.py (gpurun -- 'bash tools_dev/attn_pmc.sh <tag>'): wave-cycle breakdown, LDS and VALU/MFMA activity.
set -e
tag=${1:-attn}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/p1 -o r --output-format csv -- python tools_dev/attn_bench.py > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $out/p2 -o r --output-format csv -- python tools_dev/attn_bench.py > $out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU -d $out/p3 -o r --output-format csv -- python tools_dev/attn_bench.py > $out/p3.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $out/p4 -o r --output-format csv -- python tools_dev/attn_bench.py > $out/p4.log 2>&1 || true
python - "$out" <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
import re
agg2 = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, g), d in agg.items():
    pass
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"attn_\w+<\d+>", r["Kernel_Name"])
        if m: agg2[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg2.items()):
    print(k)
    for c, v in sorted(d.items()): print("    %-28s %14.0f" % (c, sum(v) / len(v)))
P
