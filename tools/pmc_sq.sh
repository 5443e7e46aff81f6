#!/bin/bash
# SQ counters of one chain's kernels (issue-bound or waiting?): tools/pmc_sq.sh <tag> <chain> <entropy> <blocks> <class> [kernel-regex]
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, summed over waves)
set -u
TAG=$1; CH=$2; ENT=$3; B=$4; CLS=$5; RE=${6:-.}
OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq_$CH -- python $ROOT/tools/chain_probe.py $CH $ENT $B $CLS > $OUT/sq_$CH.log 2>&1
F=$(ls $OUT/sq_$CH/*/*counter_collection.csv | head -1)
python - "$F" "$RE" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rx = re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if not rx.search(k): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    print("%-28s launches %3d  wave quad-cycles %.3e  wait_any %.0f%%  wait_inst %.0f%%  active %.0f%%  VALU %.3e SALU %.3e LDS %.3e VMEM_RD %.3e" % (
        k[:28], n[k], wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0)))
PY
rm -rf $OUT/sq_$CH
