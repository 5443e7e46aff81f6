#!/bin/bash
# the forward suffix sort's kernels on a time axis, round by round: tools/bwt_timeline.sh <tag> [blocks] [class]
set -u
TAG=$1; B=${2:-342}; CLS=${3:-0}
OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp
KZ_PROBE_ENC_ONLY=0 KZ_BWT_TRACE=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python $ROOT/tools/chain_probe.py BWT NONE $B $CLS > $OUT/bwt_tl_$CLS.log 2>&1
F=$(ls $OUT/tl/*/*kernel_trace.csv | head -1)
python - "$F" <<'PY' > $OUT/bwt_timeline_$CLS.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_bwt_init")]
st = idx[-1]
t0 = int(rows[st]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[st:]:
    name = r["Kernel_Name"].split("(")[0][:28]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    gap = (int(r["Start_Timestamp"]) - prev_end) / 1e6
    prev_end = max(prev_end, int(r["End_Timestamp"]))
    busy += e - s
    print("%8.2f .. %8.2f ms  %7.2f  gap %6.2f  %s  grid %s" % (s, e, e - s, gap, name, r.get("Grid_Size", "")))
    if name.startswith("k_bwt_emit"): break
print("busy %.1f ms" % busy)
PY
rm -rf $OUT/tl
tail -60 $OUT/bwt_timeline_$CLS.txt
