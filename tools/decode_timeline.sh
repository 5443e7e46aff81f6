#!/bin/bash
# the decoder's kernels on a time axis (start / end in ms from the first kernel of the last decode call): tools/decode_timeline.sh <tag> [blocks]
set -u
TAG=$1; B=${2:-2048}
OUT=$(pwd)/gpurun_out/$TAG; mkdir -p $OUT; ROOT=$(pwd)
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -- python $ROOT/tools/chain_probe.py BWT+RANK+ZRLT ANS0 $B > $OUT/tl.log 2>&1
F=$(ls $OUT/tl/*/*kernel_trace.csv | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode call starts at the last k_ans_dec_index (first kernel of the decoder)
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_ans_dec_index")]
st = idx[-1] if idx else 0
# (the decoder issues k_ans_dec_index per class: take the first of the last group of them)
while st > 0 and int(rows[st]["Start_Timestamp"]) - int(rows[st - 1]["End_Timestamp"]) < 50_000_000 and not rows[st - 1]["Kernel_Name"].startswith("k_frame"): st -= 1
t0 = int(rows[st]["Start_Timestamp"])
for r in rows[st:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e - s >= 3.0: print("%8.1f .. %8.1f ms  %6.1f  %s" % (s, e, e - s, r["Kernel_Name"].split("(")[0][:40]))
print("last kernel ends at %.1f ms" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e6))
PY
rm -rf $OUT/tl
