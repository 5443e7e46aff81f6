#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per access of known patterns (tools/ubench_gather.hip): tools/pmc_calibrate.sh <tag>
# -> gpurun_out/<tag>/pmc_calibration.json  (copy to profiles/<tag>_pmc_calibration.json; tools/pmc_traffic.py reads the factors)
set -u
TAG=${1:-r05}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
BIN=$ROOT/tools/ubench_gather
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN $ROOT/tools/ubench_gather.hip
cd /tmp; export TMPDIR=/tmp
: > $OUT/ubench_gather.txt
for k in stream16 gather4 chase4 chase4geo scatter4 stream16w; do
  $BIN $k 4096 256 65536 >> $OUT/ubench_gather.txt 2>&1
done
$BIN chase4 4096 1024 16384 >> $OUT/ubench_gather.txt 2>&1
$BIN chase4geo 4096 1024 16384 >> $OUT/ubench_gather.txt 2>&1
$BIN chase4 16 1024 16384 >> $OUT/ubench_gather.txt 2>&1        # one block's link array (16 MiB): the Infinity Cache / L2 ceiling
$BIN chase4 192 1024 16384 >> $OUT/ubench_gather.txt 2>&1       # twelve blocks' link arrays: inside the 256 MiB Infinity Cache
for k in stream16 gather4 chase4 scatter4 stream16w; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/cal_${k}_$c
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/cal_${k}_$c -- $BIN $k 4096 256 65536 > $OUT/cal_${k}_$c.log 2>&1
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, re, sys
out = sys.argv[1]
res = {"note": "KiB per counter unit (guide); per-access figures from the SECOND launch of each kernel (tools/ubench_gather.hip runs a warm-up first)", "patterns": {}}
acc = {}
for line in open(out + "/ubench_gather.txt"):
    m = re.match(r"(\w+) array 4096 MiB steps 256 waves 65536: ([\d.]+) ms, (\d+) accesses, ([\d.]+) G accesses/s, ([\d.]+) useful GB/s", line)
    if m: acc[m.group(1)] = {"ms": float(m.group(2)), "accesses": int(m.group(3)), "G_per_s": float(m.group(4)), "useful_GBs": float(m.group(5))}
for k, a in acc.items():
    row = dict(a)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob("%s/cal_%s_%s/*/*counter_collection.csv" % (out, k, c))
        if not fs: continue
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == c and r["Kernel_Name"].startswith("k_" + k.replace("geo", "")) ]
        if vals:
            row[c + "_KiB"] = vals[-1]
            row[c + "_bytes_per_access"] = vals[-1] * 1024.0 / a["accesses"]
    res["patterns"][k] = row
json.dump(res, open(out + "/pmc_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
cat $OUT/ubench_gather.txt
rm -rf $OUT/cal_*_FETCH_SIZE $OUT/cal_*_WRITE_SIZE
