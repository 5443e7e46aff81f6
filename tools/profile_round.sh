#!/bin/bash
# Round measurement recipe (run on the GPU box through gpurun): writes everything under gpurun_out/<tag>/.
#   tools/profile_round.sh r02
# 1. bench.py defaults (one JSON line)           -> bench_default.json
# 2. rocprofv3 --kernel-trace --stats of bench   -> kernel_stats.csv
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with traces)
#    aggregated by tools/pmc_traffic.py          -> pmc_traffic.json
# Copy the files you want judged to profiles/<tag>_*.
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
BLOCKS=${BLOCKS:-2048}
timeout 900 python bench.py --traffic-json $OUT/pmc_none.json > $OUT/bench_nopmc.json 2> $OUT/bench.err
cd /tmp
# (bench.py runs one extra instrumented step behind the timed ones: every kernel of the step shows up --steps + 1 times)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-shapes > $ROOT/$OUT/stats.log 2>&1
timeout 1500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-shapes > $ROOT/$OUT/pmc_fetch.log 2>&1
timeout 1500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_write -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-shapes > $ROOT/$OUT/pmc_write.log 2>&1
cd $ROOT
F=$(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1)
W=$(ls $OUT/pmc_write/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W $BLOCKS > $OUT/pmc_traffic.json
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
# final bench line with the measured traffic attached
timeout 900 python bench.py --traffic-json $OUT/pmc_traffic.json > $OUT/bench_default.json 2>> $OUT/bench.err
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats      # raw per-dispatch CSVs are large
cat $OUT/bench_default.json
head -12 $OUT/kernel_stats.csv
