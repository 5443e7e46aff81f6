#!/bin/bash
# Round measurement recipe (run on the GPU box through gpurun): writes everything under gpurun_out/<tag>/.
#   tools/profile_round.sh r03
# 1. bench.py defaults (one JSON line)                                   -> bench_nopmc.json
# 2. rocprofv3 --kernel-trace --stats of the headline chain and of every config.chains row
#                                                                        -> kernel_stats.csv, kernel_stats_<key>.csv
# 3. two separate --pmc passes per chain (FETCH_SIZE, WRITE_SIZE; never combined with traces), aggregated by
#    tools/pmc_traffic.py                                                -> pmc_traffic.json, pmc_traffic_<key>.json
# 4. bench.py defaults with the measured traffic attached                -> bench_default.json
# Copy the files you want judged to profiles/<tag>_*.
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
BLOCKS=${BLOCKS:-2048}
PMC_CHAINS=${PMC_CHAINS:-"head lz_ans0 bwt_srt_zrlt_fpaq level5_exact"}
# KZ_GIT_SHA=<commit> in the environment is stamped into the traffic files (there is no .git on the GPU box)
timeout 1200 python bench.py --profiles-tag none > $OUT/bench_nopmc.json 2> $OUT/bench.err
cd /tmp
# (bench.py runs one extra instrumented step behind the timed ones: every kernel of the step shows up --steps + 1 times)
run_chain() {   # key chain entropy data
  local key=$1 chain=$2 ent=$3 data=$4
  local common="--chain $chain --entropy $ent --data $data --steps 1 --no-cpu-baseline --no-shapes --no-chains --blocks $BLOCKS"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats_$key -- python $ROOT/bench.py $common --warmup 1 > $ROOT/$OUT/stats_$key.log 2>&1
  cp $(ls $ROOT/$OUT/stats_$key/*/*kernel_stats.csv | head -1) $ROOT/$OUT/kernel_stats_$key.csv
  rm -rf $ROOT/$OUT/stats_$key
  case " $PMC_CHAINS " in *" $key "*) ;; *) return ;; esac
  timeout 1500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch_$key -- python $ROOT/bench.py $common --warmup 0 > $ROOT/$OUT/pmc_fetch_$key.log 2>&1
  timeout 1500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_write_$key -- python $ROOT/bench.py $common --warmup 0 > $ROOT/$OUT/pmc_write_$key.log 2>&1
  local F=$(ls $ROOT/$OUT/pmc_fetch_$key/*/*counter_collection.csv | head -1)
  local W=$(ls $ROOT/$OUT/pmc_write_$key/*/*counter_collection.csv | head -1)
  python $ROOT/tools/pmc_traffic.py $F $W $BLOCKS $chain $ent > $ROOT/$OUT/pmc_traffic_$key.json
  rm -rf $ROOT/$OUT/pmc_fetch_$key $ROOT/$OUT/pmc_write_$key      # raw per-dispatch CSVs are large
}
run_chain head BWT+RANK+ZRLT ANS0 mix
run_chain lz_ans0 LZ ANS0 mix
run_chain bwt_srt_zrlt_fpaq BWT+SRT+ZRLT FPAQ mix
run_chain level5_exact TEXT+UTF+BWT+RANK+ZRLT ANS0 text
cd $ROOT
mv $OUT/kernel_stats_head.csv $OUT/kernel_stats.csv
mv $OUT/pmc_traffic_head.json $OUT/pmc_traffic.json
# final bench line with the measured traffic attached: bench.py reads profiles/<tag>_pmc_traffic*.json
mkdir -p profiles
for f in $OUT/pmc_traffic*.json; do cp $f profiles/${TAG}_$(basename $f); done
timeout 1200 python bench.py --profiles-tag $TAG --detail-json $OUT/bench_kernels.json > $OUT/bench_default.json 2>> $OUT/bench.err
cp $OUT/bench_kernels.json profiles/${TAG}_bench_kernels.json
cp $OUT/bench_default.json profiles/${TAG}_bench_default.json
for f in $OUT/kernel_stats*.csv; do cp $f profiles/${TAG}_$(basename $f); done
cat $OUT/bench_default.json | cut -c1-600
head -8 $OUT/kernel_stats.csv | cut -c1-160
