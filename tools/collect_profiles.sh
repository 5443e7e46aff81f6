#!/bin/bash
# Copy the judged artefacts of tools/profile_round.sh (merged back under gpurun_out/<tag>/) into profiles/<tag>_*.
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p profiles
for f in $OUT/pmc_traffic*.json $OUT/kernel_stats*.csv; do [ -f "$f" ] && cp $f profiles/${TAG}_$(basename $f); done
[ -f $OUT/bench_kernels.json ] && cp $OUT/bench_kernels.json profiles/${TAG}_bench_kernels.json
[ -f $OUT/bench_default.json ] && tail -n 1 $OUT/bench_default.json > profiles/${TAG}_bench_default.json
ls -la profiles | grep ${TAG}_
