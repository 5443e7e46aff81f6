set -x
mkdir -p gpurun_out/r04a
for K in 7 5 4; do
  KZ_BWT_K=$K KZ_BWT_TRACE=1 timeout 300 python tools/chain_probe.py BWT NONE 714 > gpurun_out/r04a/bwt_K$K.log 2>&1
done
for c in 0 1 2 3 4; do
  KZ_BWT_TRACE=1 timeout 300 python tools/chain_probe.py BWT NONE 357 $c > gpurun_out/r04a/bwt_cls$c.log 2>&1
done
tail -n 14 gpurun_out/r04a/bwt_K*.log
