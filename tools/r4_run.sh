mkdir -p gpurun_out/r04k
for c in -1 5; do echo "== BWT 2048 class $c"; KZ_BWT_TRACE=0 timeout 600 python tools/chain_probe.py BWT NONE 2048 $c 2>&1 | grep -E "rep [12]" ; done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
