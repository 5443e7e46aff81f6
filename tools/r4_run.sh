mkdir -p gpurun_out/r04g
timeout 600 python tools/bwt_diag.py quick > gpurun_out/r04g/diag.log 2>&1; tail -2 gpurun_out/r04g/diag.log
echo "== text"; timeout 300 python tools/chain_probe.py BWT NONE 357 0 2>&1 | grep -E "k_tr_|rep 2" | head -8
echo "== sparse"; timeout 300 python tools/chain_probe.py BWT NONE 357 4 2>&1 | grep -E "k_tr_|rep 2" | head -8
echo "== mix"; timeout 300 python tools/chain_probe.py BWT NONE 714 2>&1 | grep -E "k_tr_|rep 2" | head -8
