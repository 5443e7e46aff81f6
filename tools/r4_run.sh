timeout 900 python tools/bwt_diag.py quick 2>&1 | tail -1
for c in 0 4; do echo "== class $c"; timeout 300 python tools/chain_probe.py BWT NONE 357 $c 2>&1 | grep -E "rep 2|k_tr_count" | head -3; done
echo "== mix 682"; timeout 300 python tools/chain_probe.py BWT NONE 682 2>&1 | grep -E "rep 2" 
