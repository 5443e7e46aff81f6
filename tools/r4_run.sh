for c in 3 0; do for v in "" 1; do echo "== class $c OCC1=$v"; env ${v:+KZ_TRQ_OCC1=1} timeout 300 python tools/chain_probe.py BWT NONE 357 $c 2>&1 | grep -E "k_tr_sort|rep 2" | head -3; done; done
