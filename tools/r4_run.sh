timeout 900 python tools/bwt_diag.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "trie_rounds or bwt_forward_group or large_blocks" 2>&1 | tail -2
for c in 0 3; do echo "== class $c"; timeout 300 python tools/chain_probe.py BWT NONE 357 $c 2>&1 | grep -E "rep 2|k_live" | head -3; done
echo "== mix 682"; timeout 300 python tools/chain_probe.py BWT NONE 682 2>&1 | grep -E "rep 2" 
