timeout 900 python tools/bwt_diag.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "trie_rounds or bwt_forward_group or full_size or large_blocks or 45_blocks" 2>&1 | tail -2
echo "== mix 682"; timeout 300 python tools/chain_probe.py BWT NONE 682 2>&1 | grep -E "rep 2" 
echo "== full 2048"; timeout 300 python tools/chain_probe.py BWT+RANK+ZRLT ANS0 2048 2>&1 | grep -E "rep 2" 
