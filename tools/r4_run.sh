timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "entropy or block_streams or corrupted or knz_stream or fuzz or batched_decode or full_size" 2>&1 | tail -4
timeout 300 python tools/chain_probe.py BWT+RANK+ZRLT ANS0 2048 2>&1 | grep -E "rep 2|k_ans" 
