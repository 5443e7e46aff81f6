timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "transform_forward or block_streams or corrupted_input or full_size or fuzz_streams_match" 2>&1 | tail -3
timeout 300 python tools/chain_probe.py BWT+RANK+ZRLT ANS0 2048 2>&1 | grep -E "rep 2|k_zrlt" 
