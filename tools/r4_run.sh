timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "transform_forward and (RANK or MTFT or SRT)" 2>&1 | tail -2
timeout 300 python tools/chain_probe.py BWT+RANK+ZRLT ANS0 2048 2>&1 | grep -E "rep 2|k_sbrt_replay" 
