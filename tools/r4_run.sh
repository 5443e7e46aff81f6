cp scratch/libkanzi_qonly.so kanzi_amd/libkanzi_hip.so
KZ_PROBE_ENC_ONLY=1 timeout 300 python tools/chain_probe.py BWT+RANK+ZRLT ANS0 2048 2>&1 | grep -E "rep 2|k_sbrt_replay" 
