timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "level_exact or host_stage_pipeline or level5 or prestaged" 2>&1 | tail -3
for ch in 256 0; do echo "== B=51 KZ_HOST_CHUNK=$ch"; env $( [ $ch != 0 ] && echo KZ_HOST_CHUNK=$ch ) timeout 300 python tools/level5_probe.py 51 2>&1 | grep -E "TEXT.*rep 1"; done
