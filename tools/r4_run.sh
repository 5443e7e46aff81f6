mkdir -p gpurun_out/r04j
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "level_exact or host_stage_pipeline or prestaged or pipelined_stream or level5 or text_block_size or async_batches" 2>&1 | tail -6
KZ_TRACE_PIPE=1 timeout 600 python tools/level5_probe.py 2048 > gpurun_out/r04j/level5.log 2>&1; grep -E "rep|decode done|decode chunk . : host" gpurun_out/r04j/level5.log | tail -14
