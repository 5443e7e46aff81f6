mkdir -p gpurun_out/r5b
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py --detail-json gpurun_out/r5b/bench_default.json > gpurun_out/r5b/bench_default.log 2>&1; tail -1 gpurun_out/r5b/bench_default.log | cut -c1-300
