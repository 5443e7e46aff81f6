mkdir -p gpurun_out/t2
for f in 1 3; do KZ_TEXT_GPU=$f KZ_TEXT_GPU_TRACE=1 timeout 300 python bench.py --chain TEXT+UTF+BWT+RANK+ZRLT --data text --no-shapes --no-chains --no-cpu-baseline --steps 1 --warmup 1 --detail-json gpurun_out/t2/e$f.json > gpurun_out/t2/c$f.log 2>&1; grep textgpu gpurun_out/t2/c$f.log | sort | uniq -c | head -3; python - <<EOF
import json
d=json.load(open("gpurun_out/t2/e$f.json"))
print($f, d["config"]["decode_MBps"], d["config"]["round_trip_ok"], [k for k in d["kernels"] if "text" in k["kernel"]])
EOF
done
timeout 600 python -m pytest tests -m gpu -x -q -k text_inverse_on_the_device 2>&1 | tail -3
