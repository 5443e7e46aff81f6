mkdir -p gpurun_out/r04camp
SEEDS=401,402,403 CASES=700 timeout 1500 python tools/fuzz_campaign.py > gpurun_out/r04camp/fuzz_plain.log 2>&1; tail -3 gpurun_out/r04camp/fuzz_plain.log
KZ_STREAM_CHUNK=8 KZ_HOST_CHUNK=8 KZ_HOST_CHUNK_DEC=8 SEEDS=404,405 CASES=500 timeout 1500 python tools/fuzz_campaign.py > gpurun_out/r04camp/fuzz_pipe.log 2>&1; tail -3 gpurun_out/r04camp/fuzz_pipe.log
SEEDS=406,407 CASES=400 timeout 600 python tools/entropy_campaign.py > gpurun_out/r04camp/entropy.log 2>&1; tail -2 gpurun_out/r04camp/entropy.log
SEEDS=408 CASES=400 timeout 600 python tools/transform_campaign.py > gpurun_out/r04camp/transform.log 2>&1; tail -2 gpurun_out/r04camp/transform.log
