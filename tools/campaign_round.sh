#!/bin/bash
# The differential campaigns on the round's final code (fresh seeds), one line per campaign -> gpurun_out/<tag>_campaign.txt
#   tools/campaign_round.sh r06 <seed base>
TAG=${1:-r06}; S=${2:-9600}
OUT=gpurun_out/${TAG}_campaign.txt
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "# differential campaigns on the library of commit ${KZ_GIT_SHA:-?}, fresh seeds, one MI355X box; HIP path vs oracle/libkzo.so"
echo "== fuzz_campaign SEEDS=$((S+1)),$((S+2)) CASES=1500: $(SEEDS=$((S+1)),$((S+2)) CASES=1500 python tools/fuzz_campaign.py 2>&1 | tail -1)"
echo "== fuzz_campaign through the pipelines (KZ_STREAM_CHUNK=8 KZ_HOST_CHUNK=8 KZ_HOST_CHUNK_DEC=8) SEEDS=$((S+3)) CASES=1500: $(KZ_STREAM_CHUNK=8 KZ_HOST_CHUNK=8 KZ_HOST_CHUNK_DEC=8 SEEDS=$((S+3)) CASES=1500 python tools/fuzz_campaign.py 2>&1 | tail -1)"
echo "== transform_campaign SEEDS=$((S+4)) CASES=1200: $(SEEDS=$((S+4)) CASES=1200 python tools/transform_campaign.py 2>&1 | tail -1)"
echo "== tightcap_fuzz 90 s seed $((S+5)): $(python tools/tightcap_fuzz.py 90 $((S+5)) 2>&1 | tail -1)"
echo "== bwt_fuzz 60 s seed $((S+6)): $(python tools/bwt_fuzz.py 60 $((S+6)) 2>&1 | tail -1)"
echo "== entropy_count_fuzz 40 s seed $((S+7)): $(python tools/entropy_count_fuzz.py 40 $((S+7)) 2>&1 | tail -1)"
echo "== text_fwd_gpu_fuzz 120 s seed $((S+8)) big4: $(python tools/text_fwd_gpu_fuzz.py 120 $((S+8)) big4 2>&1 | tail -2 | tr '\n' ' ')"
echo "== text_gpu_fuzz 120 s seed $((S+9)) big4: $(python tools/text_gpu_fuzz.py 120 $((S+9)) big4 2>&1 | tail -2 | tr '\n' ' ')"
echo "== utf_fwd_gpu_fuzz 120 s seed $((S+10)): $(python tools/utf_fwd_gpu_fuzz.py 120 $((S+10)) 2>/dev/null | tail -1)"
echo "== KZ_SBRT_FORM=0 (the 32-bit RANK forms) fuzz_campaign SEEDS=$((S+11)) CASES=600: $(KZ_SBRT_FORM=0 SEEDS=$((S+11)) CASES=600 python tools/fuzz_campaign.py 2>&1 | tail -1)"
} > $OUT 2>&1
cat $OUT
