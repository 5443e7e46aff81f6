python - <<'PY'
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import torch
import kanzi_amd as kz
c = kz.Context(0)
print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l and 'r-xp' in l])
try:
    t = torch.zeros(4).cuda(); print("torch cuda ok after Context", t.device)
except Exception as e:
    print("FAIL", e)
PY
python - <<'PY'
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, '.')
import torch
torch.zeros(1).cuda()
import kanzi_amd as kz
c = kz.Context(0)
print("ok torch first")
PY
env | grep -i -E "hip|rocr|visible|hsa" 
git stash -q; python -m pytest tests/test_gpu_parity.py -x -q -k "test_level_exact_streams_match_oracle and ANS0" 2>&1 | tail -3; git stash pop -q
