"""Do two stages share the GPU?  Chain X on context 1 and chain Y on context 2 (own streams, own arenas), alone and side by side.
   python tools/overlap_probe.py BWT RANK [blocks]      (entropy NONE; device resident)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kanzi_amd as kz
import datagen
X, Y = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 714
bs = 4 << 20
D = min(64, B)
host = np.stack([datagen.block(i, bs) for i in range(D)])
d_in = torch.from_numpy(host).cuda().repeat((B + D - 1) // D, 1)[:B].contiguous()
os_ = kz.max_block_stream_bytes(bs)
lens = np.full(B, bs, dtype=np.int32)
c1, c2 = kz.Context(0), kz.Context(0)
o1 = torch.zeros((B, os_), dtype=torch.uint8, device="cuda")
o2 = torch.zeros((B, os_), dtype=torch.uint8, device="cuda")
def run(which):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    jobs = []
    if which & 1: jobs.append(kz.submit_encode_blocks(c1, X, "NONE", d_in.data_ptr(), bs, lens, o1.data_ptr(), os_, kz.MEM_DEVICE))
    if which & 2: jobs.append(kz.submit_encode_blocks(c2, Y, "NONE", d_in.data_ptr(), bs, lens, o2.data_ptr(), os_, kz.MEM_DEVICE))
    for j in jobs: j.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for rep in range(3):
    a, b, ab = run(1), run(2), run(3)
    print("%s alone %.0f ms, %s alone %.0f ms, side by side %.0f ms (sum %.0f, max %.0f)" % (X, a, Y, b, ab, a + b, max(a, b)), flush=True)
