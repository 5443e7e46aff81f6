"""Would a cost-class schedule pay for BWT+SRT+ZRLT & FPAQ?  The blocks whose FPAQ stage is long (iid-like: classes 1, 3 of the
   generator) on context 1, the others on context 2: one after the other, started together, and context 2 started DELAY seconds late.
   python tools/fpaq_sched_probe.py [blocks] [delay]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kanzi_amd as kz
import datagen
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
delay = float(sys.argv[2]) if len(sys.argv) > 2 else 0.45
bs = 4 << 20
D = 64
host = np.stack([datagen.block(i, bs) for i in range(D)])
cls = np.arange(B) % D % 5
d_all = torch.from_numpy(host).cuda().repeat((B + D - 1) // D, 1)[:B]
idxE = torch.from_numpy(np.nonzero((cls == 1) | (cls == 3))[0]).cuda()
idxC = torch.from_numpy(np.nonzero((cls != 1) & (cls != 3))[0]).cuda()
dE, dC = d_all[idxE].contiguous(), d_all[idxC].contiguous()
os_ = kz.max_block_stream_bytes(bs)
oE = torch.zeros((len(idxE), os_), dtype=torch.uint8, device="cuda"); oC = torch.zeros((len(idxC), os_), dtype=torch.uint8, device="cuda")
lE = np.full(len(idxE), bs, dtype=np.int32); lC = np.full(len(idxC), bs, dtype=np.int32)
c1, c2 = kz.Context(0), kz.Context(0)
for c in (c1, c2): c.set_block_size(bs)
CH, EN = "BWT+SRT+ZRLT", "FPAQ"
def enc(ctx, d, l, o): return kz.submit_encode_blocks(ctx, CH, EN, d.data_ptr(), bs, l, o.data_ptr(), os_, kz.MEM_DEVICE)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc(c1, dE, lE, oE).wait(); t1 = time.perf_counter(); enc(c2, dC, lC, oC).wait(); t2 = time.perf_counter()
    a = enc(c1, dE, lE, oE); b = enc(c2, dC, lC, oC); a.wait(); b.wait(); t3 = time.perf_counter()
    a = enc(c1, dE, lE, oE); time.sleep(delay); b = enc(c2, dC, lC, oC); b.wait(); t4 = time.perf_counter(); a.wait(); t5 = time.perf_counter()
    print("expensive %d blocks alone %.2f s, cheap %d alone %.2f s (sum %.2f); started together %.2f s; cheap %.2f s late: cheap done at %.2f, all done at %.2f s" % (
        len(idxE), t1 - t0, len(idxC), t2 - t1, t2 - t0, t3 - t2, delay, t4 - t3, t5 - t3), flush=True)
