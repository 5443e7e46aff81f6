"""Probe: one batch coded as S sub-batches on S contexts driven by S host threads (kernels of different sub-batches
interleave on the GPU) versus the whole batch on one context.  Diagnostic."""
import os, sys, time, threading
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
import kanzi_amd as kz, datagen

B = int(os.environ.get("B", "2048")); bs = 4 << 20; K = int(os.environ.get("K", "3")); S = int(os.environ.get("S", "2"))
chain, ent = os.environ.get("CHAIN", "BWT+RANK+ZRLT"), os.environ.get("ENT", "ANS0")
dev = torch.device("cuda", 0)
D = 64
host = np.empty((D, bs), dtype=np.uint8)
for i in range(D): host[i] = datagen.block(i, bs)
d_in = torch.from_numpy(host).to(dev).repeat((B + D - 1) // D, 1)[:B].contiguous()
o_stride = kz.max_block_stream_bytes(bs)
d_enc = torch.zeros((B, o_stride), dtype=torch.uint8, device=dev)
d_dec = torch.zeros((B, bs), dtype=torch.uint8, device=dev)
ctxs = [kz.Context(0) for _ in range(S)]
# sub-batch s = blocks s, s+S, s+2S ... would need strided views; use contiguous ranges but interleave classes inside each
bounds = [(B * s // S, B * (s + 1) // S) for s in range(S)]
bits_all = np.zeros(B, dtype=np.int64)

def enc_range(ctx, lo, hi):
    n = hi - lo
    res = kz.encode_blocks(ctx, chain, ent, d_in[lo:hi].data_ptr(), bs, np.full(n, bs, dtype=np.int32), d_enc[lo:hi].data_ptr(), o_stride, kz.MEM_DEVICE)
    bits_all[lo:hi] = [r.bits for r in res]
def dec_range(ctx, lo, hi):
    res = kz.decode_blocks(ctx, chain, ent, bs, d_enc[lo:hi].data_ptr(), o_stride, bits_all[lo:hi].copy(), d_dec[lo:hi].data_ptr(), bs, kz.MEM_DEVICE)
    assert all(r.status == 0 and r.length == bs for r in res)

def run(fn, split):
    if not split:
        fn(ctxs[0], 0, B); return
    th = [threading.Thread(target=fn, args=(ctxs[s],) + bounds[s]) for s in range(S)]
    for t in th: t.start()
    for t in th: t.join()

for split in (False, True):
    run(enc_range, split); run(dec_range, split)          # warm (arena sizes differ)
    torch.cuda.synchronize()
    te = td = 0.0
    for k in range(K):
        a = time.perf_counter(); run(enc_range, split); torch.cuda.synchronize(); b_ = time.perf_counter(); run(dec_range, split); torch.cuda.synchronize(); c_ = time.perf_counter()
        te += b_ - a; td += c_ - b_
    print("%s: enc %.1f dec %.1f total %.1f ms/step (%.0f MB/s)" % ("split x%d" % S if split else "whole", te / K * 1e3, td / K * 1e3, (te + td) / K * 1e3, B * bs * K / (te + td) / 1e6), flush=True)
assert torch.equal(d_in, d_dec)
