"""Probe: encode of step k+1 on one context/stream overlapped with decode of step k on another (two host threads).
Prints ms per step for the serial and the overlapped schedule.  Diagnostic."""
import os, sys, time, threading
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
import kanzi_amd as kz, datagen

B = int(os.environ.get("B", "2048")); bs = 4 << 20; K = int(os.environ.get("K", "4"))
chain, ent = os.environ.get("CHAIN", "BWT+RANK+ZRLT"), os.environ.get("ENT", "ANS0")
dev = torch.device("cuda", 0)
D = 64
host = np.empty((D, bs), dtype=np.uint8)
for i in range(D): host[i] = datagen.block(i, bs)
d_in = torch.from_numpy(host).to(dev).repeat((B + D - 1) // D, 1)[:B].contiguous()
o_stride = kz.max_block_stream_bytes(bs)
d_enc = [torch.zeros((B, o_stride), dtype=torch.uint8, device=dev) for _ in range(2)]
d_dec = torch.zeros((B, bs), dtype=torch.uint8, device=dev)
lengths = np.full(B, bs, dtype=np.int32)
ctxE, ctxD = kz.Context(0), kz.Context(0)

def enc(ctx, slot):
    res = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lengths, d_enc[slot].data_ptr(), o_stride, kz.MEM_DEVICE)
    return np.array([r.bits for r in res], dtype=np.int64)
def dec(ctx, slot, bits):
    res = kz.decode_blocks(ctx, chain, ent, bs, d_enc[slot].data_ptr(), o_stride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
    assert all(r.status == 0 and r.length == bs for r in res)

# warm both contexts (arena allocation) serially
bits = enc(ctxE, 0); dec(ctxD, 0, bits)
torch.cuda.synchronize()
t0 = time.perf_counter()
te_ = td_ = 0.0
for k in range(K):
    a = time.perf_counter(); bits = enc(ctxE, k & 1); b_ = time.perf_counter(); dec(ctxD, k & 1, bits); c_ = time.perf_counter()
    te_ += b_ - a; td_ += c_ - b_
torch.cuda.synchronize()
serial = (time.perf_counter() - t0) / K
print("enc %.1f dec %.1f ms" % (te_ / K * 1e3, td_ / K * 1e3))
print("serial ms/step %.1f  (%.0f MB/s)" % (serial * 1e3, B * bs / serial / 1e6), flush=True)

encoded = [threading.Event() for _ in range(K)]
decoded = [threading.Event() for _ in range(K)]
bits_of = [None] * K
def thread_e():
    for k in range(K):
        if k >= 2: decoded[k - 2].wait()
        bits_of[k] = enc(ctxE, k & 1); encoded[k].set()
def thread_d():
    for k in range(K):
        encoded[k].wait(); dec(ctxD, k & 1, bits_of[k]); decoded[k].set()
torch.cuda.synchronize()
t0 = time.perf_counter()
te, td = threading.Thread(target=thread_e), threading.Thread(target=thread_d)
te.start(); td.start(); te.join(); td.join()
torch.cuda.synchronize()
piped = (time.perf_counter() - t0) / K
print("overlapped ms/step %.1f  (%.0f MB/s) over %d steps" % (piped * 1e3, B * bs / piped / 1e6, K), flush=True)
assert torch.equal(d_in, d_dec)
