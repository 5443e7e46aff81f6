"""Throughput of a chain on a chosen class of 4 MiB blocks, with the per-kernel breakdown.  Diagnostic.
   CHAIN=PACK ENT=NONE DATA=text|dna|mm|mix B=1024 python tools/stage_probe.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
import kanzi_amd as kz, refinputs, datagen

B = int(os.environ.get("B", "1024")); bs = 4 << 20
chain, ent, kind = os.environ.get("CHAIN", "PACK"), os.environ.get("ENT", "NONE"), os.environ.get("DATA", "text")
rng = np.random.default_rng(1)
def gen(k):
    if kind == "text": return datagen.block(5 * k, bs)                   # class 0: Markov text
    if kind == "dna": return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, bs)]
    if kind == "uni": return datagen.block(k, bs, 3)
    if kind == "mm": return np.frombuffer(refinputs.multimedia_like(k % 5, bs, seed=k), dtype=np.uint8)
    return datagen.block(k, bs)
dev = torch.device("cuda", 0)
host = np.stack([gen(k) for k in range(10)])
d_in = torch.from_numpy(host).to(dev).repeat((B + 9) // 10, 1)[:B].contiguous()
o_stride = kz.max_block_stream_bytes(bs)
d_enc = torch.zeros((B, o_stride), dtype=torch.uint8, device=dev)
d_dec = torch.zeros((B, bs), dtype=torch.uint8, device=dev)
lengths = np.full(B, bs, dtype=np.int32)
ctx = kz.Context(0)
for it in range(2):
    ctx.set_kernel_timing(it == 1); ctx.reset_kernel_timing()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lengths, d_enc.data_ptr(), o_stride, kz.MEM_DEVICE)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    bits = np.array([r.bits for r in res], dtype=np.int64)
    res2 = kz.decode_blocks(ctx, chain, ent, bs, d_enc.data_ptr(), o_stride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
    torch.cuda.synchronize(); t2 = time.perf_counter()
assert all(r.status == 0 and r.length == bs for r in res2) and torch.equal(d_in, d_dec)
print(chain, ent, kind, "enc %.1f ms (%.0f MB/s) dec %.1f ms (%.0f MB/s) ratio %.3f skip %s" % ((t1 - t0) * 1e3, B * bs / (t1 - t0) / 1e6, (t2 - t1) * 1e3, B * bs / (t2 - t1) / 1e6,
      sum((r.bits + 7) // 8 for r in res) / (B * bs), sorted(set(r.skipFlags for r in res))))
kt = ctx.kernel_times()
print("  ", {k: round(v["ms"], 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])[:10]})
