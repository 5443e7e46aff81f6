"""Level-exact -l 5 on the text-heavy mix, one batch, with the host-stage pipeline traced (KZ_TRACE_PIPE=1):
   python tools/level5_probe.py [blocks] [chunk]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kanzi_amd as kz
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if len(sys.argv) > 2:
    os.environ["KZ_HOST_CHUNK"] = sys.argv[2]
bs = 4 << 20
host = bench.text_mix(16, bs)
ctx = kz.Context(0); ctx.set_block_size(bs)
d = torch.from_numpy(host).cuda()
d_in = d.repeat((B + 15) // 16, 1)[:B].contiguous()
os_ = kz.max_block_stream_bytes(bs)
d_enc = torch.zeros((B, os_), dtype=torch.uint8, device="cuda")
d_dec = torch.zeros((B, bs), dtype=torch.uint8, device="cuda")
lens = np.full(B, bs, dtype=np.int32)
for chain in ("TEXT+UTF+BWT+RANK+ZRLT", "BWT+RANK+ZRLT"):
    for rep in range(2):
        t0 = time.perf_counter()
        res = kz.encode_blocks(ctx, chain, "ANS0", d_in.data_ptr(), bs, lens, d_enc.data_ptr(), os_, kz.MEM_DEVICE)
        t1 = time.perf_counter()
        bits = np.array([r.bits for r in res], dtype=np.int64)
        kz.decode_blocks(ctx, chain, "ANS0", bs, d_enc.data_ptr(), os_, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
        t2 = time.perf_counter()
        print("%s rep %d: enc %.2f s (%.0f MB/s) dec %.2f s (%.0f MB/s) c=%.3f" % (chain, rep, t1 - t0, B * bs / (t1 - t0) / 1e6, t2 - t1, B * bs / (t2 - t1) / 1e6, bits.sum() / 8 / (B * bs)), flush=True)
assert torch.equal(d_in, d_dec)
