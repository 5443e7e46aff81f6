"""The metric's shape (bench.py shapes.silesia_mix_level5_exact: 51 blocks of 4 MiB in silesia.tar's member proportions, -l 5), one
batch, device resident: wall time of encode and decode, the library's per-stage timers and the per-kernel table of one decode.
   python tools/silesia_mix_probe.py [chain] [entropy]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kanzi_amd as kz
import bench
chain = sys.argv[1] if len(sys.argv) > 1 else "TEXT+UTF+BWT+RANK+ZRLT"
ent = sys.argv[2] if len(sys.argv) > 2 else "ANS0"
bs = 4 << 20
host = bench.silesia_mix(bs)
B = host.shape[0]
ctx = kz.Context(0); ctx.set_block_size(bs)
d_in = torch.from_numpy(host).cuda()
os_ = kz.max_block_stream_bytes(bs)
d_enc = torch.zeros((B, os_), dtype=torch.uint8, device="cuda")
d_dec = torch.zeros((B, bs), dtype=torch.uint8, device="cuda")
lens = np.full(B, bs, dtype=np.int32)
lens[-1] = bench.SILESIA_BYTES - (B - 1) * bs
total = int(lens.sum())
for rep in range(4):
    if rep == 2: ctx.set_timing(True); ctx.reset_timing()
    if rep == 3: ctx.set_timing(False); ctx.set_kernel_timing(True); ctx.reset_kernel_timing()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lens, d_enc.data_ptr(), os_, kz.MEM_DEVICE)
    t1 = time.perf_counter()
    bits = np.array([r.bits for r in res], dtype=np.int64)
    res2 = kz.decode_blocks(ctx, chain, ent, bs, d_enc.data_ptr(), os_, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
    t2 = time.perf_counter()
    print("rep %d: enc %.1f ms (%.0f MB/s) dec %.1f ms (%.0f MB/s) enc+dec %.0f MB/s = %.2fx the published Java row" % (rep, (t1 - t0) * 1e3, total / (t1 - t0) / 1e6, (t2 - t1) * 1e3, total / (t2 - t1) / 1e6, total / (t2 - t0) / 1e6, total / (t2 - t0) / 1e6 / 85.8), flush=True)
    if rep == 2:
        for k, v in ctx.stage_times().items(): print("   stage %-12s %8.1f ms" % (k, v["ms"]))
assert all(r.status == 0 for r in res2)
kt = ctx.kernel_times()
for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print("  %-18s %9.1f ms %5d launches (longest %.1f)" % (k, v["ms"], v["launches"], v["max_ms"]))
