"""One chain on the bulk synthetic batch, device resident: encode / decode seconds and the per-kernel table of one instrumented step.
   python tools/chain_probe.py BWT+SRT+ZRLT FPAQ [blocks] [data-class]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kanzi_amd as kz
import datagen
chain, ent = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
cls = int(sys.argv[4]) if len(sys.argv) > 4 else -1
bs = 4 << 20
D = min(64, B)
if cls == 6:                  # the text-heavy mix of bench.py's level-exact rows
    sys.path.insert(0, ROOT)
    import bench
    D = min(16, B)
    host = bench.text_mix(D, bs)
else:
    host = np.stack([datagen.block(i, bs, None if (cls < 0 or cls > 4) else cls) for i in range(D)])
if cls == 5:                  # the synthetic mix with ONE all-zero block in the batch (the slowest possible block of a suffix sort by doubling)
    host = np.concatenate([host, np.zeros((1, bs), dtype=np.uint8)])
    D += 1
ctx = kz.Context(0); ctx.set_block_size(bs)
d_in = torch.from_numpy(host[:64]).cuda().repeat((B + min(D, 64) - 1) // min(D, 64), 1)[:B].contiguous()
if cls == 5:
    d_in[B // 2] = 0
os_ = kz.max_block_stream_bytes(bs)
d_enc = torch.zeros((B, os_), dtype=torch.uint8, device="cuda")
d_dec = torch.zeros((B, bs), dtype=torch.uint8, device="cuda")
lens = np.full(B, bs, dtype=np.int32)
for rep in range(3):
    if rep == 2:
        ctx.set_kernel_timing(True); ctx.reset_kernel_timing()
    t0 = time.perf_counter()
    res = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lens, d_enc.data_ptr(), os_, kz.MEM_DEVICE)
    t1 = time.perf_counter()
    bits = np.array([r.bits for r in res], dtype=np.int64)
    enc_only = os.environ.get("KZ_PROBE_ENC_ONLY") == "1"          # timing experiments whose output is deliberately wrong
    res2 = [] if enc_only else kz.decode_blocks(ctx, chain, ent, bs, d_enc.data_ptr(), os_, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
    t2 = time.perf_counter()
    assert all(r.status == 0 for r in res) and all(r.status == 0 for r in res2)
    print("%s & %s B=%d rep %d: enc %.3f s (%.0f MB/s) dec %.3f s (%.0f MB/s) enc+dec %.0f MB/s" % (chain, ent, B, rep, t1 - t0, B * bs / (t1 - t0) / 1e6, t2 - t1, B * bs / (t2 - t1) / 1e6, B * bs / (t2 - t0) / 1e6), flush=True)
assert enc_only or torch.equal(d_in, d_dec)
kt = ctx.kernel_times()
for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])[:24]:
    print("  %-18s %9.1f ms %5d launches (longest %.1f)" % (k, v["ms"], v["launches"], v["max_ms"]))
