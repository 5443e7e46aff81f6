"""The RANK inverse alone, per data class, a few blocks at a time (every block a lone wave: the latency the small batches pay):
   python tools/sbrt_inv_probe.py [copies] [chain] [class,class...]
prints the k_sbrt_inverse time of one decode of `copies` identical 4 MiB blocks per class, for the 64-bit-key form (default) and the
32-bit forms of rounds 2-5 (KZ_SBRT_FORM=0), after checking the round trip."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kanzi_amd as kz
import datagen, textgen
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8
chain = sys.argv[2] if len(sys.argv) > 2 else "BWT+RANK"
bs = 4 << 20
classes = [("uniform", datagen.block(3, bs, 3)), ("geometric", datagen.block(1, bs, 1)), ("text-like", datagen.block(0, bs, 0)), ("records", datagen.block(2, bs, 2)),
           ("sparse", datagen.block(4, bs, 4)), ("sensor", datagen.sensor_like(bs, 4000)), ("exe", datagen.exe_like(bs, 3000)), ("english", textgen.bulk_text(bs, 2000, "english"))]
if len(sys.argv) > 3:
    classes = [c for c in classes if c[0] in sys.argv[3].split(",")]
ctx = kz.Context(0); ctx.set_block_size(bs)
os_ = kz.max_block_stream_bytes(bs)
for name, blk in classes:
    d_in = torch.from_numpy(np.ascontiguousarray(blk)).cuda().unsqueeze(0).repeat(copies, 1).contiguous()
    d_enc = torch.zeros((copies, os_), dtype=torch.uint8, device="cuda")
    d_dec = torch.zeros((copies, bs), dtype=torch.uint8, device="cuda")
    lens = np.full(copies, bs, dtype=np.int32)
    torch.cuda.synchronize()                               # (the library runs on its own stream: torch's fills must be done)
    res = kz.encode_blocks(ctx, chain, "NONE", d_in.data_ptr(), bs, lens, d_enc.data_ptr(), os_, kz.MEM_DEVICE)
    bits = np.array([r.bits for r in res], dtype=np.int64)
    row = []
    for form in ("", "0"):
        if form: os.environ["KZ_SBRT_FORM"] = form
        else: os.environ.pop("KZ_SBRT_FORM", None)
        ctx.reload_switches()
        d_dec.zero_(); torch.cuda.synchronize()
        kz.decode_blocks(ctx, chain, "NONE", bs, d_enc.data_ptr(), os_, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
        ok = torch.equal(d_in, d_dec)
        ctx.set_kernel_timing(True); ctx.reset_kernel_timing()
        kz.decode_blocks(ctx, chain, "NONE", bs, d_enc.data_ptr(), os_, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
        torch.cuda.synchronize()
        ctx.set_kernel_timing(False)
        kt = ctx.kernel_times()
        row.append((kt.get("k_sbrt_inverse", {}).get("ms", 0.0), ok))
    print("%-10s f64 %7.1f ms %s | 32-bit %7.1f ms %s | x%.2f" % (name, row[0][0], "ok" if row[0][1] else "MISMATCH", row[1][0], "ok" if row[1][1] else "MISMATCH", row[1][0] / max(row[0][0], 1e-9)), flush=True)
