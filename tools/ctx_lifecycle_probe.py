"""Context life cycle: create -> compress/decompress -> destroy, many times; free device memory must come back.
python tools/ctx_lifecycle_probe.py   (diagnostic, not a test)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
import kanzi_amd as kz, datagen

data = b"".join(datagen.block(i, 1 << 20).tobytes() for i in range(8))
free0 = None
for it in range(40):
    ctx = kz.Context(0)
    cos = kz.CompressedOutputStream(ctx, "BWT+RANK+ZRLT" if it % 2 else "PACK+MM+LZX", "ANS0" if it % 2 else "HUFFMAN", 1 << 20, checksum=32)
    cos.write(data); cos.close()
    assert kz.CompressedInputStream(ctx, cos.output).read(len(data)) == data
    ctx.close() if hasattr(ctx, "close") else None
    del ctx
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    if it == 1: free0 = free
    if it % 8 == 1: print("iteration", it, "free MiB", free >> 20, flush=True)
print("drift MiB since iteration 1:", (free0 - free) >> 20)
