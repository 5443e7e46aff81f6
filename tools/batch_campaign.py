"""Ragged batches through kz_encode_blocks / kz_decode_blocks (host and device buffers) vs the oracle block by block.
SEEDS=1,2 CASES=60 python tools/batch_campaign.py   (diagnostic)"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch
import kanzi_amd as kz, oracle, refinputs, datagen
from test_gpu_parity import _fuzz_input

ctx = kz.Context(0)
chains = ["BWT+RANK+ZRLT", "BWT+SRT+ZRLT", "LZ", "LZX", "PACK+MM+LZX", "DNA+LZ", "MM", "PACK", "ZRLT", "BWT+MTFT+ZRLT", "NONE"]
ents = ["ANS0", "HUFFMAN", "FPAQ", "NONE"]
seeds = [int(x) for x in os.environ.get("SEEDS", "1,2").split(",")]
cases = int(os.environ.get("CASES", "60"))
bad = 0; nblk = 0
for seed in seeds:
    rng = np.random.default_rng(seed)
    for case in range(cases):
        B = int(rng.integers(1, 70)); bs = int(rng.choice([4096, 65536, 1 << 18]))
        chain, ent = chains[int(rng.integers(0, len(chains)))], ents[int(rng.integers(0, len(ents)))]
        chk = int(rng.choice([0, 32, 64])); dev = bool(rng.integers(0, 2))
        lens = np.array([min(bs, int(rng.choice([0, 1, 15, 16, 17, 100, 1023, 1024, 1025, 5000, bs, int(rng.integers(0, bs + 1))]))) for _ in range(B)], dtype=np.int32)
        inp = np.zeros((B, bs), dtype=np.uint8)
        for b in range(B):
            n = int(lens[b])
            if n == 0: continue
            k = int(rng.integers(0, 3))
            d = _fuzz_input(rng, n) if k == 0 else (np.frombuffer(refinputs.multimedia_like(int(rng.integers(0, 5)), n, seed=b), dtype=np.uint8) if k == 1 else datagen.block(int(rng.integers(0, 40)), n))
            inp[b, :n] = d
        ostride = kz.max_block_stream_bytes(bs)
        ctx.set_checksum(chk)
        if dev:
            d_in = torch.from_numpy(inp).cuda(); d_out = torch.zeros((B, ostride), dtype=torch.uint8, device="cuda"); d_dec = torch.zeros((B, bs), dtype=torch.uint8, device="cuda")
            res = kz.encode_blocks(ctx, chain, ent, d_in.data_ptr(), bs, lens, d_out.data_ptr(), ostride, kz.MEM_DEVICE)
            out = d_out.cpu().numpy()
        else:
            out = np.zeros((B, ostride), dtype=np.uint8)
            res = kz.encode_blocks(ctx, chain, ent, inp, bs, lens, out, ostride)
        bits = np.array([r.bits for r in res], dtype=np.int64)
        okc = True
        for b in range(B):
            n = int(lens[b])
            ref, rbits = oracle.encode_block(chain, ent, bytes(inp[b, :n]), checksum=chk)[:2]
            nblk += 1
            if n == 0:
                if res[b].bits != 0: okc = False; print("   empty block bits", res[b].bits)
                continue
            if res[b].status != 0 or res[b].bits != rbits or bytes(out[b, :(rbits + 7) // 8]) != ref[:(rbits + 7) // 8]:
                okc = False; print("   block", b, "n", n, "status", res[b].status, "bits", res[b].bits, "oracle bits", rbits)
        if dev:
            r2 = kz.decode_blocks(ctx, chain, ent, bs, d_out.data_ptr(), ostride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE); dec = d_dec.cpu().numpy()
        else:
            dec = np.zeros((B, bs), dtype=np.uint8); r2 = kz.decode_blocks(ctx, chain, ent, bs, out, ostride, bits, dec, bs)
        for b in range(B):
            n = int(lens[b])
            if r2[b].status != 0 or r2[b].length != n or bytes(dec[b, :n]) != bytes(inp[b, :n]):
                okc = False; print("   decode block", b, "n", n, "status", r2[b].status, "len", r2[b].length)
        if not okc:
            bad += 1; print("FAIL seed", seed, "case", case, chain, ent, "B", B, "bs", bs, "chk", chk, "dev", dev, flush=True)
ctx.set_checksum(0)
print("blocks", nblk, "failing batches", bad)
