#!/usr/bin/env python3
"""bench.py -- encode+decode MB/s of the HIP block pipeline on a synthetic 4 MiB-block stream.

One "step" = one pass of the hot path over one batch: kz_encode_blocks (BWT+RANK+ZRLT & ANS0, the
level-5 core chain) followed by kz_decode_blocks of the produced block streams, inputs and outputs
resident in HBM.  Blocks are independent (K/io/CompressedOutputStream.java:792,907): with N GPUs
block g goes to rank g mod N, no collective on the data path (weak scaling: per-GPU work fixed).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched via torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# kernel -> pipeline stage (for the algorithmic-byte attribution of SURVEY.md 8d)
KERNEL_STAGE = {}
for k in ("k_bwt_init", "k_radix_hist", "k_radix_scan", "k_radix_scatter", "k_seg_reduce", "k_seg_scan",
          "k_seg_apply", "k_live_count", "k_live_scan", "k_live_emit", "k_bwt_emit"):
    KERNEL_STAGE[k] = "bwt_fwd"
for k in ("k_sbrt_last2", "k_sbrt_scan", "k_sbrt_replay"):
    KERNEL_STAGE[k] = "sbrt_fwd"
for k in ("k_zrlt_f1", "k_zrlt_f2", "k_zrlt_f3", "k_zrlt_ffin"):
    KERNEL_STAGE[k] = "zrlt_fwd"
for k in ("k_ans_enc_chunk", "k_ans_enc_scan", "k_ans_enc_concat"):
    KERNEL_STAGE[k] = "ans_enc"
for k in ("k_ans_dec_index", "k_ans_dec_chunk", "k_ans_dec_fin"):
    KERNEL_STAGE[k] = "ans_dec"
for k in ("k_zrlt_i1", "k_zrlt_i2", "k_zrlt_i3", "k_zrlt_ifin"):
    KERNEL_STAGE[k] = "zrlt_inv"
KERNEL_STAGE["k_sbrt_inverse"] = "sbrt_inv"
for k in ("k_bwti_parse", "k_bwti_hist", "k_bwti_scan", "k_bwti_scatter", "k_bwti_walk1", "k_bwti_resolve", "k_bwti_copy", "k_bwti_literal", "k_bwti_fin"):
    KERNEL_STAGE[k] = "bwt_inv"


def stage_alg_bytes_per_input_byte(z, c):
    """algorithmic HBM bytes per input byte, per stage. z = ZRLT-out/n, c = compressed/n.
    ENC total 13+3z+c, DEC total 14+2z+c."""
    return {"bwt_fwd": 10.0, "sbrt_fwd": 2.0, "zrlt_fwd": 1.0 + z, "ans_enc": 2.0 * z + c,
            "ans_dec": c + z, "zrlt_inv": z + 1.0, "sbrt_inv": 2.0, "bwt_inv": 11.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=2048, help="4 MiB blocks per GPU per step (one wave per block in the serial kernels: throughput comes from blocks in flight)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic blocks generated per GPU; the step's blocks tile them (blocks are coded independently)")
    ap.add_argument("--block-size", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--chain", default="BWT+RANK+ZRLT")
    ap.add_argument("--entropy", default="ANS0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--data-class", type=int, default=-1, help="diagnostic: force one class of the synthetic generator (0..4) instead of the mix")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r01_pmc_traffic.json"),
                    help="per-kernel HBM bytes from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)")
    ap.add_argument("--cpu-sample-blocks", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    args = ap.parse_args()

    import torch
    import kanzi_amd as kz
    import datagen

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # dry run of the N>1 path on a one-GPU box (diagnostic only): KZ_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and
    # rendezvous goes over gloo, since RCCL refuses two ranks on one device
    share = os.environ.get("KZ_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    B, bs = args.blocks, args.block_size
    ctx = kz.Context(local_rank)
    # ---- synthetic stream: global block g = i*world + rank (round-robin over ranks) ----
    D = min(args.distinct, B)
    host = np.empty((D, bs), dtype=np.uint8)
    for i in range(D):
        host[i] = datagen.block(i * world + rank, bs, None if args.data_class < 0 else args.data_class)
    d_host = torch.from_numpy(host).to(dev)
    d_in = d_host.repeat((B + D - 1) // D, 1)[:B].contiguous()
    del d_host
    o_stride = kz.max_block_stream_bytes(bs)
    d_enc = torch.zeros((B, o_stride), dtype=torch.uint8, device=dev)
    d_dec = torch.zeros((B, bs), dtype=torch.uint8, device=dev)
    lengths = np.full(B, bs, dtype=np.int32)
    torch.cuda.synchronize()

    def step():
        t0 = time.perf_counter()
        res = kz.encode_blocks(ctx, args.chain, args.entropy, d_in.data_ptr(), bs, lengths, d_enc.data_ptr(), o_stride, kz.MEM_DEVICE)
        t1 = time.perf_counter()
        bits = np.array([r.bits for r in res], dtype=np.int64)
        for r in res:
            if r.status:
                raise RuntimeError("encode status %d" % r.status)
        res2 = kz.decode_blocks(ctx, args.chain, args.entropy, bs, d_enc.data_ptr(), o_stride, bits, d_dec.data_ptr(), bs, kz.MEM_DEVICE)
        t2 = time.perf_counter()
        for r in res2:
            if r.status or r.length != bs:
                raise RuntimeError("decode status %d len %d" % (r.status, r.length))
        return t1 - t0, t2 - t1, res

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_kernel_timing(True)
    ctx.reset_kernel_timing()
    barrier()
    T0 = time.perf_counter()
    t_enc = t_dec = 0.0
    res = None
    for _ in range(args.steps):
        a, b, res = step()
        t_enc += a
        t_dec += b
    barrier()
    T1 = time.perf_counter()
    ctx.set_kernel_timing(False)
    elapsed = T1 - T0
    tt = torch.tensor([elapsed, t_enc, t_dec], dtype=torch.float64, device="cpu" if share else dev)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, t_enc, t_dec = [float(x) for x in tt.tolist()]

    # ---- correctness outside the timed region: round trip identical ----
    ok = bool(torch.equal(d_in, d_dec))
    if not ok:
        raise SystemExit("round trip mismatch: decoded blocks differ from the input")

    # ---- measured stream-copy rate of this GPU, printed next to the 8 TB/s spec peak ----
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d_dec.copy_(d_in)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        d_dec.copy_(d_in)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 3 * 2.0 * B * bs / (e0.elapsed_time(e1) * 1e-3) / 1e9

    # ---- ratios + roofline ----
    step_bytes = float(B) * bs
    comp_bytes = float(sum((r.bits + 7) // 8 for r in res))
    post_bytes = float(sum(r.length for r in res))
    z, c = post_bytes / step_bytes, comp_bytes / step_bytes
    ktimes = ctx.kernel_times()
    per_stage_alg = stage_alg_bytes_per_input_byte(z, c)
    roofline = None
    kernels = []
    for name, v in ktimes.items():
        kernels.append({"kernel": name, "ms_per_step": v["ms"] / args.steps, "launches_per_step": v["launches"] / args.steps,
                        "stage": KERNEL_STAGE.get(name, "frame")})
    kernels.sort(key=lambda k: -k["ms_per_step"])
    stage_ms = {}
    for k in kernels:
        stage_ms[k["stage"]] = stage_ms.get(k["stage"], 0.0) + k["ms_per_step"]
    # measured HBM traffic per kernel (separate rocprofv3 --pmc passes, tools/pmc_traffic.py), if it matches this workload
    tj = None
    try:
        with open(args.traffic_json) as f:
            tj = json.load(f)
        if tj.get("blocks_per_gpu_per_step") != B or args.chain != "BWT+RANK+ZRLT" or args.entropy != "ANS0" or args.data_class >= 0:
            tj = None
    except (OSError, ValueError):
        tj = None
    if tj:
        for k in kernels:
            t = tj["kernels"].get(k["kernel"])
            if t and k["ms_per_step"] > 0:
                k["hbm_traffic_GBs"] = t["hbm_bytes_per_launch"] * k["launches_per_step"] / (k["ms_per_step"] * 1e-3) / 1e9
    if kernels:
        dom = kernels[0]
        st = dom["stage"]
        alg = per_stage_alg.get(st, 0.0) * step_bytes                # algorithmic bytes of the stage per step
        launches = max(dom["launches_per_step"], 1.0)
        avg_ms = dom["ms_per_step"] / launches
        achieved = (alg / launches) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        if tj and dom["kernel"] in tj["kernels"]:
            traffic = tj["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "measured_copy_GBs": copy_gbs,
                    "kernel": dom["kernel"], "stage": st, "launches_per_step": launches, "avg_launch_ms": avg_ms,
                    "alg_bytes_per_launch": alg / launches,
                    "stage_achieved_GBs": (alg / (stage_ms[st] * 1e-3) / 1e9) if stage_ms.get(st) else None,
                    "pipeline_enc_GBs": (13 + 3 * z + c) * step_bytes * args.steps / t_enc / 1e9,
                    "pipeline_dec_GBs": (14 + 2 * z + c) * step_bytes * args.steps / t_dec / 1e9}

    total_bytes = step_bytes * world * args.steps
    value = total_bytes / elapsed / 1e6

    out = {
        "metric": "encode+decode MB/s, 4 MiB-block synthetic stream, %s & %s (level-5 core chain), bit-exact .knz" % (args.chain, args.entropy),
        "value": value, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "configs[2]: %s & %s, %d x %d B blocks per GPU per step (%d distinct synthetic blocks per GPU, SURVEY 8d generator standing in for silesia.tar, tiled), blocks round-robin over ranks" % (args.chain, args.entropy, B, bs, D),
                   "block_size": bs, "blocks_per_gpu_per_step": B, "parallelism": "blocks%%%d" % world,
                   "encode_MBps": step_bytes * world * args.steps / t_enc / 1e6,
                   "decode_MBps": step_bytes * world * args.steps / t_dec / 1e6,
                   "z_post_transform_ratio": z, "c_compressed_ratio": c, "round_trip_ok": ok},
        "roofline": roofline,
        "kernels": kernels[:12],
    }

    # ---- CPU baseline: the oracle (C restatement) on this box's host cores, bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        jobs = os.cpu_count() or 1
        # bounded sample of the same workload: enough blocks to keep every host thread busy twice
        # (blocks are independent; the B generated blocks are tiled when the box has more cores)
        ns = args.cpu_sample_blocks or int(min(max(D, 2 * jobs), 512))
        reps = (ns + D - 1) // D
        sample = np.ascontiguousarray(np.tile(host, (reps, 1))[:ns]).reshape(-1)
        t0 = time.perf_counter()
        knz = oracle.compress(args.chain, args.entropy, bs, sample, jobs=jobs)
        t1 = time.perf_counter()
        back = oracle.decompress(knz, len(sample), jobs=jobs)
        t2 = time.perf_counter()
        assert back == sample.tobytes()
        # parity of the HIP output on a sub-sample (not timed): identical .knz bytes
        npar = min(D, 8)
        psample = np.ascontiguousarray(host[:npar]).reshape(-1)
        cos = kz.CompressedOutputStream(ctx, args.chain, args.entropy, bs)
        cos.write(psample.tobytes())
        cos.close()
        pref = oracle.compress(args.chain, args.entropy, bs, psample, jobs=jobs)
        out["cpu_baseline"] = {"value": len(sample) / (t2 - t0) / 1e6, "unit": "MB/s", "cores": jobs, "kind": "port",
                               "sample": "%d blocks (%d B; the %d distinct blocks tiled) of the same stream; oracle/libkzo.so (C restatement, SA-IS BWT), %d threads over blocks; enc %.2f s dec %.2f s" % (ns, len(sample), D, jobs, t1 - t0, t2 - t1),
                               "encode_MBps": len(sample) / (t1 - t0) / 1e6, "decode_MBps": len(sample) / (t2 - t1) / 1e6,
                               "knz_identical_to_hip": bool(cos.output == pref)}
        if cos.output != pref:
            raise SystemExit("PARITY FAILURE: HIP .knz differs from the oracle on the cpu_baseline sample")
        # second row: the reference's default job count min(logical CPUs / 2, 64), on a smaller sample
        jobs2 = max(1, min(jobs // 2, 64))
        if jobs2 != jobs:
            ns2 = int(min(ns, max(D, 2 * jobs2)))
            sample2 = sample[:ns2 * bs]
            t0 = time.perf_counter()
            knz2 = oracle.compress(args.chain, args.entropy, bs, sample2, jobs=jobs2)
            t1 = time.perf_counter()
            back2 = oracle.decompress(knz2, len(sample2), jobs=jobs2)
            t2 = time.perf_counter()
            assert back2 == sample2.tobytes()
            row2 = {"value": len(sample2) / (t2 - t0) / 1e6, "unit": "MB/s", "cores": jobs2,
                    "encode_MBps": len(sample2) / (t1 - t0) / 1e6, "decode_MBps": len(sample2) / (t2 - t1) / 1e6,
                    "sample": "%d blocks (%d B) of the same stream; oracle/libkzo.so (C restatement, SA-IS BWT), %d threads over blocks; enc %.2f s dec %.2f s" % (ns2, len(sample2), jobs2, t1 - t0, t2 - t1)}
            cb = out["cpu_baseline"]
            row1 = {k: cb[k] for k in ("value", "unit", "cores", "encode_MBps", "decode_MBps", "sample")}
            # the headline CPU figure is the better of the two thread counts (oversubscribing SMT threads can lose)
            best, other = (row2, row1) if row2["value"] > row1["value"] else (row1, row2)
            cb.update(best)
            cb["other_thread_count_row"] = other
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
