#!/usr/bin/env python3
"""bench.py -- encode+decode MB/s of the HIP block pipeline on a synthetic 4 MiB-block stream.

One "step" = one pass of the hot path over one batch: kz_encode_blocks (BWT+RANK+ZRLT & ANS0, the
level-5 core chain) followed by kz_decode_blocks of the produced block streams, inputs and outputs
resident in HBM.  Blocks are independent (K/io/CompressedOutputStream.java:792,907): with N GPUs
block g goes to rank g mod N, no collective on the data path.

  python bench.py --gpus N --steps K --warmup W
With N > 1 and no WORLD_SIZE in the environment the script re-executes itself under
torch.distributed.run (one rank per GPU); launched by torch.distributed.run it uses the ranks it is given.
Rank 0 prints ONE JSON line:
  value              the bulk batch (--blocks 4 MiB blocks per GPU per step, weak scaling), timed without instrumentation
  roofline, kernels  from one extra instrumented step (HIP events around every launch, on the stream it is launched on):
                     roofline.kernel = the kernel with the largest SUMMED GPU time of the step (rocprofv3's top row);
                     achieved = the algorithmic bytes of that kernel's pipeline stage / the summed time of the
                     stage's kernels; traffic = PMC bytes per launch of that kernel from profiles/ (separate --pmc passes)
  config.chains      the other BASELINE configs through the same harness, one short timed pass each, each with its own
                     roofline: configs[1] LZ & ANS0, configs[4] BWT+SRT+ZRLT & FPAQ, and the level-exact -l 5 chain
                     TEXT+UTF+BWT+RANK+ZRLT & ANS0 on a text-heavy mix (host TEXT / UTF stages inside the timed region)
  config.shapes      the headline chain on batches of the size the metric names, split round-robin over the ranks
                     (strong scaling): silesia (50 x 4 MiB + 2 242 560 B; also one row per synthetic class), enwik9
                     (238 x 4 MiB + 1 755 648 B), device resident; and the host-buffer, PCIe-inclusive kz_compress /
                     kz_decompress rates (silesia shape, and a bulk stream of --bulk-host-blocks blocks)
  cpu_baseline       the C oracle on this box's host cores (rank 0) + the reference's published row
"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # before the first HIP call of the process (the four-stream decoder schedule)
import json
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
SILESIA_BYTES = 211957760      # 50 x 4 MiB + 2 242 560
ENWIK9_BYTES = 1000000000      # 238 x 4 MiB + 1 755 648 (config 4)
# /root/reference/README.md:86 (Kanzi 2.5.0, silesia.tar -l 5, Ryzen 9950X, 16 jobs): 1717 ms / 752 ms
REFERENCE_PUBLISHED = {"source": "flanglet/kanzi README.md:86, silesia.tar -l 5, AMD Ryzen 9950X, Java 25, default jobs",
                       "encode_MBps": 123.4, "decode_MBps": 281.9, "enc_dec_MBps": 85.8,
                       "note": "other hardware; includes the TEXT+UTF stages of -l 5; no JVM on this box to run it here"}

# kernel -> pipeline stage (for the algorithmic-byte attribution of SURVEY.md 8d)
KERNEL_STAGE = {}
for _st, _ks in {
    "bwt_fwd": ("k_bwt_init", "k_radix_hist", "k_radix_scan", "k_radix_scatter", "k_seg_reduce", "k_seg_scan", "k_seg_apply",
                "k_live_emit", "k_bwt_emit", "k_msd_hist", "k_msd_scan", "k_msd_scatter", "k_bucket_sort", "k_bucket_count", "k_bucket_count_s",
                "k_tr_hist16", "k_tr_assign", "k_tr_count", "k_tr_scatter", "k_tr_sort", "k_trk_hist16", "k_trk_count", "k_trk_scatter", "k_trk_sort"),
    "sbrt_fwd": ("k_sbrt_last2", "k_sbrt_scan", "k_sbrt_replay"),
    "zrlt_fwd": ("k_zrlt_f1", "k_zrlt_f2", "k_zrlt_f3", "k_zrlt_ffin"),
    "ans_enc": ("k_ans_enc_chunk", "k_ans_enc_scan", "k_ans_enc_concat"),
    "ans_dec": ("k_ans_dec_index", "k_ans_dec_chunk", "k_ans_dec_fin"),
    "huf_enc": ("k_huf_enc_chunk",), "huf_dec": ("k_huf_dec_index", "k_huf_dec_chunk", "k_huf_dec_fin"),
    "fpaq_enc": ("k_fpaq_enc", "k_fpaq_pack", "k_fpaq_model"), "fpaq_dec": ("k_fpaq_dec",),
    "zrlt_inv": ("k_zrlt_i1", "k_zrlt_i2", "k_zrlt_i3", "k_zrlt_ifin"),
    "sbrt_inv": ("k_sbrt_inverse",),
    "srt_fwd": ("k_srt_hist", "k_srt_prep", "k_srt_scatter"), "srt_inv": ("k_srt_inv",),
    "lz_fwd": ("k_lz_fwd",), "lz_inv": ("k_lz_inv",),
    "text_fwd": ("k_text_fwd", "k_text_walk"),        # the TEXT forward on the device (statistics / emit passes, the dictionary walk)
    "text_inv": ("k_text_inv", "k_utf_inv"),          # the TEXT / UTF inverses on the device (their forms and passes share a kernel id each)
    "bwt_inv": ("k_bwti_parse", "k_bwti_hist", "k_bwti_scan", "k_bwti_scatter", "k_bwti_walk1", "k_bwti_resolve", "k_bwti_copy",
                "k_bwti_literal", "k_bwti_fin", "k_bwti_ord"),
}.items():
    for _k in _ks:
        KERNEL_STAGE[_k] = _st


def stage_alg_bytes_per_input_byte(chain, entropy, z, c):
    """algorithmic HBM bytes per input byte, per stage.  z = post-transform length / n (ZRLT-out or
    LZ-out), c = compressed / n.  BWT+RANK+ZRLT&ANS0: ENC 13+3z+c, DEC 14+2z+c; BWT+SRT+ZRLT&FPAQ: ENC 14+2z+c,
    DEC 14+2z+c; LZ*&HUFFMAN|ANS0: ENC 1+3l+c, DEC 1+2l+c (l = z).  TEXT / UTF are host stages: no HBM bytes."""
    names = chain.upper().split("+")
    st = {}
    if "BWT" in names:
        st["bwt_fwd"], st["bwt_inv"] = 10.0, 11.0
    if "RANK" in names or "MTFT" in names:
        st["sbrt_fwd"], st["sbrt_inv"] = 2.0, 2.0
    if "SRT" in names:
        st["srt_fwd"], st["srt_inv"] = 3.0, 2.0           # count + code + write ; read + write
        st["sbrt_fwd"] = 0.0                               # the MTF replay SRT shares with RANK is part of srt_fwd's 3 B/B
    if "ZRLT" in names:
        st["zrlt_fwd"], st["zrlt_inv"] = 1.0 + z, z + 1.0
    if "LZ" in names or "LZX" in names:
        st["lz_fwd"], st["lz_inv"] = 1.0 + z, z + 1.0
    e = entropy.upper()
    if e == "ANS0":
        st["ans_enc"], st["ans_dec"] = 2.0 * z + c, c + z
    elif e == "HUFFMAN":
        st["huf_enc"], st["huf_dec"] = 2.0 * z + c, c + z
        st["ans_enc"] = 0.0                                # the bit-concat kernels Huffman shares with ANS0
    elif e == "FPAQ":
        st["fpaq_enc"], st["fpaq_dec"] = z + c, c + z
    enc = sum(v for k, v in st.items() if k.endswith("_fwd") or k.endswith("_enc"))
    dec = sum(v for k, v in st.items() if k.endswith("_inv") or k.endswith("_dec"))
    return st, enc, dec


def kernel_table(ktimes, chain):
    """rows sorted by SUMMED GPU time (what rocprofv3 --stats ranks by).  Launches of one kernel that the decoder runs side by
    side on its side streams (k_sbrt_inverse: one per cost class) also carry the longest launch = what the step waits for."""
    srt = "SRT" in chain.upper().split("+")
    rows = []
    for name, v in ktimes.items():
        st = KERNEL_STAGE.get(name, "frame")
        if srt and st == "sbrt_fwd":
            st = "srt_fwd"
        k = {"kernel": name, "ms_per_step": v["ms"], "launches_per_step": v["launches"], "stage": st}
        if v["launches"] > 1:
            k["longest_launch_ms"] = v["max_ms"]
        rows.append(k)
    rows.sort(key=lambda k: -k["ms_per_step"])
    return rows


def roofline_of(kernels, per_stage_alg, alg_enc, alg_dec, step_bytes, t_enc, t_dec, steps, traffic, copy_gbs):
    """roofline of the step's dominant kernel = largest summed GPU time.  achieved = the algorithmic bytes of that kernel's
    pipeline stage / the summed time of ALL kernels of the stage (a stage of 17 kernels is not credited to one of them);
    traffic = measured HBM bytes per launch of the kernel (PMC, separate passes) when a matching profile is committed."""
    if not kernels:
        return None
    stage_ms = {}
    for k in kernels:
        stage_ms[k["stage"]] = stage_ms.get(k["stage"], 0.0) + k["ms_per_step"]
    dom = next((k for k in kernels if per_stage_alg.get(k["stage"], 0.0) > 0), kernels[0])
    st = dom["stage"]
    alg = per_stage_alg.get(st, 0.0) * step_bytes                # algorithmic bytes of the stage per step
    sms = stage_ms[st]
    achieved = alg / (sms * 1e-3) / 1e9 if sms > 0 else 0.0
    launches = max(dom["launches_per_step"], 1)
    tk = traffic["kernels"].get(dom["kernel"]) if traffic else None
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": tk["hbm_bytes_per_launch"] if tk else None,
         "kernel": dom["kernel"], "stage": st, "kernel_ms_per_step": dom["ms_per_step"], "launches_per_step": launches,
         "avg_launch_ms": dom["ms_per_step"] / launches, "stage_ms_per_step": sms, "stage_kernels": sum(1 for k in kernels if k["stage"] == st),
         "stage_alg_bytes_per_step": alg, "alg_bytes_per_launch": alg / launches,
         "alg_bytes_per_input_byte": {"encode": alg_enc, "decode": alg_dec, "stage": per_stage_alg.get(st, 0.0)},
         "pipeline_enc_GBs": alg_enc * step_bytes * steps / t_enc / 1e9, "pipeline_dec_GBs": alg_dec * step_bytes * steps / t_dec / 1e9,
         "pipeline_frac": (alg_enc + alg_dec) * step_bytes * steps / (t_enc + t_dec) / 1e9 / HBM_PEAK_GBS}
    if copy_gbs:
        r["measured_copy_GBs"] = copy_gbs
    if tk:
        r["traffic_source"] = traffic.get("_path")
        r["traffic_GBs"] = tk["hbm_bytes_per_launch"] * launches / (dom["ms_per_step"] * 1e-3) / 1e9
        r["traffic_frac_of_peak"] = r["traffic_GBs"] / HBM_PEAK_GBS
    # per-stage table: algorithmic bytes over summed kernel time, every stage of the chain
    r["stages"] = {s: {"ms_per_step": stage_ms.get(s, 0.0), "alg_bytes_per_input_byte": a,
                       "achieved_GBs": (a * step_bytes / (stage_ms[s] * 1e-3) / 1e9) if stage_ms.get(s) else None}
                   for s, a in per_stage_alg.items() if a > 0}
    # the largest kernel that IS bandwidth bound, next to it (a serial per-block chain like the RANK inverse is bound by the
    # instruction issue of one wave per block, not by HBM: DESIGN.md 4)
    if traffic:
        hb = next((k for k in kernels if k is not dom and k["stage"] in ("bwt_fwd", "bwt_inv") and k["kernel"] in traffic["kernels"]), None)
        if hb is not None:
            hl = max(hb["launches_per_step"], 1)
            hbytes = traffic["kernels"][hb["kernel"]]["hbm_bytes_per_launch"]
            r["largest_hbm_bound_kernel"] = {"kernel": hb["kernel"], "stage": hb["stage"], "ms_per_step": hb["ms_per_step"], "launches_per_step": hl,
                                             "avg_launch_ms": hb["ms_per_step"] / hl, "traffic": hbytes,
                                             "traffic_GBs": hbytes * hl / (hb["ms_per_step"] * 1e-3) / 1e9,
                                             "traffic_frac_of_peak": hbytes * hl / (hb["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    return r


def _r(x, nd=4):
    """floats to nd significant digits (the printed line must stay small enough for the driver's record)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    return x


def compact_roofline(r):
    if not r:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "stage", "avg_launch_ms", "launches_per_step",
            "kernel_ms_per_step", "stage_ms_per_step", "alg_bytes_per_launch", "pipeline_frac", "measured_copy_GBs", "traffic_GBs", "traffic_source")
    return {k: _r(r[k]) for k in keep if k in r}


def compact(out):
    """the ONE printed JSON line: the contract's keys, roofline and cpu_baseline, and for every other BASELINE config
    (config.chains) and every shaped batch (config.shapes) its rates and the roofline of its dominant kernel"""
    c = out["config"]
    cc = {k: _r(v) for k, v in c.items() if k not in ("chains", "shapes")}
    cc["chains"] = {}
    for key, ch in c["chains"].items():
        cc["chains"][key] = {"chain": ch["chain"] + " & " + ch["entropy"], "data": ch["data"], "blocks": ch["blocks_per_gpu_per_step"],
                             "enc": _r(ch["encode_MBps"]), "dec": _r(ch["decode_MBps"]), "enc_dec": _r(ch["enc_dec_MBps"]),
                             "ms_per_step": _r(ch["ms_per_step"]), "z": _r(ch["z_post_transform_ratio"]), "c": _r(ch["c_compressed_ratio"]),
                             "roofline": compact_roofline(ch["roofline"])}
        if ch.get("host_stage_ms_per_step"):
            cc["chains"][key]["text_utf_stage_ms_incl_device_forms"] = {k: _r(v) for k, v in ch["host_stage_ms_per_step"].items()}
        for a in ("knz_identical_to_hip", "blocks_compared_with_oracle"):
            if a in ch:
                cc["chains"][key][a] = ch[a]
        if ch.get("host_share8"):
            cc["chains"][key]["host_share8"] = {k: _r(v) for k, v in ch["host_share8"].items() if k != "what"}
    cc["shapes"] = {}
    for key, sh in c["shapes"].items():
        if key == "silesia_by_class":
            cc["shapes"][key] = {n: _r(v["enc_dec_MBps"]) for n, v in sh.items()}
            continue
        row = {k: _r(v) for k, v in sh.items() if k in ("blocks", "bytes", "scaling", "chain", "file", "knz_bytes", "knz_bytes_reference", "vs_reference_published", "steps", "ms_per_step")}
        for a, b in (("encode_MBps", "enc"), ("decode_MBps", "dec"), ("compress_MBps", "enc"), ("decompress_MBps", "dec"), ("enc_dec_MBps", "enc_dec")):
            if a in sh:
                row[b] = _r(sh[a])
        for a in ("knz_identical_to_hip", "blocks_compared_with_oracle"):
            if a in sh:
                row[a] = sh[a]
        cc["shapes"][key] = row
    o = {k: _r(v) for k, v in out.items() if k not in ("config", "roofline", "kernels", "cpu_baseline")}
    o["config"] = cc
    o["roofline"] = compact_roofline(out["roofline"])
    if out.get("roofline"):
        o["roofline"]["stages_ms_GBs"] = {s: [_r(v["ms_per_step"]), _r(v["achieved_GBs"])] for s, v in out["roofline"].get("stages", {}).items()}
    cb = out.get("cpu_baseline")
    if cb:
        o["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "sample", "encode_MBps", "decode_MBps", "knz_identical_to_hip") if k in cb}
        o["cpu_baseline"]["reference_published_enc_dec_MBps"] = REFERENCE_PUBLISHED["enc_dec_MBps"]
        if "single_thread" in cb:
            o["cpu_baseline"]["single_thread"] = {k: _r(v) for k, v in cb["single_thread"].items()}
        o["cpu_baseline"]["usable_cpus"] = cb["host"]["usable_cpus"]
    else:
        o["cpu_baseline"] = None
    o["detail"] = "per-kernel tables and per-stage rooflines of THIS run: %s (--detail-json; tools/profile_round.sh copies its own run's file to profiles/%s_bench_kernels.json)" % (out.get("_detail_path", "?"), out.get("_tag", "r06"))
    return o


def load_traffic(path, B, chain, entropy):
    if not os.path.exists(path):                                       # this round's PMC passes not collected (yet): the newest earlier round's,
        import glob                                                    # named as such in roofline.traffic_source
        base = os.path.basename(path).split("_", 1)[1]
        older = sorted(glob.glob(os.path.join(os.path.dirname(path), "r[0-9][0-9]_" + base)))
        if older:
            path = older[-1]
    try:
        with open(path) as f:
            tj = json.load(f)
    except (OSError, ValueError):
        return None
    if tj.get("blocks_per_gpu_per_step") != B or tj.get("chain", "BWT+RANK+ZRLT") != chain or tj.get("entropy", "ANS0") != entropy:
        return None
    # where the PMC bytes come from: the committed profile and the commit it was measured at (tools/profile_round.sh stamps
    # KZ_GIT_SHA into the file; the kernel TIMES next to it are this run's own)
    tj["_path"] = "%s @ %s" % (os.path.relpath(path, ROOT), tj.get("git_sha") or "unstamped")
    return tj


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def text_mix(D, bs):
    """text-heavy stand-in for the level-exact rows (levels 5 / 6 lead with TEXT+UTF): of every 8 blocks 5 English-like, 1
    XML-like, 1 UTF-8 (Cyrillic) and 1 binary (records, SURVEY 8d class 2)"""
    import textgen
    import datagen
    out = np.empty((D, bs), dtype=np.uint8)
    for i in range(D):
        k = i % 8
        if k == 5:
            out[i] = textgen.bulk_text(bs, 1000 + i, "xml")
        elif k == 6:
            out[i] = textgen.bulk_text(bs, 1000 + i, "utf8")
        elif k == 7:
            out[i] = datagen.block(i, bs, 2)
        else:
            out[i] = textgen.bulk_text(bs, 1000 + i, "english")
    return out


# silesia.tar by member, in 4 MiB blocks (211 957 760 B): dickens 2.4, webster 9.9, reymont 1.6, samba 5.1, xml 1.3 (text, source,
# mark-up: 20) + nci 8.0 (chemical database text, highly redundant); mozilla 12.2, ooffice 1.5 (executables: 14); mr 2.4, sao 1.7,
# x-ray 2.0 (poorly compressible binary: 6); osdb 2.4 (database records).  The stand-in keeps those proportions:
SILESIA_MIX = (("text-like (english / source / xml / utf-8)", 24), ("executable-like, mid entropy", 14), ("poorly compressible binary", 7),
               ("64-byte records", 3), ("highly redundant (90% zeros)", 3))


def silesia_mix(bs):
    """51 blocks in silesia.tar's member proportions, members kept together as in the tar"""
    import textgen
    import datagen
    out = np.empty((51, bs), dtype=np.uint8)
    i = 0
    for k in range(24):
        kind = "xml" if k in (9, 10) else ("utf8" if k in (11, 12) else "english")
        out[i] = textgen.bulk_text(bs, 2000 + k, kind)
        i += 1
    for k in range(14):
        out[i] = datagen.exe_like(bs, 3000 + k)
        i += 1
    for k in range(7):
        out[i] = datagen.sensor_like(bs, 4000 + k)
        i += 1
    for k in range(3):
        out[i] = datagen.block(5000 + k, bs, 2)
        i += 1
    for k in range(3):
        out[i] = datagen.block(6000 + k, bs, 4)
        i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=2048, help="4 MiB blocks per GPU per step of the bulk batch")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic blocks generated per GPU; batches tile them (blocks are coded independently)")
    ap.add_argument("--block-size", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--chain", default="BWT+RANK+ZRLT")
    ap.add_argument("--entropy", default="ANS0")
    ap.add_argument("--data", default="mix", choices=("mix", "text"), help="mix = SURVEY 8d generator; text = the text-heavy mix of the level-exact rows")
    ap.add_argument("--input", default="", help="a corpus file (silesia.tar, enwik9 ...): the headline, every chain and every shape read ITS bytes in "
                    "--block-size blocks instead of the synthetic generator, and shapes.input reports the level-exact .knz size next to the README's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shapes", action="store_true", help="skip the silesia / enwik9 shaped batches and the host-buffer rates")
    ap.add_argument("--no-chains", action="store_true", help="skip config.chains (the other BASELINE configs)")
    ap.add_argument("--no-two-streams", action="store_true", help="skip shapes.bulk_two_streams (a second context: encode of step k+1 under the decode of step k)")
    ap.add_argument("--chain-steps", type=int, default=2, help="timed steps of each config.chains row")
    ap.add_argument("--bulk-host-blocks", type=int, default=2048, help="blocks of the PCIe-inclusive bulk stream (0 = skip)")
    ap.add_argument("--data-class", type=int, default=-1, help="diagnostic: force one class of the synthetic generator (0..4) instead of the mix")
    ap.add_argument("--detail-json", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="the full record (per-kernel tables, per-stage rooflines of every chain) is written here; the printed line is the compact form")
    ap.add_argument("--profiles-tag", default="r06", help="profiles/<tag>_pmc_traffic*.json: per-kernel HBM bytes from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)")
    ap.add_argument("--traffic-json", default="", help="override the PMC traffic file of the headline chain")
    ap.add_argument("--cpu-sample-blocks", type=int, default=0, help="0 = auto (about 10-30 s of CPU work)")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start one rank per GPU ourselves (the driver's `python bench.py --gpus N`) ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

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
    kz.pin_host_threads_to_gpu(local_rank, world)                     # N ranks on one host: host stages stay on the GPU's NUMA node
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(vals):
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if share else dev)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    B, bs = args.blocks, args.block_size
    ctx = kz.Context(local_rank)
    ctx.set_block_size(bs)                                             # the stream's "blockSize" entry (TEXT sizes its hash map by it)
    # ---- synthetic stream: the D distinct blocks of global ids i*world + rank (round-robin over ranks), tiled ----
    D = min(args.distinct, B)
    input_bytes = None
    if args.input:
        raw = np.fromfile(args.input, dtype=np.uint8)
        if raw.size == 0:
            raise SystemExit("--input %s is empty" % args.input)
        input_bytes = int(raw.size)
        nfile = (raw.size + bs - 1) // bs
        # this rank's share of the file's blocks (block g -> rank g mod N), the ragged last block zero padded for the tiled batches
        # (shapes.input codes the exact bytes)
        mine = list(range(rank, nfile, world)) or [0]
        D = len(mine)
        host = np.zeros((D, bs), dtype=np.uint8)
        for j, g in enumerate(mine):
            blk = raw[g * bs:(g + 1) * bs]
            host[j, :blk.size] = blk
    elif args.data == "text":
        host = text_mix(D, bs)
    else:
        host = np.empty((D, bs), dtype=np.uint8)
        for i in range(D):
            host[i] = datagen.block(i * world + rank, bs, None if args.data_class < 0 else args.data_class)
    o_stride = kz.max_block_stream_bytes(bs)

    class Batch:
        """nb blocks of this rank resident in HBM (block k = distinct block k mod D of `src`), the last one tail_len bytes long"""

        def __init__(self, src, nb, chain, entropy, tail_len=bs):
            self.nb, self.chain, self.entropy = nb, chain, entropy
            self.lengths = np.full(nb, bs, dtype=np.int32)
            if nb:
                self.lengths[-1] = tail_len
            self.nbytes = int(self.lengths.sum())
            d = src.shape[0]
            self.d_in = src.repeat((nb + d - 1) // d, 1)[:nb].contiguous() if nb else None
            self.d_enc = torch.zeros((max(nb, 1), o_stride), dtype=torch.uint8, device=dev)
            self.d_dec = torch.zeros((max(nb, 1), bs), dtype=torch.uint8, device=dev)

        def step(self):
            if self.nb == 0:
                return 0.0, 0.0, []
            t0 = time.perf_counter()
            res = kz.encode_blocks(ctx, self.chain, self.entropy, self.d_in.data_ptr(), bs, self.lengths, self.d_enc.data_ptr(), o_stride, kz.MEM_DEVICE)
            t1 = time.perf_counter()
            bits = np.array([r.bits for r in res], dtype=np.int64)
            for r in res:
                if r.status:
                    raise RuntimeError("encode status %d" % r.status)
            res2 = kz.decode_blocks(ctx, self.chain, self.entropy, bs, self.d_enc.data_ptr(), o_stride, bits, self.d_dec.data_ptr(), bs, kz.MEM_DEVICE)
            t2 = time.perf_counter()
            for k, r in enumerate(res2):
                if r.status or r.length != self.lengths[k]:
                    raise RuntimeError("decode status %d len %d" % (r.status, r.length))
            return t1 - t0, t2 - t1, res

        def round_trip_ok(self):
            if self.nb == 0:
                return True
            ok = bool(torch.equal(self.d_in[:-1], self.d_dec[:-1]))
            tl = int(self.lengths[-1])
            return ok and bool(torch.equal(self.d_in[-1, :tl], self.d_dec[-1, :tl]))

    def timed_pass(src, nb, chain, entropy, steps, warmup, traffic_path, copy_rate=False, keep=False):
        """warmup + `steps` timed steps (barrier and device sync on both sides, max over ranks), then ONE instrumented step
        (HIP events around every kernel launch; stage timers for the host TEXT / UTF stages) -> dict"""
        bt = Batch(src, nb, chain, entropy)
        for _ in range(warmup):
            bt.step()
        barrier()
        T0 = time.perf_counter()
        te = td = 0.0
        res = None
        for _ in range(steps):
            a, b, res = bt.step()
            te += a
            td += b
        barrier()
        T1 = time.perf_counter()
        elapsed, te, td = max_over_ranks([T1 - T0, te, td])
        if not bt.round_trip_ok():
            raise SystemExit("round trip mismatch (%s & %s): decoded blocks differ from the input" % (chain, entropy))
        step_bytes = float(bt.nbytes)
        comp_bytes = float(sum((r.bits + 7) // 8 for r in res))
        post_bytes = float(sum(r.length for r in res))
        z, c = post_bytes / step_bytes, comp_bytes / step_bytes
        ctx.set_kernel_timing(True)
        ctx.reset_kernel_timing()
        bt.step()
        torch.cuda.synchronize()
        ctx.set_kernel_timing(False)
        ktimes = ctx.kernel_times()
        host_ms = None
        if "TEXT" in chain.upper().split("+") or "UTF" in chain.upper().split("+"):
            ctx.set_timing(True)
            ctx.reset_timing()
            ctx.lib.kz_host_stage_blocks(0, 1)
            ctx.lib.kz_host_stage_blocks(1, 1)
            bt.step()
            torch.cuda.synchronize()
            ctx.set_timing(False)
            stt = ctx.stage_times()
            # wall time of the TEXT / UTF stage INCLUDING its device forms' kernels, and how many blocks of that step really went
            # through a host thread (kz_host_stage_blocks)
            host_ms = {"forward": stt.get("host_fwd", {}).get("ms", 0.0), "inverse": stt.get("host_inv", {}).get("ms", 0.0),
                       "forward_blocks_on_host_threads": int(ctx.lib.kz_host_stage_blocks(0, 0)), "inverse_blocks_on_host_threads": int(ctx.lib.kz_host_stage_blocks(1, 0))}
        copy_gbs = None
        if copy_rate:                                                  # measured stream-copy rate of this GPU, next to the 8 TB/s spec peak
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            bt.d_dec.copy_(bt.d_in)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                bt.d_dec.copy_(bt.d_in)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 3 * 2.0 * nb * bs / (e0.elapsed_time(e1) * 1e-3) / 1e9
        per_stage_alg, alg_enc, alg_dec = stage_alg_bytes_per_input_byte(chain, entropy, z, c)
        kernels = kernel_table(ktimes, chain)
        traffic = load_traffic(traffic_path, nb, chain, entropy) if (traffic_path and args.data_class < 0) else None
        if traffic:
            for k in kernels:
                t = traffic["kernels"].get(k["kernel"])
                if t and k["ms_per_step"] > 0:
                    k["hbm_traffic_GBs"] = t["hbm_bytes_per_launch"] * k["launches_per_step"] / (k["ms_per_step"] * 1e-3) / 1e9
        roof = roofline_of(kernels, per_stage_alg, alg_enc, alg_dec, step_bytes, te, td, steps, traffic, copy_gbs)
        out = {"chain": chain, "entropy": entropy, "blocks_per_gpu_per_step": nb, "steps": steps, "warmup": warmup,
               "elapsed": elapsed, "t_enc": te, "t_dec": td, "step_bytes": step_bytes, "z": z, "c": c,
               "enc_dec_MBps": step_bytes * world * steps / elapsed / 1e6, "encode_MBps": step_bytes * world * steps / te / 1e6,
               "decode_MBps": step_bytes * world * steps / td / 1e6, "ms_per_step": elapsed / steps * 1e3,
               "roofline": roof, "kernels": kernels, "host_stage_ms_per_step": host_ms}
        if keep:                                                       # a sample of the HIP block streams for the oracle comparison in the cpu_baseline leg
            nsm = min(nb, src.shape[0], 16)                            # (rows 0 .. d-1 of the batch are the d distinct blocks)
            out["_sample"] = [(int(res[k].bits), int(res[k].skipFlags), int(res[k].length), bt.d_enc[k, :(int(res[k].bits) + 7) // 8].cpu().numpy().tobytes(),
                               src[k, :int(bt.lengths[k])].cpu().numpy().tobytes()) for k in range(nsm)]
        return out

    d_host = torch.from_numpy(host).to(dev)
    tag = args.profiles_tag
    prof = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))

    # ================= headline: the bulk batch, weak scaling, no instrumentation inside the timed region =================
    head = timed_pass(d_host, B, args.chain, args.entropy, args.steps, args.warmup, args.traffic_json or prof("pmc_traffic.json"), copy_rate=True)
    step_bytes, z, c = head["step_bytes"], head["z"], head["c"]
    value = head["enc_dec_MBps"]
    nk = int(os.environ.get("KZ_BENCH_KERNELS", "12"))

    # ================= two independent streams on one GPU: encode of batch k+1 under the decode of batch k =================
    def two_streams(src, nb, steps):
        """The decoder is one serial RANK chain per block (two waves per SIMD) and leaves most of the GPU's issue slots idle, the encoder
        is throughput-bound: a second context encodes step k+1 while the first decodes step k (two host threads, two contexts, two
        encoded buffers).  Every step is still one encode and one decode of the whole batch; the row is the rate of `steps` such steps.
        NOT the headline (whose steps run one after the other): what a service that compresses one stream while it expands another gets."""
        import threading
        ctx2 = kz.Context(local_rank)
        ctx2.set_block_size(bs)
        bt = Batch(src, nb, args.chain, args.entropy)
        encs = [bt.d_enc, torch.zeros_like(bt.d_enc)]
        errs = []

        def enc(k):
            res = kz.encode_blocks(ctx2, bt.chain, bt.entropy, bt.d_in.data_ptr(), bs, bt.lengths, encs[k & 1].data_ptr(), o_stride, kz.MEM_DEVICE)
            if any(r.status for r in res):
                raise RuntimeError("encode status")
            return np.array([r.bits for r in res], dtype=np.int64)

        def dec(k, bits):
            res = kz.decode_blocks(ctx, bt.chain, bt.entropy, bs, encs[k & 1].data_ptr(), o_stride, bits, bt.d_dec.data_ptr(), bs, kz.MEM_DEVICE)
            if any(r.status or r.length != bt.lengths[j] for j, r in enumerate(res)):
                raise RuntimeError("decode status")

        dec(0, enc(0))                                                 # warm-up: both contexts size their arenas
        encoded = [threading.Event() for _ in range(steps)]
        decoded = [threading.Event() for _ in range(steps)]
        bits_of = [None] * steps

        def thread_e():
            try:
                for k in range(steps):
                    if k >= 2:
                        decoded[k - 2].wait()                          # the buffer of step k-2 is free again
                    bits_of[k] = enc(k)
                    encoded[k].set()
            except Exception as e:                                     # noqa: BLE001 -- reported after the join
                errs.append(e)
                for ev in encoded:
                    ev.set()

        def thread_d():
            try:
                for k in range(steps):
                    encoded[k].wait()
                    if errs:
                        return
                    dec(k, bits_of[k])
                    decoded[k].set()
            except Exception as e:                                     # noqa: BLE001
                errs.append(e)
                for ev in decoded:
                    ev.set()

        barrier()
        S0 = time.perf_counter()
        te_, td_ = threading.Thread(target=thread_e), threading.Thread(target=thread_d)
        te_.start(); td_.start(); te_.join(); td_.join()
        barrier()
        S1 = time.perf_counter()
        if errs:
            raise SystemExit("two-stream pass failed: %r" % errs[0])
        if not bt.round_trip_ok():
            raise SystemExit("round trip mismatch in the two-stream pass")
        ctx2.close()
        (sel,) = max_over_ranks([S1 - S0])
        return {"blocks": nb * world, "bytes": int(bt.nbytes) * world, "scaling": "weak", "steps": steps, "ms_per_step": sel / steps * 1e3,
                "enc_dec_MBps": bt.nbytes * world * steps / sel / 1e6,
                "chain": "%s & %s; two contexts on one GPU, the encode of step k+1 under the decode of step k (not the headline's schedule)" % (args.chain, args.entropy)}

    two = None
    if not args.no_two_streams and not args.no_shapes:
        two = two_streams(d_host, B, 4)
        torch.cuda.empty_cache()

    # ================= config.chains: the other BASELINE configs, same harness, one short pass each =================
    chains = {}
    oracle_checks = []          # (section, row, chain, entropy, sample): HIP outputs kept for the cpu_baseline leg, where the oracle may be used
    if not args.no_chains:
        d_text = None
        for key, chain, entropy, data, what in (
                ("lz_ans0", "LZ", "ANS0", "mix", "configs[1]: silesia.tar -l 3 shorthand (LZ+ANS) as the explicit chain -t LZ -e ANS0"),
                ("bwt_srt_zrlt_fpaq", "BWT+SRT+ZRLT", "FPAQ", "mix", "configs[4]: synthetic 8 GiB mixed-entropy stream, the level-6 core chain (HBM-roofline report)"),
                ("level5_exact", "TEXT+UTF+BWT+RANK+ZRLT", "ANS0", "text", "level-exact -l 5 (BlockCompressor.java:539-573) on a text-heavy mix; the TEXT forward, the TEXT and UTF inverses run on the device, the UTF forward of the UTF-8 blocks on host threads, all inside the timed region")):
            if data == "text" and d_text is None:
                d_text = d_host if args.input else torch.from_numpy(text_mix(min(16, D), bs)).to(dev)
            r = timed_pass(d_text if data == "text" else d_host, B, chain, entropy, args.chain_steps, 1, prof("pmc_traffic_%s.json" % key), keep=(key == "level5_exact"))
            chains[key] = {"what": what, "chain": chain, "entropy": entropy, "data": ("file %s" % os.path.basename(args.input)) if args.input else "synthetic %s" % data,
                           "blocks_per_gpu_per_step": B, "steps": r["steps"], "warmup": r["warmup"], "ms_per_step": r["ms_per_step"],
                           "encode_MBps": r["encode_MBps"], "decode_MBps": r["decode_MBps"], "enc_dec_MBps": r["enc_dec_MBps"],
                           "z_post_transform_ratio": r["z"], "c_compressed_ratio": r["c"], "round_trip_ok": True,
                           "host_stage_ms_per_step": r["host_stage_ms_per_step"], "roofline": r["roofline"], "kernels": r["kernels"][:6]}
            if "_sample" in r:
                oracle_checks.append(("chains", key, chain, entropy, r["_sample"]))
            if key == "level5_exact":
                # what ONE of eight ranks on a node gets: the host pool capped to 1/8 of the box's CPUs (TEXT / UTF run on the host;
                # VERDICT r4 item 4).  One timed step, same batch.
                lib = kz.load_library()
                cpus8 = int(lib.kz_host_share(8))
                nb8 = min(B, 512)                                       # a quarter of the batch: the pass runs at an eighth of the host's rate
                r8 = timed_pass(d_text, nb8, chain, entropy, 1, 0, "")
                lib.kz_host_share(max(world, 1))
                chains[key]["host_share8"] = {"host_cpus": cpus8, "blocks": nb8, "encode_MBps": r8["encode_MBps"], "decode_MBps": r8["decode_MBps"], "enc_dec_MBps": r8["enc_dec_MBps"],
                                              "what": "the same row with the library's host pool capped to 1/8 of this box's usable CPUs (kz_host_share(8)): one rank's share on an 8-GPU node"}
        del d_text
        torch.cuda.empty_cache()

    # ================= shaped batches: the sizes the metric names, split over the ranks (strong scaling) =================
    shapes = {"bulk": {"blocks": B * world, "bytes": int(step_bytes) * world, "scaling": "weak",
                       "encode_MBps": head["encode_MBps"], "decode_MBps": head["decode_MBps"], "enc_dec_MBps": value}}
    if two:
        shapes["bulk_two_streams"] = two

    def shaped(src, total, reps=2, chain=None, entropy=None, by_id=False):
        nblk = (total + bs - 1) // bs
        mine = list(range(rank, nblk, world))                         # block g -> rank g mod N
        tail = total - (nblk - 1) * bs if (mine and mine[-1] == nblk - 1) else bs
        if by_id and mine:                                            # src holds the stream's blocks by global id (not this rank's tiles)
            src = src[[g % src.shape[0] for g in mine]]
        sb = Batch(src, len(mine), chain or args.chain, entropy or args.entropy, tail)
        sb.step()
        barrier()
        S0 = time.perf_counter()
        se = sd = 0.0
        for _ in range(reps):
            a, b, _r = sb.step()
            se += a
            sd += b
        barrier()
        S1 = time.perf_counter()
        sel, se, sd = max_over_ranks([S1 - S0, se, sd])
        if not sb.round_trip_ok():
            raise SystemExit("round trip mismatch in a shaped batch")
        return {"blocks": nblk, "bytes": total, "scaling": "strong", "blocks_on_rank0": len(mine),
                "encode_MBps": total * reps / se / 1e6, "decode_MBps": total * reps / sd / 1e6, "enc_dec_MBps": total * reps / sel / 1e6}

    def knz_size(hdata, chain, entropy):
        """the .knz this library writes for a host buffer (rank 0, untimed): the aggregate known-answer of a corpus run"""
        cap = int(ctx.lib.kz_compress_bound(hdata.size, bs))
        knz = np.empty(cap, dtype=np.uint8)
        return int(ctx.check(ctx.lib.kz_compress(ctx.h, kz.transform_type(chain), kz.ENTROPY_IDS[entropy.upper()], bs, hdata.ctypes.data, hdata.size, knz.ctypes.data, cap)))

    L5 = ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0")
    if not args.no_shapes and args.input:
        # the corpus at its own size: the core chain and the level-exact -l 5 chain on the file's exact bytes (ragged tail included)
        d_file = torch.from_numpy(np.fromfile(args.input, dtype=np.uint8))
        nfile = (input_bytes + bs - 1) // bs
        pad = torch.zeros(nfile * bs, dtype=torch.uint8)
        pad[:input_bytes] = d_file
        d_file = pad.view(nfile, bs).to(dev)
        shapes["input"] = shaped(d_file, input_bytes, by_id=True)
        shapes["input"]["file"] = os.path.basename(args.input)
        if not args.no_chains:
            shapes["input_level5_exact"] = shaped(d_file, input_bytes, chain=L5[0], entropy=L5[1], by_id=True)
            shapes["input_level5_exact"]["chain"] = "%s & %s on %s" % (L5[0], L5[1], os.path.basename(args.input))
            if rank == 0:
                raw = np.fromfile(args.input, dtype=np.uint8)
                shapes["input_level5_exact"]["knz_bytes"] = knz_size(raw, *L5)
                shapes["input_level5_exact"]["knz_bytes_reference"] = ({"silesia.tar -l 5 -b 4m (README.md:86)": 53853702} if input_bytes == SILESIA_BYTES else None)
                del raw
        del d_file, pad
        torch.cuda.empty_cache()
    if not args.no_shapes and not args.input:
        for name, total in (("silesia", SILESIA_BYTES), ("enwik9", ENWIK9_BYTES)):
            shapes[name] = shaped(d_host, total)
            torch.cuda.empty_cache()
        if not args.no_chains:
            # the metric's own workload at its own size: silesia.tar -l 5 = TEXT+UTF+BWT+RANK+ZRLT & ANS0 (BlockCompressor.java:539-573)
            # on 51 blocks, TEXT / UTF on host threads inside the timed region.  Two mixes: silesia.tar's member proportions (the row to
            # quote against the README's 85.8 MB/s) and the text-heavy mix of rounds 3 / 4 (7/8 text: the cheap class)
            h_mix = silesia_mix(bs)
            d_mix = torch.from_numpy(h_mix).to(dev)
            mixdesc = ", ".join("%d %s" % (cnt, nm) for nm, cnt in SILESIA_MIX)
            shapes["silesia_mix"] = shaped(d_mix, SILESIA_BYTES, by_id=True)
            shapes["silesia_mix"]["chain"] = "%s & %s; 51 blocks: %s" % (args.chain, args.entropy, mixdesc)
            shapes["silesia_mix_level5_exact"] = shaped(d_mix, SILESIA_BYTES, chain=L5[0], entropy=L5[1], by_id=True)
            shapes["silesia_mix_level5_exact"]["chain"] = "%s & %s; 51 blocks: %s" % (L5[0], L5[1], mixdesc)
            shapes["silesia_mix_level5_exact"]["vs_reference_published"] = shapes["silesia_mix_level5_exact"]["enc_dec_MBps"] / REFERENCE_PUBLISHED["enc_dec_MBps"]
            if rank == 0:
                shapes["silesia_mix_level5_exact"]["knz_bytes"] = knz_size(h_mix.reshape(-1)[:SILESIA_BYTES], *L5)
            del d_mix, h_mix
            d_text = torch.from_numpy(text_mix(min(16, D), bs)).to(dev)
            shapes["silesia_level5_exact"] = shaped(d_text, SILESIA_BYTES, chain=L5[0], entropy=L5[1])
            shapes["silesia_level5_exact"]["chain"] = "TEXT+UTF+BWT+RANK+ZRLT & ANS0, text-heavy mix (7/8 text: not silesia's composition)"
            del d_text
            torch.cuda.empty_cache()
    if not args.no_shapes:
        if args.data == "mix" and args.data_class < 0 and not args.input:
            # the silesia shape one synthetic class at a time: small-batch decode is the serial RANK inverse of the slowest block,
            # so the worst class sets the mixed row
            names = ("text-like", "geometric skew", "64-byte records", "uniform random", "90% zeros")
            per = {}
            for cl in range(5):
                rows = [i for i in range(D) if (i * world + rank) % 5 == cl]
                if rows:
                    per[names[cl]] = {k: v for k, v in shaped(d_host[rows], SILESIA_BYTES, reps=1).items() if k.endswith("MBps")}
            shapes["silesia_by_class"] = per
            torch.cuda.empty_cache()
        # host-buffer (PCIe-inclusive) rates through the stream entry points, rank 0's GPU; SURVEY 8d "two timings"
        if rank == 0:
            tt, et = kz.transform_type(args.chain), kz.ENTROPY_IDS[args.entropy.upper()]

            def host_rate(nbytes, reps, tt=tt, et=et, blocks=None, keep_first=0):
                nblk = (nbytes + bs - 1) // bs
                src_blocks = host if blocks is None else blocks
                hdata = np.ascontiguousarray(np.tile(src_blocks, ((nblk + src_blocks.shape[0] - 1) // src_blocks.shape[0], 1))[:nblk]).reshape(-1)[:nbytes]
                cap = int(ctx.lib.kz_compress_bound(hdata.size, bs))
                knz = np.empty(cap, dtype=np.uint8)
                back = np.empty(hdata.size, dtype=np.uint8)
                best = None
                for _ in range(reps):
                    t0 = time.perf_counter()
                    m = ctx.check(ctx.lib.kz_compress(ctx.h, tt, et, bs, hdata.ctypes.data, hdata.size, knz.ctypes.data, cap))
                    t1 = time.perf_counter()
                    r = ctx.check(ctx.lib.kz_decompress(ctx.h, knz.ctypes.data, m, back.ctypes.data, hdata.size))
                    t2 = time.perf_counter()
                    if r != hdata.size or not np.array_equal(back, hdata):
                        raise SystemExit("kz_compress / kz_decompress round trip mismatch")
                    row = {"bytes": int(hdata.size), "knz_bytes": int(m), "compress_MBps": hdata.size / (t1 - t0) / 1e6,
                           "decompress_MBps": hdata.size / (t2 - t1) / 1e6, "enc_dec_MBps": hdata.size / (t2 - t0) / 1e6}
                    if best is None or row["enc_dec_MBps"] > best["enc_dec_MBps"]:
                        best = row
                best["what"] = "kz_compress / kz_decompress on host buffers: H2D, codec, D2H and host bit assembly inside the timed region; one GPU"
                if keep_first:                                         # the first blocks' bit strings of the .knz just written, for the oracle comparison
                    off = np.zeros(keep_first, dtype=np.int64)
                    nbits = np.zeros(keep_first, dtype=np.int64)
                    nb_ = int(ctx.lib.kz_knz_index(knz.ctypes.data, m, None, None, None, None, None, off.ctypes.data, nbits.ctypes.data, keep_first))
                    kf = min(keep_first, max(nb_, 0))
                    head_bytes = knz[:int((off[kf - 1] + nbits[kf - 1] + 7) // 8) + 8].tobytes() if kf else b""
                    best["_sample"] = ([(int(off[i]), int(nbits[i]), kz.extract_bits(head_bytes, int(off[i]), int(nbits[i]))) for i in range(kf)],
                                       hdata[:kf * bs].tobytes())
                return best

            shapes["input_host_pcie" if args.input else "silesia_host_pcie"] = host_rate(input_bytes if args.input else SILESIA_BYTES, 2)
            if args.bulk_host_blocks > 0:
                shapes["bulk_host_pcie"] = host_rate(args.bulk_host_blocks * bs, 2)
                if not args.no_chains and not args.input:
                    # the level-exact -l 5 chain through the same entry points, text-heavy mix: the chunks of kz_compress's pipeline run
                    # the TEXT stage on the device, kz_decompress the TEXT / UTF inverses
                    row = host_rate(args.bulk_host_blocks * bs, 2, kz.transform_type(L5[0]), kz.ENTROPY_IDS[L5[1]], text_mix(min(16, D), bs), keep_first=min(16, D, args.bulk_host_blocks))
                    row["chain"] = "%s & %s, text-heavy mix" % L5
                    if "_sample" in row:
                        oracle_checks.append(("shapes", "bulk_level5_host_pcie", L5[0], L5[1], row.pop("_sample")))
                    shapes["bulk_level5_host_pcie"] = row
        barrier()

    out = {
        "metric": "encode+decode MB/s, 4 MiB-block synthetic stream, %s & %s (level-5 core chain), bit-exact .knz" % (args.chain, args.entropy),
        "value": value, "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": ("file %s (%d B)" % (os.path.basename(args.input), input_bytes)) if args.input else "synthetic",
        "config": {"workload": "configs[2]: %s & %s, %d x %d B blocks per GPU per step (%s, tiled), blocks round-robin over ranks" % (args.chain, args.entropy, B, bs, ("the %d blocks of %s on this rank" % (D, os.path.basename(args.input))) if args.input else ("%d distinct synthetic blocks per GPU, SURVEY 8d generator standing in for silesia.tar" % D)),
                   "block_size": bs, "blocks_per_gpu_per_step": B, "parallelism": "blocks%%%d" % world,
                   "encode_MBps": head["encode_MBps"], "decode_MBps": head["decode_MBps"],
                   "z_post_transform_ratio": z, "c_compressed_ratio": c, "round_trip_ok": True,
                   "chains": chains, "shapes": shapes},
        "roofline": head["roofline"],
        "kernels": head["kernels"][:nk],
        "_tag": tag, "_detail_path": os.path.relpath(args.detail_json, ROOT),
    }

    # ---- CPU baseline: the oracle (C restatement) on this box's host cores, bounded sample; rank 0, once ----
    if rank == 0 and not args.no_cpu_baseline:
        import oracle
        logical = os.cpu_count() or 1
        kz.load_library().kz_host_share(1)                             # the other ranks sleep during this leg: the whole quota
        usable = kz.usable_cpus()                                      # affinity mask cut down to the cgroup CPU quota (the MI355X box: 256 logical, quota 16)
        what = "oracle/libkzo.so (C restatement, -O3 -march=x86-64-v3, induced-sorting BWT), %d threads over blocks"

        def cpu_row(jobs, ns):
            reps = (ns + D - 1) // D
            sample = np.ascontiguousarray(np.tile(host, (reps, 1))[:ns]).reshape(-1)
            t0 = time.perf_counter()
            knz = oracle.compress(args.chain, args.entropy, bs, sample, jobs=jobs)
            t1 = time.perf_counter()
            back = oracle.decompress(knz, len(sample), jobs=jobs)
            t2 = time.perf_counter()
            assert back == sample.tobytes()
            return {"value": len(sample) / (t2 - t0) / 1e6, "unit": "MB/s", "cores": jobs,
                    "encode_MBps": len(sample) / (t1 - t0) / 1e6, "decode_MBps": len(sample) / (t2 - t1) / 1e6,
                    "sample": ("%d blocks (%d B; the %d distinct blocks tiled) of the same stream; " + what + "; enc %.2f s dec %.2f s") % (ns, len(sample), D, jobs, t1 - t0, t2 - t1)}

        # thread counts: what the process may actually burn (CPU quota), and the reference's default job count min(logical / 2, 64)
        # (BlockCompressor.java:199-203); about 4 blocks per thread each, blocks are independent
        cands = sorted({max(1, min(usable, 256)), max(1, min(logical // 2, 64))})
        rows = [cpu_row(j, args.cpu_sample_blocks or int(min(max(D, 4 * j), 512))) for j in cands]
        best = max(rows, key=lambda r: r["value"])
        # parity of the HIP output on a sub-sample (not timed): identical .knz bytes
        npar = min(D, 8)
        psample = np.ascontiguousarray(host[:npar]).reshape(-1)
        cos = kz.CompressedOutputStream(ctx, args.chain, args.entropy, bs)
        cos.write(psample.tobytes())
        cos.close()
        pref = oracle.compress(args.chain, args.entropy, bs, psample, jobs=max(1, min(usable, 16)))
        cb = dict(best)
        cb.update({"kind": "port", "knz_identical_to_hip": bool(cos.output == pref), "reference_published": REFERENCE_PUBLISHED,
                   "host": {"logical_cpus": logical, "usable_cpus": usable,
                            "note": "usable = affinity mask cut down to the cgroup CPU quota; more busy threads than that only get throttled"}})
        others = [r for r in rows if r is not best]
        if others:
            cb["other_thread_count_row"] = others[0]
        out["cpu_baseline"] = cb
        if cos.output != pref:
            raise SystemExit("PARITY FAILURE: HIP .knz differs from the oracle on the cpu_baseline sample")
        # the level-exact rows (device TEXT / UTF kernels at 4 MiB blocks): a sample of their HIP outputs against the oracle, not against
        # the HIP decoder.  chains.*: block streams, bit counts, skip flags, lengths of up to 16 distinct blocks of the
        # bulk batch == oracle.encode_block; shapes.bulk_level5_host_pcie: the first 16 blocks' bit strings inside kz_compress's .knz ==
        # the same blocks' bit strings inside the oracle's .knz of those 16 blocks
        from concurrent.futures import ThreadPoolExecutor
        for section, key, chain_, ent_, sample in oracle_checks:
            row = out["config"][section].get(key)
            if row is None:
                continue
            if section == "chains":
                with ThreadPoolExecutor(max(1, min(usable, 16))) as ex:
                    want = list(ex.map(lambda t: oracle.encode_block(chain_, ent_, t[4], block_size=bs), sample))
                same = all((w[1], w[2], w[3]) == (t[0], t[1], t[2]) and w[0] == t[3] for w, t in zip(want, sample))
                n_cmp = len(sample)
            else:
                blocks_, raw = sample
                pref2 = oracle.compress(chain_, ent_, bs, raw, jobs=max(1, min(usable, 16)))
                idx = kz.knz_index(pref2)["blocks"]
                same = len(idx) == len(blocks_) and all(b[1] == o[1] and b[2] == kz.extract_bits(pref2, o[0], o[1]) for b, o in zip(blocks_, idx))
                n_cmp = len(blocks_)
            row["knz_identical_to_hip"] = bool(same)
            row["blocks_compared_with_oracle"] = n_cmp
            if not same:
                raise SystemExit("PARITY FAILURE: %s.%s: HIP output differs from the oracle on the sampled blocks" % (section, key))
        # one thread alone (its suffix array stays in cache): the reference's README row implies about 7.7 MB/s per thread for encode
        # INCLUDING TEXT+UTF on its 16-core host
        one = np.ascontiguousarray(host[:min(D, 5)]).reshape(-1)
        t0 = time.perf_counter()
        k1 = oracle.compress(args.chain, args.entropy, bs, one, jobs=1)
        t1 = time.perf_counter()
        oracle.decompress(k1, len(one), jobs=1)
        t2 = time.perf_counter()
        out["cpu_baseline"]["single_thread"] = {"blocks": min(D, 5), "encode_MBps": len(one) / (t1 - t0) / 1e6, "decode_MBps": len(one) / (t2 - t1) / 1e6}
    elif rank == 0:
        out["cpu_baseline"] = None
    if dist is not None:
        # the other ranks wait for rank 0's CPU leg asleep on the rendezvous store's socket, not spinning in a GPU collective (under
        # one cgroup CPU quota a spinning rank takes a CPU away from the leg being timed); then the ordinary barrier
        try:
            import datetime
            store = dist.distributed_c10d._get_default_store()
            if rank == 0:
                store.set("kz_bench_cpu_leg", "done")
            else:
                store.wait(["kz_bench_cpu_leg"], datetime.timedelta(minutes=30))
        except Exception:
            pass
        barrier()
    if rank == 0:
        # the full record goes to a file (tools/profile_round.sh copies it to profiles/<tag>_bench_kernels.json); the printed line is
        # the compact form: every BASELINE config and shape with its rates and its roofline, no per-kernel tables
        try:
            os.makedirs(os.path.dirname(args.detail_json), exist_ok=True)
            with open(args.detail_json, "w") as f:
                json.dump(out, f, indent=1)
        except OSError:
            pass
        print(json.dumps(compact(out), separators=(",", ":")))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
