#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel HBM traffic.
Usage: tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <blocks> [chain entropy] > profiles/rNN_pmc_traffic.json
Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB
(hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024); on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read (TCC_EA0_RDREQ x 64 B with 128-B requests) -> doubled for STREAMING kernels.
Calibrated per access class in round 5 (tools/ubench_gather.hip, tools/pmc_calibrate.sh -> profiles/r05_pmc_calibration.json):
  coalesced 16 B / lane reads   : FETCH_SIZE =  8 B per 16-B access  -> x2 (the guide's case, confirmed)
  scattered 4-byte loads        : FETCH_SIZE = 64 B per access (gather and dependent chase alike) at 54.5 G accesses / s; doubled,
                                  that would be 7.0 TB/s, above the 5.4 TB/s the same box streams -> a scattered access moves ONE
                                  64-byte request, the counter is right as it stands -> x1 for the kernels in SCATTERED below
  coalesced 16 B / lane writes  : WRITE_SIZE = 16 B per 16-B access  -> x1
  scattered 4-byte stores       : WRITE_SIZE = 32 B per store         -> x1 (a 32-byte sector per store)
Kernels that mix both patterns (k_bwt_emit: a streamed suffix array + gathered text bytes) keep the x2: an upper bound."""
import csv
import json
import os
import re
import sys


# kernels the library times under one id (kz_internal.h: KZ_KERNEL_NAMES): the text-sourced first radix pass
ALIASES = {"k_radix_hist0": "k_radix_hist", "k_radix_scatter0": "k_radix_scatter",
           "k_trk_hist16": "k_tr_hist16", "k_trk_count": "k_tr_count", "k_trk_scatter": "k_tr_scatter", "k_trk_sort": "k_tr_sort",   # the key trie rounds share the trie rounds' ids
           "k_fpaq_enc_wave": "k_fpaq_enc", "k_fpaq_dec_wave": "k_fpaq_dec", "k_fpaq_dec_wave2": "k_fpaq_dec",   # the one-wave-per-block forms share their ids
           "k_tf_walk": "k_text_walk", "k_tf_stats": "k_text_fwd", "k_tf_decide": "k_text_fwd", "k_tf_init": "k_text_fwd",
           "k_tf_emit": "k_text_fwd", "k_tf_scan": "k_text_fwd", "k_tf_copy_back": "k_text_fwd"}   # the device TEXT forward: the walk, and its parallel passes under one id


# kernels whose loads are scattered 4-byte accesses (one 64-byte request each): FETCH_SIZE taken as counted
SCATTERED = {"k_bwti_walk1", "k_bwti_literal", "k_lz_fwd", "k_text_walk"}   # (k_text_walk: 8-byte map entries by hash)


def agg(path, counter):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).strip()
            name = re.sub(r"<.*", "", name.replace("void ", "")).strip()      # templated kernels: "void k_x<2>"
            name = ALIASES.get(name, name)
            d = out.setdefault(name, {"launches": 0, "sum": 0.0})
            d["launches"] += 1
            d["sum"] += float(row["Counter_Value"])
    return out


def main():
    fetch = agg(sys.argv[1], "FETCH_SIZE")
    write = agg(sys.argv[2], "WRITE_SIZE")
    blocks = int(sys.argv[3])
    chain = sys.argv[4] if len(sys.argv) > 4 else "BWT+RANK+ZRLT"
    entropy = sys.argv[5] if len(sys.argv) > 5 else "ANS0"
    res = {"blocks_per_gpu_per_step": blocks, "chain": chain, "entropy": entropy, "git_sha": os.environ.get("KZ_GIT_SHA") or None, "steps_profiled": 2,   # bench.py --steps 1 --warmup 0 runs one timed and one instrumented step: every kernel appears twice per step's launches
           
           "note": "FETCH_SIZE x2 for streaming kernels (gfx950 correction), x1 for the scattered-load kernels (fetch_factor says which; profiles/r05_pmc_calibration.json), KiB -> bytes; WRITE_SIZE as measured",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f = fetch.get(k, {"launches": 0, "sum": 0.0})
        w = write.get(k, {"launches": 0, "sum": 0.0})
        n = max(f["launches"], w["launches"], 1)
        ff = 1.0 if k in SCATTERED else 2.0
        res["kernels"][k] = {"launches": n, "fetch_factor": ff,
                             "fetch_bytes_per_launch": ff * f["sum"] * 1024.0 / n,
                             "write_bytes_per_launch": w["sum"] * 1024.0 / n,
                             "hbm_bytes_per_launch": (ff * f["sum"] + w["sum"]) * 1024.0 / n}
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
