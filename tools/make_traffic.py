#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed summaries
profiles/<tag>_{bench.log,kernel_stats.csv,pmc_*.csv,traffic.json}.   usage: python tools/make_traffic.py r01k"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
KERN = ("encode_fused_kernel", "encode_frames_kernel", "compact_frames_kernel", "decode_frames_kernel", "float32_range_kernel")


def one(pattern):
    files = glob.glob(os.path.join(src, pattern), recursive=True)
    if not files:
        raise SystemExit("missing " + pattern)
    return files[0]


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if not any(k in name for k in KERN):
            continue
        key = name.replace("void ", "").split("(")[0]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


shutil.copy(os.path.join(src, f"{tag}_bench.log"), os.path.join(dst, f"{tag}_bench.log"))
shutil.copy(one(f"{tag}_stats/**/*kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
pf, pw, pi = (one(f"{tag}_pmc{x}/**/*counter_collection.csv") for x in "FWI")


def copy_own_kernels(path, out):
    """keep the rows of this library's kernels only (torch's fill / copy kernels make up most of the raw file)"""
    rows = list(csv.reader(open(path)))
    ki = rows[0].index("Kernel_Name")
    w = csv.writer(open(out, "w", newline=""))
    w.writerow(rows[0])
    w.writerows(r for r in rows[1:] if "fa::" in r[ki])


copy_own_kernels(pf, os.path.join(dst, f"{tag}_pmc_fetch_size.csv"))
copy_own_kernels(pw, os.path.join(dst, f"{tag}_pmc_write_size.csv"))
copy_own_kernels(pi, os.path.join(dst, f"{tag}_pmc_sq_insts.csv"))
F, W, I = counters(pf), counters(pw), counters(pi)
out = {
    "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 1`; "
    "counters are KB per launch. hbm_bytes applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts half of "
    "a wide 16 B/lane coalesced read) only to encode_frames_kernel, whose reads are such; the 64-byte-per-lane chunk reads "
    "of decode_frames_kernel and the 4 B/lane reads of compact_frames_kernel are uncalibrated and reported raw.",
    "kernels": {},
    "instruction_mix": {},
}
for k in F:
    corr = 2.0 if ("encode_frames" in k or "encode_fused" in k or "float32_range" in k) else 1.0
    f, w = F[k]["FETCH_SIZE"], W.get(k, {}).get("WRITE_SIZE", 0.0)
    out["kernels"][k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "fetch_correction": corr, "hbm_bytes": (f * corr + w) * 1024.0}
for k, c in I.items():
    if "compact" in k:
        continue
    waves = c.get("SQ_WAVES", 0.0) or 1.0
    out["instruction_mix"][k] = {
        "waves": waves,
        "valu_per_wave": round(c.get("SQ_INSTS_VALU", 0.0) / waves, 1),
        "salu_per_wave": round(c.get("SQ_INSTS_SALU", 0.0) / waves, 1),
        "lds_per_wave": round(c.get("SQ_INSTS_LDS", 0.0) / waves, 1),
    }
# issue side (separate PMC pass): share of a wave's life in which it issues vector instructions, times the waves that
# share a SIMD (K3F: 3 by its 168 registers; K7: 2 by its 228 registers -- the 16.9 KB of LDS per workgroup would allow 9 per CU)
WAVES_PER_SIMD = {"encode_fused_kernel": 3.0, "decode_frames_kernel": 2.0, "encode_frames_kernel": 2.0}
pu = glob.glob(os.path.join(src, f"{tag}_pmcU/**/*counter_collection.csv"), recursive=True)
if pu:
    copy_own_kernels(pu[0], os.path.join(dst, f"{tag}_pmc_issue.csv"))
    out["issue"] = {}
    for k, c in counters(pu[0]).items():
        wps = next((v for n, v in WAVES_PER_SIMD.items() if n in k), None)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if not wps or not wc:
            continue
        out["issue"][k] = {
            "waves_per_simd": wps,
            "valu_active_share_of_wave_life": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
            "lds_active_share_of_wave_life": round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 4),
            "wait_any_share": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4),
            "wait_inst_any_share": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
            "issue_frac": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc * wps, 4),
        }
# FETCH_SIZE calibration for K7's read shape: the micro-benchmark's reads-only kernel moves a known number of bytes
pc = glob.glob(os.path.join(src, f"{tag}_pmcC/**/*counter_collection.csv"), recursive=True)
if pc:
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(pc[0]))
            if "frame_stream<64, 128, 17408, false, 1>" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    if vals:
        known = 2048 * 256 * 9392.0  # bytes the kernel reads (frames x bytes per frame, tools/ubench/frame_stream.hip)
        kb = sum(vals) / len(vals)
        factor = known / (kb * 1024.0)
        out["k7_fetch_calibration"] = {"known_bytes": known, "FETCH_SIZE_KB": kb, "bytes_per_counted_byte": round(factor, 4),
                                       "note": "same per-lane 4 x 16 B chunk reads as K7; K7's hbm_bytes below uses this factor"}
        for k, rec in out["kernels"].items():
            if "decode_frames" in k:
                rec["fetch_correction"] = round(factor, 4)
                rec["hbm_bytes"] = (rec["FETCH_SIZE_KB"] * factor + rec["WRITE_SIZE_KB"]) * 1024.0
json.dump(out, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
