#!/usr/bin/env python3
"""Static instruction counts of K3F between its phase marks.

    python tools/isa_phases.py [-D...more defines] [--kernel SUBSTR] [--keep]

Compiles csrc/fused_unit.hip to assembly with -DFA_PHASE_MARKS (every FA_STAMP becomes a fenced comment `; FA_MARK k`,
encode_kernels.hpp) and counts, for the chosen kernel (default: encode_fused_kernel<8, false>), the instructions between
consecutive marks in layout order: VALU (of which f64 and DPP), SALU, LDS, global / scratch memory, waits.  The hot path
of the kernel is laid out in one piece from the first mark to the last; blocks the compiler moved behind s_endpgm (rare
paths: wide samples, slow partition search, VERBATIM) are reported as one "cold" line.  Backward branches inside a range
are listed as loops (label, body size) -- their bodies are counted ONCE in the table; `--trips name=n,...` multiplies.
No GPU needed.  The table goes to stdout as markdown.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = {
    "init": "P0 stage the frame (16 row loads, image store, min / max / or)",
    "0": "P2 fixed predictors 0-4 (lane sums over A then B)",
    "1": "P2 wave sums, order choice, (unfused) partition search",
    "2": "P3 lag products (9 f64 chains per sample) + fused fixed partition search",
    "4": "P3 nine wave sums (butterfly)",
    "5": "Levinson-Durbin, order choice, coefficient quantisation",
    "6": "P4 LPC residual in place + magnitude sums",
    "7": "LPC partition search, winner bookkeeping",
    "8": "materialise the winner (FIXED recompute), exact size of the winner",
    "9": "publish size, writer state, look-back issue, preamble",
    "10": "rows: Rice codes -> ring, flush of completed blocks (loop bodies counted once)",
    "11": "offset resolve at the tail",
    "13": "tail: CRC-16 of the remaining words, last stores",
    "12": "(end)",
}


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="encode_fused_kernelILi8ELb0E")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--trips", default="")
    ap.add_argument("defines", nargs="*")
    args, extra = ap.parse_known_args()
    defines = [d for d in extra + args.defines if d.startswith("-D") or d.startswith("-m")]
    tmp = tempfile.mkdtemp(prefix="isa_")
    out = os.path.join(tmp, "fused.s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-DFA_DEV_MINIMAL",
           "-DFA_PHASE_MARKS", "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "flacarray_amd", "csrc", "fused_unit.hip")] + defines
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(args.kernel) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    meta = {}
    for l in lines[end:end + 400]:
        m = re.match(r"\s*;\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs):\s*(\d+)", l)
        if m and m.group(1) not in meta:
            meta[m.group(1)] = int(m.group(2))
    ranges, cur, cur_name = [], None, "prologue"
    counts = lambda: {"valu": 0, "f64": 0, "dpp": 0, "salu": 0, "lds": 0, "vmem": 0, "scratch": 0, "wait": 0, "loops": []}  # noqa: E731
    cur = counts()
    labels = {}
    seen_end = False
    cold = counts()
    for i, l in enumerate(body):
        t = l.strip()
        m = re.match(r";\s*FA_MARK\s+(\S+)", t)
        if m:
            ranges.append((cur_name, cur))
            cur_name, cur = m.group(1), counts()
            if m.group(1) == "end":
                seen_end = True
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = (i, cur_name)
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        c = classify(op)
        tgt = cold if seen_end else cur
        if c:
            tgt[c] += 1
            if c == "valu":
                if "_f64" in op or "_b64" in op or "_i64" in op or "_u64" in op:
                    tgt["f64"] += 1
                if "dpp" in t or "row_" in t or "quad_perm" in t or "wave_sh" in t:
                    tgt["dpp"] += 1
        m = re.match(r"s_cbranch_\S+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m and not seen_end:
            lab = m.group(1) or m.group(2)
            if lab in labels and labels[lab][1] == cur_name:  # backward branch inside the range: a loop
                j = labels[lab][0]
                n = sum(1 for x in body[j:i] if classify((x.strip().split() or ["."])[0]) == "valu")
                cur["loops"].append((lab, n))
    ranges.append((cur_name, cur))
    trips = dict(kv.split("=") for kv in args.trips.split(",") if kv)
    print(f"kernel `{args.kernel}` {' '.join(defines)}: " + ", ".join(f"{k} {v}" for k, v in meta.items()))
    print()
    print("| after mark | phase | VALU | of which 64-bit | DPP | SALU | LDS | global | scratch | waits | loops (label: VALU in body) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    tot = counts()
    for name, c in ranges:
        if name == "end":
            continue
        for k in ("valu", "f64", "dpp", "salu", "lds", "vmem", "scratch", "wait"):
            tot[k] += c[k]
        loops = "; ".join(f"{lab}: {n}" for lab, n in c["loops"])
        print(f"| {name} | {PHASES.get(name, name)} | {c['valu']} | {c['f64']} | {c['dpp']} | {c['salu']} | {c['lds']} | {c['vmem']} | {c['scratch']} | {c['wait']} | {loops} |")
    print(f"| | **layout total (loop bodies once)** | {tot['valu']} | {tot['f64']} | {tot['dpp']} | {tot['salu']} | {tot['lds']} | {tot['vmem']} | {tot['scratch']} | {tot['wait']} | |")
    print(f"| | cold blocks behind s_endpgm | {cold['valu']} | {cold['f64']} | {cold['dpp']} | {cold['salu']} | {cold['lds']} | {cold['vmem']} | {cold['scratch']} | {cold['wait']} | |")
    if args.keep:
        print("\nassembly kept at", out, file=sys.stderr)
    else:
        os.remove(out)
        os.rmdir(tmp)


if __name__ == "__main__":
    main()
