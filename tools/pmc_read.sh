#!/bin/bash
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_read -o run -- python tools/read_one.py > gpurun_out/pmc_read.log 2>&1
f=$(find gpurun_out/pmc_read -name '*counter_collection.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fa::decode_frames" in n:
        g = int(r.get("Grid_Size", 0) or 0)
        key = "small" if g <= 64 * 8 else "big"
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
    print(k, "launches", len(cs["SQ_WAVES"]), "waves/launch", w, {c: round(sum(v) / len(v) / w, 1) for c, v in cs.items() if c != "SQ_WAVES"})
PY
