#!/bin/bash
tag=$1; shift
export TMPDIR=/tmp
for name in "$@"; do
  export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip_${name}.so
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${tag}_${name}_A -o run -- python tools/kbench.py --channels 1024 --reps 1 > gpurun_out/${tag}_${name}_A.log 2>&1
  f=$(find gpurun_out/${tag}_${name}_A -name '*counter_collection.csv' | head -1)
  python - "$f" "$name" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fa::encode_fused" in n:
        acc[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
    print(sys.argv[2], {c: round(sum(v) / len(v) / w, 1) for c, v in cs.items() if c != "SQ_WAVES"})
PY
done
