#!/bin/bash
# instruction counts of the encode kernel for library variants:  bash tools/pmc_insts.sh <tag> name ...
tag=$1; shift
export TMPDIR=/tmp
for name in "$@"; do
  export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip_${name}.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/${tag}_${name}_I -o run -- python tools/kbench.py --channels 1024 --reps 1 > gpurun_out/${tag}_${name}_I.log 2>&1
  f=$(find gpurun_out/${tag}_${name}_I -name '*counter_collection.csv' | head -1)
  python - "$f" "$name" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fa::" in n and ("encode" in n or "compact" in n):
        acc[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
    print(f"{sys.argv[2]:10s} {k:42s} waves {w:9.0f}  per wave: VALU {sum(cs['SQ_INSTS_VALU'])/len(cs['SQ_INSTS_VALU'])/w:8.1f}  SALU {sum(cs['SQ_INSTS_SALU'])/len(cs['SQ_INSTS_SALU'])/w:8.1f}  LDS {sum(cs['SQ_INSTS_LDS'])/len(cs['SQ_INSTS_LDS'])/w:7.1f}")
PY
done
