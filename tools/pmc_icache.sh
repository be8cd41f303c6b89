#!/bin/bash
# instruction-cache picture of the encode kernel:  bash tools/pmc_icache.sh <tag> <variant name | shipped>
tag=$1; name=$2
export TMPDIR=/tmp
export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip_${name}.so
[ "$name" = shipped ] && export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip.so
i=0
for set in "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_${name}_I$i -o run -- python tools/kbench.py --channels 1024 --reps 1 > gpurun_out/${tag}_${name}_I$i.log 2>&1
  f=$(find gpurun_out/${tag}_${name}_I$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fa::encode" in n or "fa::decode_frames" in n:
        acc[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    w = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
    print(k, "waves", int(w), " per wave:", {c: round(sum(v) / len(v) / w, 1) for c, v in cs.items() if c != "SQ_WAVES"})
PY
done
