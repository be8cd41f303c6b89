#!/bin/bash
# HBM-side traffic of the encode kernel for library variants (rocprofv3 --pmc, one counter per pass as
# MI355X_MICROARCH.md prescribes):  bash tools/pmc_variants.sh <tag> name ...   -> gpurun_out/<tag>_<name>_{W,F}.csv
tag=$1; shift
export TMPDIR=/tmp
for name in "$@"; do
  export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip_${name}.so
  [ "$name" = shipped ] && export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip.so
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/${tag}_${name}_$c -o run -- python tools/kbench.py --channels 1024 --reps 1 > gpurun_out/${tag}_${name}_$c.log 2>&1
    f=$(find gpurun_out/${tag}_${name}_$c -name '*counter_collection.csv' | head -1)
    python - "$f" "$name" "$c" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "fa::" in n and ("encode" in n or "compact" in n):
        acc[n.split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{sys.argv[2]:10s} {sys.argv[3]:10s} {k:45s} launches {len(v)} mean {sum(v)/len(v)/1e6:9.3f} GB (counter KB/1e6)")
PY
  done
done
