#!/bin/bash
# issue / stall picture of the encode kernel:  bash tools/pmc_util.sh <tag> <variant name | shipped>
tag=$1; name=$2
export TMPDIR=/tmp
export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip_${name}.so
[ "$name" = shipped ] && export FLACARRAY_HIP_LIB=$PWD/flacarray_amd/lib/libflacarray_hip.so
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_CYCLES_SALU" "SQ_WAVES SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/${tag}_${name}_U$i -o run -- python tools/kbench.py --channels 1024 --reps 1 > gpurun_out/${tag}_${name}_U$i.log 2>&1
  f=$(find gpurun_out/${tag}_${name}_U$i -name '*counter_collection.csv' | head -1)
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
