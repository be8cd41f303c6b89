#!/bin/bash
# Produce the per-round measurement set under gpurun_out/<tag>_* (run on the GPU box through gpurun):
#   bench line, rocprofv3 kernel stats, and the PMC passes (FETCH_SIZE / WRITE_SIZE / instruction mix), each in
#   its own run as MI355X_MICROARCH.md prescribes.  tools/make_traffic.py turns them into profiles/<tag>_*.
#   usage: bash tools/profile_round.sh r01k
set -e
tag=${1:-rXX}
out=gpurun_out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench.log 2> $out/${tag}_bench.err
echo "bench done" && tail -c 300 $out/${tag}_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/${tag}_stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmcF -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/${tag}_pmcF.log 2>&1
echo "pmcF done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmcW -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/${tag}_pmcW.log 2>&1
echo "pmcW done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/${tag}_pmcI -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/${tag}_pmcI.log 2>&1
echo "pmcI done"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $out/${tag}_pmcU -o run -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > $out/${tag}_pmcU.log 2>&1
echo "pmcU done"
# K7's read shape with a KNOWN byte count (tools/ubench/frame_stream: "reads only" kernel): calibrates FETCH_SIZE for it
if [ -x tools/ubench/frame_stream ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmcC -o run -- tools/ubench/frame_stream > $out/${tag}_pmcC.log 2>&1
  echo "pmcC done"
fi

