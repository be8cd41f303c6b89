#!/bin/bash
# Round-end check on the GPU box: GPU tests, smoke, the default bench line.  Usage (from the repo root, through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/final_check.sh r04z'
# Outputs: gpurun_out/<tag>_pytest.log, <tag>_bench.log, <tag>_bench.err.  Steps are chained: a failing step ends the run.
set -eo pipefail
tag="${1:?usage: tools/final_check.sh <tag>}"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "gpurun_out/${tag}_pytest.log" 2>&1 || { tail -30 "gpurun_out/${tag}_pytest.log"; exit 1; }
tail -3 "gpurun_out/${tag}_pytest.log"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python bench.py > "gpurun_out/${tag}_bench.log" 2> "gpurun_out/${tag}_bench.err"
tail -c 2500 "gpurun_out/${tag}_bench.log"
