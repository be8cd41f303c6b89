#!/bin/bash
# Reproduces INTEGRATION.md section A's "checked here": the reference's own Cython binding, unmodified, compiles
# against include/flacarray.h (the stand-in for the reference header of that name) and the compiled object refers to
# exactly the symbols libflacarray_hip.so exports.  Build container only: it reads /root/reference (never copied into
# the repo; the generated C goes to a scratch directory) and skips where that tree is absent (the GPU box).
#   bash tools/check_pyx_binding.sh
set -e
REF=/root/reference/src/flacarray/libflacarray/libflacarray.pyx
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ ! -f "$REF" ]; then echo "SKIP: $REF not present (not the build container)"; exit 0; fi
python3 -c "import Cython" 2>/dev/null || { echo "SKIP: Cython not importable"; exit 0; }
W=$(mktemp -d /tmp/pyxcheck.XXXXXX)
trap 'rm -rf "$W"' EXIT
# the scratch copy is what cython needs next to its output; it never enters the repo
cp "$REF" "$W/libflacarray.pyx"
(cd "$W" && python3 -m cython -3 libflacarray.pyx -o libflacarray.c)
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
gcc -O1 -fPIC -c -I"$ROOT/include" -I"$PYINC" -I"$NPINC" -Wno-deprecated-declarations -o "$W/libflacarray.o" "$W/libflacarray.c"
# every undefined symbol of the binding that belongs to the flacarray C layer must be exported by the library
need=$(nm -u "$W/libflacarray.o" | awk '{print $2}' | grep -E '^(encode_|decode_|float(32|64)_to_|int(32|64)_to_)' | sort -u)
have=$(nm -D --defined-only "$ROOT/flacarray_amd/lib/libflacarray_hip.so" | awk '{print $3}' | sort -u)
missing=0
for s in $need; do
  if ! echo "$have" | grep -qx "$s"; then echo "MISSING in libflacarray_hip.so: $s"; missing=1; fi
done
echo "binding needs: $(echo $need | tr '\n' ' ')"
[ $missing -eq 0 ] && echo "OK: the reference's libflacarray.pyx compiles with -I include and every C symbol it calls is exported"
exit $missing
