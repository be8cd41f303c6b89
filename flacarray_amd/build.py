"""Build the HIP extension in-tree: flacarray_amd/lib/libflacarray_hip.so (gfx950).

hipcc cross-compiles without a GPU; the built .so travels with the repo snapshot.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "flacarray_hip.hip")
OUT = os.path.join(_HERE, "lib", "libflacarray_hip.so")
DEPS = [
    SRC,
    os.path.join(_HERE, "csrc", "flac_math.hpp"),
    os.path.join(_HERE, "csrc", "encode_kernels.hpp"),
    os.path.join(_HERE, "csrc", "decode_kernels.hpp"),
    os.path.join(_HERE, "csrc", "quantize_kernels.hpp"),
    os.path.join(os.path.dirname(_HERE), "include", "flacarray_hip.h"),
]
SRC_COMPACT = os.path.join(_HERE, "csrc", "compact_unit.hip")
SRC_FUSED = os.path.join(_HERE, "csrc", "fused_unit.hip")
SRC_PLACED = os.path.join(_HERE, "csrc", "placed_unit.hip")
DEPS += [SRC_COMPACT, SRC_FUSED, SRC_PLACED, os.path.join(_HERE, "csrc", "encode_placed.hpp"), os.path.join(_HERE, "csrc", "encode_fused.hpp"), os.path.join(_HERE, "csrc", "verify_kernels.hpp"),
         os.path.join(_HERE, "csrc", "decode_latency.hpp")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared"]
# The shipped library is built from four translation units (the fourth, the placing encoder K3G, like the first): the slot encoder and the decoder (K3, K7) with LLVM's
# max-ILP scheduling strategy (measured: K3 -4 %, K7 -2 %), the compaction kernels (K5) and the single-pass encoder
# (K3F, which has no registers to spare: 168 for three waves per SIMD) with the default one (max-ILP slows K5 by 10 %
# and adds spills to K3F).  Variants (diagnostic builds) stay single-unit, default strategy.
# K3's occupancy is two waves per SIMD by its LDS image, so the scheduler may as well use the 256 VGPRs (-1.5 %).
MAIN_UNIT_FLAGS = [
    "-DFA_SPLIT_UNITS",
    "-mllvm",
    "-amdgpu-sched-strategy=max-ilp",
    "-DFA_K3_WAVES_ATTR=__attribute__((amdgpu_waves_per_eu(2,2)))",
]


# The placing encoder K3G shares K3's frame body but not its flags: default scheduling strategy, and its own register
# bounds (encode_placed.hpp, FA_PG_ATTR).
PLACED_UNIT_FLAGS = ["-DFA_SPLIT_UNITS"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=(), name=None):
    """Compile the library if it is missing or older than its sources; returns its path.  `name` / `extra`: a variant
    lib/libflacarray_hip_<name>.so of the SHIPPED structure (four units, their flags) with more -D flags -- what an A/B
    against the shipped library needs when the kernels' co-residency depends on the units' flags (K9 beside K7)."""
    if name:
        return _build_units(OUT.replace(".so", f"_{name}.so"), list(extra), verbose, tag=name)
    if not force and not needs_build():
        return OUT
    return _build_units(OUT, [], verbose)


def _build_units(OUT, more, verbose, tag=""):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in FLAGS if f != "-shared"]
    objs, procs = [], []
    # (the units compile side by side)
    for src, extra in ((SRC, MAIN_UNIT_FLAGS), (SRC_COMPACT, []), (SRC_FUSED, []), (SRC_PLACED, PLACED_UNIT_FLAGS)):
        obj = os.path.join(os.path.dirname(OUT), os.path.basename(src).replace(".hip", f"{tag}.o"))
        cmd = [hipcc] + cflags + extra + more + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    for o in objs:
        os.remove(o)
    return OUT


def build_variant(name, defines, verbose=False):
    """Compile a (diagnostic) variant lib/libflacarray_hip_<name>.so with extra -D flags (single unit, default
    scheduling strategy unless the flags say otherwise).  -DFA_DEV_MINIMAL instantiates the level 3-5 int32 kernels
    only: seconds instead of minutes."""
    out = OUT if not name else OUT.replace(".so", f"_{name}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(defines) + ["-o", out + ".tmp", SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    if "--split-variant" in sys.argv:
        i = sys.argv.index("--split-variant")
        print(build(verbose=True, name=sys.argv[i + 1], extra=sys.argv[i + 2:]))
    elif "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    elif "--stamps" in sys.argv:
        print(build_variant("stamps", ["-DFA_STAMPS"], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
