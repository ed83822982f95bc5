"""Build libomnifusion_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.

    python -m omnifusion_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the
GPU box with the gpurun snapshot.

-ffp-contract=off: every kernel variant (layouts, dtypes, the candidate-mask builder vs the
blend) must evaluate the shared geometry functions to the SAME bits; FMAs are written
explicitly (fmaf) where they are wanted.

-target-feature -packed-fp32-ops: no v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 anywhere in the
library.  Measured on MI355X (tools/conc3.py): a kernel whose f32 products hipcc had packed —
pers2equi's bilinear weights — returned wrong values in a 16-lane group of some waves in most
launches WHILE a convolution kernel issued dense MFMAs on another stream of the same GPU (never
alone, never beside copy / element-wise kernels, never with one MFMA per product block); without
packed instructions: 0 of thousands.  The host pass prints "'-packed-fp32-ops' is not a recognized
feature" (ignored there); resample timings are unchanged.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libomnifusion_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-munsafe-fp-atomics", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-Wall", "-Wno-unused-function"]


LIB_DEBUG = os.path.join(CSRC, "libomnifusion_hip_dbg.so")
DEBUG_ONLY = ("omni_debug.hip",)             # micro-benchmark scaffolding: never part of the product library


def sources(debug=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    return srcs if debug else [f for f in srcs if os.path.basename(f) not in DEBUG_ONLY]


def source_hash():
    """sha256 over the library's sources (csrc/*.hip, csrc/*.h, include/*.h): ties a measurement file under
    profiles/ to the build it was taken on (bench.py reports PMC traffic only when the hashes agree)."""
    import hashlib
    h = hashlib.sha256()
    files = sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + \
        sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(lib=LIB, debug=False):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources(debug) + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, debug=False):
    """debug=True builds libomnifusion_hip_dbg.so: -DOMNI_DEBUG_BUILD (result-changing ablation bits, OMNI_*_DBG) plus
    omni_debug.hip — for tools/ only; the product library carries neither."""
    lib = LIB_DEBUG if debug else LIB
    if not force and not _stale(lib, debug):
        return lib
    objs = []
    procs = []
    for src in sources(debug):
        obj = src[:-4] + (".dbg.o" if debug else ".o")
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + (["-DOMNI_DEBUG_BUILD"] if debug else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src}\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    if not debug:
        # the ISA guard (omnifusion_amd/isa.py): no packed-fp32 arithmetic anywhere, no scratch in the kernels that count their own
        # s_waitcnt vmcnt(N).  A violation fails the build BEFORE a library exists that could be loaded.
        from . import isa
        bad = isa.check(objs)
        if bad:
            if os.path.exists(lib):
                os.remove(lib)
            raise RuntimeError("ISA check failed:\n  " + "\n  ".join(bad))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
