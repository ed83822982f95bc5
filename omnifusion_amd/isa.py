"""Build-time guard over the gfx950 ISA of libomnifusion_hip.so.

    python -m omnifusion_amd.isa            # table of every kernel: VGPR / AGPR / SGPR / LDS / scratch
    python -m omnifusion_amd.isa --check    # the rules below; exit status 1 on a violation

Two facts about this library are properties of the generated code, not of the source, and were
found by measurement on MI355X:

 1. no packed-fp32 arithmetic (`v_pk_mul_f32`, `v_pk_add_f32`, `v_pk_fma_f32`) may be issued:
    `pers2equi`'s bilinear weights came out wrong in a 16-lane group while another stream's
    convolution issued dense MFMAs (`v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` returns a wrong low product in lanes 48-63 then; profiles/r04g_pkfp32_rootcause.txt).  The switch that
    prevents it is a compiler flag in build.py — one object built without it, or a toolchain
    that ignores it, would re-open the hazard silently.  So the DISASSEMBLY is checked.
 2. the kernels that count their own `s_waitcnt vmcnt(N)` by hand (`e2p_box_kernel`,
    `p2e_lds_kernel`, `p2e_walk_kernel`) must not spill: a scratch store/load in the middle of
    the LDS-DMA pipeline shifts every hand-counted wait.  So `.private_segment_fixed_size` and
    the spill counts of those kernels are checked to be 0.

`build()` runs `check()` after linking and fails the build on a violation.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("OMNI_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FORBIDDEN = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")
# kernels whose correctness depends on hand-counted vmcnt: no scratch, no spills
COUNTED = ("e2p_box_kernel", "p2e_lds_kernel", "p2e_walk_kernel", "e2p_ref_kernel")
# kernels that hold >= 1 % of the bench's GPU time (profiles/r04e_bench_kernel_stats.txt): a spill there is a performance bug —
# scratch 0 is required of every instantiation (conv_sh_kernel<128,128,4,2,3,4> carried 236 B of it)
HOT = ("conv_sh_kernel", "conv3x3_halo_sh_kernel", "conv3x3_up2_g1_kernel", "stem_f16x3_pc_kernel", "heads_kernel", "maxpool_kernel",
       "layernorm512_kernel", "attention_kernel", "upsample_sh8_kernel")
# the translation units WITHOUT device code (host-only): every other object must yield a code object, or the check fails
HOST_ONLY = ("omni_png.o", "omni_geometry.o")


def _device_object(obj, tmp):
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "dev.co")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj, os.path.join(tmp, "copy.o")],
                       stderr=subprocess.DEVNULL)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None                                                # a host-only translation unit (no kernels)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                           f"--targets={TARGET}", f"--output={co}"])
    return co


def disassemble(obj):
    """gfx950 disassembly (text) of the device code embedded in one host object file."""
    with tempfile.TemporaryDirectory() as tmp:
        co = _device_object(obj, tmp)
        if co is None:
            return ""
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co]).decode()


def kernel_meta(obj):
    """[{name, vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill}] from the code object's AMDGPU metadata note."""
    with tempfile.TemporaryDirectory() as tmp:
        co = _device_object(obj, tmp)
        if co is None:
            return []
        txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    out, cur = [], None
    keys = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".group_segment_fixed_size": "lds",
            ".private_segment_fixed_size": "scratch", ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill",
            ".name": "name"}
    for line in txt.splitlines():
        m = re.match(r"^  - \.(\w+):\s*(\S+)\s*$", line)           # first key of a kernel record
        if m:
            cur = {}
            out.append(cur)
            line = "    ." + m.group(1) + ": " + m.group(2)
        m = re.match(r"^    (\.\w+):\s*(\S+)\s*$", line)
        if m and cur is not None and m.group(1) in keys:
            v = m.group(2)
            cur[keys[m.group(1)]] = v if m.group(1) == ".name" else int(v)
    return [k for k in out if "name" in k]


def demangle(names):
    try:
        p = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")], input="\n".join(names).encode(), stdout=subprocess.PIPE, check=True)
        return p.stdout.decode().splitlines()
    except Exception:
        return list(names)


def objects():
    from . import build as b
    return [s[:-4] + ".o" for s in b.sources()]


def check(objs=None):
    """Returns the list of violations (empty = fine)."""
    bad = []
    found = {c: 0 for c in COUNTED + HOT}
    full = objs is None                                              # the whole library: every guarded kernel must be SEEN
    for obj in objs or objects():
        base = os.path.basename(obj)
        if not os.path.exists(obj):
            bad.append(f"{base}: object missing (build first)")
            continue
        asm = disassemble(obj)
        if not asm.strip():
            # fail CLOSED: an object that yields no device code is either a known host-only unit or a sign that the extraction broke
            # (objcopy / bundler / target string) — in which case nothing below would have been checked at all
            if base not in HOST_ONLY:
                bad.append(f"{base}: no gfx950 device code could be extracted (not in HOST_ONLY): the ISA rules were NOT checked")
            continue
        cur = "?"
        for line in asm.splitlines():
            if line.endswith(">:"):
                cur = line.split("<", 1)[1][:-2]
            elif FORBIDDEN.search(line):
                bad.append(f"{base}: packed-fp32 instruction in {cur}: {line.strip()[:80]}")
                break
        for k in kernel_meta(obj):
            for c in COUNTED + HOT:
                if c in k["name"]:
                    found[c] += 1
            if any(c in k["name"] for c in COUNTED):
                if k.get("scratch", 0) or k.get("vgpr_spill", 0):          # (SGPR spills go to VGPR lanes — v_writelane — not to memory)
                    bad.append(f"{base}: {k['name']} uses scratch {k.get('scratch')} B / spills {k.get('vgpr_spill')} VGPR "
                               f"{k.get('sgpr_spill')} SGPR — its hand-counted s_waitcnt vmcnt(N) would be off")
            elif any(c in k["name"] for c in HOT):
                if k.get("scratch", 0) or k.get("vgpr_spill", 0):
                    bad.append(f"{base}: {k['name']} (a kernel with >= 1 % of the bench's GPU time) uses scratch {k.get('scratch')} B / "
                               f"spills {k.get('vgpr_spill')} VGPR")
    if full:
        for c, n in found.items():
            if n == 0:                                              # a rename, or a change of the notes format: the rule would be silently off
                bad.append(f"guarded kernel `{c}` was not found in any code object's metadata: the scratch rule for it was NOT checked")
    return bad


def table(objs=None):
    rows = []
    for obj in objs or objects():
        meta = kernel_meta(obj)
        for k, nm in zip(meta, demangle([k["name"] for k in meta])):
            nm = nm.replace("(anonymous namespace)::", "")
            nm = re.sub(r"\(.*$", "", nm)
            rows.append((os.path.basename(obj), nm, k.get("vgpr", 0), k.get("agpr", 0), k.get("sgpr", 0), k.get("lds", 0), k.get("scratch", 0)))
    return rows


if __name__ == "__main__":
    if "--check" in sys.argv:
        v = check()
        for line in v:
            print("ISA CHECK FAILED:", line)
        print("isa check:", "FAILED" if v else "ok (no v_pk_*_f32; counted-wait and hot kernels without scratch; every guarded kernel found)")
        sys.exit(1 if v else 0)
    print(f"{'object':22s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'lds':>6s} {'scr':>4s}  kernel")
    for o, nm, v, a, s, l, sc in table():
        print(f"{o:22s} {v:4d} {a:4d} {s:4d} {l:6d} {sc:4d}  {nm[:150]}")
