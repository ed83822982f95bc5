"""Investigation: the assembly variants of fail_base.s that name the failing instruction.  fail_base.s is the gfx950 assembly hipcc 7.2 produced
for an earlier form of taps_dump2.hip (the bug depends on the instruction the compiler happens to choose: the current source no longer yields it).
  B: the packed multiply with crossed src1 halves replaced by two scalar multiplies      -> no wrong value
  D: the same packed multiply WITHOUT op_sel (the halves swapped by two moves first)     -> no wrong value
  E: the crossing moved to src0 (op_sel:[1,0] op_sel_hi:[0,1], operands exchanged)       -> no wrong value
  base: v_pk_mul_f32 v[16:17], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]           -> wrong low product in lanes 48-63 beside an MFMA kernel
usage: python variants.py (writes fail_{B,D,E}.s and the .co code objects of all four); then TAPS_CO=fail_base.co python taps_run.py on the GPU."""
import os, subprocess
here = os.path.dirname(os.path.abspath(__file__))
base = open(os.path.join(here, "fail_base.s")).read()
B = "\tv_pk_mul_f32 v[16:17], v[10:11], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]\n"
assert base.count(B) == 2
reps = {"B": "\tv_mul_f32_e32 v16, v10, v13\n\tv_mul_f32_e32 v17, v11, v12\n",
        "D": "\tv_mov_b32_e32 v30, v13\n\tv_mov_b32_e32 v31, v12\n\tv_pk_mul_f32 v[16:17], v[10:11], v[30:31]\n",
        "E": "\tv_pk_mul_f32 v[16:17], v[12:13], v[10:11] op_sel:[1,0] op_sel_hi:[0,1]\n"}
for name, rep in reps.items():
    s = base.replace(B, rep)
    s = s.replace(".amdhsa_next_free_vgpr 26", ".amdhsa_next_free_vgpr 32").replace(".amdhsa_accum_offset 28", ".amdhsa_accum_offset 32").replace(".vgpr_count:     26", ".vgpr_count:     32")
    open(os.path.join(here, "fail_%s.s" % name), "w").write(s)
llvm = "/opt/rocm/lib/llvm/bin/"
for name in ["base"] + list(reps):
    subprocess.check_call([llvm + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", os.path.join(here, "fail_%s.s" % name), "-o", os.path.join(here, "fail_%s.o" % name)])
    subprocess.check_call([llvm + "ld.lld", "-shared", os.path.join(here, "fail_%s.o" % name), "-o", os.path.join(here, "fail_%s.co" % name)])
    print("built fail_%s.co" % name)
