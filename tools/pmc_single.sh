#!/bin/bash
# usage (GPU box, repo root): tools/pmc_single.sh <tag>  -> gpurun_out/<tag>_single_plane_pmc.txt
# The resample kernels of the SINGLE-PANORAMA configurations (VERDICT r3 #1b): BASELINE cfg 3 (1024x2048, nrows 6, 46 x 256^2, fp32) and cfg 5
# (2048x4096, 46 x 512^2, fp16): instruction counts, busy cycles and HBM bytes per launch — separate --pmc passes (kernel-trace only beside them),
# FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; out=$O/pmcs_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {   # name, kbench args...
  name=$1; shift
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$name/p$i -o p$i -- python $R/tools/kbench.py --iters 5 "$@" > $out/$name.p$i.log 2>&1
  done
}
run cfg3 --B 1 --H 1024 --W 2048 --nrows 6 --P 256
run cfg5h --B 1 --H 2048 --W 4096 --nrows 6 --P 512 --half
run cfg5 --B 1 --H 2048 --W 4096 --nrows 6 --P 512
python - "$out" > $O/${tag}_single_plane_pmc.txt <<'PY'
import csv, glob, collections, sys, re
out = sys.argv[1]
print("# tools/pmc_single.sh: the resample kernels of ONE panorama (B = 1) under rocprofv3 --kernel-trace --pmc (4 separate passes per shape); means per launch.")
print("# valu/px-patch = SQ_INSTS_VALU x 64 lanes / (ERP pixels x covering patches: 4.76 at nrows 6) for pers2equi, / patch samples for equi2pers;")
print("# fetch = FETCH_SIZE x 2 KiB (gfx950), write = WRITE_SIZE KiB; us = kernel duration under the counters")
for name, (H, W, N, P, s) in (("cfg3", (1024, 2048, 46, 256, 4)), ("cfg5h", (2048, 4096, 46, 512, 2)), ("cfg5", (2048, 4096, 46, 512, 4))):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{out}/{name}/p*/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"== {name}: {H}x{W} ERP, {N} x {P}^2 patches, {s}-byte elements, B = 1 (equi2pers C = 3, pers2equi C = 1)")
    for k in sorted(agg, key=lambda n: -sum(dur[n])):
        if not any(t in k for t in ("e2p_box_kernel", "p2e_walk_kernel", "p2e_kernel", "p2e_lds_kernel")) or len(dur[k]) < 5: continue
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        nm = re.sub(r"\(anonymous namespace\)::|void ", "", k).split("(")[0]
        units = H * W * 4.76 if "p2e" in k else 3 * N * P * P / 3.0          # (equi2pers: the geometry of a sample is shared by its 3 planes)
        alg = (N * P * P + H * W) * s if "p2e" in k else 3 * (H * W + N * P * P) * s
        us = sum(dur[k]) / len(dur[k])
        print(f"{nm:44s} {us:7.1f} us  waves {m.get('SQ_WAVES', 0):8.0f}  VALU {m.get('SQ_INSTS_VALU', 0):10.0f} ({m.get('SQ_INSTS_VALU', 0) * 64 / units:5.1f} per {'pixel-patch' if 'p2e' in k else 'sample'})"
              f"  SALU {m.get('SQ_INSTS_SALU', 0):9.0f}  LDS {m.get('SQ_INSTS_LDS', 0):8.0f}  VMEM rd/wr {m.get('SQ_INSTS_VMEM_RD', 0):8.0f}/{m.get('SQ_INSTS_VMEM_WR', 0):7.0f}"
              f"  busy {m.get('SQ_BUSY_CYCLES', 0):9.0f}  valu-active {m.get('SQ_ACTIVE_INST_VALU', 0):9.0f}  fetch {2 * m.get('FETCH_SIZE', 0) * 1024 / 1e6:6.1f} MB  write {m.get('WRITE_SIZE', 0) * 1024 / 1e6:6.1f} MB"
              f"  algorithmic {alg / 1e6:6.1f} MB -> {alg / us / 1e6:5.2f} TB/s")
PY
cat $O/${tag}_single_plane_pmc.txt
