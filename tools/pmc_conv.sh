#!/bin/bash
# usage (GPU box, repo root): tools/pmc_conv.sh <tag>  -> gpurun_out/<tag>_conv_pmc.txt
# MFMA utilisation of the network's convolution kernels: rocprofv3 --kernel-trace --pmc (two separate passes, no other trace domain) on
# tools/convbench.py (every convolution shape of the network at 8 panoramas = 144 patches, f16x3 SH path, the decoder's first convolutions
# with the up-sampling inside).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of the SIMD-cycles
# of the launch in which a matrix instruction was executing (32 cycles per v_mfma_f32_32x32x16_f16).
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/pmcc_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/convbench.py > $out/p$i.log 2>&1
done
python - > $GRAFT_REPO_ROOT/gpurun_out/${tag}_conv_pmc.txt <<PY
import csv, glob, collections, json, re, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from omnifusion_amd.build import source_hash
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$out/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:78]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/p*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:78]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# snapshot $tag — rocprofv3 --kernel-trace --pmc <set> (2 separate passes, tools/pmc_conv.sh) on tools/convbench.py: every convolution shape of the")
print("# network at 8 panoramas (144 patches), f16x3 SH path; means per dispatch over the shapes a kernel form serves.")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): share of the launch's SIMD-cycles with a matrix instruction executing")
busy = {}
for k, d in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]])):
    if "conv" not in k: continue
    m = lambda c: sum(d[c]) / len(d[c]) if c in d else float("nan")
    mt = re.search(r"(\w+<[^>]*>)", k)
    if mt:
        key = mt.group(1).replace(" ", "")
        if key.startswith("conv_sh_kernel<"):                   # <BM,BN,WM,WN,NST,NL,PP,WINO>: the two flags by name, defaults dropped
            ar = key[len("conv_sh_kernel<"):-1].split(",")
            key = "conv_sh_kernel<" + ",".join(ar[:6] + (["PP"] if len(ar) > 6 and ar[6] == "true" else []) + (["WINO"] if len(ar) > 7 and ar[7] == "true" else [])) + ">"
        busy[key] = m("SQ_VALU_MFMA_BUSY_CYCLES") / (m("GRBM_GUI_ACTIVE") / 8 * 1024)
    print("%s avg_us=%.1f n=%d   mfma_busy=%.1f%%" % (k, sum(dur[k]) / len(dur[k]), len(dur[k]), 100 * m("SQ_VALU_MFMA_BUSY_CYCLES") / (m("GRBM_GUI_ACTIVE") / 8 * 1024)))
    for c, v in sorted(d.items()):
        print(f"   {c:40s} mean={sum(v)/len(v):.4g}")
json.dump({"build": source_hash(), "mfma_busy": busy, "note": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024) per kernel form, each alone on the GPU (tools/pmc_conv.sh on tools/convbench.py, 144 patches); snapshot $tag"},
          open("$GRAFT_REPO_ROOT/gpurun_out/conv_pmc.json", "w"), indent=1)
PY
head -30 $GRAFT_REPO_ROOT/gpurun_out/${tag}_conv_pmc.txt
