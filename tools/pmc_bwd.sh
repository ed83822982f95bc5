#!/bin/bash
# usage (GPU box, repo root): tools/pmc_bwd.sh <tag> -> gpurun_out/<tag>_bwd_pmc.txt
# The two operator backwards (tools/kbench_bwd.py: B = 8, 512x1024, 18 x 256^2, planar and reference layout) under rocprofv3 --kernel-trace --pmc,
# separate passes; means per launch of every kernel that ran >= 10 times.
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; out=$O/pmcb_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python $R/tools/kbench_bwd.py > $out/p$i.log 2>&1
done
python - "$out" > $O/${tag}_bwd_pmc.txt <<'PY'
import csv, glob, collections, sys, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"], str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# tools/pmc_bwd.sh: operator backwards at B = 8, 512x1024, 18 x 256^2 (equi2pers C = 3, pers2equi C = 1); means per launch; fetch = FETCH_SIZE x 2 KiB (gfx950), write = WRITE_SIZE KiB")
for k in sorted(agg, key=lambda n: -sum(dur[n])):
    if len(dur[k]) < 10 or not any(t in k[0] for t in ("sp_gather", "sp_interleave", "p2e_nlast", "bwd")): continue
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    nm = re.sub(r"\(anonymous namespace\)::|void ", "", k[0]).split("(")[0]
    print(f"== {nm} grid {k[1]}: {sum(dur[k]) / len(dur[k]):.1f} us under the counters, {len(dur[k])} launches")
    for c in sorted(m): print(f"   {c:40s} {m[c]:14.1f}")
    if "FETCH_SIZE" in m: print(f"   fetch {m['FETCH_SIZE'] * 2 / 1024:.1f} MB, write {m.get('WRITE_SIZE', 0) / 1024:.1f} MB")
PY
cat $O/${tag}_bwd_pmc.txt | head -120
