#!/bin/bash
# usage: tools/pmc3.sh <tag> <kernel substring> <script> [args]   — several PMC passes on one script
tag=$1; shift; kern=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python $GRAFT_REPO_ROOT/"$@" > $out/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$out/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/p*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in agg.items():
    if "$kern" in k:
        print(k, "avg_us=%.1f n=%d" % (sum(dur[k])/len(dur[k]), len(dur[k])))
        for c, v in sorted(d.items()):
            print(f"   {c:40s} mean={sum(v)/len(v):.4g}")
PY
