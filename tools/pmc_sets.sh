#!/bin/bash
# usage: tools/pmc_sets.sh <tag> <kernel substring> "<set1>;<set2>;..." <script> [args]   (env passes through)
tag=$1; shift; kern=$1; shift; sets=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra SETS <<< "$sets"
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python $GRAFT_REPO_ROOT/"$@" > $out/p$i.log 2>&1
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
