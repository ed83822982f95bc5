#!/bin/bash
# usage: tools/pmc2.sh <tag> "<counter list>" <python args...>
tag=$1; shift; ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py "$@" > $out/p.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "e2p" in k or "p2e_kernel" in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c:40s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
