#!/bin/bash
# usage: tools/convprof.sh <tag> [ENV=VAL ...]   — kernel durations (rocprofv3 kernel trace) of tools/convbench.py
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/cp_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $out -o $tag -- python $GRAFT_REPO_ROOT/tools/convbench.py > $out.log 2>&1
f=$(find $out -name "${tag}_kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# aggregate by (kernel, grid): mean of the last 20 dispatches of each
agg = collections.OrderedDict()
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
    if "conv" not in n and "reduce" not in n: continue
    key = (n, r["Grid_Size_X"], r["Grid_Size_Y"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, ds in agg.items():
    if len(ds) >= 10:
        tail = ds[-20:]
        print(f"{key[0]:62s} grid={key[1]:>9s}x{key[2]:<3s} n={len(ds):3d} mean={sum(tail)/len(tail):7.1f} us  min={min(tail):7.1f}")
PY
