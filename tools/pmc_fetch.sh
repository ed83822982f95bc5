#!/bin/bash
# usage (GPU box, repo root): tools/pmc_fetch.sh "<ENV=..> <ENV=..>" ...   — FETCH_SIZE (doubled: MI355X_MICROARCH.md) of the two resample
# kernels at the benchmark shape under one option set per argument (environment assignments, may be empty)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_fetch; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/v$i -o v$i -- python $R/tools/kbench.py --B ${B:-8} --iters 5 > $O/v$i.log 2>&1
  python - "$O/v$i" "$v" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list); dur = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
line = f"[{sys.argv[2]}]"
for sub in ("e2p_box_kernel", "p2e_lds_kernel"):
    for k in agg:
        if sub in k and len(agg[k]) >= 5:
            line += f"  {sub}: fetch {2 * 1024 * sum(agg[k]) / len(agg[k]) / 1e6:.1f} MB, {sum(dur[k]) / len(dur[k]):.1f} us under the counters"
print(line)
PY
done
