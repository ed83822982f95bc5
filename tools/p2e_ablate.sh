#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the ablated pers2equi LDS kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in ${DBGS:-0 1 2 4 8 15}; do
  rm -rf /tmp/abl; OMNI_P2E_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl -o abl -- python $R/tools/p2e_ablate.py > /tmp/abl.log 2>&1
  grep "OMNI_P2E_DBG" /tmp/abl.log
  python - <<'PY'
import csv, glob
f = glob.glob("/tmp/abl/**/abl_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "p2e" in r["Name"] and "tiles" not in r["Name"] and "candidates" not in r["Name"]:
        print("    ", r["Name"][:60], "calls", r["Calls"], "avg_us %.2f" % (float(r["AverageNs"]) / 1e3), "min %.2f" % (float(r["MinNs"]) / 1e3))
PY
done
