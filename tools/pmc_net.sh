#!/bin/bash
# usage (GPU box, repo root): tools/pmc_net.sh <tag> [B]
# HBM-side traffic of ONE forward of the single-pass model at B panoramas (512x1024, 18 x 128^2 patches), run AS bench.py's TIMED REGION runs it
# (net.pipelined(3): whole-batch kernels — VERDICT r4 #7a: rounds 2-4 measured the two-lane plain forward, other kernels): separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE; FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes) over tools/fwd.py, summed over every kernel of the run and
# divided by the number of forwards.  Writes gpurun_out/<tag>_network_traffic.json with the hash of the sources it was measured on: copy it to
# profiles/network_traffic.json — bench.py reports `roofline.traffic` only when the hash is THIS build's.
tag=$1; B=${2:-8}; STEPS=8
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; out=$O/pmcn_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python $R/tools/fwd.py --batch $B --steps $STEPS --pipelined 3 > $out/p$i.log 2>&1
done
cd $R
python - "$out" "$B" "$O/${tag}_network_traffic.json" <<'PY'
import csv, glob, collections, json, sys, re
sys.path.insert(0, ".")
from omnifusion_amd.build import source_hash
out, B, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tot = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); per[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE": calls[n] += 1
stem = [n for n in calls if "stem" in n]
lanes = 1                                   # pipelined: every kernel works on the whole batch
nfwd = max(1, calls[stem[0]] // lanes) if stem else 1
fetch, write = 2 * tot["FETCH_SIZE"] * 1024 / nfwd, tot["WRITE_SIZE"] * 1024 / nfwd
top = sorted(per.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:12]
json.dump({"build": source_hash(), "B": B, "forwards": nfwd, "traffic_bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
           "mode": "spherical_fusion.pipelined(3): the kernels of bench.py's timed region", "note": f"one forward of {B} panoramas (whole-batch kernels, as the timed region runs them): {fetch/1e6:.0f} MB fetched + {write/1e6:.0f} MB written by all its kernels (FETCH_SIZE x 2 on gfx950, WRITE_SIZE), mean over {nfwd} forwards",
           "kernels_MB_per_forward": {n: {"fetch": round(2 * v["FETCH_SIZE"] * 1024 / nfwd / 1e6, 1), "write": round(v["WRITE_SIZE"] * 1024 / nfwd / 1e6, 1), "calls": calls[n] // nfwd} for n, v in top}},
          open(dst, "w"), indent=1)
print(open(dst).read())
PY
