#!/bin/bash
# usage (GPU box, repo root): tools/prof_bench.sh <tag> [bench args]  -> gpurun_out/<tag>_bench_kernel_stats.txt (+ the raw trace dir prof_<tag>)
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
d=$O/prof_$tag; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o $tag -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > $d.log 2>&1
python - "$d" "$tag" "$*" > $O/${tag}_bench_kernel_stats.txt <<'PY'
import csv, sys, glob, json, re, os, collections
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from omnifusion_amd.build import source_hash
d, tag, extra = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(f"{d}/**/{tag}_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 {extra}  (snapshot {tag})")
print(f"{'kernel':104s}{'calls':>7s}{'total_us':>13s}{'avg_us':>11s}{'pct':>8s}")
for r in rows:
    print(f"{r['Name'][:102]:104s}{int(r['Calls']):7d}{float(r['TotalDurationNs'])/1e3:13.1f}{float(r['AverageNs'])/1e3:11.2f}{float(r['Percentage']):8.2f}")
fam = collections.OrderedDict()
for r in rows:
    n = re.sub(r"^void |\(anonymous namespace\)::", "", r["Name"])
    n = re.split(r"[<(]", n)[0].strip()
    fam[n] = fam.get(n, 0.0) + float(r["Percentage"])
json.dump({"build": source_hash(), "shares": {k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]) if v >= 0.3},
           "top": [{"kernel": re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:90], "pct": float(r["Percentage"]), "avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"])} for r in rows[:8]],
           "note": f"per-kernel-family share (%) of the GPU time of `python bench.py --no-cpu-baseline --steps 10 --warmup 3 {extra}` under rocprofv3 --kernel-trace --stats (the whole run: resample sweeps and the other legs included); snapshot {tag}"},
          open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/bench_kernel_shares.json", "w"), indent=1)
PY
head -40 $O/${tag}_bench_kernel_stats.txt
