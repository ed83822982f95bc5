#!/bin/bash
# usage (on the GPU box, from the repo root): tools/make_profiles.sh <tag>
# Writes gpurun_out/<tag>_bench.json, <tag>_bench_kernel_stats.txt, <tag>_layers.txt, <tag>_conv_pmc.txt, <tag>_resample_pmc.txt
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py > $O/${tag}_bench.log 2>&1; tail -1 $O/${tag}_bench.log > $O/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
d=$O/prof_$tag; rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o $tag -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $d.log 2>&1
python - "$d" "$tag" > $O/${tag}_bench_kernel_stats.txt <<'PY'
import csv, sys, glob
d, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{d}/**/{tag}_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 10 --warmup 3   (snapshot {tag})")
print(f"{'kernel':104s}{'calls':>7s}{'total_us':>13s}{'avg_us':>11s}{'pct':>8s}")
for r in rows:
    print(f"{r['Name'][:102]:104s}{int(r['Calls']):7d}{float(r['TotalDurationNs'])/1e3:13.1f}{float(r['AverageNs'])/1e3:11.2f}{float(r['Percentage']):8.2f}")
PY
python $R/tools/layerprof.py $(find $d -name "${tag}_kernel_trace.csv" | head -1) v > $O/${tag}_layers.txt
cd $R
tools/pmc3.sh ${tag}c conv tools/convbench.py > $O/${tag}_conv_pmc.txt 2>&1
tools/pmc.sh ${tag}r --iters 5 > $O/${tag}_resample_pmc.txt 2>&1
ls -la $O | grep $tag
