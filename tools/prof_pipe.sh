#!/bin/bash
# usage (GPU box, repo root): tools/prof_pipe.sh <tag>  -> gpurun_out/<tag>_pipe_kernel_stats.txt: kernel statistics of the STEADY pipelined state
# (8 panoramas per forward, 3 forwards in flight, nothing else in the process — bench.py's profile mixes in its single-panorama and host-fed legs)
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
d=$O/profp_$tag; rm -rf $d; mkdir -p $d
cat > /tmp/pipe_only.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from omnifusion_amd.model.spherical_model import spherical_fusion
from omnifusion_amd.weights import make_state_dict
net = spherical_fusion(4, 18, (128, 128), (80, 80)).cuda(); net.load_state_dict(make_state_dict(42, 18, False))
batches = [torch.rand((8, 3, 512, 1024), device="cuda") for _ in range(4)]
run = net.pipelined(3); pend = []
for i in range(80):
    pend.append(run(batches[i % 4], confidence=True))
    if len(pend) > 3: pend.pop(0).get()
for p in pend: p.get()
torch.cuda.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o $tag -- python /tmp/pipe_only.py > $d.log 2>&1
python - "$d" "$tag" > $O/${tag}_pipe_kernel_stats.txt <<'PY'
import csv, sys, glob
d, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{d}/**/{tag}_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f"# rocprofv3 --kernel-trace --stats: 80 forwards of 8 panoramas, 3 in flight (spherical_fusion.pipelined), nothing else  (snapshot {tag}); calls / 80 = per forward")
print(f"{'kernel':104s}{'calls':>7s}{'total_us':>13s}{'avg_us':>11s}{'pct':>8s}")
for r in rows:
    print(f"{r['Name'][:102]:104s}{int(r['Calls']):7d}{float(r['TotalDurationNs'])/1e3:13.1f}{float(r['AverageNs'])/1e3:11.2f}{float(r['Percentage']):8.2f}")
PY
head -40 $O/${tag}_pipe_kernel_stats.txt
