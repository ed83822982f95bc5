#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh <tag> [B]
# HBM-side traffic of the resample pair at the benchmark shape (B panoramas of 512x1024, 18 x 256^2 patches, fp32, planar): separate
# --pmc passes (FETCH_SIZE, WRITE_SIZE), collected and corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE doubled on gfx950,
# WRITE_SIZE as reported).  Writes gpurun_out/<tag>_resample_traffic.json carrying the hash of the sources it was measured on:
# copy it to profiles/resample_traffic.json — bench.py reports `traffic` only when the hash is THIS build's.
tag=$1; B=${2:-8}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; out=$O/pmc_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python $R/tools/kbench.py --B $B --iters 5 > $out/p$i.log 2>&1
done
cd $R
python - "$out" "$B" "$O/${tag}_resample_traffic.json" <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, ".")
from omnifusion_amd.build import source_hash
out, B, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{out}/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for key, sub in (("equi2pers", "e2p_box_kernel"), ("pers2equi", "p2e_lds_kernel")):
    ks = [k for k in agg if sub in k]
    if not ks:
        continue
    k = max(ks, key=lambda n: sum(dur[n]))
    m = {c: sum(v) / len(v) for c, v in agg[k].items()}
    res[key] = {"kernel": k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], "avg_us_under_pmc": sum(dur[k]) / len(dur[k]),
                "fetch_bytes": 2 * m["FETCH_SIZE"] * 1024, "write_bytes": m["WRITE_SIZE"] * 1024,
                "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])}
alg = {"equi2pers": B * 3 * (512 * 1024 + 256 * 256 * 18) * 4, "pers2equi": B * (256 * 256 * 18 + 512 * 1024) * 4}
tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in res.values())
note = "; ".join(f"{k} {v['fetch_bytes']/1e6:.1f} MB fetched + {v['write_bytes']/1e6:.1f} MB written vs {alg[k]/1e6:.1f} MB algorithmic (L2 hit rate {v['l2_hit_rate']:.2f})" for k, v in res.items())
json.dump({"build": source_hash(), "B": B, "traffic_bytes": tot, "algorithmic_bytes": sum(alg.values()), "note": note, "kernels": res}, open(dst, "w"), indent=1)
print(open(dst).read())
PY
