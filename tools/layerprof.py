#!/usr/bin/env python
"""Print the kernel sequence of the LAST forward in a rocprofv3 kernel trace (csv), with durations.
usage: python tools/layerprof.py gpurun_out/prof_x/**/x_kernel_trace.csv"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "e2p_box_kernel" in r["Kernel_Name"] or "e2p_lds_kernel" in r["Kernel_Name"]]
# last forward = from the last e2p at P=128 (grid differs from P=256) .. next p2e kernel
start = None
for i in reversed(idx):
    start = i
    nxt = [j for j in range(i + 1, len(rows)) if any(t in rows[j]["Kernel_Name"] for t in ("p2e_lds_kernel", "p2e_walk_kernel", "p2e_kernel"))]
    if nxt and nxt[0] - i > 50:
        end = nxt[0]
        break
tot = collections.Counter()
# optional 3rd argument: the call labels of that forward (tools/fwd.py --labels, one lane): kernel k of the sequence is call k, except
# that a split-K reduce (and a one-time weight re-pack) belongs to the call in front of it
labels = [l.strip() for l in open(sys.argv[3])] if len(sys.argv) > 3 else None
li = (labels.index("equi2pers") - 1) if labels and "equi2pers" in labels else -1
for r in rows[start:end + 1]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n] += d
    lab = ""
    if labels is not None:
        if not ("splitk_reduce" in n or "gemm_rows_pack" in n):
            li += 1
        lab = (labels[li] if 0 <= li < len(labels) else "?") + ("  (split-K reduce)" if "splitk_reduce" in n else "")
    if len(sys.argv) > 2:
        print(f"{lab:34s} {n[:58]:58s} grid={r['Grid_Size_X']:>9s} {d:8.1f}")
print("---- totals (us)")
for n, d in tot.most_common():
    print(f"{n:72s} {d:8.1f}")
print("sum", sum(tot.values()))
