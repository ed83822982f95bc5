#!/bin/bash
# usage (GPU box, repo root; needs the debug build): tools/fetch_calib.sh <tag>  ->  gpurun_out/<tag>_fetch_calibration.txt
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; out=$O/calib_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p1 -o p1 -- python $R/tools/fetch_calib.py > $out/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/p2 -o p2 -- python $R/tools/fetch_calib.py > $out/p2.log 2>&1
cd $R
python - "$out" > $O/${tag}_fetch_calibration.txt <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
BYTES = 96 << 20
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "calib" in r["Kernel_Name"]]
    # launches in program order: 8 patterns x 3 repetitions
    byc = collections.defaultdict(list)
    for r in rows: byc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    for c, lst in byc.items():
        lst.sort()
        for k, (_, name, v) in enumerate(lst): vals[k % 8][c].append(v)
names = ["LDS-DMA, 16 B per lane (buffer_load_dwordx4 ... lds)", "16-byte loads per lane", "8-byte loads per lane", "4-byte loads per lane, contiguous",
         "4-byte gathers, one per 32 B", "4-byte gathers, one per 64 B", "4-byte gathers, one per 128 B", "4-byte gathers, one per 256 B"]
expect = [BYTES, BYTES, BYTES, BYTES, BYTES, BYTES, BYTES, BYTES]      # bytes of the address range touched (every 32-byte sector / 64-byte / 128-byte line of it for the first six)
print("# FETCH_SIZE calibration on MI355X (gfx950): each kernel reads a 96-MB range ONCE (omni_debug_calib; three buffers in rotation).  Columns: FETCH_SIZE as")
print("# reported (KB), x 1024 x 2 (the correction every traffic figure under profiles/ applies) against the bytes of the range; TCC_EA0_RDREQ (all) / _32B requests.")
print("# A gather of one float per 128 B / 256 B touches every line / every other line: the range's bytes / half of them are the least a 128-byte-line cache can fetch.")
for k, n in enumerate(names):
    fs = vals[k].get("FETCH_SIZE", [0]); rq = vals[k].get("TCC_EA0_RDREQ_sum", [0]); r32 = vals[k].get("TCC_EA0_RDREQ_32B_sum", [0])
    f = sum(fs) / len(fs)
    print("%-52s FETCH_SIZE %10.0f KB  -> x2: %7.1f MB  (%.3f of the 100.7-MB range)   RDREQ %9.0f, 32-byte %9.0f" % (n, f, 2 * f * 1024 / 1e6, 2 * f * 1024 / BYTES, sum(rq) / len(rq), sum(r32) / len(r32)))
PY
cat $O/${tag}_fetch_calibration.txt
