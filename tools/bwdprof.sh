# per-kernel durations of the two operator backwards (tools/kbench_bwd.py) -> gpurun_out/bwd_kernels.txt
cd /root/repo
export TMPDIR=/tmp; d=gpurun_out/bwd; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o bwd -- python tools/kbench_bwd.py > $d.log 2>&1
python - <<'PY' | tee gpurun_out/bwd_kernels.txt
import csv,glob
f=glob.glob('gpurun_out/bwd/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-100s calls %5s avg %10.1f us  %5s %%" % (r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
