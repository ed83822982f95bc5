#!/bin/bash
# usage (GPU box, repo root): tools/final_profiles.sh <tag> — the subset of make_profiles.sh that must be taken on the FINAL sources: the bench line, its
# kernel stats, the labelled layer sequences, and the two HBM-traffic JSONs whose build hash bench.py compares with the library's.
tag=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
# the counters first: bench.py reports them (roofline.traffic) only from files whose build hash is the library's
tools/pmc_traffic.sh $tag 8 > /dev/null 2>&1
tools/pmc_net.sh $tag 8 > /dev/null 2>&1
tools/pmc_conv.sh $tag > /dev/null 2>&1                      # mfma_busy per kernel form -> conv_pmc.json (roofline.dominant.mfma_busy_from_profile)
cp $O/${tag}_resample_traffic.json $R/profiles/resample_traffic.json; cp $O/${tag}_network_traffic.json $R/profiles/network_traffic.json; cp $O/conv_pmc.json $R/profiles/conv_pmc.json
tools/prof_bench.sh $tag > /dev/null 2>&1                    # per-kernel stats of the bench command -> bench_kernel_shares.json (roofline.kernel_time_shares)
cp $O/bench_kernel_shares.json $R/profiles/bench_kernel_shares.json
python bench.py > $O/${tag}_bench.log 2>&1; tail -1 $O/${tag}_bench.log > $O/${tag}_bench.json
python tools/clocks_under_load.py 2>&1 | grep -v amdgpu.ids > $O/${tag}_clocks_under_load.txt
cd /tmp && export TMPDIR=/tmp
for b in 8 1; do
  d=$O/prof_${tag}_b$b; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $d -o l -- python $R/tools/fwd.py --batch $b --steps 6 --labels $d/labels.txt > $d.log 2>&1
  { echo "# kernel sequence of ONE forward of the single-pass model at $b panorama(s) per GPU (512x1024, 18 x 128^2 patches), ONE lane (the whole batch per kernel, as in the pipelined steady state), each kernel alone on the GPU:";
    echo "# rocprofv3 --kernel-trace -- python tools/fwd.py --batch $b --steps 6 --labels ..., last forward; columns: library call (layer), kernel, grid (threads), duration us";
    python $R/tools/layerprof.py $(find $d -name "l_kernel_trace.csv" | head -1) v $d/labels.txt; } > $O/${tag}_layers_b$b.txt
done
cd $R
# the raw traces (50 MB each) stay on the box: gpurun copies back at most 64 MiB
rm -rf $O/prof_${tag} $O/prof_${tag}_b8 $O/prof_${tag}_b1
ls -la $O | grep $tag
