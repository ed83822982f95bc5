#!/bin/bash
# usage (on the GPU box, from the repo root): tools/make_profiles.sh <tag>
# Writes into gpurun_out/: <tag>_bench.json, <tag>_bench_kernel_stats.txt (rocprofv3 --kernel-trace --stats of the same bench command),
# <tag>_layers_b8.txt / <tag>_layers_b1.txt (kernel sequence of ONE forward at 8 / 1 panoramas), <tag>_resample_shapes.txt (per-shape
# durations of the resample pair), <tag>_resample_pmc.txt (counters), <tag>_resample_traffic.json (HBM bytes + build hash), <tag>_bwd.txt / <tag>_{planar,reference}_bwd_pmc.txt (operator backwards).
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py > $O/${tag}_bench.log 2>&1; tail -1 $O/${tag}_bench.log > $O/${tag}_bench.json
tools/prof_bench.sh $tag > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for b in 8 1; do
  d=$O/prof_${tag}_b$b; rm -rf $d; mkdir -p $d
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $d -o l -- python $R/tools/fwd.py --batch $b --steps 6 --labels $d/labels.txt > $d.log 2>&1
  { echo "# kernel sequence of ONE forward of the single-pass model at $b panorama(s) per GPU (512x1024, 18 x 128^2 patches), ONE lane (the whole batch per kernel, as in the pipelined steady state), each kernel alone on the GPU:";
    echo "# rocprofv3 --kernel-trace -- python tools/fwd.py --batch $b --steps 6 --labels ..., last forward; columns: library call (layer), kernel, grid (threads), duration us";
    python $R/tools/layerprof.py $(find $d -name "l_kernel_trace.csv" | head -1) v $d/labels.txt; } > $O/${tag}_layers_b$b.txt
done
cd $R
{ echo "# resample pair, per shape: tools/kbench.py (HIP events on the launch stream, 20 launches each); algorithmic bytes = B*C*(H*W + P*P*N)*s per operator (SURVEY 8d)";
  python tools/kbench.py --B 8 --P 256 --copy
  python tools/kbench.py --B 8 --P 128
  python tools/kbench.py --B 1 --P 256
  python tools/kbench.py --B 16 --P 256
  python tools/kbench.py --B 1 --P 256 --H 1024 --W 2048 --nrows 6
  python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6
  python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 --half
  python tools/kbench.py --B 4 --P 512 --H 2048 --W 4096 --nrows 6 --half; } 2>&1 | grep -v amdgpu.ids > $O/${tag}_resample_shapes.txt
tools/pmc_resample.sh ${tag}r --iters 5 > $O/${tag}_resample_pmc.txt 2>&1
tools/pmc_traffic.sh $tag 8 > /dev/null 2>&1
tools/pmc_net.sh $tag 8 > /dev/null 2>&1
tools/pmc_conv.sh $tag > /dev/null 2>&1
tools/pmc_single.sh $tag > /dev/null 2>&1
{ echo "# equi2pers / pers2equi in the reference layout [B,C,ph,pw,N] (what the drop-in functions return / take): tools/kbench.py --layout ref; ref_lds=0: the gather kernel";
  for v in 1 0; do for a in "--B 8" "--B 8 --P 128" "--B 1" "--B 16"; do OMNI_E2P_REF_LDS=$v python tools/kbench.py --layout ref $a 2>&1 | grep -v amdgpu.ids | sed "s/^/ref_lds=$v /"; done; done; } > $O/${tag}_ref_layout.txt
LAYOUT=planar tools/pmc_bwd.sh ${tag}_planar > /dev/null 2>&1
LAYOUT=reference tools/pmc_bwd.sh ${tag}_reference > /dev/null 2>&1
{ echo "# tools/kbench_bwd.py (B = 8, 512x1024, 18 x 256^2, fp32; equi2pers^T 24 planes, pers2equi^T 8 planes), HIP-event means of 20 calls"
  echo "## default (sparse-matrix gathers through the plane-interleaved copy)"; python tools/kbench_bwd.py 2>&1 | grep bwd
  echo "## OMNI_BWD_WIDE=0 (4-byte gathers from the gradient itself, no scratch)"; OMNI_BWD_WIDE=0 python tools/kbench_bwd.py 2>&1 | grep bwd
  echo "## the kernels of rounds 1-3: OMNI_P2E_BWD_SIMPLE=2 OMNI_E2P_BWD_SIMPLE=3 (tile gathers with LDS atomics)"; OMNI_P2E_BWD_SIMPLE=2 OMNI_E2P_BWD_SIMPLE=3 python tools/kbench_bwd.py 2>&1 | grep bwd
  echo "## OMNI_P2E_BWD_SIMPLE=1 OMNI_E2P_BWD_SIMPLE=2 (global atomics; LDS boxes + global atomics)"; OMNI_P2E_BWD_SIMPLE=1 OMNI_E2P_BWD_SIMPLE=2 python tools/kbench_bwd.py 2>&1 | grep bwd
  echo "## tables"; OMNI_E2P_VERBOSE=1 LAYOUT=planar python tools/kbench_bwd.py 2>&1 | grep "sparse"; } > $O/${tag}_bwd.txt
ls -la $O | grep $tag
