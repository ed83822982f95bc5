#!/bin/bash
tag=$1; B=${2:-1}; out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 --batch $B > $out.log 2>&1
f=$(find $out -name "${tag}_kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/layerprof.py $f v > $out/layers.txt
tail -24 $out/layers.txt
