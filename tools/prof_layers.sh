R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for b in 8; do d=$O/prof_x_b$b; rm -rf $d; mkdir -p $d
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $d -o l -- python $R/tools/fwd.py --batch $b --steps 6 --labels $d/labels.txt > $d.log 2>&1
python $R/tools/layerprof.py $(find $d -name "l_kernel_trace.csv" | head -1) v $d/labels.txt > $O/x_layers_b$b.txt; done
grep -A12 "de_conv3_1" $O/x_layers_b8.txt | head -8; tail -28 $O/x_layers_b8.txt
