# same-box A/B of whole forwards between THIS library and a variant build (gpurun_abl/libold_conv.so: the library linked with omni_conv_sh.o of another revision;
# omnifusion_amd/_lib.py loads $OMNI_LIB_VARIANT instead of the product library); first the tests that pin the kernels touched
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "fused_upsample or fused_up_conv_heads or single_pass_model_golden or kernel_choices" 2>&1 | grep -E "passed|failed"
for rep in 1 2; do for v in new old; do
  if [ $v = old ]; then export OMNI_LIB_VARIANT=$PWD/gpurun_abl/libold_conv.so; else unset OMNI_LIB_VARIANT; fi
  echo "== $v"; OPT=conv_stem_pc VALS=1 BS=8 python tools/plain_ab.py 2>&1 | grep plain; OPT=conv_stem_pc VALS=1 python tools/pipe_ab.py 2>&1 | grep median
done; done
