#!/bin/bash
# Compile-time ablations of the f16x3 convolution kernels (OMNI_CONV_ABL in csrc/omni_conv_sh.hip): builds one library variant per bit set HERE
# (no GPU needed), then `tools/convabl.sh run` on the GPU box times tools/convbench.py with each.   BITS="0 112 128 ..." ONLY=0,2,4
cd "$(dirname "$0")/.."
BITS=${BITS:-"0 112 128 256 512 640 752 4"}
C=omnifusion_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++20 -munsafe-fp-atomics -fPIC -fno-gpu-rdc -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-function"
if [ "$1" != run ]; then
  python -m omnifusion_amd.build > /dev/null 2>&1
  mkdir -p gpurun_abl
  for b in $BITS; do ( /opt/rocm/bin/hipcc $FL -DOMNI_CONV_ABL=$b -c $C/omni_conv_sh.hip -o gpurun_abl/conv_sh_$b.o 2>/dev/null && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_abl/libabl_$b.so gpurun_abl/conv_sh_$b.o $(ls $C/*.o | grep -v 'dbg\|omni_conv_sh.o\|omni_debug') && rm gpurun_abl/conv_sh_$b.o ) & done
  wait; ls gpurun_abl
else
  for b in $BITS; do echo "== bits=$b"; LIBPATH=gpurun_abl/libabl_$b.so python tools/convbench.py; done
fi
