#!/bin/bash
# pers2equi (walk kernel, B = 8, 18 x 256^2) with parts switched off, cache-warm against rotating buffers: which part pays for HBM instead of the
# memory-side cache.  DEBUG build (python -m omnifusion_amd.build --debug); OMNI_P2E_DBG bits: 1 no tap geometry, 2 no LDS tap reads, 4 no DMA, 8 no stores
export OMNI_LIB_VARIANT=$PWD/omnifusion_amd/csrc/libomnifusion_hip_dbg.so
for bits in 0 4 8 12 15; do
  export OMNI_P2E_DBG=$bits
  echo "== OMNI_P2E_DBG=$bits"; python tools/p2e_rot.py --walk 2>&1 | grep round
done
