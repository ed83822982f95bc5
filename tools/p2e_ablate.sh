# the one-plane pers2equi at BASELINE cfg 5 / cfg 3 with parts switched off (DEBUG build, OMNI_P2E_DBG bits: 1 no tap geometry, 2 no LDS tap reads, 4 no DMA,
# 8 no stores, 32 blocks read their header and end) — timing only, the results are wrong by construction.
export OMNI_LIB_VARIANT=$PWD/omnifusion_amd/csrc/libomnifusion_hip_dbg.so
for bits in 0 32 1 2 4 8 7 15; do
  export OMNI_P2E_DBG=$bits
  echo "== OMNI_P2E_DBG=$bits"; python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 --half 2>&1 | grep -v amdgpu | sed 's/.*| pers2equi/cfg5 f16 pers2equi/;s/|.*//'; python tools/kbench.py --B 1 --P 256 --H 1024 --W 2048 --nrows 6 2>&1 | grep -v amdgpu| sed 's/.*| pers2equi/cfg3 pers2equi/;s/|.*//'
done
