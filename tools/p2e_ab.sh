#!/bin/bash
# Same-box A/B of pers2equi between THIS library and a variant build of it: tools/p2e_ab.sh <path to the variant .so>
# (a variant = the library linked with another omni_pers2equi.o, e.g. built from an earlier revision of csrc/omni_pers2equi.hip the way tools/convabl.sh
#  builds its variants; omnifusion_amd/_lib.py loads $OMNI_LIB_VARIANT instead of the product library).  Used for profiles/r05c_p2e_single_plane.txt (a).
V=${1:?usage: tools/p2e_ab.sh <variant.so>}
for rep in 1 2; do for v in new old; do
  if [ $v = old ]; then export OMNI_LIB_VARIANT=$V; else unset OMNI_LIB_VARIANT; fi
  echo "== $v"
  python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 --half 2>&1 | grep -v amdgpu | sed 's/.*| pers2equi/cfg5 f16 pers2equi/'
  python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 2>&1 | grep -v amdgpu | sed 's/.*| pers2equi/cfg5 f32 pers2equi/'
  python tools/kbench.py --B 1 --P 256 --H 1024 --W 2048 --nrows 6 2>&1 | grep -v amdgpu | sed 's/.*| pers2equi/cfg3 pers2equi/'
  python tools/kbench.py --B 8 --P 256 2>&1 | grep -v amdgpu | sed 's/.*| pers2equi/cfg1 B=8 pers2equi/'
done; done
