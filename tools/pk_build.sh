#!/bin/bash
# Investigation build (never the product): the library with packed fp32 arithmetic ALLOWED in the files named on the command line
# (default omni_pers2equi.hip), every other object as the product has it.  -> tools/pk/libomnifusion_pk.so
# extra hipcc flags for the packed files: PKFLAGS
cd "$(dirname "$0")/.."; C=omnifusion_amd/csrc; O=tools/pk; mkdir -p $O
PK=${@:-omni_pers2equi.hip}
BASE="--offload-arch=gfx950 -O3 -std=c++20 -munsafe-fp-atomics -fPIC -fno-gpu-rdc -ffp-contract=off -Wall -Wno-unused-function -Iinclude"
objs=""
for f in $C/*.hip; do
  b=$(basename $f .hip); [ "$b" = omni_debug ] && continue
  if echo " $PK " | grep -q " $b.hip "; then
    /opt/rocm/bin/hipcc $BASE $PKFLAGS -c $f -o $O/$b.pk.o 2>&1 | grep -v "packed-fp32" ; objs="$objs $O/$b.pk.o"
  else
    objs="$objs $C/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs -lz -o $O/libomnifusion_pk.so && echo built $O/libomnifusion_pk.so
