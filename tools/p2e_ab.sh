# same-box A/B of the one-plane pers2equi forms: OMNI_P2E_WALK1 = 0 (p2e_walk_kernel<T,1>: one stage) | 2 | 3 patches in flight
for rep in 1 2; do for v in 0 2 3; do
  export OMNI_P2E_WALK1=$v
  echo "== p2e_walk1=$v"; python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 --half 2>&1 | grep -v amdgpu | sed 's/.*| pers2equi/cfg5 f16 pers2equi/'; python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 2>&1 | grep -v amdgpu| sed 's/.*| pers2equi/cfg5 f32 pers2equi/'; python tools/kbench.py --B 1 --P 256 --H 1024 --W 2048 --nrows 6 2>&1 | grep -v amdgpu| sed 's/.*| pers2equi/cfg3 pers2equi/'; python tools/kbench.py --B 1 --P 256 2>&1 | grep -v amdgpu| sed 's/.*| pers2equi/cfg1 B=1 pers2equi/'
done; done
