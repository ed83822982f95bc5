set -x
timeout 900 python -m pytest tests/test_resample_gpu.py -x -q 2>&1 | tail -5
for rep in 1 2; do for v in 1 0; do
  export OMNI_P2E_TILE8=$v; echo "== tile8=$v"
  python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 --half 2>&1 | grep pers2equi | sed 's/.*| pers2equi/cfg5 f16 pers2equi/'
  python tools/kbench.py --B 1 --P 512 --H 2048 --W 4096 --nrows 6 2>&1 | grep pers2equi | sed 's/.*| pers2equi/cfg5 f32 pers2equi/'
  python tools/kbench.py --B 1 --P 256 --H 1024 --W 2048 --nrows 6 2>&1 | grep pers2equi | sed 's/.*| pers2equi/cfg3 pers2equi/'
  python tools/kbench.py --B 1 --P 256 2>&1 | grep pers2equi | sed 's/.*| pers2equi/cfg1 B=1 pers2equi/'
done; done
