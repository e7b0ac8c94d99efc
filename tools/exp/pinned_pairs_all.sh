export TMPDIR=/tmp; ulimit -c 0
for V in pinflow pinflow_tn pinflow_r; do
  export MI_PHYSICS_LIB=$PWD/build_exp/libmi_physics_$V.so
  echo "== variant $V"
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-220
done
