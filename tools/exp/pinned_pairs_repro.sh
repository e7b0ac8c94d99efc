#!/bin/bash
# Reproducer of the round-5 "pinned register pairs give WRONG results in the dispatch-ordered kernels" case (EXPERIMENTS.md).
#   python tools/build_variant.py pinflow -DMI_EXP_PIN_FLOW -- pinflow_nv -DMI_EXP_PIN_FLOW -DMI_EXP_PIN_NONVOLATILE \
#          -- pinflow_r -DMI_EXP_PIN_FLOW -DMI_EXP_PIN_MASK=0x7 -- pinflow_tn -DMI_EXP_PIN_FLOW -DMI_EXP_PIN_MASK=0x1F8
# then on the GPU box: bash tools/exp/pinned_pairs_repro.sh   (the flow-kernel cases of the solver-variant test against the oracle, per variant)
export TMPDIR=/tmp; ulimit -c 0
for V in "" pinflow pinflow_nv pinflow_r pinflow_tn; do
  if [ -n "$V" ]; then export MI_PHYSICS_LIB=$PWD/build_exp/libmi_physics_$V.so; else unset MI_PHYSICS_LIB; fi
  echo "== variant '${V:-in-tree}'"
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "other_contact_solvers and (flow or granules)" 2>&1 | tail -3 | cut -c1-200
  timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "ragdoll or joint" 2>&1 | tail -1 | cut -c1-200
done
