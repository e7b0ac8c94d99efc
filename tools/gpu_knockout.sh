#!/bin/bash
# dev helper — the knock-out harness of the persistent solver.  Libraries: build_exp/libmi_physics_knock.so (-DMI_DBG_KNOCKOUT) and
# build_exp/libmi_physics_knocktl.so (+ -DMI_DBG_TIMELINE), both from tools/build_variant.py.  Every step launches k_contact_solve_persist twice: first on scratch
# copies of the velocity arrays with part of the tile visit removed (MI_DBG_KNOCKOUT bits: 1 no row stream, 2 every tile's rows from contact-tile 0 (L2), 4 no tag
# waits, 8 nothing removed = the harness itself), then the real launch.  Prints the mean device time of the knock-out launch beside the real solve stage.
ulimit -c 0
mkdir -p gpurun_out
cat > /tmp/ko.py <<'PY'
import sys
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
settle = int(sys.argv[1]) if len(sys.argv) > 1 else 245
sc = scenes.obb_pile(128, 16, 128)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, settle)
w.set_stage_timing(3)
w.accumulated_stage_times(reset=True)
w.step_fixed(s, sc.dt, 40)
acc, n, ci = w.accumulated_stage_times()
print("real solve stage us %.1f over %d steps, contacts %d colours %d kind %s" % (acc["solve"] * 1e3 / n, n, w.counts()["num_contacts"], w.counts()["num_colors"], w.solver_kind()), flush=True)
del w
PY
OUT=gpurun_out/knockout.txt; : > $OUT
for SETTLE in ${KO_SETTLES:-245 1500}; do
  for K in ${KO_BITS:-8 1 2 4 5}; do
    echo "== settle $SETTLE MI_DBG_KNOCKOUT=$K" >> $OUT
    MI_DBG_KNOCKOUT=$K MI_PHYSICS_LIB=build_exp/libmi_physics_knock.so timeout 300 python /tmp/ko.py $SETTLE 2>&1 | grep -E "knockout|real solve" >> $OUT
  done
done
for K in ${KO_EMIT_BITS:-0x8000 0x100 0x200 0x400 0x1000 0x300 0x700 0x1700}; do
  echo "== settle 245 emit MI_DBG_KNOCKOUT=$K (bits: 0x100 no bodyUsed atomics, 0x200 no history insert, 0x400 no history probe, 0x800 no round-0 proposals, 0x1000 no material gathers; 0x8000 nothing removed)" >> $OUT
  MI_DBG_KNOCKOUT=$K MI_PHYSICS_LIB=build_exp/libmi_physics_knock.so timeout 300 python /tmp/ko.py 245 2>&1 | grep -E "knockout|real solve" >> $OUT
done
for K in ${KO_BP_BITS:-0x800000 0x10000 0x20000 0x40000 0x60000}; do
  echo "== settle 245 bp grid pass MI_DBG_KNOCKOUT=$K (bits: 0x10000 no candidate loop, 0x20000 hits neither keyed nor staged, 0x40000 no block flush; 0x800000 nothing removed)" >> $OUT
  MI_DBG_KNOCKOUT=$K MI_PHYSICS_LIB=build_exp/libmi_physics_knock.so timeout 300 python /tmp/ko.py 245 2>&1 | grep -E "knockout|real solve" >> $OUT
done
cat $OUT
if [ -f build_exp/libmi_physics_knocktl.so ]; then
  cat > /tmp/tlk.py <<'PY'
import sys
sys.path.insert(0, ".")
import torch; torch.cuda.set_device(0)
import d3d12renderer_amd as mi
from d3d12renderer_amd import scenes
sc = scenes.obb_pile(128, 16, 128)
w = sc.populate(mi.create_world(0)); s = sc.settings()
w.step_fixed(s, sc.dt, 285)
PY
  for K in ${KO_TL_BITS:-1}; do
    MI_DBG_KNOCKOUT=$K MI_DBG_TIMELINE_STEP=283 MI_DBG_TIMELINE_OUT=gpurun_out/timeline_ko$K.bin MI_PHYSICS_LIB=build_exp/libmi_physics_knocktl.so timeout 300 python /tmp/tlk.py 2>&1 | tail -1
    for F in gpurun_out/timeline_ko$K.bin gpurun_out/timeline_ko$K.bin.knock; do
      echo "== visit stamps $F (MI_DBG_KNOCKOUT=$K; .knock = the knock-out launch)" | tee -a $OUT
      python tools/visit_stamps.py $F | tee -a $OUT
      rm -f $F      # (16 MB each: gpurun_out/ travels back only while it stays under 64 MiB)
    done
  done
fi
