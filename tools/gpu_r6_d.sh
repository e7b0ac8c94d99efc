#!/bin/bash
# round 6, last session: the terrain kernel's two instances (plain / large window) — the terrain tests first, the terrain configuration's step time, then the whole GPU suite + smoke
# and the bench at the driver's flags
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "terrain or heightmap" > gpurun_out/d_terrain_pytest.log 2>&1; echo "terrain pytest rc=$?" >> gpurun_out/d_terrain_pytest.log
tail -3 gpurun_out/d_terrain_pytest.log
CFGS=terrain bash tools/gpu_cfgs.sh 2>&1 | tail -2 | cut -c1-900; cp gpurun_out/cfgs.json gpurun_out/d_terrain_cfg.json
echo "terrain at $(( $(date +%s) - T0 )) s"
timeout 1500 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/d_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/d_pytest.log
grep -E "passed|failed|rc=" gpurun_out/d_pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
echo "suite + smoke at $(( $(date +%s) - T0 )) s"
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/d_bench_driver_flags.json; cut -c1-300 gpurun_out/d_bench_driver_flags.json
echo "all done at $(( $(date +%s) - T0 )) s"
