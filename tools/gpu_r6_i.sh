#!/bin/bash
# round 6, last call: at HEAD — smoke, the default bench (what `python bench.py` without flags prints), the other configurations' step times
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
T0=$(date +%s)
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 python bench.py 2>/dev/null | tail -1 > gpurun_out/i_bench_default.json; cut -c1-260 gpurun_out/i_bench_default.json
echo "bench at $(( $(date +%s) - T0 )) s"
CFGS=cfg1,cfg2,cfg4,cfg5,terrain,zones,cfg3 timeout 200 bash tools/gpu_cfgs.sh 2>&1 | tail -7 | cut -c1-160; cp gpurun_out/cfgs.json gpurun_out/i_other_configs.json
echo "all done at $(( $(date +%s) - T0 )) s"
