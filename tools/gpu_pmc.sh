#!/bin/bash
# rocprofv3 passes of the default workload (kernel stats; FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc runs, kernel-trace only);
# summaries -> gpurun_out/prof_summary (copy what is to be judged into profiles/)
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp
RAW=/tmp/prof_raw; rm -rf $RAW; mkdir -p $RAW
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o r06 -- $B > gpurun_out/stats_bench.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -o r06 -- $B > gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -o r06 -- $B > gpurun_out/pmc_write.log 2>&1
python tools/summarize_prof.py $RAW gpurun_out/prof_summary gpurun_out/pmc_fetch.log
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json
