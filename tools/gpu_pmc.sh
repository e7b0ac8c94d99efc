#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in SEPARATE runs, kernel-trace only); summaries -> gpurun_out/prof_summary
ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
mkdir -p gpurun_out
export TMPDIR=/tmp
RAW=/tmp/prof_raw; rm -rf $RAW; mkdir -p $RAW
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -o r01 -- python bench.py --steps 10 --warmup 250 --no-cpu-baseline > gpurun_out/stats_bench.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -o r01 -- python bench.py --steps 4 --warmup 250 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -o r01 -- python bench.py --steps 4 --warmup 250 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
python tools/summarize_prof.py $RAW gpurun_out/prof_summary
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.log
