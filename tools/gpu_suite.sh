#!/bin/bash
# the GPU suite + smoke at HEAD (what the driver runs at round end)
mkdir -p gpurun_out; export TMPDIR=/tmp; ulimit -c 0
cd oracle && make >/dev/null 2>&1; cd ..
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/suite_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/suite_pytest.log
grep -E "passed|failed|rc=" gpurun_out/suite_pytest.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
