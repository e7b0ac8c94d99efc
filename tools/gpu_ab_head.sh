#!/bin/bash
# same-box A/B: the working tree against a checkout of HEAD built under build_exp/head_src (git worktree add --detach build_exp/head_src HEAD; python -m build there)
export TMPDIR=/tmp
one() { (cd "$1" && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-at-rest 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"); }
for i in 1 2 3; do one . tree; one build_exp/head_src head; done
