#!/bin/bash
# round 5, GPU call C: shaped-march parity, flush-Adam tests, and the march placement sweep (profiles/microbench/march_placement.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flush_adam.py tests/test_gpu_parity.py -q -k "flush_adam or march" 2>&1 | tail -15 > $O/pytest_sel.txt
timeout 600 python profiles/microbench/march_placement.py --json $O/march_placement.json > $O/march_placement.txt 2> $O/march_placement.err
tail -n 5 $O/pytest_sel.txt
cat $O/march_placement.txt
tail -n 5 $O/march_placement.err
