#!/bin/bash
# round 5, GPU call G: chunked tests with tracebacks, the whole GPU suite, the unchanged train.py (20 000 steps) + its cProfile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunked.py -q --tb=short 2>&1 | tail -120 > $O/pytest_chunked.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest_gpu.txt
timeout 600 python scripts/run_reference_train.py --out $O/reference_train_py.json --log $O/reference_train_py.log > $O/ref_train.out 2>&1
timeout 300 python scripts/run_reference_train.py --max_steps 4000 --wh 400 --n_train 25 --n_test 2 --cprofile $O/train_py_cprofile.txt --out $O/reference_train_py_cprofile_run.json > $O/ref_train_cprof.out 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --path modules > $O/bench_modules.json 2> $O/bench_modules.err
NGP_BENCH_TORCH_ADAM=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --path modules > $O/bench_modules_torch_adam.json 2>> $O/bench_modules.err
grep -n "FAILED\|Error\|assert" $O/pytest_chunked.txt | head -30
tail -n 8 $O/pytest_gpu.txt
tail -n 4 $O/ref_train.out
head -30 $O/train_py_cprofile.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), d['config'].get('path'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
