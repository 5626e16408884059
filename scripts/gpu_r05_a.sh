#!/bin/bash
# round 5, GPU call A: the flush-Adam + deterministic-mode tests, the whole GPU suite, the default bench twice (pinned state?),
# the two-launch A/B, the modules path
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flush_adam.py -x -q 2>&1 | tail -25 > $O/pytest_flush.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_1.json 2> $O/bench_default_1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_default_2.json 2> $O/bench_default_2.err
NGP_FLUSH_ADAM=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_two_launch.json 2> $O/bench_two_launch.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline > $O/bench_steps200.json 2> $O/bench_steps200.err
NGP_FLUSH_ADAM=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline > $O/bench_steps200_two_launch.json 2> $O/bench_steps200_two_launch.err
tail -3 $O/pytest_flush.txt $O/pytest_gpu.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'ns/live', d.get('ns_per_live_sample'), 'noprefetch', d.get('ms_per_step_no_prefetch'), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})
    for c in d.get('configs',[]): print('   ', c['name'], c.get('value'), c.get('ms_per_step'), c.get('live_samples_per_step'), c.get('error'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
