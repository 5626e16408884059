#!/bin/bash
# round 5, GPU call E: chunked forward (tests, C3 bench with and without), single-trip flush (default bench), two-rank bench with the exchange variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunked.py tests/test_gpu_flush_adam.py -q -x 2>&1 | tail -30 > $O/pytest_chunked.txt
timeout 600 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_configs.py -q 2>&1 | tail -30 > $O/pytest_bench.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512 > $O/bench_garden_chunked.json 2> $O/bench_garden_chunked.err
NGP_CHUNKED_FWD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512 > $O/bench_garden_all.json 2> $O/bench_garden_all.err
tail -n 12 $O/pytest_chunked.txt $O/pytest_bench.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'shaded', d.get('shaded_samples_last_step'), 'rm/ray', d.get('rm_samples_per_ray'), 'vr/ray', d.get('vr_samples_per_ray'), {k:(round(v['avg_ms']*1e3,1), v['launches']) for k,v in d['kernels'].items()})
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
tail -3 $O/bench_garden_chunked.err
