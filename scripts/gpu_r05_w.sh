#!/bin/bash
# round 5, GPU call W (the round's last): the placement test and the config tests at HEAD, then the driver's bench line with its configs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_configs.py -q --tb=short -x -k "placement or prefetch or garden or c3 or C3" 2>&1 | tail -8 > $O/pytest_sel.txt
timeout 110 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20_nocpu.json 2> $O/err.txt
tail -n 3 $O/pytest_sel.txt
python - $O/bench_steps20_nocpu.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'))
for c in d.get('configs',[]): print('   ', c['name'], round(c['value']/1e6,2), round(c['ms_per_step'],3), c.get('error'))
PY
