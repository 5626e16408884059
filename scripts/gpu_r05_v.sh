#!/bin/bash
# round 5, GPU call V (last): the placement rule for heavy marches -- in line for dense grids, 4-wave low-priority blocks otherwise
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05v; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_trainer.py -q --tb=short -x -k "placement or prefetch" 2>&1 | tail -15 > $O/pytest_sel.txt
H="--steps 20 --warmup 5 --no-configs --no-cpu-baseline"
timeout 100 python bench.py $H --regime random50 > $O/init_rule.json 2> $O/err.txt
timeout 100 python bench.py $H --rays 65536 > $O/lego65536_rule.json 2>> $O/err.txt
tail -n 6 $O/pytest_sel.txt
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'noprefetch', d.get('ms_per_step_no_prefetch'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
grep -v "amdgpu.ids\|^Hash" $O/err.txt | tail -3
