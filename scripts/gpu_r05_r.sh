#!/bin/bash
# round 5, GPU call R: C3 small kernels (list builders with one atomic per 64 rays, per-ray MSE gradient, block-ordered live list) + march position for C3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunked.py tests/test_gpu_configs.py tests/test_gpu_trainer.py -q --tb=short 2>&1 | tail -15 > $O/pytest_sel.txt
G="--steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512"
timeout 300 python bench.py $G > $O/garden_default.json 2> $O/err.txt
NGP_PREFETCH_AT=2 NGP_MARCH_SHAPE=16,0 NGP_SIDE_PRIORITY=low timeout 300 python bench.py $G > $O/garden_pos2_w16_low.json 2>> $O/err.txt
NGP_PREFETCH_AT=1 NGP_MARCH_SHAPE=16,0 NGP_SIDE_PRIORITY=low timeout 300 python bench.py $G > $O/garden_pos1_w16_low.json 2>> $O/err.txt
timeout 300 python bench.py $G > $O/garden_default_2.json 2>> $O/err.txt
tail -n 6 $O/pytest_sel.txt
for f in $O/garden_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'shaded', d.get('shaded_samples_last_step'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
tail -3 $O/err.txt
