#!/bin/bash
# round 5, GPU call F: chunk scheduler test, where the prefetched march goes on the C3 shape (chunked forward)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_chunked.py -q 2>&1 | tail -8 > $O/pytest_chunked.txt
G="--steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512"
timeout 300 python bench.py $G > $O/garden_pos0_w4_def.json 2> $O/err.txt
NGP_PREFETCH_AT=3 NGP_MARCH_SHAPE=16,0 NGP_SIDE_PRIORITY=low timeout 300 python bench.py $G > $O/garden_pos3_w16_low.json 2>> $O/err.txt
NGP_PREFETCH_AT=3 NGP_MARCH_SHAPE=4,0 NGP_SIDE_PRIORITY=default timeout 300 python bench.py $G > $O/garden_pos3_w4_def.json 2>> $O/err.txt
NGP_PREFETCH_AT=3 NGP_MARCH_SHAPE=16,0 NGP_SIDE_PRIORITY=default timeout 300 python bench.py $G > $O/garden_pos3_w16_def.json 2>> $O/err.txt
NGP_PREFETCH_AT=2 NGP_MARCH_SHAPE=16,0 NGP_SIDE_PRIORITY=default timeout 300 python bench.py $G > $O/garden_pos2_w16_def.json 2>> $O/err.txt
timeout 300 python bench.py $G --no-prefetch > $O/garden_noprefetch.json 2>> $O/err.txt
tail -n 5 $O/pytest_chunked.txt
for f in $O/garden_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'shaded', d.get('shaded_samples_last_step'), 'noprefetch', d.get('ms_per_step_no_prefetch'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
tail -3 $O/err.txt
