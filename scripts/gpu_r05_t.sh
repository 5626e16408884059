#!/bin/bash
# round 5, GPU call T: with the concentrated-scene scatter-add (1.7 ms instead of 2.4) the C3 march outlives it: where the march goes now
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05t2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chunked.py tests/test_gpu_flush_adam.py -q --tb=short -k "sliced or chunked or flush" 2>&1 | tail -5 > $O/pytest_sel.txt
G="--steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $G > $O/garden_$name.json 2>> $O/err.txt; }
run adaptive X=1
run at3_low NGP_PREFETCH_AT=3 NGP_SIDE_PRIORITY=low
run at3_def NGP_PREFETCH_AT=3 NGP_SIDE_PRIORITY=default
run at275_low NGP_PREFETCH_AT=2.75 NGP_SIDE_PRIORITY=low
run at25_low NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=low
run at25_def NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=default
run at25_low_w4 NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
run at2_low NGP_PREFETCH_AT=2 NGP_SIDE_PRIORITY=low
run at25_low_nomerge NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=low NGP_BWD_MERGE_HASHED=0
run at275_low_nomerge NGP_PREFETCH_AT=2.75 NGP_SIDE_PRIORITY=low NGP_BWD_MERGE_HASHED=0
run at25_low_off NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=low NGP_BWD_CONCENTRATED=0
tail -n 3 $O/pytest_sel.txt
for f in $O/garden_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d['kernels']
    print(sys.argv[1].split('/')[-1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'prep', round(k['hash_bwd_prep']['avg_ms']*1e3,1), 'mlp_bwd', round(k['mlp_bwd']['avg_ms']*1e3,1), 'scatter', round(k['hash_bwd_f32']['avg_ms']*1e3,1), 'adam', round(k['adam']['avg_ms']*1e3,1))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
grep -v "amdgpu.ids\|^Hash" $O/err.txt | tail -3
