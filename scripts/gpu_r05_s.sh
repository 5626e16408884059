#!/bin/bash
# round 5, GPU call S: the scatter-add's concentrated-scene plan on the C3 shape: parity, per-level timeline, bench variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_chunked.py tests/test_gpu_flush_adam.py -q --tb=short -k "sliced or chunked or flush" 2>&1 | tail -12 > $O/pytest_sel.txt
timeout 500 python profiles/microbench/scatter_timeline.py --scene garden 2>&1 | grep -v "amdgpu.ids\|^Hash" > $O/timeline_concentrated.txt
G="--steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512"
timeout 300 python bench.py $G > $O/garden_conc_default.json 2> $O/err.txt
NGP_BWD_CONCENTRATED=0 timeout 300 python bench.py $G > $O/garden_conc_off.json 2>> $O/err.txt
NGP_BWD_HASHED_REP=2 timeout 300 python bench.py $G > $O/garden_rep2.json 2>> $O/err.txt
NGP_BWD_HASHED_REP_RES=512 NGP_BWD_HASHED_REP=2 timeout 300 python bench.py $G > $O/garden_res512_rep2.json 2>> $O/err.txt
NGP_BWD_MERGE_HASHED=0 timeout 300 python bench.py $G > $O/garden_nomerge.json 2>> $O/err.txt
NGP_BWD_HASHED_REP=1 timeout 300 python bench.py $G > $O/garden_mergeonly.json 2>> $O/err.txt
tail -n 4 $O/pytest_sel.txt
cat $O/timeline_concentrated.txt
for f in $O/garden_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'scatter', round(d['kernels']['hash_bwd_f32']['avg_ms']*1e3,1), 'adam', round(d['kernels']['adam']['avg_ms']*1e3,1))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
tail -2 $O/err.txt
