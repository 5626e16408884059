#!/bin/bash
# round 5, GPU call U: C3 march placement with 4-wave blocks at low priority (call T: 14.72 M at 2.5 against 14.46 adaptive), repeated
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05u; mkdir -p $O
G="--steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $G > $O/garden_$name.json 2>> $O/err.txt; }
run adaptive_a X=1
run at25_low_w4_a NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
run at0_low_w4 NGP_PREFETCH_AT=0 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
run at1_low_w4 NGP_PREFETCH_AT=1 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
run at2_low_w4 NGP_PREFETCH_AT=2 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
run at275_low_w4 NGP_PREFETCH_AT=2.75 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
run at3_low_w4 NGP_PREFETCH_AT=3 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
run at25_low_w8 NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=8,0
run at25_def_w4 NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=default NGP_MARCH_SHAPE=4,0
run adaptive_b X=1
run at25_low_w4_b NGP_PREFETCH_AT=2.5 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0
for f in $O/garden_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k=d['kernels']
    print(sys.argv[1].split('/')[-1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'prep', round(k['hash_bwd_prep']['avg_ms']*1e3,1), 'mlp_bwd', round(k['mlp_bwd']['avg_ms']*1e3,1), 'scatter', round(k['hash_bwd_f32']['avg_ms']*1e3,1), 'adam', round(k['adam']['avg_ms']*1e3,1), 'hash_fwd', round(k['hash_fwd_f32']['avg_ms']*1e3,1))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
grep -v "amdgpu.ids\|^Hash" $O/err.txt | tail -3
