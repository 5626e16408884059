#!/bin/bash
# round 5, GPU call M: the scatter-add's plan knobs re-swept on round 5's kernel (optimizer in the flush), pinned bench state
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05m; mkdir -p $O
B="--steps 200 --warmup 20 --no-configs --no-cpu-baseline"
for cfg in "48 8 128" "48 16 128" "48 16 512" "48 8 512" "48 16 256" "48 8 128" "48 16 128" "48 16 512" "48 8 512" "48 16 256" "48 8 128" "48 16 128" "48 16 512" "48 8 512" "48 16 256"; do
  set -- $cfg
  NGP_BWD_REP_TARGET=$1 NGP_BWD_DENSE_MIN_REP=$2 NGP_BWD_MERGE_RES=$3 timeout 300 python bench.py $B > $O/b_$1_$2_$3.json 2>/dev/null
  python - "$O/b_$1_$2_$3.json" "$cfg" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("rep_target dense_min_rep merge_res = %-12s ms %.4f  live %.0f  ns/live %.4f  scatter-add %.1f us" % (sys.argv[2], d['ms_per_step'], d['live_samples_per_step'], d['ns_per_live_sample'], d['kernels']['hash_bwd_f32']['avg_ms']*1e3))
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
done
