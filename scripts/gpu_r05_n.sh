#!/bin/bash
# round 5, GPU call N: the flush's loads issued in front of the end-of-task barrier: NGP_FLUSH_EARLY = 0 / 4 / 8 entries per thread
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05n; mkdir -p $O
B="--steps 200 --warmup 20 --no-configs --no-cpu-baseline"
timeout 900 python -m pytest tests/test_gpu_flush_adam.py -q 2>&1 | tail -3 > $O/pytest_flush.txt
for e in 4 0 8 4 0 8; do
  export NGP_HIPCC_EXTRA="-DNGP_FLUSH_EARLY=$e"
  python -c "
import sys; sys.path.insert(0, 'taichi-nerfs_amd')
from ngp_hip import lib; lib.build()" > /dev/null 2>&1
  for rep in 1 2; do
    timeout 300 python bench.py $B > $O/b_${e}_$rep.json 2>/dev/null
    python - "$O/b_${e}_$rep.json" "$e" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("EARLY %s  ms %.4f  live %.0f  ns/live %.4f  scatter-add %.1f us" % (sys.argv[2], d['ms_per_step'], d['live_samples_per_step'], d['ns_per_live_sample'], d['kernels']['hash_bwd_f32']['avg_ms']*1e3))
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  done
done
cat $O/pytest_flush.txt
