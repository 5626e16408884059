#!/bin/bash
# round 5, GPU call H: tests touched since call G; train.py cProfile with FusedAdam.step(grad_scaler=); then a -DNGP_BWD_DIAG -DNGP_HASH_FWD_DIAG
# build for two gate experiments: scatter-add pieces switched off, gathers of levels 0-1 made free
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_apex_adam.py tests/test_gpu_chunked.py tests/test_gpu_e2e_state.py tests/test_gpu_fused.py tests/test_dropin_imports.py tests/test_gpu_half_fused.py -q --tb=short 2>&1 | tail -60 > $O/pytest_sel.txt
timeout 300 python scripts/run_reference_train.py --max_steps 4000 --wh 400 --n_train 25 --n_test 2 --cprofile $O/train_py_cprofile.txt --out $O/reference_train_py_cprofile_run.json > $O/ref_train_cprof.out 2>&1
timeout 600 python scripts/run_reference_train.py --out $O/reference_train_py.json --log $O/reference_train_py.log > $O/ref_train.out 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --path modules > $O/bench_modules.json 2> $O/bench_modules.err
export NGP_HIPCC_EXTRA="-DNGP_BWD_DIAG -DNGP_HASH_FWD_DIAG"
timeout 600 python -c "
import sys; sys.path.insert(0, 'taichi-nerfs_amd')
from ngp_hip import lib; lib.build(); print('diag build done')" > $O/diag_build.txt 2>&1
timeout 600 python profiles/microbench/encoder_ab.py NGP_HASH_FWD_FREE_LEVELS 0 0x3 0x3f 0xffc0 0xffff > $O/hash_fwd_free_levels.txt 2>&1
timeout 600 python profiles/microbench/hash_bwd_variants.py > $O/hash_bwd_gate.txt 2>&1
tail -n 12 $O/pytest_sel.txt
tail -n 3 $O/ref_train.out | cut -c1-600
head -24 $O/train_py_cprofile.txt
cat $O/hash_fwd_free_levels.txt | tail -20
tail -45 $O/hash_bwd_gate.txt
python - $O/bench_modules.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), d['config'].get('path'))
PY
