#!/bin/bash
# round 5, GPU call J: one C entry per direction for the fused render: tests, the unchanged train.py, cProfile, bench modules path
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_half_fused.py tests/test_gpu_configs.py tests/test_gpu_e2e_state.py tests/test_gpu_bf16_table.py tests/test_gpu_reference_train.py tests/test_gpu_apex_adam.py -q --tb=short 2>&1 | tail -40 > $O/pytest_sel.txt
timeout 300 python scripts/run_reference_train.py --max_steps 4000 --wh 400 --n_train 25 --n_test 2 --cprofile $O/train_py_cprofile.txt --out $O/reference_train_py_cprofile_run.json > $O/ref_train_cprof.out 2>&1
timeout 600 python scripts/run_reference_train.py --out $O/reference_train_py.json --log $O/reference_train_py.log > $O/ref_train.out 2>&1
timeout 600 python scripts/run_reference_train.py --extra "--half_opt" --out $O/reference_train_py_half.json --log $O/reference_train_py_half.log > $O/ref_train_half.out 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline --path modules > $O/bench_modules.json 2> $O/bench_modules.err
tail -n 8 $O/pytest_sel.txt
grep -n "grad_scaler.py:360\|rendering.py:104\|run_backward\|fused.py.*forward\|fused.py.*backward\|optimizers.py.*step" $O/train_py_cprofile.txt | head
python - <<'PY'
import json
for f in ("reference_train_py.json","reference_train_py_half.json"):
    try:
        d=json.load(open("gpurun_out/r05j/"+f)); print(f, d["train_seconds"], d["train_rays_per_sec"], d["test_psnr_avg"])
    except Exception as e: print(f, "ERR", e)
d=json.loads(open("gpurun_out/r05j/bench_modules.json").read().strip().splitlines()[-1])
print('modules', round(d['value']/1e6,3), round(d['ms_per_step'],4))
PY
