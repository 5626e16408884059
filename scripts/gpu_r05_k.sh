#!/bin/bash
# round 5, GPU call K: A/B on ONE box of the fused render as one C entry per direction (HEAD) against the launch sequence in Python
# (the previous commit's ngp_hip/fused.py, swapped in for the B legs): the unchanged train.py, 20 000 steps, twice each, alternating
# (the B legs' file is made beforehand, outside the history:  mkdir -p scratch && git show 7e9a514:taichi-nerfs_amd/ngp_hip/fused.py > scratch/fused_before_entry.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
cp taichi-nerfs_amd/ngp_hip/fused.py /tmp/fused_entry.py
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_half_fused.py tests/test_gpu_configs.py -q 2>&1 | tail -3 > $O/pytest_sel.txt
for rep in 1 2; do
  cp /tmp/fused_entry.py taichi-nerfs_amd/ngp_hip/fused.py
  timeout 600 python scripts/run_reference_train.py --out $O/train_entry_$rep.json --log $O/train_entry_$rep.log > /dev/null 2>&1
  cp scratch/fused_before_entry.py taichi-nerfs_amd/ngp_hip/fused.py
  timeout 600 python scripts/run_reference_train.py --out $O/train_calls_$rep.json > /dev/null 2>&1
done
cp /tmp/fused_entry.py taichi-nerfs_amd/ngp_hip/fused.py
timeout 300 python scripts/run_reference_train.py --max_steps 4000 --wh 400 --n_train 25 --n_test 2 --cprofile $O/train_py_cprofile_entry.txt --out $O/cprof_entry.json > /dev/null 2>&1
timeout 600 python scripts/run_reference_train.py --extra=--half_opt --out $O/train_entry_half.json --log $O/train_entry_half.log > /dev/null 2>&1
cat $O/pytest_sel.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05k/train_*.json")):
    d=json.load(open(f)); print(f, d["train_seconds"], round(d["train_rays_per_sec"]/1e6,2), round(d["test_psnr_avg"],2))
PY
grep -n "grad_scaler.py:360\|rendering.py:104\|run_backward\|fused.py.*forward\|optimizers.py.*step" $O/train_py_cprofile_entry.txt | head -6
