#!/bin/bash
# Round-4 training artefacts on a GPU box (through gpurun, from the repo root; ref_lease.tgz must have been made by
# scripts/make_ref_lease.sh first):   bash scripts/collect_training_runs_r04.sh <outdir under gpurun_out/>
# the reference's UNCHANGED train.py (json summary + its own stdout log, verbatim) on the procedural NSVF scene: default recipe
# (compat/apex FusedAdam is picked up by train.py:143-149), distortion loss, half2 encoder (fused render), the Garden recipe;
# a 2-rank functional bench line with the overlapped exchange; the full GPU test suite.
set -u
O=gpurun_out/${1:-r04_train}
mkdir -p "$O"
run() { name=$1; shift; timeout 600 python scripts/run_reference_train.py "$@" --out $O/reference_train_py$name.json --log $O/reference_train_py$name.log > $O/run$name.txt 2>&1; tail -c 200 $O/run$name.txt; echo; }
run "" 
run _distortion --max_steps 5000 --extra="--distortion_loss_w 1e-3"
run _half --max_steps 20000 --extra=--half_opt
run _garden --scene garden --wh 200 --batch_size 4096 --max_steps 20000 --extra="--distortion_loss_w 1e-3"
run _torch_adam --env "NGP_NO_APEX=1"
NGP_BENCH_BACKEND=gloo NGP_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo_one_gpu.json 2> /dev/null
NGP_COMM_OVERLAP=1 NGP_BENCH_BACKEND=gloo NGP_BENCH_ONE_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo_one_gpu_overlap.json 2> /dev/null
timeout 1500 python -m pytest tests -m gpu -q --timeout 300 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
ls $O
