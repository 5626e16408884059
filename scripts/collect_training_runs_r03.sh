#!/bin/bash
# Round-3 training artefacts on a GPU box (through gpurun, from the repo root; ref_lease.tgz must have been made by
# scripts/make_ref_lease.sh first):   bash scripts/collect_training_runs_r03.sh <outdir under gpurun_out/>
# the reference's UNCHANGED train.py on the procedural NSVF scene (default recipe, distortion loss, half2 encoder), the
# FusedTrainer / modules loops of examples/train_procedural.py, and the full GPU test suite (incl. the reference-train tests).
set -u
O=gpurun_out/${1:-r03_train}
mkdir -p "$O"
timeout 600 python scripts/run_reference_train.py --out $O/reference_train_py.json > $O/ref_train.log 2>&1; tail -c 300 $O/ref_train.log; echo
timeout 300 python scripts/run_reference_train.py --max_steps 5000 --extra="--distortion_loss_w 1e-3" --out $O/reference_train_py_distortion.json > $O/ref_train_dist.log 2>&1; tail -c 200 $O/ref_train_dist.log; echo
timeout 300 python scripts/run_reference_train.py --max_steps 20000 --extra=--half_opt --out $O/reference_train_py_half.json > $O/ref_train_half.log 2>&1; tail -c 200 $O/ref_train_half.log; echo
timeout 300 python examples/train_procedural.py --steps 3000 --path trainer --out $O/procedural_training_trainer.json > $O/proc_trainer.log 2>&1; tail -c 200 $O/proc_trainer.log; echo
timeout 300 python examples/train_procedural.py --steps 3000 --path trainer --encoder half --out $O/procedural_training_trainer_half.json > $O/proc_half.log 2>&1; tail -c 200 $O/proc_half.log; echo
timeout 300 python examples/train_procedural.py --steps 3000 --path modules --out $O/procedural_training_modules.json > $O/proc_modules.log 2>&1; tail -c 200 $O/proc_modules.log; echo
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
