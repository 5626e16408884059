#!/bin/bash
# round 6: rocprofv3 kernel trace of the reference-shaped loop (bench.py --path modules): where the 0.60 ms go against FusedTrainer's 0.48
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06modules; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-configs --path modules > $O/bench.json 2> $O/prof.err
python profiles/timed_region_modules_r06.py $O/prof 40 > $O/rocprofv3_timed_region.txt 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
head -120 $O/rocprofv3_timed_region.txt
