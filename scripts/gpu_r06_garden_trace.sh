#!/bin/bash
# round 6: rocprofv3 kernel trace of the C3 shape (chunked forward, march under the scatter-add): timed region + one step's timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06garden; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-configs --scene garden --condition 512 > $O/bench.json 2> $O/prof.err
python profiles/timed_region_r05.py $O/prof 40 512 8 > $O/rocprofv3_timed_region.txt 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
head -60 $O/rocprofv3_timed_region.txt
