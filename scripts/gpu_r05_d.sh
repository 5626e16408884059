#!/bin/bash
# round 5, GPU call D: prologue folded into the scatter-add, march at the start of the step as 4-wave blocks: tests, sweep, bench, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flush_adam.py tests/test_gpu_trainer.py tests/test_gpu_bench_contract.py -q 2>&1 | tail -15 > $O/pytest_sel.txt
timeout 600 python profiles/microbench/march_placement.py --configs="-1:16:0:def;3:16:0:low;0:4:0:def;0:4:0:low;0:8:0:def;0:16:0:def;0:4:20480:def;2:4:0:def" --json $O/march_placement.json > $O/march_placement.txt 2> $O/march_placement.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline > $O/bench_steps200.json 2> $O/bench_steps200.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $O/prof_bench.json 2> $O/prof.err
python profiles/timed_region_r05.py $O/prof 200 > $O/rocprofv3_timed_region.txt 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats.csv \;
rm -rf $O/prof
tail -n 6 $O/pytest_sel.txt
cat $O/march_placement.txt
head -50 $O/rocprofv3_timed_region.txt
for f in $O/bench_*.json $O/prof_bench.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'ns/live', d.get('ns_per_live_sample'), 'noprefetch', d.get('ms_per_step_no_prefetch'), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, 'roof', d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'],3)))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
