#!/bin/bash
# round 5, GPU call B: fixed tests, default bench twice (pinned state), two-launch A/B, modules path, rocprofv3 timeline of the long run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flush_adam.py tests/test_gpu_bench_contract.py tests/test_gpu_apex_adam.py tests/test_gpu_mlp.py -q 2>&1 | tail -40 > $O/pytest_sel.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_1.json 2> $O/bench_default_1.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-configs --no-cpu-baseline > $O/bench_default_2.json 2> $O/bench_default_2.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline > $O/bench_steps200.json 2> $O/bench_steps200.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-deterministic-condition > $O/bench_steps200_nodet.json 2> $O/bench_steps200_nodet.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $O/prof_bench.json 2> $O/prof.err
python profiles/timed_region_r05.py $O/prof 200 > $O/rocprofv3_timed_region.txt 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/rocprofv3_kernel_stats.csv \;
rm -rf $O/prof
tail -n 6 $O/pytest_sel.txt
head -30 $O/rocprofv3_timed_region.txt
for f in $O/bench_*.json $O/prof_bench.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'ns/live', d.get('ns_per_live_sample'), 'noprefetch', d.get('ms_per_step_no_prefetch'), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()}, 'roof', d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'],3)))
    for c in d.get('configs',[]): print('   ', c['name'], c.get('value'), c.get('ms_per_step'), c.get('live_samples_per_step'), c.get('error'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
