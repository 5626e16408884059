#!/bin/bash
# round 5, final GPU call: the whole GPU suite, smoke(), the driver's bench invocation, the long run, the no-flag run, a functional 2-rank run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05final; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_steps200.json 2> /dev/null
timeout 300 python bench.py > $O/bench_noflags.json 2> /dev/null
NGP_BENCH_BACKEND=gloo NGP_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --condition 256 > $O/bench_2ranks_gloo_one_gpu.json 2> $O/bench_2ranks.err
G="--steps 20 --warmup 5 --no-configs --no-cpu-baseline --scene garden --condition 512"
timeout 300 python bench.py $G > $O/garden_c3_default.json 2> /dev/null
NGP_PREFETCH_AT=3 NGP_SIDE_PRIORITY=low timeout 300 python bench.py $G > $O/garden_c3_march16.json 2> /dev/null
NGP_BWD_CONCENTRATED=0 timeout 300 python bench.py $G > $O/garden_c3_plan_off.json 2> /dev/null
# (information for the next round, last in the call: the other heavy-march cases with the 4-wave low-priority march C3 now uses)
H="--steps 20 --warmup 5 --no-configs --no-cpu-baseline"
timeout 200 python bench.py $H --rays 65536 > $O/lego65536_default.json 2> /dev/null
NGP_PREFETCH_AT=3 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0 timeout 200 python bench.py $H --rays 65536 > $O/lego65536_march4.json 2> /dev/null
timeout 200 python bench.py $H --regime random50 > $O/init_default.json 2> /dev/null
NGP_PREFETCH_AT=3 NGP_SIDE_PRIORITY=low NGP_MARCH_SHAPE=4,0 timeout 200 python bench.py $H --regime random50 > $O/init_march4.json 2> /dev/null
tail -n 4 $O/pytest_gpu.txt | head -2; tail -1 $O/smoke.txt
for f in $O/bench_steps20.json $O/bench_steps200.json $O/bench_noflags.json $O/bench_2ranks_gloo_one_gpu.json $O/garden_c3_*.json $O/lego65536_*.json $O/init_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'ns/live', d.get('ns_per_live_sample'), 'noprefetch', d.get('ms_per_step_no_prefetch'))
    for c in d.get('configs',[]): print('   ', c['name'], round(c['value']/1e6,2), round(c['ms_per_step'],3), c.get('comm_ms'), c.get('exposed_comm_ms'), c.get('bus_bandwidth_GBs'), c.get('error'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
