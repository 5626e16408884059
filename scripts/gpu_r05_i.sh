#!/bin/bash
# round 5, GPU call I: the whole GPU suite at HEAD, the driver's bench invocation (with configs + cpu baseline), the long run, smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs > $O/bench_steps200.json 2> /dev/null
timeout 300 python bench.py > $O/bench_noflags.json 2> /dev/null
tail -n 6 $O/pytest_gpu.txt; tail -2 $O/smoke.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], 'Mrays/s', round(d['value']/1e6,3), 'ms', round(d['ms_per_step'],4), 'live', d.get('live_samples_per_step'), 'ns/live', d.get('ns_per_live_sample'), 'noprefetch', d.get('ms_per_step_no_prefetch'), 'roof', d['roofline'] and (d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline'].get('traffic')))
    for c in d.get('configs',[]): print('   ', c['name'], round(c['value']/1e6,2), round(c['ms_per_step'],3), c.get('live_samples_per_step'), c.get('shaded_samples_last_step'), c.get('error'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
