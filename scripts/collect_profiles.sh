#!/bin/bash
# Collects the artefacts profiles/README.md lists for the current round on a GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh <outdir under gpurun_out/> [quick]
# bench lines (driver invocation, long run, float-atomic A/B, config lines), the rocprofv3 kernel trace + stats of the long run,
# and the two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the driver invocation.
set -u
OUT=gpurun_out/${1:-collect}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_steps200.json" 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-events > "$OUT/bench_steps200_no_kernel_events.json" 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --kernel-events-every 1 > "$OUT/bench_steps200_events_every_step.json" 2>/dev/null
NGP_HASH_BWD=atomic python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/bench_steps200_atomic_bwd.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline > "$OUT/prof_bench.json" 2> "$OUT/prof.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err"
if [ "${2:-}" != "quick" ]; then
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --rays 65536 > "$OUT/bench_65536rays.json" 2>/dev/null
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --table bf16 > "$OUT/bench_bf16_table.json" 2>/dev/null
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --half > "$OUT/bench_half_c5.json" 2>/dev/null
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --scene garden > "$OUT/bench_garden_c3.json" 2>/dev/null
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --path modules > "$OUT/bench_modules_path.json" 2>/dev/null
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --regime lego > "$OUT/bench_legacy_lego_regime.json" 2>/dev/null
fi
# keep the merge-back small: the kernel traces of the pmc passes are large, only the counter tables are needed
rm -f "$OUT"/pmc_*/p_kernel_trace.csv "$OUT"/pmc_*/p_agent_info.csv
ls -la "$OUT" "$OUT"/prof "$OUT"/pmc_fetch 2>/dev/null | head -40
