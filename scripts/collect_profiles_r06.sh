#!/bin/bash
# Round-6 artefacts on a GPU box (through gpurun, from the repo root):   bash scripts/collect_profiles_r06.sh <outdir under gpurun_out/>
# bench lines (driver invocation with `configs`, long run), the rocprofv3 kernel trace + stats of the long run, and SEPARATE --pmc
# passes of the driver invocation: FETCH_SIZE, WRITE_SIZE (HBM bytes per launch, MI355X_MICROARCH.md), two SQ passes (wave /
# wait / active cycles; instruction mix + LDS conflicts) and TCP / TCC passes (vector-L1 accesses, L2 requests, hits / misses).
# Raw counter tables are summarised on the box (profiles/pmc_summarize.py: mean over the last 25 launches of each kernel = the 20
# timed steps + the 5 warm-up steps of the run) and deleted: gpurun merges at most 64 MiB back.
set -u
OUT=gpurun_out/${1:-r06_collect}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --steps 20 --warmup 5 > "$OUT/bench_steps20.json" 2> "$OUT/bench_steps20.err"
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs > "$OUT/bench_steps200.json" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs > "$OUT/prof_bench.json" 2> "$OUT/prof.err"
pass() {   # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs > "$OUT/pmc_${name}_bench.json" 2> "$OUT/pmc_$name.err"
  local csv=$(find "$OUT/pmc_$name" -name "*counter_collection.csv" | head -1)
  python profiles/pmc_summarize.py --tail 25 "ngp" "$csv" > "$OUT/pmc_$name.json"
  rm -rf "$OUT/pmc_$name"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16
pass sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
# the stats table of the long run is small; the raw kernel trace (tens of MB) is reduced to the timed region by timed_region_r05.py
python profiles/timed_region_r05.py "$OUT/prof" 200 > "$OUT/rocprofv3_timed_region.txt" 2>&1
find "$OUT/prof" -name "*kernel_trace.csv" -delete
du -sh "$OUT"; ls "$OUT"
