"""profiles/r06_pmc.json from the per-pass summaries scripts/collect_profiles_r06.sh leaves behind (pmc_<pass>.json = mean per launch
over the last 25 launches of every kernel, pmc_<pass>_bench.json = the bench line that pass printed).  Every pass is its own run of
`rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs`, as
MI355X_MICROARCH.md prescribes (TCC counters do not share a pass; never combined with other trace domains).
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B for wide
coalesced reads (calibrated in round 1 on the Adam pass); for 8/12-byte gathers the factor 2 is an upper bound.
The workload state of the FETCH / WRITE runs is stored: bench.py attaches `traffic` only when its own run matches it within 15 %.
usage: python profiles/make_pmc_json_r05.py <dir with pmc_*.json> <out.json>"""
import json
import os
import sys

KEYS = {"hash_fwd_f32_xcd_kernel<0": "hash_fwd_f32", "hash_bwd_lds_kernel<false": "hash_bwd_f32", "hash_bwd_prep_kernel": "hash_bwd_prep",
        "mlp_fwd_kernelILb1": "mlp_fwd", "mlp_bwd_kernel": "mlp_bwd", "adam_all_kernel": "adam", "march_count_kernel": "march_count", "march_fused_kernel": "march_count",
        "composite_train_fused_kernel": "composite_fused", "train_prologue_reduce_kernel": "prologue_reduce", "train_prologue_kernel": "prologue", "march_write_kernel": "march_write", "live_scan_kernel": "live_scan"}
PASSES = ("fetch", "write", "sq1", "sq2", "sq3", "tcp", "tcc")


def main(d, outp):
    kernels = {}
    for p in PASSES:
        j = json.load(open(os.path.join(d, "pmc_%s.json" % p)))
        for name, vals in j.items():
            key = next((k for pat, k in KEYS.items() if pat in name), None)
            if key is None:
                continue
            kernels.setdefault(key, {"counters": {}})["counters"].update({c: v for c, v in vals.items() if c != "dispatches"})
    for key, k in kernels.items():
        c = k["counters"]
        if "FETCH_SIZE" in c:
            k["fetch_kib_raw"], k["write_kib"] = c["FETCH_SIZE"], c.get("WRITE_SIZE", 0.0)
            k["hbm_bytes_per_launch"] = int(2 * c["FETCH_SIZE"] * 1024 + c.get("WRITE_SIZE", 0.0) * 1024)
        w = c.get("SQ_WAVE_CYCLES")
        if w:
            k["wave_cycle_shares"] = {"wait_any": c.get("SQ_WAIT_ANY", 0) / w, "wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0) / w,
                                      "active_inst_any": c.get("SQ_ACTIVE_INST_ANY", 0) / w}
        if c.get("SQ_BUSY_CU_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            k["mfma_busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_BUSY_CU_CYCLES"])          # 4 SIMDs per CU
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None:
            k["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0)
    lines = [json.loads(open(os.path.join(d, "pmc_%s_bench.json" % p)).read().strip().splitlines()[-1]) for p in ("fetch", "write")]
    state = {"regime": lines[0]["config"]["workload_state"]["regime"], "rays": lines[0]["config"]["rays_per_gpu"],
             "live_samples_per_step": sum(l["live_samples_per_step"] for l in lines) / 2,
             "marched_samples_per_step": sum(l["rm_samples_per_ray"] * l["config"]["rays_per_gpu"] for l in lines) / 2,
             "per_run_live": [l["live_samples_per_step"] for l in lines]}
    out = {"source": "rocprofv3 --kernel-trace --pmc <one pass> --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                     "--no-configs; passes: FETCH_SIZE | WRITE_SIZE | SQ wave/wait/active cycles | SQ instruction mix + LDS | SQ MFMA busy | TCP | TCC",
           "summary": "mean per launch over the last 25 launches of each kernel (profiles/pmc_summarize.py --tail 25)",
           "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE tallies 128-B requests at 64 B; upper bound for narrow gathers)",
           "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over all waves; SQ_BUSY_CU_CYCLES and "
                    "SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over CUs (MI355X_MICROARCH.md); mfma_busy_fraction = MFMA_BUSY / (4 SIMDs x BUSY_CU)",
           "state": state, "kernels": kernels}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} for k, v in kernels.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
