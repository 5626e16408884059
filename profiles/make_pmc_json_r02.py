"""profiles/r02_pmc.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, each in its own run of the SAME bench.py
command, as MI355X_MICROARCH.md prescribes: the two TCC counters do not fit one pass).  HBM bytes per launch =
2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B for wide coalesced reads
(calibrated in round 1 on the Adam pass: 4 x 45.7 MB of reads reported as 91 MB); for 8/12-byte gathers that factor is an upper
bound.  Medians over the launches of each kernel in the last `--tail` launches (the timed region + the no-prefetch leg).
The workload state of the runs (live / marched samples per step, taken from the bench lines the two runs printed) is stored with
the numbers: bench.py attaches `traffic` only when its own run is in the same state (10 %).
usage: python profiles/make_pmc_json_r02.py <fetch_counter_collection.csv> <fetch_bench.json> <write_counter_collection.csv> <write_bench.json> <out.json>"""
import csv
import json
import statistics
import sys

KEYS = {"hash_fwd_f32_xcd_kernel": "hash_fwd_f32", "hash_bwd_lds_kernel": "hash_bwd_f32", "hash_bwd_prep_kernel": "hash_bwd_prep",
        "hash_bwd_f32x2_kernel": "hash_bwd_f32_atomic", "mlp_fwd_kernelILb1": "mlp_fwd", "mlp_bwd_kernel": "mlp_bwd", "adam_all_kernel": "adam",
        "HIP_vector_typeIfLj4EEPvS3_S3_l": "adam", "march_count_kernel": "march_count", "composite_train_fused_kernel": "composite_fused"}


def medians(path, counter, tail):
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for pat, key in KEYS.items():
            if pat in r["Kernel_Name"]:
                per.setdefault(key, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                break
    out = {}
    for k, v in per.items():
        v = [x for _, x in sorted(v)][-tail:]
        out[k] = (statistics.median(v), len(v))
    return out


if __name__ == "__main__":
    fcsv, fjson, wcsv, wjson, outp = sys.argv[1:6]
    tail = 200
    f, w = medians(fcsv, "FETCH_SIZE", tail), medians(wcsv, "WRITE_SIZE", tail)
    lines = [json.loads(open(p).read().strip().splitlines()[-1]) for p in (fjson, wjson)]
    state = {"regime": lines[0]["config"]["workload_state"]["regime"], "rays": lines[0]["config"]["rays_per_gpu"],
             "live_samples_per_step": sum(l["live_samples_per_step"] for l in lines) / 2,
             "marched_samples_per_step": sum(l["rm_samples_per_ray"] * l["config"]["rays_per_gpu"] for l in lines) / 2,
             "per_run_live": [l["live_samples_per_step"] for l in lines]}
    kernels = {k: {"fetch_kib_raw": f[k][0], "write_kib": w.get(k, (0.0, 0))[0], "launches_sampled": f[k][1],
                   "hbm_bytes_per_launch": int(2 * f[k][0] * 1024 + w.get(k, (0.0, 0))[0] * 1024)} for k in f}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline",
               "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE tallies 128-B requests at 64 B; upper bound for narrow gathers)",
               "state": state, "kernels": kernels}, open(outp, "w"), indent=1)
    print(json.dumps({"state": state, "kernels": kernels}, indent=1))
