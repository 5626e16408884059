"""profiles/r01_pmc.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, each in its own run as
MI355X_MICROARCH.md prescribes).  HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B for wide coalesced reads (calibrated on the Adam pass here: 4 x 45.7 MB of reads reported
as 91 MB); for the 8-byte gathers of the hash kernels that factor is an upper bound.  Medians over the launches of the
timed loop (the 1-in-16 occupancy-update launches of hash_fwd are outliers by design).
usage: python profiles/make_pmc_json.py <fetch.db> <write.db> <out.json>"""
import json
import sqlite3
import statistics
import sys

KEYS = {"hash_fwd_f32_xcd_kernel": "hash_fwd_f32", "hash_fwd_f32_kernel<2>": "hash_fwd_f32_generic", "hash_bwd_f32x2_kernel": "hash_bwd_f32", "mlp_fwd_kernelILb1": "mlp_fwd",
        "mlp_bwd_kernel": "mlp_bwd", "adam_kernel": "adam", "adam_all_kernel": "adam", "march_count_kernel": "march_count",
        "composite_fwd_kernel": "composite_fwd", "composite_bwd_kernel": "composite_bwd"}


def medians(path, counter):
    db = sqlite3.connect(path)
    out = {}
    for name, val in db.execute("select name, counter_value from pmc_events where counter_name=?", (counter,)):
        for pat, key in KEYS.items():
            if pat in name:
                out.setdefault(key, []).append(val)
    res = {}
    for k, v in out.items():
        if k == "adam":
            v = [x for x in v if x > max(v) / 2]          # table pass only (the 9 408-weight pass is tiny)
        res[k] = statistics.median(v)
    return res


if __name__ == "__main__":
    f, w = medians(sys.argv[1], "FETCH_SIZE"), medians(sys.argv[2], "WRITE_SIZE")
    kernels = {k: {"fetch_kib_raw": f[k], "write_kib": w.get(k, 0.0),
                   "hbm_bytes_per_launch": int(2 * f[k] * 1024 + w.get(k, 0.0) * 1024)} for k in f}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --steps 24 --warmup 8",
               "correction": "hbm = 2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE reads half of a wide stream)", "kernels": kernels},
              open(sys.argv[3], "w"), indent=1)
    print(json.dumps(kernels, indent=1))
