"""NGP_EXPERIMENT -- the one environment variable behind which every A/B knob of the package lives.

    NGP_EXPERIMENT="flush_adam=0;march_shape=4,0"        (items separated by ';', key=value, keys as listed in KEYS)

Until round 5 each of these was an NGP_* variable of its own (32 of them, most combinations never run).  What a USER may want to touch
stays a named variable: NGP_FUSED_RENDER / NGP_FUSED_MLP / NGP_FUSED_EVAL / NGP_FUSED_OCCUPANCY (the reference's
operator-by-operator paths), NGP_HASH_BWD (the float-atomic scatter-add), NGP_DETERMINISTIC, NGP_NO_APEX, NGP_HIPCC_EXTRA.  Everything here
is for timing experiments and recorded negatives; none is needed -- every default is the measured-fastest form -- and a key that is not in
KEYS is an error, not a silent no-op.  The keys marked [C] are read by the library itself (csrc/ngp_device.h: ngp_experiment)."""
import os

KEYS = {
    # ---- FusedTrainer: the launch sequence of one step
    "flush_adam": "0: the table's optimizer as its own launch (rounds 1-4) instead of inside the scatter-add's flush",
    "live_backward": "0: backward over every marched sample instead of the live ones",
    "march_fused": "0: the count / scan / write chain instead of the one-launch march",
    "march_rng": "torch: jitter from torch.rand instead of the march kernel's counter-based generator",
    "prefetch_host_wait": "1: the HOST waits for the prefetched march where it was issued a step earlier (no cross-queue wait packet on the main stream; default: the stream-side wait)",
    "prefetch_at": "0 / 1 / 2 / 2.5 / 2.75 / 3 / 4: where in the step the next batch's march goes on the side stream (unset: adaptive)",
    "side_priority": "low / default: stream priority of that march (unset: adaptive)",
    "march_shape": "'waves,idle_lds_bytes': launch shape of that march (unset: adaptive)",
    "chunked_fwd": "1 / 0: shade in chunks of 64 / 64 / 128 / ... samples per live ray (unset: on for multi-cascade scenes)",
    "bwd_concentrated": "1 / 0: the scatter-add's concentrated-scene task plan (unset: on for multi-cascade scenes)",
    "mlp_dw": "atomic: round 3's float-atomic weight gradients instead of per-block slabs",
    "mlp_dw_reduce": "prologue: the slab sum in the prologue launch instead of the head of the scatter-add launch",
    # ---- the overlapped multi-GPU exchange (default off until a multi-GPU run has decided)
    "comm_overlap": "1: one scatter-add launch + async reduce-scatter / Adam / all-gather per level group",
    "comm_groups": "first level of each group in launch order (default '8,0')",
    "comm_scatter_blocks": "workgroup cap of the later group launches (default 240)",
    "comm_stub": "1: every collective replaced by its local part (bench.py: what of comm_ms is exposed)",
    # ---- [C] launch-shape knobs of the library
    "bwd_rep_target": "[C] tasks a replicated level of the scatter-add gets (48)",
    "bwd_merge_res": "[C] run pre-summing on levels up to this resolution (128)",
    "bwd_dense_min_rep": "[C] sample ranges per dense slice, at least (8)",
    "bwd_merge_chunks": "[C] run-pre-summing levels: at least this many super-chunks per task (32; 0: round 5's slice-count rule alone)",
    "bwd_hashed_rep_res": "[C] concentrated plan: hashed levels up to this resolution get replicas (256)",
    "bwd_hashed_rep": "[C] ... this many (3)",
    "bwd_merge_hashed": "[C] concentrated plan: run pre-summing on those levels too (0)",
    "bwd_knobs_dynamic": "[C] re-read the bwd_* knobs on every call (one process A/Bs them on the same inputs)",
    "bwd_levels": "[C] -DNGP_BWD_DIAG builds only: level mask (wrong results by construction)",
    "bwd_diag": "[C] -DNGP_BWD_DIAG builds only: 1 no LDS adds, 2 no gathers, 4 no accumulate, 8 one gather per hit",
    "bwd_blocks": "[C] -DNGP_BWD_DIAG builds only: fewer persistent workgroups",
    "prep_batch": "[C] 3 / 6 / 12: LDS-form levels per fence in the scatter-add's prepass (1)",
    "hash_fwd_v1": "[C] 1: the round-1..3 loop of the forward hash gather",
    "hash_fwd_tiles": "[C] tile cap of the forward hash gather's persistent grid (768)",
    "hash_fwd_free_levels": "[C] -DNGP_HASH_FWD_DIAG builds only: levels whose gathers all read one line",
    "march_group": "[C] 16 / 32 / 64 lanes per ray of the march's count pass (32)",
    "mlp_fwd_blocks": "[C] persistent grid of the MLP forward (768)",
    "mlp_bwd": "[C] reg: round 4's register-resident MLP backward (-DNGP_MLP_BWD_REG builds only)",
}


def parse(text=None):
    """{key: value} of NGP_EXPERIMENT (or of `text`); an unknown key raises."""
    text = os.environ.get("NGP_EXPERIMENT", "") if text is None else text
    out = {}
    for item in text.split(";"):
        item = item.strip()
        if not item:
            continue
        key, sep, value = item.partition("=")
        key = key.strip()
        if not sep or key not in KEYS:
            raise ValueError("NGP_EXPERIMENT: %r is not 'key=value' with a key of ngp_hip.experiment.KEYS" % item)
        out[key] = value.strip()
    return out


def get(key, default=None):
    if key not in KEYS:
        raise KeyError(key)
    return parse().get(key, default)


def has(key):
    return get(key) is not None
