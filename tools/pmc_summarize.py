"""Turn the two rocprofv3 --pmc passes over tools/pmc_probe.py into per-launch HBM bytes per decode stage.

usage: python tools/pmc_summarize.py FETCH_SIZE_counter_collection.csv WRITE_SIZE_counter_collection.csv out.json [batch]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled because gfx950 reports half of a wide
coalesced stream (MI355X_MICROARCH.md, HBM / rocprofv3 section); the median over the profiled launches of a stage is used
(the probe launches every stage a few times)."""
import csv
import json
import statistics
import sys

STAGE_OF = [  # (substring of the kernel name, stage)
    # round 3: the K-split kernel of decode_ks.hip <B, MODE, NORM, XATTN, CPL> serves batch 1-2 on bf16 weights
    # (round 5: <B, MODE, NORM, XS, CPL, R32> with XS = 0 global rows / 1 attention split partials / 2 embedding rows)
    ("emmax_decode_ks_kernel<1, 0,", "qkv_gemv"), ("emmax_decode_ks_kernel<1, 1, false, 1,", "oproj_gemv"),
    ("emmax_decode_ks_kernel<1, 2,", "gateup_gemv"), ("emmax_decode_ks_kernel<1, 1, false, 0,", "down_gemv"),
    ("emmax_decode_ks_kernel<1, 1, false, true,", "oproj_gemv"), ("emmax_decode_ks_kernel<1, 1, false, false,", "down_gemv"),   # rounds 3-4
    ("emmax_decode_ks_kernel<1, 3,", "lmhead_argmax"),
    ("emmax_decode_gemv_kernel<1, 0,", "qkv_gemv"), ("emmax_decode_attn_kernel", "paged_attn"), ("emmax_x_decode_attn_kernel", "paged_attn"),   # (x: exact numerics, round 6)
    ("emmax_decode_gemv_kernel<1, 1, false, true,", "oproj_gemv"), ("emmax_decode_gemv_kernel<1, 2,", "gateup_gemv"),
    ("emmax_decode_gemv_kernel<1, 1, false, false,", "down_gemv"), ("emmax_decode_gemv_kernel<1, 3,", "lmhead_argmax"),
    # batch >= 3 (PROBE_BATCH=8), round 3: the K-split MFMA kernels of decode_km.hip <MODE, NORM, XATTN, FP8, F8N> / the two-phase down kernel
    ("emmax_decode_km_kernel<0,", "qkv_gemv"), ("emmax_decode_km_kernel<1,", "oproj_gemv"), ("emmax_decode_km_kernel<2,", "gateup_gemv"),
    ("emmax_decode_km_kernel<3,", "lmhead_argmax"), ("emmax_decode_kmd_kernel<", "down_gemv"),
    # batch 17-32 (round 5): decode_kmp.hip <MODE, NORM, R32, TMAX, NPH>; the down projection is its 11-phase RESID form
    ("emmax_decode_kmp_kernel<0,", "qkv_gemv"), ("emmax_decode_kmp_kernel<1, false, true, 1, 11>", "down_gemv"),
    ("emmax_decode_kmp_kernel<1, false, false, 1, 11>", "down_gemv"), ("emmax_decode_kmp_kernel<1,", "oproj_gemv"),
    ("emmax_decode_kmp_kernel<2,", "gateup_gemv"), ("emmax_decode_kmp_kernel<3,", "lmhead_argmax"),
    # batch >= 3 (PROBE_BATCH=8): the MFMA small-batch kernel <MODE, NORM, XATTN, FP8>
    ("emmax_decode_mfma_kernel<0,", "qkv_gemv"), ("emmax_decode_mfma_kernel<1, false, true,", "oproj_gemv"),
    ("emmax_decode_mfma_kernel<2,", "gateup_gemv"), ("emmax_decode_mfma_kernel<1, false, false,", "down_gemv"),
    ("emmax_decode_mfma_kernel<3,", "lmhead_argmax"),
]


def read(path, counter):
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            for sub, stage in STAGE_OF:   # first match wins (more specific patterns are listed first)
                if sub in row["Kernel_Name"]:
                    per.setdefault(stage, []).append(float(row["Counter_Value"]))
                    break
    return {k: statistics.median(v) for k, v in per.items()}


def main():
    fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
    out = {"how": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --kernel-trace --kernel-include-regex emmax_decode "
                  "-- python tools/pmc_probe.py (PROBE_BATCH rows, context 768, full-size layer shapes). FETCH_SIZE doubled: on gfx950 it reports 1/2 of a "
                  "wide coalesced stream (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as reported (uncalibrated).",
           "batch": int(sys.argv[4]) if len(sys.argv) > 4 else 1, "stages": {}}
    for stage in fetch:
        w = write.get(stage, 0.0)
        out["stages"][stage] = {"FETCH_SIZE_KB": round(fetch[stage], 1), "WRITE_SIZE_KB": round(w, 1),
                                "hbm_bytes_per_launch": int((2 * fetch[stage] + w) * 1024)}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["stages"]))


if __name__ == "__main__":
    main()
