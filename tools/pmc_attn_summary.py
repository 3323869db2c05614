"""Summarise the rocprofv3 --pmc passes of tools/gpucmd_attn_pmc.sh: per attention kernel instantiation, mean counter values
and the derived ratios (MFMA pipe busy, VALU busy, LDS conflict share, HBM traffic).  SQ counters are summed over the 8 XCDs;
GRBM_GUI_ACTIVE likewise (per-XCD elapsed cycles = value / 8)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        i, j = k.find("AttnCfg<"), k.find("ResCfg<")
        if i >= 0:
            k = "emmax_attention_kernel<" + k[i:k.find(">", i) + 1] + ">"
        elif j >= 0:
            k = "emmax_attention_resident_kernel<" + k[j:k.find(">", j) + 1] + ">"
        else:
            k = k[max(k.find("emmax_attention"), 0):][:48]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_vgpr"] = [float(r["VGPR_Count"])]
        agg[k]["_lds"] = [float(r["LDS_Block_Size"])]
out = {}
for k, v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8.0          # elapsed shader cycles
    d = dict(m)
    if cyc:
        d["elapsed_cycles"] = cyc
        d["mfma_busy_frac"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        d["valu_active_frac_of_wave_cycles"] = 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
    if "SQ_LDS_IDX_ACTIVE" in m:
        d["lds_conflict_share"] = m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1)
    if "FETCH_SIZE" in m:
        d["hbm_read_MB(2xFETCH_SIZE)"] = 2 * m["FETCH_SIZE"] * 1024 / 1e6
    if "WRITE_SIZE" in m:
        d["hbm_write_MB"] = m["WRITE_SIZE"] * 1024 / 1e6
    if "TCC_HIT_sum" in m:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1)
    out[k] = d
print(json.dumps(out, indent=1))
json.dump(out, open(root + "/summary.json", "w"), indent=1)
