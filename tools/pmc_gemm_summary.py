"""Summarise a rocprofv3 --pmc pass over tools/gemm_bench.py: per GEMM launch shape, MFMA pipe utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (elapsed shader cycles x 1024 SIMDs).  SQ_VALU_MFMA_BUSY_CYCLES counts 16 cycles per
v_mfma_f32_16x16x32_bf16 summed over the chip; GRBM_GUI_ACTIVE is summed over the 8 XCDs (elapsed cycles = value / 8)."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][r["Kernel_Name"].find("emmax_"):][:110], r["Grid_Size"], r["LDS_Block_Size"], r["Dispatch_Id"])
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
by_shape = collections.defaultdict(list)
for (name, grid, lds, _), c in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    busy, cyc = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(c["GRBM_GUI_ACTIVE"]) / 8.0
    by_shape[(name, grid, lds)].append((busy, cyc, sum(c.get("SQ_INSTS_VALU", [0])), sum(c.get("SQ_INSTS_MFMA", [0])) or busy / 16.0 / 64.0))
out = {}
for (name, grid, lds), v in sorted(by_shape.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    busy = sum(x[0] for x in v) / len(v)
    cyc = sum(x[1] for x in v) / len(v)
    out[f"{name} grid {grid} lds {lds}"] = {"launches": len(v), "mfma_busy_cycles": busy, "elapsed_shader_cycles": cyc,
                                            "mfma_util": round(busy / (cyc * 1024.0), 4),
                                            "valu_insts_per_mfma_inst": round(sum(x[2] for x in v) / max(sum(x[3] for x in v), 1.0), 2)}
if len(sys.argv) > 2:   # label: nest under the shape the pass ran
    out = {sys.argv[2]: out}
print(json.dumps(out, indent=1))
