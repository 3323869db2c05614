"""Gaps between consecutive kernels of the decode steps, from a rocprofv3 --kernel-trace CSV (not part of the product): for the steady region
of a bench.py run -- the launches between two consecutive lm-head launches are one decode step -- the sum of the kernel durations, the sum of
the gaps (start of a kernel minus end of the one before it) per step, and both per launch.  usage: kernel_gaps.py <kernel_trace.csv> [label]
RESULT (round 5, profiles/r05_kernel_gaps.txt): on this stack the trace's stamps of back-to-back dependent launches are CONTIGUOUS -- every gap is
0.00 us at batch 1 / 8 / 32, eager and graph replay: the time between two kernels of a stream (the guide's ~1.2-1.9 us boundary) is folded into the
durations rocprofv3 reports, so the per-kernel averages of profiles/*kernel_stats.csv INCLUDE their boundary; in-kernel phase stamps
(tools/decode_stage_trace.py) are what separates ramp from stream."""
import csv
import statistics as st
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
label = sys.argv[2] if len(sys.argv) > 2 else ""
# a decode step ends with emmax_decode_finish_kernel; keep steps made of emmax_decode* launches only
steps, cur = [], []
for s, e, n in rows:
    if "emmax_decode" not in n and "emmax_gemv" not in n:
        cur = []
        continue
    cur.append((s, e, n))
    if "finish" in n:
        if len(cur) > 100:
            steps.append(cur)
        cur = []
if not steps:
    raise SystemExit("no decode steps found")
steps = steps[len(steps) // 4:]          # steady region
dur = [sum(e - s for s, e, _ in stp) / 1e3 for stp in steps]
gap = [sum(stp[i + 1][0] - stp[i][1] for i in range(len(stp) - 1)) / 1e3 for stp in steps]
span = [(stp[-1][1] - stp[0][0]) / 1e3 for stp in steps]
n = len(steps[0])
gaps_all = [(stp[i + 1][0] - stp[i][1]) / 1e3 for stp in steps for i in range(len(stp) - 1)]
print(f"{label}: {len(steps)} steps of {n} launches: span {st.median(span):.1f} us per step = kernels {st.median(dur):.1f} + gaps {st.median(gap):.1f} "
      f"({st.median(gap) / st.median(span) * 100:.1f} %); per launch: kernel {st.median(dur) / n:.2f} us, gap median {st.median(gaps_all):.2f} / p10 {sorted(gaps_all)[len(gaps_all) // 10]:.2f} / "
      f"p90 {sorted(gaps_all)[len(gaps_all) * 9 // 10]:.2f} us")
# per kernel family: mean duration and mean gap BEFORE it
fam = {}
for stp in steps:
    for i, (s, e, nm) in enumerate(stp):
        key = nm.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
        d = fam.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        d[2] += (s - stp[i - 1][1]) / 1e3 if i else 0.0
for k, (c, d, g) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"   {k:70s} x{c / len(steps):5.1f} per step  {d / c:7.2f} us  gap before {g / c:5.2f} us")
