"""Print scratch / SGPR / VGPR / spill figures of every kernel in a hipcc `-save-temps` .s file (gfx950 metadata block)."""
import re
import sys

s = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in s.split("  - .agpr_count:")[1:]:
    def f(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = f("name")
    if filt and filt not in name:
        continue
    print(f"{name[:90]:90s} scratch {f('private_segment_fixed_size'):>4s} sgpr {f('sgpr_count'):>3s} vgpr {f('vgpr_count'):>3s} spill {f('vgpr_spill_count'):>3s} lds {f('group_segment_fixed_size')}")
