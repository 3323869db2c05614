#!/bin/bash
# builds tools/bin/attn_lab[_<variant>] : plain + the knock-out variants named on the command line (e.g. EXP SOFTMAX DMA)
cd "$(dirname "$0")/.." && mkdir -p tools/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -Wno-unused-variable -Wno-unused-function -Iemma-x_amd/csrc -Iinclude"
/opt/rocm/bin/hipcc $F tools/attn_lab.hip -o tools/bin/attn_lab &
for v in "$@"; do
  D=""; for k in ${v//+/ }; do D="$D -DATTN_LAB_KO_$k"; done
  /opt/rocm/bin/hipcc $F $D tools/attn_lab.hip -o tools/bin/attn_lab_$v &
done
wait
