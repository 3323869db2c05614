#!/bin/bash
# builds the stamp-enabled library and (on the GPU box) runs tools/decode_stage_trace.py with it; usage: tools/decode_stage_trace.sh build | run [B]
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/bin/trace_build
  F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable -Wno-inline-asm -DDECODE_LAB_TRACE -Iemma-x_amd/csrc -Iinclude"
  for f in gemm norm attention misc decode decode_ks decode_km decode_kmp decode_mfma model; do /opt/rocm/bin/hipcc $F -c emma-x_amd/csrc/$f.hip -o tools/bin/trace_build/$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libemmax_hip_trace.so tools/bin/trace_build/*.o
else
  cp tools/bin/libemmax_hip_trace.so emma-x_amd/emmax/libemmax_hip.so
  python tools/decode_stage_trace.py ${2:-1}
fi
