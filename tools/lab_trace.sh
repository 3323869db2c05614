#!/bin/bash
# on the GPU box: run tools/ks_trace.py with the stamp-enabled library (tools/decode_stage_trace.sh build made it here)
cd "$(dirname "$0")/.."
cp emma-x_amd/emmax/libemmax_hip.so /tmp/libemmax_hip_product.so
cp tools/bin/libemmax_hip_trace.so emma-x_amd/emmax/libemmax_hip.so
timeout 600 python tools/${2:-ks_trace}.py ${1:-1}
cp /tmp/libemmax_hip_product.so emma-x_amd/emmax/libemmax_hip.so
