cd $GRAFT_REPO_ROOT
for v in 0 2 4 5 6; do
  echo "== variant $v"; EMMAX_ATTN_VARIANT=$v python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
  EMMAX_ATTN_VARIANT=$v python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention" 2>&1 | tail -1
done
