cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python tools/attn_probe.py 2>&1 | grep -v amdgpu
python tools/stage_bench.py 2>/dev/null | tail -1 > $O/stage_bench.json; cat $O/stage_bench.json
bash tools/gpucmd_attn_pmc.sh r02m/attn_pmc > /dev/null 2>&1; python tools/pmc_attn_summary.py $O/attn_pmc | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k, {x: round(v[x],4) for x in ('mfma_busy_frac','lds_conflict_share','hbm_read_MB(2xFETCH_SIZE)','hbm_write_MB','l2_hit_rate','SQ_INSTS_VALU','SQ_INSTS_MFMA') if x in v})
"
