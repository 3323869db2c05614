import os, sys
sys.path[:0]=['.','emma-x_amd','tests']
import numpy as np, torch
from emmax.config import EmmaXConfig
from emmax.modeling import EmmaXForActionPrediction
from test_fullsize_gpu import _rows
cfg = EmmaXConfig.emma_x_7b()
model = EmmaXForActionPrediction.from_synthetic(cfg, seed=0, device="cuda:0", planted=True, max_batch=4, max_prompt=512, max_ctx=256 + 512 + 64 + 1)
frames, rows = _rows(cfg, [300, 64, 128, 33], [20, 33, 9, 28], seed=17)
fr = frames.to("cuda:0")
os.environ["EMMAX_GRAPH"]="0"
_, ids_e, lens_e = model.generate_actions_batch(fr, rows, max_new_tokens=40)
def forced(sel, T=24):
    model._prefill([rows[i] for i in sel], None, fr[sel].contiguous(), max_new=41)
    out=[]
    for t in range(T):
        out.append(model.engine.last_logits().float().cpu())
        model.engine.set_current_tokens([int(ids_e[b, min(t, int(lens_e[b]) - 1)]) for b in sel])
        model.engine.decode_step()
    return out
a = forced([0,1,2,3])
a2 = forced([0,1,2,3])
print("run-to-run identical:", all(torch.equal(x,y) for x,y in zip(a,a2)))
os.environ["EMMAX_STREAMK"]="0"
model.engine.new_session(4, 512, 256+512+64+1)
w = forced([0,1,2,3])
os.environ.pop("EMMAX_STREAMK")
model.engine.new_session(4, 512, 256+512+64+1)
singles = [forced([b]) for b in range(4)]
for t in range(24):
    mx = a[t].abs().amax(dim=1)
    d_aw = ((a[t]-w[t]).abs().amax(dim=1)/mx).tolist()
    d_a1 = [float(((a[t][b]-singles[b][t][0]).abs().max()/mx[b])) for b in range(4)]
    d_w1 = [float(((w[t][b]-singles[b][t][0]).abs().max()/mx[b])) for b in range(4)]
    print(t, "sk-vs-whole", ["%.4f"%x for x in d_aw], "sk-vs-bs1", ["%.4f"%x for x in d_a1], "whole-vs-bs1", ["%.4f"%x for x in d_w1])
