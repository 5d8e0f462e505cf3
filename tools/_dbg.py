import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_fullsize as TF
from test_gpu_fullsize import *
Lb = TF._lib()
W = O.init_weights(V + 1, E, H, A, D, D, seed=11)
n_img, S = 128, 5
b = O.synthetic_batch(n_img, S, R, D, V, L, seed=77, ragged_regions=True)
batch = {k: v.cuda() for k, v in b.items()}
model = build_model(CFG, W, "bf16", drop=0.5)
model.train()
eng = model.engine
N, T = n_img * S, L + 1
t_run = model._steps_to_run(batch["labels"])
pd = {k: v.detach() for k, v in model.param_dict().items()}
td = torch.bfloat16
def run(rec):
    eng.recurrence = rec
    logp, ws, _ = eng.forward(pd, batch["fc_feats"], batch["att_feats"], batch["att_masks"], batch["labels"], t_run, True, 99)
    out = {n: eng.workspace_tensor(ws, n, shp, dt)[: (t_run + 1 if shp[0] == T + 1 else t_run)].float().clone() for n, shp, dt in NAMES(T, N, td)}
    torch.cuda.synchronize()
    eng.release(ws)
    return out
ref = run(Lb.REC_FWD_CHAIN)
g1, g1b, g2 = run(0), run(0), run(Lb.REC_SAFE)
for k in ("alpha", "ctx", "att_h", "h_att"):
    d12 = (g1[k] - g2[k]).abs()
    d1r = (g1[k] - ref[k]).abs()
    d2r = (g2[k] - ref[k]).abs()
    print(k, "fast-vs-safe max %.3e; fast-vs-chain %.3e; safe-vs-chain %.3e" % (d12.max().item(), d1r.max().item(), d2r.max().item()))
    if d12.max() > 0:
        tt, nn = torch.nonzero(d12.amax(2) > 0, as_tuple=True)
        print("   first differing (t, row, row%80, group):", [(int(a), int(b), int(b) % 80, int(b) // 80) for a, b in list(zip(tt, nn))[:12]], "count", len(tt))
# NaN check on alpha rows of shared rows
a = g1["alpha"]
print("alpha row sums fast: min %.6f max %.6f; safe: min %.6f max %.6f" % (a.sum(2).min().item(), a.sum(2).max().item(), g2["alpha"].sum(2).min().item(), g2["alpha"].sum(2).max().item()))
