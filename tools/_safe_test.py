import sys, os
sys.path.insert(0, "/root/repo"); sys.argv=[sys.argv[0]]
import torch
from bench import CFG, make_opt
from unpaired_image_captioning_amd import _lib as L, models
from unpaired_image_captioning_amd.synthetic import synthetic_batch
lib = L.load(); c = CFG
torch.manual_seed(1234)
model = models.setup(make_opt("bf16", 1234)).cuda(); model.train(True); eng = model.engine
batch = synthetic_batch(128, c["S"], c["R"], c["D"], c["V"], c["L"], seed=1234)
t_run = model._steps_to_run(batch["labels"]); params = model.param_dict()
N, T, H = 640, 17, 512
def run(mode):
    L.check(lib.uic_set_persistent_rnn(mode))
    logp, ws, _ = eng.forward(params, batch["fc_feats"], batch["att_feats"], None, batch["labels"], t_run, 1, 77, want_logprobs=True)
    out = {}
    for n, shp, dt in (("h_att", (T + 1, N, H), torch.bfloat16), ("att_h", (T, N, H), torch.float32), ("ctx", (T, N, H), torch.bfloat16), ("h_lang", (T + 1, N, H), torch.bfloat16), ("c_att", (T+1, N, H), torch.float32)):
        out[n] = eng.workspace_tensor(ws, n, shp, dt).float().clone()
    torch.cuda.synchronize(); eng.release(ws); return out
ref = run(0)
for mode in (1, 2):
    g = run(mode)
    for k in g:
        d = (g[k] - ref[k]).abs()
        bad = torch.isnan(g[k]) | (d > 0.05)
        if bad.any():
            idx = bad.nonzero()
            t0 = int(idx[:, 0].min())
            sub = idx[idx[:, 0] == t0]
            rows = sorted(set(sub[:, 1].tolist())); cols = sorted(set(sub[:, 2].tolist()))
            print(mode, k, "first bad step", t0, "count", len(sub), "rows", rows[:12], "...", rows[-3:], "n_rows", len(rows), "cols", cols[:8], "...", cols[-3:], "n_cols", len(cols), "nan", int(torch.isnan(g[k][t0]).sum()))
        else:
            print(mode, k, "ok", float(d.max()))
g = run(2)
b = (torch.isnan(g["h_lang"][1]) | ((g["h_lang"][1] - ref["h_lang"][1]).abs() > 0.02)).float()   # [N, H]
bt = b.view(8, 80, 512)
print("bad frac per group:", [round(float(x), 2) for x in bt.mean(dim=(1, 2))])
print("bad frac per row tile:", [round(float(bt[:, 16*i:16*i+16].mean()), 2) for i in range(5)])
print("bad frac per unit%16:", [round(float(b.view(640, 32, 16)[:, :, j].mean()), 2) for j in range(16)])
print("bad frac per rank (unit//16) first 32:", [round(float(b.view(640, 32, 16)[:, j, :].mean()), 2) for j in range(32)])
print("bad frac per row%4:", [round(float(b.view(160, 4, 512)[:, j].mean()), 2) for j in range(4)])
print("bad frac per (row%16)//4:", [round(float(b.view(40, 4, 4, 512)[:, j].mean()), 2) for j in range(4)])
