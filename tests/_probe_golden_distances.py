"""Diagnostic (not a test): every distance `_check_model_vs_golden` gates, printed instead of asserted, plus the bf16-emulating oracle's own
distance to the same fixture.    python tests/_probe_golden_distances.py <name> <batch>"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..", "cross-attention-vit_amd")]
import ref_cpu as R  # noqa: E402
from _util import rel  # noqa: E402
from test_modules_gpu import _run_model, _t  # noqa: E402

name, batch = sys.argv[1], int(sys.argv[2])
g = np.load(os.path.join(HERE, "golden", f"model_cross_{name}.npz"))
cfg, sd, img, labels, model, caps, logits, loss = _run_model(name, batch)
print("logits gpu", logits.detach().cpu().tolist(), "golden", g["logits"].tolist())
print(f"logits vs fp32 golden {rel(logits, _t(g['logits'])):.3e}   loss diff {abs(float(loss.detach()) - float(g['loss'])):.3e}")
cap = {}
with R.emulate_bf16():
    el, _ = R.model_cross_forward(sd, img, labels, cfg, capture=cap)
print(f"emulating oracle vs fp32 golden: logits {rel(el, torch.from_numpy(g['logits'])):.3e};  gpu vs emulating oracle {rel(logits, el):.3e}")
for b in range(cfg.num_multi_blocks):
    for m in range(cfg.num_modalities):
        t = caps[b][m]
        rows = _t(g[f"msb{b}/mod{m}/rows_idx"])
        print(f"msb{b} mod{m}: cls {rel(t[:, 0], _t(g[f'msb{b}/mod{m}/cls'])):.3e} (emu vs golden {rel(cap[f'msb{b}'][m][:, 0], torch.from_numpy(g[f'msb{b}/mod{m}/cls'])):.3e}, gpu vs emu {rel(t[:, 0], cap[f'msb{b}'][m][:, 0]):.3e})"
              f"  rownorm {rel(t.norm(dim=-1), _t(g[f'msb{b}/mod{m}/rownorm'])):.3e}  rows {rel(t[:, rows.to(t.device)], _t(g[f'msb{b}/mod{m}/rows'])):.3e}  all vs emu {rel(t, cap[f'msb{b}'][m]):.3e}")
worst_n, worst_s = (0, ""), (0, "")
for i, (k, p) in enumerate(sorted(model.named_parameters())):
    ref_n = float(g[f"gnorm/{k}"])
    if k.endswith("wk.bias"):
        continue
    dn = abs(float(p.grad.double().norm()) - ref_n) / (ref_n + 1e-30)
    idx = R.sample_idx(p.numel(), 16, 7919 + i)
    got = p.grad.reshape(-1)[idx.to(p.device)].cpu().double()
    ref = torch.from_numpy(g[f"gsamp/{k}"]).double()
    ds = float((got - ref).norm()) / (float(ref.norm()) + 1e-30)
    worst_n, worst_s = max(worst_n, (dn, k)), max(worst_s, (ds, k))
print("worst grad-norm deviation", worst_n, " worst 16-sample deviation", worst_s)
