import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_cpu as R
import xvit
dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())
for batch in (2, 6):
    cfg = R.make_config("base"); sd = R.make_state_dict(cfg, seed=0); img, labels = R.make_inputs(cfg, batch, seed=0)
    sd_r = {k: R.bf16_round(v) for k, v in sd.items()}; img_r = R.bf16_round(img)
    cap = {}
    ref_logits, ref_loss = R.model_cross_forward(sd_r, img_r, labels, cfg, capture=cap)
    cap_e = {}
    with R.emulate_bf16():
        emu_logits, _ = R.model_cross_forward(sd_r, img_r, labels, cfg, capture=cap_e)
    model = xvit.ModelCross(cfg).to(dev); model.load_state_dict(sd_r); model.train()
    caps = {}
    hooks = [blk.register_forward_hook(lambda m, i, o, b=b: caps.__setitem__(b, [t.detach() for t in o])) for b, blk in enumerate(model.transformer)]
    logits, loss = model(img_r.to(dev), labels.to(dev))
    for b in range(cfg.num_multi_blocks):
        for m in range(cfg.num_modalities):
            print(f"B={batch} msb{b} mod{m}: all tokens gpu-fp32q {rel(caps[b][m], cap[f'msb{b}'][m]):.2e}  CLS rows {rel(caps[b][m][:, 0], cap[f'msb{b}'][m][:, 0]):.2e} | gpu-emu {rel(caps[b][m], cap_e[f'msb{b}'][m]):.2e} CLS {rel(caps[b][m][:, 0], cap_e[f'msb{b}'][m][:, 0]):.2e}")
    print(f"B={batch} logits gpu-fp32q {rel(logits, ref_logits):.2e}  gpu-emu {rel(logits, emu_logits):.2e} emu-fp32q {rel(emu_logits, ref_logits):.2e} loss diff {abs(float(loss) - float(ref_loss)):.2e}", flush=True)
