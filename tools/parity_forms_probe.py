import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_cpu as R
import xvit, xvit.functional as XF
dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm())
for name, batch in (("base", 2), ("small", 2)):
    cfg = R.make_config(name); sd = R.make_state_dict(cfg, seed=0); img, labels = R.make_inputs(cfg, batch, seed=0)
    ref_logits, _ = R.model_cross_forward(sd, img, labels, cfg)
    cap_ref = {}; R.model_cross_forward(sd, img, labels, cfg, capture=cap_ref)
    for emu_low in (True, False):
        saved = R._cls_cross_attention_lowrank_emulated
        if not emu_low:
            R._cls_cross_attention_lowrank_emulated = None
            orig = R.cls_cross_attention
        cap = {}
        with R.emulate_bf16():
            if not emu_low:
                # literal emulation: bypass the low-rank branch
                q0 = R._QUANT
                def lit(sd_, p, x, H, _orig=orig):
                    d = x.shape[-1]
                    q = R._split_heads(R.linear(x[:, 0:1], sd_[p + ".wq.weight"], sd_[p + ".wq.bias"], exact=True), H)
                    k = R._split_heads(R.linear(x, sd_[p + ".wk.weight"], sd_[p + ".wk.bias"], store=True), H)
                    v = R._split_heads(R.linear(x, sd_[p + ".wv.weight"], sd_[p + ".wv.bias"], store=True), H)
                    o, _ = R.softmax_attention(q, k, v, (d // H) ** -0.5)
                    return R.linear(R._merge_heads(o), sd_[p + ".proj.weight"], sd_[p + ".proj.bias"], exact=True)
                R.cls_cross_attention = lit
            emu_logits, _ = R.model_cross_forward(sd, img, labels, cfg, capture=cap)
        if not emu_low:
            R.cls_cross_attention = orig; R._cls_cross_attention_lowrank_emulated = saved
        for form in ("lowrank", "dense"):
            XF.XATTN_FORM = form
            model = xvit.ModelCross(cfg).to(dev); model.load_state_dict(sd); model.train()
            caps = {}
            hooks = [blk.register_forward_hook(lambda m, i, o, b=b: caps.__setitem__(b, [t.detach() for t in o])) for b, blk in enumerate(model.transformer)]
            logits, loss = model(img.to(dev), labels.to(dev))
            last = len(model.transformer) - 1
            cls_e = max(rel(caps[last][m][:, 0], cap[f"msb{last}"][m][:, 0]) for m in range(cfg.num_modalities))
            cls_r = max(rel(caps[last][m][:, 0], cap_ref[f"msb{last}"][m][:, 0]) for m in range(cfg.num_modalities))
            print(f"{name} gpu={form:8s} emu={'lowrank' if emu_low else 'literal'}: logits gpu-emu {rel(logits, emu_logits):.2e} gpu-fp32 {rel(logits, ref_logits):.2e} emu-fp32 {rel(emu_logits, ref_logits):.2e} | last-block CLS gpu-emu {cls_e:.2e} gpu-fp32 {cls_r:.2e}", flush=True)
