#!/usr/bin/env python3
"""BASELINE.json configs[4] (patch 8^3 -> N = 4097 tokens per modality): ModelCross fwd+bwd step time with the bf16 attention
forward and with the MX-fp8 one (XVIT_ATTN_FP8=1), same process, same weights and inputs.  One JSON line per variant.

    python tools/long_seq_bench.py [batch=8] [steps=10]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
sys.path.insert(0, ROOT)
import xvit  # noqa: E402
from bench import base_config, flops_per_sample  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
cfg = base_config()
cfg.patch_size = (8, 8, 8)
fwd_f, both_f, P = flops_per_sample(cfg)
torch.manual_seed(0)
model = xvit.ModelCross(cfg).to(dev)
model.train()
img = torch.randn(B, cfg.num_modalities, 1, *cfg.img_size).to(dev, torch.bfloat16)
labels = torch.randint(0, 2, (B,)).to(dev)


def step():
    for p in model.parameters():
        p.grad = None
    xvit.invalidate_shadows()
    logits, loss = model(img, labels)
    loss.backward()
    return logits, loss


res = {}
for flag in ("0", "1", "0", "1"):
    os.environ["XVIT_ATTN_FP8"] = flag
    for _ in range(3):
        logits, loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        logits, loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    res.setdefault(flag, []).append((ms, logits.detach().float().cpu(), float(loss)))
    print(json.dumps({"metric": "patch_tokens_per_sec_fwd_bwd", "value": round(B * cfg.num_modalities * P / ms * 1e3, 1), "unit": "patch-tokens/s", "ms_per_step": round(ms, 3),
                      "config": {"workload": f"configs[4]: ModelCross d=768 H=12, 128^3 volume, 8^3 patches (N={P + 1}), batch {B}", "attention_forward": "mx-fp8" if flag == "1" else "bf16"},
                      "model_tflops": round(B * both_f / ms / 1e9, 1), "loss": round(float(loss), 5)}), flush=True)
lb, l8 = res["0"][-1][1], res["1"][-1][1]
print(f"logits rel-L2 fp8 vs bf16 forward attention: {float((l8 - lb).norm() / lb.norm()):.3e};  step {min(m for m, _, _ in res['1']):.2f} vs {min(m for m, _, _ in res['0']):.2f} ms")
