#!/usr/bin/env python3
"""ModelCross fwd+bwd step time at any of BASELINE.json's shapes — `base` (configs[1]), `ucsf` (configs[2]: 4 modalities of 240^3,
N = 3376) or `long` (configs[4]: 8^3 patches, N = 4097) — or at the reference's own run shape `mist` (config2.py + main_mist.py:71: d = 1024,
16 heads, three 128 x 128 x 64 modalities in a ring, 16 x 16 x 8 patches; `drop=0.25` for its dropout rate), eagerly and, with `graph`, as
one HIP graph.  One JSON line per variant.

    python tools/config_step_bench.py <base|ucsf|long|mist> [batch] [steps] [graph] [drop=P]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "base"
nums = [int(a) for a in sys.argv[2:] if a.isdigit()]
B = nums[0] if nums else {"base": 126, "ucsf": 2, "long": 8, "mist": 8}[name]
drop = ([float(a[5:]) for a in sys.argv if a.startswith("drop=")] + [0.0])[0]
steps = nums[1] if len(nums) > 1 else 10
dev = torch.device("cuda:0")
cfg = base_config(dropout=drop)
if name == "mist":
    cfg.hidden_dim, cfg.mlp_dim, cfg.num_heads, cfg.img_size, cfg.patch_size = 1024, 4096, 16, (128, 128, 64), (16, 16, 8)
    cfg.num_modalities, cfg.attn_order = 3, {"0": "1", "1": "2", "2": "0"}
elif name == "ucsf":
    cfg.img_size, cfg.num_modalities, cfg.attn_order = (240, 240, 240), 4, {"0": "1", "1": "2", "2": "3", "3": "0"}
elif name == "long":
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


def report(kind, fn):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"metric": "patch_tokens_per_sec_fwd_bwd", "value": round(B * cfg.num_modalities * P / ms * 1e3, 1), "unit": "patch-tokens/s", "ms_per_step": round(ms, 3),
                      "config": {"workload": f"{name}: ModelCross d={cfg.hidden_dim} H={cfg.num_heads}, {cfg.num_modalities} x {'x'.join(map(str, cfg.img_size))}, {'x'.join(map(str, cfg.patch_size))} patches (N={P + 1}), batch {B}, dropout {drop}", "launch": kind},
                      "model_tflops": round(B * both_f / ms / 1e9, 1), "loss": round(float(out[1].detach()), 5), "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 1e9, 2)}), flush=True)


report("eager", step)
if "graph" in sys.argv:
    from xvit.graph import GraphedStep
    g = GraphedStep(model, img, labels)
    report("hip graph", lambda: g(img, labels))
