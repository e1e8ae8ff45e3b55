#!/usr/bin/env python3
"""Where the HOST time of an eager step goes at the reference's batch (8 volume pairs, configs[1] shape): cProfile over N steps, top entries
by own time.  At this batch the step is bound by Python + launch issue (8-9 ms) rather than by the GPU (4.4 ms as one HIP graph).
    python tools/host_profile.py [batch=8] [steps=30]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-attention-vit_amd"))
sys.path.insert(0, ROOT)
import xvit  # noqa: E402
from bench import base_config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
cfg = base_config()
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


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.2f} ms per eager step (batch {B})")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

# ---- the backward runs on the autograd engine's thread, which cProfile above does not see: wall time inside every Function.backward
# (launches are asynchronous, so this is host time) and a cProfile of one block's backward called directly
import collections  # noqa: E402
import xvit.functional as XF  # noqa: E402
from xvit import cross_vit as CV  # noqa: E402

acc = collections.Counter()
cnt = collections.Counter()


def timed(cls):
    orig = cls.backward

    def wrapper(ctx, *a):
        t = time.perf_counter()
        out = orig(ctx, *a)
        acc[cls.__name__] += time.perf_counter() - t
        cnt[cls.__name__] += 1
        return out
    cls.backward = staticmethod(wrapper)


for c in (XF.SelfAttentionBlockFn, XF.CrossFusionFn, XF.PatchEmbedFn, XF.HeadFn, XF.MeanCrossEntropyFn, CV._FanOut):
    timed(c)
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("host time inside Function.backward, ms per step:")
for k, v in acc.most_common():
    print(f"  {k:28s} {v / steps * 1e3:7.3f} ms  ({cnt[k] // steps} calls, {v / cnt[k] * 1e6:6.1f} us each)")
print(f"  total {sum(acc.values()) / steps * 1e3:.3f} ms")
