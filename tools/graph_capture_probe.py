#!/usr/bin/env python3
"""Which stream-fork pattern survives HIP-graph capture of the training step?  Each variant runs in its own child process
(a crash in hipStreamEndCapture must not take the probe down).  GPU box only.

    python tools/graph_capture_probe.py [config] [batch] [eager steps before the capture]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "cross-attention-vit_amd")); sys.path.insert(0, os.path.join(%r, "oracle"))
import ref_cpu as R, xvit
from xvit.graph import GraphedStep
name, B, pre = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = R.make_config(name)
torch.manual_seed(0)
model = xvit.ModelCross(cfg).cuda(); model.train()
img, lab = R.make_inputs(cfg, B, seed=0)
img, lab = img.cuda().bfloat16(), lab.cuda()
for _ in range(pre):                       # eager history on the default stream (what bench.py has behind it)
    for p in model.parameters(): p.grad = None
    model(img, lab)[1].backward()
torch.cuda.synchronize()
g = GraphedStep(model, img, lab)
l1 = float(g()[1]); l2 = float(g()[1]); torch.cuda.synchronize()
print("CAPTURE_OK", l1, l2)
''' % (ROOT, ROOT)

VARIANTS = {
    "branches only": {"XVIT_GRAPH_STREAMS": "branches"},
    "branches + fusions, engine sums grads across streams": {"XVIT_GRAPH_STREAMS": "1", "XVIT_FANOUT": "0"},
    "branches + fusions, explicit fan-out node": {"XVIT_GRAPH_STREAMS": "1", "XVIT_FANOUT": "1"},
    "one stream": {"XVIT_GRAPH_STREAMS": "0"},
}
args = (sys.argv[1:] + ["tiny", "4", "0"])[:3] if len(sys.argv) < 4 else sys.argv[1:4]
print("config", args)
for name, env in VARIANTS.items():
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", CHILD, *args], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("CAPTURE_OK")]
    err = [ln for ln in r.stderr.splitlines() if "Error" in ln or "error" in ln]
    print(f"{name:55s} rc={r.returncode:4d} {ok[0] if ok else (err[0][:150] if err else '')}", flush=True)
