#!/usr/bin/env python3
"""Which stream-fork pattern survives HIP-graph capture of the training step?  Each variant runs in its own child process
(a crash in hipStreamEndCapture must not take the probe down).  GPU box only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "cross-attention-vit_amd")); sys.path.insert(0, os.path.join(%r, "oracle"))
import ref_cpu as R, xvit
from xvit.graph import GraphedStep
cfg = R.make_config("tiny")
model = xvit.ModelCross(cfg).cuda(); model.train()
img, lab = R.make_inputs(cfg, 4, seed=0)
g = GraphedStep(model, img.cuda(), lab.cuda())
l1 = float(g()[1]); l2 = float(g()[1]); torch.cuda.synchronize()
print("CAPTURE_OK", l1, l2)
''' % (ROOT, ROOT)

VARIANTS = {
    "branches only": {"XVIT_GRAPH_STREAMS": "branches"},
    "branches + fusions, engine sums grads across streams": {"XVIT_GRAPH_STREAMS": "1", "XVIT_FANOUT": "0"},
    "branches + fusions, explicit fan-out node": {"XVIT_GRAPH_STREAMS": "1", "XVIT_FANOUT": "1"},
    "one stream": {"XVIT_GRAPH_STREAMS": "0"},
}
for name, env in VARIANTS.items():
    r = subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("CAPTURE_OK")]
    print(f"{name:55s} rc={r.returncode:4d} {ok[0] if ok else r.stderr.strip().splitlines()[0][:120] if r.stderr.strip() else ''}", flush=True)
