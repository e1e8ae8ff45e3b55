"""Fused Adam for the drop-in models: same update as the reference's
`torch.optim.Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)` (model_cross.py:277), one HIP launch
per parameter group instead of a foreach chain, and the bf16 GEMM-operand copy of every flat-stored weight is written
by the same kernel (so `FlatWeights.refresh` has nothing to re-cast after a step).

    opt = xvit.optim.FusedAdam(model.parameters(), lr=1e-4, weight_decay=0.0)
    ... loss.backward(); opt.step(); opt.zero_grad()
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from . import functional as XF

CHUNK = 16384


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @staticmethod
    def _shadow_of(p):
        for grp in XF.SHADOWS.groups:
            i = grp.index.get(id(p))
            if i is not None and grp.params[i] is p and grp.intact():
                return grp, grp.view16[i]
        return None, None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        touched, unshadowed = set(), []
        for group in self.param_groups:
            buckets, keep = {}, []          # step count -> (rows, chunks): torch keeps the step per parameter
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("xvit FusedAdam: parameters and gradients must be dense fp32 tensors on the GPU")
                if not p.is_contiguous():
                    raise RuntimeError("xvit FusedAdam: non-contiguous parameter")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                grp, sh = self._shadow_of(p)
                if grp is not None:
                    touched.add(id(grp))
                else:
                    unshadowed.append(p)
                keep.append(g)
                rows, chunks = buckets.setdefault(st["step"], ([], []))
                t = len(rows)
                rows.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             sh.data_ptr() if sh is not None else 0, p.numel()))
                chunks.extend((t, c) for c in range((p.numel() + CHUNK - 1) // CHUNK))
            b1, b2 = group["betas"]
            for step, (rows, chunks) in buckets.items():
                dev = group["params"][0].device
                table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev, non_blocking=True)
                chunk_t = torch.from_numpy(np.asarray(chunks, dtype=np.int32)).to(dev, non_blocking=True)
                _lib.check(lib.xvit_adam_step(table.data_ptr(), chunk_t.data_ptr(), len(chunks), float(group["lr"]), b1, b2, group["eps"],
                                              group["weight_decay"], step, 1.0, torch.cuda.current_stream().cuda_stream), "xvit_adam_step")
                table.record_stream(torch.cuda.current_stream()); chunk_t.record_stream(torch.cuda.current_stream())
        # the kernel rewrote p in place through raw pointers (no version bump) AND its bf16 copy: mark the flat groups fresh
        for grp in XF.SHADOWS.groups:
            if id(grp) in touched:
                grp.stamp = sum(q._version for q in grp.params)
        # ... and every other parameter (ModelVIT, the model.py Encoder, stand-alone modules, XVIT_FLAT_WEIGHTS=0) only in place:
        # their per-parameter bf16 copies are keyed by the version counter, which a raw-pointer write does not move
        XF.SHADOWS.drop(unshadowed)
        return loss
