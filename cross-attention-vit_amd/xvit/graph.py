"""HIP-graph capture of one training step (zero-grad + weight cast + forward + loss + backward).

At the reference's batch size (8 volume pairs per GPU, main_mist.py:206) the ~480 kernel launches of a step cost more
host time (8 ms of Python + launch issue) than GPU time (~4 ms): replaying them as ONE graph launch removes the host
from the loop.  Every libxvit_hip entry point only enqueues on the stream it is given (no allocation, no
synchronisation), so the whole step is capturable; torch's caching allocator serves the step's tensors from the
graph's private pool.

    step = GraphedStep(model, img_example, labels_example)   # warm-up + capture
    logits, loss = step(img, labels)                         # copies into the static inputs, replays; p.grad are filled
    optimizer.step()                                         # outside the graph

Restrictions: static shapes.  Dropout (the reference trains with p = 0.1 .. 0.25, main_mist.py:71-77): the host-side seeds of the
capture are frozen into the graph, so the step registers a device counter (xvit_set_dropout_epoch), increments it at the head of the
graph, and every dropout kernel mixes its value into the seed at run time: each replay draws fresh masks, the same in its forward and
its backward.  The per-modality self-attention branches and the fusions fork from and join back into the capture stream, so the
graph keeps them as parallel paths (XVIT_GRAPH_STREAMS=branches forks the branches only, =0 captures on one stream).

The captured step runs the model on detached leaf ALIASES of its parameters (torch.func.functional_call; same storage, so
optimizer updates are seen) and takes the gradients with torch.autograd.grad, assigning them to the real p.grad.  Reason:
backward delivers a leaf's gradient through its AccumulateGrad node, which is bound to the stream it was created on and
stays alive as long as ANY autograd graph that used the parameter does.  If an earlier eager iteration's graph is still
referenced (a kept `loss` is enough), those nodes sit on the default stream, the captured backward would have to hand
gradients to a stream outside the capture, and hipStreamEndCapture crashes (torch only warns "AccumulateGrad node's
stream does not match", once per process).  Fresh aliases get fresh nodes, created inside the capture.
"""
from __future__ import annotations

import os

import torch

from . import functional as XF


def _has_dropout(model) -> bool:
    """Any active rate on the path (the module facades keep their rates in nn.Dropout children, like the reference)."""
    return any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in model.modules())


class GraphedStep:
    """reducer (xvit.ddp.BucketedGradReducer, optional): data-parallel runs.  The reducer's bucket views become the gradient buffers
    (weight-gradient kernels write into them, functional.GRAD_SINK) and its bucket all-reduces are captured INTO the graph, each
    forked onto the comm stream behind its last gradient and joined before the graph ends — the reference's setting (8 volume
    pairs per GPU under DDP, main_mist.py:206, 211-218) is exactly where only the graph removes the host bound."""

    def __init__(self, model, img, labels, warmup: int = 3, reducer=None):
        if not img.is_cuda:
            raise RuntimeError("GraphedStep needs GPU tensors")
        # dropout active anywhere in the step: a device-side epoch makes every replay draw new masks (module docstring)
        self._epoch = torch.zeros(1, dtype=torch.int64, device=img.device) if model.training and _has_dropout(model) else None
        if os.environ.get("XVIT_FANOUT", "1") != "1" and os.environ.get("XVIT_GRAPH_STREAMS", "1") == "1":
            # without the fan-out node a branch output read by two forked fusions has its gradients accumulated by the autograd
            # engine ACROSS the side streams: exactly the pattern hipStreamEndCapture crashes on (tools/graph_capture_probe.py)
            raise RuntimeError("GraphedStep: XVIT_FANOUT=0 cannot be captured with forked fusions; unset it, or set XVIT_GRAPH_STREAMS=branches (or 0)")
        self.model = model
        if hasattr(model, "_sync_flat_weights") and next(model.parameters()).is_cuda and os.environ.get("XVIT_FLAT_WEIGHTS", "1") != "0":
            model._sync_flat_weights()             # parameters move into the flat buffer at the first forward: alias them afterwards
        self._named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.params = [p for _, p in self._named]
        self._alias = {n: p.detach().requires_grad_() for n, p in self._named}
        self.reducer = reducer
        self._sink = reducer.grad_sink(model) if reducer is not None else None
        if reducer is not None:
            with torch.no_grad():
                for b in reducer.buckets:          # a parameter the step never reaches keeps reducing these zeros (eager mode: zeros_like)
                    b.flat.zero_()
        self.img = img.clone()
        self.labels = labels.clone()
        # XVIT_GRAPH_STREAMS: "1" (default) forks the branches AND the fusions inside the capture (parallel graph paths),
        # "branches" only the self-attention branches, "0" captures everything on one stream.  Set as a context variable
        # (cross_vit.STREAM_MODE), not through os.environ: another thread's model keeps its own mode.
        from .cross_vit import STREAM_MODE
        form = XF.XATTN_FORM
        if form == "auto":
            XF.XATTN_FORM = "lowrank"              # warm up the kernels the capture will run (auto picks them only while capturing)
        self._mode_token = STREAM_MODE.set({"0": "0", "branches": "branches"}.get(os.environ.get("XVIT_GRAPH_STREAMS", "1"), "1"))
        try:
            side = torch.cuda.Stream(device=img.device)
            side.wait_stream(torch.cuda.current_stream(img.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager()
            torch.cuda.current_stream(img.device).wait_stream(side)
            torch.cuda.synchronize(img.device)
            self.graph = torch.cuda.CUDAGraph()
            for p in self.params:
                p.grad = None                      # gradients are (re)allocated from the graph's private pool
            XF.arena_reset()                       # no zero arena crosses the capture boundary in either direction (functional._zeros)
            try:
                with torch.cuda.graph(self.graph):
                    self.logits, self.loss = self._eager(zero=False)
            finally:
                XF.arena_reset()
                XF.release_capture_keep()         # tensors that crossed streams were kept alive up to here (functional.keep)
            if reducer is not None:
                for p in self.params:              # from now on p.grad IS the bucket view the graph's collectives reduce in place
                    p.grad = reducer._view_of[id(p)]
            self._grads = [p.grad for p in self.params]   # static buffers of the graph: handed back at every call (zero_grad(set_to_none=True) drops them)
            # the graph holds raw addresses: remember where the parameters (and the flat weight buffers) live
            self._ptrs = [p.data_ptr() for p in self.params]
            self._flat = getattr(model, "_flat", None)
        finally:
            XF.XATTN_FORM = form
            self._restore_env()                    # also when warm-up or capture raises: never leave the process in capture mode

    def _restore_env(self):
        from .cross_vit import STREAM_MODE
        if self._mode_token is not None:
            STREAM_MODE.reset(self._mode_token)
            self._mode_token = None

    def _eager(self, zero=True):
        if zero:
            for p in self.params:
                p.grad = None
        XF.SHADOWS.force = True                # the captured step always re-casts the weights (they change every step)
        if self._epoch is None:
            return self._step()
        from . import ops
        ops.set_dropout_epoch(self._epoch)     # process-wide while this step is being issued (warm-up and capture); off again below
        try:
            self._epoch.add_(1)                # first node of the graph: a new epoch per replay
            return self._step()
        finally:
            ops.set_dropout_epoch(None)

    def _step(self):
        red = self.reducer
        if red is None:
            logits, loss = torch.func.functional_call(self.model, self._alias, (self.img, self.labels))
            grads = torch.autograd.grad(loss, [self._alias[n] for n, _ in self._named], allow_unused=True)   # see the module docstring
            for p, g in zip(self.params, grads):
                p.grad = g
            return logits.detach(), loss.detach()
        prev_sink, XF.GRAD_SINK = XF.GRAD_SINK, self._sink
        handles = red.attach_leaves([(p, self._alias[n]) for n, p in self._named])
        red.set_graph_mode(True)
        try:
            logits, loss = torch.func.functional_call(self.model, self._alias, (self.img, self.labels))
            torch.autograd.grad(loss, [self._alias[n] for n, _ in self._named], allow_unused=True)   # the hooks move / count the gradients
            red.finish()                       # unused parameters' buckets, then the capture stream waits for every collective
        finally:
            red.set_graph_mode(False)
            XF.GRAD_SINK = prev_sink
            for h in handles:
                h.remove()
        return logits.detach(), loss.detach()

    def __call__(self, img=None, labels=None):
        if any(p.data_ptr() != q for p, q in zip(self.params, self._ptrs)) or getattr(self.model, "_flat", None) is not self._flat:
            raise RuntimeError("GraphedStep: the model's parameter storage changed after capture (model.to(...), a re-built flat weight buffer): "
                               "the captured graph would keep training the old buffers; build a new GraphedStep")
        if img is not None:
            self.img.copy_(img, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        self.graph.replay()
        for p, g in zip(self.params, self._grads):
            p.grad = g
        return self.logits, self.loss
