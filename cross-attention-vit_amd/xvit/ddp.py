"""Data-parallel gradient reduction for the hot path: one process per GPU, batch sharded across
ranks, ONE exchange step per iteration — the mean of all parameter gradients (SURVEY.md §8(e);
the reference gets this implicitly from Lightning's DDP, main_mist.py:211-218).

Design for xGMI (point-to-point links, per-link-bound rings): few, large buckets (32 MiB fp32 by
default -> 12 collectives for the 93 M-parameter config) filled in the order gradients become
ready during backward (reverse registration order: heads -> last MultiScaleBlock -> ... -> the
shared patch embedding last).  When the last gradient of a bucket has been accumulated its
all-reduce is issued immediately on a SIDE stream behind an event, so RCCL traffic overlaps the
rest of backward; `finish()` makes the compute stream wait for the tail.  After `finish()` every
`p.grad` is a view into its bucket (no copy back).

No pack pass for the weight matrices: `grad_sink()` maps every 2-D parameter (by the address of the fp32 master and of
its bf16 operand copy) to its bucket view, and with `xvit.functional.GRAD_SINK` set to it the weight-gradient kernels
write straight into the buckets (the gradient autograd hands on IS the view); the 1/world of the mean rides on the
collective itself (ReduceOp.AVG on RCCL).  What is left to copy are the 1-D gradients (biases, norms: 0.3 % of the bytes).

With a captured step (`xvit.graph.GraphedStep(model, img, labels, reducer=...)`) the same logic runs from TENSOR hooks on
the step's leaf aliases while the graph is being captured, so the bucket all-reduces become nodes of the graph, forked
onto the comm stream behind the events of their last gradients and joined before the graph ends: a replay launches the
whole step, collectives included, and they overlap the rest of the captured backward exactly as in the eager form.

The class is device-agnostic (CPU tensors + gloo skip the stream logic), which is what the
world_size-2 tests in tests/test_ddp_gloo.py run.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "flat", "views", "pending", "work", "launched", "events")

    def __init__(self, params, device):
        self.params = params
        slot = lambda p: (p.numel() + 63) // 64 * 64          # noqa: E731 — every view starts on a 256-byte boundary: kernels write into them (16-byte stores)
        self.flat = torch.zeros(sum(slot(p) for p in params), dtype=torch.float32, device=device)
        self.views, o = [], 0
        for p in params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += slot(p)
        self.pending, self.work, self.launched = len(params), None, False
        self.events = {}      # stream id -> (stream, event): the latest gradient write of this bucket on each stream


def plan_buckets(params, bucket_bytes: int):
    """Partition `params` into buckets of >= bucket_bytes fp32 (the last one may be smaller), walking the list in
    REVERSE registration order — approximately the order backward produces the gradients.  For ModelCross that is
    heads -> final norms -> transformer.1 -> transformer.0 -> the shared cls_token / patch_to_embedding / pos_embedding
    last (their gradients are complete only after the embed backward of every modality)."""
    buckets, cur, size = [], [], 0
    for p in reversed(list(params)):
        cur.append(p)
        size += p.numel() * 4
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    return buckets


class BucketedGradReducer:
    def __init__(self, params, process_group=None, bucket_bytes: int = 32 << 20, broadcast: bool = True):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("no parameters to reduce")
        self.device = params[0].device
        self.cuda = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        if broadcast:  # replicas start identical (DDP does the same at wrap time)
            for p in params:
                dist.broadcast(p.data, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
        self.buckets = [_Bucket(ps, self.device) for ps in plan_buckets(params, bucket_bytes)]
        self._bucket_of = {id(p): b for b in self.buckets for p in b.params}
        self._view_of = {id(p): v for b in self.buckets for p, v in zip(b.params, b.views)}
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        self.exposed_launches = 0
        self._avg = self.cuda and dist.get_backend(process_group) == "nccl"     # RCCL applies the 1/world itself; gloo has no AVG
        self._graph_mode = False

    def grad_sink(self, model=None):
        """{address -> bucket view} for every weight matrix: its fp32 master and, when `model` keeps a flat bf16 operand copy
        (ModelCross: xvit.functional.FlatWeights), that copy's view.  Assign to xvit.functional.GRAD_SINK."""
        sink = {}
        for b in self.buckets:
            for p, v in zip(b.params, b.views):
                if p.dim() == 2:
                    sink[p.data_ptr()] = v
        flat = getattr(model, "_flat", None) if model is not None else None
        if flat is not None:
            for p, v16 in zip(flat.params, flat.view16):
                v = self._view_of.get(id(p))
                if v is not None:
                    sink[v16.data_ptr()] = v
        return sink

    # ---- per-gradient hook (runs inside backward) ----------------------------------------------
    def _on_grad(self, p):
        b = self._bucket_of[id(p)]
        if self.cuda:
            # The model's modality branches and fusions run on their own streams, and autograd replays each backward node
            # on the stream of its forward, so the gradients of ONE bucket are written on several streams.  The hook runs
            # with the stream that produced p.grad current: (re-)record that stream's event; _launch waits for all of them.
            st = torch.cuda.current_stream(self.device)
            ent = b.events.get(st.cuda_stream)
            if ent is None:
                ent = b.events[st.cuda_stream] = (st, torch.cuda.Event())
            ent[1].record(st)
        b.pending -= 1
        if b.pending == 0 and not b.launched:
            self._launch(b)

    # ---- captured step: the same bookkeeping from tensor hooks on the step's leaf aliases (xvit.graph.GraphedStep) -----------
    def attach_leaves(self, leaves):
        """`leaves`: [(parameter, alias tensor the captured step differentiates)].  Returns hook handles.  Each alias gradient is
        moved into its bucket view (a no-op for the weight matrices written there by the kernels) and counted like p.grad."""
        handles = []
        for p, a in leaves:
            def hook(g, p=p):
                v = self._view_of[id(p)]
                if g.data_ptr() != v.data_ptr():
                    v.copy_(g)
                self._on_grad(p)
                return None
            handles.append(a.register_hook(hook))
        return handles

    def set_graph_mode(self, on: bool):
        """While True the gradients are taken to sit in the bucket views already (attach_leaves put them there) and p.grad is not read."""
        self._graph_mode = bool(on)

    def _launch(self, b):
        b.launched = True
        if self._graph_mode:
            grads = list(b.views)
        else:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b.params]
        scale = 1.0 / self.world
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            ent = b.events.get(cur.cuda_stream)        # the zero-filled stand-ins above / a launch from finish() are on `cur`
            if ent is None:
                ent = b.events[cur.cuda_stream] = (cur, torch.cuda.Event())
            ent[1].record(cur)
            with torch.cuda.stream(self.comm_stream):
                for _, ev in b.events.values():        # every stream that wrote one of this bucket's gradients
                    self.comm_stream.wait_event(ev)
                if not torch.cuda.is_current_stream_capturing():
                    for g in grads:                    # allocated on a branch stream, last read here
                        g.record_stream(self.comm_stream)
                self._pack(b, grads, None if self._avg else scale)
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self._pack(b, grads, scale)
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def _pack(b, grads, scale):
        """Gradients that are not in their bucket view yet (1-D ones; every one without a GRAD_SINK) are copied there; `scale` is
        the 1/world of the mean where the collective cannot apply it itself."""
        srcs = [g for g, v in zip(grads, b.views) if g.data_ptr() != v.data_ptr()]
        dsts = [v for g, v in zip(grads, b.views) if g.data_ptr() != v.data_ptr()]
        if srcs:
            torch._foreach_copy_(dsts, srcs)
        if scale is not None:
            b.flat.mul_(scale)

    # ---- after backward ------------------------------------------------------------------------
    def finish(self):
        """Wait (stream-side on GPU) for every bucket and point p.grad at the reduced values."""
        for b in self.buckets:
            if not b.launched:  # some gradient never arrived (unused parameter): reduce zeros for it
                self.exposed_launches += 1
                self._launch(b)
        for b in self.buckets:
            b.work.wait()  # NCCL/RCCL: the current stream waits; gloo: the host waits
            for p, v in zip(b.params, b.views):
                p.grad = v
            b.pending, b.work, b.launched = len(b.params), None, False

    def zero_grad(self):
        """Drop gradients (set_to_none semantics) so the next backward writes fresh tensors."""
        for b in self.buckets:
            for p in b.params:
                p.grad = None

    def remove(self):
        for h in self._hooks:
            h.remove()
