"""Data-parallel gradient synchronisation: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference is single-process (no collectives anywhere), so this is new work: samples are
independent (LayerNorm only), every rank holds a full replica, and the only exchange is one averaged all-reduce of
the gradients per step.  The backbone's gradients live in ONE flat fp32 buffer (VitEngine.grad_flat) written by the
HIP backward block range by block range; each finished range is all-reduced on a side stream while the next range's
kernels run (bucketed overlap; default 2 buckets tapering 8 : 4 blocks = 14.2 / 7.1 MB -- every extra bucket costs 0.09 ms
of range hand-over on one MI355X, measured with a one-rank RCCL group; 3 buckets taper 6 : 4 : 2 blocks = 10.6 / 7.1 / 4.1 MB: large enough for xGMI's
per-link bandwidth, and the last bucket, whose all-reduce is exposed, is the smallest).  Head/KAN gradients (0.7 MB) go in one flat bucket after backward.

The class only needs ``flat`` tensors and ranges, so its bucket logic is exercised on CPU with the gloo backend.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def block_ranges(depth: int, buckets: int, taper: bool = False) -> List[Tuple[int, int]]:
    """Split blocks depth-1..0 into `buckets` contiguous ranges in backward order, e.g. 12,3 -> (11,8),(7,4),(3,0).
    taper=True makes the ranges shrink (weights buckets : ... : 2 : 1, e.g. 12,3 -> (11,6),(5,2),(1,0)): the early,
    large buckets have the rest of the backward to hide behind, the last one -- whose all-reduce is exposed -- is small."""
    buckets = max(1, min(buckets, depth))
    if taper and buckets > 1:
        total = buckets * (buckets + 1) // 2
        cum, edges = 0, [0]
        for i in range(buckets):
            cum += buckets - i
            edges.append(round(cum * depth / total))
        for i in range(1, buckets + 1):                      # every bucket keeps at least one block
            edges[i] = min(max(edges[i], edges[i - 1] + 1), depth - (buckets - i))
    else:
        edges = [round(i * depth / buckets) for i in range(buckets + 1)]
    return [(depth - 1 - edges[i], depth - edges[i + 1]) for i in range(buckets)]


class FlatBucketAllReduce:
    """Averaging all-reduce of slices of one flat gradient buffer, issued on a side stream as slices become ready."""

    def __init__(self, group=None, use_side_stream: bool = True, force: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()      # run the collectives even with one rank (code-path test)
        self.use_side_stream = use_side_stream
        self._stream: Optional[torch.cuda.Stream] = None
        self._pending = []
        self._avg_ok: Optional[bool] = None            # decided by _check_avg on first use (RCCL only)
        self.issued: List[Tuple[int, int]] = []          # (offset, numel) log, for tests

    @property
    def op_name(self) -> str:
        return {True: 'AVG', False: 'SUM then scale by 1/world', None: 'unchecked'}[self._avg_ok]

    def _check_avg(self, device):
        """ReduceOp.AVG had never met more than one RCCL rank before round 4's first multi-GPU run: check it ONCE on a
        4-element tensor (rank r contributes r+1; the mean is (world+1)/2) and fall back to SUM + scale if the backend
        rejects it or returns anything else.  Every rank takes the same decision (the check itself is a collective)."""
        ok = True
        try:
            probe = torch.full((4,), float(dist.get_rank(self.group) + 1), device=device)
            dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.group)
            ok = bool((probe - (self.world + 1) / 2).abs().max().item() < 1e-6)
        except Exception:                                 # noqa: BLE001 -- an unsupported op must not cost the run
            ok = False
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        self._avg_ok = bool(flag.item() > 0.5)

    def _avg(self, t: torch.Tensor):
        if self.world == 1 and not self.force:
            return None
        backend = dist.get_backend(self.group)
        if backend == 'nccl':
            if self._avg_ok is None:
                self._check_avg(t.device)
            if self._avg_ok:
                return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w.wait()                                      # stream-level: the scale is ordered behind the sum on this stream
            t.mul_(1.0 / self.world)
            return None
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        w.wait()
        t.div_(self.world)
        return None

    def stream_for(self, device) -> Optional[torch.cuda.Stream]:
        if self._stream is None and self.use_side_stream and torch.cuda.is_available():
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def reduce_slice(self, flat: torch.Tensor, offset: int, numel: int, ordered: bool = False):
        """ordered=True: the side stream has already been made to wait for the slice (rovit_vit_backward_notify)."""
        self.issued.append((offset, numel))
        if (self.world == 1 and not self.force) or numel == 0:
            return
        piece = flat[offset:offset + numel]
        if flat.is_cuda and self.use_side_stream:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            if not ordered:
                self._stream.wait_stream(torch.cuda.current_stream(flat.device))     # slice is final on the main stream
            with torch.cuda.stream(self._stream):
                w = self._avg(piece)
            if w is not None:
                self._pending.append(w)
        else:
            w = self._avg(piece)
            if w is not None:
                w.wait()

    def finish(self, device=None):
        """Make the main stream wait for every issued reduction (no host sync)."""
        for w in self._pending:
            w.wait()                                   # stream-level wait for NCCL work objects
        self._pending.clear()
        if self._stream is not None:
            torch.cuda.current_stream(device).wait_stream(self._stream)


class GradSync:
    """Wires FlatBucketAllReduce into a RoViTKAN model: backbone buckets overlap with backward, heads/KAN after."""

    def __init__(self, model, buckets: int = 2, group=None, force: bool = False, broadcast_init: bool = True, optimizer=None):
        self.model = model
        self.engine = model.backbone.model.engine
        self.depth = model.backbone.model.depth
        self.reducer = FlatBucketAllReduce(group, force=force)
        self.world = self.reducer.world
        self.active = self.world > 1 or self.reducer.force
        self.ranges = block_ranges(self.depth, buckets, taper=True)
        params = model.backbone.model.ordered_parameters()
        sizes = [p.numel() for p in params]
        self.prefix = sum(sizes[:6])
        self.block_numel = sum(sizes[6:18])
        self.other_params = [p for n, p in model.named_parameters() if not n.startswith('backbone.')]
        # a RoViTAdamW keeps the head / KAN gradients in one flat buffer: reduce that buffer in place instead of flattening
        # the ~30 gradient tensors, reducing the copy and scattering it back (three small launches per step less)
        self.optimizer = optimizer if hasattr(optimizer, 'pack_and_install_grads') else None
        if self.active and broadcast_init and self.world > 1:
            self.broadcast_parameters(group)
        if self.active:
            self.engine.backward_ranges = self.ranges
            self.engine.range_hook = self._on_range
            self.engine.pre_backward_hook = self._on_backbone_backward_start
            dev = params[0].device
            if dev.type == 'cuda':
                self.engine.notify_stream = self.reducer.stream_for(dev)
                if dist.get_backend(group) == 'nccl':
                    self.reducer._check_avg(dev)        # decide AVG vs SUM + scale here, not inside the first backward

    def broadcast_parameters(self, group=None):
        """Replica identity does not depend on seeds or on which checkpoint a rank loaded: rank 0's parameters and
        buffers (the KAN knots) are broadcast once, in place (views into flat optimizer buffers stay views)."""
        with torch.no_grad():
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                dist.broadcast(t.data, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.engine._prep_key = None          # bf16 weights must be re-prepared from the received values
        # ... and so must the KAN module's prepared (transposed / matrix-core) layouts: `.data` writes do not bump _version
        for m in self.model.modules():
            if hasattr(m, 'invalidate_prepared'):
                m.invalidate_prepared()
                m.__dict__.pop('_uniform_cache', None)      # the knot buffers were broadcast too

    def slice_for(self, first: int, last: int) -> Tuple[int, int]:
        off = self.prefix + last * self.block_numel
        n = (first - last + 1) * self.block_numel
        if last == 0:                                   # cls/pos/patch/final-norm grads are complete now too
            off, n = 0, n + self.prefix
        return off, n

    def _on_range(self, engine, first: int, last: int, ordered: bool = False):
        off, n = self.slice_for(first, last)
        self.reducer.reduce_slice(engine.grad_flat, off, n, ordered)

    def _reduce_others(self):
        """One flat bucket for the head/KAN gradients; returns (flat, grads) or None."""
        grads = [p.grad for p in self.other_params if p.grad is not None]
        if not grads:
            return None
        if grads[0].is_cuda:
            # the fused head phase computes these gradients on a stream of its own (functions.HeadPhaseFn): whoever reads them before
            # the backward pass has ended -- this bucket does -- waits for that launch first
            from .functions import HeadPhaseFn
            # with a flat optimizer buffer the reduction stream waits and the backward's own stream does not; the flatten copy below runs
            # on the backward's stream, which then has to wait itself
            rs = self.reducer.stream_for(grads[0].device) if (self.reducer.use_side_stream and self.optimizer is not None) else None
            HeadPhaseFn.wait_param_grads(grads[0].device, rs)
        if self.optimizer is not None:
            for off, n in self.optimizer.pack_and_install_grads():       # param.grad are views of o_grad from here on
                self.reducer.reduce_slice(self.optimizer.o_grad, off, n)
            return self.optimizer.o_grad, None
        flat = torch._utils._flatten_dense_tensors(grads)            # one cat kernel
        self.reducer.reduce_slice(flat, 0, flat.numel())
        return flat, grads

    def _on_backbone_backward_start(self, engine):
        # Autograd runs the heads' and the KAN's backward (and their AccumulateGrad nodes, which have top priority)
        # before the backbone's: their gradients are final here, so their all-reduce hides behind the whole backbone
        # backward instead of being exposed after it.
        self._early = self._reduce_others()

    def finish(self):
        """Call after loss.backward(): the small head/KAN bucket (already in flight if the backbone ran a backward,
        else reduced now), then join the reduction stream."""
        if not self.active:
            return
        pending = getattr(self, '_early', None)
        self._early = None
        late = None
        if pending is None:
            pending = self._reduce_others()
        else:
            # anything that received its gradient only after the early bucket went out (not expected: AccumulateGrad
            # nodes run as soon as their gradient is ready) still gets reduced
            seen = {id(g) for g in pending[1]} if pending[1] is not None else None
            if seen is None:
                packed = self.optimizer.grad_view_ptrs()
                rest = [p.grad for p in self.other_params if p.grad is not None and p.grad.data_ptr() not in packed]
            else:
                rest = [p.grad for p in self.other_params if p.grad is not None and id(p.grad) not in seen]
            if rest:
                late = (torch._utils._flatten_dense_tensors(rest), rest)
                self.reducer.reduce_slice(late[0], 0, late[0].numel())
        if pending is not None:
            self.reducer.finish(pending[0].device)
            for flat, grads in ((pending,) if late is None else (pending, late)):
                if grads is not None:
                    torch._foreach_copy_(grads, torch._utils._unflatten_dense_tensors(flat, grads))   # one multi-tensor copy
        else:
            self.reducer.finish()
