"""Optimizer for the HIP path: global-norm gradient clipping + AdamW with the reference's parameter grouping.

Reference: training/optimizer.py:7-32 (AdamW; parameters whose name contains 'backbone' at lr/10, the rest at lr;
weight_decay 1e-4) and training/trainer.py:123-128,137-141 (clip_grad_norm_(parameters, 1.0) before the step).

Every parameter is re-homed into a flat fp32 buffer (each nn.Parameter keeps its identity and becomes a view):
* the backbone's 5.5 M parameters next to the flat gradient buffer the HIP backward already writes;
* the ~0.18 M head / KAN parameters in a second buffer, one segment per top-level module with EVERY parameter
  starting on a 16-byte boundary (the head / KAN kernels read them with 16-byte loads and check the alignment; the
  padding floats stay zero in the parameter, gradient and moment buffers, so norms and AdamW are unchanged); their
  gradients (separate autograd tensors) are packed by one multi-tensor copy per step.
The whole step is then: squared norms (rovit_sq_norm_accum per buffer) -> clip coefficient on the device
(rovit_clip_coef) -> fused clip-scale + decoupled weight decay + Adam (rovit_adamw_flat per buffer / active segment).
A module whose parameters received no gradient this step (curriculum stage gating) is skipped entirely, like
torch.optim.AdamW skips parameters with ``grad is None``; its bias-correction step count does not advance.

``RoViTAdamW`` is a ``torch.optim.Optimizer`` with the reference's two parameter groups (``param_groups[0]`` =
backbone at lr/10, ``param_groups[1]`` = heads/KAN at lr), so ``CosineAnnealingLR`` (training/optimizer.py:35-44) and
``get_lr`` (:47-49) work on it unchanged; ``build_optimizer`` / ``build_scheduler`` / ``get_lr`` mirror that file.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import native
from .native import call, ptr, stream_ptr


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class _Segment:
    """Parameters of one top-level module, contiguous in the flat buffer; every parameter starts on a 16-byte
    boundary (offsets[i], relative to the flat buffer) and `numel` counts the padded extent."""

    def __init__(self, name: str, params: List[torch.nn.Parameter], offset: int):
        self.name, self.params, self.offset = name, params, offset
        self.offsets, o = [], offset
        for p in params:
            self.offsets.append(o)
            o += _pad4(p.numel())
        self.numel = o - offset
        self.t = 0
        self.grad_views: List[torch.Tensor] = []


class RoViTAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr: float = 1e-4, weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_grad_norm: Optional[float] = 1.0):
        self.model = model
        self.max_grad_norm = max_grad_norm
        self.vit = model.backbone.model
        self.engine = self.vit.engine
        self.bb_params = self.vit.ordered_parameters()
        if any(p.numel() % 4 for p in self.bb_params):
            raise native.RovitHipError('backbone parameters are expected to be multiples of 4 floats (16-byte aligned views)')
        self._bb_offsets, o = [], 0
        for p in self.bb_params:
            self._bb_offsets.append(o)
            o += p.numel()
        self._bb_total = o
        # head / KAN parameters: one segment per top-level module, segment starts padded to 4 floats (16 bytes)
        groups = {}
        for n, p in model.named_parameters():
            if not n.startswith('backbone.'):
                groups.setdefault(n.split('.')[0], []).append(p)
        self.segments: List[_Segment] = []
        off = 0
        for name, ps in groups.items():
            self.segments.append(_Segment(name, ps, off))
            off += self.segments[-1].numel
        self._o_total = max(off, 4)
        self.other_params = [p for s in self.segments for p in s.params]
        self.t = 0
        self.p_flat = self.m_flat = self.v_flat = self.o_flat = self.o_grad = self.o_m = self.o_v = None
        self._pending_flat = None
        super().__init__([{'params': list(self.bb_params), 'lr': lr / 10.0}, {'params': list(self.other_params), 'lr': lr}],
                         dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.last_grad_norm: Optional[torch.Tensor] = None
        # The reference builds its optimizer BEFORE Trainer.__init__ moves the model to the device (scripts/train.py:104 vs
        # training/trainer.py:32): the flat buffers are therefore created when the parameters are first seen on a
        # CUDA/HIP device -- here if they already are, else at the first step / state_dict call -- and again if the model
        # is moved afterwards (moments are carried over).
        if self.bb_params[0].device.type == 'cuda':
            self._build()

    def _stale(self) -> bool:
        p0 = self.bb_params[0]
        return self.p_flat is None or p0.device != self.p_flat.device or p0.data_ptr() != self.p_flat.data_ptr()

    def _build(self):
        dev = self.bb_params[0].device
        if dev.type != 'cuda':
            raise native.RovitHipError('RoViTAdamW needs the model on a CUDA/HIP device before the first step '
                                       '(model.to(device); building the optimizer earlier is fine)')
        if any(p.device != dev for p in list(self.bb_params) + self.other_params):
            raise native.RovitHipError('RoViTAdamW: all parameters must live on one device')
        old = None if self.p_flat is None else (self.m_flat, self.v_flat, self.o_m, self.o_v)
        self.p_flat = self._rehome(self.bb_params, self._bb_offsets, self._bb_total, dev)
        self.m_flat = torch.zeros_like(self.p_flat)
        self.v_flat = torch.zeros_like(self.p_flat)
        self.o_flat = torch.zeros(self._o_total, dtype=torch.float32, device=dev)
        for s in self.segments:
            self._rehome(s.params, s.offsets, s.numel, dev, self.o_flat)
        self.o_grad = torch.zeros_like(self.o_flat)
        for s in self.segments:
            s.grad_views = [self.o_grad[o:o + p.numel()].view_as(p) for o, p in zip(s.offsets, s.params)]
        self.o_m = torch.zeros_like(self.o_flat)
        self.o_v = torch.zeros_like(self.o_flat)
        # the fused head phase (models/rovit_kan.py) may write its parameter gradients straight into these views: no per-step pack copy
        self.model._head_grad_views = {p.data_ptr(): v for s in self.segments for p, v in zip(s.params, s.grad_views)}
        if old is not None:                                  # the model was moved after the buffers existed: keep the moments
            for dst, src in zip((self.m_flat, self.v_flat, self.o_m, self.o_v), old):
                dst.copy_(src)
        self._sq = torch.zeros((), dtype=torch.float32, device=dev)
        # ticket + fixed-order block partials of rovit_sq_norm_clip (one per 4096 gradient floats; bit-reproducible norm)
        self._sq_scratch = torch.zeros(16 + (self._bb_total + 4095) // 4096 + (self._o_total + 4095) // 4096 + 8, dtype=torch.float32, device=dev)
        self._coef = torch.ones((), dtype=torch.float32, device=dev)
        self._norm = torch.zeros((), dtype=torch.float32, device=dev)
        self.engine._prep_key = None
        if hasattr(self.model, 'kan_module') and hasattr(self.model.kan_module, 'invalidate_prepared'):
            self.model.kan_module.invalidate_prepared()
        if self._pending_flat is not None:
            flat, self._pending_flat = self._pending_flat, None
            self._load_flat(flat)

    @staticmethod
    def _rehome(params, offsets, total, dev, flat=None):
        if flat is None:
            flat = torch.empty(total, dtype=torch.float32, device=dev)
        for off, p in zip(offsets, params):
            view = flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        return flat

    def _backbone_active(self) -> bool:
        return self.bb_params[0].requires_grad and self.bb_params[0].grad is not None

    def _pack_grads(self) -> List[_Segment]:
        """Copy the active segments' gradients into o_grad (one multi-tensor copy for all of them; the padding floats
        between parameters are never written and stay zero).  A gradient that already IS its view of o_grad
        (pack_and_install_grads ran, e.g. for the data-parallel bucket) is not copied again."""
        active, dst, src = [], [], []
        for s in self.segments:
            grads = [p.grad for p in s.params]
            if all(g is None for g in grads):
                continue
            for v, g in zip(s.grad_views, grads):
                if g is None:
                    v.zero_()
                elif g.data_ptr() != v.data_ptr():
                    dst.append(v); src.append(g)
            active.append(s)
        if dst:
            if dst[0].is_cuda:           # gradients that are still being written on the head phase's own stream (functions.HeadPhaseFn)
                from .functions import HeadPhaseFn
                HeadPhaseFn.wait_param_grads(dst[0].device)
            torch._foreach_copy_(dst, src)
        return active

    def grad_view_ptrs(self):
        """Addresses of the o_grad views (cached per buffer): tells a packed gradient from a fresh autograd tensor."""
        key = None if self.o_grad is None else self.o_grad.data_ptr()
        if getattr(self, '_gvp_key', None) != key:
            self._gvp = frozenset(v.data_ptr() for s in self.segments for v in s.grad_views)
            self._gvp_key = key
        return self._gvp

    def pack_and_install_grads(self):
        """Pack the head / KAN gradients into the flat o_grad buffer NOW and make every ``param.grad`` its view of that buffer,
        so that a data-parallel all-reduce can run in place on contiguous slices (GradSync(optimizer=...)) and step() finds
        the gradients already packed.  Returns the (offset, numel) runs of o_grad that hold live gradients."""
        if self._stale():
            self._build()
        with torch.no_grad():
            active = self._pack_grads()
            for s in active:
                for p, v in zip(s.params, s.grad_views):
                    if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                        p.grad = v
        return [(first.offset, last.offset + last.numel - first.offset) for first, last in self._runs(active)]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._stale():
            if any(p.grad is not None for p in self.bb_params) and self.p_flat is not None:
                raise native.RovitHipError('the model was moved between backward and optimizer.step(): run the step again')
            self._build()
        eng = self.engine
        gb, gh = self.param_groups[0], self.param_groups[1]
        sp = stream_ptr()
        bb = self._backbone_active()
        if bb and (eng.grad_views is None or self.bb_params[0].grad.data_ptr() != eng.grad_views[0].data_ptr()):
            raise native.RovitHipError('backbone gradients are not the engine-owned flat buffer')
        active = self._pack_grads()
        import ctypes as C
        coef = None
        if self.max_grad_norm is not None:
            # clip_grad_norm_ in ONE launch: squared norm over the backbone's flat buffer and the live runs of o_grad (padding floats
            # stay zero: whole aligned runs are summed), block partials in fixed order, then the coefficient
            bufs = [(eng.grad_flat, eng.grad_flat.numel())] if bb else []
            bufs += [(self.o_grad[first.offset:], last.offset + last.numel - first.offset) for first, last in self._runs(active)]
            if bufs and len(bufs) <= 4:
                arr = (C.c_void_p * len(bufs))(*[ptr(b) for b, _ in bufs])
                cnt = (C.c_size_t * len(bufs))(*[n for _, n in bufs])
                call('rovit_sq_norm_clip', arr, cnt, len(bufs), float(self.max_grad_norm), ptr(self._coef), ptr(self._norm),
                     ptr(self._sq_scratch), self._sq_scratch.numel(), sp)
            elif bufs:                                    # many disjoint runs (a model with many gated modules): one launch per run
                self._sq.zero_()
                for b, n in bufs:
                    call('rovit_sq_norm_accum', ptr(b), n, ptr(self._sq), None, sp)
                call('rovit_clip_coef', ptr(self._sq), float(self.max_grad_norm), ptr(self._coef), ptr(self._norm), sp)
            if bufs:
                self.last_grad_norm = self._norm
                coef = ptr(self._coef)
        # AdamW over the backbone and every run of active segments that share a step count: ONE launch
        segs = []
        hyper = lambda g_: (float(g_['betas'][0]), float(g_['betas'][1]), float(g_['eps']), float(g_['weight_decay']))
        if bb:
            self.t += 1
            segs.append((self.p_flat, eng.grad_flat, self.m_flat, self.v_flat, self.p_flat.numel(), float(gb['lr']), self.t, hyper(gb)))
            eng._prep_key = None            # parameters changed behind torch's version counters: re-prepare weights
        if active and hasattr(self.model, 'kan_module') and hasattr(self.model.kan_module, 'invalidate_prepared'):
            self.model.kan_module.invalidate_prepared()      # parameters change behind torch's version counters
        for first, last in self._runs(active, same_t=True):
            n = last.offset + last.numel - first.offset
            for s in active[active.index(first):active.index(last) + 1]:
                s.t += 1
            o = first.offset
            segs.append((self.o_flat[o:], self.o_grad[o:], self.o_m[o:], self.o_v[o:], n, float(gh['lr']), first.t, hyper(gh)))
        # (one launch when the groups share betas / eps / weight decay, as the reference's do; else one per distinct setting)
        for hp in dict.fromkeys(c[7] for c in segs):
            same = [c for c in segs if c[7] == hp]
            for i in range(0, len(same), 4):
                chunk = same[i:i + 4]
                pa = lambda k: (C.c_void_p * len(chunk))(*[ptr(c[k]) for c in chunk])
                call('rovit_adamw_flat_multi', pa(0), pa(1), pa(2), pa(3), (C.c_size_t * len(chunk))(*[c[4] for c in chunk]),
                     (C.c_float * len(chunk))(*[c[5] for c in chunk]), (C.c_int * len(chunk))(*[c[6] for c in chunk]), len(chunk), coef,
                     hp[0], hp[1], hp[2], hp[3], sp)
        return loss

    @staticmethod
    def _runs(active, same_t: bool = False):
        """(first, last) of every run of segments that are adjacent in the flat buffer (and share a step count)."""
        runs, i = [], 0
        while i < len(active):
            j = i
            while (j + 1 < len(active) and (not same_t or active[j + 1].t == active[i].t) and
                   active[j + 1].offset == active[j].offset + active[j].numel):
                j += 1
            runs.append((active[i], active[j]))
            i = j + 1
        return runs

    def state_dict(self):
        """param_groups as torch reports them + the flat moment buffers and step counts."""
        sd = super().state_dict()
        if self.p_flat is None:                              # nothing stepped yet (optimizer built before model.to(device))
            sd['rovit_flat'] = self._pending_flat
            return sd
        sd['rovit_flat'] = {'m_flat': self.m_flat.clone(), 'v_flat': self.v_flat.clone(), 'o_m': self.o_m.clone(),
                            'o_v': self.o_v.clone(), 't': self.t, 'segment_t': {s.name: s.t for s in self.segments}}
        return sd

    def _load_flat(self, flat):
        self.m_flat.copy_(flat['m_flat']); self.v_flat.copy_(flat['v_flat'])
        self.o_m.copy_(flat['o_m']); self.o_v.copy_(flat['o_v'])
        self.t = int(flat['t'])
        for s in self.segments:
            s.t = int(flat['segment_t'].get(s.name, 0))

    def load_state_dict(self, state_dict):
        flat = state_dict.get('rovit_flat')
        super().load_state_dict({k: v for k, v in state_dict.items() if k != 'rovit_flat'})
        if flat is not None:
            if self.p_flat is None and self.bb_params[0].device.type != 'cuda':
                self._pending_flat = flat                    # applied when the buffers are built on the device
                return
            if self._stale():
                self._build()
            self._load_flat(flat)


def build_optimizer(model, config) -> RoViTAdamW:
    """training/optimizer.py:7-32 on the HIP path (clip_grad_norm_ of trainer.py:123-126 is inside step())."""
    clip = getattr(getattr(config, 'flags', None), 'gradient_clip', 1.0)
    return RoViTAdamW(model, lr=config.train.learning_rate, weight_decay=config.train.weight_decay, max_grad_norm=clip)


def build_scheduler(optimizer, config):
    """training/optimizer.py:35-44."""
    return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=config.train.epochs, eta_min=1e-6)


def get_lr(optimizer) -> float:
    """training/optimizer.py:47-49: the first group's rate (the backbone's)."""
    for group in optimizer.param_groups:
        return group['lr']
