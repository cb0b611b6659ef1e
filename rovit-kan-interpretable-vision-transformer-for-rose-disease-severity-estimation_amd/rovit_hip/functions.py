"""torch.autograd.Function wrappers around the C ABI (include/rovit_hip.h).

torch is used for device memory, the current stream and autograd bookkeeping only; every FLOP of the
hot path runs in librovit_hip.so.  CPU tensors are rejected (no fallback).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import native
from .native import call, ptr, ptr_array, stream_ptr

ACT_NONE, ACT_RELU, ACT_SIGMOID3 = 0, 1, 2


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# KAN layer  (reference: models/kan.py:70-95 + the activation applied after it, :138-149)
# ------------------------------------------------------------------------------------------------
class KANLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, spline_w, knots, lin_w, lin_b, act: int):
        x, spline_w, knots, lin_w, lin_b = map(_f32c, (x, spline_w, knots, lin_w, lin_b))
        B, in_f = x.shape
        out_f = lin_w.shape[0]
        out = torch.empty(B, out_f, device=x.device, dtype=torch.float32)
        call('rovit_kan_layer_fwd', ptr(x), ptr(spline_w), ptr(knots), ptr(lin_w), ptr(lin_b), ptr(out),
             B, in_f, out_f, knots.numel(), act, stream_ptr())
        ctx.save_for_backward(x, spline_w, knots, lin_w, out)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, g):
        x, spline_w, knots, lin_w, out = ctx.saved_tensors
        g = _f32c(g)
        B, in_f = x.shape
        out_f = lin_w.shape[0]
        need_dx = ctx.needs_input_grad[0]
        need_dw = ctx.needs_input_grad[1] or ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        dx = torch.empty_like(x) if need_dx else None
        dws = torch.empty_like(spline_w) if need_dw else None
        dlw = torch.empty_like(lin_w) if need_dw else None
        dlb = torch.empty(out_f, device=x.device, dtype=torch.float32) if need_dw else None
        call('rovit_kan_layer_bwd', ptr(x), ptr(spline_w), ptr(knots), ptr(lin_w), ptr(out), ptr(g), ptr(dx), ptr(dws),
             ptr(dlw), ptr(dlb), B, in_f, out_f, knots.numel(), ctx.act, 0, stream_ptr())
        return dx, dws, None, dlw, dlb, None


class KANStackFn(torch.autograd.Function):
    """KANSeverityModule.forward (reference models/kan.py:138-149) and its backward as whole-stack launches.
    forward: mode 0 = one rovit_kan_layer_fwd launch per layer (the faster form below ~2048 samples), 1 = rovit_kan_stack_fwd
    (one launch, vector ALU), 2 = rovit_kan_stack_fwd_mfma (one launch, matrix cores).  backward (round 3): rovit_kan_stack_bwd,
    two launches for the whole stack (the per-sample dx chain, then the parameter gradients of every layer).
    inputs: x, acts (tuple of ROVIT_ACT_*), prep (list of per-layer (spline_wt, lin_wt, wm) prepared tensors or None for mode 0),
    mode, then per layer (spline_w, knots, lin_w, lin_b); outputs: every layer's output."""

    @staticmethod
    def forward(ctx, x, acts, prep, mode, *params):
        import ctypes as C
        x = _f32c(x)
        ptr(x)                                     # a CPU tensor is refused here (RovitHipError), before anything asks for a stream
        params = [_f32c(p) for p in params]
        n = len(params) // 4
        B = x.shape[0]
        dims = [x.shape[1]] + [params[4 * l + 2].shape[0] for l in range(n)]
        nks = [params[4 * l + 1].numel() for l in range(n)]
        outs = [torch.empty(B, dims[l + 1], device=x.device, dtype=torch.float32) for l in range(n)]
        arr = lambda xs: (C.c_int * len(xs))(*xs)
        if mode == 2:
            call('rovit_kan_stack_fwd_mfma', ptr(x), ptr_array([p[2] for p in prep]), ptr_array(params[1::4]), ptr_array(params[3::4]),
                 ptr_array(outs), B, arr(dims), arr(nks), arr(list(acts)), n, stream_ptr())
        elif mode == 1:
            call('rovit_kan_stack_fwd', ptr(x), ptr_array([p[0] for p in prep]), ptr_array(params[1::4]), ptr_array([p[1] for p in prep]),
                 ptr_array(params[3::4]), ptr_array(outs), B, arr(dims), arr(nks), arr(list(acts)), n, stream_ptr())
        else:
            st = stream_ptr()
            for l in range(n):
                xin = x if l == 0 else outs[l - 1]
                call('rovit_kan_layer_fwd', ptr(xin), ptr(params[4 * l]), ptr(params[4 * l + 1]), ptr(params[4 * l + 2]), ptr(params[4 * l + 3]),
                     ptr(outs[l]), B, dims[l], dims[l + 1], nks[l], acts[l], st)
        ctx.save_for_backward(x, *params, *outs)
        ctx.n, ctx.acts, ctx.dims, ctx.nks = n, tuple(acts), dims, nks
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        import ctypes as C
        n = ctx.n
        saved = ctx.saved_tensors
        x, params, outs = saved[0], saved[1:1 + 4 * n], saved[1 + 4 * n:]
        B = x.shape[0]
        gouts = [(_f32c(g) if g is not None else None) for g in gouts]
        # layers above the topmost output that received a gradient contribute nothing
        top = max((l for l in range(n) if gouts[l] is not None), default=-1)
        if top < 0:
            return (None,) * (4 + 4 * n)
        need_dx = ctx.needs_input_grad[0]
        need_dw = any(ctx.needs_input_grad[4 + 4 * l + k] for l in range(top + 1) for k in (0, 2, 3))
        nl = top + 1
        dev = x.device
        gz = [torch.empty(B, ctx.dims[l + 1], device=dev, dtype=torch.float32) for l in range(nl)]
        dx = torch.empty_like(x) if need_dx else None
        dws = [torch.empty_like(params[4 * l]) for l in range(nl)] if need_dw else None
        dlw = [torch.empty_like(params[4 * l + 2]) for l in range(nl)] if need_dw else None
        dlb = [torch.empty(ctx.dims[l + 1], device=dev, dtype=torch.float32) for l in range(nl)] if need_dw else None
        arr = lambda xs: (C.c_int * len(xs))(*xs)
        call('rovit_kan_stack_bwd', ptr(x), ptr_array(params[0:4 * nl:4]), ptr_array(params[1:4 * nl:4]), ptr_array(params[2:4 * nl:4]),
             ptr_array(outs[:nl]), ptr_array(gouts[:nl]), ptr_array(gz), ptr(dx),
             ptr_array(dws) if need_dw else None, ptr_array(dlw) if need_dw else None, ptr_array(dlb) if need_dw else None,
             B, arr(ctx.dims[:nl + 1]), arr(ctx.nks[:nl]), arr(list(ctx.acts[:nl])), nl, stream_ptr())
        grads = [None] * (4 * n)
        if need_dw:
            for l in range(nl):
                grads[4 * l], grads[4 * l + 2], grads[4 * l + 3] = dws[l], dlw[l], dlb[l]
        return (dx, None, None, None, *grads)


# ------------------------------------------------------------------------------------------------
# The three MLP heads (reference: models/heads.py:17-22, 38-43, 91-102; gate: models/rovit_kan.py:93-116)
# ------------------------------------------------------------------------------------------------
class HeadsFn(torch.autograd.Function):
    """inputs: features, stage, masks (list of 3 tensors or None), then the 14 parameters.
    outputs: (cls_logits, ordinal_logits, mu, log_var); inactive heads give zero-size placeholders."""

    @staticmethod
    def forward(ctx, features, stage: int, masks, *params):
        features = _f32c(features)
        params = [_f32c(p) for p in params]
        B, embed = features.shape
        hid = params[0].shape[0]
        ncls = params[2].shape[0]
        dev = features.device
        hidden = torch.empty(3, B, hid, device=dev, dtype=torch.float32)
        cls = torch.empty(B, ncls, device=dev, dtype=torch.float32)
        ordl = torch.empty(B, ncls - 1, device=dev, dtype=torch.float32) if stage >= 2 else None
        mu = torch.empty(B, 1, device=dev, dtype=torch.float32) if stage >= 3 else None
        lv = torch.empty(B, 1, device=dev, dtype=torch.float32) if stage >= 3 else None
        parr = ptr_array(params)
        marr = ptr_array([_f32c(m) if m is not None else None for m in masks]) if masks is not None else None
        call('rovit_heads_fwd', ptr(features), parr, marr, ptr(hidden), ptr(cls), ptr(ordl), ptr(mu), ptr(lv),
             B, embed, hid, ncls, stage, stream_ptr())
        ctx.save_for_backward(features, hidden, lv if lv is not None else cls.new_empty(0), *params)
        ctx.masks = masks
        ctx.stage = stage
        ctx.dims = (B, embed, hid, ncls)
        empty = cls.new_empty(0)
        outs = (cls, ordl if ordl is not None else empty, mu if mu is not None else empty, lv if lv is not None else empty)
        ctx.mark_non_differentiable(*[o for o in outs if o.numel() == 0])
        return outs

    @staticmethod
    def backward(ctx, g_cls, g_ord, g_mu, g_lv):
        features, hidden, lv, *params = ctx.saved_tensors
        B, embed, hid, ncls = ctx.dims
        stage = ctx.stage
        dev = features.device

        def act(g, active):
            return _f32c(g) if (active and g is not None) else None
        g_cls = act(g_cls, True)
        g_ord = act(g_ord, stage >= 2)
        g_mu = act(g_mu, stage >= 3)
        g_lv = act(g_lv, stage >= 3)
        if (g_mu is None) != (g_lv is None):          # one of the pair unused by the loss: treat as zero
            z = torch.zeros(B, 1, device=dev, dtype=torch.float32)
            g_mu = g_mu if g_mu is not None else z
            g_lv = g_lv if g_lv is not None else z
        grads: List[Optional[torch.Tensor]] = [None] * 14
        live = [g_cls is not None] * 4 + [g_ord is not None] * 4 + [g_mu is not None] * 6
        for i, p in enumerate(params):
            if live[i]:
                grads[i] = torch.empty_like(p)
        dfeat = torch.empty_like(features)
        scratch = torch.empty(3, B, hid, device=dev, dtype=torch.float32)
        masks = ctx.masks
        marr = ptr_array([_f32c(m) if m is not None else None for m in masks]) if masks is not None else None
        call('rovit_heads_bwd', ptr(features), ptr_array(params), marr, ptr(hidden), ptr(lv) if lv.numel() else None,
             ptr(g_cls), ptr(g_ord), ptr(g_mu), ptr(g_lv), ptr(dfeat), ptr_array(grads), ptr(scratch),
             B, embed, hid, ncls, 0, stream_ptr())
        return (dfeat, None, None, *grads)


class HeadPhaseFn(torch.autograd.Function):
    """Everything RoViTKAN.forward does with the features (reference models/rovit_kan.py:93-124) as ONE forward launch and a
    two-launch backward (rovit_head_phase_fwd / _bwd, csrc/head_phase.hip): the three heads and the KAN stack.

    inputs: features, cfg, the 14 head parameters (order of rovit_heads_fwd), then (spline_w, lin_w, lin_b) per KAN layer.
    cfg: stage; masks (list of 3 scaled keep-masks / None entries) or None; drop_p, seed, offset (dropout drawn in the kernel
    when masks is None and drop_p > 0); kan_knots (list of knot buffers), kan_acts, kan_dims ([] = no KAN stack);
    grad_views: optional {param.data_ptr(): tensor} of flat-buffer views the parameter gradients may be written into directly.
    outputs: cls, ord, mu, log_var, kan (the last KAN layer's output); inactive ones are zero-size placeholders."""

    _streams = {}          # device index -> the stream the parameter-gradient launch runs on
    _pending = {}          # device index -> event recorded behind the most recent parameter-gradient launch (until waited for)

    @staticmethod
    def param_grad_stream(dev):
        s = HeadPhaseFn._streams.get(dev.index)
        if s is None:
            s = HeadPhaseFn._streams[dev.index] = torch.cuda.Stream(device=dev)
        return s

    @staticmethod
    def wait_param_grads(dev, stream=None):
        """Make `stream` (default: the current one) wait for the head / KAN parameter gradients of the backward in flight."""
        ev = HeadPhaseFn._pending.pop(dev.index, None) if stream is None else HeadPhaseFn._pending.get(dev.index)
        if ev is not None:
            (stream if stream is not None else torch.cuda.current_stream(dev)).wait_event(ev)

    @staticmethod
    def _desc(features, cfg, head_params, kan_params):
        import ctypes as C
        d = native.HeadPhase()
        B, E = features.shape
        d.batch, d.embed, d.hid, d.num_classes, d.stage = B, E, head_params[0].shape[0], head_params[2].shape[0], cfg['stage']
        dims = cfg['kan_dims']
        nl = len(dims) - 1 if dims else 0
        d.kan_layers = nl
        for l in range(nl + 1 if nl else 0):
            d.kan_dims[l] = dims[l]
        for l in range(nl):
            d.kan_knots[l] = cfg['kan_knots'][l].numel()
            d.kan_acts[l] = cfg['kan_acts'][l]
            d.kan_w[l], d.kan_lw[l], d.kan_lb[l] = (ptr(kan_params[3 * l + q]) for q in range(3))
            d.kan_knots_p[l] = ptr(cfg['kan_knots'][l])
        d.drop_p, d.seed, d.offset = float(cfg.get('drop_p', 0.0)), int(cfg.get('seed', 0)), int(cfg.get('offset', 0))
        d.features = ptr(features)
        for i, p_ in enumerate(head_params):
            d.head_params[i] = ptr(p_)
        masks = cfg.get('masks')
        if masks is not None:
            for h, m in enumerate(masks):
                d.masks[h] = ptr(m) if m is not None else None
        return d

    @staticmethod
    def forward(ctx, features, cfg, *params):
        features = _f32c(features)
        ptr(features)                              # a CPU tensor is refused here (RovitHipError)
        params = [_f32c(p) for p in params]
        head_params, kan_params = params[:14], params[14:]
        if cfg.get('masks') is not None:
            cfg = dict(cfg, masks=[_f32c(m) if m is not None else None for m in cfg['masks']])
        B = features.shape[0]
        dev = features.device
        hid, C_ = head_params[0].shape[0], head_params[2].shape[0]
        stage = cfg['stage']
        dims = cfg['kan_dims']
        nl = len(dims) - 1 if dims else 0
        d = HeadPhaseFn._desc(features, cfg, head_params, kan_params)
        hidden = torch.empty(3, B, hid, device=dev, dtype=torch.float32)
        cls = torch.empty(B, C_, device=dev, dtype=torch.float32)
        empty = cls.new_empty(0)
        ordl = torch.empty(B, C_ - 1, device=dev, dtype=torch.float32) if stage >= 2 else empty
        mu = torch.empty(B, 1, device=dev, dtype=torch.float32) if stage >= 3 else empty
        lv = torch.empty(B, 1, device=dev, dtype=torch.float32) if stage >= 3 else empty
        kouts = [torch.empty(B, dims[l + 1], device=dev, dtype=torch.float32) for l in range(nl)]
        d.hidden, d.cls = ptr(hidden), ptr(cls)
        d.ord, d.mu, d.lv = (ptr(t) if t.numel() else None for t in (ordl, mu, lv))
        for l in range(nl):
            d.kan_out[l] = ptr(kouts[l])
        import ctypes as C
        call('rovit_head_phase_fwd', C.byref(d), stream_ptr())
        ctx.save_for_backward(features, hidden, lv, *params, *kouts)
        ctx.cfg, ctx.nl = cfg, nl
        kan = kouts[-1] if nl else empty
        outs = (cls, ordl, mu, lv, kan)
        ctx.mark_non_differentiable(*[o for o in outs if o.numel() == 0])
        return outs

    @staticmethod
    def backward(ctx, g_cls, g_ord, g_mu, g_lv, g_kan):
        import ctypes as C
        saved = ctx.saved_tensors
        nl, cfg = ctx.nl, ctx.cfg
        features, hidden, lv = saved[:3]
        params = saved[3:3 + 14 + 3 * nl]
        kouts = saved[3 + 14 + 3 * nl:]
        head_params, kan_params = params[:14], params[14:]
        B, E = features.shape
        hid = head_params[0].shape[0]
        stage = cfg['stage']
        dev = features.device

        def act(g, active):
            return _f32c(g) if (active and g is not None) else None
        g_cls, g_ord = act(g_cls, True), act(g_ord, stage >= 2)
        g_mu, g_lv = act(g_mu, stage >= 3), act(g_lv, stage >= 3)
        g_kan = act(g_kan, nl > 0)
        if (g_mu is None) != (g_lv is None):          # one of the pair unused by the loss: treat as zero
            z = torch.zeros(B, 1, device=dev, dtype=torch.float32)
            g_mu = g_mu if g_mu is not None else z
            g_lv = g_lv if g_lv is not None else z
        d = HeadPhaseFn._desc(features, cfg, head_params, kan_params)
        d.hidden = ptr(hidden)
        d.lv = ptr(lv) if lv.numel() else None
        for l in range(nl):
            d.kan_out[l] = ptr(kouts[l])
        d.g_cls, d.g_ord, d.g_mu, d.g_lv, d.g_kan = ptr(g_cls), ptr(g_ord), ptr(g_mu), ptr(g_lv), ptr(g_kan)
        need_dx = ctx.needs_input_grad[0]
        dfeat = torch.empty_like(features) if need_dx else None
        d.d_features = ptr(dfeat)
        dims = cfg['kan_dims']
        scratch = torch.empty(3 * B * hid + B * sum(dims[1:]) if nl else 3 * B * hid, device=dev, dtype=torch.float32)
        d.dpre = ptr(scratch)
        off = 3 * B * hid
        keep = [scratch]
        for l in range(nl):
            d.kan_gz[l] = scratch.data_ptr() + 4 * off
            off += B * dims[l + 1]
        # parameter gradients: written straight into the optimizer's flat gradient buffer when it offered views and nothing has been
        # accumulated yet (autograd then installs the view as .grad without a copy); otherwise into fresh tensors
        live_h = [g_cls is not None] * 4 + [g_ord is not None] * 4 + [g_mu is not None] * 6
        live_k = [g_kan is not None] * (3 * nl)
        need = [ctx.needs_input_grad[2 + i] for i in range(14 + 3 * nl)]
        want = any(n and l for n, l in zip(need, live_h + live_k))
        views = cfg.get('grad_views')
        direct = bool(views) and all(p_.grad is None for p_ in params)
        grads = [None] * (14 + 3 * nl)
        if want:
            for i, (p_, l) in enumerate(zip(params, live_h + live_k)):
                if not l:
                    continue
                v = views.get(p_.data_ptr()) if direct else None
                grads[i] = v.view_as(p_) if v is not None else torch.empty_like(p_)      # a FRESH alias: see AccumulateGrad's stealing rule
            for i in range(14):
                d.head_grads[i] = ptr(grads[i])
            for l in range(nl):
                d.kan_dw[l], d.kan_dlw[l], d.kan_dlb[l] = (ptr(grads[14 + 3 * l + q]) for q in range(3))
        side = HeadPhaseFn.param_grad_stream(dev) if (want and cfg.get('dw_side_stream', True)) else None
        d.want_param_grads = int(want and side is None)
        call('rovit_head_phase_bwd', C.byref(d), stream_ptr())
        if side is not None:
            # The parameter gradients are sample sums nobody waits for before the optimizer (or the data-parallel bucket): their launch
            # goes to a stream of its own, beside the backbone's backward, and the caller's stream is made to wait for it when the whole
            # backward pass ends (autograd's final callbacks run on the stream that surrounded .backward(), like DDP's).
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                call('rovit_head_phase_bwd_params', C.byref(d), stream_ptr())
            for t in (features, hidden, lv, *kouts, g_cls, g_ord, g_mu, g_lv, g_kan, scratch, *(g for g in grads if g is not None)):
                if t is not None and t.numel():
                    t.record_stream(side)
            done = torch.cuda.Event()
            done.record(side)
            HeadPhaseFn._pending[dev.index] = done
            from torch.autograd import Variable
            Variable._execution_engine.queue_callback(lambda: HeadPhaseFn.wait_param_grads(dev))
        del keep
        out = [g if n else None for g, n in zip(grads, need)]
        return (dfeat, None, *out)


class MLPHeadFn(torch.autograd.Function):
    """One stand-alone head: Linear -> ReLU -> dropout mask -> one or more output Linears
    (models/heads.py:17-22, 38-43, 91-102).  inputs: x, mask|None, flags-per-output, w1, b1, (w, b)*."""

    @staticmethod
    def forward(ctx, x, mask, out_flags, w1, b1, *outs):
        x, w1, b1 = _f32c(x), _f32c(w1), _f32c(b1)
        outs = [_f32c(t) for t in outs]
        mask = _f32c(mask) if mask is not None else None
        B, embed = x.shape
        hid = w1.shape[0]
        st = stream_ptr()
        h = torch.empty(B, hid, device=x.device, dtype=torch.float32)
        call('rovit_linear_fwd', ptr(x), ptr(w1), ptr(b1), ptr(mask), ptr(h), B, embed, hid, 1, st)
        ys = []
        for k, flags in enumerate(out_flags):
            w, b = outs[2 * k], outs[2 * k + 1]
            y = torch.empty(B, w.shape[0], device=x.device, dtype=torch.float32)
            call('rovit_linear_fwd', ptr(h), ptr(w), ptr(b), None, ptr(y), B, hid, w.shape[0], flags, st)
            ys.append(y)
        ctx.save_for_backward(x, h, w1, *outs, *ys)
        ctx.mask, ctx.out_flags = mask, tuple(out_flags)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gys):
        n = len(ctx.out_flags)
        x, h, w1, *rest = ctx.saved_tensors
        outs, ys = rest[:2 * n], rest[2 * n:]
        B, embed = x.shape
        hid = w1.shape[0]
        st = stream_ptr()
        dh = torch.zeros(B, hid, device=x.device, dtype=torch.float32)
        grads = []
        for k in range(n):
            w = outs[2 * k]
            g = gys[k]
            if g is None:
                grads += [None, None]
                continue
            g = _f32c(g)
            dw, db = torch.empty_like(w), torch.empty(w.shape[0], device=x.device, dtype=torch.float32)
            yc = ys[k] if (ctx.out_flags[k] & 2) else None
            call('rovit_linear_bwd', ptr(h), ptr(w), ptr(g), ptr(yc), ptr(ctx.mask), ptr(h), ptr(dh), ptr(dw), ptr(db),
                 B, hid, w.shape[0], 1, st)
            grads += [dw, db]
        dx = torch.empty_like(x)
        dw1, db1 = torch.empty_like(w1), torch.empty(hid, device=x.device, dtype=torch.float32)
        call('rovit_linear_bwd', ptr(x), ptr(w1), ptr(dh), None, None, None, ptr(dx), ptr(dw1), ptr(db1), B, embed, hid, 0, st)
        return (dx, None, None, dw1, db1, *grads)


# ------------------------------------------------------------------------------------------------
# DeiT-Tiny backbone (reference: models/backbone.py:23-25 -> timm VisionTransformer.forward)
# ------------------------------------------------------------------------------------------------
class VitEngine:
    """Owns the device buffers of one backbone instance: prepared bf16 weights, activation workspaces and the
    flat gradient buffer.  One engine per DeiTTinyBackbone module."""
    # mlp_path of engines that do not set their own (tests run the small parity cases through BOTH MLP-half paths by changing this)
    default_mlp_path = native.MLP_AUTO

    def __init__(self, depth: int):
        self.depth = depth
        self.n_params = native.load().rovit_vit_num_params(depth)
        self.prep: Optional[torch.Tensor] = None
        self._prep_key = None
        self._ws_pool = {}          # (batch, training) -> list of free workspaces
        self.grad_flat: Optional[torch.Tensor] = None
        self.grad_stage: Optional[torch.Tensor] = None
        self.grad_views: Optional[List[torch.Tensor]] = None
        self.stage_views: Optional[List[torch.Tensor]] = None
        # called as hook(engine, first_block, last_block, ordered) after each backward range has been enqueued
        # (`ordered` = the reduction stream already waits for the range's gradients, see rovit_vit_backward_notify)
        self.backward_ranges: Optional[Sequence] = None
        self.range_hook = None
        self.notify_stream: Optional[torch.cuda.Stream] = None
        self.pre_backward_hook = None    # called as hook(engine) when the backbone's backward starts (head/KAN grads are final)
        # explainability: {block: callback(block, dL/d(norm1 output))}, set by DeiTTiny.forward for the next forward only
        self.grad_taps = {}
        # which kernels run the MLP half (native.MLP_AUTO: by size; tests force MLP_TWO_LAUNCH / MLP_ONE_LAUNCH): an ARGUMENT of the
        # forward and backward calls -- the forward's choice is saved with the graph, so a later change cannot split a step
        self.mlp_path: Optional[int] = None          # None: VitEngine.default_mlp_path
        self.last_ws = None              # (workspace, batch) of the most recent training-mode forward (read by rovit_hip.taps)

    # -- prepared weights ---------------------------------------------------------------------
    def prepare(self, params: Sequence[torch.Tensor], defer: bool = False) -> int:
        """Bring the bf16 weight images in line with `params`.  defer=True (the training forward): nothing is launched here; the
        return value tells the caller to use rovit_vit_forward_prepare (1; 2 = this buffer's constant tables are not written yet),
        which prepares the blocks' weights beside the patch embedding.  0: the images are current."""
        key = tuple((p.data_ptr(), p._version) for p in params)
        if key == self._prep_key:
            return 0
        dev = params[0].device
        lib = native.load()
        if self.prep is None or self.prep.device != dev:
            self.prep = torch.empty(lib.rovit_vit_prep_bytes(self.depth), dtype=torch.uint8, device=dev)
            self._tables_written = False
        self._prep_key = key
        mode = 1 if getattr(self, '_tables_written', False) else 2
        self._tables_written = True
        if defer:
            return mode
        call('rovit_vit_prepare', ptr_array(params), ptr(self.prep), self.depth, stream_ptr())
        return 0

    # -- workspaces ---------------------------------------------------------------------------
    def take_ws(self, batch: int, training: bool, dev) -> torch.Tensor:
        pool = self._ws_pool.setdefault((batch, training, str(dev)), [])
        if pool:
            return pool.pop()
        nbytes = native.load().rovit_vit_workspace_bytes(batch, self.depth, int(training))
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def give_ws(self, batch: int, training: bool, ws: torch.Tensor):
        pool = self._ws_pool.setdefault((batch, training, str(ws.device)), [])
        if len(pool) < 2:
            pool.append(ws)

    # -- flat gradients -----------------------------------------------------------------------
    def ensure_grads(self, params: Sequence[torch.Tensor]):
        dev = params[0].device
        total = sum(p.numel() for p in params)
        if self.grad_flat is None or self.grad_flat.device != dev or self.grad_flat.numel() != total:
            self.grad_flat = torch.zeros(total, dtype=torch.float32, device=dev)
            self.grad_stage = torch.empty(total, dtype=torch.float32, device=dev)
            self.grad_views, self.stage_views = [], []
            off = 0
            for p in params:
                n = p.numel()
                self.grad_views.append(self.grad_flat[off:off + n].view_as(p))
                self.stage_views.append(self.grad_stage[off:off + n].view_as(p))
                off += n


class VitFn(torch.autograd.Function):
    """features = backbone(images).  Parameter gradients are written by the HIP backward into the engine's
    flat buffer and installed as ``param.grad`` directly (autograd's per-tensor accumulation would cost ~150
    tiny copy kernels); accumulation semantics are preserved (grad += new when a grad already exists)."""

    @staticmethod
    def forward(ctx, images, engine: VitEngine, training: bool, *params):
        images = _f32c(images)
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, 224, 224):
            raise native.RovitHipError(f'backbone expects (B,3,224,224) images, got {tuple(images.shape)}')
        B = images.shape[0]
        dev = images.device
        prep_mode = engine.prepare(params, defer=True)
        need_bwd = training and any(p.requires_grad for p in params) and torch.is_grad_enabled()
        # (inside Function.forward grad mode is disabled; the caller passes the real flag via `training`)
        need_bwd = training and any(p.requires_grad for p in params)
        ws = engine.take_ws(B, need_bwd, dev)
        feats = torch.empty(B, 192, device=dev, dtype=torch.float32)
        parr = ptr_array(params)
        mlp_path = int(engine.mlp_path if engine.mlp_path is not None else VitEngine.default_mlp_path)
        if prep_mode:        # the parameters changed (optimizer step): prepare the weight images inside the forward call
            try:
                call('rovit_vit_forward_prepare', ptr(images), parr, ptr(engine.prep), ptr(ws), ptr(feats), B, engine.depth,
                     int(need_bwd), mlp_path, int(prep_mode == 2), stream_ptr())
            except Exception:
                engine._prep_key, engine._tables_written = None, False       # nothing can be assumed about the images
                raise
        else:
            call('rovit_vit_forward', ptr(images), parr, ptr(engine.prep), ptr(ws), ptr(feats), B, engine.depth,
                 int(need_bwd), mlp_path, stream_ptr())
        ctx.engine, ctx.batch, ctx.need_bwd, ctx.mlp_path = engine, B, need_bwd, mlp_path
        if need_bwd:
            ctx.ws = ws
            ctx.save_for_backward(images)           # the patch-embedding weight gradient gathers its pixels from the batch itself
            ctx.params = params
            ctx.prep_key = engine._prep_key       # (data_ptr, version) of every parameter the bf16 weights were built from
            ctx.grad_taps = dict(engine.grad_taps)
            engine.last_ws = (ws, B, mlp_path)
        else:
            engine.give_ws(B, need_bwd, ws)
        return feats

    @staticmethod
    def backward(ctx, dfeat):
        if not ctx.need_bwd:
            return (None,) * (3 + ctx.engine.n_params)
        engine: VitEngine = ctx.engine
        params = ctx.params
        if ctx.ws is None:
            raise native.RovitHipError('backbone backward called twice on the same graph: the activation workspace is '
                                       'recycled after the first backward (run the forward again instead of retain_graph=True)')
        if engine._prep_key != ctx.prep_key:
            raise native.RovitHipError('backbone parameters changed between forward and backward (optimizer step or '
                                       'in-place update): the prepared bf16 weights no longer match the saved activations')
        dfeat = _f32c(dfeat)
        images, = ctx.saved_tensors
        engine.ensure_grads(params)
        fresh = all(p.grad is None for p in params)
        owned = (not fresh) and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                                    for p, v in zip(params, engine.grad_views))
        # fresh: write straight into the flat buffer; owned: stage + one fused add; else: per-tensor fallback
        targets = engine.grad_views if fresh else engine.stage_views
        parr, garr = ptr_array(params), ptr_array(targets)
        depth = engine.depth
        ranges = engine.backward_ranges or [(depth - 1, 0)]
        hooked = engine.range_hook is not None and fresh
        grad_taps = ctx.grad_taps
        if grad_taps and engine.range_hook is not None:
            # the tapped backward is cut at the tapped blocks and does not hand its ranges to the reduction stream: the backbone
            # gradients would stay rank-local and the replicas would drift apart without any error
            raise native.RovitHipError('a Grad-CAM full-backward hook on blocks[i].norm1 cannot be combined with data-parallel '
                                       'gradient sync (GradSync): run explainability passes on a model without GradSync')
        if grad_taps:
            # explainability taps: every tapped block ends a range of its own, so that its dqkv buffer can be read before
            # block-2 reuses it; the data-parallel notify path is not combined with taps
            cuts = sorted(grad_taps, reverse=True)
            ranges, first = [], depth - 1
            for c in cuts:
                ranges.append((first, c))
                first = c - 1
            if first >= 0:
                ranges.append((first, 0))
            hooked = False
        if hooked and engine.pre_backward_hook is not None:
            engine.pre_backward_hook(engine)
        for first, last in ranges:
            if hooked and engine.notify_stream is not None and last > 0:
                # the range's gradients become visible on the reduction stream; this stream is not stalled
                call('rovit_vit_backward_notify', ptr(images), ptr(dfeat), parr, ptr(engine.prep), ptr(ctx.ws), garr, ctx.batch, depth,
                     first, last, ctx.mlp_path, stream_ptr(), engine.notify_stream.cuda_stream)
                engine.range_hook(engine, first, last, True)
            else:
                call('rovit_vit_backward', ptr(images), ptr(dfeat), parr, ptr(engine.prep), ptr(ctx.ws), garr, ctx.batch, depth,
                     first, last, ctx.mlp_path, stream_ptr())
                if hooked:
                    engine.range_hook(engine, first, last, False)
                if last in grad_taps:
                    from . import taps
                    grad_taps[last](last, taps.norm1_output_grad(params, ctx.ws, ctx.batch, depth, last))
        if fresh:
            for p, v in zip(params, engine.grad_views):
                if p.requires_grad:
                    p.grad = v
        elif owned:
            engine.grad_flat.add_(engine.grad_stage)
            if engine.range_hook is not None:          # accumulated gradients: one bucket over everything
                engine.range_hook(engine, depth - 1, 0, False)
        else:
            if engine.range_hook is not None:
                raise native.RovitHipError('data-parallel sync needs engine-owned gradients: do not replace '
                                           'backbone .grad tensors between backward passes')
            for p, v in zip(params, engine.stage_views):
                if p.requires_grad:
                    p.grad = v.clone() if p.grad is None else p.grad.add_(v)
        if engine.last_ws is not None and engine.last_ws[0] is ctx.ws:
            engine.last_ws = None            # the workspace goes back to the pool: its saved activations are no longer valid
        engine.give_ws(ctx.batch, True, ctx.ws)
        ctx.ws = None
        return (None,) * (3 + engine.n_params)
