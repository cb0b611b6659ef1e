"""DeiT-Tiny feature extractor on the HIP path.  Mirrors /root/reference/models/backbone.py (:7-82): the
``DeiTTinyBackbone(pretrained, freeze)`` wrapper with ``.model`` / ``.embed_dim`` and the timm state_dict key set
(SURVEY.md section 2).  ``timm`` is not a dependency: ``.model`` is a parameter container whose forward runs the
fused HIP backbone (rovit_vit_forward / rovit_vit_backward)."""
import os
import warnings

import torch
import torch.nn as nn

from rovit_hip.functions import VitEngine, VitFn

EMBED_DIM, DEPTH, HEADS, MLP_DIM, PATCH, IMG = 192, 12, 3, 768, 16, 224
TOKENS = (IMG // PATCH) ** 2 + 1


_NO_HOOK = ('the HIP backbone runs blocks as fused kernels: a hook on `{name}` would never fire.  Supported taps: forward '
            'hooks on `blocks[i].attn` and forward / full-backward hooks on `blocks[i].norm1` (fired from the fused path), '
            '`DeiTTinyBackbone.get_attention_maps(x)`, `.get_attention_probabilities(x)` and rovit_hip.taps')


class _NoHooks:
    """Parameter containers whose forward is never called: registering a hook raises instead of silently not firing."""

    def _refuse(self, *a, **k):
        raise NotImplementedError(_NO_HOOK.format(name=type(self).__name__))

    register_forward_hook = register_forward_pre_hook = register_full_backward_hook = _refuse
    register_backward_hook = register_full_backward_pre_hook = _refuse


class _Linear(_NoHooks, nn.Linear):
    pass


class _LayerNorm(_NoHooks, nn.LayerNorm):
    pass


class _Conv2d(_NoHooks, nn.Conv2d):
    pass


class _TapLayerNorm(nn.LayerNorm):
    """norm1 of a block: forward hooks see its output (B,197,192), full-backward hooks the gradient w.r.t. that output
    (what explainability/gradcam.py:18-26,40 registers on blocks[-1].norm1); both are fired by DeiTTiny.forward /
    the fused backward from the kernels' saved buffers.  Other hook kinds cannot be honoured."""

    def _refuse(self, *a, **k):
        raise NotImplementedError(_NO_HOOK.format(name='norm1 (pre-hooks / non-full backward hooks)'))

    register_forward_pre_hook = register_backward_hook = register_full_backward_pre_hook = _refuse


class _PatchEmbed(_NoHooks, nn.Module):
    def __init__(self):
        super().__init__()
        self.proj = _Conv2d(3, EMBED_DIM, kernel_size=PATCH, stride=PATCH)


class _Attention(nn.Module):
    """Forward hooks see the module output (B,197,192), like timm's Attention under the reference's
    get_attention_maps / AttentionRollout hooks (models/backbone.py:37-62, explainability/attention_maps.py:24-32)."""

    def __init__(self):
        super().__init__()
        self.num_heads = HEADS
        self.scale = (EMBED_DIM // HEADS) ** -0.5
        self.qkv = _Linear(EMBED_DIM, 3 * EMBED_DIM, bias=True)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = _Linear(EMBED_DIM, EMBED_DIM)
        self.proj_drop = nn.Dropout(0.0)

    def _refuse(self, *a, **k):
        raise NotImplementedError(_NO_HOOK.format(name='attn (pre-hooks / backward hooks)'))

    register_forward_pre_hook = register_full_backward_hook = register_backward_hook = register_full_backward_pre_hook = _refuse


class _Mlp(_NoHooks, nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = _Linear(EMBED_DIM, MLP_DIM)
        self.act = nn.GELU()
        self.fc2 = _Linear(MLP_DIM, EMBED_DIM)


class _Block(_NoHooks, nn.Module):
    def __init__(self):
        super().__init__()
        self.norm1 = _TapLayerNorm(EMBED_DIM, eps=1e-6)
        self.attn = _Attention()
        self.norm2 = _LayerNorm(EMBED_DIM, eps=1e-6)
        self.mlp = _Mlp()


class DeiTTiny(nn.Module):
    """Parameter layout of timm's ``deit_tiny_patch16_224`` with ``num_classes=0``; forward = fused HIP path."""

    def __init__(self, depth: int = DEPTH):
        super().__init__()
        self.num_features = self.embed_dim = EMBED_DIM
        self.depth = depth
        self.cls_token = nn.Parameter(torch.zeros(1, 1, EMBED_DIM))
        self.pos_embed = nn.Parameter(torch.zeros(1, TOKENS, EMBED_DIM))
        self.patch_embed = _PatchEmbed()
        self.blocks = nn.ModuleList(_Block() for _ in range(depth))
        self.norm = _LayerNorm(EMBED_DIM, eps=1e-6)
        self._engine = None
        # 'bf16' (default): bf16 MFMA operands, fp32 accumulation -- the training / fast path.  'fp32': inference-only
        # reference-precision mode (rovit_vit_forward_f32: every product and sum in fp32) for end-to-end parity at 1e-3.
        self.precision = 'bf16'
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def ordered_parameters(self):
        """Order expected by rovit_vit_* (include/rovit_hip.h)."""
        ps = [self.cls_token, self.pos_embed, self.patch_embed.proj.weight, self.patch_embed.proj.bias,
              self.norm.weight, self.norm.bias]
        for b in self.blocks:
            ps += [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight, b.attn.proj.bias,
                   b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias]
        return ps

    @property
    def engine(self) -> VitEngine:
        if self._engine is None:
            self._engine = VitEngine(self.depth)
        return self._engine

    def _forward_fp32(self, x: torch.Tensor) -> torch.Tensor:
        from rovit_hip import native
        from rovit_hip.native import call, ptr, ptr_array, stream_ptr
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise native.RovitHipError("precision='fp32' is an inference-only parity mode: call it under torch.no_grad() "
                                       '(training runs on the bf16 MFMA path)')
        x = x.detach().float().contiguous()
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, IMG, IMG):
            raise native.RovitHipError(f'backbone expects (B,3,224,224) images, got {tuple(x.shape)}')
        B = x.shape[0]
        params = [p.detach().float().contiguous() for p in self.ordered_parameters()]
        ws = torch.empty(native.load().rovit_vit_f32_workspace_bytes(B), dtype=torch.uint8, device=x.device)
        feats = torch.empty(B, EMBED_DIM, device=x.device, dtype=torch.float32)
        call('rovit_vit_forward_f32', ptr(x), ptr_array(params), ptr(ws), ptr(feats), B, self.depth, stream_ptr())
        return feats

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.precision == 'fp32':
            return self._forward_fp32(x)
        if self.precision != 'bf16':
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {self.precision!r}")
        training = torch.is_grad_enabled()
        attn_hooked = [i for i, b in enumerate(self.blocks) if b.attn._forward_hooks]
        norm_fwd = [i for i, b in enumerate(self.blocks) if b.norm1._forward_hooks]
        norm_bwd = [i for i, b in enumerate(self.blocks) if b.norm1._backward_hooks]
        if not (attn_hooked or norm_fwd or norm_bwd):
            return VitFn.apply(x, self.engine, training, *self.ordered_parameters())
        return self._forward_with_taps(x, training, attn_hooked, norm_fwd, norm_bwd)

    def _forward_with_taps(self, x, training, attn_hooked, norm_fwd, norm_bwd):
        """Hooks registered on blocks[i].attn / blocks[i].norm1 are fired from the fused path's own buffers
        (rovit_hip/taps.py); a hook that returns a value (i.e. wants to REPLACE the activation) cannot be honoured."""
        from rovit_hip import native, taps

        def fire(module, hooks, *args):
            for h in list(hooks.values()):
                if h(module, *args) is not None:
                    raise NotImplementedError('hooks on the fused backbone are observers: returning a replacement is not supported')
        if attn_hooked:
            outs = taps.attention_outputs(self, x.detach())          # one extra inference-mode forward
            for i in attn_hooked:
                fire(self.blocks[i].attn, self.blocks[i].attn._forward_hooks, (None,), outs[i])
        eng = self.engine
        if (norm_fwd or norm_bwd) and not (training and any(p.requires_grad for p in self.parameters())):
            raise native.RovitHipError('hooks on blocks[i].norm1 read the training workspace: call the model with grad enabled '
                                       'and trainable backbone parameters (as explainability/gradcam.py does)')
        eng.grad_taps = {i: self._fire_norm1_backward for i in norm_bwd}
        try:
            feats = VitFn.apply(x, eng, training, *self.ordered_parameters())
        finally:
            eng.grad_taps = {}
        for i in norm_fwd:
            fire(self.blocks[i].norm1, self.blocks[i].norm1._forward_hooks, (None,), taps.norm1_output(self, i))
        return feats

    def _fire_norm1_backward(self, block: int, grad_output: torch.Tensor):
        m = self.blocks[block].norm1
        for h in list(m._backward_hooks.values()):
            if h(m, (None,), (grad_output,)) is not None:
                raise NotImplementedError('hooks on the fused backbone are observers: returning a replacement is not supported')


class DeiTTinyBackbone(nn.Module):
    def __init__(self, pretrained: bool = True, freeze: bool = False):
        super().__init__()
        self.model = DeiTTiny()
        if pretrained:
            self._load_pretrained()
        self.embed_dim = self.model.num_features
        if freeze:
            self.freeze()

    def _load_pretrained(self):
        """The reference downloads ImageNet weights through timm (backbone.py:12-16).  There is no network here:
        weights are read from $ROVIT_DEIT_TINY_WEIGHTS (a timm-keyed state_dict saved with torch.save / safetensors)
        when set, otherwise the seeded random initialisation is kept and a warning says so."""
        path = os.environ.get('ROVIT_DEIT_TINY_WEIGHTS')
        if not path:
            warnings.warn('pretrained=True requested but no network/timm: set ROVIT_DEIT_TINY_WEIGHTS to a local '
                          'deit_tiny_patch16_224 state_dict; keeping random initialisation')
            return
        if path.endswith('.safetensors'):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            sd = torch.load(path, map_location='cpu', weights_only=True)
        sd = {k: v for k, v in sd.items() if not k.startswith('head')}
        self.model.load_state_dict(sd, strict=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x)

    def freeze(self):
        for p in self.model.parameters():
            p.requires_grad = False
        print("Backbone frozen")

    def unfreeze(self):
        for p in self.model.parameters():
            p.requires_grad = True
        print("Backbone unfrozen")

    def get_attention_maps(self, x: torch.Tensor):
        """Reference backbone.py:37-62 hooks ``block.attn`` and collects each attention module's OUTPUT
        ((B,197,192) on current timm).  The fused path keeps that tensor per block; return copies of it."""
        from rovit_hip import taps
        return taps.attention_outputs(self.model, x)

    def get_attention_probabilities(self, x: torch.Tensor):
        """Extension (not in the reference): the softmax probabilities (B,3,197,197) per block, for attention rollout."""
        from rovit_hip import taps
        return taps.attention_probabilities(self.model, x)


def freeze_backbone(model: nn.Module, freeze: bool = True):
    if not hasattr(model, 'backbone'):
        raise AttributeError("Model does not have 'backbone' attribute")
    model.backbone.freeze() if freeze else model.backbone.unfreeze()


def get_backbone_output_dim(backbone_name: str = 'deit_tiny_patch16_224') -> int:
    # same table as the reference (backbone.py:75-82), including its stale 384 for deit_tiny
    return {'deit_tiny_patch16_224': 384, 'deit_small_patch16_224': 384, 'deit_base_patch16_224': 768}.get(backbone_name, 384)
