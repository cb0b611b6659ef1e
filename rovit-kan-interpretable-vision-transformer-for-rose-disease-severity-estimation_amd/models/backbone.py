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


class _PatchEmbed(nn.Module):
    def __init__(self):
        super().__init__()
        self.proj = nn.Conv2d(3, EMBED_DIM, kernel_size=PATCH, stride=PATCH)


class _Attention(nn.Module):
    def __init__(self):
        super().__init__()
        self.num_heads = HEADS
        self.scale = (EMBED_DIM // HEADS) ** -0.5
        self.qkv = nn.Linear(EMBED_DIM, 3 * EMBED_DIM, bias=True)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(EMBED_DIM, EMBED_DIM)
        self.proj_drop = nn.Dropout(0.0)


class _Mlp(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(EMBED_DIM, MLP_DIM)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(MLP_DIM, EMBED_DIM)


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.norm1 = nn.LayerNorm(EMBED_DIM, eps=1e-6)
        self.attn = _Attention()
        self.norm2 = nn.LayerNorm(EMBED_DIM, eps=1e-6)
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
        self.norm = nn.LayerNorm(EMBED_DIM, eps=1e-6)
        self._engine = None
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

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        training = torch.is_grad_enabled()
        return VitFn.apply(x, self.engine, training, *self.ordered_parameters())


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
