"""RoViT-KAN on the HIP path.  Mirrors /root/reference/models/rovit_kan.py (:9-181): constructor (config object or
scalars, plus the ``embed_dim=`` keyword the reference's own callers use: scripts/train.py:88-97,
evaluation/evaluator.py:233-242), curriculum-stage gate, 6-key output dict, predict/freeze/count helpers."""
from typing import Dict

import torch
import torch.nn as nn

from rovit_hip.functions import ACT_RELU, ACT_SIGMOID3, HeadPhaseFn, HeadsFn

from .backbone import DeiTTinyBackbone
from .heads import ClassificationHead, OrdinalHead, UncertaintyHead, dropout_mask
from .kan import KANSeverityModule


class RoViTKAN(nn.Module):
    def __init__(self, config_or_embed_dim=None, hidden_dim: int = 128, num_classes: int = 4, kan_layers: list = None,
                 kan_num_knots: int = 5, kan_degree: int = 3, dropout: float = 0.3, pretrained: bool = True,
                 embed_dim: int = None):
        super().__init__()
        if hasattr(config_or_embed_dim, 'model'):
            cfg = config_or_embed_dim
            embed_dim = cfg.model.embed_dim
            hidden_dim, num_classes = cfg.model.hidden_dim, cfg.data.num_classes
            kan_layers, kan_num_knots, kan_degree = cfg.model.kan_layers, cfg.model.kan_num_knots, cfg.model.kan_degree
            dropout, pretrained = cfg.model.dropout, cfg.model.pretrained
        elif config_or_embed_dim is not None:
            embed_dim = config_or_embed_dim
        self.backbone = DeiTTinyBackbone(pretrained=pretrained, freeze=False)
        if embed_dim is None:
            embed_dim = self.backbone.embed_dim
        if kan_layers is None:
            kan_layers = [embed_dim, 64, 16, 1]
        self.classification_head = ClassificationHead(embed_dim, hidden_dim, num_classes, dropout)
        self.ordinal_head = OrdinalHead(embed_dim, hidden_dim, num_classes, dropout)
        self.uncertainty_head = UncertaintyHead(embed_dim, hidden_dim, dropout)
        self.kan_module = KANSeverityModule(layers=kan_layers, num_knots=kan_num_knots, degree=kan_degree)
        self._curriculum_stage = 4

    @property
    def curriculum_stage(self) -> int:
        return self._curriculum_stage

    @curriculum_stage.setter
    def curriculum_stage(self, stage: int):
        assert 1 <= stage <= 4, "Stage must be between 1 and 4"
        self._curriculum_stage = stage

    def _head_params(self):
        c, o, u = self.classification_head, self.ordinal_head, self.uncertainty_head
        return [c.fc1.weight, c.fc1.bias, c.fc2.weight, c.fc2.bias, o.fc1.weight, o.fc1.bias, o.fc2.weight, o.fc2.bias,
                u.fc1.weight, u.fc1.bias, u.fc_mu.weight, u.fc_mu.bias, u.fc_logvar.weight, u.fc_logvar.bias]

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        stage = self._curriculum_stage
        if x.shape[0] == 0:
            return self._empty_outputs(x, stage)
        features = self.backbone(x)
        if self._head_phase_fusable(features):
            return self._forward_head_phase(features, stage)
        B, hid = features.shape[0], self.classification_head.fc1.out_features
        masks = None
        if self.training:
            heads = ((self.classification_head, 1), (self.ordinal_head, 2), (self.uncertainty_head, 3))
            live = [stage >= need and h.dropout.p > 0.0 for h, need in heads]
            ps = {h.dropout.p for (h, _), a in zip(heads, live) if a}
            if len(ps) == 1:                      # one random draw for all active heads (3 launches instead of 9-12)
                keep = 1.0 - ps.pop()
                m = torch.empty(sum(live), B, hid, device=features.device).bernoulli_(keep).mul_(1.0 / keep)
                it = iter(m.unbind(0))
                masks = [next(it) if a else None for a in live]
            elif ps:
                masks = [dropout_mask(h.dropout, True, (B, hid), features.device) if a else None for (h, _), a in zip(heads, live)]
        cls_logits, ordinal_logits, mu, log_var = HeadsFn.apply(features, stage, masks, *self._head_params())
        return {
            'cls_logits': cls_logits,
            'features': features,
            'ordinal_logits': ordinal_logits if stage >= 2 else None,
            'mu': mu if stage >= 3 else None,
            'log_var': log_var if stage >= 3 else None,
            'kan_severity': self.kan_module(features) if stage >= 4 else None,
        }

    # ---- the head phase as one forward launch and a two-launch backward (csrc/head_phase.hip) ----------------------------
    head_phase_max_batch = 1024        # one workgroup per sample: beyond this the per-module kernels (sample tiles, matrix cores) win

    def _head_phase_fusable(self, features: torch.Tensor) -> bool:
        """Shapes the fused kernels cover, and nothing a caller could observe differently: a forward / backward hook on any
        head or KAN sub-module would not fire from inside the fused launch, so such a model takes the per-module path."""
        if not features.is_cuda or features.dim() != 2 or not (0 < features.shape[0] <= self.head_phase_max_batch):
            return False
        c, o, u, k = self.classification_head, self.ordinal_head, self.uncertainty_head, self.kan_module
        if not (type(c) is ClassificationHead and type(o) is OrdinalHead and type(u) is UncertaintyHead and type(k) is KANSeverityModule):
            return False
        E, hid, C = features.shape[1], c.fc1.out_features, c.fc2.out_features
        if not (c.fc1.in_features == o.fc1.in_features == u.fc1.in_features == E and o.fc1.out_features == hid == u.fc1.out_features):
            return False
        if not (E % 4 == 0 and E <= 768 and hid % 4 == 0 and hid <= 256 and 2 <= C <= 8 and o.fc2.out_features == C - 1):
            return False
        d = k.layers_dims
        if not (k.degree == 3 and 1 <= len(k.kan_layers) <= 4 and d[0] == E and all(1 <= w <= 64 for w in d[1:]) and
                all(8 <= l.knots.numel() <= 64 for l in k.kan_layers)):
            # (num_knots 5 / 6: dense basis rows; any other grid: one 16-byte load of the four live weights per (input, output) pair --
            # BASELINE configs[4], num_knots 32 at batch 512: 50 us forward against 54 for the three per-layer launches, backward 144 against 126)
            return False
        if self.training and len({h.dropout.p for h in (c, o, u)}) != 1:
            return False
        for top in (c, o, u, k):
            for m in top.modules():
                if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, '_backward_pre_hooks', None):
                    return False
        return True

    def _kan_params(self):
        out = []
        for l in self.kan_module.kan_layers:
            out += [l.spline_weights, l.linear.weight, l.linear.bias]
        return out

    def _forward_head_phase(self, features: torch.Tensor, stage: int) -> Dict[str, torch.Tensor]:
        k = self.kan_module
        nl = len(k.kan_layers)
        cfg = {'stage': stage, 'masks': None, 'drop_p': 0.0, 'seed': 0, 'offset': 0,
               'kan_dims': list(k.layers_dims) if stage >= 4 else [],
               'kan_knots': [l.knots for l in k.kan_layers] if stage >= 4 else [],
               'kan_acts': [ACT_SIGMOID3 if i == nl - 1 else ACT_RELU for i in range(nl)],
               'grad_views': getattr(self, '_head_grad_views', None)}
        p = self.classification_head.dropout.p
        if self.training and p > 0.0:
            # dropout drawn inside the kernel (Philox keyed by the device generator's seed, counter advanced like torch's own kernels do)
            gen = torch.cuda.default_generators[features.device.index]
            n = features.shape[0] * self.classification_head.fc1.out_features
            off = gen.get_offset()
            gen.set_offset(off + 4 * ((n + 3) // 4))
            cfg.update(drop_p=float(p), seed=gen.initial_seed(), offset=off)
        cls_logits, ordinal_logits, mu, log_var, kan = HeadPhaseFn.apply(features, cfg, *self._head_params(),
                                                                         *(self._kan_params() if stage >= 4 else []))
        return {
            'cls_logits': cls_logits,
            'features': features,
            'ordinal_logits': ordinal_logits if stage >= 2 else None,
            'mu': mu if stage >= 3 else None,
            'log_var': log_var if stage >= 3 else None,
            'kan_severity': kan if stage >= 4 else None,
        }

    def _empty_outputs(self, x: torch.Tensor, stage: int) -> Dict[str, torch.Tensor]:
        """An empty batch gives empty outputs of the right widths, as the reference's modules do (every op of
        rovit_kan.py:88-124 accepts a zero-length batch dimension); no kernel is launched."""
        from rovit_hip import native
        native.ptr(x)                             # same device / dtype rules as a real batch (CPU tensors raise)
        z = lambda w: torch.zeros(0, w, device=x.device, dtype=torch.float32)
        return {'cls_logits': z(self.classification_head.fc2.out_features), 'features': z(self.backbone.embed_dim),
                'ordinal_logits': z(self.ordinal_head.fc2.out_features) if stage >= 2 else None,
                'mu': z(1) if stage >= 3 else None, 'log_var': z(1) if stage >= 3 else None,
                'kan_severity': z(self.kan_module.layers_dims[-1]) if stage >= 4 else None}

    def predict(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            out = self.forward(x)
            probs = torch.softmax(out['cls_logits'], dim=1)
            pred = {'class': torch.argmax(probs, dim=1), 'class_probs': probs, 'features': out['features']}
            if out['ordinal_logits'] is not None:
                p = OrdinalHead.probabilities_from_logits(out['ordinal_logits'])
                levels = torch.arange(p.shape[1], dtype=torch.float32, device=p.device)
                pred['ordinal_probs'] = p
                pred['ordinal_severity'] = (p * levels).sum(dim=1, keepdim=True)
            if out['mu'] is not None:
                pred['uncertainty_mu'] = out['mu']
                pred['uncertainty_std'] = torch.exp(0.5 * out['log_var'])
            if out['kan_severity'] is not None:
                pred['kan_severity'] = out['kan_severity']
            return pred

    def freeze_backbone(self):
        self.backbone.freeze()

    def unfreeze_backbone(self):
        self.backbone.unfreeze()

    def get_attention_maps(self, x: torch.Tensor):
        return self.backbone.get_attention_maps(x)

    def count_parameters(self) -> Dict[str, int]:
        def n(m):
            return sum(p.numel() for p in m.parameters() if p.requires_grad)
        counts = {'backbone': n(self.backbone), 'classification_head': n(self.classification_head),
                  'ordinal_head': n(self.ordinal_head), 'uncertainty_head': n(self.uncertainty_head),
                  'kan_module': n(self.kan_module)}
        counts['total'] = sum(counts.values())
        return counts
