"""CPU oracle for the RoViT-KAN forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU, fp32) restatement of the
reference's algorithm for the hot path.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- always as the
checker, never as the thing that is measured or shipped.  The product path
(``rovit_hip`` + ``models``) never imports it and raises when the HIP library is missing.

Pinning (see oracle/make_golden.py, tests/golden/):
  * KAN / heads / losses are checked against the reference's own classes imported from
    /root/reference (models/kan.py, models/heads.py, training/losses.py) in this container;
    the resulting input/output/gradient vectors are committed under tests/golden/.
  * The DeiT-Tiny backbone arithmetic lives in `timm` (un-vendored, unpinned
    ``timm>=0.6.0``, requirements.txt:9; not installed here).  The restatement follows timm's
    published VisionTransformer semantics and is cross-checked against
    the locally installed ``transformers.ViTModel`` built from a local config object.  The
    reference holds no golden vector for the backbone, so backbone parity is pinned only by
    the parameter counts it publishes (5 524 416 backbone / 5 706 394 full model,
    outputs/ablation/full_model/test_metrics.json:11) and by that HF cross-check.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# KAN head  (reference: models/kan.py)
# ----------------------------------------------------------------------------------------


def make_knots(num_knots: int = 5, degree: int = 3) -> torch.Tensor:
    """Uniform fp32 knot vector, exactly as the reference registers it (models/kan.py:59-60).

    Note the fp32 ``linspace`` artefacts (e.g. knots[5] == -1.4901e-08 for the default
    configuration); they are part of the state_dict and of the parity contract."""
    return torch.linspace(-1, 1, num_knots + 2 * degree)


def truncated_bspline_basis(x: torch.Tensor, knots: torch.Tensor, degree: int = 3) -> torch.Tensor:
    """Truncated Cox-de Boor basis, restating models/kan.py:8-44.

    The reference keeps only ``num_basis = len(knots) - degree - 1`` degree-0 indicators
    (kan.py:13,23-25) and drops the right recursion term when ``i + 1 >= num_basis``
    (kan.py:39-40).  The result equals the cubic B-spline for ``x < knots[num_basis]`` and is
    identically zero beyond.  Vectorised over the basis index; the floating-point operation
    order per element is the reference's (left term first, then right term)."""
    nk = knots.numel()
    nb = nk - degree - 1
    x = torch.clamp(x, knots[0], knots[-1])                      # kan.py:16
    xe = x.unsqueeze(-1)
    basis = ((xe >= knots[:nb]) & (xe < knots[1:nb + 1])).to(torch.float32)   # kan.py:23-25
    zero = torch.zeros_like(basis[..., :1])
    for d in range(1, degree + 1):                               # kan.py:28-42
        t_i = knots[:nb]
        t_id = knots[d:d + nb]
        left_den = t_id - t_i
        left = torch.where(left_den != 0, (xe - t_i) / torch.where(left_den != 0, left_den, torch.ones_like(left_den)),
                           torch.zeros_like(xe)) * basis
        t_i1 = knots[1:nb + 1]
        t_id1 = knots[d + 1:d + 1 + nb]
        right_den = t_id1 - t_i1
        shifted = torch.cat([basis[..., 1:], zero], dim=-1)     # basis[i+1]; absent for i = nb-1 (kan.py:39)
        right = torch.where(right_den != 0, (t_id1 - xe) / torch.where(right_den != 0, right_den, torch.ones_like(right_den)),
                            torch.zeros_like(xe)) * shifted
        basis = left + right
    return basis


def closed_form_basis(x_norm: torch.Tensor, knots: torch.Tensor):
    """Closed form of the truncated cubic basis on uniform knots (SURVEY.md section 8(a) addendum).

    Returns (j, vals) where ``j`` is the knot interval index (``knots[j] <= x_c < knots[j+1]``)
    and ``vals[..., m]`` is the value of basis ``j - m`` (m = 0..3); entries whose basis index
    falls outside ``[0, nb-1]`` or with ``j >= nb`` are zero.  This is the formula the HIP
    kernel evaluates; the tests check it against ``truncated_bspline_basis``."""
    nb = knots.numel() - 4
    xc = torch.clamp(x_norm, knots[0], knots[-1])
    j = (xc.unsqueeze(-1) >= knots).sum(-1) - 1
    jc = j.clamp(0, knots.numel() - 2)
    tj = knots[jc]
    h = knots[jc + 1] - tj
    u = (xc - tj) / h
    u2, u3 = u * u, u * u * u
    om = 1.0 - u
    vals = torch.stack([u3 / 6.0,
                        (-3 * u3 + 3 * u2 + 3 * u + 1) / 6.0,
                        (3 * u3 - 6 * u2 + 4) / 6.0,
                        om * om * om / 6.0], dim=-1)
    idx = j.unsqueeze(-1) - torch.arange(4)
    ok = (j.unsqueeze(-1) < nb) & (idx >= 0) & (idx < nb)
    return j, torch.where(ok, vals, torch.zeros_like(vals))


def kan_layer_forward(x: torch.Tensor, spline_weights: torch.Tensor, knots: torch.Tensor,
                      lin_w: torch.Tensor, lin_b: torch.Tensor, degree: int = 3) -> torch.Tensor:
    """models/kan.py:70-95: tanh -> basis -> sum_i sum_k basis[b,i,k] W[i,j,k] + Linear(x).

    The reference's ``in x out`` Python loop (kan.py:85-89) is exactly this einsum."""
    basis = truncated_bspline_basis(torch.tanh(x), knots, degree)
    spline = torch.einsum('bik,ijk->bj', basis, spline_weights)
    return F.linear(x, lin_w, lin_b) + spline


def kan_layer_forward_loop(x, spline_weights, knots, lin_w, lin_b, degree: int = 3):
    """Loop-faithful variant of kan.py:83-93 (same accumulation order as the reference);
    used for the honest CPU baseline and to bound the einsum's reordering error."""
    basis = truncated_bspline_basis(torch.tanh(x), knots, degree)
    out = torch.zeros(x.shape[0], spline_weights.shape[1])
    for i in range(spline_weights.shape[0]):
        for j in range(spline_weights.shape[1]):
            out[:, j] += (basis[:, i, :] * spline_weights[i, j]).sum(dim=1)
    return F.linear(x, lin_w, lin_b) + out


def kan_module_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = '',
                       degree: int = 3, loop: bool = False) -> torch.Tensor:
    """models/kan.py:138-149: KAN -> ReLU -> ... -> KAN -> 3*sigmoid."""
    n = 0
    while f'{prefix}kan_layers.{n}.spline_weights' in sd:
        n += 1
    fn = kan_layer_forward_loop if loop else kan_layer_forward
    for i in range(n):
        p = f'{prefix}kan_layers.{i}.'
        x = fn(x, sd[p + 'spline_weights'], sd[p + 'knots'], sd[p + 'linear.weight'], sd[p + 'linear.bias'], degree)
        if i < n - 1:
            x = torch.relu(x)
    return 3.0 * torch.sigmoid(x)


def kan_module_layer_inputs(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = '', degree: int = 3) -> List[torch.Tensor]:
    """Inputs of every KAN layer of models/kan.py:138-149 (features, then the post-ReLU activations): the values whose
    distance to the spline cutoff decides whether two precisions of the same features are comparable."""
    xs, n = [], 0
    while f'{prefix}kan_layers.{n}.spline_weights' in sd:
        n += 1
    for i in range(n):
        xs.append(x)
        p = f'{prefix}kan_layers.{i}.'
        x = kan_layer_forward(x, sd[p + 'spline_weights'], sd[p + 'knots'], sd[p + 'linear.weight'], sd[p + 'linear.bias'], degree)
        if i < n - 1:
            x = torch.relu(x)
    return xs


def kan_cutoff(knots: torch.Tensor, degree: int = 3) -> float:
    """The input value at which the reference's truncated basis jumps to zero: tanh(x) = knots[num_basis]
    (models/kan.py:24,33-40)."""
    nb = knots.numel() - degree - 1
    return float(torch.atanh(knots[nb].double()))


def init_kan_state(layers: List[int], num_knots: int = 5, degree: int = 3,
                   generator: Optional[torch.Generator] = None, prefix: str = '') -> Dict[str, torch.Tensor]:
    """Parameter shapes/initial distributions of models/kan.py:48-68,118-131."""
    sd = {}
    nb = num_knots + degree - 1
    for i in range(len(layers) - 1):
        fi, fo = layers[i], layers[i + 1]
        p = f'{prefix}kan_layers.{i}.'
        sd[p + 'spline_weights'] = torch.randn(fi, fo, nb, generator=generator) * 0.1
        sd[p + 'knots'] = make_knots(num_knots, degree)
        bound = 1.0 / math.sqrt(fi)
        sd[p + 'linear.weight'] = (torch.rand(fo, fi, generator=generator) * 2 - 1) * bound
        sd[p + 'linear.bias'] = (torch.rand(fo, generator=generator) * 2 - 1) * bound
    return sd


# ----------------------------------------------------------------------------------------
# Heads  (reference: models/heads.py)
# ----------------------------------------------------------------------------------------


def mlp_head(x, w1, b1, w2, b2, drop_mask: Optional[torch.Tensor] = None):
    """Linear -> ReLU -> Dropout -> Linear (models/heads.py:17-22, 38-43).  ``drop_mask`` is the
    already-scaled keep mask (1/(1-p) or 0); ``None`` = eval mode."""
    h = torch.relu(F.linear(x, w1, b1))
    if drop_mask is not None:
        h = h * drop_mask
    return F.linear(h, w2, b2)


def heads_forward(features: torch.Tensor, sd: Dict[str, torch.Tensor], stage: int = 4,
                  masks: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, Optional[torch.Tensor]]:
    """The three MLP heads with the curriculum gate of models/rovit_kan.py:93-116."""
    masks = masks or {}
    out: Dict[str, Optional[torch.Tensor]] = {}
    out['cls_logits'] = mlp_head(features, sd['classification_head.fc1.weight'], sd['classification_head.fc1.bias'],
                                 sd['classification_head.fc2.weight'], sd['classification_head.fc2.bias'],
                                 masks.get('cls'))
    out['ordinal_logits'] = None
    out['mu'] = None
    out['log_var'] = None
    if stage >= 2:
        out['ordinal_logits'] = mlp_head(features, sd['ordinal_head.fc1.weight'], sd['ordinal_head.fc1.bias'],
                                         sd['ordinal_head.fc2.weight'], sd['ordinal_head.fc2.bias'],
                                         masks.get('ord'))
    if stage >= 3:
        h = torch.relu(F.linear(features, sd['uncertainty_head.fc1.weight'], sd['uncertainty_head.fc1.bias']))
        if masks.get('unc') is not None:
            h = h * masks['unc']
        out['mu'] = F.linear(h, sd['uncertainty_head.fc_mu.weight'], sd['uncertainty_head.fc_mu.bias'])
        lv = F.linear(h, sd['uncertainty_head.fc_logvar.weight'], sd['uncertainty_head.fc_logvar.bias'])
        out['log_var'] = torch.clamp(lv, min=-10, max=10)        # heads.py:100
    return out


def ordinal_probabilities(cum_logits: torch.Tensor) -> torch.Tensor:
    """models/heads.py:45-67."""
    cp = torch.sigmoid(cum_logits)
    first = cp[:, :1]
    mid = cp[:, 1:] - cp[:, :-1]
    last = 1.0 - cp[:, -1:]
    return torch.cat([first, mid, last], dim=1)


def ordinal_severity(cum_logits: torch.Tensor) -> torch.Tensor:
    """models/heads.py:69-77."""
    p = ordinal_probabilities(cum_logits)
    lv = torch.arange(p.shape[1], dtype=torch.float32)
    return (p * lv).sum(dim=1, keepdim=True)


def init_heads_state(embed_dim=192, hidden=128, num_classes=4,
                     generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """nn.Linear default init for the shapes of models/heads.py:8-15,26-36,81-89."""
    def lin(o, i):
        bound = 1.0 / math.sqrt(i)
        return ((torch.rand(o, i, generator=generator) * 2 - 1) * bound,
                (torch.rand(o, generator=generator) * 2 - 1) * bound)
    sd = {}
    for name, outs in (('classification_head', {'fc2': num_classes}),
                       ('ordinal_head', {'fc2': num_classes - 1}),
                       ('uncertainty_head', {'fc_mu': 1, 'fc_logvar': 1})):
        sd[f'{name}.fc1.weight'], sd[f'{name}.fc1.bias'] = lin(hidden, embed_dim)
        for k, o in outs.items():
            sd[f'{name}.{k}.weight'], sd[f'{name}.{k}.bias'] = lin(o, hidden)
    return sd


# ----------------------------------------------------------------------------------------
# DeiT-Tiny backbone  (reference: models/backbone.py:12-25 -> timm VisionTransformer)
# ----------------------------------------------------------------------------------------

VIT_DIM, VIT_HEADS, VIT_DEPTH, VIT_MLP, VIT_PATCH, VIT_IMG = 192, 3, 12, 768, 16, 224
VIT_TOKENS = (VIT_IMG // VIT_PATCH) ** 2 + 1


def vit_param_shapes(depth: int = VIT_DEPTH, dim: int = VIT_DIM, mlp: int = VIT_MLP, prefix: str = ''):
    """state_dict keys/shapes of timm's deit_tiny_patch16_224 with num_classes=0."""
    s = {prefix + 'cls_token': (1, 1, dim), prefix + 'pos_embed': (1, VIT_TOKENS, dim),
         prefix + 'patch_embed.proj.weight': (dim, 3, VIT_PATCH, VIT_PATCH), prefix + 'patch_embed.proj.bias': (dim,)}
    for i in range(depth):
        b = f'{prefix}blocks.{i}.'
        s[b + 'norm1.weight'] = (dim,); s[b + 'norm1.bias'] = (dim,)
        s[b + 'attn.qkv.weight'] = (3 * dim, dim); s[b + 'attn.qkv.bias'] = (3 * dim,)
        s[b + 'attn.proj.weight'] = (dim, dim); s[b + 'attn.proj.bias'] = (dim,)
        s[b + 'norm2.weight'] = (dim,); s[b + 'norm2.bias'] = (dim,)
        s[b + 'mlp.fc1.weight'] = (mlp, dim); s[b + 'mlp.fc1.bias'] = (mlp,)
        s[b + 'mlp.fc2.weight'] = (dim, mlp); s[b + 'mlp.fc2.bias'] = (dim,)
    s[prefix + 'norm.weight'] = (dim,); s[prefix + 'norm.bias'] = (dim,)
    return s


def init_vit_state(depth: int = VIT_DEPTH, generator: Optional[torch.Generator] = None, prefix: str = '',
                   std: float = 0.02, perturb_norm: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded random weights of the DeiT-Tiny shapes (no checkpoint exists offline).

    Linear/conv/pos/cls ~ N(0, std); biases ~ N(0, std) (non-zero so bias handling is
    exercised); LayerNorm gamma = 1 + N(0, 0.1), beta = N(0, 0.1) when ``perturb_norm``."""
    sd = {}
    for k, shp in vit_param_shapes(depth, prefix=prefix).items():
        if 'norm' in k:
            if k.endswith('weight'):
                sd[k] = torch.ones(shp) + (0.1 * torch.randn(shp, generator=generator) if perturb_norm else 0)
            else:
                sd[k] = 0.1 * torch.randn(shp, generator=generator) if perturb_norm else torch.zeros(shp)
        else:
            sd[k] = torch.randn(shp, generator=generator) * std
    return sd


def vit_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = '', heads: int = VIT_HEADS,
                eps: float = 1e-6, return_tokens: bool = False, attn_taps: Optional[list] = None,
                attn_probs: Optional[list] = None, tap_norm1: Optional[tuple] = None):
    """timm VisionTransformer.forward for deit_tiny_patch16_224, num_classes=0:
    patch conv k16/s16 -> [cls | patches] + pos_embed -> 12 pre-norm blocks -> LayerNorm -> token 0."""
    B = x.shape[0]
    dim = sd[prefix + 'cls_token'].shape[-1]
    t = F.conv2d(x, sd[prefix + 'patch_embed.proj.weight'], sd[prefix + 'patch_embed.proj.bias'], stride=VIT_PATCH)
    t = t.flatten(2).transpose(1, 2)                                   # (B, 196, dim)
    t = torch.cat([sd[prefix + 'cls_token'].expand(B, -1, -1), t], dim=1) + sd[prefix + 'pos_embed']
    hd = dim // heads
    i = 0
    while f'{prefix}blocks.{i}.norm1.weight' in sd:
        b = f'{prefix}blocks.{i}.'
        h = F.layer_norm(t, (dim,), sd[b + 'norm1.weight'], sd[b + 'norm1.bias'], eps)
        if tap_norm1 is not None and tap_norm1[0] == i:                # what hooks on blocks[i].norm1 see (reference
            h = h.detach().requires_grad_(True)                        # explainability/gradcam.py:18-26,40): the output
            tap_norm1[1]['y'] = h                                      # and, by autograd, the gradient w.r.t. it
        qkv = F.linear(h, sd[b + 'attn.qkv.weight'], sd[b + 'attn.qkv.bias'])
        qkv = qkv.reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = torch.softmax((q * hd ** -0.5) @ k.transpose(-2, -1), dim=-1)
        if attn_probs is not None:
            attn_probs.append(a)                                       # (B, heads, N, N) softmax probabilities
        o = (a @ v).transpose(1, 2).reshape(B, -1, dim)
        ao = F.linear(o, sd[b + 'attn.proj.weight'], sd[b + 'attn.proj.bias'])
        if attn_taps is not None:                                      # what a forward hook on blocks[i].attn sees
            attn_taps.append(ao)                                       # (reference models/backbone.py:37-62)
        t = t + ao
        h = F.layer_norm(t, (dim,), sd[b + 'norm2.weight'], sd[b + 'norm2.bias'], eps)
        h = F.gelu(F.linear(h, sd[b + 'mlp.fc1.weight'], sd[b + 'mlp.fc1.bias']))
        t = t + F.linear(h, sd[b + 'mlp.fc2.weight'], sd[b + 'mlp.fc2.bias'])
        i += 1
    if return_tokens:
        return t
    t = F.layer_norm(t, (dim,), sd[prefix + 'norm.weight'], sd[prefix + 'norm.bias'], eps)
    return t[:, 0]


def hf_vit_state_to_timm(hf_sd: Dict[str, torch.Tensor], depth: int) -> Dict[str, torch.Tensor]:
    """Weight mapping transformers.ViTModel -> timm key names."""
    sd = {'cls_token': hf_sd['embeddings.cls_token'], 'pos_embed': hf_sd['embeddings.position_embeddings'],
          'patch_embed.proj.weight': hf_sd['embeddings.patch_embeddings.projection.weight'],
          'patch_embed.proj.bias': hf_sd['embeddings.patch_embeddings.projection.bias'],
          'norm.weight': hf_sd['layernorm.weight'], 'norm.bias': hf_sd['layernorm.bias']}
    for i in range(depth):
        s, d = f'layers.{i}.', f'blocks.{i}.'
        if s + 'attention.q_proj.weight' not in hf_sd:      # older HF naming
            s = f'encoder.layer.{i}.'
            q, k, v = (s + f'attention.attention.{n}' for n in ('query', 'key', 'value'))
            o = s + 'attention.output.dense'
            f1, f2 = s + 'intermediate.dense', s + 'output.dense'
        else:
            q, k, v = (s + f'attention.{n}_proj' for n in 'qkv')
            o = s + 'attention.o_proj'
            f1, f2 = s + 'mlp.fc1', s + 'mlp.fc2'
        sd[d + 'attn.qkv.weight'] = torch.cat([hf_sd[q + '.weight'], hf_sd[k + '.weight'], hf_sd[v + '.weight']])
        sd[d + 'attn.qkv.bias'] = torch.cat([hf_sd[q + '.bias'], hf_sd[k + '.bias'], hf_sd[v + '.bias']])
        sd[d + 'attn.proj.weight'], sd[d + 'attn.proj.bias'] = hf_sd[o + '.weight'], hf_sd[o + '.bias']
        sd[d + 'norm1.weight'], sd[d + 'norm1.bias'] = hf_sd[s + 'layernorm_before.weight'], hf_sd[s + 'layernorm_before.bias']
        sd[d + 'norm2.weight'], sd[d + 'norm2.bias'] = hf_sd[s + 'layernorm_after.weight'], hf_sd[s + 'layernorm_after.bias']
        sd[d + 'mlp.fc1.weight'], sd[d + 'mlp.fc1.bias'] = hf_sd[f1 + '.weight'], hf_sd[f1 + '.bias']
        sd[d + 'mlp.fc2.weight'], sd[d + 'mlp.fc2.bias'] = hf_sd[f2 + '.weight'], hf_sd[f2 + '.bias']
    return sd


# ----------------------------------------------------------------------------------------
# Full model + loss  (reference: models/rovit_kan.py:88-124, training/losses.py)
# ----------------------------------------------------------------------------------------


def init_rovit_state(depth: int = VIT_DEPTH, kan_layers=(192, 64, 16, 1), num_knots: int = 5, degree: int = 3,
                     seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded full-model state_dict with the reference key set."""
    g = torch.Generator().manual_seed(seed)
    sd = init_vit_state(depth, g, prefix='backbone.model.')
    sd.update(init_heads_state(generator=g))
    sd.update(init_kan_state(list(kan_layers), num_knots, degree, g, prefix='kan_module.'))
    return sd


def rovit_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], stage: int = 4, degree: int = 3,
                  masks=None) -> Dict[str, Optional[torch.Tensor]]:
    """models/rovit_kan.py:88-124: backbone -> cls head; ordinal (stage>=2); mu/log_var (>=3); KAN (>=4)."""
    feats = vit_forward(x, sd, prefix='backbone.model.')
    out = heads_forward(feats, sd, stage, masks)
    out['features'] = feats
    out['kan_severity'] = kan_module_forward(feats, sd, 'kan_module.', degree) if stage >= 4 else None
    return out


def focal_loss(logits, targets, gamma=2.0, alpha=None):
    """training/losses.py:15-38."""
    ce = F.cross_entropy(logits, targets, reduction='none')
    pt = torch.softmax(logits, dim=1).gather(1, targets.unsqueeze(1)).squeeze(1)
    fl = (1 - pt) ** gamma * ce
    if alpha is not None:
        fl = alpha[targets] * fl
    return fl.mean()


def ordinal_bce_loss(cum_logits, targets):
    """training/losses.py:48-72: BCE-with-logits on (targets > k), mean over thresholds then batch."""
    k = torch.arange(cum_logits.shape[1], device=cum_logits.device)
    bt = (targets.unsqueeze(1) > k).float()
    return F.binary_cross_entropy_with_logits(cum_logits, bt, reduction='none').mean(dim=1).mean()


def uncertainty_loss(mu, log_var, targets):
    """training/losses.py:80-101: 0.5 * (exp(-s) (y - mu)^2 + s)."""
    y = targets.unsqueeze(1).float() if targets.dim() == 1 else targets
    return (0.5 * ((y - mu) ** 2 * torch.exp(-log_var) + log_var)).mean()


def kan_regression_loss(pred, targets):
    """training/losses.py:109-114."""
    y = targets.unsqueeze(1).float() if targets.dim() == 1 else targets
    return F.mse_loss(pred, y)


def joint_loss(out, class_targets, severity_targets, stage=4, lambda_ord=1.0, mu_unc=0.5, nu_kan=0.5,
               gamma=2.0, alpha=None):
    """training/losses.py:139-181 (stage-gated weighted sum)."""
    losses = {'cls_loss': focal_loss(out['cls_logits'], class_targets, gamma, alpha)}
    total = losses['cls_loss']
    zero = torch.zeros((), device=total.device)
    losses['ord_loss'] = losses['unc_loss'] = losses['kan_loss'] = zero
    if stage >= 2 and out['ordinal_logits'] is not None:
        losses['ord_loss'] = ordinal_bce_loss(out['ordinal_logits'], severity_targets)
        total = total + lambda_ord * losses['ord_loss']
    if stage >= 3 and out['mu'] is not None and out['log_var'] is not None:
        losses['unc_loss'] = uncertainty_loss(out['mu'], out['log_var'], severity_targets)
        total = total + mu_unc * losses['unc_loss']
    if stage >= 4 and out['kan_severity'] is not None:
        losses['kan_loss'] = kan_regression_loss(out['kan_severity'], severity_targets)
        total = total + nu_kan * losses['kan_loss']
    losses['total_loss'] = total
    return losses
