"""Generate tests/golden/*.npz from the REFERENCE implementation (run in the build container only).

TEST INFRASTRUCTURE.  Imports the reference's own ``models.kan``, ``models.heads`` and
``training.losses`` from /root/reference (they import and run on CPU here; ``models.backbone``
cannot, it needs ``timm``), and a locally-constructed ``transformers.ViTModel`` (from a config
object; no download) as the independent implementation of the DeiT-Tiny arithmetic.  Only numeric
inputs/outputs are written; no reference source is copied.  /root/reference does not exist on the
GPU box, so nothing here is imported by tests -- they read the .npz files.

    python oracle/make_golden.py           # rewrites tests/golden/
"""
import hashlib
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
sys.path.insert(1, '/root/reference')

from oracle import ref_cpu  # noqa: E402


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def kan_fixture(name, layers, num_knots, degree, batch, seed, scale=1.5):
    from models.kan import KANSeverityModule          # reference class
    torch.manual_seed(seed)
    m = KANSeverityModule(list(layers), num_knots, degree)
    x = (torch.randn(batch, layers[0]) * scale).requires_grad_(True)
    w = torch.randn(batch, layers[-1])
    y = m(x)
    (y * w).sum().backward()
    traj = m.get_activation_trajectory(x.detach())
    out = {'x': x.detach().numpy(), 'y': y.detach().numpy(), 'w': w.numpy(), 'dx': x.grad.numpy(),
           'layers': np.array(layers), 'num_knots': np.array(num_knots), 'degree': np.array(degree)}
    for k, v in m.state_dict().items():
        out['sd.' + k] = v.numpy()
    for k, p in m.named_parameters():
        out['grad.' + k] = p.grad.numpy()
    for i, t in enumerate(traj):
        out[f'traj.{i}'] = t.detach().numpy()
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print(name, 'y', y.detach().flatten()[:4].tolist())


def basis_fixture():
    from models.kan import BSplineBasis
    out = {}
    for G in (5, 32):
        knots = torch.linspace(-1, 1, G + 6)
        nb = G + 2
        pts = [torch.linspace(-1, 1, 21)]
        ulp = torch.tensor(1.0).nextafter(torch.tensor(2.0)) - 1.0
        for t in knots:
            pts.append(torch.stack([t - 4 * ulp * t.abs().clamp_min(1e-3), t, t + 4 * ulp * t.abs().clamp_min(1e-3)]))
        pts.append(torch.tensor([-1.0, 0.3999, 0.4, 0.8378, 0.84, 0.999999, 1.0, 1.5, -1.5]))
        g = torch.Generator().manual_seed(7)
        pts.append(torch.tanh(torch.randn(256, generator=g) * 1.5))
        x = torch.cat(pts).unsqueeze(0)
        b = BSplineBasis.compute_basis(x, knots, 3)
        out[f'g{G}.x'] = x.numpy()
        out[f'g{G}.knots'] = knots.numpy()
        out[f'g{G}.basis'] = b.numpy()
        assert b.shape[-1] == nb
    np.savez_compressed(os.path.join(GOLD, 'kan_basis.npz'), **out)


def heads_fixture():
    from models.heads import ClassificationHead, OrdinalHead, UncertaintyHead
    torch.manual_seed(3)
    c, o, u = ClassificationHead(192, 128, 4, 0.3).eval(), OrdinalHead(192, 128, 4, 0.3).eval(), UncertaintyHead(192, 128, 0.3).eval()
    x = torch.randn(8, 192) * 2.0
    x.requires_grad_(True)
    cl, ol = c(x), o(x)
    mu, lv = u(x)
    probs, sev = o.predict_probabilities(x), o.predict_severity(x)
    g = torch.Generator().manual_seed(4)
    ws = [torch.randn(t.shape, generator=g) for t in (cl, ol, mu, lv)]
    (cl * ws[0]).sum().add((ol * ws[1]).sum()).add((mu * ws[2]).sum()).add((lv * ws[3]).sum()).backward()
    out = {'x': x.detach().numpy(), 'cls_logits': cl.detach().numpy(), 'ordinal_logits': ol.detach().numpy(),
           'mu': mu.detach().numpy(), 'log_var': lv.detach().numpy(), 'ord_probs': probs.detach().numpy(),
           'ord_severity': sev.detach().numpy(), 'dx': x.grad.numpy(),
           'w.cls': ws[0].numpy(), 'w.ord': ws[1].numpy(), 'w.mu': ws[2].numpy(), 'w.lv': ws[3].numpy()}
    for name, m in (('classification_head', c), ('ordinal_head', o), ('uncertainty_head', u)):
        for k, v in m.state_dict().items():
            out[f'sd.{name}.{k}'] = v.numpy()
        for k, p in m.named_parameters():
            out[f'grad.{name}.{k}'] = p.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, 'heads.npz'), **out)


def loss_fixture():
    from training.losses import JointLoss
    torch.manual_seed(0)
    B = 8
    outd = {'cls_logits': torch.randn(B, 4, requires_grad=True), 'ordinal_logits': torch.randn(B, 3, requires_grad=True),
            'mu': torch.randn(B, 1, requires_grad=True), 'log_var': torch.randn(B, 1, requires_grad=True),
            'kan_severity': (3 * torch.rand(B, 1)).requires_grad_(True)}
    y = torch.randint(0, 4, (B,))
    alpha = torch.tensor([1.0, 0.7, 1.3, 2.0])
    out = {'y': y.numpy(), 'alpha': alpha.numpy()}
    for k, v in outd.items():
        out['in.' + k] = v.detach().numpy()
    for stage in (1, 2, 3, 4):
        for k in outd:
            outd[k].grad = None
        l = JointLoss(1.0, 0.5, 0.5, 2.0, alpha)(outd, y, y, stage)
        l['total_loss'].backward()
        for k, v in l.items():
            out[f's{stage}.{k}'] = v.detach().numpy()
        for k, v in outd.items():
            out[f's{stage}.grad.{k}'] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
    np.savez_compressed(os.path.join(GOLD, 'joint_loss.npz'), **out)
    print('joint_loss s4', float(out['s4.total_loss']))


def vit_fixture(name, depth, batch, seed):
    """Independent DeiT-Tiny arithmetic: transformers.ViTModel built from a local config."""
    from transformers import ViTConfig, ViTModel
    cfg = ViTConfig(hidden_size=192, num_hidden_layers=depth, num_attention_heads=3, intermediate_size=768,
                    layer_norm_eps=1e-6, qkv_bias=True, hidden_act='gelu', hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0, image_size=224, patch_size=16)
    vm = ViTModel(cfg, add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(seed)
    sd = ref_cpu.init_vit_state(depth, g)
    # load OUR seeded weights into the HF model through the inverse of hf_vit_state_to_timm
    hf = vm.state_dict()
    mapped = ref_cpu.hf_vit_state_to_timm(hf, depth)            # view-sharing for non-cat entries
    with torch.no_grad():
        hf['embeddings.cls_token'].copy_(sd['cls_token'])
        hf['embeddings.position_embeddings'].copy_(sd['pos_embed'])
        hf['embeddings.patch_embeddings.projection.weight'].copy_(sd['patch_embed.proj.weight'])
        hf['embeddings.patch_embeddings.projection.bias'].copy_(sd['patch_embed.proj.bias'])
        hf['layernorm.weight'].copy_(sd['norm.weight']); hf['layernorm.bias'].copy_(sd['norm.bias'])
        for i in range(depth):
            s, d = f'layers.{i}.', f'blocks.{i}.'
            qw, kw, vw = sd[d + 'attn.qkv.weight'].chunk(3)
            qb, kb, vb = sd[d + 'attn.qkv.bias'].chunk(3)
            for n, w, b in (('q', qw, qb), ('k', kw, kb), ('v', vw, vb)):
                hf[s + f'attention.{n}_proj.weight'].copy_(w); hf[s + f'attention.{n}_proj.bias'].copy_(b)
            hf[s + 'attention.o_proj.weight'].copy_(sd[d + 'attn.proj.weight']); hf[s + 'attention.o_proj.bias'].copy_(sd[d + 'attn.proj.bias'])
            hf[s + 'layernorm_before.weight'].copy_(sd[d + 'norm1.weight']); hf[s + 'layernorm_before.bias'].copy_(sd[d + 'norm1.bias'])
            hf[s + 'layernorm_after.weight'].copy_(sd[d + 'norm2.weight']); hf[s + 'layernorm_after.bias'].copy_(sd[d + 'norm2.bias'])
            hf[s + 'mlp.fc1.weight'].copy_(sd[d + 'mlp.fc1.weight']); hf[s + 'mlp.fc1.bias'].copy_(sd[d + 'mlp.fc1.bias'])
            hf[s + 'mlp.fc2.weight'].copy_(sd[d + 'mlp.fc2.weight']); hf[s + 'mlp.fc2.bias'].copy_(sd[d + 'mlp.fc2.bias'])
    vm.load_state_dict(hf)
    del mapped
    x = torch.randn(batch, 3, 224, 224, generator=g)
    with torch.no_grad():
        feats = vm(pixel_values=x).last_hidden_state[:, 0]
        ours = ref_cpu.vit_forward(x, sd)
    print(name, 'HF vs restatement max|diff|', float((feats - ours).abs().max()))
    flat = torch.cat([sd[k].flatten() for k in sorted(sd)])
    digest = hashlib.sha256(flat.numpy().tobytes()).hexdigest()
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), features=feats.numpy(), depth=np.array(depth),
                        batch=np.array(batch), seed=np.array(seed), weights_sha256=np.array(digest),
                        x_probe=x[0, :, :2, :4].numpy(), n_params=np.array(flat.numel()))


def main():
    os.makedirs(GOLD, exist_ok=True)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:      # reference configs mkdir in CWD on import (config.py:80-84)
        os.chdir(tmp)
        try:
            kan_fixture('kan_mini', (16, 8, 1), 5, 3, 8, seed=0)
            kan_fixture('kan_default', (192, 64, 16, 1), 5, 3, 8, seed=1)
            kan_fixture('kan_g32', (24, 8, 1), 32, 3, 16, seed=2)
            basis_fixture()
            heads_fixture()
            loss_fixture()
            vit_fixture('vit_depth2', 2, 2, seed=11)
            vit_fixture('vit_depth12', 12, 4, seed=12)
        finally:
            os.chdir(cwd)


if __name__ == '__main__':
    main()
