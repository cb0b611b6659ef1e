"""CPU tests of the boundary: the C-ABI library builds/loads and exports every symbol include/rovit_hip.h
declares; host-side logic that needs no GPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'rovit_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(rovit_[a-z0-9_]+)\s*\(', txt)))


@pytest.fixture(scope='module')
def native():
    from rovit_hip import native as n
    if not os.path.exists(n.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return n


def test_library_exports_every_declared_symbol(native):
    lib = ctypes.CDLL(native.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/rovit_hip.h but not exported'
    assert set(native.SIGNATURES) == set(syms)
    lib.rovit_version.restype = ctypes.c_int
    assert lib.rovit_version() >= 100


def test_product_library_has_no_knobs(native):
    """"re-entrant and thread-safe (no global mutable state besides a per-thread last-error string)".  Round 4: the
    shipped library exports no setter and no developer entry point, and does not read the environment (the developer library,
    `make -C csrc dev`, is where A/B switches and ablation bits live)."""
    import subprocess
    lib_path = os.path.join(os.path.dirname(native.LIB_PATH), 'librovit_hip.so')
    dyn = subprocess.run(['nm', '-D', lib_path], capture_output=True, text=True, check=True).stdout
    exported = [l.split()[-1] for l in dyn.splitlines() if ' T ' in l]
    undefined = [l.split()[-1] for l in dyn.splitlines() if ' U ' in l]
    assert not [e for e in exported if e.startswith('rovit_set_') or e.startswith('rovit_dev_')], exported
    assert not [u for u in undefined if u.split('@')[0] in ('getenv', 'secure_getenv')], 'the product library reads the environment'
    assert set(e for e in exported if e.startswith('rovit_')) == set(_header_symbols())


def test_host_side_size_queries(native):
    lib = native.load()
    assert lib.rovit_vit_num_params(12) == 6 + 12 * 12
    assert lib.rovit_vit_prep_bytes(12) > 12 * 2 * (3 * 192 * 192 + 192 * 192 + 2 * 768 * 192) * 2
    infer = lib.rovit_vit_workspace_bytes(256, 12, 0)
    train = lib.rovit_vit_workspace_bytes(256, 12, 1)
    assert train > infer > 256 * 197 * 192 * 4
    assert lib.rovit_wgrad_splits(50432, 768, 192) % 8 == 0


def test_product_path_has_no_cpu_fallback(native):
    from rovit_hip import RovitHipError
    from models.kan import KANLayer, KANSeverityModule
    from models.rovit_kan import RoViTKAN
    with pytest.raises(RovitHipError):
        KANLayer(8, 4)(torch.randn(2, 8))
    with pytest.raises(RovitHipError):
        KANSeverityModule([8, 4, 1])(torch.randn(2, 8))
    m = RoViTKAN(pretrained=False)
    with pytest.raises(RovitHipError):
        m(torch.randn(1, 3, 224, 224))


def test_module_surface_matches_reference_contract(native):
    """constructor spellings, attributes and state_dict keys the reference's callers touch."""
    from oracle import ref_cpu
    from models import RoViTKAN, DeiTTinyBackbone, KANSeverityModule, KANLayer, BSplineBasis, freeze_backbone  # noqa: F401
    m = RoViTKAN(pretrained=False)
    sd = ref_cpu.init_rovit_state(seed=0)
    assert set(m.state_dict().keys()) == set(sd.keys())
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    assert m.count_parameters() == {'backbone': 5524416, 'classification_head': 25220, 'ordinal_head': 25091,
                                    'uncertainty_head': 24962, 'kan_module': 106705, 'total': 5706394}
    assert m.backbone.embed_dim == 192 and m.backbone.model.num_features == 192
    assert hasattr(m.backbone.model.blocks[0].attn, 'attn_drop') and hasattr(m.backbone.model.blocks[-1], 'norm1')
    assert m.kan_module.kan_layers[0].in_features == 192 and m.kan_module.kan_layers[0].knots.numel() == 11
    assert torch.equal(m.kan_module.kan_layers[0].knots, ref_cpu.make_knots(5, 3))
    m.curriculum_stage = 2
    assert m.curriculum_stage == 2
    with pytest.raises(AssertionError):
        m.curriculum_stage = 0
    m.freeze_backbone()
    assert m.count_parameters()['backbone'] == 0 and m.count_parameters()['total'] == 181978
    m.unfreeze_backbone()
    freeze_backbone(m, True)
    with pytest.raises(AttributeError):
        freeze_backbone(torch.nn.Linear(2, 2))
    # the spelling scripts/train.py:88-97 and evaluation/evaluator.py:233-242 use
    m2 = RoViTKAN(embed_dim=192, hidden_dim=128, num_classes=4, kan_layers=[192, 64, 16, 1], kan_num_knots=5,
                  kan_degree=3, dropout=0.3, pretrained=False)
    assert m2.count_parameters()['total'] == 5706394

    class Cfg:          # config-object spelling (experiments/ablation.py:264)
        class model:
            embed_dim, hidden_dim, kan_layers, kan_num_knots, kan_degree, dropout, pretrained = 192, 128, [192, 64, 16, 1], 5, 3, 0.3, False

        class data:
            num_classes = 4
    assert RoViTKAN(Cfg).count_parameters()['total'] == 5706394
    assert len([n for n, _ in m.named_parameters() if 'backbone' in n]) == 6 + 12 * 12


def test_synthetic_dataset_and_loaders_on_cpu():
    """data.dataset: calling conventions of scripts/train.py:73-84,110-111 and evaluate.py:40-46,
    synthetic mode, class weights, seeded split; no GPU needed (device='cpu')."""
    import torch
    from data.dataset import RoseLeafDataset, create_dataloaders, DeviceBatchLoader
    from data.transforms import augmented_transforms, original_transforms
    names = ["Healthy Leaf", "Leaf Holes", "Black Spot", "Dry Leaf"]
    sev = {n: i for i, n in enumerate(names)}
    tr, va, te = create_dataloaders(augmented_root='nope/Augmented Image', original_root='nope/Original Image', class_names=names,
                                    severity_map=sev, augmented_transform=augmented_transforms(), original_transform=original_transforms(),
                                    batch_size=8, train_val_split=0.8, num_workers=0, seed=3, synthetic=50, device=torch.device('cpu'))
    assert isinstance(tr, DeviceBatchLoader) and len(tr) == 5 and len(va) == 2 and len(te) == 2
    base = tr.dataset.dataset
    assert isinstance(base, RoseLeafDataset) and len(base) == 50
    w = base.get_class_weights()
    assert w.shape == (4,) and abs(float(w.mean()) - 1.0) < 0.2
    seen = 0
    for x, c, s in tr:
        assert x.shape[1:] == (3, 224, 224) and x.dtype == torch.float32 and torch.equal(c, s)
        seen += x.shape[0]
    assert seen == 40
    tr2, _, _ = create_dataloaders(None, None, names, sev, batch_size=8, seed=3, synthetic=50, device=torch.device('cpu'))
    assert tr2.dataset.indices == tr.dataset.indices                 # seeded split
    img, c, s = RoseLeafDataset(None, names, sev, None, 'original', synthetic=5)[2]
    assert img.shape == (3, 224, 224) and int(s) == sev[names[int(c)]]
    import pytest
    with pytest.raises(FileNotFoundError):
        RoseLeafDataset('does/not/exist', names, sev)


def test_hooks_that_cannot_fire_are_refused():
    import pytest
    from models.backbone import DeiTTiny
    m = DeiTTiny(1)
    for mod in (m.blocks[0].mlp.fc1, m.blocks[0].attn.qkv, m.blocks[0].norm2, m.patch_embed, m.norm, m.blocks[0].mlp, m.blocks[0]):
        with pytest.raises(NotImplementedError):
            mod.register_forward_hook(lambda *a: None)
    m.blocks[0].norm1.register_forward_hook(lambda *a: None).remove()            # supported taps register fine
    m.blocks[0].norm1.register_full_backward_hook(lambda *a: None).remove()
    m.blocks[0].attn.register_forward_hook(lambda *a: None).remove()
    with pytest.raises(NotImplementedError):
        m.blocks[0].attn.register_full_backward_hook(lambda *a: None)
