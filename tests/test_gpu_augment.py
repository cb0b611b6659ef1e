"""GPU tests of the on-device CutMix / MixUp.  The reference's data/transforms.py is not in
its checkout (only the call site training/trainer.py:84-96 is), so the checker is a torch restatement of the
published definitions: parity unpinned against the reference, exact against the restatement (a copy / one fp32 FMA)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('shape', [(5, 3, 224, 224), (1, 3, 224, 224), (4, 1, 8, 12), (256, 3, 224, 224)])
def test_mixup_matches_torch(shape):
    from data.transforms import mix_images
    torch.manual_seed(1)
    x = torch.randn(*shape, device=dev())
    perm = torch.randperm(shape[0], device=dev())
    for lam in (0.0, 0.3, 1.0):
        got = mix_images(x, perm, 'mixup', lam=lam)
        ref = lam * x + (1.0 - lam) * x[perm]
        assert float((got - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize('box', [(0, 224, 0, 224), (0, 0, 0, 0), (17, 93, 5, 6), (100, 224, 1, 223), (3, 4, 218, 224)])
def test_cutmix_matches_torch_bit_exact(box):
    from data.transforms import mix_images
    torch.manual_seed(2)
    x = torch.randn(6, 3, 224, 224, device=dev())
    perm = torch.randperm(6, device=dev())
    got = mix_images(x, perm, 'cutmix', box=box)
    ref = x.clone()
    y0, y1, x0, x1 = box
    ref[:, :, y0:y1, x0:x1] = x[perm][:, :, y0:y1, x0:x1]
    assert torch.equal(got, ref)


def test_cutmix_or_mixup_contract():
    """Replays the host draws (same RandomState seed, same torch seed for the permutation) and checks the returned
    batch, label pair and lam against the torch restatement."""
    from data.transforms import cutmix_or_mixup, rand_bbox
    torch.manual_seed(3)
    x = torch.randn(8, 3, 224, 224, device=dev())
    y = torch.arange(8, device=dev()) % 4
    rng, replay = np.random.RandomState(0), np.random.RandomState(0)
    kinds = set()
    for k in range(12):
        torch.manual_seed(100 + k)
        m, ya, yb, lam = cutmix_or_mixup(x, y, True, True, 1.0, 0.2, rng=rng)
        torch.manual_seed(100 + k)
        perm = torch.randperm(8, device=dev())
        cut = replay.rand() < 0.5
        l0 = float(replay.beta(1.0, 1.0) if cut else replay.beta(0.2, 0.2))
        assert torch.equal(ya, y) and torch.equal(yb, y[perm])
        if cut:
            y0, y1, x0, x1 = rand_bbox(224, 224, l0, replay)
            ref = x.clone()
            ref[:, :, y0:y1, x0:x1] = x[perm][:, :, y0:y1, x0:x1]
            assert torch.equal(m, ref)
            assert abs(lam - (1.0 - (y1 - y0) * (x1 - x0) / 224.0 ** 2)) < 1e-12
        else:
            ref = l0 * x + (1.0 - l0) * x[perm]
            assert float((m - ref).abs().max()) < 1e-5 and lam == l0
        assert 0.0 <= lam <= 1.0
        kinds.add(cut)
    assert kinds == {True, False}
    out = cutmix_or_mixup(x, y, False, False)
    assert out[0] is x and out[3] == 1.0
    for flags in ((False, True), (True, False)):                        # only one enabled: no coin is drawn
        m, _, _, lam = cutmix_or_mixup(x, y, *flags, 1.0, 0.2, rng=rng)
        assert m.shape == x.shape and 0.0 <= lam <= 1.0


def test_errors_are_loud():
    from data.transforms import mix_images
    from rovit_hip.native import RovitHipError
    with pytest.raises(RovitHipError):
        mix_images(torch.zeros(2, 3, 8, 8), torch.arange(2), 'mixup')          # CPU tensor
    x = torch.zeros(2, 3, 8, 8, device=dev())
    with pytest.raises(RovitHipError):
        mix_images(x, torch.arange(2), 'cutmix', box=(0, 9, 0, 8))             # box outside the image
    with pytest.raises(RovitHipError):
        mix_images(torch.zeros(2, 3, 8, 6, device=dev()), torch.arange(2), 'mixup')   # width not a multiple of 4
