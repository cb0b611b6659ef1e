"""KAN severity head on the HIP path.  Mirrors /root/reference/models/kan.py (BSplineBasis :8-44,
KANLayer :47-114, KANSeverityModule :117-170): same constructors, attributes, parameter/buffer names."""
from typing import List, Tuple

import numpy as np
import torch
import torch.nn as nn

from rovit_hip import native
from rovit_hip.functions import ACT_NONE, ACT_RELU, ACT_SIGMOID3, KANLayerFn, KANStackFn


def _check_degree(degree: int):
    if degree != 3:
        raise NotImplementedError('the HIP KAN kernels implement the cubic (degree=3) basis the reference configures '
                                  '(configs/config.py:65)')


class BSplineBasis:
    @staticmethod
    def compute_basis(x: torch.Tensor, knots: torch.Tensor, degree: int = 3) -> torch.Tensor:
        """(B, dim) normalised inputs -> (B, dim, num_basis) truncated cubic basis (reference kan.py:8-44)."""
        _check_degree(degree)
        xf = x.detach().float().contiguous()
        kf = knots.detach().float().contiguous()
        nb = kf.numel() - degree - 1
        out = torch.empty(xf.numel(), nb, device=xf.device, dtype=torch.float32)
        native.call('rovit_kan_basis', native.ptr(xf), native.ptr(kf), native.ptr(out), xf.numel(), kf.numel(),
                    native.stream_ptr())
        return out.view(*x.shape, nb)


class KANLayer(nn.Module):
    def __init__(self, in_features: int, out_features: int, num_knots: int = 5, degree: int = 3):
        super().__init__()
        _check_degree(degree)
        self.in_features, self.out_features = in_features, out_features
        self.num_knots, self.degree = num_knots, degree
        self.num_basis = num_knots + degree - 1
        self.register_buffer('knots', torch.linspace(-1, 1, num_knots + 2 * degree))
        self.spline_weights = nn.Parameter(torch.randn(in_features, out_features, self.num_basis) * 0.1)
        self.linear = nn.Linear(in_features, out_features, bias=True)

    def _run(self, x: torch.Tensor, act: int) -> torch.Tensor:
        return KANLayerFn.apply(x, self.spline_weights, self.knots, self.linear.weight, self.linear.bias, act)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._run(x, ACT_NONE)

    def get_spline_weights(self) -> torch.Tensor:
        return self.spline_weights.detach()

    def plot_activation(self, input_idx: int = 0, output_idx: int = 0, num_points: int = 100) -> Tuple[np.ndarray, np.ndarray]:
        xs = torch.linspace(-1, 1, num_points, device=self.knots.device)
        basis = BSplineBasis.compute_basis(xs.unsqueeze(0), self.knots, self.degree)[0]
        ys = (basis * self.spline_weights[input_idx, output_idx].detach()).sum(dim=1)
        return xs.cpu().numpy(), ys.cpu().numpy()


class KANSeverityModule(nn.Module):
    def __init__(self, layers: List[int] = [384, 64, 16, 1], num_knots: int = 5, degree: int = 3):
        super().__init__()
        self.layers_dims, self.num_knots, self.degree = layers, num_knots, degree
        self.kan_layers = nn.ModuleList(KANLayer(a, b, num_knots, degree) for a, b in zip(layers[:-1], layers[1:]))
        self.activations = nn.ModuleList(nn.ReLU() for _ in range(len(layers) - 2))
        # forward kernel by batch size (MI355X, kernels only, [192,64,16,1]; tools/bench_kan.py --sweep, profiles/r02_bench_kan_sweep.jsonl):
        #   G=5:  per-layer 52 / 89 / 163 us at 1024 / 2048 / 4096; matrix-core stack 103 us flat up to 8192, 186 us at 65536
        #         (VALU stack 114 us flat, 520 us at 65536)
        #   G=32: per-layer 129 / 287 us at 1024 / 2048; VALU stack 213 us flat up to 16384, 446 / 884 us at 32768 / 65536;
        #         matrix-core stack 314 us flat, 340 / 698 us at 32768 / 65536 (it multiplies 36 slots where 5 are non-zero)
        self.fused_min_batch = 2048 if num_knots > 8 else 4096
        self.mfma_min_batch = 32768 if num_knots > 8 else 4096

    def _fusable(self) -> bool:
        """rovit_kan_stack_fwd: up to 4 layers, widths after the input <= 64 and a multiple of 4 (or < 8)."""
        d = self.layers_dims
        return (1 <= len(self.kan_layers) <= 4 and all(w <= 64 and (w % 4 == 0 or w < 8) for w in d[1:]) and
                all(l.knots.numel() <= 64 for l in self.kan_layers))

    def _prepared(self):
        """Per layer (spline_wt (in, nb, out), lin_wt (in, out), wm = the rovit_kan_stack_fwd_mfma layout or None): rebuilt (one tiny
        launch per layer) whenever a parameter changed -- torch bumps ``_version`` on in-place updates; optimizers that
        write through flat buffers (RoViTAdamW) call ``invalidate_prepared()``."""
        key = tuple((p.data_ptr(), p._version) for l in self.kan_layers for p in (l.spline_weights, l.linear.weight))
        if key != getattr(self, '_prep_key', None) or self._prep[0][0].device != self.kan_layers[0].spline_weights.device:
            prep = []
            for l in self.kan_layers:
                w, lw = l.spline_weights.detach().float().contiguous(), l.linear.weight.detach().float().contiguous()
                wt = torch.empty(l.in_features, l.num_basis, l.out_features, device=w.device, dtype=torch.float32)
                lwt = torch.empty(l.in_features, l.out_features, device=w.device, dtype=torch.float32)
                native.call('rovit_kan_prepare', native.ptr(w), native.ptr(lw), native.ptr(wt), native.ptr(lwt), l.in_features,
                            l.out_features, l.num_basis, native.stream_ptr())
                nm = native.load().rovit_kan_mfma_prepared_floats(l.in_features, l.out_features, l.num_basis)
                wm = None
                # the matrix-core kernel computes interval indices arithmetically: uniform grids only (kan.py:59 builds
                # the knots with torch.linspace; a non-uniform buffer could only come from a hand-edited state_dict)
                if nm and self._uniform_knots(l):
                    wm = torch.empty(nm, device=w.device, dtype=torch.float32)
                    native.call('rovit_kan_prepare_mfma', native.ptr(w), native.ptr(lw), native.ptr(wm), l.in_features, l.out_features,
                                l.num_basis, native.stream_ptr())
                prep.append((wt, lwt, wm))
            self._prep, self._prep_key = prep, key
        return self._prep

    def _uniform_knots(self, layer) -> bool:
        """Knot uniformity is a property of a BUFFER the optimizer never touches: checked once per (storage, version) with one
        device-to-host copy, not at every rebuild of the prepared weights (RoViTAdamW invalidates those every step)."""
        cache = self.__dict__.setdefault('_uniform_cache', {})
        key = (layer.knots.data_ptr(), layer.knots._version, str(layer.knots.device))
        hit = cache.get(id(layer))
        if hit is None or hit[0] != key:
            kn = layer.knots.detach().float().cpu()
            hstep = float(kn[-1] - kn[0]) / (kn.numel() - 1)
            hit = (key, bool(((kn[1:] - kn[:-1]) - hstep).abs().max() <= 1e-4 * abs(hstep)))
            cache[id(layer)] = hit
        return hit[1]

    def invalidate_prepared(self):
        self._prep_key = None

    def _trajectory(self, x: torch.Tensor) -> List[torch.Tensor]:
        last = len(self.kan_layers) - 1
        codes = tuple(ACT_SIGMOID3 if i == last else ACT_RELU for i in range(last + 1))
        # One launch with the activations on the CU pays off when there are enough samples to fill the chip with sample
        # tiles (measured on MI355X, kernels only: batch 65536 4.4x / 11x faster than the per-layer kernels at G = 5 / 32;
        # batch 256-512 2-4x SLOWER: a tile walks its 192 input features serially, the per-layer kernels split them)
        if self._fusable() and x.shape[0] >= self.fused_min_batch:
            flat = []
            for layer in self.kan_layers:
                flat += [layer.spline_weights, layer.knots, layer.linear.weight, layer.linear.bias]
            prep = self._prepared()
            # from mfma_min_batch samples up the dense form on the matrix cores (rovit_kan_stack_fwd_mfma) is the faster one
            mfma = x.shape[0] >= self.mfma_min_batch and all(p[2] is not None for p in prep)
            return [x, *KANStackFn.apply(x, codes, prep, 2 if mfma else 1, *flat)]
        if self._fusable():
            # below that: one forward launch per layer (they split the input features over threads), but still ONE autograd node
            # whose backward is the two-launch rovit_kan_stack_bwd instead of a dx + dW launch pair per layer
            flat = []
            for layer in self.kan_layers:
                flat += [layer.spline_weights, layer.knots, layer.linear.weight, layer.linear.bias]
            return [x, *KANStackFn.apply(x, codes, None, 0, *flat)]
        acts = [x]
        for i, layer in enumerate(self.kan_layers):      # activation fused into the layer kernel
            x = layer._run(x, codes[i])
            acts.append(x)
        return acts

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._trajectory(x)[-1]

    def get_spline_weights(self) -> List[torch.Tensor]:
        return [layer.get_spline_weights() for layer in self.kan_layers]

    def get_activation_trajectory(self, x: torch.Tensor) -> List[torch.Tensor]:
        return self._trajectory(x)

    def count_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
