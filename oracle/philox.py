"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) restated with numpy.

Test infrastructure (like the rest of oracle/): the dropout masks drawn INSIDE the fused head-phase kernel
(csrc/head_phase.hip: philox4x32_10, key = seed, counter = (sample * hid + unit, 0, offset lo, offset hi)) are checked against this
restatement on the GPU (tests/test_gpu_head_phase.py), and this restatement against the published known-answer vectors of the Random123
distribution on the CPU (tests/test_oracle.py) -- the reference itself draws its masks with torch.nn.Dropout and holds no vectors for them.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: four arrays / scalars of 32-bit words, key: two.  Returns the four output words as uint64 arrays holding 32-bit values."""
    c = [np.asarray(x, dtype=np.uint64) & _MASK for x in counter]
    k0, k1 = (np.asarray(x, dtype=np.uint64) & _MASK for x in key)
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK, p1 >> np.uint64(32), p1 & _MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + _W0) & _MASK
        k1 = (k1 + _W1) & _MASK
    return c


def head_phase_masks(batch, hid, p, seed, offset):
    """The three heads' scaled keep-masks (batch, hid) as the kernel draws them: words x / y / z of the output serve heads 0 / 1 / 2; a unit is
    kept iff (word >> 8) * 2^-24 < 1 - p (fp32), kept units carry 1 / (1 - p)."""
    n = batch * hid
    out = philox4x32_10([np.arange(n, dtype=np.uint64), np.zeros(n, np.uint64), np.full(n, offset & 0xFFFFFFFF, np.uint64),
                         np.full(n, (offset >> 32) & 0xFFFFFFFF, np.uint64)], [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF])
    keep = np.float32(1.0) - np.float32(p)
    masks = []
    for w in out[:3]:
        u = (w >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        masks.append(np.where(u < keep, np.float32(1.0) / keep, np.float32(0.0)).astype(np.float32).reshape(batch, hid))
    return masks
