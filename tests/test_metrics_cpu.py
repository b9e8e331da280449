"""oracle/metrics.py (the restatement of torchmetrics' SSIM the GPU tests check tn_ssim_fwd against) against an independent
float64 computation: separable gaussian filtering with scipy.ndimage on the un-padded image, cropped to the windows that lie
inside it (torchmetrics pads by reflection and crops the same margin again, so the padding never reaches the result)."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import metrics as M


def ssim_float64(p: np.ndarray, t: np.ndarray) -> float:
    """p, t [C,H,W] float64."""
    g = np.exp(-((np.arange(11) - 5) / 1.5) ** 2 / 2)
    g /= g.sum()

    def filt(x):
        return ndimage.correlate1d(ndimage.correlate1d(x, g, axis=-1, mode="constant"), g, axis=-2, mode="constant")[..., 5:-5, 5:-5]

    data_range = max(p.max() - p.min(), t.max() - t.min())
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mp, mt = filt(p), filt(t)
    sp = np.maximum(filt(p * p) - mp * mp, 0)
    st = np.maximum(filt(t * t) - mt * mt, 0)
    spt = filt(p * t) - mp * mt
    idx = ((2 * mp * mt + c1) * (2 * spt + c2)) / ((mp * mp + mt * mt + c1) * (sp + st + c2))
    return float(idx.mean())


@pytest.mark.parametrize("shape", [(1, 11, 11), (3, 23, 17), (1, 64, 48), (3, 40, 90)])
def test_oracle_ssim_matches_float64_filtering(shape):
    g = torch.Generator().manual_seed(sum(shape))
    t = torch.rand(shape, generator=g)
    p = (t + 0.1 * torch.randn(shape, generator=g)).clamp(0, 1)
    got = float(M.ssim(p[None], t[None]))
    want = ssim_float64(p.double().numpy(), t.double().numpy())
    assert abs(got - want) <= 2e-6, (got, want)
    assert 0.0 < got < 1.0


def test_oracle_ssim_properties():
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(1, 3, 32, 40, generator=g), torch.rand(1, 3, 32, 40, generator=g)
    assert float(M.ssim(a, a)) == pytest.approx(1.0, abs=1e-6)
    assert float(M.ssim(a, b)) == pytest.approx(float(M.ssim(b, a)), abs=1e-6)  # symmetric
    assert float(M.ssim(a, b)) < 0.2                                            # independent noise
    # smooth structure, small perturbation: close to 1
    y, x = torch.meshgrid(torch.linspace(0, 1, 32), torch.linspace(0, 1, 40), indexing="ij")
    s = (0.5 + 0.5 * torch.sin(6 * x) * torch.cos(4 * y))[None, None]
    assert float(M.ssim(s, (s + 0.01 * torch.randn(s.shape, generator=g)).clamp(0, 1))) > 0.9
    assert float(M.psnr(a, a + 0.1)) == pytest.approx(20.0, abs=1e-4)
