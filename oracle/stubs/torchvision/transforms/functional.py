"""torchvision.transforms.functional.gaussian_blur restated from its documented algorithm
(reference call site: src/utils/event_image_converter.py:175), and `resize` for float tensors, which
torchvision forwards to torch.nn.functional.interpolate (reference call site: src/utils/flow.py:9-10).
Third-party stand-in."""
import torch
import torch.nn.functional as F


def _k1d(k, s, dtype, device):
    lim = (k - 1) * 0.5
    x = torch.linspace(-lim, lim, steps=k, dtype=dtype, device=device)
    pdf = torch.exp(-0.5 * (x / s).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(img, kernel_size, sigma=None):
    ks = [kernel_size] * 2 if isinstance(kernel_size, int) else list(kernel_size)
    sg = [float(sigma)] * 2 if isinstance(sigma, (int, float)) else list(sigma)
    kx = _k1d(ks[0], sg[0], img.dtype, img.device)
    ky = _k1d(ks[1], sg[1], img.dtype, img.device)
    k2 = torch.mm(ky[:, None], kx[None, :])
    C = img.shape[-3]
    x = F.pad(img, [ks[0] // 2, ks[0] // 2, ks[1] // 2, ks[1] // 2], mode='reflect')
    return F.conv2d(x, k2.expand(C, 1, *k2.shape), groups=C)


def resize(img, size, interpolation=None, max_size=None, antialias=True):
    """Float tensors [..., C, H, W]: interpolate(size, mode, align_corners=False, antialias), no clamping."""
    mode = getattr(interpolation, 'value', interpolation) or 'bilinear'
    lead = img.shape[:-3]
    x = img.reshape((-1,) + tuple(img.shape[-3:]))
    ac = False if mode in ('bilinear', 'bicubic') else None
    aa = bool(antialias) and mode in ('bilinear', 'bicubic')
    y = F.interpolate(x, size=list(size), mode=mode, align_corners=ac, antialias=aa)
    return y.reshape(lead + tuple(y.shape[-3:]))
