"""The four tf.image colour adjustments the reference's augmentation draws from
(dataloaders/generic.py:186-206), as elementwise torch ops on [..., H, W, 3] float tensors."""
import torch


def adjust_brightness(im, delta):
    """tf.image.adjust_brightness: add delta."""
    return im + delta


def adjust_contrast(im, factor):
    """tf.image.adjust_contrast: (x - mean) * factor + mean, mean per image and channel."""
    mean = im.mean(dim=(-3, -2), keepdim=True)
    return (im - mean) * factor + mean


def rgb_to_hsv(im):
    r, g, b = im.unbind(-1)
    mx = torch.maximum(torch.maximum(r, g), b)
    mn = torch.minimum(torch.minimum(r, g), b)
    rng = mx - mn
    safe = torch.where(rng > 0, rng, torch.ones_like(rng))
    s = torch.where(mx > 0, rng / torch.where(mx > 0, mx, torch.ones_like(mx)), torch.zeros_like(mx))
    hr = ((g - b) / safe) / 6.0
    hg = (2.0 + (b - r) / safe) / 6.0
    hb = (4.0 + (r - g) / safe) / 6.0
    h = torch.where(mx == r, hr, torch.where(mx == g, hg, hb))
    h = torch.where(rng > 0, h, torch.zeros_like(h))
    h = torch.where(h < 0, h + 1.0, h)
    return torch.stack([h, s, mx], dim=-1)


def hsv_to_rgb(hsv):
    h, s, v = hsv.unbind(-1)
    dh = h * 6.0
    dr = torch.clamp(torch.abs(dh - 3.0) - 1.0, 0.0, 1.0)
    dg = torch.clamp(2.0 - torch.abs(dh - 2.0), 0.0, 1.0)
    db = torch.clamp(2.0 - torch.abs(dh - 4.0), 0.0, 1.0)
    one_minus_s = 1.0 - s
    return torch.stack([(one_minus_s + s * dr) * v, (one_minus_s + s * dg) * v, (one_minus_s + s * db) * v], dim=-1)


def adjust_saturation(im, factor):
    """tf.image.adjust_saturation: scale S in HSV, clipped to [0,1]."""
    hsv = rgb_to_hsv(im)
    h, s, v = hsv.unbind(-1)
    return hsv_to_rgb(torch.stack([h, torch.clamp(s * factor, 0.0, 1.0), v], dim=-1))


def adjust_hue(im, delta):
    """tf.image.adjust_hue: rotate H by delta (in [-1,1] of a full turn)."""
    hsv = rgb_to_hsv(im)
    h, s, v = hsv.unbind(-1)
    h = torch.remainder(h + delta, 1.0)
    return hsv_to_rgb(torch.stack([h, s, v], dim=-1))
