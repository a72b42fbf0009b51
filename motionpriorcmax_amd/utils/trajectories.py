"""Tile mask and coefficient-grid -> list helpers (host side, plain torch on the caller's
device; tiny tensors, autograd handles their backward).  Mirrors reference
src/utils/trajectories.py:3-52."""
import torch


def get_optical_flow_tile_mask(image_shape, tile_size):
    """One trajectory per tile_size x tile_size tile, at offset tile_size // 2."""
    mask = torch.zeros(tuple(image_shape), dtype=torch.bool)
    s = tile_size // 2
    mask[s::tile_size, s::tile_size] = True
    return mask


def coeffs_grid_to_list(coeff_grid, mask, num_coeffs):
    """coeff_grid [b, s, 2k, h, w] (first k channels: y, next k: x), mask [h, w] ->
    (coeffs [b, s, 2, n, k], pixel_positions [n, 2] (y, x), orig_shape)."""
    orig_shape = coeff_grid.shape
    b, s, c2, h, w = orig_shape
    assert c2 == 2 * num_coeffs
    pixel_positions = torch.nonzero(mask)
    sel = coeff_grid.reshape(b, s, c2, h * w)[..., mask.reshape(-1)]
    coeffs = sel.reshape(b, s, 2, num_coeffs, -1).permute(0, 1, 2, 4, 3)
    return coeffs, pixel_positions, orig_shape
