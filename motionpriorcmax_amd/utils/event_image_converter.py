"""`imager` attribute of FocusLoss: the subset of reference EventImageConverter that sits on
the path (src/utils/event_image_converter.py:45-74,134-176,333-391: tensor 'bilinear_vote' +
3x3 blur).  Count / polarity / numpy variants are not part of the CMax path and raise."""
from typing import Tuple, Union

import torch

from .. import ops


class EventImageConverter(object):
    def __init__(self, image_size: tuple, outer_padding: Union[int, Tuple[int, int]] = 0):
        if isinstance(outer_padding, (int, float)):
            self.outer_padding = (int(outer_padding), int(outer_padding))
        else:
            self.outer_padding = tuple(outer_padding)
        if self.outer_padding != (0, 0):
            raise NotImplementedError('outer_padding != 0 is not used on the CMax path')
        self.image_size = tuple(int(i) for i in image_size)

    def create_iwe(self, events, method: str = "bilinear_vote", sigma: int = 1, weight=1.0):
        """events [(b,) n, >=2] with columns (y, x, ...) -> [(b,) H, W]; a 2-dim result gets a
        leading batch axis, as in the reference (event_image_converter.py:72-73)."""
        if not isinstance(events, torch.Tensor):
            raise RuntimeError(f"Non-supported type of events. {type(events)}")
        if method != "bilinear_vote":
            raise NotImplementedError(f"{method = } is not implemented")
        ev = events if events.dim() == 3 else events[None]
        nimg, m, _ = ev.shape
        rows = torch.zeros((nimg, m, 6), dtype=torch.float32, device=ev.device)
        rows[..., :2] = ev[..., :2]
        unit = not isinstance(weight, torch.Tensor)
        if not unit:
            assert weight.shape == events.shape[:-1]
            rows[..., 5] = weight.reshape(nimg, m)
        img = ops.splat_events(self.image_size, rows, unit, sigma > 0)
        if unit and float(weight) != 1.0:
            img = img * float(weight)
        img = torch.squeeze(img)
        if img.dim() == 2:
            img = img[None]
        return img
