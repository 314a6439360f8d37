"""Tube masks for the VideoMAE pre-training path: one spatial keep / drop pattern per clip, shared by every temporal slot.

Behavioural contract (reference masking_generator.py:17-24, pinned by tests/golden/g6_tube_mask.npz and g10_vitl_mae.npz): a
float64 vector of ``H' * W' - k`` zeros followed by ``k = int(ratio * H' * W')`` ones is shuffled in place by ONE call of numpy's
generator, then repeated T' times; 1 = masked.  The draw is a function here (``tube_mask``) so that callers can own the RNG; the
class keeps the reference's constructor / call / attribute surface and its use of numpy's GLOBAL state.
"""
from __future__ import annotations

import numpy as np


def masked_per_slot(grid_hw: int, mask_ratio: float) -> int:
    return int(mask_ratio * grid_hw)


def tube_mask(slots: int, grid_hw: int, n_masked: int, shuffle=None) -> np.ndarray:
    """flat [slots * grid_hw] float64 mask; ``shuffle`` defaults to ``np.random.shuffle`` (the global stream the reference uses)"""
    pattern = np.zeros(grid_hw, dtype=np.float64)
    pattern[grid_hw - n_masked:] = 1.0
    (np.random.shuffle if shuffle is None else shuffle)(pattern)
    return np.broadcast_to(pattern, (slots, grid_hw)).reshape(-1).copy()


class TubeMaskingGenerator:
    """``TubeMaskingGenerator((T', H', W'), ratio)()`` -> mask; attributes as read by the reference's data pipeline."""

    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = (int(v) for v in input_size)
        self.num_patches_per_frame = self.height * self.width
        self.num_masks_per_frame = masked_per_slot(self.num_patches_per_frame, mask_ratio)
        self.total_patches = self.frames * self.num_patches_per_frame
        self.total_masks = self.frames * self.num_masks_per_frame

    def __call__(self) -> np.ndarray:
        return tube_mask(self.frames, self.num_patches_per_frame, self.num_masks_per_frame)

    def __repr__(self) -> str:
        return f"TubeMaskingGenerator(total patches {self.total_patches}, masked {self.total_masks})"
