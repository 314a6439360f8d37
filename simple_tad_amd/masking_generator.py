"""Tube masking for VideoMAE pre-training (masking_generator.py:3-26): ONE random frame mask of ``int(ratio * H' * W')`` ones,
repeated over the temporal axis.  Host-side integer bookkeeping (numpy RNG, as in the reference: ``np.random.shuffle``)."""
import numpy as np


class TubeMaskingGenerator:
    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = input_size
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.frames * self.num_patches_per_frame
        self.num_masks_per_frame = int(mask_ratio * self.num_patches_per_frame)
        self.total_masks = self.frames * self.num_masks_per_frame

    def __repr__(self):
        return "Maks: total patches {}, mask patches {}".format(self.total_patches, self.total_masks)

    def __call__(self):
        mask_per_frame = np.hstack([np.zeros(self.num_patches_per_frame - self.num_masks_per_frame),
                                    np.ones(self.num_masks_per_frame)])
        np.random.shuffle(mask_per_frame)
        return np.tile(mask_per_frame, (self.frames, 1)).flatten()
