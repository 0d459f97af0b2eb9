"""N2 -- the test-time frame transform of the few-shot datasets on the GPU (the step before the hot path).

Reference: datasets/base/ssv2_few_shot.py:614-642 (test/val transform list) and
datasets/utils/transformations.py:663-716 (``KineticsResizedCropFewshot._get_controlled_crop``): ToTensorVideo ->
bilinear resize to TEST_SCALE (int -> square, [h, w] list -> that size) -> crop of TEST_CROP_SIZE selected by
``num_spatial_crops`` / ``idx`` -> NormalizeVideo(DATA.MEAN, DATA.STD) -> ``permute(1,0,2,3)`` to [T, 3, crop, crop]
(:439).  Decoding (decord) stays on the host; this takes the decoded uint8 frames.
"""
from __future__ import annotations

import torch

from . import hip


def crop_window(scale_hw, crop, num_spatial_crops=1, idx=1):
    """(y0, x0) of KineticsResizedCropFewshot._get_controlled_crop (transformations.py:689-716)."""
    sh, sw = int(scale_hw[0]), int(scale_hw[1])
    x_max, y_max = sw - int(crop), sh - int(crop)
    length = sh                                         # short_side_range[0]
    if num_spatial_crops == 1 or idx == 1:
        return y_max // 2, x_max // 2
    if num_spatial_crops == 3 and idx in (0, 2):
        if sw == length:
            return (0 if idx == 0 else y_max), x_max // 2
        if sh == length:
            return y_max // 2, (0 if idx == 0 else x_max)
    raise ValueError("unsupported spatial crop selection (num_spatial_crops=%r, idx=%r)" % (num_spatial_crops, idx))


def preprocess_video(frames_u8: torch.Tensor, test_scale, crop_size, mean, std, num_spatial_crops=1, idx=1) -> torch.Tensor:
    """frames_u8: uint8 device tensor [T, H, W, 3] -> fp32 [T, 3, crop, crop] on the same device."""
    scale_hw = (test_scale, test_scale) if isinstance(test_scale, int) else (test_scale[0], test_scale[1])
    y0, x0 = crop_window(scale_hw, crop_size, num_spatial_crops, idx)
    out = torch.empty(frames_u8.shape[0], 3, int(crop_size), int(crop_size), device=frames_u8.device, dtype=torch.float32)
    hip.preprocess_frames(frames_u8.contiguous(), out, scale_hw, crop_size, y0, x0, mean, std)
    return out
