"""Oracle (CPU, test infrastructure only): ClassMix / cross-domain mixture.

Restates the inline blocks at
  * G5/train_DiGA_gta2city_warm_up.py:240-259          (image paste)
  * G5/train_DiGA_gta2city_self_training.py:306-325    (image + label paste)
SURVEY App. A-8: classes present are listed ascending (torch.unique), half of
them are drawn with ONE call of Python's global `random.sample` per image, 255
is always added, the mask is broadcast over the 3 channels.
Pinned by tests/golden (G-classmix, KAT-6).
"""
import random

import numpy as np
import torch

IGNORE = 255


def classes_present(label_img):
    """Ascending list of the distinct label values of one image (== torch.unique().tolist())."""
    return np.unique(label_img.cpu().numpy()).tolist()


def select_classes(present, rng=random):
    """warm_up.py:247-250: one random.sample call, then force-add 255."""
    sel = rng.sample(present, len(present) // 2)
    if IGNORE not in sel:
        sel.append(IGNORE)
    return sel


def class_mask(labels, selections):
    """mask[b,y,x] = 1.0 where labels[b,y,x] is in selections[b] (warm_up.py:251-252)."""
    mask = torch.zeros(labels.shape, dtype=torch.float32)
    for b, sel in enumerate(selections):
        lut = torch.zeros(256, dtype=torch.bool)
        lut[torch.tensor(sel, dtype=torch.int64)] = True
        mask[b] = lut[labels[b]].to(torch.float32)
    return mask


def paste(background, foreground, mask):
    """warm_up.py:253-259: out = background*(1-m) + foreground*m, m broadcast over C."""
    m = mask.unsqueeze(1)
    return background * (1.0 - m) + foreground * m


def paste_labels(bg_labels, fg_labels, mask):
    """self_training.py:318-319: label_out = fg label where mask else bg label."""
    return torch.where(mask > 0, fg_labels, bg_labels)


def classmix(background, foreground, fg_labels, rng=random, bg_labels=None):
    """Whole block.  Returns (mixed, mask, selections[, mixed_labels])."""
    sels = [select_classes(classes_present(fg_labels[b]), rng)
            for b in range(fg_labels.shape[0])]
    mask = class_mask(fg_labels, sels)
    mixed = paste(background, foreground, mask)
    if bg_labels is None:
        return mixed, mask, sels
    return mixed, mask, sels, paste_labels(bg_labels, fg_labels, mask)
