"""Host-side helpers of the path (numpy, no kernels): mirrors pipeline/train_utils.py of the reference.

  EarlyStopping  pipeline/train_utils.py:8-60   (np.Inf there breaks on NumPy 2; np.inf here)
  zscore         pipeline/train_utils.py:228-250
  zscore_patch   pipeline/train_utils.py:252-274
"""
import os

import numpy as np
import torch


class EarlyStopping:
    """Stop when the validation loss has not improved for `patience` calls; checkpoint on improvement."""

    def __init__(self, patience=7, verbose=False, delta=0, path='checkpoint.pt', trace_func=print):
        self.patience = patience
        self.verbose = verbose
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = np.inf
        self.delta = delta
        self.path = path
        self.trace_func = trace_func
        self.writes = True

    def __call__(self, val_loss, model):
        score = -val_loss
        if self.best_score is None or score >= self.best_score + self.delta:
            if self.best_score is not None:
                self.counter = 0
            self.best_score = score
            self.save_checkpoint(val_loss, model)
            return
        self.counter += 1
        self.trace_func(f'EarlyStopping counter: {self.counter} out of {self.patience}')
        if self.counter >= self.patience:
            self.early_stop = True

    def save_checkpoint(self, val_loss, model):
        """Weights only (state_dict), exactly what process_VAE later loads as <weights>/model.pt.  `writes` (set False on
        the non-zero ranks of a data-parallel run) keeps the bookkeeping and skips the file; the file appears atomically
        (temporary name + os.replace), so a reader never sees a torn checkpoint."""
        if self.verbose:
            self.trace_func(f'Validation loss decreased ({self.val_loss_min:.6f} --> {val_loss:.6f}).  Saving model ...')
        if self.writes:
            tmp = str(self.path) + '.tmp'
            torch.save(model.state_dict(), tmp)
            os.replace(tmp, self.path)
        self.val_loss_min = val_loss


def zscore(input_image, channel_mean=None, channel_std=None):
    """Dataset-wide per-channel z-score of an (N, C, H, W) array; eps in the denominator."""
    input_image = np.asarray(input_image)
    if not channel_mean:
        channel_mean = np.mean(input_image, axis=(0, 2, 3))
    if not channel_std:
        channel_std = np.std(input_image, axis=(0, 2, 3))
    eps = np.finfo(float).eps
    norm_img = np.stack([(input_image[:, c, ...] - channel_mean[c]) / (channel_std[c] + eps)
                         for c in range(len(channel_mean))], 1)
    print('channel_mean:', channel_mean)
    print('channel_std:', channel_std)
    return norm_img


def zscore_patch(imgs):
    """Per-patch, per-channel z-score over H x W (population std)."""
    imgs = np.asarray(imgs)
    means = np.mean(imgs, axis=(2, 3), keepdims=True)
    stds = np.std(imgs, axis=(2, 3), keepdims=True)
    return (imgs - means) / (stds + np.finfo(float).eps)
