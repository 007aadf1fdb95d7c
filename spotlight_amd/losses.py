"""The loss functions of spotlight/losses.py:18-244 as plain torch functions, for code that
imports them directly (custom training loops, tests).  The models of this package do NOT call
these: their forward, loss, backward and update are fused in the gfx950 kernels
(csrc/slk_kernels.h: slk_pair_loss, csrc/slk_bilinear.hip: k_adaptive_select / k_explicit_loss),
which follow the same formulas operation by operation.

Implicit-feedback losses take (positive_predictions, negative_predictions[, mask]); with a mask
(sequence models: mask = sequence != 0) the mean is over the unmasked entries only.
"""
import torch
import torch.nn.functional as F

from spotlight_amd.torch_utils import assert_no_grad


def _masked_mean(values, mask):
    if mask is None:
        return values.mean()
    weights = mask.float()
    return (values * weights).sum() / weights.sum()


def pointwise_loss(positive_predictions, negative_predictions, mask=None):
    """Logistic loss: (1 - sigmoid(pos)) + sigmoid(neg)   (losses.py:18-50)."""
    return _masked_mean((1.0 - torch.sigmoid(positive_predictions)) + torch.sigmoid(negative_predictions), mask)


def bpr_loss(positive_predictions, negative_predictions, mask=None):
    """Spotlight's BPR variant: 1 - sigmoid(pos - neg) -- not -log sigmoid   (losses.py:53-90)."""
    return _masked_mean(1.0 - torch.sigmoid(positive_predictions - negative_predictions), mask)


def hinge_loss(positive_predictions, negative_predictions, mask=None):
    """max(0, neg - pos + 1)   (losses.py:93-124)."""
    return _masked_mean(torch.clamp(negative_predictions - positive_predictions + 1.0, 0.0), mask)


def adaptive_hinge_loss(positive_predictions, negative_predictions, mask=None):
    """Hinge loss against the highest-scoring of several sampled negatives: negative_predictions has the
    candidates along dim 0   (losses.py:127-166)."""
    hardest, _ = negative_predictions.max(0)
    return hinge_loss(positive_predictions, hardest.squeeze(), mask=mask)


def regression_loss(observed_ratings, predicted_ratings):
    """Mean squared error   (losses.py:169-191)."""
    assert_no_grad(observed_ratings)
    return ((observed_ratings - predicted_ratings) ** 2).mean()


def poisson_loss(observed_ratings, predicted_ratings):
    """Poisson negative log-likelihood up to a constant: pred - obs * log(pred)   (losses.py:194-216)."""
    assert_no_grad(observed_ratings)
    return (predicted_ratings - observed_ratings * torch.log(predicted_ratings)).mean()


def logistic_loss(observed_ratings, predicted_ratings):
    """Binary cross-entropy with logits; observed ratings are -1 / +1   (losses.py:219-244)."""
    assert_no_grad(observed_ratings)
    targets = torch.clamp(observed_ratings, 0, 1)
    return F.binary_cross_entropy_with_logits(predicted_ratings, targets, reduction='mean')
