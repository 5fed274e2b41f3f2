"""
The reference's loss callables (``collie/loss/*.py``), same names and arguments, computed by the
``collie_loss_forward_backward`` HIP kernel.  They are differentiable w.r.t. both score tensors.

The fused training step (``MatrixFactorizationModel.calculate_loss``) does not go through these:
it evaluates the same formulas inside the gather kernel.  These stand-alone functions exist for
direct callers (``docs/source/loss.rst``) and for the generic, un-fused route.
"""
from typing import Any, Dict, Optional

import torch

from collie_amd import ops


def metadata_to_codes(metadata: Optional[Dict[str, torch.Tensor]],
                      metadata_weights: Optional[Dict[str, float]], device):
    """
    Categorical metadata -> int32 code arrays on ``device`` (equality-preserving), plus the weights
    in the same order.  Raises as ``ideal_difference_from_metadata`` does when the weights sum to
    more than 1 (``collie/loss/metadata_utils.py:60-62``).
    """
    if metadata is None or len(metadata) == 0:
        return [], []
    metadata_weights = metadata_weights or {}
    weight_sum = sum(metadata_weights.values())
    if weight_sum > 1:
        raise ValueError(f'sum of metadata weights was {weight_sum}, must be <=1')
    codes, weights = [], []
    for key, arr in metadata.items():
        arr = torch.as_tensor(arr).reshape(-1)
        _, inverse = torch.unique(arr.cpu(), return_inverse=True)
        codes.append(inverse.to(torch.int32).to(device).contiguous())
        weights.append(float(metadata_weights[key]))
    return codes, weights


class _PairwiseLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s_pos, s_neg_KB, kind, num_items, pos_items, neg_items_KB, codes, weights):
        cfg = ops.make_loss_cfg(kind, num_items if num_items is not None else 0, codes, weights)
        want_grad = s_pos.requires_grad or s_neg_KB.requires_grad
        loss, g_pos, g_neg = ops.loss_forward_backward(
            cfg, s_pos.detach(), s_neg_KB.detach(), pos_items if codes else None,
            neg_items_KB if codes else None, want_grad=True)
        ctx.save_for_backward(g_pos, g_neg)
        ctx.neg_shape = s_neg_KB.shape
        del want_grad
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        g_pos, g_neg = ctx.saved_tensors
        return grad_out * g_pos, (grad_out * g_neg).reshape(ctx.neg_shape), None, None, None, \
            None, None, None


def _call(kind, positive_scores, negative_scores, num_items, positive_items, negative_items,
          metadata, metadata_weights):
    s_neg = negative_scores if negative_scores.dim() == 2 else negative_scores.reshape(1, -1)
    codes, weights = metadata_to_codes(metadata, metadata_weights, positive_scores.device)
    pos_items = neg_items = None
    if codes:
        pos_items = positive_items.reshape(-1).to(positive_scores.device)
        neg_items = negative_items.to(positive_scores.device).reshape(s_neg.shape)
    out = _PairwiseLoss.apply(positive_scores, s_neg, kind, num_items, pos_items, neg_items, codes,
                              weights)
    return out


def bpr_loss(positive_scores: torch.Tensor,
             negative_scores: torch.Tensor,
             num_items: Optional[Any] = None,
             positive_items: Optional[torch.Tensor] = None,
             negative_items: Optional[torch.Tensor] = None,
             metadata: Optional[Dict[str, torch.Tensor]] = dict(),
             metadata_weights: Optional[Dict[str, float]] = dict()) -> torch.Tensor:
    """``collie/loss/bpr.py:8-89``: ``l = c - sigmoid(s+ - s-)``, ``(sum l + sum l^2) / B``."""
    return _call('bpr', positive_scores, negative_scores, num_items, positive_items,
                 negative_items, metadata, metadata_weights)


def hinge_loss(positive_scores: torch.Tensor,
               negative_scores: torch.Tensor,
               num_items: Optional[Any] = None,
               positive_items: Optional[torch.Tensor] = None,
               negative_items: Optional[torch.Tensor] = None,
               metadata: Optional[Dict[str, torch.Tensor]] = dict(),
               metadata_weights: Optional[Dict[str, float]] = dict()) -> torch.Tensor:
    """``collie/loss/hinge.py:8-85``: ``l = max(c - (s+ - s-), 0)``."""
    return _call('hinge', positive_scores, negative_scores, num_items, positive_items,
                 negative_items, metadata, metadata_weights)


def adaptive_bpr_loss(positive_scores: torch.Tensor,
                      many_negative_scores: torch.Tensor,
                      num_items: Optional[Any] = None,
                      positive_items: Optional[torch.Tensor] = None,
                      negative_items: Optional[torch.Tensor] = None,
                      metadata: Optional[Dict[str, torch.Tensor]] = dict(),
                      metadata_weights: Optional[Dict[str, float]] = dict()) -> torch.Tensor:
    """``collie/loss/bpr.py:92-167``: BPR against the highest-scoring of K negatives."""
    return _call('adaptive_bpr', positive_scores, many_negative_scores, num_items, positive_items,
                 negative_items, metadata, metadata_weights)


def adaptive_hinge_loss(positive_scores: torch.Tensor,
                        many_negative_scores: torch.Tensor,
                        num_items: Optional[Any] = None,
                        positive_items: Optional[torch.Tensor] = None,
                        negative_items: Optional[torch.Tensor] = None,
                        metadata: Optional[Dict[str, torch.Tensor]] = dict(),
                        metadata_weights: Optional[Dict[str, float]] = dict()) -> torch.Tensor:
    """``collie/loss/hinge.py:88-171``: hinge against the highest-scoring of K negatives."""
    return _call('adaptive_hinge', positive_scores, many_negative_scores, num_items,
                 positive_items, negative_items, metadata, metadata_weights)


def warp_loss(positive_scores: torch.Tensor,
              many_negative_scores: torch.Tensor,
              num_items: int,
              positive_items: Optional[torch.Tensor] = None,
              negative_items: Optional[torch.Tensor] = None,
              metadata: Optional[Dict[str, torch.Tensor]] = dict(),
              metadata_weights: Optional[Dict[str, float]] = dict()) -> torch.Tensor:
    """``collie/loss/warp.py:8-154``: first violating negative, weight ``ln(num_items / tries)``."""
    return _call('warp', positive_scores, many_negative_scores, num_items, positive_items,
                 negative_items, metadata, metadata_weights)


LOSS_FUNCTIONS = {'bpr': bpr_loss, 'hinge': hinge_loss, 'adaptive_bpr': adaptive_bpr_loss,
                  'adaptive_hinge': adaptive_hinge_loss, 'warp': warp_loss}
