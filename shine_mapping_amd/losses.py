"""utils/loss.py:17-24 and utils/tools.py:175-185 under their reference names (torch composites, Tier A)."""
import torch
import torch.nn as nn
from torch.autograd import grad


def sdf_bce_loss(pred, label, sigma, weight, weighted=False, bce_reduction="mean"):
    """utils/loss.py:17-24"""
    loss_bce = nn.BCEWithLogitsLoss(reduction=bce_reduction, weight=weight if weighted else None)
    return loss_bce(pred, torch.sigmoid(label / sigma))


def get_gradient(inputs, outputs):
    """utils/tools.py:175-185"""
    d_points = torch.ones_like(outputs, requires_grad=False, device=outputs.device)
    return grad(outputs=outputs, inputs=inputs, grad_outputs=d_points, create_graph=True, retain_graph=True,
                only_inputs=True)[0]
