"""Dense losses and metrics with the reference's names and signatures (/root/reference/utils/loss.py),
each one a fused gfx950 kernel.  All reductions are SUMs over the batch, as in the reference.

The sparse / legacy functions of the reference (get_bce, get_focal, get_acc, *_legacy) work on
MinkowskiEngine tensors for an encoder-side print only; they are out of scope.
"""
import torch

from . import functional as NF
from . import ops


def _same_shape(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)


def get_focal_dense(data, groud_truth, alpha=0.97, gamma=2):
    """sum -a_t (1 - p_t)^2 ln p_t  (loss.py:61-72)."""
    _same_shape(data, groud_truth)
    if gamma != 2:
        raise NotImplementedError("the fused focal kernel implements gamma = 2")
    return NF.FocalLoss.apply(data, groud_truth, None, float(alpha), 0.0)


def get_surf_focal_dense(data, groud_truth, dist, beta=1, alpha=0.97, gamma=2):
    """Focal term weighted by dist + gt * beta  (loss.py:94-111)."""
    _same_shape(data, groud_truth)
    if gamma != 2:
        raise NotImplementedError("the fused focal kernel implements gamma = 2")
    return NF.FocalLoss.apply(data, groud_truth, dist, float(alpha), float(beta))


def get_acc_dense(data, groud_truth, thh=0.5, alpha=0.9, gamma=2):
    """(true-positive rate, true-negative rate) at threshold thh  (loss.py:74-84)."""
    m = ops.metrics(data.detach().contiguous(), groud_truth.contiguous(), None, thh, thh)
    return m[0] / m[1], m[2] / m[3]


def get_sse1(data, groud_truth, dist, thh, maxv=1023):
    """(sum ((p > thh) * dist)^2, count(p > thh))  (loss.py:113-121)."""
    m = ops.metrics(data.detach().contiguous(), groud_truth.contiguous(), dist.contiguous(), thh, thh)
    return m[4], m[5]


def get_se(data, dist, thh):
    """[B,2,...]: per-voxel squared predicted distance stacked with p  (loss.py:123-128)."""
    return ops.squared_error_map(data.detach().contiguous(), dist.contiguous(), thh)


def get_surf_dual_dense(data, ground_truth, dist, beta=1):
    raise NotImplementedError("get_surf_dual_dense is reachable only by editing NVFPCC.py's module constant "
                              "main_loss (NVFPCC.py:27,177-188); the codec uses 'wfocal'")


def get_surface_loss_dense(data, groud_truth, dist, alpha=1):
    raise NotImplementedError("unused by NVFPCC.py")
