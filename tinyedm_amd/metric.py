"""sigma-weighted MSE (reference metric.py:8-49): state = (sum_i mean_j w_i d_ij^2, N)."""
import torch
from torch import Tensor

from . import ops


def _scaled(dD: Tensor, g: Tensor) -> Tensor:
    """dD * g -- without the launch when g is the marked unit gradient of LightningModule.backward (trainer.py)"""
    return dD if getattr(g, "_edm_unit", False) else dD * g


class _WeightedMSEFn(torch.autograd.Function):
    """loss = (1/B) sum_i mean_j w_i (p_ij - t_ij)^2 ; gradient w.r.t. preds produced in the same pass."""

    @staticmethod
    def forward(ctx, weight, preds, target):
        loss, dD = ops.weighted_mse(preds, target, None, 0.0, weight=weight, want_grad=True)
        ctx.save_for_backward(dD)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dD,) = ctx.saved_tensors
        return None, _scaled(dD, g), None


class _SigmaMSEFn(torch.autograd.Function):
    """The same loss with the weight lambda(sigma) = (sigma^2 + sd^2) / (sigma sd)^2 (edm.py:212) evaluated inside the
    kernel and the metric's epoch state (sum, count: metric.py:38-49) accumulated in the same pass: one launch instead
    of the kernel plus ~10 scalar ATen launches (pow / add / mul / div on a B-vector, state += , count +=)."""

    @staticmethod
    def forward(ctx, sigma, sigma_data, preds, target, acc_sum, acc_total, want_grad):
        loss, dD = ops.weighted_mse(preds, target, sigma, sigma_data, want_grad=want_grad, acc_sum=acc_sum,
                                    acc_total=acc_total)
        ctx.save_for_backward(dD)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dD,) = ctx.saved_tensors
        return None, None, _scaled(dD, g), None, None, None, None


def weighted_mse_loss(weight: Tensor, preds: Tensor, target: Tensor) -> Tensor:
    preds32, target32 = preds.float().contiguous(), target.float().contiguous()
    w = weight.float().flatten().contiguous()
    if w.requires_grad:  # uncertainty branch (edm.py:213-219): d loss / d weight needed -> elementwise torch math
        n = target32.shape[0]
        d = (preds32 - target32).view(n, -1)
        return (w.view(n, 1) * d * d).mean(dim=1).sum() / n
    return _WeightedMSEFn.apply(w.detach(), preds32, target32)


def _weighted_sum_squared_error_update(weights: Tensor, preds: Tensor, target: Tensor):
    """metric.py:8-18 -> (sum_i mean_j w_i d_ij^2, N)."""
    n = target.shape[0]
    return weighted_mse_loss(weights, preds, target) * n, n


class WeightedMeanSquaredError(torch.nn.Module):
    """Stand-in for the reference's torchmetrics Metric (torchmetrics is not installed): ``forward``
    returns the differentiable batch value and accumulates the epoch state; ``compute`` returns
    state_sum / total, summed over ranks when torch.distributed is initialised (dist_reduce_fx="sum")."""

    is_differentiable = True
    higher_is_better = False
    full_state_update = False

    def __init__(self, **kwargs) -> None:
        super().__init__()
        self.register_buffer("weighted_sum_squared_error", torch.zeros(1), persistent=False)
        self.register_buffer("total", torch.tensor(0), persistent=False)

    def update(self, weight: Tensor, preds: Tensor, target: Tensor) -> None:
        s, n = _weighted_sum_squared_error_update(weight, preds, target)
        self.weighted_sum_squared_error += s.detach().to(self.weighted_sum_squared_error.device)
        self.total += n

    def forward(self, weight: Tensor, preds: Tensor, target: Tensor) -> Tensor:
        s, n = _weighted_sum_squared_error_update(weight, preds, target)
        self.weighted_sum_squared_error += s.detach().to(self.weighted_sum_squared_error.device)
        self.total += n
        return s / n

    def forward_sigma(self, sigma: Tensor, sigma_data: float, preds: Tensor, target: Tensor) -> Tensor:
        """`forward(lambda(sigma), preds, target)` (the call of edm.py:212,221 / 243) in one kernel launch: same value,
        same state update."""
        st, tot = self.weighted_sum_squared_error, self.total
        if not (preds.is_cuda and st.is_cuda and tot.is_cuda and tot.dtype == torch.int64 and tot.dim() == 0):
            w = (sigma ** 2 + sigma_data ** 2) / (sigma * sigma_data) ** 2
            return self.forward(w, preds, target)
        return _SigmaMSEFn.apply(sigma.detach().float().flatten().contiguous(), float(sigma_data),
                                 preds.float().contiguous(), target.float().contiguous(), st, tot,
                                 torch.is_grad_enabled() and preds.requires_grad)

    def compute(self) -> Tensor:
        s, t = self.weighted_sum_squared_error.clone(), self.total.clone().to(self.weighted_sum_squared_error.dtype)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            pack = torch.cat([s, t.view(1)])
            torch.distributed.all_reduce(pack)
            s, t = pack[:1], pack[1]
        return s / t

    def reset(self) -> None:
        self.weighted_sum_squared_error.zero_()
        self.total.zero_()
