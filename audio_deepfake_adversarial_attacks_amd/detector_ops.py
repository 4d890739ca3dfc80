"""Fused elementwise / pooling ops of the SpecRNet and RawNet3 detectors as differentiable functions backed by the HIP
kernels of include/advstep_detector.h (csrc/detector_elem.hip).  Input gradients only for the folded parameters' part:
the per-channel constants (conv biases, eval-mode BatchNorm statistics and affine terms) are treated as constants, which
is what an attack needs — `Attack.__call__` freezes the attacked model's parameters, and the models take these paths only
then (otherwise they run their plain torch modules).  HIP tensors only."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from .hip_ops import _Launch, _require, _stream

MODE_AFFINE_LRELU, MODE_RELU_AFFINE = 0, 1


def bn_eval_affine(bn: torch.nn.modules.batchnorm._BatchNorm) -> Tuple[torch.Tensor, torch.Tensor]:
    """(scale, shift) with  bn(x) = x * scale + shift  for an eval-mode BatchNorm with running statistics; cached on the
    module until a parameter or buffer changes."""
    tensors = [bn.running_mean, bn.running_var] + ([bn.weight, bn.bias] if bn.affine else [])
    key = tuple((t.data_ptr(), t._version) for t in tensors)
    if getattr(bn, "_advstep_affine_key", None) != key:
        with torch.no_grad():
            invstd = torch.rsqrt(bn.running_var + bn.eps)
            scale = invstd * bn.weight if bn.affine else invstd
            shift = (bn.bias if bn.affine else 0.0) - bn.running_mean * scale
        bn._advstep_affine_key, bn._advstep_affine = key, (scale.contiguous(), shift.contiguous())
    return bn._advstep_affine


def foldable_bn(bn) -> bool:
    return (isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and not bn.training and bn.track_running_stats
            and bn.running_mean is not None)


class _AffineAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, pre, mode, slope):
        _require(x, "x"), _require(scale, "scale"), _require(shift, "shift")
        N, C = x.shape[0], x.shape[1]
        P = x[0, 0].numel()
        if scale.numel() != C or shift.numel() != C or (pre is not None and pre.numel() != C):
            raise ValueError("per-channel constants must have one entry per channel")
        y = torch.empty_like(x)
        with _Launch("affine_act_forward", x.device):
            st = _lib.load().advstep_affine_act_forward_f32(x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                            pre.data_ptr() if pre is not None else None, y.data_ptr(), N, C, P,
                                                            mode, slope, _stream(x.device))
        _lib.check(st, "advstep_affine_act_forward_f32")
        ctx.save_for_backward(x, scale, shift, *([pre] if pre is not None else []))
        ctx.meta = (N, C, P, mode, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, shift, *pre = ctx.saved_tensors
        N, C, P, mode, slope = ctx.meta
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        with _Launch("affine_act_backward", x.device):
            st = _lib.load().advstep_affine_act_backward_f32(gy.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                             pre[0].data_ptr() if pre else None, gx.data_ptr(), N, C, P, mode,
                                                             slope, _stream(x.device))
        _lib.check(st, "advstep_affine_act_backward_f32")
        return gx, None, None, None, None, None


def affine_lrelu(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, slope: float) -> torch.Tensor:
    """leaky_relu(x * scale[c] + shift[c], slope) over (N, C, ...)."""
    return _AffineAct.apply(x.contiguous(), scale, shift, None, MODE_AFFINE_LRELU, float(slope))


def relu_affine(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu(x + pre[c]) * scale[c] + shift[c] over (N, C, ...)."""
    return _AffineAct.apply(x.contiguous(), scale, shift, pre, MODE_RELU_AFFINE, 0.0)


class _AddMaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, bias):
        _require(a, "a")
        if a.dim() != 4:
            raise ValueError(f"expected (N, C, H, W), got {tuple(a.shape)}")
        if b is not None:
            _require(b, "b")
            if b.shape != a.shape:
                raise ValueError("a and b must have the same shape")
        N, C, H, W = a.shape
        y = torch.empty((N, C, H // 2, W // 2), dtype=a.dtype, device=a.device)
        sel = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=a.device)
        with _Launch("add_maxpool2_forward", a.device):
            st = _lib.load().advstep_add_maxpool2_forward_f32(a.data_ptr(), b.data_ptr() if b is not None else None,
                                                              bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                                              sel.data_ptr(), N, C, H, W, _stream(a.device))
        _lib.check(st, "advstep_add_maxpool2_forward_f32")
        ctx.save_for_backward(sel)
        ctx.shape = (N, C, H, W)
        ctx.two = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        (sel,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        g = torch.empty((N, C, H, W), dtype=gy.dtype, device=gy.device)
        with _Launch("maxpool2_backward", gy.device):
            st = _lib.load().advstep_maxpool2_backward_f32(gy.data_ptr(), sel.data_ptr(), g.data_ptr(), N, C, H, W,
                                                           _stream(gy.device))
        _lib.check(st, "advstep_maxpool2_backward_f32")
        return g, (g if ctx.two else None), None


def add_maxpool2(a: torch.Tensor, b: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MaxPool2d(2)(a + b + bias[c]); b and bias optional."""
    return _AddMaxPool2.apply(a.contiguous(), None if b is None else b.contiguous(), bias)


class _AddMaxPool1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, k):
        _require(a, "a")
        if a.dim() != 3:
            raise ValueError(f"expected (N, C, L), got {tuple(a.shape)}")
        if b is not None:
            _require(b, "b")
            if b.shape != a.shape:
                raise ValueError("a and b must have the same shape")
        N, C, L = a.shape
        y = torch.empty((N, C, L // k), dtype=a.dtype, device=a.device)
        sel = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=a.device)
        with _Launch("add_maxpool1d_forward", a.device):
            st = _lib.load().advstep_add_maxpool1d_forward_f32(a.data_ptr(), b.data_ptr() if b is not None else None, y.data_ptr(),
                                                               sel.data_ptr(), N, C, L, k, _stream(a.device))
        _lib.check(st, "advstep_add_maxpool1d_forward_f32")
        ctx.save_for_backward(sel)
        ctx.meta = (N, C, L, k, b is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        (sel,) = ctx.saved_tensors
        N, C, L, k, two = ctx.meta
        gy = gy.contiguous()
        g = torch.empty((N, C, L), dtype=gy.dtype, device=gy.device)
        with _Launch("maxpool1d_backward", gy.device):
            st = _lib.load().advstep_maxpool1d_backward_f32(gy.data_ptr(), sel.data_ptr(), g.data_ptr(), N, C, L, k,
                                                            _stream(gy.device))
        _lib.check(st, "advstep_maxpool1d_backward_f32")
        return g, (g if two else None), None


def add_maxpool1d(a: torch.Tensor, b: Optional[torch.Tensor], k: int) -> torch.Tensor:
    """MaxPool1d(k)(a + b) over (N, C, L), kernel = stride = k in 2..8; b optional."""
    return _AddMaxPool1d.apply(a.contiguous(), None if b is None else b.contiguous(), int(k))


def maxpool1d_supported(pool) -> bool:
    """nn.MaxPool1d with kernel = stride in 2..8, no padding / dilation / ceil mode / indices."""
    k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
    st = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
    pad = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
    dil = pool.dilation if isinstance(pool.dilation, int) else pool.dilation[0]
    return 2 <= k <= 8 and st == k and pad == 0 and dil == 1 and not pool.ceil_mode and not pool.return_indices


class _GateMaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate):
        _require(x, "x"), _require(gate, "gate")
        N, C, H, W = x.shape
        if gate.numel() != N * C:
            raise ValueError("gate must hold one value per (sample, channel)")
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        sel = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=x.device)
        with _Launch("gate_maxpool2_forward", x.device):
            st = _lib.load().advstep_gate_maxpool2_forward_f32(x.data_ptr(), gate.data_ptr(), y.data_ptr(), sel.data_ptr(), N, C,
                                                               H, W, _stream(x.device))
        _lib.check(st, "advstep_gate_maxpool2_forward_f32")
        ctx.save_for_backward(sel, x, gate)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        sel, x, gate = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        lib = _lib.load()
        blocks = max(lib.advstep_gate_maxpool2_blocks(H, W), 1)
        gx = torch.empty_like(x)
        partial = torch.empty((N * C, blocks), dtype=x.dtype, device=x.device)
        with _Launch("gate_maxpool2_backward", x.device):
            st = lib.advstep_gate_maxpool2_backward_f32(gy.data_ptr(), sel.data_ptr(), x.data_ptr(), gate.data_ptr(),
                                                        gx.data_ptr(), partial.data_ptr(), N, C, H, W, _stream(x.device))
        _lib.check(st, "advstep_gate_maxpool2_backward_f32")
        return gx, partial.sum(dim=1).view_as(gate)


def gate_maxpool2(x: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """MaxPool2d(2)(x * gate[n, c] + gate[n, c]), gate (N, C) or (N, C, 1, 1); differentiable in x and gate."""
    return _GateMaxPool2.apply(x.contiguous(), gate.contiguous())
