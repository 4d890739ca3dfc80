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

MODE_AFFINE_LRELU, MODE_RELU_AFFINE, MODE_AFFINE_SELU = 0, 1, 2


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
        P = 1
        for d in x.shape[2:]:
            P *= d
        if scale.numel() != C or shift.numel() != C or (pre is not None and pre.numel() != C):
            raise ValueError("per-channel constants must have one entry per channel")
        y = torch.empty_like(x)
        with _Launch("affine_act_forward", x.device, tensors=(x, y)):
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
        with _Launch("affine_act_backward", x.device, tensors=(gy, x, gx)):
            st = _lib.load().advstep_affine_act_backward_f32(gy.data_ptr(), x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                             pre[0].data_ptr() if pre else None, gx.data_ptr(), N, C, P, mode,
                                                             slope, _stream(x.device))
        _lib.check(st, "advstep_affine_act_backward_f32")
        return gx, None, None, None, None, None


def affine_lrelu(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, slope: float) -> torch.Tensor:
    """leaky_relu(x * scale[c] + shift[c], slope) over (N, C, ...)."""
    return _AffineAct.apply(x.contiguous(), scale, shift, None, MODE_AFFINE_LRELU, float(slope))


def affine_selu(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """selu(x * scale[c] + shift[c]) over (N, C, ...): an eval-mode BatchNorm followed by SELU, one pass each way."""
    return _AffineAct.apply(x.contiguous(), scale, shift, None, MODE_AFFINE_SELU, 0.0)


def relu_affine(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu(x + pre[c]) * scale[c] + shift[c] over (N, C, ...)."""
    return _AffineAct.apply(x.contiguous(), scale, shift, pre, MODE_RELU_AFFINE, 0.0)


def _add_maxpool2_raw(a, b, bias):
    """(MaxPool2d(2)(a + b + bias[c]), selection bytes) — the kernel call, no autograd."""
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
    with _Launch("add_maxpool2_forward", a.device, tensors=(a, b, y, sel)):
        st = _lib.load().advstep_add_maxpool2_forward_f32(a.data_ptr(), b.data_ptr() if b is not None else None,
                                                          bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                                          sel.data_ptr(), N, C, H, W, _stream(a.device))
    _lib.check(st, "advstep_add_maxpool2_forward_f32")
    return y, sel


class _AddMaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, bias):
        y, sel = _add_maxpool2_raw(a, b, bias)
        N, C, H, W = a.shape
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
        with _Launch("maxpool2_backward", gy.device, tensors=(gy, sel, g)):
            st = _lib.load().advstep_maxpool2_backward_f32(gy.data_ptr(), sel.data_ptr(), g.data_ptr(), N, C, H, W,
                                                           _stream(gy.device))
        _lib.check(st, "advstep_maxpool2_backward_f32")
        return g, (g if ctx.two else None), None


def add_maxpool2(a: torch.Tensor, b: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MaxPool2d(2)(a + b + bias[c]); b and bias optional."""
    return _AddMaxPool2.apply(a.contiguous(), None if b is None else b.contiguous(), bias)


def _plane_view(t: torch.Tensor, name: str) -> int:
    """Batch stride (elements) of a (N, C, P) float32 HIP tensor whose (c, p) planes are dense: a contiguous tensor or a
    channel slice of one."""
    _require_strided(t, name)
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(1) != t.shape[2] or t.stride(0) < t.shape[1] * t.shape[2]:
        raise ValueError(f"{name}: expected a (N, C, P) tensor or a channel slice of one, got strides {t.stride()}")
    return t.stride(0)


def _require_strided(t: torch.Tensor, name: str) -> None:
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise _lib.AdvstepError(f"{name}: needs a float32 tensor on a HIP device (no CPU fallback)")


def res2net_link_forward(h: torch.Tensor, scale, shift, pre, y: torch.Tensor, other: Optional[torch.Tensor],
                         z: Optional[torch.Tensor]) -> None:
    """y[...] = relu(h + pre[c]) * scale[c] + shift[c] (y: a channel slice of the concatenated tensor); z = y + other when
    `other` (the next group, a channel slice) is given.  In place on y / z — no autograd (building block)."""
    _require(h, "h")
    N, C, P = h.shape
    y_bs = _plane_view(y, "y")
    o_bs = _plane_view(other, "other") if other is not None else 0
    if z is not None:
        _require(z, "z")
    st = _lib.load().advstep_res2net_link_forward_f32(h.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                      None if pre is None else pre.data_ptr(), y.data_ptr(), y_bs,
                                                      None if other is None else other.data_ptr(), o_bs,
                                                      None if z is None else z.data_ptr(), N, C, P, _stream(h.device))
    _lib.check(st, "advstep_res2net_link_forward_f32")


def res2net_link_backward(g1: torch.Tensor, g2: Optional[torch.Tensor], h: torch.Tensor, scale, shift, pre) -> torch.Tensor:
    """(h + pre <= 0) ? 0 : (g1 + g2) * scale  -> contiguous (N, C, P); g1 / g2 may be channel slices; g2 optional."""
    _require(h, "h")
    N, C, P = h.shape
    gx = torch.empty_like(h)
    st = _lib.load().advstep_res2net_link_backward_f32(g1.data_ptr(), _plane_view(g1, "g1"), None if g2 is None else g2.data_ptr(),
                                                       _plane_view(g2, "g2") if g2 is not None else 0, h.data_ptr(),
                                                       scale.data_ptr(), shift.data_ptr(), None if pre is None else pre.data_ptr(),
                                                       gx.data_ptr(), N, C, P, _stream(h.device))
    _lib.check(st, "advstep_res2net_link_backward_f32")
    return gx


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
        with _Launch("add_maxpool1d_forward", a.device, tensors=(a, b, y, sel)):
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
        with _Launch("maxpool1d_backward", gy.device, tensors=(gy, sel, g)):
            st = _lib.load().advstep_maxpool1d_backward_f32(gy.data_ptr(), sel.data_ptr(), g.data_ptr(), N, C, L, k,
                                                            _stream(gy.device))
        _lib.check(st, "advstep_maxpool1d_backward_f32")
        return g, (g if two else None), None


def add_maxpool1d(a: torch.Tensor, b: Optional[torch.Tensor], k: int) -> torch.Tensor:
    """MaxPool1d(k)(a + b) over (N, C, L), kernel = stride = k in 2..8; b optional."""
    return _AddMaxPool1d.apply(a.contiguous(), None if b is None else b.contiguous(), int(k))


def _afms_row(mode: int, a, b, alpha, r0, r1, out, rows: int, C: int, L: int) -> None:
    ptr = lambda t: None if t is None else t.data_ptr()
    with _Launch("afms_row", a.device, tensors=(a, b, out) if out.numel() >= a.numel() else (a, b)):
        st = _lib.load().advstep_afms_row_f32(mode, a.data_ptr(), ptr(b), ptr(alpha), ptr(r0), ptr(r1), out.data_ptr(), rows, C, L,
                                              _stream(a.device))
    _lib.check(st, "advstep_afms_row_f32")


class _Afms(torch.autograd.Function):
    """(x + alpha[c]) * sigmoid(fc(mean_t x)) — RawNet3's AFMS with frozen parameters; input gradient only."""

    @staticmethod
    def forward(ctx, x, alpha, weight, bias):
        _require(x, "x")
        N, C, L = x.shape
        mean = torch.empty((N, C), dtype=x.dtype, device=x.device)
        _afms_row(0, x, None, None, None, None, mean, N * C, C, L)
        y = torch.sigmoid(torch.addmm(bias, mean, weight.t()) if bias is not None else mean @ weight.t()).contiguous()
        out = torch.empty_like(x)
        _afms_row(1, x, None, alpha, y, None, out, N * C, C, L)
        ctx.save_for_backward(x, alpha, weight, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, alpha, weight, y = ctx.saved_tensors
        N, C, L = x.shape
        g = g.contiguous()
        gy = torch.empty((N, C), dtype=x.dtype, device=x.device)
        _afms_row(2, g, x, alpha, None, None, gy, N * C, C, L)
        g_mean = ((gy * y * (1.0 - y)) @ weight) / L                  # sigmoid', fc^T, the mean's 1 / L
        gx = torch.empty_like(x)
        _afms_row(3, g, None, None, y, g_mean.contiguous(), gx, N * C, C, L)
        return gx, None, None, None


def afms(x: torch.Tensor, alpha: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """RawNet3's AFMS on a (N, C, L) HIP tensor: alpha (C,) or (C, 1), fc weight (C, C) and bias (C,)."""
    return _Afms.apply(x.contiguous(), alpha.reshape(-1).contiguous(), weight, bias)


class _WeightedStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        _require(x, "x"), _require(w, "w")
        if x.shape != w.shape or x.dim() != 3:
            raise ValueError("x and w must be (N, C, L) tensors of the same shape")
        N, C, L = x.shape
        mu = torch.empty((N, C), dtype=x.dtype, device=x.device)
        m2 = torch.empty_like(mu)
        with _Launch("weighted_stats_forward", x.device, tensors=(x, w)):
            st = _lib.load().advstep_weighted_stats_forward_f32(x.data_ptr(), w.data_ptr(), mu.data_ptr(), m2.data_ptr(), N * C, L,
                                                                _stream(x.device))
        _lib.check(st, "advstep_weighted_stats_forward_f32")
        ctx.save_for_backward(x, w)
        return mu, m2

    @staticmethod
    def backward(ctx, gmu, gm2):
        x, w = ctx.saved_tensors
        N, C, L = x.shape
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        with _Launch("weighted_stats_backward", x.device, tensors=(x, w, gx, gw)):
            st = _lib.load().advstep_weighted_stats_backward_f32(x.data_ptr(), w.data_ptr(), gmu.contiguous().data_ptr(),
                                                                 gm2.contiguous().data_ptr(), gx.data_ptr(), gw.data_ptr(), N * C, L,
                                                                 _stream(x.device))
        _lib.check(st, "advstep_weighted_stats_backward_f32")
        return gx, gw


def weighted_stats(x: torch.Tensor, w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(sum_t x w, sum_t x^2 w) over the last dimension of two (N, C, L) tensors; differentiable in both."""
    return _WeightedStats.apply(x.contiguous(), w.contiguous())


class _LogMeanNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, eps):
        _require(y, "y")
        L = y.shape[-1]
        x = torch.empty_like(y)
        with _Launch("log_meannorm_forward", y.device, tensors=(y, x)):
            st = _lib.load().advstep_log_meannorm_forward_f32(y.data_ptr(), eps, x.data_ptr(), y.numel() // max(L, 1), L,
                                                              _stream(y.device))
        _lib.check(st, "advstep_log_meannorm_forward_f32")
        ctx.save_for_backward(y)
        ctx.eps = eps
        return x

    @staticmethod
    def backward(ctx, gx):
        (y,) = ctx.saved_tensors
        L = y.shape[-1]
        gx = gx.contiguous()
        gy = torch.empty_like(y)
        with _Launch("log_meannorm_backward", y.device, tensors=(gx, y, gy)):
            st = _lib.load().advstep_log_meannorm_backward_f32(gx.data_ptr(), y.data_ptr(), ctx.eps, gy.data_ptr(),
                                                               y.numel() // max(L, 1), L, _stream(y.device))
        _lib.check(st, "advstep_log_meannorm_backward_f32")
        return gy, None


def log_meannorm_supported(L: int) -> bool:
    return 0 < L <= _lib.load().advstep_log_meannorm_max_length()


def log_meannorm(y: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """log(|y| + eps) minus its mean over the last dimension, one pass each way (rows of at most 8192 elements)."""
    return _LogMeanNorm.apply(y.contiguous(), float(eps))


class _TailPool1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, res, scale, shift, pre, k):
        _require(h, "h"), _require(res, "res")
        if h.dim() != 3 or res.shape != h.shape:
            raise ValueError("h and res must be (N, C, L) tensors of the same shape")
        N, C, L = h.shape
        y = torch.empty((N, C, L // k), dtype=h.dtype, device=h.device)
        sel = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=h.device)
        with _Launch("tail_pool1d_forward", h.device, tensors=(h, res, y, sel)):
            st = _lib.load().advstep_tail_pool1d_forward_f32(h.data_ptr(), res.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                             None if pre is None else pre.data_ptr(), y.data_ptr(), sel.data_ptr(),
                                                             N, C, L, k, _stream(h.device))
        _lib.check(st, "advstep_tail_pool1d_forward_f32")
        ctx.save_for_backward(h, sel, scale, *([pre] if pre is not None else []))
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, gy):
        h, sel, scale, *pre = ctx.saved_tensors
        N, C, L = h.shape
        gy = gy.contiguous()
        g_h, g_res = torch.empty_like(h), torch.empty_like(h)
        with _Launch("tail_pool1d_backward", h.device, tensors=(gy, sel, h, g_h, g_res)):
            st = _lib.load().advstep_tail_pool1d_backward_f32(gy.data_ptr(), sel.data_ptr(), h.data_ptr(), scale.data_ptr(),
                                                              pre[0].data_ptr() if pre else None, g_h.data_ptr(), g_res.data_ptr(),
                                                              N, C, L, ctx.k, _stream(h.device))
        _lib.check(st, "advstep_tail_pool1d_backward_f32")
        return g_h, g_res, None, None, None, None


def tail_pool1d(h: torch.Tensor, res: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, pre: Optional[torch.Tensor],
                k: int) -> torch.Tensor:
    """MaxPool1d(k)(relu(h + pre[c]) * scale[c] + shift[c] + res) over (N, C, L), kernel = stride = k in 2..8; differentiable in h
    and res (per-channel constants are constants)."""
    return _TailPool1d.apply(h.contiguous(), res.contiguous(), scale, shift, pre, int(k))


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
        with _Launch("gate_maxpool2_forward", x.device, tensors=(x, y, sel)):
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
        with _Launch("gate_maxpool2_backward", x.device, tensors=(gy, sel, gx)):
            st = lib.advstep_gate_maxpool2_backward_f32(gy.data_ptr(), sel.data_ptr(), x.data_ptr(), gate.data_ptr(),
                                                        gx.data_ptr(), partial.data_ptr(), N, C, H, W, _stream(x.device))
        _lib.check(st, "advstep_gate_maxpool2_backward_f32")
        return gx, partial.sum(dim=1).view_as(gate)


def _attend_xw_enabled() -> bool:
    import os
    return os.environ.get("ADVSTEP_ATTEND_XW", "1") != "0"


class _AttendPool(torch.autograd.Function):
    """MaxPool2d(2)(x * g + g), g = sigmoid(fc(mean_hw x)) — SpecRNet's attention + pooling after a block (specrnet.py:145-149,
    163-172) with a frozen fc; input gradient only.  Backward: the gate's partial sums, sigmoid' and fc^T on (N, C), then ONE
    pass writes  gate * scatter(gy) + (the mean's share of d x)  — autograd ran the gate-pool backward, materialised the
    broadcast of the mean's gradient and added the two x-sized tensors."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require(x, "x")
        N, C, H, W = x.shape
        mean = torch.empty((N, C), dtype=x.dtype, device=x.device)
        _afms_row(0, x, None, None, None, None, mean, N * C, C, H * W)
        lib = _lib.load()
        if C <= 256 and tuple(weight.shape) == (C, C) and weight.is_contiguous():
            gate = torch.empty((N, C), dtype=x.dtype, device=x.device)
            with _Launch("attend_gate_fc", x.device, tensors=(mean, weight, gate)):
                st = lib.advstep_gate_fc_forward_f32(mean.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                     gate.data_ptr(), N, C, _stream(x.device))
            _lib.check(st, "advstep_gate_fc_forward_f32")
        else:
            gate = torch.sigmoid(torch.addmm(bias, mean, weight.t()) if bias is not None else mean @ weight.t()).contiguous()
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        sel = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=x.device)
        # the gate's gradient needs x only at the pooling winners: the forward writes them next to y (pooled size) and x itself
        # is not kept for backward (ADVSTEP_ATTEND_XW=0: gather them out of x in backward, A/B)
        compact = _attend_xw_enabled()
        xw = torch.empty_like(y) if compact else None
        with _Launch("attend_pool_forward", x.device, tensors=(x, y, sel, xw)):
            st = _lib.load().advstep_gate_maxpool2_forward_xw_f32(x.data_ptr(), gate.data_ptr(), y.data_ptr(), sel.data_ptr(),
                                                                  None if xw is None else xw.data_ptr(), N, C, H, W, _stream(x.device))
        _lib.check(st, "advstep_gate_maxpool2_forward_xw_f32")
        ctx.compact, ctx.shape = compact, (N, C, H, W)
        ctx.save_for_backward(xw if compact else x, sel, gate, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        kept, sel, gate, weight = ctx.saved_tensors          # kept: the pooling winners xw (compact) or x itself
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        dev, lib = gy.device, _lib.load()
        if ctx.compact:
            blocks = 1
            partial = torch.empty((N * C, 1), dtype=gy.dtype, device=dev)
            with _Launch("attend_pool_backward", dev, tensors=(gy, kept)):
                st = lib.advstep_gate_maxpool2_backward_gate_pooled_f32(gy.data_ptr(), kept.data_ptr(), partial.data_ptr(), N, C, H, W,
                                                                        _stream(dev))
            _lib.check(st, "advstep_gate_maxpool2_backward_gate_pooled_f32")
        else:
            blocks = max(lib.advstep_gate_maxpool2_blocks(H, W), 1)
            partial = torch.empty((N * C, blocks), dtype=gy.dtype, device=dev)
            st = lib.advstep_gate_maxpool2_backward_gate_f32(gy.data_ptr(), sel.data_ptr(), kept.data_ptr(), partial.data_ptr(), N, C,
                                                             H, W, _stream(dev))
            _lib.check(st, "advstep_gate_maxpool2_backward_gate_f32")
        if C <= 256 and tuple(weight.shape) == (C, C) and weight.is_contiguous():
            g_mean = torch.empty((N, C), dtype=gy.dtype, device=dev)
            with _Launch("attend_gate_fc", dev, tensors=(partial, gate, weight, g_mean)):
                st = lib.advstep_gate_fc_backward_f32(partial.data_ptr(), blocks, gate.data_ptr(), weight.data_ptr(), 1.0 / float(H * W),
                                                      g_mean.data_ptr(), N, C, _stream(dev))
            _lib.check(st, "advstep_gate_fc_backward_f32")
        else:
            ggate = partial.sum(dim=1).view(N, C)
            g_mean = (((ggate * gate * (1.0 - gate)) @ weight) / float(H * W)).contiguous()
        gx = torch.empty((N, C, H, W), dtype=gy.dtype, device=dev)
        with _Launch("attend_pool_backward", dev, tensors=(gy, sel, gx)):
            st = lib.advstep_gate_maxpool2_backward_input_f32(gy.data_ptr(), sel.data_ptr(), gate.data_ptr(), g_mean.data_ptr(),
                                                              gx.data_ptr(), N, C, H, W, _stream(dev))
        _lib.check(st, "advstep_gate_maxpool2_backward_input_f32")
        return gx, None, None


def attend_pool(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """MaxPool2d(2)(x * g + g) with g = sigmoid(fc(mean_hw x)) over (N, C, H, W); fc frozen, differentiable in x."""
    return _AttendPool.apply(x.contiguous(), weight, bias)


def gate_maxpool2(x: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """MaxPool2d(2)(x * gate[n, c] + gate[n, c]), gate (N, C) or (N, C, 1, 1); differentiable in x and gate."""
    return _GateMaxPool2.apply(x.contiguous(), gate.contiguous())


# ---- SpecRNet's residual block on the matrix cores (advstep_resconv_*, csrc/lcnn_wino.hip) -------------------------------

def resconv_supported(K1: int, K2: int, rows: int) -> bool:
    return bool(_lib.load().advstep_resconv_supported(K1, K2, rows))


def resconv_prepare(w3: torch.Tensor, w1: Optional[torch.Tensor] = None, rscale: Optional[torch.Tensor] = None,
                    kscale: Optional[torch.Tensor] = None, transpose: bool = False) -> torch.Tensor:
    """Transformed weights U of one operator (see advstep_detector.h).  transpose=False: w3 (rows, K1, 3, 3), w1 (rows, K2[, 1, 1]);
    transpose=True: the input gradient of a convolution with forward weights w3 (K1, rows, 3, 3), w1 (K2, rows[, 1, 1])."""
    _require(w3, "w3")
    rows, K1 = (w3.shape[1], w3.shape[0]) if transpose else (w3.shape[0], w3.shape[1])
    K2 = 0 if w1 is None else (w1.shape[0] if transpose else w1.shape[1])
    lib = _lib.load()
    n = lib.advstep_resconv_prepared_floats(K1, K2, rows)
    if n == 0:
        raise ValueError(f"unsupported convolution: K1={K1}, K2={K2}, rows={rows}")
    U = torch.empty(n, dtype=torch.float32, device=w3.device)
    ptr = lambda t: None if t is None else t.contiguous().data_ptr()
    keep = [t.contiguous() for t in (w3, w1, rscale, kscale) if t is not None]
    with _Launch("resconv_prepare", w3.device):
        st = lib.advstep_resconv_prepare_f32(keep[0].data_ptr(), ptr(w1), ptr(rscale), ptr(kscale), U.data_ptr(), rows, K1, K2,
                                             int(transpose), _stream(w3.device))
    _lib.check(st, "advstep_resconv_prepare_f32")
    del keep
    return U


def _sign_bytes(N: int, rows: int, H: int, W: int, device) -> torch.Tensor:
    """Room for an activation's sign bytes: one per (sample, channel, 2x2 tile), bit 2 i + j = output > 0 at tile position (i, j)."""
    return torch.empty((N, rows, (H + 1) // 2, (W + 1) // 2), dtype=torch.uint8, device=device)


def resconv(x1: torch.Tensor, x2: Optional[torch.Tensor], U: torch.Tensor, rows: int, shift: Optional[torch.Tensor] = None,
            slope: float = 1.0, with_act: bool = False):
    """leaky_relu(conv3x3(x1) + conv1x1(x2) + shift[row], slope) with prepared weights U — no autograd (building block).
    with_act: returns (y, act) with the sign bytes of y (see `_sign_bytes`) for `resconv_pooled_grad(act=...)`."""
    _require(x1, "x1")
    N, K1, H, W = x1.shape
    K2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None:
        _require(x2, "x2")
        if x2.shape[0] != N or tuple(x2.shape[2:]) != (H, W):
            raise ValueError("x1 and x2 must share batch and spatial dimensions")
    y = torch.empty((N, rows, H, W), dtype=x1.dtype, device=x1.device)
    act = _sign_bytes(N, rows, H, W, x1.device) if with_act else None
    with _Launch("resconv_forward", x1.device, work=8.0 * N * rows * (K1 + K2) * H * W):
        st = _lib.load().advstep_resconv_forward_act_f32(x1.data_ptr(), None if x2 is None else x2.data_ptr(), U.data_ptr(),
                                                         None if shift is None else shift.data_ptr(), float(slope), y.data_ptr(),
                                                         None if act is None else act.data_ptr(), N, K1, K2, rows, H, W,
                                                         _stream(x1.device))
    _lib.check(st, "advstep_resconv_forward_act_f32")
    return (y, act) if with_act else y


def resconv_pool2(x1: torch.Tensor, x2: Optional[torch.Tensor], U: torch.Tensor, rows: int,
                  bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(MaxPool2d(2)(conv3x3(x1) + conv1x1(x2) + bias[row]), selection bytes) — no autograd (building block)."""
    _require(x1, "x1")
    N, K1, H, W = x1.shape
    K2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None:
        _require(x2, "x2")
    y = torch.empty((N, rows, H // 2, W // 2), dtype=x1.dtype, device=x1.device)
    sel = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=x1.device)
    with _Launch("resconv_pool2_forward", x1.device, work=8.0 * N * rows * (K1 + K2) * H * W):
        st = _lib.load().advstep_resconv_pool2_forward_f32(x1.data_ptr(), None if x2 is None else x2.data_ptr(), U.data_ptr(),
                                                           None if bias is None else bias.data_ptr(), y.data_ptr(), sel.data_ptr(),
                                                           N, K1, K2, rows, H, W, _stream(x1.device))
    _lib.check(st, "advstep_resconv_pool2_forward_f32")
    return y, sel


def resconv_pool2_few(x1: torch.Tensor, x2: torch.Tensor, U: torch.Tensor, wd: torch.Tensor, rows: int,
                      bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(MaxPool2d(2)(conv3x3(x1) + conv1x1(x2; wd) + bias[row]), selection bytes) for an x2 of 1-2 channels: U prepared from the
    3x3 weights alone, the 1x1 part wd (rows, K2) applied in the kernel's epilogue — no autograd (building block)."""
    _require(x1, "x1"), _require(x2, "x2"), _require(wd, "wd")
    N, K1, H, W = x1.shape
    K2 = x2.shape[1]
    if K2 not in (1, 2) or x2.shape[0] != N or tuple(x2.shape[2:]) != (H, W) or tuple(wd.shape) != (rows, K2) or not wd.is_contiguous():
        raise ValueError("x2 must be (N, 1-2, H, W) and wd a contiguous (rows, K2)")
    y = torch.empty((N, rows, H // 2, W // 2), dtype=x1.dtype, device=x1.device)
    sel = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=x1.device)
    with _Launch("resconv_pool2_forward", x1.device, work=8.0 * N * rows * K1 * H * W):
        st = _lib.load().advstep_resconv_pool2_forward_few_f32(x1.data_ptr(), x2.data_ptr(), U.data_ptr(), wd.data_ptr(),
                                                               None if bias is None else bias.data_ptr(), y.data_ptr(),
                                                               sel.data_ptr(), N, K1, K2, rows, H, W, _stream(x1.device))
    _lib.check(st, "advstep_resconv_pool2_forward_few_f32")
    return y, sel


def resconv_pooled_grad(gy: torch.Tensor, sel: torch.Tensor, U: torch.Tensor, rows: int, H: int, W: int,
                        h: Optional[torch.Tensor] = None, slope: float = 1.0, act: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3(unpool(gy, sel)) [* leaky_relu'(h)] with U prepared with transpose=True: the input gradient of a convolution whose
    output went through MaxPool2d(2), straight from the pooled gradient (N, K, H//2, W//2) — no autograd (building block).
    The activation is given as its output h or as the sign bytes `act` a forward kernel wrote next to h (`with_act=True`)."""
    _require(gy, "gy")
    N, K = gy.shape[0], gy.shape[1]
    if tuple(gy.shape[2:]) != (H // 2, W // 2) or sel.numel() < gy.numel() or sel.dtype != torch.uint8:
        raise ValueError("gy / sel do not match a 2x2 pooling of an (H, W) plane")
    if h is not None:
        _require(h, "h")
        if tuple(h.shape) != (N, rows, H, W):
            raise ValueError("h must have the output's shape")
    if act is not None:
        if h is not None:
            raise ValueError("give the activation as h or as act, not both")
        if act.dtype != torch.uint8 or not act.is_contiguous() or tuple(act.shape) != (N, rows, (H + 1) // 2, (W + 1) // 2):
            raise ValueError("act must be the contiguous (N, rows, ceil(H/2), ceil(W/2)) sign bytes of the activation")
    g = torch.empty((N, rows, H, W), dtype=gy.dtype, device=gy.device)
    with _Launch("resconv_pooled_grad", gy.device, work=8.0 * N * rows * K * H * W):
        if act is not None:
            st = _lib.load().advstep_resconv_pooled_grad_act_f32(gy.data_ptr(), sel.data_ptr(), U.data_ptr(), act.data_ptr(),
                                                                 float(slope), g.data_ptr(), N, K, rows, H, W, _stream(gy.device))
        else:
            st = _lib.load().advstep_resconv_pooled_grad_f32(gy.data_ptr(), sel.data_ptr(), U.data_ptr(),
                                                             None if h is None else h.data_ptr(), float(slope), g.data_ptr(), N, K,
                                                             rows, H, W, _stream(gy.device))
    _lib.check(st, "advstep_resconv_pooled_grad_f32")
    return g


def conv3x3_fewin_supported(channels: int) -> bool:
    return bool(_lib.load().advstep_conv3x3_fewin_supported(channels))


def conv3x3_fewin(x: torch.Tensor, w: torch.Tensor, shift: Optional[torch.Tensor], slope: float, with_act: bool = False):
    """leaky_relu(conv3x3(x, w, pad 1) + shift[co], slope) for 1-2 input channels (vector-ALU kernel) — no autograd.
    with_act: returns (y, act) with the sign bytes of y (see `_sign_bytes`)."""
    _require(x, "x"), _require(w, "w")
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = torch.empty((N, Cout, H, W), dtype=x.dtype, device=x.device)
    act = _sign_bytes(N, Cout, H, W, x.device) if with_act else None
    with _Launch("conv3x3_fewin_forward", x.device, work=18.0 * N * Cout * x.shape[1] * H * W, tensors=(x, y, act)):
        st = _lib.load().advstep_conv3x3_fewin_forward_act_f32(x.data_ptr(), w.data_ptr(), None if shift is None else shift.data_ptr(),
                                                               float(slope), y.data_ptr(), None if act is None else act.data_ptr(),
                                                               N, Cin, Cout, H, W, _stream(x.device))
    _lib.check(st, "advstep_conv3x3_fewin_forward_act_f32")
    return (y, act) if with_act else y


def conv3x3_fewout_grad(g1: torch.Tensor, w3: torch.Tensor, gp: Optional[torch.Tensor] = None, sel: Optional[torch.Tensor] = None,
                        wd: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Input gradient (1-2 channels) of conv3x3(x, w3 (K, rows, 3, 3)) for d(out) = g1, plus that of the 1x1 convolution wd (K, rows)
    whose d(out) is the unpooled (gp, sel) — no autograd."""
    _require(g1, "g1"), _require(w3, "w3")
    N, K, H, W = g1.shape
    rows = w3.shape[1]
    if gp is not None:
        _require(gp, "gp"), _require(wd, "wd")
        if tuple(gp.shape) != (N, K, H // 2, W // 2) or sel.numel() < gp.numel() or tuple(wd.shape) != (K, rows):
            raise ValueError("gp / sel / wd do not match")
        if gp.numel() == 0:                    # nothing was pooled (H or W < 2): the identity path carries no gradient
            gp = None
    gx = torch.empty((N, rows, H, W), dtype=g1.dtype, device=g1.device)
    with _Launch("conv3x3_fewout_grad", g1.device, work=18.0 * N * K * rows * H * W, tensors=(g1, gp, gx)):
        st = _lib.load().advstep_conv3x3_fewout_grad_f32(g1.data_ptr(), w3.data_ptr(), None if gp is None else gp.data_ptr(),
                                                         None if gp is None else sel.data_ptr(), None if gp is None else wd.data_ptr(),
                                                         gx.data_ptr(), N, K, rows, H, W, _stream(g1.device))
    _lib.check(st, "advstep_conv3x3_fewout_grad_f32")
    return gx


class ResBlockPlan:
    """Everything `res_block` needs from one `Residual_block2D` with frozen parameters: folded per-channel constants and
    the transformed weights of its convolutions.  Built by `res_block_plan`, cached on the module until a parameter changes."""

    def __init__(self, conv1, bn2, conv2, down, slope: float):
        with torch.no_grad():
            self.cin, self.cout, self.slope = conv1.in_channels, conv1.out_channels, float(slope)
            scale, shift = bn_eval_affine(bn2)
            if conv1.bias is not None:
                shift = shift + conv1.bias * scale
            self.scale, self.shift = scale.contiguous(), shift.contiguous()
            w1, w2 = conv1.weight.detach(), conv2.weight.detach()
            self.downsample = down is not None
            wd = None if down is None else down.weight.detach().reshape(self.cout, self.cin).contiguous()
            bias = None if conv2.bias is None else conv2.bias.detach()
            if down is not None and down.bias is not None:
                bias = down.bias.detach() if bias is None else bias + down.bias.detach()
            self.bias = None if bias is None else bias.contiguous()
            # 1-2 input channels (the spectrogram end): conv1 and the d x convolution on the vector ALUs, bn2's scale inside
            # the weights; otherwise everything on the matrix cores
            self.fewin = self.downsample and conv3x3_fewin_supported(self.cin)
            # forward: conv1 (+ bn2 scale on its rows), conv2 [+ downsample as centre taps over x]
            self.w1_scaled = (w1 * self.scale.view(-1, 1, 1, 1)).contiguous() if self.fewin else None
            self.wd = wd
            self.U1 = None if self.fewin else resconv_prepare(w1, rscale=self.scale)
            # block0 (1-2 input channels): the downsample's channels would pad a whole k-step of conv2's reduction — they are
            # applied in the kernel's epilogue instead (ADVSTEP_RESBLOCK_FEW=0: as reduction channels)
            self.few = self.fewin and _few_epilogue_enabled()
            self.U2 = resconv_prepare(w2, None if self.few else wd)
            # input gradients: d h1 from d h2; d x from d(conv1 out) [+ d h2 through the downsample]
            self.U2T = resconv_prepare(w2, transpose=True)
            self.U1T = None if self.fewin else resconv_prepare(w1, wd, kscale=self.scale, transpose=True)


def res_block_supported(conv1, conv2, down) -> bool:
    def plain3(c):
        return (c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1
                and c.padding_mode == "zeros")
    if not (plain3(conv1) and plain3(conv2) and conv2.in_channels == conv1.out_channels == conv2.out_channels):
        return False
    cin, cout = conv1.in_channels, conv1.out_channels
    if down is not None and not (down.kernel_size == (1, 1) and down.stride == (1, 1) and down.padding == (0, 0)
                                 and down.groups == 1 and down.in_channels == cin and down.out_channels == cout):
        return False
    if down is None and cin != cout:
        return False
    return (resconv_supported(cout, cin if down is not None else 0, cout)
            and resconv_supported(cout, 0, cout) and resconv_supported(cin, 0, cout)
            and resconv_supported(cout, cout if down is not None else 0, cin))


def res_block_shape_supported(x_shape, cout: int) -> bool:
    """The size limits the block's kernels enforce on the host side (csrc/lcnn_wino.hip:resconv_check, csrc/detector_conv.hip:
    32-bit buffer offsets per tensor, 31-bit pooled-cell indices, grid.y <= 65535 samples in the vector-ALU kernels) for an
    input of shape (N, cin, H, W).  A batch beyond them takes the MIOpen + elementwise-kernel path instead of raising."""
    N, cin, H, W = (int(v) for v in x_shape)
    big = max(cin, cout)
    return (N <= 65535 and N * big * H * W * 4 < (1 << 31) and N * cout * H * W * 4 < (1 << 33)
            and N * ((H + 1) // 2) * ((W + 1) // 2) < (1 << 31))


def res_block_plan(block, conv1, bn2, conv2, down, slope: float) -> ResBlockPlan:
    tensors = [t for m in (conv1, bn2, conv2, down) if m is not None for t in list(m.parameters()) + list(m.buffers())]
    key = tuple((t.data_ptr(), t._version) for t in tensors) + (_few_epilogue_enabled(),)
    if getattr(block, "_advstep_plan_key", None) != key:
        block._advstep_plan_key, block._advstep_plan = key, ResBlockPlan(conv1, bn2, conv2, down, slope)
    return block._advstep_plan


def _few_epilogue_enabled() -> bool:
    """ADVSTEP_RESBLOCK_FEW=0 (read when a block's plan is built): block0's downsample as reduction channels (A/B measurements)."""
    import os
    return os.environ.get("ADVSTEP_RESBLOCK_FEW", "1") != "0"


def _act_bytes_enabled() -> bool:
    """ADVSTEP_RESBLOCK_ACT=0 (read per call): the residual blocks save h1 itself for backward (A/B measurements)."""
    import os
    return os.environ.get("ADVSTEP_RESBLOCK_ACT", "1") != "0"


class _ResBlock(torch.autograd.Function):
    """MaxPool2d(2)(conv2(leaky_relu(bn2(conv1(x)))) + identity) of SpecRNet's Residual_block2D (src/models/specrnet.py:73-91),
    identity = conv_downsample(x) or x; input gradient only.  The downsample convolution is part of conv2's reduction and the
    pooling part of its epilogue; backward runs the transposed convolutions through the same kernel."""

    @staticmethod
    def forward(ctx, x, plan):
        _require(x, "x")
        p = plan
        # backward needs only the SIGN of h1 (LeakyReLU's derivative): the kernel that writes h1 also writes one byte per 2x2
        # tile with the four signs, and that — 1/16 of h1's bytes — is what is saved and read back (ADVSTEP_RESBLOCK_ACT=0: h1)
        compact = p.slope > 0 and _act_bytes_enabled()
        if p.fewin:
            h1 = conv3x3_fewin(x, p.w1_scaled, p.shift, p.slope, with_act=compact)
        else:
            h1 = resconv(x, None, p.U1, p.cout, p.shift, p.slope, with_act=compact)
        h1, act = h1 if compact else (h1, None)
        if p.downsample and p.few:
            y, sel = resconv_pool2_few(h1, x, p.U2, p.wd, p.cout, p.bias)
        elif p.downsample:
            y, sel = resconv_pool2(h1, x, p.U2, p.cout, p.bias)
        else:
            y, sel = _add_maxpool2_raw(resconv(h1, None, p.U2, p.cout), x, p.bias)
        ctx.plan, ctx.compact = p, compact
        ctx.save_for_backward(x, act if compact else h1, sel)
        return y

    @staticmethod
    def backward(ctx, gy):
        p = ctx.plan
        x, kept, sel = ctx.saved_tensors                     # kept: h1's sign bytes (compact) or h1 itself
        N, _, H, W = x.shape
        gy = gy.contiguous()
        lib = _lib.load()
        # d(conv1 out) / bn2 scale = conv2^T(unpool(gy)) * lrelu'(h1): unpooling in the operand load, lrelu' in the epilogue
        if ctx.compact:
            g_pre = resconv_pooled_grad(gy, sel, p.U2T, p.cout, H, W, None, p.slope, act=kept)
        else:
            g_pre = resconv_pooled_grad(gy, sel, p.U2T, p.cout, H, W, kept, p.slope)
        if p.fewin:
            return conv3x3_fewout_grad(g_pre, p.w1_scaled, gy, sel, p.wd), None
        g_h2 = torch.empty((N, p.cout, H, W), dtype=gy.dtype, device=gy.device)       # the identity path's gradient
        with _Launch("maxpool2_backward", gy.device, tensors=(gy, sel, g_h2)):
            st = lib.advstep_maxpool2_backward_f32(gy.data_ptr(), sel.data_ptr(), g_h2.data_ptr(), N, p.cout, H, W, _stream(gy.device))
        _lib.check(st, "advstep_maxpool2_backward_f32")
        gx = resconv(g_pre, g_h2 if p.downsample else None, p.U1T, p.cin)
        if not p.downsample:
            gx += g_h2
        return gx, None


def res_block(x: torch.Tensor, plan: ResBlockPlan) -> torch.Tensor:
    return _ResBlock.apply(x.contiguous(), plan)
