"""Max-feature-map (+ 2x2 max-pool) of LCNN as differentiable ops backed by the HIP kernels of
include/advstep_lcnn.h.

`mfm(x, bias=None)` replaces  `(x + bias[None, :, None, None]).view(N, 2, C, H, W).max(1)[0]`
(reference: src/models/lcnn.py:76-95 after the Conv2d's bias add) and `mfm_pool2(x, bias=None)` additionally the
`MaxPool2d((2, 2), (2, 2))` that follows it (lcnn.py:123,129,137,154).  Forward values and input gradients are
bit-identical to the ATen ops they replace, ties and NaN included (tests/test_gpu_lcnn_ops.py).
HIP tensors only: these functions raise on CPU tensors (the model keeps its plain torch path for CPU runs)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .hip_ops import _Launch, _require, _stream


def _gemm(fn, a2d: torch.Tensor, b2d: torch.Tensor, *lead):
    """A library GEMM (rocBLAS through torch.addmm / torch.mm) under a launch bracket of its own, so bench.py's `roofline_step`
    sees the recurrent layers' projections: a2d (M, K) x b2d (K, N), 2 M N K flop."""
    with _Launch("rnn_projection_gemm", a2d.device, work=2.0 * a2d.shape[0] * a2d.shape[1] * b2d.shape[1]):
        return fn(*lead, a2d, b2d)


def _check_input(x: torch.Tensor, bias: Optional[torch.Tensor]):
    _require(x, "x")
    if x.dim() != 4 or x.shape[1] % 2 != 0:
        raise ValueError(f"expected a contiguous (N, 2C, H, W) tensor, got {tuple(x.shape)}")
    if bias is not None:
        _require(bias, "bias")
        if bias.numel() != x.shape[1]:
            raise ValueError(f"bias must have {x.shape[1]} entries, got {bias.numel()}")


def _bias_grad(gx: torch.Tensor, needed: bool):
    # only for training-style use (parameters that require grad); the attacks freeze the parameters
    return gx.sum(dim=(0, 2, 3)) if needed else None


def _bn_ptrs(bn):
    """bn = (mean (C), invstd (C)) of an eval-mode BatchNorm2d(affine=False) folded into the kernel, or None."""
    if bn is None:
        return None, None
    mean, invstd = bn
    _require(mean, "bn mean"), _require(invstd, "bn invstd")
    return mean.data_ptr(), invstd.data_ptr()


class _Mfm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, bn):
        _check_input(x, bias)
        bn_mean, bn_invstd = _bn_ptrs(bn)
        N, C2, H, W = x.shape
        C, HW = C2 // 2, H * W
        y = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device)
        lib = _lib.load()
        sel = torch.empty(max(lib.advstep_mfm_sel_bytes(N, C, HW), 1), dtype=torch.uint8, device=x.device)
        with _Launch("mfm_forward", x.device, tensors=(x, y, sel)):
            st = lib.advstep_mfm_forward_f32(x.data_ptr(), bias.data_ptr() if bias is not None else None, bn_mean,
                                             bn_invstd, y.data_ptr(), sel.data_ptr(), N, C, HW, _stream(x.device))
        _lib.check(st, "advstep_mfm_forward_f32")
        ctx.save_for_backward(sel, *([bn[1]] if bn is not None else []))
        ctx.shape = (N, C, H, W)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        sel, *scale = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty((N, 2 * C, H, W), dtype=gy.dtype, device=gy.device)
        with _Launch("mfm_backward", gy.device, tensors=(gy, sel, gx)):
            st = _lib.load().advstep_mfm_backward_f32(gy.data_ptr(), sel.data_ptr(), scale[0].data_ptr() if scale else None,
                                                      gx.data_ptr(), N, C, H * W, _stream(gy.device))
        _lib.check(st, "advstep_mfm_backward_f32")
        return gx, _bias_grad(gx, ctx.has_bias and ctx.needs_input_grad[1]), None


class _MfmPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, bn):
        _check_input(x, bias)
        bn_mean, bn_invstd = _bn_ptrs(bn)
        N, C2, H, W = x.shape
        C = C2 // 2
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        idx = torch.empty(max(y.numel(), 2), dtype=torch.uint8, device=x.device)
        with _Launch("mfm_pool2_forward", x.device, tensors=(x, y, idx)):
            st = _lib.load().advstep_mfm_pool2_forward_f32(x.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                           bn_mean, bn_invstd, y.data_ptr(), idx.data_ptr(), N, C, H, W,
                                                           _stream(x.device))
        _lib.check(st, "advstep_mfm_pool2_forward_f32")
        ctx.save_for_backward(idx, *([bn[1]] if bn is not None else []))
        ctx.shape = (N, C, H, W)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        idx, *scale = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty((N, 2 * C, H, W), dtype=gy.dtype, device=gy.device)
        with _Launch("mfm_pool2_backward", gy.device, tensors=(gy, idx, gx)):
            st = _lib.load().advstep_mfm_pool2_backward_f32(gy.data_ptr(), idx.data_ptr(),
                                                            scale[0].data_ptr() if scale else None, gx.data_ptr(), N, C, H,
                                                            W, _stream(gy.device))
        _lib.check(st, "advstep_mfm_pool2_backward_f32")
        return gx, _bias_grad(gx, ctx.has_bias and ctx.needs_input_grad[1]), None


class _Conv5MfmPool2(torch.autograd.Function):
    """First LCNN block in one kernel; differentiable w.r.t. the input only (weights must not require grad)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require(x, "x"), _require(weight, "weight")
        if bias is not None:
            _require(bias, "bias")
        if x.dim() != 4 or x.shape[1] != 1 or weight.dim() != 4 or tuple(weight.shape[1:]) != (1, 5, 5) \
                or weight.shape[0] % 2 != 0 or (bias is not None and bias.numel() != weight.shape[0]):
            raise ValueError(f"expected x (N, 1, H, W), weight (2C, 1, 5, 5), bias (2C); got {tuple(x.shape)}, "
                             f"{tuple(weight.shape)}")
        N, _, H, W = x.shape
        C = weight.shape[0] // 2
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        idx = torch.empty(max(y.numel(), 1), dtype=torch.uint8, device=x.device)
        with _Launch("conv5_mfm_pool2_forward", x.device, work=100.0 * N * C * H * W, tensors=(x, y, idx)):
            st = _lib.load().advstep_conv5_mfm_pool2_forward_f32(
                x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                idx.data_ptr(), N, C, H, W, _stream(x.device))
        _lib.check(st, "advstep_conv5_mfm_pool2_forward_f32")
        ctx.save_for_backward(idx, weight)
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.needs_input_grad[1] or (len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2]):
            raise RuntimeError("conv5_mfm_pool2 provides the input gradient only; call it with frozen weights "
                               "(the model falls back to Conv2d + mfm_pool2 otherwise)")
        idx, weight = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty((N, 1, H, W), dtype=gy.dtype, device=gy.device)
        # work = the multiply-adds the gradient needs: 25 taps per pooled output and channel (2 x 25 x N C H/2 W/2) - an eighth of
        # the dense transposed convolution rounds 1-4 priced here, which no kernel of this library executes
        with _Launch("conv5_mfm_pool2_backward", gy.device, work=12.5 * N * C * H * W, tensors=(gy, idx, gx)):
            st = _lib.load().advstep_conv5_mfm_pool2_backward_f32(gy.data_ptr(), idx.data_ptr(), weight.data_ptr(),
                                                                  gx.data_ptr(), N, C, H, W, _stream(gy.device))
        _lib.check(st, "advstep_conv5_mfm_pool2_backward_f32")
        return gx, None, None


def conv5_mfm_pool2(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """MaxPool2d(2,2)(MFM(conv2d(x, weight, bias, stride 1, padding 2))) for a one-channel input, one kernel."""
    return _Conv5MfmPool2.apply(x.contiguous(), weight.contiguous(), bias)


class _Conv1x1Mfm(torch.autograd.Function):
    """Conv2d(Cin, 2C, 1x1) + bias + MFM in one kernel; differentiable w.r.t. the input only."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        _require(x, "x"), _require(weight, "weight")
        if bias is not None:
            _require(bias, "bias")
        bn_mean, bn_invstd = _bn_ptrs(bn)
        if x.dim() != 4 or weight.dim() != 4 or tuple(weight.shape[2:]) != (1, 1) or weight.shape[1] != x.shape[1] \
                or weight.shape[0] % 2 != 0 or (bias is not None and bias.numel() != weight.shape[0]):
            raise ValueError(f"expected x (N, Cin, H, W), weight (2C, Cin, 1, 1), bias (2C); got {tuple(x.shape)}, "
                             f"{tuple(weight.shape)}")
        N, Cin, H, W = x.shape
        C, P = weight.shape[0] // 2, H * W
        lib = _lib.load()
        if not lib.advstep_conv1x1_mfm_supported(Cin):
            raise ValueError(f"conv1x1_mfm supports Cin in (32, 48, 64), got {Cin}")
        y = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device)
        sel = torch.empty(max(lib.advstep_conv1x1_mfm_sel_bytes(N, C, P), 4), dtype=torch.uint8, device=x.device)
        with _Launch("conv1x1_mfm_forward", x.device, work=4.0 * N * Cin * C * P, tensors=(x, y, sel)):
            st = lib.advstep_conv1x1_mfm_forward_f32(x.data_ptr(), weight.data_ptr(),
                                                     bias.data_ptr() if bias is not None else None, bn_mean, bn_invstd,
                                                     y.data_ptr(), sel.data_ptr(), N, Cin, C, P, _stream(x.device))
        _lib.check(st, "advstep_conv1x1_mfm_forward_f32")
        ctx.save_for_backward(sel, weight, *([bn[1]] if bn is not None else []))
        ctx.shape = (N, Cin, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.needs_input_grad[1] or (len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2]):
            raise RuntimeError("conv1x1_mfm provides the input gradient only; call it with frozen weights "
                               "(the model falls back to Conv2d + mfm otherwise)")
        sel, weight, *scale = ctx.saved_tensors
        N, Cin, C, H, W = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty((N, Cin, H, W), dtype=gy.dtype, device=gy.device)
        with _Launch("conv1x1_mfm_backward", gy.device, work=4.0 * N * Cin * C * H * W, tensors=(gy, sel, gx)):
            st = _lib.load().advstep_conv1x1_mfm_backward_f32(gy.data_ptr(), sel.data_ptr(), weight.data_ptr(),
                                                              scale[0].data_ptr() if scale else None, gx.data_ptr(), N, Cin,
                                                              C, H * W, _stream(gy.device))
        _lib.check(st, "advstep_conv1x1_mfm_backward_f32")
        return gx, None, None, None


def conv1x1_mfm_supported(in_channels: int) -> bool:
    return bool(_lib.load().advstep_conv1x1_mfm_supported(in_channels))


def conv1x1_mfm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None) -> torch.Tensor:
    """[BN_eval](MFM(conv2d(x, weight (2C, Cin, 1, 1), bias))) in one kernel (Cin in 32/48/64); bn = (mean, invstd)."""
    return _Conv1x1Mfm.apply(x.contiguous(), weight.contiguous(), bias, bn)


# ---- 3x3 blocks on the matrix cores (csrc/lcnn_wino.hip) -------------------------------------------------------------------

def _prepared_weights(weight: torch.Tensor, mode: int, gscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """U = G g G^T in the kernel's layout.  Cached ON the weight tensor object (a model's Parameter lives as long as the
    model) and keyed by its version counter, which every in-place update — optimizer step, load_state_dict — bumps."""
    cache = getattr(weight, "_advstep_wino", None)
    if cache is None or cache[0] != (weight._version, weight.data_ptr()):
        cache = ((weight._version, weight.data_ptr()), {})
        try:
            weight._advstep_wino = cache
        except AttributeError:      # exotic tensor subclasses without a __dict__: just do not cache
            pass
    slot = (mode, None if gscale is None else (gscale.data_ptr(), gscale._version))
    U = cache[1].get(slot)
    if U is None:
        Cout, Cin = weight.shape[0], weight.shape[1]
        lib = _lib.load()
        U = torch.empty(lib.advstep_conv3x3_prepared_floats(Cin, Cout, mode), dtype=torch.float32, device=weight.device)
        with _Launch("conv3x3_prepare", weight.device):
            st = lib.advstep_conv3x3_prepare_f32(weight.data_ptr(), None if gscale is None else gscale.data_ptr(), U.data_ptr(),
                                                 Cin, Cout, mode, _stream(weight.device))
        _lib.check(st, "advstep_conv3x3_prepare_f32")
        if len(cache[1]) >= 8:
            cache[1].clear()
        cache[1][slot] = (U, gscale)       # keeps the scale tensor alive: its address is part of the key
        return U
    return U[0]


def _batch_chunks(N: int, per_sample_floats: int):
    """The kernels address a tensor through a 32-bit buffer descriptor: < 2 GiB per call, split the batch otherwise."""
    step = max(1, ((1 << 31) - 1) // (4 * per_sample_floats))
    return [(lo, min(N, lo + step)) for lo in range(0, N, step)]


class _Conv3x3MfmPool2(torch.autograd.Function):
    """Conv2d(3x3, pad 1) + bias + max-feature-map + 2x2 pool [+ eval BatchNorm] in one kernel; differentiable w.r.t.
    the input only (the weights must not require grad: the attacks freeze them)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, U, U_grad):
        _require(x, "x"), _require(weight, "weight")
        if bias is not None:
            _require(bias, "bias")
        N, Cin, H, W = x.shape
        C = weight.shape[0] // 2
        bn_mean, bn_invstd = _bn_ptrs(bn)
        y = torch.empty((N, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        idx = torch.empty(max(y.numel(), 2), dtype=torch.uint8, device=x.device)
        lib = _lib.load()
        per_out = C * (H // 2) * (W // 2)
        for lo, hi in _batch_chunks(N, max(Cin * H * W, 1)):
            with _Launch("conv3x3_mfm_pool2_forward", x.device, work=16.0 * (hi - lo) * C * Cin * H * W):
                st = lib.advstep_conv3x3_mfm_pool2_forward_f32(
                    x[lo:hi].data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None, bn_mean, bn_invstd,
                    y[lo:hi].data_ptr() if hi > lo else None, idx.data_ptr() + lo * per_out, hi - lo, Cin, C, H, W,
                    _stream(x.device))
            _lib.check(st, "advstep_conv3x3_mfm_pool2_forward_f32")
        ctx.save_for_backward(idx, U_grad)
        ctx.shape = (N, Cin, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        idx, U = ctx.saved_tensors
        N, Cin, C, H, W = ctx.shape
        gy = gy.contiguous()
        lib = _lib.load()
        # d(conv out) — gy routed to the winning position of the winning half — is expanded inside the kernel's operand
        # load; the BatchNorm scale is folded into the prepared weights
        gx = torch.empty((N, Cin, H, W), dtype=gy.dtype, device=gy.device)
        per_cell = C * (H // 2) * (W // 2)
        for lo, hi in _batch_chunks(N, max(4 * per_cell, Cin * H * W, 1)):
            with _Launch("conv3x3_mfm_pool2_backward", gy.device, work=16.0 * (hi - lo) * C * Cin * H * W):
                st = lib.advstep_conv3x3_mfm_pool2_backward_f32(gy[lo:hi].data_ptr(), idx.data_ptr() + lo * per_cell,
                                                                U.data_ptr(), gx[lo:hi].data_ptr(), hi - lo, Cin, C, H, W,
                                                                _stream(gy.device))
            _lib.check(st, "advstep_conv3x3_mfm_pool2_backward_f32")
        return gx, None, None, None, None, None


class _Conv3x3Mfm(torch.autograd.Function):
    """Conv2d(3x3, pad 1) + bias + max-feature-map [+ eval BatchNorm], no pool; input gradient only."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, U, U_grad):
        _require(x, "x"), _require(weight, "weight")
        if bias is not None:
            _require(bias, "bias")
        N, Cin, H, W = x.shape
        C = weight.shape[0] // 2
        bn_mean, bn_invstd = _bn_ptrs(bn)
        lib = _lib.load()
        y = torch.empty((N, C, H, W), dtype=x.dtype, device=x.device)
        per_sel = C * ((H + 1) // 2) * ((W + 1) // 2)
        sel = torch.empty(max(N * per_sel, 1), dtype=torch.uint8, device=x.device)
        for lo, hi in _batch_chunks(N, max(Cin * H * W, 1)):
            with _Launch("conv3x3_mfm_forward", x.device, work=16.0 * (hi - lo) * C * Cin * H * W):
                st = lib.advstep_conv3x3_mfm_forward_f32(
                    x[lo:hi].data_ptr(), U.data_ptr(), bias.data_ptr() if bias is not None else None, bn_mean, bn_invstd,
                    y[lo:hi].data_ptr(), sel.data_ptr() + lo * per_sel, hi - lo, Cin, C, H, W, _stream(x.device))
            _lib.check(st, "advstep_conv3x3_mfm_forward_f32")
        ctx.save_for_backward(sel, U_grad, *([bn[1]] if bn is not None else []))
        ctx.shape = (N, Cin, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        sel, U, *scale = ctx.saved_tensors
        N, Cin, C, H, W = ctx.shape
        gy = gy.contiguous()
        lib = _lib.load()
        gconv = torch.empty((N, 2 * C, H, W), dtype=gy.dtype, device=gy.device)
        with _Launch("conv3x3_mfm_backward", gy.device, tensors=(gy, sel, gconv)):
            st = lib.advstep_conv3x3_mfm_backward_f32(gy.data_ptr(), sel.data_ptr(), scale[0].data_ptr() if scale else None,
                                                      gconv.data_ptr(), N, C, H, W, _stream(gy.device))
        _lib.check(st, "advstep_conv3x3_mfm_backward_f32")
        gx = torch.empty((N, Cin, H, W), dtype=gy.dtype, device=gy.device)
        for lo, hi in _batch_chunks(N, max(2 * C * H * W, 1)):
            with _Launch("conv3x3_backward_data", gy.device, work=16.0 * (hi - lo) * C * Cin * H * W):
                st = lib.advstep_conv3x3_backward_data_f32(gconv[lo:hi].data_ptr(), U.data_ptr(), gx[lo:hi].data_ptr(), hi - lo,
                                                           Cin, 2 * C, H, W, _stream(gy.device))
            _lib.check(st, "advstep_conv3x3_backward_data_f32")
        return gx, None, None, None, None, None


def conv3x3_mfm(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None) -> torch.Tensor:
    """MFM(conv2d(x, weight, bias, padding=1)) [then eval BatchNorm], Winograd on the matrix cores."""
    weight = weight if weight.is_contiguous() else weight.contiguous()
    return _Conv3x3Mfm.apply(x.contiguous(), weight, bias, bn, _prepared_weights(weight, 0), _prepared_weights(weight, 1))


def conv3x3_supported(in_channels: int, out_channels: int) -> bool:
    return bool(_lib.load().advstep_conv3x3_supported(in_channels, out_channels))


def conv3x3_mfm_pool2(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None) -> torch.Tensor:
    """MaxPool2d(2, 2)(MFM(conv2d(x, weight, bias, padding=1))) [then eval BatchNorm], Winograd on the matrix cores."""
    weight = weight if weight.is_contiguous() else weight.contiguous()
    return _Conv3x3MfmPool2.apply(x.contiguous(), weight, bias, bn, _prepared_weights(weight, 0),
                                  _prepared_weights(weight, 2, None if bn is None else bn[1]))



class _LstmLayer(torch.autograd.Function):
    """One (bi)directional LSTM layer, sequence-first, zero initial state; input gradient only."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, bias):
        # x (T, B, I); w_ih (D*4H, I); w_hh (D, 4H, H); bias (D*4H) = b_ih + b_hh
        _require(x, "x"), _require(w_ih, "w_ih"), _require(w_hh, "w_hh"), _require(bias, "bias")
        T, B, I = x.shape
        D, H4, H = w_hh.shape
        if H4 != 4 * H or tuple(w_ih.shape) != (D * H4, I) or bias.numel() != D * H4:
            raise ValueError("inconsistent LSTM parameter shapes")
        gx = _gemm(torch.addmm, x.reshape(T * B, I), w_ih.t(), bias)   # one GEMM for all steps and directions
        out = torch.empty((T, B, D * H), dtype=x.dtype, device=x.device)
        gates = torch.empty((T, B, D, H4), dtype=x.dtype, device=x.device)
        cell = torch.empty((T, B, D, H), dtype=x.dtype, device=x.device)
        with _Launch("lstm_forward", x.device, tensors=(gx, out, gates, cell)):
            st = _lib.load().advstep_lstm_forward_f32(gx.data_ptr(), w_hh.data_ptr(), out.data_ptr(), gates.data_ptr(),
                                                      cell.data_ptr(), T, B, D, H, _stream(x.device))
        _lib.check(st, "advstep_lstm_forward_f32")
        ctx.save_for_backward(gates, cell, w_hh, w_ih)
        ctx.dims = (T, B, I, D, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        if any(ctx.needs_input_grad[1:]):
            raise RuntimeError("lstm_layer provides the input gradient only; call it with frozen weights "
                               "(the model falls back to torch.nn.LSTM otherwise)")
        gates, cell, w_hh, w_ih = ctx.saved_tensors
        T, B, I, D, H = ctx.dims
        dout = dout.contiguous()
        dgx = torch.empty((T, B, D, 4 * H), dtype=dout.dtype, device=dout.device)
        with _Launch("lstm_backward", dout.device, tensors=(dout, gates, cell, dgx)):
            st = _lib.load().advstep_lstm_backward_f32(dout.data_ptr(), w_hh.data_ptr(), gates.data_ptr(),
                                                       cell.data_ptr(), dgx.data_ptr(), T, B, D, H, _stream(dout.device))
        _lib.check(st, "advstep_lstm_backward_f32")
        dx = _gemm(torch.mm, dgx.view(T * B, D * 4 * H), w_ih).view(T, B, I)
        return dx, None, None, None


def lstm_supported(hidden_size: int) -> bool:
    return bool(_lib.load().advstep_lstm_supported(hidden_size))


def lstm_layer(x: torch.Tensor, w_ih: torch.Tensor, w_hh: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """x (T, B, I) -> (T, B, D*H) for one LSTM layer with D directions (parameters packed per direction)."""
    return _LstmLayer.apply(x.contiguous(), w_ih, w_hh, bias)

class _IdKeyed:
    """Weak table keyed by tensor IDENTITY (tensors hash by id but compare elementwise: WeakKeyDictionary cannot hold them)."""

    def __init__(self):
        self._rows = {}

    def get(self, t):
        row = self._rows.get(id(t))
        return row[1] if row is not None and row[0]() is t else None

    def __setitem__(self, t, value):
        import weakref
        key = id(t)
        self._rows[key] = (weakref.ref(t, lambda _, k=key: self._rows.pop(k, None)), value)


_OVER_T = _IdKeyed()


class _LcnnTail(torch.autograd.Function):
    """Everything between LCNN's convolution trunk and its logit (src/models/lcnn.py:196-205) as one autograd node:
    pack -> [projection GEMM, recurrent kernel] x 2 -> skip + mean + Linear;  input gradient only.  12 launches forward +
    backward instead of ~20 (the permute copies, the skip add, the mean, the expanded mean gradient, the gradient sum)."""

    @staticmethod
    def forward(ctx, x4, w_ih1, w_hh1, b1, w_ih2, w_hh2, b2, w_out, b_out):
        _require(x4, "x4")
        B, C, T, W = x4.shape
        F = C * W
        D, H4, H = w_hh1.shape
        lib, dev = _lib.load(), x4.device
        st = _stream(dev)

        def layer(xin, w_ih, w_hh, bias):
            gx = _gemm(torch.addmm, xin.view(T * B, -1), w_ih.t(), bias)
            out = torch.empty((T, B, D * H), dtype=x4.dtype, device=dev)
            gates = torch.empty((T, B, D, H4), dtype=x4.dtype, device=dev)
            cell = torch.empty((T, B, D, H), dtype=x4.dtype, device=dev)
            with _Launch("lstm_forward", dev, tensors=(gx, out, gates, cell)):
                s_ = lib.advstep_lstm_forward_f32(gx.data_ptr(), w_hh.data_ptr(), out.data_ptr(), gates.data_ptr(), cell.data_ptr(),
                                                  T, B, D, H, st)
            _lib.check(s_, "advstep_lstm_forward_f32")
            return out, gates, cell

        xt = torch.empty((T, B, F), dtype=x4.dtype, device=dev)
        with _Launch("lcnn_tail_pack", dev, tensors=(x4, xt)):
            s_ = lib.advstep_lcnn_tail_pack_f32(x4.data_ptr(), xt.data_ptr(), B, C, T, W, st)
        _lib.check(s_, "advstep_lcnn_tail_pack_f32")
        out1, gates1, cell1 = layer(xt, w_ih1, w_hh1, b1)
        out2, gates2, cell2 = layer(out1, w_ih2, w_hh2, b2)
        z = torch.empty((B, 1), dtype=x4.dtype, device=dev)
        with _Launch("lcnn_tail_forward", dev, tensors=(out2, xt)):
            s_ = lib.advstep_lcnn_tail_forward_f32(out2.data_ptr(), xt.data_ptr(), w_out.data_ptr(),
                                                   b_out.data_ptr() if b_out is not None else None, z.data_ptr(), T, B, F, st)
        _lib.check(s_, "advstep_lcnn_tail_forward_f32")
        # w / T for the mean's gradient row: one launch per weight version, not per call.  Kept in a module-level weak table
        # (an attribute on the Parameter would be pickled by torch.save(model))
        cache = _OVER_T.get(w_out)
        key = (w_out._version, w_out.data_ptr(), T)
        if cache is None or cache[0] != key:
            cache = (key, (w_out.detach().reshape(1, F) / T).contiguous())
            _OVER_T[w_out] = cache
        ctx.save_for_backward(gates1, cell1, gates2, cell2, w_ih1, w_hh1, w_ih2, w_hh2, cache[1])
        ctx.dims = (B, C, T, W, D, H)
        return z

    @staticmethod
    def backward(ctx, dz):
        if any(ctx.needs_input_grad[1:]):
            raise RuntimeError("lcnn_tail provides the input gradient only; call it with frozen parameters")
        gates1, cell1, gates2, cell2, w_ih1, w_hh1, w_ih2, w_hh2, w_over_t = ctx.saved_tensors
        B, C, T, W, D, H = ctx.dims
        F = C * W
        lib, dev = _lib.load(), dz.device
        st = _stream(dev)
        # the mean's gradient dz[b] * (w / T)[f]: the same row for every frame — of the second layer's output and of the skip
        # connection — formed inside the two kernels that read it (one elementwise launch less)
        dzc = dz.reshape(B).contiguous()
        dgx2 = torch.empty((T, B, D, 4 * H), dtype=dz.dtype, device=dev)
        with _Launch("lstm_backward", dev, tensors=(gates2, cell2, dgx2)):
            s_ = lib.advstep_lstm_backward_outer_f32(dzc.data_ptr(), w_over_t.data_ptr(), w_hh2.data_ptr(), gates2.data_ptr(),
                                                     cell2.data_ptr(), dgx2.data_ptr(), T, B, D, H, st)
        _lib.check(s_, "advstep_lstm_backward_outer_f32")
        dout1 = _gemm(torch.mm, dgx2.view(T * B, D * 4 * H), w_ih2)                     # (T B, F) = d(first layer's output)
        dgx1 = torch.empty((T, B, D, 4 * H), dtype=dz.dtype, device=dev)
        with _Launch("lstm_backward", dev, tensors=(dout1, gates1, cell1, dgx1)):
            s_ = lib.advstep_lstm_backward_f32(dout1.data_ptr(), w_hh1.data_ptr(), gates1.data_ptr(), cell1.data_ptr(),
                                               dgx1.data_ptr(), T, B, D, H, st)
        _lib.check(s_, "advstep_lstm_backward_f32")
        dxt = _gemm(torch.mm, dgx1.view(T * B, D * 4 * H), w_ih1)
        dx4 = torch.empty((B, C, T, W), dtype=dz.dtype, device=dev)
        with _Launch("lcnn_tail_unpack_add", dev, tensors=(dxt, dx4)):
            s_ = lib.advstep_lcnn_tail_unpack_add_outer_f32(dxt.data_ptr(), dzc.data_ptr(), w_over_t.data_ptr(), dx4.data_ptr(), B, C,
                                                            T, W, st)
        _lib.check(s_, "advstep_lcnn_tail_unpack_add_outer_f32")
        return (dx4,) + (None,) * 8


def lcnn_tail_supported(features: int, hidden_size: int, out_features: int) -> bool:
    return lstm_supported(hidden_size) and 2 * hidden_size == features and features <= 256 and out_features == 1


def lcnn_tail(x4: torch.Tensor, packed1, packed2, w_out: torch.Tensor, b_out: Optional[torch.Tensor]) -> torch.Tensor:
    """x4 (B, C, T, W) conv-trunk output -> logits (B, 1): two BLSTM layers with the skip connection, mean over frames, Linear.
    packed1 / packed2 = (w_ih (D*4H, F), w_hh (D, 4H, H), bias (D*4H)) of the two layers (models/lcnn.py:BLSTMLayer._packed)."""
    return _LcnnTail.apply(x4.contiguous(), *packed1, *packed2, w_out.contiguous(), b_out)



class _GruLayer(torch.autograd.Function):
    """One (bi)directional GRU layer, sequence-first, zero initial state; input gradient only (csrc/specrnet_gru.hip)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        # x (T, B, I); w_ih (D*3H, I); w_hh (D, 3H, H); b_ih (D*3H); b_hh (D, 3H)
        for name, t in (("x", x), ("w_ih", w_ih), ("w_hh", w_hh), ("b_ih", b_ih), ("b_hh", b_hh)):
            _require(t, name)
        T, B, I = x.shape
        D, H3, H = w_hh.shape
        if H3 != 3 * H or tuple(w_ih.shape) != (D * H3, I) or b_ih.numel() != D * H3 or b_hh.numel() != D * H3:
            raise ValueError("inconsistent GRU parameter shapes")
        gx = _gemm(torch.addmm, x.reshape(T * B, I), w_ih.t(), b_ih)   # one GEMM for all steps and directions
        out = torch.empty((T, B, D * H), dtype=x.dtype, device=x.device)
        saved = torch.empty((T, B, D, 4 * H), dtype=x.dtype, device=x.device)
        with _Launch("gru_forward", x.device, tensors=(gx, out, saved)):
            st = _lib.load().advstep_gru_forward_f32(gx.data_ptr(), w_hh.data_ptr(), b_hh.data_ptr(), out.data_ptr(),
                                                     saved.data_ptr(), T, B, D, H, _stream(x.device))
        _lib.check(st, "advstep_gru_forward_f32")
        ctx.save_for_backward(saved, out, w_hh, w_ih)
        ctx.dims = (T, B, I, D, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        if any(ctx.needs_input_grad[1:]):
            raise RuntimeError("gru_layer provides the input gradient only; call it with frozen weights "
                               "(the model falls back to torch.nn.GRU otherwise)")
        saved, out, w_hh, w_ih = ctx.saved_tensors
        T, B, I, D, H = ctx.dims
        dout = dout.contiguous()
        dgx = torch.empty((T, B, D, 3 * H), dtype=dout.dtype, device=dout.device)
        with _Launch("gru_backward", dout.device, tensors=(dout, saved, out, dgx)):
            st = _lib.load().advstep_gru_backward_f32(dout.data_ptr(), w_hh.data_ptr(), saved.data_ptr(), out.data_ptr(),
                                                      dgx.data_ptr(), T, B, D, H, _stream(dout.device))
        _lib.check(st, "advstep_gru_backward_f32")
        dx = _gemm(torch.mm, dgx.view(T * B, D * 3 * H), w_ih).view(T, B, I)
        return dx, None, None, None, None


def gru_supported(hidden_size: int) -> bool:
    return bool(_lib.load().advstep_gru_supported(hidden_size))


def gru_layer(x: torch.Tensor, w_ih: torch.Tensor, w_hh: torch.Tensor, b_ih: torch.Tensor, b_hh: torch.Tensor) -> torch.Tensor:
    """x (T, B, I) -> (T, B, D*H) for one GRU layer with D directions (parameters packed per direction)."""
    return _GruLayer.apply(x.contiguous(), w_ih, w_hh, b_ih, b_hh)


def mfm(x: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None) -> torch.Tensor:
    """(N, 2C, H, W) -> (N, C, H, W): max(x[:, :C] + bias[:C], x[:, C:] + bias[C:]) [then (. - mean) * invstd]."""
    return _Mfm.apply(x.contiguous(), bias, bn)


def mfm_pool2(x: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None) -> torch.Tensor:
    """(N, 2C, H, W) -> (N, C, H//2, W//2): MaxPool2d(2, 2) of the max-feature-map [then (. - mean) * invstd]."""
    return _MfmPool2.apply(x.contiguous(), bias, bn)


def bn_eval_stats(bn: torch.nn.modules.batchnorm._BatchNorm):
    """(running_mean, 1 / sqrt(running_var + eps)) of an eval-mode, affine-free BatchNorm, cached on the module."""
    key = (bn.running_var.data_ptr(), bn.running_var._version, bn.running_mean._version, str(bn.running_var.device))
    if getattr(bn, "_advstep_key", None) != key:
        with torch.no_grad():
            invstd = (1.0 / torch.sqrt(bn.running_var + bn.eps)).contiguous()
        bn._advstep_key, bn._advstep_stats = key, (bn.running_mean.detach().contiguous(), invstd)
    return bn._advstep_stats
