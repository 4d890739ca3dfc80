"""RawNet3 detector (reference: src/models/rawnet3.py:11-291): raw waveform (B, T) -> logit (B, 1).

Pre-emphasis -> InstanceNorm -> parameterised sinc encoder -> log|.| -> mean-norm -> three Res2Net-style
`Bottle2neck` blocks with AFMS -> 1x1 conv -> channel/context attentive statistics pooling -> BN -> FC.
Module / parameter names follow the reference so its checkpoints load; everything after the sinc encoder
is plain torch and runs forward + input-backward under PyTorch-ROCm (1x1 Conv1d = GEMMs on MFMA)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .sincfb import Encoder, ParamSincFB


class PreEmphasis(nn.Module):
    """y[t] = x[t] - coef * x[t-1] with reflect padding on the left (rawnet3.py:140-158)."""

    def __init__(self, coef: float = 0.97) -> None:
        super().__init__()
        self.coef = coef
        self.register_buffer("flipped_filter", torch.FloatTensor([-self.coef, 1.0]).unsqueeze(0).unsqueeze(0))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        assert len(input.size()) == 2, "The number of dimensions of input tensor must be 2!"
        if input.is_cuda and input.shape[1] >= 2 and _elementwise_preemphasis_enabled():
            # the same two-tap filter as elementwise arithmetic, in the convolution's order (tap 0 first): MIOpen runs this
            # convolution with its naive reference kernel (0.2 ms forward, 0.27 ms backward at B = 64)
            f0, f1 = self.flipped_filter[0, 0, 0], self.flipped_filter[0, 0, 1]
            prev = torch.cat([input[:, 1:2], input[:, :-1]], dim=1)            # reflect padding on the left: x[-1] := x[1]
            return (prev * f0 + input * f1).unsqueeze(1)
        return F.conv1d(F.pad(input.unsqueeze(1), (1, 0), "reflect"), self.flipped_filter)


class AFMS(nn.Module):
    """Alpha-feature-map scaling: (x + alpha) * sigmoid(fc(mean_t x))   (rawnet3.py:161-182)."""

    def __init__(self, nb_dim: int) -> None:
        super().__init__()
        self.alpha = nn.Parameter(torch.ones((nb_dim, 1)))
        self.fc = nn.Linear(nb_dim, nb_dim)
        self.sig = nn.Sigmoid()

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and _fused_elem_enabled() and _afms_enabled():
            frozen = not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
            if frozen:
                from .. import detector_ops as D
                return D.afms(x, self.alpha.detach(), self.fc.weight.detach(), None if self.fc.bias is None else self.fc.bias.detach())
        y = F.adaptive_avg_pool1d(x, 1).view(x.size(0), -1)
        y = self.sig(self.fc(y)).view(x.size(0), x.size(1), -1)
        return (x + self.alpha) * y


def _elementwise_preemphasis_enabled() -> bool:
    """ADVSTEP_RAWNET3_PREEMPH=0 keeps the two-tap convolution (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_RAWNET3_PREEMPH", "1") != "0"


def _gemm_conv1d_enabled() -> bool:
    import os
    return os.environ.get("ADVSTEP_RAWNET3_GEMM_CONV", "1") != "0"


def _inplace_conv1d_enabled() -> bool:
    """ADVSTEP_RAWNET3_INPLACE_CONV=0 keeps the autograd formulation over a padded copy (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_RAWNET3_INPLACE_CONV", "1") != "0"


def _chain_enabled() -> bool:
    """ADVSTEP_RAWNET3_CHAIN=0 runs the Res2Net branches as separate torch ops (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_RAWNET3_CHAIN", "1") != "0"


def _afms_enabled() -> bool:
    """ADVSTEP_RAWNET3_AFMS=0 keeps AFMS as torch ops (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_RAWNET3_AFMS", "1") != "0"


def _context_split_enabled() -> bool:
    """ADVSTEP_RAWNET3_CONTEXT=0 builds the concatenated context tensor for the attention convolution (A/B); default on."""
    import os
    return os.environ.get("ADVSTEP_RAWNET3_CONTEXT", "1") != "0"


def _tail_enabled() -> bool:
    """ADVSTEP_RAWNET3_TAIL=0 keeps activation and `+= residual` / MaxPool1d as two passes (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_RAWNET3_TAIL", "1") != "0"


def _fused_elem_enabled() -> bool:
    import os
    return os.environ.get("ADVSTEP_RAWNET3_ELEM", "1") != "0"


def _conv_relu_bn(x: torch.Tensor, conv: nn.Conv1d, bn: nn.BatchNorm1d) -> torch.Tensor:
    """`bn(relu(conv(x)))` (rawnet3.py:240-242, 252-254, 262-264).  On a HIP tensor with frozen parameters and an eval-mode
    BatchNorm the convolution runs without its bias and  relu(. + bias[c]) * scale[c] + shift[c]  is ONE pass
    (detector_ops.relu_affine) instead of bias add, ReLU and BatchNorm kernels over activations of up to 1.7 GB."""
    if x.is_cuda and x.dtype == torch.float32 and _fused_elem_enabled():
        from .. import detector_ops as D
        frozen = not (torch.is_grad_enabled() and (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)
                                                   or any(p.requires_grad for p in bn.parameters())))
        if frozen and D.foldable_bn(bn):
            scale, shift = D.bn_eval_affine(bn)
            h = _same_conv1d(x, conv, with_bias=False)
            return D.relu_affine(h, scale, shift, conv.bias.detach() if conv.bias is not None else None)
    return bn(torch.relu(_same_conv1d(x, conv)))


def _add_pool(a: torch.Tensor, b, pool: nn.MaxPool1d) -> torch.Tensor:
    """`pool(a + b)` (b may be None): fused on a HIP tensor (detector_ops.add_maxpool1d: one byte per output instead of
    int64 indices, the sum never written), the torch modules otherwise."""
    if a.is_cuda and a.dtype == torch.float32 and _fused_elem_enabled():
        from .. import detector_ops as D
        if D.maxpool1d_supported(pool):
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            return D.add_maxpool1d(a, b, k)
    return pool(a if b is None else a + b)


def _tap_ranges(k: int, d: int, T: int):
    """(tap j, output range, input range) of a "same" dilated convolution: y[t] += W_j x[t + (j - k//2) d] where both exist."""
    for j in range(k):
        o = (j - k // 2) * d
        if abs(o) >= T:
            continue
        yield (j, slice(0, T), slice(0, T)) if o == 0 else ((j, slice(-o, T), slice(0, T + o)) if o < 0 else
                                                            (j, slice(0, T - o), slice(o, T)))


def _dilated_gemm(x, weight, bias, d: int, transpose: bool = False, out=None):
    """The dilated "same" convolution (transpose=False) or its input gradient (transpose=True) as k GEMMs accumulating in
    place into sub-ranges of ONE buffer — `out` (any (B, C, T) view with unit last stride, e.g. a channel slice) or a new tensor.
    x may be a channel slice too.  No autograd."""
    B, _, T = x.shape
    taps = weight if isinstance(weight, (list, tuple)) else None      # per-tap (Cout, Cin) matrices, already contiguous
    k = len(taps) if taps is not None else weight.shape[-1]
    y = None
    for j, ys, xs in sorted(_tap_ranges(k, d, T), key=lambda r: r[0] != k // 2):      # the full-range (centre) tap first
        if transpose:
            ys, xs = xs, ys                                                          # tap j moves gradient from y-range to x-range
        wj = taps[j] if taps is not None else weight[:, :, j]                        # (a (Cout, Cin, k) slice has no unit stride:
        wj = (wj.t() if transpose else wj).unsqueeze(0).expand(B, -1, -1)            #  bmm copies it, B times, on every call)
        if y is None:
            if bias is not None:
                y = torch.baddbmm(bias.view(1, -1, 1), wj, x[:, :, xs], out=out) if out is not None else \
                    torch.baddbmm(bias.view(1, -1, 1), wj, x[:, :, xs])
            else:
                y = torch.bmm(wj, x[:, :, xs], out=out) if out is not None else torch.bmm(wj, x[:, :, xs])
        else:
            y[:, :, ys].baddbmm_(wj, x[:, :, xs])
    return y


class _SameConv1dFrozen(torch.autograd.Function):
    """The dilated "same" Conv1d as k GEMMs that accumulate IN PLACE into sub-ranges of one output buffer — no zero-padded
    copy of the input — and the same for its input gradient (the transposed taps into sub-ranges of one gradient buffer).
    Differentiable w.r.t. the input only: built for the attack loop, where the weights are frozen.  Through autograd the
    shifted views of a padded input cost, per convolution and direction, a pad / slice copy plus (backward) three zero-filled
    full-size buffers, three slice copies and two adds around the three GEMMs: 10 ms of an 81 ms iteration at B = 64."""

    @staticmethod
    def forward(ctx, x, weight, bias, d):
        ctx.save_for_backward(weight)
        ctx.d = d
        return _dilated_gemm(x, weight, bias, d)

    @staticmethod
    def backward(ctx, g):
        (weight,) = ctx.saved_tensors
        return _dilated_gemm(g, weight, None, ctx.d, transpose=True), None, None, None


class _Res2NetChain(torch.autograd.Function):
    """The hierarchical branches of a Bottle2neck (src/models/rawnet3.py:244-258) on the (B, scale * width, T) tensor that leaves
    conv1 / bn1:   sp_0 = spx[0],  sp_i = piece_{i-1} + spx[i],  piece_i = bns[i](relu(convs[i](sp_i))),
    out = cat(piece_0 .. piece_{nums-1}, spx[nums])   — input gradient only (parameters frozen).
    Per branch: the convolution's k GEMMs (`_dilated_gemm`) and ONE elementwise pass each way (`advstep_res2net_link_*`): the
    activation is written straight into its slice of the concatenated tensor together with the next branch's input, and on
    the way back the two gradients of a piece are summed inside the activation's backward while the branch's input gradient
    lands in its slice of d(input) — no `torch.cat`, no strided adds, no gradient-accumulation adds."""

    @staticmethod
    def forward(ctx, out, plan):
        from .. import detector_ops as D
        B, _, T = out.shape
        w, nums = plan["width"], plan["nums"]
        cat = torch.empty_like(out)
        hs, inp = [], out[:, :w]
        for i in range(nums):
            h = _dilated_gemm(inp, plan["weight"][i], None, plan["d"])
            nxt = out[:, (i + 1) * w:(i + 2) * w] if i + 1 < nums else None
            z = torch.empty((B, w, T), dtype=out.dtype, device=out.device) if nxt is not None else None
            D.res2net_link_forward(h, plan["scale"][i], plan["shift"][i], plan["bias"][i], cat[:, i * w:(i + 1) * w], nxt, z)
            hs.append(h)
            inp = z
        cat[:, nums * w:].copy_(out[:, nums * w:])
        ctx.save_for_backward(*hs)
        ctx.plan = plan
        return cat

    @staticmethod
    def backward(ctx, g):
        from .. import detector_ops as D
        plan, hs = ctx.plan, ctx.saved_tensors
        w, nums = plan["width"], plan["nums"]
        g = g.contiguous()
        gout = torch.empty_like(g)
        carry = None
        for i in reversed(range(nums)):
            gh = D.res2net_link_backward(g[:, i * w:(i + 1) * w], carry, hs[i], plan["scale"][i], plan["shift"][i], plan["bias"][i])
            carry = gout[:, i * w:(i + 1) * w]
            _dilated_gemm(gh, plan["weight"][i], None, plan["d"], transpose=True, out=carry)
        gout[:, nums * w:].copy_(g[:, nums * w:])
        return gout, None


def _same_conv1d(x: torch.Tensor, conv: nn.Conv1d, with_bias: bool = True) -> torch.Tensor:
    """`conv(x)` for the dilated "same" Conv1d(width, width, 3, dilation=d, padding=d) of the Res2Net branches.

    On a HIP device MIOpen (no tuned solver for these 128-channel dilated 1-D convolutions) runs them with its NAIVE
    reference kernel — 5.9 ms each, 22 per forward: 130 ms of a 190 ms RawNet3 iteration at B = 64.  The same arithmetic
    as k GEMMs over shifted ranges, W[:, :, j] (Cout x Cin) . x[:, :, t + (j - k//2) d], goes to rocBLAS / hipBLASLt instead:
    with frozen parameters (an attack is running) through `_SameConv1dFrozen` (in-place accumulation, no padded copy,
    hand-written input gradient), otherwise as baddbmm over views of the padded input with autograd.  CPU tensors and any
    other geometry take the module itself."""
    k, d = conv.kernel_size[0], conv.dilation[0]
    bias = conv.bias if with_bias else None
    if not (x.is_cuda and _gemm_conv1d_enabled() and conv.stride == (1,) and conv.groups == 1 and k % 2 == 1 and k > 1
            and conv.padding == ((k // 2) * d,) and conv.padding_mode == "zeros"):
        # the module's own forward path, so a non-zero padding_mode keeps its reflect / circular padding
        return conv(x) if (with_bias or conv.bias is None) else conv._conv_forward(x, conv.weight, None)
    frozen = not (torch.is_grad_enabled() and (conv.weight.requires_grad or (bias is not None and bias.requires_grad)))
    if frozen and x.dtype == torch.float32 and x.shape[-1] > (k // 2) * d and _inplace_conv1d_enabled():
        return _SameConv1dFrozen.apply(x, conv.weight.detach(), None if bias is None else bias.detach(), d)
    T = x.shape[-1]
    xp = torch.nn.functional.pad(x, ((k // 2) * d, (k // 2) * d))
    B = x.shape[0]
    # bias and the running sum ride in the GEMM epilogue (C operand of baddbmm): no separate add kernels
    out = bias.view(1, -1, 1).expand(B, -1, T) if bias is not None else None
    for j in range(k):
        wj = conv.weight[:, :, j].unsqueeze(0).expand(B, -1, -1)
        xj = xp[:, :, j * d:j * d + T]
        out = torch.bmm(wj, xj) if out is None else torch.baddbmm(out, wj, xj)
    return out


class _PointwiseConv1dFrozen(torch.autograd.Function):
    """Conv1d(kernel 1) as one batched GEMM each way, input gradient only.  For `layer4` (3072 -> 1536 channels over 429 frames)
    MIOpen picks a composable-kernel backward-data solver that needs four layout transposes around it (2.8 ms at B = 64)."""

    @staticmethod
    def forward(ctx, x, w2, bias):
        ctx.save_for_backward(w2)
        B = x.shape[0]
        wb = w2.unsqueeze(0).expand(B, -1, -1)
        return torch.bmm(wb, x) if bias is None else torch.baddbmm(bias.view(1, -1, 1), wb, x)

    @staticmethod
    def backward(ctx, g):
        (w2,) = ctx.saved_tensors
        return torch.bmm(w2.t().unsqueeze(0).expand(g.shape[0], -1, -1), g), None, None


def _pointwise_conv1d(x: torch.Tensor, conv: nn.Conv1d) -> torch.Tensor:
    frozen = not (torch.is_grad_enabled() and (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)))
    if (frozen and x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (1,) and conv.stride == (1,)
            and conv.padding == (0,) and conv.groups == 1 and _gemm_conv1d_enabled() and _inplace_conv1d_enabled()):
        return _PointwiseConv1dFrozen.apply(x, conv.weight.detach()[:, :, 0], None if conv.bias is None else conv.bias.detach())
    return conv(x)


class Bottle2neck(nn.Module):
    """Res2Net bottleneck over time with hierarchical dilated convs (rawnet3.py:185-274)."""

    def __init__(self, inplanes, planes, kernel_size=None, dilation=None, scale=4, pool=False):
        super().__init__()
        width = int(math.floor(planes / scale))
        self.conv1 = nn.Conv1d(inplanes, width * scale, kernel_size=1)
        self.bn1 = nn.BatchNorm1d(width * scale)
        self.nums = scale - 1
        pad = math.floor(kernel_size / 2) * dilation
        self.convs = nn.ModuleList(
            [nn.Conv1d(width, width, kernel_size=kernel_size, dilation=dilation, padding=pad) for _ in range(self.nums)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(width) for _ in range(self.nums)])
        self.conv3 = nn.Conv1d(width * scale, planes, kernel_size=1)
        self.bn3 = nn.BatchNorm1d(planes)
        self.relu = nn.ReLU()
        self.width = width
        self.mp = nn.MaxPool1d(pool) if pool else False
        self.afms = AFMS(planes)
        if inplanes != planes:
            self.residual = nn.Sequential(nn.Conv1d(inplanes, planes, kernel_size=1, stride=1, bias=False))
        else:
            self.residual = nn.Identity()

    def forward(self, x):
        residual = self.residual(x)
        out = _conv_relu_bn(x, self.conv1, self.bn1)

        plan = self._chain_plan(out)
        if plan is not None:
            out = _Res2NetChain.apply(out.contiguous(), plan)
            return self._tail(out, residual)
        groups = torch.split(out, self.width, 1)
        pieces, carry = [], None
        for i in range(self.nums):
            carry = groups[i] if i == 0 else carry + groups[i]
            carry = _conv_relu_bn(carry, self.convs[i], self.bns[i])
            pieces.append(carry)
        pieces.append(groups[self.nums])
        out = torch.cat(pieces, 1)

        return self._tail(out, residual)

    def _tail(self, out, residual):
        if (self.mp and out.is_cuda and out.dtype == torch.float32 and _fused_elem_enabled() and _tail_enabled()
                and isinstance(residual, torch.Tensor)):
            from .. import detector_ops as D
            conv, bn = self.conv3, self.bn3
            frozen = not (torch.is_grad_enabled() and (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)
                                                       or any(p.requires_grad for p in bn.parameters())))
            if frozen and D.foldable_bn(bn) and D.maxpool1d_supported(self.mp):
                # conv3 -> relu -> bn3 -> `+= residual` -> MaxPool1d: the convolution without its bias, everything else one pass
                scale, shift = D.bn_eval_affine(bn)
                h = _same_conv1d(out, conv, with_bias=False)
                if h.shape == residual.shape:
                    k = self.mp.kernel_size if isinstance(self.mp.kernel_size, int) else self.mp.kernel_size[0]
                    return self.afms(D.tail_pool1d(h, residual, scale, shift, conv.bias.detach() if conv.bias is not None else None, k))
                return self.afms(_add_pool(D.relu_affine(h, scale, shift, conv.bias.detach() if conv.bias is not None else None),
                                           residual, self.mp))
        out = _conv_relu_bn(out, self.conv3, self.bn3)
        if self.mp:
            out = _add_pool(out, residual, self.mp)        # `out += residual` -> MaxPool1d, one pass on a HIP tensor
        else:
            out = out + residual
        return self.afms(out)

    def _chain_plan(self, out):
        """Frozen, foldable parameters on a HIP tensor: the per-branch constants `_Res2NetChain` needs, cached until a
        parameter changes; None = run the branches as torch ops."""
        if not (out.is_cuda and out.dtype == torch.float32 and _fused_elem_enabled() and _gemm_conv1d_enabled()
                and _inplace_conv1d_enabled() and _chain_enabled()):
            return None
        from .. import detector_ops as D
        convs, bns = list(self.convs), list(self.bns)
        params = [p for m in convs + bns for p in m.parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return None
        c0 = convs[0]
        k, d = c0.kernel_size[0], c0.dilation[0]
        same = all(c.kernel_size == (k,) and c.dilation == (d,) and c.stride == (1,) and c.groups == 1 and c.padding == ((k // 2) * d,)
                   and c.padding_mode == "zeros" and c.in_channels == c.out_channels == self.width for c in convs)
        if not (same and k % 2 == 1 and k > 1 and out.shape[-1] > (k // 2) * d and all(D.foldable_bn(b) for b in bns)):
            return None
        tensors = params + [b for m in bns for b in m.buffers()]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if getattr(self, "_chain_key", None) != key:
            affine = [D.bn_eval_affine(b) for b in bns]
            self._chain_key = key
            self._chain_val = {"width": self.width, "nums": self.nums, "d": d,
                               "weight": [[c.weight.detach()[:, :, j].contiguous() for j in range(k)] for c in convs],
                               "bias": [None if c.bias is None else c.bias.detach() for c in convs],
                               "scale": [a[0] for a in affine], "shift": [a[1] for a in affine]}
        return self._chain_val


class RawNet3(nn.Module):
    def __init__(self, block, model_scale, context, summed, C=1024, **kwargs):
        super().__init__()
        nOut = kwargs["nOut"]
        self.context = context
        self.encoder_type = kwargs["encoder_type"]
        self.log_sinc = kwargs["log_sinc"]
        self.norm_sinc = kwargs["norm_sinc"]
        self.out_bn = kwargs["out_bn"]
        self.summed = summed

        self.preprocess = nn.Sequential(PreEmphasis(), nn.InstanceNorm1d(1, eps=1e-4, affine=True))
        self.conv1 = Encoder(ParamSincFB(C // 4, 251, stride=kwargs["sinc_stride"]))
        self.relu = nn.ReLU()
        self.bn1 = nn.BatchNorm1d(C // 4)

        self.layer1 = block(C // 4, C, kernel_size=3, dilation=2, scale=model_scale, pool=5)
        self.layer2 = block(C, C, kernel_size=3, dilation=3, scale=model_scale, pool=3)
        self.layer3 = block(C, C, kernel_size=3, dilation=4, scale=model_scale)
        self.layer4 = nn.Conv1d(3 * C, 1536, kernel_size=1)

        attn_input = 1536 * 3 if self.context else 1536
        print("self.encoder_type", self.encoder_type)
        if self.encoder_type == "ECA":
            attn_output = 1536
        elif self.encoder_type == "ASP":
            attn_output = 1
        else:
            raise ValueError("Undefined encoder")

        self.attention = nn.Sequential(
            nn.Conv1d(attn_input, 128, kernel_size=1), nn.ReLU(), nn.BatchNorm1d(128),
            nn.Conv1d(128, attn_output, kernel_size=1), nn.Softmax(dim=2))
        self.bn5 = nn.BatchNorm1d(3072)
        self.fc6 = nn.Linear(3072, nOut)
        self.bn6 = nn.BatchNorm1d(nOut)
        self.mp3 = nn.MaxPool1d(3)

    def forward(self, x):
        """x: (B, samples)   (rawnet3.py:73-137)."""
        x = self.conv1(self.preprocess(x))
        fused_norm = False
        if self.log_sinc and self.norm_sinc == "mean" and x.is_cuda and x.dtype == torch.float32 and _fused_elem_enabled():
            from .. import detector_ops as D
            if D.log_meannorm_supported(x.shape[-1]):
                x = D.log_meannorm(x, 1e-6)               # abs, + 1e-6, log, mean, subtract: one pass each way
                fused_norm = True
        if not fused_norm:
            x = torch.abs(x)
            if self.log_sinc:
                x = torch.log(x + 1e-6)
        if fused_norm:
            pass
        elif self.norm_sinc == "mean":
            x = x - torch.mean(x, dim=-1, keepdim=True)
        elif self.norm_sinc == "mean_std":
            m = torch.mean(x, dim=-1, keepdim=True)
            s = torch.std(x, dim=-1, keepdim=True).clamp(min=0.001)
            x = (x - m) / s

        x1 = self.layer1(x)
        x2 = self.layer2(x1)
        x1p = _add_pool(x1, None, self.mp3)      # the reference evaluates mp3(x1) twice (:96, :102): same values, once here
        x3 = self.layer3(x1p + x2) if self.summed else self.layer3(x2)
        x = self.relu(_pointwise_conv1d(torch.cat((x1p, x2, x3), dim=1), self.layer4))

        t = x.size()[-1]
        first = self.attention[0]
        if (self.context and x.is_cuda and _context_split_enabled() and isinstance(first, nn.Conv1d) and first.kernel_size == (1,)
                and first.stride == (1,) and first.padding == (0,) and first.groups == 1 and first.in_channels == 3 * x.shape[1]):
            # `attention[0](cat(x, mean.repeat(t), std.repeat(t)))` without the repeated / concatenated (B, 4608, t) tensor: a
            # kernel-1 convolution of time-constant channels is a per-utterance constant,
            #     W [x; mean; std] = W_x x + (W_mean mean + W_std std + bias) 1^T,
            # so the GEMM runs over the 1536 real channels only and the two statistics enter as its (B, 128, 1) C operand.
            # Plain torch ops (autograd, parameter gradients included); the reduction order differs from the module's.
            C = x.shape[1]
            mean = torch.mean(x, dim=2)
            std = torch.sqrt(torch.var(x, dim=2).clamp(min=1e-4, max=1e4))
            W = first.weight[:, :, 0]
            const = mean @ W[:, C:2 * C].t() + std @ W[:, 2 * C:].t()
            if first.bias is not None:
                const = const + first.bias
            h = torch.baddbmm(const.unsqueeze(2), W[:, :C].unsqueeze(0).expand(x.shape[0], -1, -1), x)
            w = self.attention[1:](h)
        else:
            if self.context:
                mean = torch.mean(x, dim=2, keepdim=True).repeat(1, 1, t)
                std = torch.sqrt(torch.var(x, dim=2, keepdim=True).clamp(min=1e-4, max=1e4)).repeat(1, 1, t)
                global_x = torch.cat((x, mean, std), dim=1)
            else:
                global_x = x
            w = self.attention(global_x)

        if x.is_cuda and x.dtype == torch.float32 and w.shape == x.shape and _fused_elem_enabled():
            from .. import detector_ops as D
            mu, m2 = D.weighted_stats(x, w)                   # both weighted sums in one pass each way
        else:
            mu, m2 = torch.sum(x * w, dim=2), torch.sum((x ** 2) * w, dim=2)
        sg = torch.sqrt((m2 - mu ** 2).clamp(min=1e-4, max=1e4))
        x = self.fc6(self.bn5(torch.cat((mu, sg), 1)))
        return self.bn6(x) if self.out_bn else x


def prepare_model():
    """rawnet3.py:277-291 — the configuration the repository uses."""
    return RawNet3(Bottle2neck, model_scale=8, context=True, summed=True, encoder_type="ECA", nOut=1, out_bn=False,
                   sinc_stride=10, log_sinc=True, norm_sinc="mean", grad_mult=1)
