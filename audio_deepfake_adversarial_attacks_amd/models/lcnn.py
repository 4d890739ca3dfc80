"""LCNN detector (reference: src/models/lcnn.py:24-243; upstream ASVspoof2021 LFCC-LCNN baseline).

Same architecture, same `state_dict` key names (`m_transform.<i>.*`, `m_before_pooling.<i>.l_blstm.*`,
`m_output_act.*`, `frontend.*`), so reference checkpoints load unchanged.  Forward + input-backward run
under PyTorch-ROCm (MIOpen convolutions / RNN); tests/test_models.py pins `BaseLCNN` against logits and input
gradients produced by the reference's own BaseLCNN (tests/golden/lcnn_body.npz)."""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import frontends


def _fused_mfm_enabled() -> bool:
    """ADVSTEP_LCNN_FUSED=0 switches the HIP max-feature-map kernels off (A/B measurements); default on."""
    return os.environ.get("ADVSTEP_LCNN_FUSED", "1") != "0"


def _fused_tail_enabled() -> bool:
    return os.environ.get("ADVSTEP_LCNN_TAIL", "1") != "0"


class BLSTMLayer(nn.Module):
    """Bidirectional LSTM that keeps (batch, length, dim) on both sides (lcnn.py:24-46)."""

    def __init__(self, input_dim: int, output_dim: int):
        super().__init__()
        if output_dim % 2 != 0:
            raise ValueError(f"BLSTMLayer expects an even layer size, got {output_dim}")
        self.l_blstm = nn.LSTM(input_dim, output_dim // 2, bidirectional=True)

    def _packed(self):
        """Per-direction parameters packed for lcnn_ops.lstm_layer, cached until a parameter changes."""
        lstm = self.l_blstm
        names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
        params = [getattr(lstm, n + s) for s in ("", "_reverse") for n in names]
        key = tuple((p.data_ptr(), p._version) for p in params)
        if getattr(self, "_pack_key", None) != key:
            with torch.no_grad():
                w_ih = torch.cat([lstm.weight_ih_l0, lstm.weight_ih_l0_reverse], dim=0).contiguous()
                w_hh = torch.stack([lstm.weight_hh_l0, lstm.weight_hh_l0_reverse], dim=0).contiguous()
                bias = torch.cat([lstm.bias_ih_l0 + lstm.bias_hh_l0,
                                  lstm.bias_ih_l0_reverse + lstm.bias_hh_l0_reverse], dim=0).contiguous()
            self._pack_key, self._pack = key, (w_ih, w_hh, bias)
        return self._pack

    def forward(self, x):
        lstm = self.l_blstm
        frozen = not (torch.is_grad_enabled() and any(p.requires_grad for p in lstm.parameters()))
        if x.is_cuda and frozen and _fused_lstm_enabled() and lstm.num_layers == 1 and lstm.bidirectional \
                and lstm.bias and lstm.proj_size == 0:
            from .. import lcnn_ops
            if lcnn_ops.lstm_supported(lstm.hidden_size):
                # one workgroup per (utterance, direction) instead of MIOpen's per-time-step kernel chain
                out = lcnn_ops.lstm_layer(x.permute(1, 0, 2), *self._packed())
                return out.permute(1, 0, 2)
        out, _ = lstm(x.permute(1, 0, 2))  # the LSTM is sequence-first
        return out.permute(1, 0, 2)


class MaxFeatureMap2D(nn.Module):
    """Max-feature-map: split `max_dim` in two halves and keep the element-wise maximum (lcnn.py:49-95)."""

    def __init__(self, max_dim: int = 1):
        super().__init__()
        self.max_dim = max_dim

    def forward(self, inputs):
        shape = list(inputs.size())
        if self.max_dim >= len(shape):
            raise ValueError(f"MaxFeatureMap: cannot maximise dim {self.max_dim} of a {len(shape)}-d input")
        if shape[self.max_dim] % 2 != 0:
            raise ValueError(f"MaxFeatureMap: dim {self.max_dim} has an odd size {shape[self.max_dim]}")
        shape[self.max_dim] //= 2
        shape.insert(self.max_dim, 2)
        return inputs.view(*shape).max(self.max_dim)[0]


# (kind, args) rows of `m_transform`, in the reference's order so the Sequential indices (= state_dict keys)
# match lcnn.py:120-157.  conv: (in, out, kernel, padding); bn: channels (affine=False).
_TRANSFORM = (
    ("conv", (None, 64, 5, 2)), ("mfm",), ("pool",),
    ("conv", (32, 64, 1, 0)), ("mfm",), ("bn", 32),
    ("conv", (32, 96, 3, 1)), ("mfm",), ("pool",), ("bn", 48),
    ("conv", (48, 96, 1, 0)), ("mfm",), ("bn", 48),
    ("conv", (48, 128, 3, 1)), ("mfm",), ("pool",),
    ("conv", (64, 128, 1, 0)), ("mfm",), ("bn", 64),
    ("conv", (64, 64, 3, 1)), ("mfm",), ("bn", 32),
    ("conv", (32, 64, 1, 0)), ("mfm",), ("bn", 32),
    ("conv", (32, 64, 3, 1)), ("mfm",), ("pool",),
    ("dropout", 0.7),
)


def _make_transform(input_channels: int) -> nn.Sequential:
    layers = []
    for row in _TRANSFORM:
        kind = row[0]
        if kind == "conv":
            cin, cout, k, pad = row[1]
            layers.append(nn.Conv2d(input_channels if cin is None else cin, cout, (k, k), 1, padding=(pad, pad)))
        elif kind == "mfm":
            layers.append(MaxFeatureMap2D())
        elif kind == "pool":
            layers.append(nn.MaxPool2d((2, 2), (2, 2)))
        elif kind == "bn":
            layers.append(nn.BatchNorm2d(row[1], affine=False))
        else:
            layers.append(nn.Dropout(row[1]))
    return nn.Sequential(*layers)


def _fused_conv0_enabled() -> bool:
    """ADVSTEP_LCNN_CONV0=0 keeps MIOpen's convolution for the first block (A/B measurements); default on."""
    return os.environ.get("ADVSTEP_LCNN_CONV0", "1") != "0"


def _fused_bn_enabled() -> bool:
    """ADVSTEP_LCNN_BN=0 keeps ATen's batch-norm kernels after the fused blocks (A/B measurements); default on."""
    return os.environ.get("ADVSTEP_LCNN_BN", "1") != "0"


def _foldable_bn(mod) -> bool:
    return (isinstance(mod, nn.BatchNorm2d) and not mod.training and not mod.affine and mod.track_running_stats
            and mod.running_mean is not None)


def _fused_lstm_enabled() -> bool:
    """ADVSTEP_LCNN_LSTM=0 keeps MIOpen's RNN path (A/B measurements); default on."""
    return os.environ.get("ADVSTEP_LCNN_LSTM", "1") != "0"


def _fused_conv1x1_enabled() -> bool:
    """ADVSTEP_LCNN_CONV1X1=0 keeps MIOpen's GEMM for the 1x1 blocks (A/B measurements); default on."""
    return os.environ.get("ADVSTEP_LCNN_CONV1X1", "1") != "0"


def _fused_conv3x3_enabled() -> bool:
    """ADVSTEP_LCNN_CONV3X3=0 keeps MIOpen's convolution for the 3x3 + pool blocks (A/B measurements); default on."""
    return os.environ.get("ADVSTEP_LCNN_CONV3X3", "1") != "0"


def _is_same_conv3x3(conv: nn.Conv2d) -> bool:
    return (_pair(conv.kernel_size) == (3, 3) and _pair(conv.stride) == (1, 1) and _pair(conv.padding) == (1, 1)
            and _pair(conv.dilation) == (1, 1) and conv.groups == 1)


def _is_pointwise_conv(conv: nn.Conv2d) -> bool:
    return (_pair(conv.kernel_size) == (1, 1) and _pair(conv.stride) == (1, 1) and _pair(conv.padding) == (0, 0)
            and _pair(conv.dilation) == (1, 1) and conv.groups == 1 and conv.out_channels % 2 == 0)


def _is_first_block_conv(conv: nn.Conv2d) -> bool:
    return (conv.in_channels == 1 and _pair(conv.kernel_size) == (5, 5) and _pair(conv.stride) == (1, 1)
            and _pair(conv.padding) == (2, 2) and _pair(conv.dilation) == (1, 1) and conv.groups == 1
            and conv.out_channels % 2 == 0)


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _is_pool2(pool: nn.MaxPool2d) -> bool:
    return (_pair(pool.kernel_size) == (2, 2) and _pair(pool.stride) == (2, 2) and _pair(pool.padding) == (0, 0)
            and _pair(pool.dilation) == (1, 1) and not pool.ceil_mode and not pool.return_indices)


class BaseLCNN(nn.Module):
    """Spectrogram (B, C, n_coeff, frames) -> logit (B, 1)   (lcnn.py:102-217)."""

    def __init__(self, **kwargs):
        super().__init__()
        input_channels = kwargs.get("input_channels", 1)
        self.num_coefficients = kwargs.get("num_coefficients", 80)
        self.v_emd_dim = 1

        self.m_transform = _make_transform(input_channels)
        width = (self.num_coefficients // 16) * 32
        self.m_before_pooling = nn.Sequential(BLSTMLayer(width, width), BLSTMLayer(width, width))
        self.m_output_act = nn.Linear(width, self.v_emd_dim)

    def _transform(self, x):
        """`self.m_transform(x)`.  On a HIP device every Conv2d -> MaxFeatureMap2D [-> MaxPool2d(2, 2)] run of the
        Sequential goes through the fused kernels of lcnn_ops: one pass over the conv
        output instead of ATen's max-reduce (+ int64 indices) and pool kernels, and — when the bias needs no gradient,
        which is the case inside an attack — the conv's bias add is folded in as well.  Bit-identical to the plain path.
        CPU tensors (oracle / CPU-baseline runs) take the plain Sequential."""
        if not (x.is_cuda and _fused_mfm_enabled()):
            return self.m_transform(x)
        from .. import lcnn_ops
        mods = list(self.m_transform)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (isinstance(m, nn.Conv2d) and m.padding_mode == "zeros" and isinstance(nxt, MaxFeatureMap2D)
                    and nxt.max_dim == 1):
                after = mods[i + 2] if i + 2 < len(mods) else None
                params_frozen = not (torch.is_grad_enabled() and (m.weight.requires_grad or
                                                                  (m.bias is not None and m.bias.requires_grad)))
                if (params_frozen and _is_first_block_conv(m) and isinstance(after, nn.MaxPool2d) and _is_pool2(after)
                        and _fused_conv0_enabled()):
                    # one-channel 5x5 conv + MFM + pool in ONE kernel: the (B, 64, 404, 80) conv output never exists
                    x = lcnn_ops.conv5_mfm_pool2(x, m.weight, m.bias)
                    i += 3
                    continue
                pooled = isinstance(after, nn.MaxPool2d) and _is_pool2(after)
                consumed = 3 if pooled else 2
                # an eval-mode BatchNorm2d(affine=False) right after the block is folded into the kernel's epilogue
                tail = mods[i + consumed] if i + consumed < len(mods) else None
                bn = None
                if _foldable_bn(tail) and _fused_bn_enabled():
                    bn = lcnn_ops.bn_eval_stats(tail)
                    consumed += 1
                if (params_frozen and _is_pointwise_conv(m) and lcnn_ops.conv1x1_mfm_supported(m.in_channels)
                        and not isinstance(after, nn.MaxPool2d) and _fused_conv1x1_enabled()):
                    # 1x1 conv + bias + MFM (+ BN) in ONE kernel: the 2C-channel conv output never exists
                    x = lcnn_ops.conv1x1_mfm(x, m.weight, m.bias, bn)
                    i += consumed
                    continue
                if (params_frozen and _is_same_conv3x3(m) and _fused_conv3x3_enabled()
                        and lcnn_ops.conv3x3_supported(m.in_channels, m.out_channels)):
                    # 3x3 conv (Winograd on the matrix cores) + bias + MFM [+ pool] (+ BN) in ONE kernel
                    x = (lcnn_ops.conv3x3_mfm_pool2 if pooled else lcnn_ops.conv3x3_mfm)(x, m.weight, m.bias, bn)
                    i += consumed
                    continue
                fold_bias = m.bias is not None and not (torch.is_grad_enabled() and m.bias.requires_grad)
                h = F.conv2d(x, m.weight, None if fold_bias else m.bias, m.stride, m.padding, m.dilation, m.groups)
                bias = m.bias if fold_bias else None
                x = lcnn_ops.mfm_pool2(h, bias, bn) if pooled else lcnn_ops.mfm(h, bias, bn)
                i += consumed
            else:
                x = m(x)
                i += 1
        return x

    def _fused_tail(self, hidden4):
        """The two BLSTM layers, the skip connection, the mean over frames and the read-out as ONE autograd node
        (lcnn_ops.lcnn_tail): HIP tensor, frozen parameters (an attack or a scoring pass), the reference's layer shapes."""
        if not (hidden4.is_cuda and hidden4.dtype == torch.float32 and _fused_tail_enabled() and _fused_lstm_enabled()):
            return None
        layers = list(self.m_before_pooling)
        if len(layers) != 2 or not all(isinstance(m, BLSTMLayer) for m in layers) or not isinstance(self.m_output_act, nn.Linear):
            return None
        if hidden4.shape[2] == 0:          # an input too short for the trunk: the plain path's behaviour, not EINVAL
            return None
        # the node bypasses these modules' forward(): anyone listening on them (feature extraction, debugging hooks) must
        # keep being called, so the plain path runs instead
        bypassed = [self.m_before_pooling, self.m_output_act] + layers + [m.l_blstm for m in layers]
        if any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None)
               for m in bypassed):
            return None
        # ... and so must hooks registered for EVERY module (torch.nn.modules.module.register_module_forward_hook and its pre /
        # backward variants): they fire from Module.__call__, which the node does not go through either
        from torch.nn.modules import module as _nn_module
        if any(getattr(_nn_module, name, None) for name in ("_global_forward_hooks", "_global_forward_pre_hooks",
                                                            "_global_forward_hooks_always_called", "_global_backward_hooks",
                                                            "_global_backward_pre_hooks")):
            return None
        params = [p for m in layers for p in m.parameters()] + list(self.m_output_act.parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return None
        from .. import lcnn_ops
        feats = hidden4.shape[1] * hidden4.shape[3]
        lstms = [m.l_blstm for m in layers]
        if not all(l.num_layers == 1 and l.bidirectional and l.bias and l.proj_size == 0 and l.input_size == feats
                   and 2 * l.hidden_size == feats for l in lstms):
            return None
        if not lcnn_ops.lcnn_tail_supported(feats, lstms[0].hidden_size, self.m_output_act.out_features):
            return None
        # (the Parameter itself, not a detached alias: lcnn_tail caches w / T on the tensor object it is handed)
        return lcnn_ops.lcnn_tail(hidden4, layers[0]._packed(), layers[1]._packed(), self.m_output_act.weight,
                                  self.m_output_act.bias)

    def _compute_embedding(self, x):
        batch_size = x.shape[0]
        # (B, C, coeff, frames) -> (B, C, frames, coeff) -> conv trunk -> (B, frames', C' * coeff')   (:190-199)
        hidden = self._transform(x.permute(0, 1, 3, 2))
        fused = self._fused_tail(hidden) if hidden.dim() == 4 else None
        if fused is not None:
            return fused
        hidden = hidden.permute(0, 2, 1, 3).contiguous()
        hidden = hidden.view(batch_size, hidden.shape[1], -1)
        # two BLSTMs with a skip connection, mean over frames, linear read-out   (:202-205)
        lstm = self.m_before_pooling(hidden)
        return self.m_output_act((lstm + hidden).mean(1))

    def _compute_score(self, feature_vec):
        return torch.sigmoid(feature_vec).squeeze(1)

    def forward(self, x):
        return self._compute_embedding(x)


class LCNN(BaseLCNN):
    """Waveform (B, T) -> logit (B, 1): frontend + BaseLCNN   (lcnn.py:221-243)."""

    def __init__(self, device: str = "cuda", **kwargs):
        super().__init__(**kwargs)
        self.device = device
        frontend_name = kwargs.get("frontend_algorithm", [])
        self.frontend = frontends.get_frontend(frontend_name)
        print(f"Using {frontend_name} frontend")

    def _compute_frontend(self, x):
        feats = self.frontend(x)
        return feats.unsqueeze(1) if feats.ndim < 4 else feats  # (B, 1|n, n_coeff, frames)

    def forward(self, x):
        return self._compute_embedding(self._compute_frontend(x))
