"""SpecRNet detector (reference: src/models/specrnet.py:23-214; https://github.com/piotrkawa/specrnet).

Same architecture and `state_dict` keys as the reference, including its two quirks: `Residual_block2D`
feeds `conv1` with the block INPUT (the bn1/leaky-relu result is discarded, specrnet.py:75-81) and
`get_config` is mutated while the blocks are built (:106-107).  Pinned against the reference's BaseSpecRNet
logits in tests/test_models.py (tests/golden/specrnet_body.npz)."""
from typing import Dict

import torch
import torch.nn as nn

from .. import frontends


def get_config(input_channels: int) -> Dict:
    return {
        "filts": [input_channels, [input_channels, 20], [20, 64], [64, 64]],
        "nb_fc_node": 64,
        "gru_node": 64,
        "nb_gru_layer": 2,
        "nb_classes": 1,
    }


class Residual_block2D(nn.Module):
    def __init__(self, nb_filts, first=False):
        super().__init__()
        cin, cout = nb_filts
        self.first = first
        if not first:
            self.bn1 = nn.BatchNorm2d(num_features=cin)  # parameters exist (checkpoint keys); output unused
        self.lrelu = nn.LeakyReLU(negative_slope=0.3)
        self.conv1 = nn.Conv2d(cin, cout, kernel_size=3, padding=1, stride=1)
        self.bn2 = nn.BatchNorm2d(num_features=cout)
        self.conv2 = nn.Conv2d(cout, cout, kernel_size=3, padding=1, stride=1)
        self.downsample = cin != cout
        if self.downsample:
            self.conv_downsample = nn.Conv2d(cin, cout, kernel_size=1, padding=0, stride=1)
        self.mp = nn.MaxPool2d(2)

    def _fused(self, x) -> bool:
        """The fused kernels of detector_ops apply: HIP tensor, frozen parameters (an attack is running: only the input
        gradient is needed, so conv biases and BatchNorm terms are per-channel constants), eval-mode bn2."""
        if not (x.is_cuda and x.dtype == torch.float32 and _fused_elem_enabled()):
            return False
        from .. import detector_ops
        frozen = not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
        return frozen and detector_ops.foldable_bn(self.bn2)

    def _forward_fused(self, x):
        """Same arithmetic as `forward` with the full-resolution elementwise passes folded:
        conv1 bias + bn2 + LeakyReLU in one pass; conv2 bias + downsample bias + residual add + MaxPool2d in one pass."""
        import torch.nn.functional as F
        from .. import detector_ops as D
        down = self.conv_downsample if self.downsample else None
        if (_fused_conv_enabled() and x.dim() == 4 and D.res_block_supported(self.conv1, self.conv2, down)
                and D.res_block_shape_supported(x.shape, self.conv1.out_channels)):
            # the whole block on the matrix cores (detector_ops._ResBlock): downsample inside conv2's reduction, pooling in
            # its epilogue, transposed convolutions through the same kernel on the way back
            return D.res_block(x, D.res_block_plan(self, self.conv1, self.bn2, self.conv2, down, self.lrelu.negative_slope))
        scale, shift = D.bn_eval_affine(self.bn2)
        if self.conv1.bias is not None:
            shift = shift + self.conv1.bias.detach() * scale
        h = F.conv2d(x, self.conv1.weight, None, 1, 1)
        h = D.affine_lrelu(h, scale, shift, self.lrelu.negative_slope)
        h = F.conv2d(h, self.conv2.weight, None, 1, 1)
        bias = self.conv2.bias.detach() if self.conv2.bias is not None else None
        if self.downsample:
            identity = F.conv2d(x, self.conv_downsample.weight, None)
            if self.conv_downsample.bias is not None:
                bias = self.conv_downsample.bias.detach() if bias is None else bias + self.conv_downsample.bias.detach()
        else:
            identity = x
        return D.add_maxpool2(h, identity, bias)

    def forward(self, x):
        # specrnet.py:73-91.  NB conv1 consumes x, not lrelu(bn1(x)) — reference behaviour, kept.
        if self._fused(x):
            return self._forward_fused(x)
        if not self.first and self.bn1.training:
            # the reference computes bn1(x) and discards it (:75-78); in train mode that still updates bn1's running
            # statistics, which end up in checkpoints — keep the side effect, skip the wasted pass otherwise
            with torch.no_grad():
                self.bn1(x)
        out = self.conv2(self.lrelu(self.bn2(self.conv1(x))))
        identity = self.conv_downsample(x) if self.downsample else x
        return self.mp(out + identity)


def _fused_elem_enabled() -> bool:
    """ADVSTEP_SPECRNET_ELEM=0 keeps the plain ATen / MIOpen elementwise chains (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_SPECRNET_ELEM", "1") != "0"


def _fused_conv_enabled() -> bool:
    """ADVSTEP_SPECRNET_CONV=0 keeps ATen / MIOpen convolutions inside the residual blocks (A/B measurements); default on."""
    import os
    return os.environ.get("ADVSTEP_SPECRNET_CONV", "1") != "0"


def _fused_gru_enabled() -> bool:
    import os
    return os.environ.get("ADVSTEP_SPECRNET_GRU", "1") != "0"


class BaseSpecRNet(nn.Module):
    """Spectrogram (B, C, 80, frames) -> logit (B, 1)   (specrnet.py:94-190)."""

    def __init__(self, d_args, **kwargs):
        super().__init__()
        self.device = kwargs.get("device", "cuda")
        filts = d_args["filts"]

        self.first_bn = nn.BatchNorm2d(num_features=filts[0])
        self.selu = nn.SELU(inplace=True)
        self.block0 = nn.Sequential(Residual_block2D(nb_filts=filts[1], first=True))
        self.block2 = nn.Sequential(Residual_block2D(nb_filts=filts[2]))
        filts[2][0] = filts[2][1]
        self.block4 = nn.Sequential(Residual_block2D(nb_filts=filts[2]))
        self.avgpool = nn.AdaptiveAvgPool2d(1)

        self.fc_attention0 = self._make_attention_fc(filts[1][-1], filts[1][-1])
        self.fc_attention2 = self._make_attention_fc(filts[2][-1], filts[2][-1])
        self.fc_attention4 = self._make_attention_fc(filts[2][-1], filts[2][-1])

        self.bn_before_gru = nn.BatchNorm2d(num_features=filts[2][-1])
        self.gru = nn.GRU(input_size=filts[2][-1], hidden_size=d_args["gru_node"],
                          num_layers=d_args["nb_gru_layer"], batch_first=True, bidirectional=True)
        self.fc1_gru = nn.Linear(d_args["gru_node"] * 2, d_args["nb_fc_node"] * 2)
        self.fc2_gru = nn.Linear(d_args["nb_fc_node"] * 2, d_args["nb_classes"], bias=True)
        self.sig = nn.Sigmoid()
        self.pool = nn.MaxPool2d(2)

    def _attend(self, feats, fc):
        """Squeeze-excite style gate used after every block: x * y + y  (specrnet.py:145-149)."""
        gate = self.sig(fc(self.avgpool(feats).view(feats.size(0), -1)))
        gate = gate.view(gate.size(0), gate.size(1), 1, 1)
        return feats * gate + gate

    def _attend_pool(self, feats, fc):
        """`self.pool(self._attend(feats, fc))` (specrnet.py:163-172); on a HIP tensor the gate multiply-add and the 2x2 pooling
        are one pass (detector_ops.gate_maxpool2: differentiable in the features and in the gate)."""
        if feats.is_cuda and feats.dtype == torch.float32 and _fused_elem_enabled():
            from .. import detector_ops
            lin = fc[0] if isinstance(fc, nn.Sequential) and len(fc) == 1 else fc
            if (isinstance(lin, nn.Linear) and feats.dim() == 4
                    and not (torch.is_grad_enabled() and any(p.requires_grad for p in lin.parameters()))):
                return detector_ops.attend_pool(feats, lin.weight.detach(), None if lin.bias is None else lin.bias.detach())
            gate = self.sig(fc(self.avgpool(feats).view(feats.size(0), -1)))
            return detector_ops.gate_maxpool2(feats, gate)
        return self.pool(self._attend(feats, fc))

    def _bn_selu(self, x, bn):
        """`self.selu(bn(x))` (specrnet.py:159, 173); eval-mode BatchNorm with frozen parameters on a HIP tensor: one pass."""
        if x.is_cuda and x.dtype == torch.float32 and _fused_elem_enabled():
            from .. import detector_ops as D
            frozen = not (torch.is_grad_enabled() and any(p.requires_grad for p in bn.parameters()))
            if frozen and D.foldable_bn(bn):
                scale, shift = D.bn_eval_affine(bn)
                return D.affine_selu(x, scale, shift)
        return self.selu(bn(x))

    def _compute_embedding(self, x):
        x = self._bn_selu(x, self.first_bn)
        x = self._attend_pool(self.block0(x), self.fc_attention0)
        x = self._attend_pool(self.block2(x), self.fc_attention2)
        x = self._attend_pool(self.block4(x), self.fc_attention4)
        x = self._bn_selu(x, self.bn_before_gru)
        x = x.squeeze(-2).permute(0, 2, 1)
        x = self._run_gru(x)
        return self.fc2_gru(self.fc1_gru(x[:, -1, :]))

    def _packed_gru(self):
        """Per-layer, per-direction GRU parameters packed for lcnn_ops.gru_layer, cached until a parameter changes."""
        gru = self.gru
        params = list(gru.parameters())
        key = tuple((p.data_ptr(), p._version) for p in params)
        if getattr(self, "_gru_key", None) != key:
            with torch.no_grad():
                packed = []
                for layer in range(gru.num_layers):
                    sfx = [f"_l{layer}", f"_l{layer}_reverse"]
                    w_ih = torch.cat([getattr(gru, "weight_ih" + s) for s in sfx], dim=0).contiguous()
                    w_hh = torch.stack([getattr(gru, "weight_hh" + s) for s in sfx], dim=0).contiguous()
                    b_ih = torch.cat([getattr(gru, "bias_ih" + s) for s in sfx], dim=0).contiguous()
                    b_hh = torch.stack([getattr(gru, "bias_hh" + s) for s in sfx], dim=0).contiguous()
                    packed.append((w_ih, w_hh, b_ih, b_hh))
            self._gru_key, self._gru_packed = key, packed
        return self._gru_packed

    def _run_gru(self, x):
        """(B, T, C) -> (B, T, 2H).  With frozen parameters on a HIP device (an attack is running: only the input gradient
        is needed) each layer's recurrence is one kernel per direction pair (csrc/specrnet_gru.hip) instead of MIOpen's
        per-time-step kernel chain; otherwise torch.nn.GRU."""
        gru = self.gru
        frozen = not (torch.is_grad_enabled() and any(p.requires_grad for p in gru.parameters()))
        if x.is_cuda and frozen and _fused_gru_enabled() and gru.bidirectional and gru.bias and gru.dropout == 0.0:
            from .. import lcnn_ops
            if lcnn_ops.gru_supported(gru.hidden_size):
                seq = x.permute(1, 0, 2)
                for w_ih, w_hh, b_ih, b_hh in self._packed_gru():
                    seq = lcnn_ops.gru_layer(seq, w_ih, w_hh, b_ih, b_hh)
                return seq.permute(1, 0, 2)
        gru.flatten_parameters()
        out, _ = gru(x)
        return out

    def forward(self, x):
        return self._compute_embedding(x)

    def _make_attention_fc(self, in_features, l_out_features):
        return nn.Sequential(nn.Linear(in_features=in_features, out_features=l_out_features))


class SpecRNet(BaseSpecRNet):
    """Waveform (B, T) -> logit (B, 1)   (specrnet.py:193-214)."""

    def __init__(self, d_args, **kwargs):
        super().__init__(d_args, **kwargs)
        self.device = kwargs["device"]
        frontend_name = kwargs.get("frontend_algorithm", [])
        self.frontend = frontends.get_frontend(frontend_name)
        print(f"Using {frontend_name} frontend")

    def _compute_frontend(self, x):
        feats = self.frontend(x)
        return feats.unsqueeze(1) if feats.ndim < 4 else feats

    def forward(self, x):
        return self._compute_embedding(self._compute_frontend(x))
