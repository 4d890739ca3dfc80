"""`Attack` base class — the drop-in boundary (reference: adversarial_attacks/torchattacks/attack.py:5-331).

Same public surface as the reference: ``Attack(name, model)``, ``atk(images, labels)``, the mode setters,
``set_training_mode``, ``set_return_type``, ``save`` and ``__str__``.  What differs is below the surface:
subclasses do their waveform arithmetic through ``self.ops`` — the HIP kernels of libadvstep.so
(``hip_ops``).  There is no CPU implementation in this package; tests inject the oracle's op table.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch

_TARGETED_DEFAULT = "targeted"
_TARGETED_LEAST_LIKELY = "targeted(least-likely)"
_TARGETED_RANDOM = "targeted(random)"


def _frozen_layer(module: torch.nn.Module, batchnorm_training: bool, dropout_training: bool) -> bool:
    """attack.py:313-319 — layers forced back to eval while the rest of the model is put in train mode."""
    cls = module.__class__.__name__
    return (not batchnorm_training and "BatchNorm" in cls) or (not dropout_training and "Dropout" in cls)


class Attack(object):
    r"""Base class for all attacks (attack.py:5-13).

    The device is taken from the model's first parameter; by default the model is switched to eval mode
    for the duration of an attack call (see `set_training_mode`)."""

    # the attack's inner loop can replay from a hipGraph (torchattacks/graphed.py): PGD, PGDL2.  evaluation.generate_attacks
    # keeps two batches in flight for those (additive attribute; the reference has no counterpart)
    replays_from_graph = False

    def __init__(self, name, model):
        # attack.py:14-35
        self.attack = name
        self.model = model
        self.model_name = str(model).split("(")[0]
        self.device = next(model.parameters()).device

        self._attack_mode = "default"
        self._targeted = False
        self._return_type = "float"
        self._supported_mode = ["default"]

        self._model_training = False
        self._batchnorm_training = False
        self._dropout_training = False

        # Not in the reference: the table of waveform ops (HIP kernels) and an optional explicit random-start
        # draw for parity runs.  Underscore-prefixed so `__str__` matches the reference's output.
        self._ops = None
        self._init_noise = None

    # ---- op table ----------------------------------------------------------------------------------------

    @property
    def ops(self):
        """Waveform op table.  Defaults to the HIP kernels; bound lazily so constructing an attack on a box
        without the library still raises at the first call, loudly, rather than silently doing something else."""
        if self._ops is None:
            from .. import hip_ops
            hip_ops._lib.load()
            self._ops = hip_ops
        return self._ops

    @ops.setter
    def ops(self, table):
        self._ops = table

    def set_init_noise(self, draw) -> None:
        """Use an explicit random-start draw for the next calls (None = in-kernel Philox).
        PGD: noise (B,T) ~ U(-eps, eps) as in pgd.py:56.  PGDL2: (normal (B,T), r (B)) as in pgdl2.py:57,60."""
        self._init_noise = draw

    def _fresh_seed(self) -> int:
        """One 62-bit Philox key per attack call, drawn from torch's global CPU generator (so
        `torch.manual_seed` / `set_seed` make random starts reproducible; no device sync)."""
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    # ---- to be overridden ----------------------------------------------------------------------------------

    def forward(self, *input):
        # attack.py:37-42
        raise NotImplementedError

    # ---- attack mode (attack.py:44-110) --------------------------------------------------------------------

    def get_mode(self):
        return self._attack_mode

    def set_mode_default(self):
        self._attack_mode = "default"
        self._targeted = False
        print("Attack mode is changed to 'default.'")

    def _enter_targeted(self, mode: str, fn: Optional[Callable]):
        if "targeted" not in self._supported_mode:
            raise ValueError("Targeted mode is not supported.")
        self._attack_mode = mode
        self._targeted = True
        self._target_map_function = fn
        print(f"Attack mode is changed to '{mode}.'")

    def set_mode_targeted_by_function(self, target_map_function=None):
        self._enter_targeted(_TARGETED_DEFAULT, target_map_function)

    def set_mode_targeted_least_likely(self, kth_min=1):
        if "targeted" not in self._supported_mode:
            raise ValueError("Targeted mode is not supported.")
        assert kth_min > 0
        self._kth_min = kth_min
        self._enter_targeted(_TARGETED_LEAST_LIKELY, self._get_least_likely_label)

    def set_mode_targeted_random(self):
        self._enter_targeted(_TARGETED_RANDOM, self._get_random_target_label)

    def set_return_type(self, type):
        # attack.py:112-130
        if type not in ("float", "int"):
            raise ValueError(type + " is not a valid type. [Options: float, int]")
        self._return_type = type

    def set_training_mode(self, model_training=False, batchnorm_training=False, dropout_training=False):
        # attack.py:132-147 — RNNs need train mode for their backward pass
        self._model_training = model_training
        self._batchnorm_training = batchnorm_training
        self._dropout_training = dropout_training

    # ---- bulk generation helper (attack.py:149-233) ------------------------------------------------------------

    def save(self, data_loader, save_path=None, verbose=True, return_verbose=False, save_pred=False):
        keep = save_path is not None
        adv_all, label_all, pred_all = [], [], []
        correct = total = 0
        l2_parts = []
        n_batches = len(data_loader)

        was_training = self.model.training
        requested_type, self._return_type = self._return_type, "float"
        rob_acc = l2 = elapsed = progress = 0.0

        for step, (images, labels) in enumerate(data_loader):
            t0 = time.time()
            adv = self.__call__(images, labels)
            n = len(images)

            if verbose or return_verbose:
                with torch.no_grad():
                    if was_training:
                        self.model.eval()
                    pred = self.model(adv).data.max(1)[1]
                    total += labels.size(0)
                    right = pred == labels.to(self.device)
                    correct += int(right.sum())
                    elapsed = time.time() - t0
                    diff = (adv - images.to(self.device)).view(n, -1)
                    l2_parts.append(torch.norm(diff[~right], p=2, dim=1))
                    rob_acc = 100 * float(correct) / total
                    l2 = torch.cat(l2_parts).mean().item()
                    progress = (step + 1) / n_batches * 100
                    if verbose:
                        self._save_print(progress, rob_acc, l2, elapsed, end="\r")

            if keep:
                adv_cpu = adv.detach().cpu()
                adv_all.append(self._to_uint(adv_cpu) if requested_type == "int" else adv_cpu)
                label_all.append(labels.detach().cpu())
                payload = [torch.cat(adv_all, 0), torch.cat(label_all, 0)]
                if save_pred:
                    pred_all.append(pred.detach().cpu())
                    payload.append(torch.cat(pred_all, 0))
                torch.save(tuple(payload), save_path)

        if verbose:
            self._save_print(progress, rob_acc, l2, elapsed, end="\n")
        if was_training:
            self.model.train()
        # (attack.py:173-175 switches the return type to 'float' for the duration of save() and never switches it back: kept —
        # tests/golden/attack_save.npz records the reference leaving 'float' behind after an 'int' save)
        if return_verbose:
            return rob_acc, l2, elapsed

    def _save_print(self, progress, rob_acc, l2, elapsed_time, end):
        print("- Save progress: %2.2f %% / Robust accuracy: %2.2f %% / L2: %1.5f (%2.3f it/s) \t"
              % (progress, rob_acc, l2, elapsed_time), end=end)

    # ---- target-label helpers (attack.py:235-282) ----------------------------------------------------------------

    @torch.no_grad()
    def _get_target_label(self, images, labels=None):
        if not self._targeted:
            raise ValueError("Please define target_map_function.")
        was_training = self.model.training
        if was_training:
            self.model.eval()
        target = self._target_map_function(images, labels)
        if was_training:
            self.model.train()
        return target

    def _other_classes(self, outputs, labels):
        if labels is None:
            labels = outputs.max(dim=1)[1]
        return labels, outputs.shape[-1]

    @torch.no_grad()
    def _get_least_likely_label(self, images, labels=None):
        outputs = self.model(images)
        labels, n_classes = self._other_classes(outputs, labels)
        target = torch.zeros_like(labels)
        for b in range(labels.shape[0]):
            others = [c for c in range(n_classes) if c != int(labels[b])]
            _, t = torch.kthvalue(outputs[b][others], self._kth_min)
            target[b] = others[t]
        return target.long().to(self.device)

    @torch.no_grad()
    def _get_random_target_label(self, images, labels=None):
        outputs = self.model(images)
        labels, n_classes = self._other_classes(outputs, labels)
        target = torch.zeros_like(labels)
        for b in range(labels.shape[0]):
            others = [c for c in range(n_classes) if c != int(labels[b])]
            t = (len(others) * torch.rand([1])).long()
            target[b] = others[t]
        return target.long().to(self.device)

    def _to_uint(self, images):
        # attack.py:284-289
        return (images * 255).type(torch.uint8)

    def __str__(self):
        # attack.py:291-306 — public hyper-parameters only
        info = {k: v for k, v in self.__dict__.items() if not k.startswith("_") and k not in ("model", "attack")}
        info["attack_mode"] = self._attack_mode
        info["return_type"] = self._return_type
        return self.attack + "(" + ", ".join("{}={}".format(k, v) for k, v in info.items()) + ")"

    # ---- the call (attack.py:308-331) -------------------------------------------------------------------------------

    def __call__(self, *input, **kwargs):
        was_training = self.model.training

        if self._model_training:
            self.model.train()
            for _, module in self.model.named_modules():
                if _frozen_layer(module, self._batchnorm_training, self._dropout_training):
                    module.eval()
        else:
            self.model.eval()

        # Not in the reference (no observable difference: the attacks only ever differentiate w.r.t. the input):
        # parameters are frozen for the duration of the call, so no parameter-gradient bookkeeping is built and the
        # model's fused kernels may fold bias adds (models/lcnn.py::_transform).
        frozen = [p for p in self.model.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        try:
            images = self.forward(*input, **kwargs)
        finally:
            for p in frozen:
                p.requires_grad_(True)

        if was_training:
            self.model.train()
        if self._return_type == "int":
            images = self._to_uint(images)
        return images

    # ---- shared by the gradient attacks -------------------------------------------------------------------------------

    def _prepare(self, images, labels):
        """fgsm.py:37-41 / pgd.py:44-48: private copies on the model's device (+ target labels if targeted)."""
        images = images.clone().detach().to(self.device)
        labels = labels.clone().detach().to(self.device)
        target = self._get_target_label(images, labels) if self._targeted else None
        if images.dtype != torch.float32:
            raise TypeError(f"images must be float32 waveforms, got {images.dtype}")
        return images.contiguous(), labels, target

    def _input_gradient(self, adv, labels, target):
        """One forward + input-backward of the attacked model (pgd.py:60-72).

        The reference builds `cat([-z, z], 1)` and back-propagates `CrossEntropyLoss` (or its negative when
        targeted); here d cost / d z comes from the closed-form kernel (`ops.ce2_loss_grad`) and only the model
        itself is differentiated by autograd.  Returns (grad w.r.t. adv, cost (1,))."""
        adv.requires_grad_(True)
        z = self.model(adv)
        if z.dim() != 2 or z.shape[1] != 1:
            raise ValueError(f"the attacked model must emit one logit per utterance, got {tuple(z.shape)}")
        wanted = target if self._targeted else labels
        dz, cost = self.ops.ce2_loss_grad(z.detach().contiguous(), wanted.to(torch.int64).contiguous(),
                                          -1.0 if self._targeted else 1.0)
        (grad,) = torch.autograd.grad(z, adv, grad_outputs=dz.view_as(z), retain_graph=False, create_graph=False)
        return grad.contiguous(), cost
