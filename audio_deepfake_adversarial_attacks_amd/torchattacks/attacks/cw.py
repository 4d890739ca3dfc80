"""Carlini-Wagner L2 (reference: adversarial_attacks/torchattacks/attacks/cw.py:8-134)."""
import torch

from ..attack import Attack

# torch.optim.Adam defaults used by cw.py:68 (`optim.Adam([w], lr=self.lr)`)
_ADAM_BETA1, _ADAM_BETA2, _ADAM_EPS = 0.9, 0.999, 1e-8


class CW(Attack):
    r"""CW, 'Towards Evaluating the Robustness of Neural Networks' [https://arxiv.org/abs/1608.04644], L2.

    Arguments:
        model (nn.Module): model to attack.
        c (float): box-constraint weight: minimise ||1/2(tanh(w)+1) - x||^2 + c * f(1/2(tanh(w)+1)). (Default: 1e-4)
        kappa (float): confidence: f(x') = max(Z(x')_y - max{Z(x')_i : i != y}, -kappa). (Default: 0)
        steps (int): number of steps. (Default: 1000)
        lr (float): learning rate of the Adam optimizer. (Default: 0.01)

    .. warning:: With default c, you can't easily get adversarial images. Set higher c like 1.
    .. note:: Binary search for c is not implemented (as in the reference).

    Examples::
        >>> attack = torchattacks.CW(model, c=1e-4, kappa=0, steps=1000, lr=0.01)
        >>> adv_images = attack(images, labels)
    """

    def __init__(self, model, c=1e-4, kappa=0, steps=1000, lr=0.01):
        super().__init__("CW", model)
        self.c = c
        self.kappa = kappa
        self.steps = steps
        self.lr = lr
        self._supported_mode = ["default", "targeted"]
        self._early_stop = True

    def set_early_stop(self, enabled: bool = True) -> None:
        """Additive (not in the reference): False keeps iterating when the loss check of cw.py:105-110 would return early —
        for timing all `steps` iterations; the default reproduces the reference."""
        self._early_stop = bool(enabled)

    def forward(self, images, labels):
        ops = self.ops
        images, labels, target = self._prepare(images, labels)
        wanted = target if self._targeted else labels

        # cw.py:56-58: w = atanh(2x - 1); Adam state lives next to it (the optimiser is fused into one kernel)
        w = ops.cw_init_w(images)
        m = torch.zeros_like(w)
        v = torch.zeros_like(w)

        best_adv = images.clone()
        best_l2 = 1e10 * torch.ones(len(images), device=self.device)
        prev_cost = 1e10
        check_every = max(self.steps // 10, 1)
        adv_buf = None

        for step in range(self.steps):
            # cw.py:72-77: adv = 1/2 (tanh w + 1), per-utterance squared L2 distance — one fused pass
            adv, current_l2 = ops.cw_tanh_sqdist(w, images, adv_out=adv_buf)

            # cw.py:79-85: model term c * sum f, differentiated w.r.t. adv by autograd
            adv.requires_grad_(True)
            z = self.model(adv)
            outputs = torch.cat([-z, z], dim=1)
            f_loss = self.f(outputs, wanted).sum()
            (grad_adv,) = torch.autograd.grad(self.c * f_loss, adv)
            adv = adv_buf = adv.detach()  # the next step's tanh pass overwrites this buffer

            # cw.py:87-91: cost = sum L2 + c * f; chain rule through tanh space + Adam(w), fused
            cost = current_l2.sum() + self.c * f_loss.detach()
            ops.cw_adam_step(w, m, v, images, grad_adv.contiguous(), step + 1, self.lr, _ADAM_BETA1, _ADAM_BETA2,
                             _ADAM_EPS)

            # cw.py:93-103: keep the closest adversarial that fools the model
            pre = outputs.detach().max(1)[1]
            correct = (pre == labels).float()
            mask = (1 - correct) * (best_l2 > current_l2).float()
            best_l2 = mask * current_l2 + (1 - mask) * best_l2
            ops.cw_best_update(adv, mask.contiguous(), best_adv)

            # cw.py:105-110: early stop when the loss stops decreasing (one host sync per check)
            if step % check_every == 0:
                cost_now = cost.item()
                if cost_now > prev_cost and self._early_stop:
                    return best_adv
                prev_cost = cost_now

        return best_adv

    def tanh_space(self, x):
        return 1 / 2 * (torch.tanh(x) + 1)

    def inverse_tanh_space(self, x):
        return self.atanh(x * 2 - 1)

    def atanh(self, x):
        return 0.5 * torch.log((1 + x) / (1 - x))

    def f(self, outputs, labels):
        """cw.py:125-134 — one-hot built on the logits' device (the reference builds it on the CPU first)."""
        one_hot = torch.eye(outputs.shape[1], device=outputs.device, dtype=outputs.dtype)[labels]
        other = ((1 - one_hot) * outputs).max(dim=1)[0]
        true = torch.masked_select(outputs, one_hot.bool())
        if self._targeted:
            return torch.clamp(other - true, min=-self.kappa)
        return torch.clamp(true - other, min=-self.kappa)
