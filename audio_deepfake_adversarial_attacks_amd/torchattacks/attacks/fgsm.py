"""FGSM (reference: adversarial_attacks/torchattacks/attacks/fgsm.py:7-62)."""
from ..attack import Attack


class FGSM(Attack):
    r"""FGSM, 'Explaining and harnessing adversarial examples' [https://arxiv.org/abs/1412.6572], L-inf.

    Arguments:
        model (nn.Module): model to attack, (B, T) waveform in [0, 1] -> (B, 1) logit.
        eps (float): maximum perturbation. (Default: 0.007)

    Examples::
        >>> attack = torchattacks.FGSM(model, eps=0.007)
        >>> adv_images = attack(images, labels)
    """

    def __init__(self, model, eps=0.007):
        super().__init__("FGSM", model)
        self.eps = eps
        self._supported_mode = ["default", "targeted"]

    def forward(self, images, labels):
        images, labels, target = self._prepare(images, labels)
        grad, _ = self._input_gradient(images, labels, target)      # fgsm.py:45-57
        # fgsm.py:59-60 fused: clamp(x + eps * sign(g), 0, 1)
        return self.ops.fgsm_step(images.detach(), grad, self.eps)
