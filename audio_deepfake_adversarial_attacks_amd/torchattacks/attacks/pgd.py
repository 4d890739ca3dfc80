"""PGD, L-inf (reference: adversarial_attacks/torchattacks/attacks/pgd.py:7-78)."""
from .. import graphed
from ..attack import Attack


class PGD(Attack):
    r"""PGD, 'Towards Deep Learning Models Resistant to Adversarial Attacks' [https://arxiv.org/abs/1706.06083].

    Arguments:
        model (nn.Module): model to attack.
        eps (float): maximum perturbation. (Default: 0.3)
        alpha (float): step size. (Default: 2/255)
        steps (int): number of steps. (Default: 40)
        random_start (bool): using random initialization of delta. (Default: True)

    Examples::
        >>> attack = torchattacks.PGD(model, eps=8/255, alpha=1/255, steps=40, random_start=True)
        >>> adv_images = attack(images, labels)
    """

    replays_from_graph = True

    def __init__(self, model, eps=0.3, alpha=2 / 255, steps=40, random_start=True):
        super().__init__("PGD", model)
        self.eps = eps
        self.alpha = alpha
        self.steps = steps
        self.random_start = random_start
        self._supported_mode = ["default", "targeted"]

    def forward(self, images, labels):
        ops = self.ops
        images, labels, target = self._prepare(images, labels)

        if self.random_start:
            # pgd.py:54-57: clamp(x + U(-eps, eps), 0, 1); the draw is generated inside the kernel (Philox)
            # unless an explicit draw was installed with set_init_noise()
            if self._init_noise is not None:
                adv = ops.pgd_linf_init(images, self.eps, noise=self._init_noise.to(self.device).contiguous())
            else:
                adv = ops.pgd_linf_init(images, self.eps, seed=self._fresh_seed())
        else:
            adv = images.clone()

        # pgd.py:60-76, `steps` times: model forward + input-backward, then the fused sign step / eps-ball projection around
        # `images` / [0, 1] clamp, writing where the model is not reading (ping-pong).  Replayed from a hipGraph once the
        # same (model state, shape) has been seen twice (graphed.py); eager otherwise — bit-identical either way.
        def step(cur, grad, orig, out):
            ops.pgd_linf_step(cur, grad, orig, self.alpha, self.eps, out=out)

        return graphed.run_iterations(self, adv, images, labels, target, self.steps, step, (self.eps, self.alpha))
