"""PGD, L2 (reference: adversarial_attacks/torchattacks/attacks/pgdl2.py:7-90)."""
from .. import graphed
from ..attack import Attack


class PGDL2(Attack):
    r"""PGD with an L2 ball, 'Towards Deep Learning Models Resistant to Adversarial Attacks'.

    Arguments:
        model (nn.Module): model to attack.
        eps (float): maximum perturbation. (Default: 1.0)
        alpha (float): step size. (Default: 0.2)
        steps (int): number of steps. (Default: 40)
        random_start (bool): using random initialization of delta. (Default: True)
        eps_for_division (float): added to the gradient norm before dividing. (Default: 1e-10)

    Examples::
        >>> attack = torchattacks.PGDL2(model, eps=1.0, alpha=0.2, steps=40, random_start=True)
        >>> adv_images = attack(images, labels)
    """

    replays_from_graph = True

    def __init__(self, model, eps=1.0, alpha=0.2, steps=40, random_start=True, eps_for_division=1e-10):
        super().__init__("PGDL2", model)
        self.eps = eps
        self.alpha = alpha
        self.steps = steps
        self.random_start = random_start
        self.eps_for_division = eps_for_division
        self._supported_mode = ["default", "targeted"]

    def forward(self, images, labels):
        ops = self.ops
        images, labels, target = self._prepare(images, labels)

        if self.random_start:
            # pgdl2.py:55-62: gaussian direction, radius r * eps with r ~ U(0, 1), clamp
            if self._init_noise is not None:
                normal, r = self._init_noise
                adv = ops.pgd_l2_init(images, self.eps, draws=(normal.to(self.device).contiguous(),
                                                               r.to(self.device).reshape(-1).contiguous()))
            else:
                adv = ops.pgd_l2_init(images, self.eps, seed=self._fresh_seed())
        else:
            adv = images.clone()

        # pgdl2.py:64-88, `steps` times: model forward + input-backward, then the fused step (normalise the gradient row-wise,
        # step, project onto the L2 ball, clamp), ping-pong; hipGraph replay as in PGD (graphed.py)
        def step(cur, grad, orig, out):
            ops.pgd_l2_step(cur, grad, orig, self.alpha, self.eps, self.eps_for_division, out=out)

        return graphed.run_iterations(self, adv, images, labels, target, self.steps, step,
                                      (self.eps, self.alpha, self.eps_for_division))
