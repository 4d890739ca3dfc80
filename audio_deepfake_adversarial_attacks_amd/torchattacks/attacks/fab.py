"""FAB (reference: adversarial_attacks/torchattacks/attacks/fab.py:19-559, projections :562-717).

Same constructor, attributes and public methods as the reference class; the per-iteration tensor work runs in the
FAB kernels of libadvstep.so (include/advstep_fab.h):

    reference, per iteration (fab.py:208-292)                        here
    ------------------------------------------------------------     ---------------------------------------------
    model forward + ONE BACKWARD PER LOGIT COLUMN of cat([-z, z])    one forward + one backward (the columns'
      (get_diff_logits_grads_batch, :90-112)                           gradients are -gz and +gz exactly)
    df / dg / dist / argmin / b  on (B, 2, T) tensors  (:210-229)     ops.fab_hyperplane   (one pass over gz, x1)
    projection_linf|l2|l1 on cat((x1, x0)), cat((w, w))  (:231-245)   ops.fab_projection   (no sort, no cat of w)
    a0 / alpha / convex combination / clamp  (:246-267)                ops.fab_combine      (in place)
    prediction, `if is_adv.sum() > 0` host sync, norms, masked        ops.fab_backward_step (flags stay on the
      blends, backward step  (:269-290)                                device: no host sync inside the loop)

Deviations, all documented in INTEGRATION.md: floating-point (not bit) parity of the projections; `norm='L1'` works
through `forward` (the reference raises UnboundLocalError at fab.py:522 because `res` is never assigned for L1)."""
import time

import torch

from ..attack import Attack

_DEFAULT_EPS = {"Linf": 0.3, "L2": 1.0, "L1": 5.0}


def _row_norm(v, norm):
    v = v.reshape(v.shape[0], -1)
    if norm == "Linf":
        return v.abs().max(dim=1)[0]
    if norm == "L2":
        return (v ** 2).sum(dim=-1).sqrt()
    if norm == "L1":
        return v.abs().sum(dim=-1)
    raise ValueError("norm not supported")


class FAB(Attack):
    r"""Fast Adaptive Boundary attack, 'Minimally distorted Adversarial Examples with a Fast Adaptive Boundary Attack'
    [https://arxiv.org/abs/1907.02044].  Distance measure: Linf, L2, L1.

    Arguments:
        model (nn.Module): model to attack.
        norm (str): Lp-norm to minimize. ['Linf', 'L2', 'L1'] (Default: 'Linf')
        eps (float): maximum perturbation. (Default: None -> 0.3 / 1.0 / 5.0 by norm)
        steps (int): number of steps. (Default: 100)
        n_restarts (int): number of random restarts. (Default: 1)
        alpha_max (float): alpha_max. (Default: 0.1)
        eta (float): overshooting. (Default: 1.05)
        beta (float): backward step. (Default: 0.9)
        verbose (bool): print progress. (Default: False)
        seed (int): random seed for the starting point. (Default: 0)
        targeted (bool): accepted and ignored exactly like the reference (fab.py:63 sets `self.targeted = False`).
        n_classes (int): number of classes. (Default: 10)

    Examples::
        >>> attack = torchattacks.FAB(model, norm='Linf', steps=100, eps=None, n_restarts=1, alpha_max=0.1, eta=1.05,
        ...                           beta=0.9, verbose=False, seed=0, targeted=False, n_classes=10)
        >>> adv_images = attack(images, labels)
    """

    def __init__(self, model, norm="Linf", eps=None, steps=100, n_restarts=1, alpha_max=0.1, eta=1.05, beta=0.9,
                 verbose=False, seed=0, targeted=False, n_classes=10):
        super().__init__("FAB", model)
        self.norm = norm
        self.n_restarts = n_restarts
        self.eps = eps if eps is not None else _DEFAULT_EPS[norm]
        self.alpha_max = alpha_max
        self.eta = eta
        self.beta = beta
        self.steps = steps
        self.targeted = False
        self.verbose = verbose
        self.seed = seed
        self.target_class = None
        self.n_target_classes = n_classes - 1
        self._supported_mode = ["default"]

    def forward(self, images, labels):
        images = images.clone().detach().to(self.device)
        labels = labels.clone().detach().to(self.device)
        return self.perturb(images, labels)

    # ---- model access ----------------------------------------------------------------------------------------------

    def _logits2(self, x):
        out = self.model(x)
        return torch.cat([-out, out], dim=1)

    def _get_predicted_label(self, x):
        with torch.no_grad():
            return self._logits2(x).max(dim=1)[1]

    def check_shape(self, x):
        return x if len(x.shape) > 0 else x.unsqueeze(0)

    def _logit_and_gradient(self, x1):
        """z (B) and d(sum z)/dx (B, T): one forward + one input-backward."""
        im = x1.detach().requires_grad_()
        with torch.enable_grad():
            z = self.model(im)
            (gz,) = torch.autograd.grad(z, im, grad_outputs=torch.ones_like(z))
        return z.detach().reshape(-1).contiguous(), gz.contiguous()

    def get_diff_logits_grads_batch(self, imgs, la):
        """fab.py:90-112 with the reference's shapes: df (B, 2), dg (B, 2, *imgs.shape[1:]).  Kept for callers of the
        public method; the attack loop itself never materialises dg."""
        z, gz = self._logit_and_gradient(imgs)
        y2 = torch.stack([-z, z], dim=1)
        g2 = torch.stack([-gz, gz], dim=1)
        u = torch.arange(imgs.shape[0])
        df = y2 - y2[u, la].unsqueeze(1)
        dg = g2 - g2[u, la].unsqueeze(1)
        df[u, la] = 1e10
        return df, dg

    def get_diff_logits_grads_batch_targeted(self, imgs, la, la_target):
        """fab.py:114-129: df (B, 1), dg (B, 1, ...) of the single difference -(y[la] - y[la_target])."""
        z, gz = self._logit_and_gradient(imgs)
        y2 = torch.stack([-z, z], dim=1)
        col = torch.tensor([-1.0, 1.0], device=z.device)
        u = torch.arange(imgs.shape[0])
        df = -(y2[u, la] - y2[u, la_target])
        coef = col[la_target] - col[la]
        return df.unsqueeze(1), (coef.reshape(-1, *[1] * (gz.dim() - 1)) * gz).unsqueeze(1)

    # ---- one run -------------------------------------------------------------------------------------------------------

    def _random_start(self, x0, res2):
        """fab.py:174-205.  The draw is made on the CPU generator and moved, exactly as the reference's
        `torch.rand(x1.shape).to(self.device)`, so a seeded run starts from the reference's points."""
        radius = torch.min(res2, self.eps * torch.ones_like(res2)).reshape(-1, 1)
        if self.norm == "Linf":
            t = 2 * torch.rand(x0.shape).to(self.device) - 1
            x1 = x0 + radius * t / t.abs().max(dim=1, keepdim=True)[0] * 0.5
        elif self.norm == "L2":
            t = torch.randn(x0.shape).to(self.device)
            x1 = x0 + radius * t / (t ** 2).sum(dim=-1, keepdim=True).sqrt() * 0.5
        elif self.norm == "L1":
            t = torch.randn(x0.shape).to(self.device)
            x1 = x0 + radius * t / t.abs().sum(dim=-1, keepdim=True) / 2
        else:
            raise ValueError("norm not supported")
        return x1.clamp(0.0, 1.0)

    def _single_run(self, x, y, use_rand_start, targeted):
        ops = self.ops
        if self.norm not in _DEFAULT_EPS:
            raise ValueError("norm not supported")
        self.orig_dim = list(x.shape[1:])
        self.ndims = len(self.orig_dim)

        x = x.detach().clone().float().to(self.device)
        y_pred = self._get_predicted_label(x)
        y = y_pred.detach().clone().long() if y is None else y.detach().clone().long().to(self.device)
        pred = y_pred == y
        corr_classified = pred.float().sum()
        if self.verbose:
            print("Clean accuracy: {:.2%}".format(pred.float().mean()))
        if pred.sum() == 0:
            return x
        rows = self.check_shape(pred.nonzero().squeeze())
        la_target = None
        if targeted:
            la_target = self._logits2(x).sort(dim=-1)[1][:, -self.target_class][rows].detach().clone()

        startt = time.time()
        shape = x.shape
        flat = x.reshape(shape[0], -1)
        bs, T = rows.numel(), flat.shape[1]
        # points buffer of the projection: rows [0, bs) = x1 (updated in place), rows [bs, 2 bs) = the clean points
        pts = torch.empty(2 * bs, T, device=self.device)
        x1, x0 = pts[:bs], pts[bs:]
        x0.copy_(flat[rows])
        la = y[rows].contiguous()
        adv = x0.clone()
        res2 = torch.full((bs,), 1e10, device=self.device)
        x1.copy_(self._random_start(x0, res2) if use_rand_start else x0)
        d3 = torch.empty_like(pts)
        col = torch.tensor([-1.0, 1.0], device=self.device)

        for _ in range(self.steps):
            z, gz = self._logit_and_gradient(x1.view(bs, *shape[1:]))
            gz = gz.reshape(bs, T)
            if targeted:
                _, _, _, gdot = ops.fab_hyperplane(gz, x1, None, None, self.norm)
                wscale = (col[la_target] - col[la]).contiguous()
                y2 = torch.stack([-z, z], dim=1)
                u = torch.arange(bs, device=self.device)
                b = (y2[u, la] - y2[u, la_target]) + wscale * gdot
            else:
                wscale, b, _, _ = ops.fab_hyperplane(gz, x1, z, la, self.norm)
            _, n3 = ops.fab_projection(pts, gz, b.repeat(2), self.norm, wscale, out=d3)
            ops.fab_combine(x1, x0, d3[:bs], d3[bs:], n3[:bs], n3[bs:], self.eta, self.alpha_max, out=x1)
            is_adv = self._get_predicted_label(x1.view(bs, *shape[1:])) != la
            ops.fab_backward_step(x1, x0, adv, res2, is_adv, self.beta, self.norm)

        ind_succ = res2 < 1e10
        if self.verbose:
            print("success rate: {:.0f}/{:.0f}".format(ind_succ.float().sum(), corr_classified)
                  + " (on correctly classified points) in {:.1f} s".format(time.time() - startt))
        adv_c = flat.clone()
        ind_succ = self.check_shape(ind_succ.nonzero().squeeze())
        adv_c[rows[ind_succ]] = adv[ind_succ]
        return adv_c.view(shape)

    def attack_single_run(self, x, y=None, use_rand_start=False):
        """:param x: clean images  :param y: clean labels, if None we use the predicted labels  (fab.py:131-307)"""
        with torch.no_grad():
            return self._single_run(x, y, use_rand_start, targeted=False)

    def attack_single_run_targeted(self, x, y=None, use_rand_start=False):
        """fab.py:309-493; needs `self.target_class` (set by `perturb` when `self.targeted`)."""
        with torch.no_grad():
            return self._single_run(x, y, use_rand_start, targeted=True)

    # ---- restarts --------------------------------------------------------------------------------------------------------

    def perturb(self, x, y):
        """fab.py:495-559."""
        adv = x.clone()
        with torch.no_grad():
            acc = self._logits2(x).max(1)[1] == y
            startt = time.time()

            torch.random.manual_seed(self.seed)
            torch.cuda.random.manual_seed(self.seed)

            targets = range(2, self.n_target_classes + 2) if self.targeted else (None,)
            for target_class in targets:
                self.target_class = target_class
                for counter in range(self.n_restarts):
                    ind_to_fool = self.check_shape(acc.nonzero().squeeze())
                    if ind_to_fool.numel() == 0:
                        continue
                    x_to_fool, y_to_fool = x[ind_to_fool].clone(), y[ind_to_fool].clone()
                    run = self.attack_single_run_targeted if self.targeted else self.attack_single_run
                    adv_curr = run(x_to_fool, y_to_fool, use_rand_start=(counter > 0))

                    acc_curr = self._logits2(adv_curr).max(1)[1] == y_to_fool
                    res = _row_norm(x_to_fool - adv_curr, self.norm)
                    acc_curr = torch.max(acc_curr, res > self.eps)

                    ind_curr = (acc_curr == 0).nonzero().squeeze()
                    acc[ind_to_fool[ind_curr]] = 0
                    adv[ind_to_fool[ind_curr]] = adv_curr[ind_curr].clone()

                    if self.verbose:
                        print("restart {} - robust accuracy: {:.2%} at eps = {:.5f} - cum. time: {:.1f} s".format(
                            counter, acc.float().mean(), self.eps, time.time() - startt))
        return adv
