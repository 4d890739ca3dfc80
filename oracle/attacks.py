"""TEST INFRASTRUCTURE — whole-attack CPU restatement of the reference, one torch op per reference op.

Functional restatements of FGSM / PGD / PGDL2 / CW and of the per-batch body of the evaluation loop, written
against plain torch CPU ops in the reference's order, so on the same torch build and thread count they are
BIT-IDENTICAL to the reference classes (asserted against tests/golden/*.npz in tests/test_oracle_golden.py).
They serve as (i) the checker for the product's Attack classes and (ii) the multi-threaded `cpu_baseline` of
bench.py ("port": the reference itself cannot travel to the GPU box).

Reference: adversarial_attacks/torchattacks/attack.py:308-331 (mode juggling), attacks/fgsm.py:33-62,
attacks/pgd.py:40-78, attacks/pgdl2.py:40-90, attacks/cw.py:46-134, src/aa/utils.py:4-14,
evaluate_models_on_adversarial_attacks.py:211-265."""
from contextlib import contextmanager

import torch
import torch.nn as nn


@contextmanager
def attack_mode(model, model_training=True, batchnorm_training=False, dropout_training=False):
    """attack.py:308-326 with the flags evaluate_models_on_adversarial_attacks.py:170 sets."""
    was_training = model.training
    if model_training:
        model.train()
        for _, m in model.named_modules():
            name = m.__class__.__name__
            if (not batchnorm_training and "BatchNorm" in name) or (not dropout_training and "Dropout" in name):
                m.eval()
    else:
        model.eval()
    try:
        yield
    finally:
        if was_training:
            model.train()


def _cost_and_grad(model, adv, labels, with_cost=False):
    """pgd.py:60-72: logits -> cat([-z, z]) -> CrossEntropyLoss -> gradient w.r.t. the input.
    with_cost (tests only): also return the scalar loss and the logits the reference computes on the way."""
    adv.requires_grad = True
    z = model(adv)
    outputs = torch.cat([-z, z], dim=1)
    cost = nn.CrossEntropyLoss()(outputs, labels)
    grad = torch.autograd.grad(cost, adv, retain_graph=False, create_graph=False)[0]
    return (grad, cost.detach(), z.detach()) if with_cost else grad


def to_minmax(batch_x):
    mn, _ = torch.min(batch_x, dim=1, keepdim=True)
    mx, _ = torch.max(batch_x, dim=1, keepdim=True)
    r = mx - mn
    return (batch_x - mn) / r, mn, mx


def revert_minmax(batch_x, mn, mx):
    return (batch_x * (mx - mn)) + mn


def fgsm(model, images, labels, eps=0.007):
    images = images.clone().detach()
    labels = labels.clone().detach()
    grad = _cost_and_grad(model, images, labels)
    return torch.clamp(images + eps * grad.sign(), min=0, max=1).detach()


def pgd(model, images, labels, eps=0.3, alpha=2 / 255, steps=40, random_start=True, noise=None, trace=None):
    images = images.clone().detach()
    labels = labels.clone().detach()
    adv = images.clone().detach()
    if random_start:
        if noise is None:
            noise = torch.empty_like(adv).uniform_(-eps, eps)
        adv = torch.clamp(adv + noise, min=0, max=1).detach()
    for _ in range(steps):
        grad, cost, z = _cost_and_grad(model, adv, labels, with_cost=True)
        if trace is not None:
            trace.append((adv.detach().clone(), grad.clone(), cost.clone(), z.clone()))
        adv = adv.detach() + alpha * grad.sign()
        delta = torch.clamp(adv - images, min=-eps, max=eps)
        adv = torch.clamp(images + delta, min=0, max=1).detach()
    return adv


def pgdl2(model, images, labels, eps=1.0, alpha=0.2, steps=40, random_start=True, eps_for_division=1e-10,
          draws=None, trace=None):
    images = images.clone().detach()
    labels = labels.clone().detach()
    adv = images.clone().detach()
    batch_size = len(images)
    if random_start:
        if draws is None:
            delta = torch.empty_like(adv).normal_()
            n = delta.view(batch_size, -1).norm(p=2, dim=1).view(batch_size, 1)
            r = torch.zeros_like(n).uniform_(0, 1)
        else:
            delta = draws[0].clone()
            n = delta.view(batch_size, -1).norm(p=2, dim=1).view(batch_size, 1)
            r = draws[1].view(batch_size, 1)
        delta *= r / n * eps
        adv = torch.clamp(adv + delta, min=0, max=1).detach()
    for _ in range(steps):
        grad, cost, z = _cost_and_grad(model, adv, labels, with_cost=True)
        if trace is not None:
            trace.append((adv.detach().clone(), grad.clone(), cost.clone(), z.clone()))
        grad_norms = torch.norm(grad.view(batch_size, -1), p=2, dim=1) + eps_for_division
        grad = grad / grad_norms.view(batch_size, 1)
        adv = adv.detach() + alpha * grad
        delta = adv - images
        delta_norms = torch.norm(delta.view(batch_size, -1), p=2, dim=1)
        factor = eps / delta_norms
        factor = torch.min(factor, torch.ones_like(delta_norms))
        delta = delta * factor.view(-1, 1)
        adv = torch.clamp(images + delta, min=0, max=1).detach()
    return adv


def _cw_f(outputs, labels, kappa):
    one_hot = torch.eye(len(outputs[0]))[labels].to(outputs.device)
    i, _ = torch.max((1 - one_hot) * outputs, dim=1)
    j = torch.masked_select(outputs, one_hot.bool())
    return torch.clamp((j - i), min=-kappa)


def cw(model, images, labels, c=1e-4, kappa=0, steps=1000, lr=0.01, trace=None):
    images = images.clone().detach()
    labels = labels.clone().detach()
    y = images * 2 - 1
    w = (0.5 * torch.log((1 + y) / (1 - y))).detach()
    w.requires_grad = True
    best_adv = images.clone().detach()
    best_l2 = 1e10 * torch.ones(len(images))
    prev_cost = 1e10
    dim = len(images.shape)
    mse = nn.MSELoss(reduction="none")
    flat = nn.Flatten()
    optimizer = torch.optim.Adam([w], lr=lr)
    for step in range(steps):
        adv = 1 / 2 * (torch.tanh(w) + 1)
        current_l2 = mse(flat(adv), flat(images)).sum(dim=1)
        l2_loss = current_l2.sum()
        outputs = model(adv)
        outputs = torch.cat([-outputs, outputs], dim=1)
        f_loss = _cw_f(outputs, labels, kappa).sum()
        cost = l2_loss + c * f_loss
        optimizer.zero_grad()
        cost.backward()
        optimizer.step()
        _, pre = torch.max(outputs.detach(), 1)
        correct = (pre == labels).float()
        mask = (1 - correct) * (best_l2 > current_l2.detach())
        best_l2 = mask * current_l2.detach() + (1 - mask) * best_l2
        mask = mask.view([-1] + [1] * (dim - 1))
        best_adv = mask * adv.detach() + (1 - mask) * best_adv
        if trace is not None:
            trace.append((cost.detach().clone(), current_l2.detach().clone(), outputs.detach()[:, 1].clone(),
                          adv.detach().clone()))
        if step % max(steps // 10, 1) == 0:
            if cost.item() > prev_cost:
                return best_adv
            prev_cost = cost.item()
    return best_adv


ATTACKS = {"FGSM": fgsm, "PGD": pgd, "PGDL2": pgdl2, "CW": cw}


def attack_and_score(target_model, attack_model, attack_name, attack_params, batch_x, batch_y):
    """One pass of the hot-loop body (evaluate_models_on_adversarial_attacks.py:212-238) on CPU tensors."""
    target_model.eval()
    x01, mn, mx = to_minmax(batch_x)
    with attack_mode(attack_model):
        adv01 = ATTACKS[attack_name](attack_model, x01, batch_y, **attack_params)
    adv = revert_minmax(adv01, mn, mx)
    with torch.no_grad():
        preds = torch.sigmoid(target_model(adv).squeeze(1).detach())
        labels = (preds + 0.5).int()
    return adv, preds, labels
