"""Attack plugin API of the hot path (reference: adversarial_attacks/torchattacks/__init__.py).

The attacks the north-star path names — FGSM, PGD, PGDL2, CW — and FAB (the attack that completes the
reference's AttackEnum), with the reference's constructor signatures and the reference's 1-logit -> 2-logit adapter
(`cat([-z, z], 1)`)."""
from .attack import Attack
from .attacks.cw import CW
from .attacks.fab import FAB
from .attacks.fgsm import FGSM
from .attacks.pgd import PGD
from .attacks.pgdl2 import PGDL2

__version__ = "3.2.7+advstep"
__all__ = ["Attack", "FGSM", "PGD", "PGDL2", "CW", "FAB"]
