"""Attack registry: CLI name -> (attack class, kwargs)  (reference: src/aa/aa_types.py:5-24).

The reference's members are kept verbatim, FAB included.
Additive members carry the configurations BASELINE.json names, which the reference's enum cannot express
(SURVEY.md F7): 40-step PGD at eps = 0.003, 40-step PGDL2, and CW."""
from enum import Enum

from .. import torchattacks


class AttackEnum(Enum):

    # --- reference members (aa_types.py:8-22) ---
    PGD = (torchattacks.PGD, {"eps": 0.0005, "steps": 10})
    PGD_eps00075 = (torchattacks.PGD, {"eps": 0.00075, "steps": 10})
    PGD_eps001 = (torchattacks.PGD, {"eps": 0.001, "steps": 10})

    PGDL2 = (torchattacks.PGDL2, {"eps": 0.1, "steps": 10})
    PGDL2_eps15 = (torchattacks.PGDL2, {"eps": 0.15, "steps": 10})
    PGDL2_eps20 = (torchattacks.PGDL2, {"eps": 0.20, "steps": 10})

    FGSM = (torchattacks.FGSM, {"eps": 0.0005})
    FGSM_eps00075 = (torchattacks.FGSM, {"eps": 0.00075})
    FGSM_eps001 = (torchattacks.FGSM, {"eps": 0.001})

    FAB = (torchattacks.FAB, {"n_classes": 2, "eta": 10})
    FAB_eta20 = (torchattacks.FAB, {"n_classes": 2, "eta": 20})
    FAB_eta30 = (torchattacks.FAB, {"n_classes": 2, "eta": 30})

    # --- additive members for BASELINE.json's configurations ---
    PGD40_eps003 = (torchattacks.PGD, {"eps": 0.003, "steps": 40})          # configs 2 and 5 (alpha default 2/255)
    PGDL2_40 = (torchattacks.PGDL2, {"eps": 0.1, "steps": 40})             # config 3 (alpha default 0.2)
    CW = (torchattacks.CW, {"c": 1.0, "kappa": 0, "steps": 100, "lr": 0.01})  # config 4 (cw.py:27 advises c ~ 1)

    NO_ATTACK = (None, {})
