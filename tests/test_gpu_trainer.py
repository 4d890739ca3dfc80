"""`-m gpu`: adversarial training on the MI355X — LCNN + LFCC, attacks through the HIP kernels, one short run per
strategy family.  Checks the contract between the two consumers of the model: while an attack runs the parameters are
frozen (fused forward + input-backward kernels), the training step itself differentiates w.r.t. the parameters."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lcnn(cuda):
    from audio_deepfake_adversarial_attacks_amd.models.models import get_model
    torch.manual_seed(0)
    return get_model("lcnn", {"frontend_algorithm": ["lfcc"], "input_channels": 1}, str(cuda)).to(cuda)


@pytest.mark.parametrize("strategy,attacks", [("RANDOM", ["FGSM", "PGDL2", "FAB"]), ("EQUAL", ["PGD"]),
                                               ("ADAPTIVE_V2", ["FGSM_eps001", "PGDL2_eps20"])])
def test_adversarial_training_runs_on_lcnn(cuda, strategy, attacks):
    from audio_deepfake_adversarial_attacks_amd import trainer as T
    from audio_deepfake_adversarial_attacks_amd.aa.aa_trainer_types import AdversarialGDTrainerEnum
    from audio_deepfake_adversarial_attacks_amd.datasets.synthetic import SyntheticDetectionDataset
    random.seed(1)
    torch.manual_seed(1)
    model = _lcnn(cuda)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    tr = AdversarialGDTrainerEnum[strategy].value(epochs=1, batch_size=8, device=str(cuda), optimizer_kwargs={"lr": 1e-4})
    calls = []
    orig = tr._attack_batch
    tr._attack_batch = staticmethod(lambda atk, bx, by: (calls.append(type(atk).__name__), orig(atk, bx, by))[1])
    out = tr.train(dataset=SyntheticDetectionDataset(24, return_meta=False), model=model, attack_model=model, adversarial_attacks=attacks,
                   test_dataset=SyntheticDetectionDataset(8, seed=99, return_meta=False))
    assert out is model and calls                               # single process: no wrapper; attacks were applied
    assert all(p.requires_grad for p in model.parameters())       # the attacks' parameter freeze is always undone
    changed = [k for k, v in model.state_dict().items() if v.dtype.is_floating_point and not torch.equal(v, before[k])]
    assert len(changed) > 10 and all(torch.isfinite(v).all() for v in model.state_dict().values())
    if hasattr(tr, "adv_attacks_weights"):
        assert abs(sum(float(w) for w in tr.adv_attacks_weights) - 1.0) < 1e-5


def test_training_cli_end_to_end(cuda, tmp_path):
    import yaml

    import train_models_on_adversarial_attacks as cli
    args = cli.parse_args(["--config", "configs/aa_training/finetune/lcnn_fgsm.yaml", "--synthetic", "16,8", "--batch_size", "8",
                           "--epochs", "1", "--ckpt", str(tmp_path / "ckpt"), "--config_save_path", str(tmp_path),
                           "--adv_training_strategy", "ONLY_ADV"])
    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    cfg["data"]["adversarial_attacks"] = ["FGSM_eps001"]
    one = tmp_path / "one.yaml"
    one.write_text(yaml.dump(cfg))
    args.config = str(one)
    cli.main(args)
    saved = list((tmp_path / "ckpt").glob("aad__lcnn_*/ckpt*.pth"))
    assert len(saved) == 2                                         # per-epoch checkpoint + final (reference layout)
    written = [p for p in tmp_path.glob("aad__lcnn__*.yaml")]
    assert len(written) == 1
    test_cfg = yaml.safe_load(written[0].read_text())
    # the written config evaluates straight away with the evaluation CLI's loader
    from audio_deepfake_adversarial_attacks_amd.utils import load_model
    m = load_model(test_cfg, str(cuda))
    assert m.weights_path.endswith("ckpt.pth")
