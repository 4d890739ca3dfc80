"""Qualitative results of an attack run: the utterances whose verdict the attack flipped, as WAV pairs
(reference: src/aa/qualitative/attacks_analysis.py).

`AttackAnalyser(result_dst).analyse(...)` is the `on_attack_end_callback` of `generate_attacks`
(evaluate_models_on_adversarial_attacks.py:126-129, 249-258) and keeps the reference's keyword contract and output:
for every utterance that the target model classified correctly before the attack and differently after it,
`<name>_<subset>_<seconds>sec_{fp|fn}_original.wav` and `..._attacked.wav`, 16 kHz IEEE-float WAVE files
(`fp`: a spoof now accepted, `fn`: a bonafide utterance now rejected), plus the per-utterance diff line on stdout.

The reference copies both (B, 64600) batches and all predictions to the host and selects there
(`tensor_to_ndarray`, :17-49).  Here the selection runs on the device (`advstep_qual_select`), the flipped rows of both
batches are packed by `advstep_wave_gather_rows_f32`, and only those rows — usually a handful — cross PCIe, into a
pinned staging buffer that the WAVE writer streams to disk.  The per-row mean |x - x_adv| of the diff line is one
device reduction over the batch ((B) floats to the host)."""
from __future__ import annotations

import logging
from pathlib import Path

import numpy as np
import torch

from ...datasets import wave_ops
from ...datasets.audio_io import write_wav_f32

LOGGER = logging.getLogger()

SAMPLE_RATE = 16_000  # the rate the reference writes (:129,:135), whatever the source file had


def result_file_stem(src_path, subset, sec_length) -> str:
    """`<folder>_<stem>` for WaveFake / FakeAVCeleb sources (their stems repeat across folders), `<stem>` otherwise,
    then `_<subset>_<seconds:.2f>sec` (reference :114-122)."""
    src_path = Path(src_path)
    name = src_path.stem
    if "WaveFake" in str(src_path) or "FakeAVCeleb" in str(src_path):
        name = f"{src_path.parent.name}_{name}"
    return f"{name}_{subset}_{float(sec_length):.2f}sec"


class AttackAnalyser:
    def __init__(self, result_dst):
        self.result_dst = Path(result_dst)
        self.result_dst.mkdir(parents=True, exist_ok=True)

    @staticmethod
    def batch_metadata_rows(batch_metadata):
        """default_collate turns B metadata tuples into 4 columns; back to one (attack, path, subset, seconds) per row."""
        return [tuple(v.item() if isinstance(v, torch.Tensor) else v for v in row) for row in zip(*batch_metadata)]

    def analyse(self, batch_x, batch_x_attacked, batch_y, batch_preds_label, batch_preds, batch_preds_noattack_label,
                batch_preds_noattack, batch_metadata):
        rows_meta = self.batch_metadata_rows(batch_metadata)
        y = batch_y.to(torch.int64).contiguous()
        clean = batch_preds_noattack_label.to(torch.int32).contiguous()
        attacked = batch_preds_label.to(torch.int32).contiguous()

        rows, counts = wave_ops.qual_select(y, clean, attacked)
        mean_abs = (batch_x - batch_x_attacked).abs().mean(dim=1)
        counts_h = counts.cpu()                      # one small sync per batch: how many rows to fetch
        n_fp, n_fn = int(counts_h[0]), int(counts_h[1])
        n = n_fp + n_fn

        self.sample_diffs(mean_abs.cpu().numpy(), y.cpu().numpy(), clean.cpu().numpy(), attacked.cpu().numpy(),
                          rows_meta)
        rows_h = rows[:n].cpu().numpy()
        LOGGER.info("false_positives: {}".format(rows_h[:n_fp]))
        LOGGER.info("false_negatives: {}".format(rows_h[n_fp:]))
        if n == 0:
            return
        T = batch_x.shape[1]
        staging = torch.empty(2, n, T, pin_memory=True)
        staging[0].copy_(wave_ops.gather_rows(batch_x.contiguous(), rows, n), non_blocking=True)
        staging[1].copy_(wave_ops.gather_rows(batch_x_attacked.contiguous(), rows, n), non_blocking=True)
        torch.cuda.current_stream(batch_x.device).synchronize()
        original, adversarial = staging[0].numpy(), staging[1].numpy()
        self.save_waves(rows_h[:n_fp], original[:n_fp], adversarial[:n_fp], rows_meta, "fp")
        self.save_waves(rows_h[n_fp:], original[n_fp:], adversarial[n_fp:], rows_meta, "fn")

    @staticmethod
    def sample_diffs(mean_abs, batch_y, batch_preds_noattack_label, batch_preds_label, rows_meta):
        """The reference's stdout line per utterance (:51-68)."""
        import torch.distributed as dist
        # multi-rank runs: every line names its rank (ranks print concurrently; the reference has one process)
        prefix = (f"[rank {dist.get_rank()}]",) if dist.is_available() and dist.is_initialized() else ()
        for i in range(len(batch_y)):
            print(*prefix, i, mean_abs[i], batch_preds_noattack_label[i] != batch_preds_label[i], "y:", batch_y[i],
                  "y_noadvatk_pred:", batch_preds_noattack_label[i], "y_pred:", batch_preds_label[i], *rows_meta[i])

    def save_waves(self, batch_rows, waves, waves_attacked, rows_meta, suffix):
        """waves[k] / waves_attacked[k]: the packed host copies of batch row batch_rows[k]."""
        for k, i in enumerate(batch_rows):
            stem = result_file_stem(rows_meta[i][1], rows_meta[i][2], rows_meta[i][3])
            write_wav_f32(self.result_dst / f"{stem}_{suffix}_original.wav", SAMPLE_RATE, waves[k])
            write_wav_f32(self.result_dst / f"{stem}_{suffix}_attacked.wav", SAMPLE_RATE, waves_attacked[k])
