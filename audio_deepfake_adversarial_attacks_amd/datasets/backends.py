"""Start-up registration of the optional third-party audio backends (reference: src/datasets/base_dataset.py:278-337 calls
`torchaudio.sox_effects.apply_effects_tensor` / `torchaudio.functional.apply_codec`; `torchaudio.load` decodes FLAC / MP3
through libsndfile / ffmpeg, :165).

Neither SoX nor the FLAC / MP3 codecs are restated by this build: when the libraries the reference
uses are importable they are plugged into the two plug-in points (`base_dataset.register_sox_backend`,
`audio_io.register_decoder`); when a run needs one that is missing, the CLIs stop at START-UP with one message that says
what is missing and which flag avoids it — not with an exception from a DataLoader worker minutes later."""
from __future__ import annotations

from typing import Dict, List, Optional

from . import audio_io, base_dataset


def register_available_backends() -> Dict[str, bool]:
    """Plug in whatever is importable; returns {"sox": bool, "codec": bool, "flac": bool, "mp3": bool}."""
    have = {"sox": base_dataset._sox_backend is not None, "codec": base_dataset._codec_backend is not None,
            "flac": ".flac" in audio_io._decoders, "mp3": ".mp3" in audio_io._decoders}
    if not have["sox"]:
        try:
            from torchaudio import sox_effects
            apply_codec = None
            try:
                from torchaudio.functional import apply_codec
            except ImportError:
                pass
            base_dataset.register_sox_backend(sox_effects.apply_effects_tensor, apply_codec)
            have["sox"], have["codec"] = True, apply_codec is not None
        except (ImportError, AttributeError, OSError):
            pass
    try:
        import soundfile

        def read(path):
            data, rate = soundfile.read(str(path), dtype="float32", always_2d=True)
            return data, rate

        formats = {f.upper() for f in soundfile.available_formats()}
        for ext, fmt in ((".flac", "FLAC"), (".mp3", "MP3")):
            if not have[ext[1:]] and fmt in formats:
                audio_io.register_decoder(ext, read)
                have[ext[1:]] = True
    except (ImportError, OSError):
        pass
    return have


def missing_for(asv_path: Optional[str], wavefake_path: Optional[str], celeb_path: Optional[str], trim: bool,
                have: Dict[str, bool]) -> List[str]:
    """What a run over these corpora needs and does not have (empty list = good to go)."""
    problems = []
    if asv_path is not None and not have["flac"]:
        problems.append("--asv_path: ASVspoof2021 files are FLAC and no FLAC decoder is available (install `soundfile` with "
                        "libsndfile, or call datasets.audio_io.register_decoder('.flac', fn))")
    if celeb_path is not None and not have["mp3"]:
        problems.append("--celeb_path: FakeAVCeleb audio is MP3 and no MP3 decoder is available (`soundfile` with "
                        "libsndfile >= 1.1, or datasets.audio_io.register_decoder('.mp3', fn))")
    real = any(p is not None for p in (asv_path, wavefake_path, celeb_path))
    if real and trim and not have["sox"]:
        problems.append("the reference preprocessing trims silence with SoX (base_dataset.py:29-33, 278-290) and no SoX backend "
                        "is available (install a `torchaudio` that ships sox_effects, or call "
                        "datasets.base_dataset.register_sox_backend(fn)); --no_trim skips the trim, which DEPARTS from the "
                        "reference preprocessing (the detector sees untrimmed audio) and still needs SoX's `rate` effect for "
                        "files that are not 16 kHz")
    return problems


def require_for(asv_path, wavefake_path, celeb_path, trim: bool) -> Dict[str, bool]:
    """CLI start-up check: register what is importable, then stop with ONE message if the run cannot work."""
    have = register_available_backends()
    problems = missing_for(asv_path, wavefake_path, celeb_path, trim, have)
    if problems:
        raise SystemExit("cannot run on the requested corpora with this installation:\n  - " + "\n  - ".join(problems) +
                         "\n(--synthetic N runs without any audio backend)")
    return have
