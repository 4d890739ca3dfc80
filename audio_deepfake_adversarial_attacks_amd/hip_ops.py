"""Torch-facing wrappers of the C ABI (include/advstep.h): device tensors in, raw pointers + HIP stream out.

PyTorch is used for plumbing only — device memory, the current HIP stream, autograd around the model.  Every
function requires contiguous float32 tensors that live on a HIP device and launches on torch's *current*
stream of that device, so the kernels are ordered with the model's forward/backward work without any
synchronisation.  There is NO CPU path here: a CPU tensor, a missing libadvstep.so or a non-zero status
raises.

Reference op chains replaced (see the header for line-by-line citations):
  to_minmax / revert_minmax          src/aa/utils.py:4-14
  fgsm_step                          adversarial_attacks/torchattacks/attacks/fgsm.py:59-60
  pgd_linf_init / pgd_linf_step      .../attacks/pgd.py:54-57, 74-76
  pgd_l2_init / pgd_l2_step          .../attacks/pgdl2.py:55-62, 78-88
  cw_*                               .../attacks/cw.py:57, 72-77, 87-103
  ce2_loss_grad                      .../attacks/pgd.py:62,50,68 (and the same lines of fgsm.py / pgdl2.py)
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import _lib

NAME = "hip"

# ---------------------------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------------------------

_workspaces: Dict[Tuple[int, int, int, int], torch.Tensor] = {}     # (device, stream handle, B, T) -> row-reduction scratch
_profile: Optional[Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event]]]] = None
_profile_work: Dict[str, List[float]] = {}      # per bracketed launch: the arithmetic the caller says it did (flop), or 0
_profile_bytes: Dict[str, List[float]] = {}     # per bracketed launch: the bytes of the operands the caller named, or 0
_profile_all = False                            # start_profile("*"): every entry point
_profile_graph_ok = False                       # start_profile(..., graph_ok=True): only launches OUTSIDE captured graphs matter
_launch_depth = 0                               # > 0 while a bracketed launch is being issued (bench.py's library-op brackets skip what is inside)


def _require(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _lib.AdvstepError(
            f"{name}: tensor lives on '{t.device}'. The attack kernels run only on a HIP device "
            "(there is no CPU fallback in this package; the CPU restatement is the test-only oracle/ tree).")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _same_shape(*named) -> None:
    first_name, first = named[0]
    for name, t in named[1:]:
        if t.shape != first.shape or t.device != first.device:
            raise ValueError(f"{name} {tuple(t.shape)}@{t.device} does not match {first_name} "
                             f"{tuple(first.shape)}@{first.device}")


def _rows(t: torch.Tensor, name: str) -> Tuple[int, int]:
    if t.dim() < 2:
        raise ValueError(f"{name}: expected (B, ...) with at least 2 dims, got {tuple(t.shape)}")
    B = t.shape[0]
    return B, (t.numel() // B if B else 0)


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


_WORKSPACE_SLOTS = 32      # bound of the scratch table: distinct (device, stream, B, T) kept alive at once


def _workspace(device: torch.device, B: int, T: int) -> Tuple[int, int]:
    # one scratch per (device, stream, batch shape): the layout of the single-pass PGD-L2 exchange area depends on (B, T), and a
    # buffer that only ever sees one shape and one stream never shows a call words another layout or another stream's call left
    # behind (include/advstep.h).  The table is a small LRU: variable-length clips or short last batches
    # would otherwise add a zero-filled buffer per distinct shape for the life of the process.  Evicting is safe: the tensor
    # was allocated and only ever used on the stream of its key, so the caching allocator re-issues its memory in stream order;
    # a captured graph OWNS the buffers it baked in (`release_stream_workspaces`), they are not in this table any more.
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(device), int(B), int(T))
    ws = _workspaces.pop(key, None)
    if ws is None:
        need = _lib.load().advstep_row_workspace_bytes(B, T)
        # zero-filled ONCE: epoch 0, no row flagged (include/advstep.h)
        ws = torch.zeros(max(need, 256), dtype=torch.uint8, device=device)
        while len(_workspaces) >= _WORKSPACE_SLOTS:
            _workspaces.pop(next(iter(_workspaces)))                 # least recently used first (dicts keep insertion order)
    _workspaces[key] = ws                                            # (re-)inserted last = most recently used
    return ws.data_ptr(), ws.numel()


def release_stream_workspaces(stream_handle: int) -> list:
    """Hand over — and forget — every scratch buffer keyed by `stream_handle`.  A captured graph calls this right after its
    capture: the graph replays kernels that write to these ADDRESSES, so it must own the buffers, and an eager call on a stream
    that torch later gives the same handle (its pool is round-robin) must NOT get the same buffer — two streams sharing one
    workspace is outside the C ABI's contract (the exchange's call counter is advanced by a plain read-modify-write)."""
    keys = [k for k in _workspaces if k[1] == stream_handle]
    return [_workspaces.pop(k) for k in keys]


def _out_like(ref: torch.Tensor, out: Optional[torch.Tensor], name: str = "out") -> torch.Tensor:
    if out is None:
        return torch.empty_like(ref, memory_format=torch.contiguous_format)
    _require(out, name)
    _same_shape(("input", ref), (name, out))
    return out


class _Launch:
    """Device guard + optional HIP-event bracket around one C-ABI call (events sit on the launch stream).  `work` = the
    floating-point operations the launch must do — for the Winograd F(2x2, 3x3) kernels what goes through the matrix cores: 2 x 16
    products per 2x2 output tile, output row and reduction channel = a direct 3x3 convolution's count / 2.25, unpadded;
    `tensors` = the operands the launch reads or writes once (its algorithmic HBM bytes = the sum of their sizes).  Both are
    kept next to the bracket's duration for bench.py's rooflines (`roofline_model`, `roofline_step`)."""

    def __init__(self, name: str, device: torch.device, work: float = 0.0, tensors=()):
        self.name, self.device, self.work = name, device, work
        self.nbytes = float(sum(t.numel() * t.element_size() for t in tensors if t is not None)) if tensors else 0.0
        self.guard = torch.cuda.device(device)

    def __enter__(self):
        global _launch_depth
        _launch_depth += 1
        self.guard.__enter__()
        self.pair = None
        if (_profile is not None and (_profile_all or self.name in _profile)
                and not torch.cuda.is_current_stream_capturing()):       # an event cannot be recorded into a graph
            self.pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.pair[0].record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, *exc):
        global _launch_depth
        _launch_depth -= 1
        if self.pair is not None:
            self.pair[1].record(torch.cuda.current_stream(self.device))
            _profile.setdefault(self.name, []).append(self.pair)
            _profile_work.setdefault(self.name, []).append(self.work)
            _profile_bytes.setdefault(self.name, []).append(self.nbytes)
        return self.guard.__exit__(*exc)


def start_profile(*entry_points: str, graph_ok: bool = False) -> None:
    """Bracket every later launch of the named entry points ("*": of every entry point) with HIP events on their launch
    stream.  graph_ok: the caller prices launches that stay OUTSIDE the attacks' captured graphs (the update steps,
    min-max, random starts) and the iteration may keep replaying its model part from a hipGraph (torchattacks/graphed.py);
    without it profiling keeps the whole loop eager so that launches of the model part can be bracketed too."""
    global _profile, _profile_all, _profile_graph_ok
    _profile_graph_ok = bool(graph_ok)
    _profile_all = "*" in entry_points
    _profile = {n: [] for n in entry_points if n != "*"}
    _profile_work.clear()
    _profile_bytes.clear()


def stop_profile(with_work=False):
    """Synchronise and return {entry point: [milliseconds per launch]}; with_work: also {entry point: [flop per launch]};
    with_work="bytes": also {entry point: [algorithmic bytes per launch]} as a third result."""
    global _profile, _profile_all, _profile_graph_ok
    prof, _profile, _profile_all, _profile_graph_ok = _profile or {}, None, False, False
    torch.cuda.synchronize()
    ms = {n: [a.elapsed_time(b) for a, b in pairs] for n, pairs in prof.items()}
    if with_work == "bytes":
        return (ms, {n: list(_profile_work.get(n, [])) for n in prof}, {n: list(_profile_bytes.get(n, [])) for n in prof})
    if with_work:
        return ms, {n: list(_profile_work.get(n, [])) for n in prof}
    return ms


# ---------------------------------------------------------------------------------------------------------
# a1 / a2
# ---------------------------------------------------------------------------------------------------------

def to_minmax(batch_x: torch.Tensor):
    """src/aa/utils.py:4-9 — returns (x01 (B,T), mn (B,1), mx (B,1))."""
    x = _require(batch_x, "batch_x")
    B, T = _rows(x, "batch_x")
    x01 = torch.empty_like(x)
    mn = torch.empty((B, 1), dtype=torch.float32, device=x.device)
    mx = torch.empty((B, 1), dtype=torch.float32, device=x.device)
    with _Launch("minmax_normalize", x.device, tensors=(x, x, x01)):
        ws, ws_bytes = _workspace(x.device, B, T)
        st = _lib.load().advstep_minmax_normalize_f32(x.data_ptr(), x01.data_ptr(), mn.data_ptr(), mx.data_ptr(), B, T,
                                                      ws, ws_bytes, _stream(x.device))
    _lib.check(st, "advstep_minmax_normalize_f32")
    return x01, mn, mx


def revert_minmax(batch_x: torch.Tensor, mn: torch.Tensor, mx: torch.Tensor, out: Optional[torch.Tensor] = None):
    """src/aa/utils.py:12-14."""
    x = _require(batch_x, "batch_x")
    B, T = _rows(x, "batch_x")
    _require(mn, "mn"), _require(mx, "mx")
    if mn.numel() != B or mx.numel() != B:
        raise ValueError(f"mn/mx must hold one value per row ({B}), got {mn.numel()} / {mx.numel()}")
    out = _out_like(x, out)
    with _Launch("minmax_revert", x.device, tensors=(x, out)):
        st = _lib.load().advstep_minmax_revert_f32(x.data_ptr(), mn.data_ptr(), mx.data_ptr(), out.data_ptr(), B, T,
                                                   _stream(x.device))
    _lib.check(st, "advstep_minmax_revert_f32")
    return out


# ---------------------------------------------------------------------------------------------------------
# a4 / a5
# ---------------------------------------------------------------------------------------------------------

def fgsm_step(x, grad, eps: float, lo: float = 0.0, hi: float = 1.0, out=None):
    _require(x, "x"), _require(grad, "grad")
    _same_shape(("x", x), ("grad", grad))
    out = _out_like(x, out)
    with _Launch("fgsm_step", x.device, tensors=(x, grad, out)):
        st = _lib.load().advstep_fgsm_step_f32(x.data_ptr(), grad.data_ptr(), out.data_ptr(), x.numel(), eps, lo, hi,
                                               _stream(x.device))
    _lib.check(st, "advstep_fgsm_step_f32")
    return out


def pgd_linf_init(x, eps: float, noise=None, seed: Optional[int] = None, offset: int = 0, lo: float = 0.0,
                  hi: float = 1.0, out=None):
    """Random start of pgd.py:54-57.  `noise` (the caller's U(-eps, eps) draw) or a Philox `seed`."""
    _require(x, "x")
    out = _out_like(x, out)
    if noise is not None:
        _require(noise, "noise")
        _same_shape(("x", x), ("noise", noise))
        with _Launch("pgd_linf_init", x.device, tensors=(x, noise, out)):
            st = _lib.load().advstep_pgd_linf_init_noise_f32(x.data_ptr(), noise.data_ptr(), out.data_ptr(), x.numel(),
                                                             lo, hi, _stream(x.device))
        _lib.check(st, "advstep_pgd_linf_init_noise_f32")
    else:
        if seed is None:
            raise ValueError("pgd_linf_init needs either `noise` or a Philox `seed`")
        with _Launch("pgd_linf_init", x.device, tensors=(x, out)):
            st = _lib.load().advstep_pgd_linf_init_philox_f32(x.data_ptr(), out.data_ptr(), x.numel(), eps, lo, hi,
                                                              seed, offset, _stream(x.device))
        _lib.check(st, "advstep_pgd_linf_init_philox_f32")
    return out


def pgd_linf_step(adv, grad, orig, alpha: float, eps: float, lo: float = 0.0, hi: float = 1.0, out=None):
    _require(adv, "adv"), _require(grad, "grad"), _require(orig, "orig")
    _same_shape(("adv", adv), ("grad", grad), ("orig", orig))
    out = _out_like(adv, out)
    with _Launch("pgd_linf_step", adv.device, tensors=(adv, grad, orig, out)):
        st = _lib.load().advstep_pgd_linf_step_f32(adv.data_ptr(), grad.data_ptr(), orig.data_ptr(), out.data_ptr(),
                                                   adv.numel(), alpha, eps, lo, hi, _stream(adv.device))
    _lib.check(st, "advstep_pgd_linf_step_f32")
    return out


# ---------------------------------------------------------------------------------------------------------
# a6
# ---------------------------------------------------------------------------------------------------------

def pgd_l2_init(x, eps: float, draws=None, seed: Optional[int] = None, offset: int = 0, lo: float = 0.0,
                hi: float = 1.0, out=None):
    """Random start of pgdl2.py:55-62.  `draws` = (normal (B,T), r (B)) or a Philox `seed`."""
    _require(x, "x")
    B, T = _rows(x, "x")
    out = _out_like(x, out)
    with _Launch("pgd_l2_init", x.device, tensors=(x, out)):
        ws, ws_bytes = _workspace(x.device, B, T)
        if draws is not None:
            normal, r = draws
            _require(normal, "normal"), _require(r, "r")
            _same_shape(("x", x), ("normal", normal))
            if r.numel() != B:
                raise ValueError(f"r must hold one value per row ({B}), got {r.numel()}")
            st = _lib.load().advstep_pgd_l2_init_noise_f32(x.data_ptr(), normal.data_ptr(), r.data_ptr(), out.data_ptr(),
                                                           B, T, eps, lo, hi, ws, ws_bytes, _stream(x.device))
            what = "advstep_pgd_l2_init_noise_f32"
        else:
            if seed is None:
                raise ValueError("pgd_l2_init needs either `draws` or a Philox `seed`")
            st = _lib.load().advstep_pgd_l2_init_philox_f32(x.data_ptr(), out.data_ptr(), B, T, eps, lo, hi, seed,
                                                            offset, ws, ws_bytes, _stream(x.device))
            what = "advstep_pgd_l2_init_philox_f32"
    _lib.check(st, what)
    return out


def pgd_l2_step(adv, grad, orig, alpha: float, eps: float, eps_div: float = 1e-10, lo: float = 0.0, hi: float = 1.0,
                out=None, return_norms: bool = False):
    _require(adv, "adv"), _require(grad, "grad"), _require(orig, "orig")
    _same_shape(("adv", adv), ("grad", grad), ("orig", orig))
    B, T = _rows(adv, "adv")
    out = _out_like(adv, out)
    gn = dn = None
    if return_norms:
        gn = torch.empty(B, dtype=torch.float32, device=adv.device)
        dn = torch.empty(B, dtype=torch.float32, device=adv.device)
    with _Launch("pgd_l2_step", adv.device, tensors=(adv, grad, orig, out)):
        ws, ws_bytes = _workspace(adv.device, B, T)
        st = _lib.load().advstep_pgd_l2_step_f32(adv.data_ptr(), grad.data_ptr(), orig.data_ptr(), out.data_ptr(), B, T,
                                                 alpha, eps, eps_div, lo, hi, gn.data_ptr() if return_norms else None,
                                                 dn.data_ptr() if return_norms else None, ws, ws_bytes,
                                                 _stream(adv.device))
    _lib.check(st, "advstep_pgd_l2_step_f32")
    return (out, gn, dn) if return_norms else out


def pgd_l2_repaired_rows(like: torch.Tensor) -> int:
    """Diagnostics (synchronises): how many rows of the LAST single-pass pgd_l2_step / pgd_l2_init call on this device and
    stream were recomputed by the repair kernel because their in-launch norm exchange was abandoned (include/advstep.h)."""
    _require(like, "like")
    B, T = _rows(like, "like")
    count = torch.zeros(1, dtype=torch.int32, device=like.device)
    with torch.cuda.device(like.device):
        ws, ws_bytes = _workspace(like.device, B, T)
        st = _lib.load().advstep_pgd_l2_repaired_rows(ws, ws_bytes, B, T, count.data_ptr(), _stream(like.device))
    _lib.check(st, "advstep_pgd_l2_repaired_rows")
    return int(count.item())


# ---------------------------------------------------------------------------------------------------------
# a7
# ---------------------------------------------------------------------------------------------------------

def cw_init_w(x, out=None):
    _require(x, "x")
    out = _out_like(x, out)
    with _Launch("cw_init_w", x.device, tensors=(x, out)):
        st = _lib.load().advstep_cw_init_w_f32(x.data_ptr(), out.data_ptr(), x.numel(), _stream(x.device))
    _lib.check(st, "advstep_cw_init_w_f32")
    return out


def cw_tanh_sqdist(w, x, adv_out=None):
    """Returns (adv = 1/2 (tanh w + 1), l2 (B) = row sums of (adv - x)^2)."""
    _require(w, "w"), _require(x, "x")
    _same_shape(("w", w), ("x", x))
    B, T = _rows(w, "w")
    adv = _out_like(w, adv_out, "adv_out")
    l2 = torch.empty(B, dtype=torch.float32, device=w.device)
    with _Launch("cw_tanh_sqdist", w.device, tensors=(w, x, adv)):
        ws, ws_bytes = _workspace(w.device, B, T)
        st = _lib.load().advstep_cw_tanh_sqdist_f32(w.data_ptr(), x.data_ptr(), adv.data_ptr(), l2.data_ptr(), B, T, ws,
                                                    ws_bytes, _stream(w.device))
    _lib.check(st, "advstep_cw_tanh_sqdist_f32")
    return adv, l2


def cw_adam_step(w, m, v, x, grad_adv, step: int, lr: float = 0.01, beta1: float = 0.9, beta2: float = 0.999,
                 adam_eps: float = 1e-8) -> None:
    """In-place Adam step on (w, m, v); `step` is 1-based."""
    for name, t in (("w", w), ("m", m), ("v", v), ("x", x), ("grad_adv", grad_adv)):
        _require(t, name)
    _same_shape(("w", w), ("m", m), ("v", v), ("x", x), ("grad_adv", grad_adv))
    with _Launch("cw_adam_step", w.device, tensors=(w, w, m, m, v, v, x, grad_adv)):
        st = _lib.load().advstep_cw_adam_step_f32(w.data_ptr(), m.data_ptr(), v.data_ptr(), x.data_ptr(),
                                                  grad_adv.data_ptr(), w.numel(), step, lr, beta1, beta2, adam_eps,
                                                  _stream(w.device))
    _lib.check(st, "advstep_cw_adam_step_f32")


def cw_best_update(adv, mask, best) -> None:
    """In place: best = mask * adv + (1 - mask) * best, mask (B) float32 in {0, 1}."""
    _require(adv, "adv"), _require(mask, "mask"), _require(best, "best")
    _same_shape(("adv", adv), ("best", best))
    B, T = _rows(adv, "adv")
    if mask.numel() != B:
        raise ValueError(f"mask must hold one value per row ({B}), got {mask.numel()}")
    with _Launch("cw_best_update", adv.device, tensors=(adv, best, best)):
        st = _lib.load().advstep_cw_best_update_f32(adv.data_ptr(), mask.data_ptr(), best.data_ptr(), B, T,
                                                    _stream(adv.device))
    _lib.check(st, "advstep_cw_best_update_f32")


# ---------------------------------------------------------------------------------------------------------
# a8
# ---------------------------------------------------------------------------------------------------------

def ce2_loss_grad(z, labels, scale: float = 1.0):
    """CE(cat([-z, z], 1), labels) (mean) in closed form: returns (d cost / d z shaped like z, cost (1,))."""
    _require(z, "z")
    _require(labels, "labels", torch.int64)
    B = z.numel()
    if labels.numel() != B:
        raise ValueError(f"labels must hold one value per logit ({B}), got {labels.numel()}")
    dz = torch.empty_like(z)
    loss = torch.empty(1, dtype=torch.float32, device=z.device)
    with _Launch("ce2_loss_grad", z.device):
        st = _lib.load().advstep_ce2_loss_grad_f32(z.data_ptr(), labels.data_ptr(), dz.data_ptr(), loss.data_ptr(), B,
                                                   scale, _stream(z.device))
    _lib.check(st, "advstep_ce2_loss_grad_f32")
    return dz, loss


# ---------------------------------------------------------------------------------------------------------
# f3: FAB (include/advstep_fab.h; reference adversarial_attacks/torchattacks/attacks/fab.py:208-292, 562-717)
# ---------------------------------------------------------------------------------------------------------

FAB_NORMS = {"Linf": 0, "L2": 1, "L1": 2}


def _fab_kind(norm: str) -> int:
    try:
        return FAB_NORMS[norm]
    except KeyError:
        raise ValueError("norm not supported") from None    # fab.py:224


def fab_hyperplane(gz, x, z=None, labels=None, norm: str = "Linf"):
    """Row statistics of the logit gradient + the closest-boundary selection for cat([-z, z]) (fab.py:90-112, 210-229).
    Returns (wscale, b, gnorm, gdot), each (B); wscale / b are None when z / labels are not given."""
    _require(gz, "gz"), _require(x, "x")
    _same_shape(("gz", gz), ("x", x))
    B, T = _rows(gz, "gz")
    dev = gz.device
    gnorm, gdot = torch.empty(B, device=dev), torch.empty(B, device=dev)
    wscale = b = None
    zp = lp = wp = bp = None
    if z is not None:
        _require(z, "z"), _require(labels, "labels", torch.int64)
        if z.numel() != B or labels.numel() != B:
            raise ValueError(f"z / labels must hold one value per row ({B})")
        wscale, b = torch.empty(B, device=dev), torch.empty(B, device=dev)
        zp, lp, wp, bp = z.data_ptr(), labels.data_ptr(), wscale.data_ptr(), b.data_ptr()
    with _Launch("fab_hyperplane", dev):
        st = _lib.load().advstep_fab_hyperplane_f32(gz.data_ptr(), x.data_ptr(), zp, lp, wp, bp, gnorm.data_ptr(),
                                                    gdot.data_ptr(), B, T, _fab_kind(norm), _stream(dev))
    _lib.check(st, "advstep_fab_hyperplane_f32")
    return wscale, b, gnorm, gdot


def fab_projection(points, w, b, norm: str = "Linf", wscale=None, out=None):
    """projection_{linf,l2,l1}(points, w, b) (fab.py:562-717) for R rows: returns (d (R, T), its attack norm (R)).
    w may hold fewer rows than points (R % w_rows == 0): row r uses w[r % w_rows] * wscale[r % w_rows]."""
    _require(points, "points"), _require(w, "w"), _require(b, "b")
    R, _ = _rows(points, "points")
    w_rows, _ = _rows(w, "w")
    T, Tw = points[0].numel() if R else w.shape[1:].numel(), w.shape[1:].numel()
    if Tw != T or w_rows == 0 or R % w_rows or b.numel() != R or w.device != points.device:
        raise ValueError(f"points {tuple(points.shape)}, w {tuple(w.shape)}, b {tuple(b.shape)} do not line up")
    if wscale is not None:
        _require(wscale, "wscale")
        if wscale.numel() != w_rows:
            raise ValueError(f"wscale must hold one value per row of w ({w_rows})")
    d = _out_like(points, out, "out")
    dnorm = torch.empty(R, device=points.device)
    with _Launch("fab_projection", points.device):
        st = _lib.load().advstep_fab_projection_f32(points.data_ptr(), w.data_ptr(),
                                                    None if wscale is None else wscale.data_ptr(), b.data_ptr(),
                                                    d.data_ptr(), dnorm.data_ptr(), R, w_rows, T, _fab_kind(norm),
                                                    _stream(points.device))
    _lib.check(st, "advstep_fab_projection_f32")
    return d, dnorm


def fab_combine(x1, x0, d1, d2, n1, n2, eta: float, alpha_max: float, out=None):
    """clamp((x1 + eta d1)(1 - alpha) + (x0 + eta d2) alpha, 0, 1), alpha from the two move norms (fab.py:257-267)."""
    for name, t in (("x1", x1), ("x0", x0), ("d1", d1), ("d2", d2), ("n1", n1), ("n2", n2)):
        _require(t, name)
    _same_shape(("x1", x1), ("x0", x0), ("d1", d1), ("d2", d2))
    B, T = _rows(x1, "x1")
    if n1.numel() != B or n2.numel() != B:
        raise ValueError(f"n1 / n2 must hold one value per row ({B})")
    res = _out_like(x1, out, "out")
    with _Launch("fab_combine", x1.device):
        st = _lib.load().advstep_fab_combine_f32(x1.data_ptr(), x0.data_ptr(), d1.data_ptr(), d2.data_ptr(), n1.data_ptr(),
                                                 n2.data_ptr(), res.data_ptr(), B, T, eta, alpha_max, _stream(x1.device))
    _lib.check(st, "advstep_fab_combine_f32")
    return res


def fab_backward_step(x1, x0, adv, res2, is_adv, beta: float, norm: str = "Linf") -> None:
    """In place, rows with is_adv != 0: keep the closest adversarial point so far, then step back towards the clean
    point by beta (fab.py:271-290).  is_adv (B) uint8 or bool."""
    _require(x1, "x1"), _require(x0, "x0"), _require(adv, "adv"), _require(res2, "res2")
    _same_shape(("x1", x1), ("x0", x0), ("adv", adv))
    B, T = _rows(x1, "x1")
    if is_adv.dtype == torch.bool:
        is_adv = is_adv.view(torch.uint8)
    _require(is_adv, "is_adv", torch.uint8)
    if res2.numel() != B or is_adv.numel() != B:
        raise ValueError(f"res2 / is_adv must hold one value per row ({B})")
    with _Launch("fab_backward_step", x1.device):
        st = _lib.load().advstep_fab_backward_step_f32(x1.data_ptr(), x0.data_ptr(), adv.data_ptr(), res2.data_ptr(),
                                                       is_adv.data_ptr(), B, T, beta, _fab_kind(norm), _stream(x1.device))
    _lib.check(st, "advstep_fab_backward_step_f32")
