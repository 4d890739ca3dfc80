"""hipGraph replay of the iterative attacks' inner loop (PGD, PGDL2).

One attack iteration on LCNN is ~37 kernel launches (model forward + input-backward through the fused kernels, the
closed-form loss gradient, the fused update step) of fixed shapes; launched eagerly the launching thread is busy for as
long as the GPU is, so with the data loader's collation on the same thread — or eight ranks sharing one host — the loop
turns host-bound.  Here the MODEL PART of an iteration — forward, loss gradient, input-backward —
is captured once per (model state, batch shape, attack hyper-parameters, launch stream) into two hipGraphs, one reading
each of the two ping-pong buffers, and the update step is launched between the replays:
    replay G_a: grad_a = d cost / d adv_a      step(adv_a, grad_a) -> adv_b
    replay G_b: grad_b = d cost / d adv_b      step(adv_b, grad_b) -> adv_a            (steps / 2 times)
Round 6: rounds 2-5 captured BOTH iterations including their steps in one graph ("fused", still behind
ADVSTEP_ATTACK_GRAPH=fused for A/B).  The split form costs the host one more graph launch and two plain launches per pair
(microseconds against a 3.4 ms pair) and buys two things: the step kernel — the kernel bench.py prices — can be bracketed
with HIP events while the loop replays (events cannot be recorded into a graph), so the measured loop IS the shipped loop;
and nothing of the launch stream's scratch is baked into a graph.

Two batches in flight (round 6, evaluation.generate_attacks / bench.py): the key carries the LAUNCH STREAM, so two streams
that alternate batches get a capture each — a capture owns its static buffers and its activation pool and cannot serve
two batches at once — over the same read-only model.

What is baked into a captured graph and therefore part of its key: the device pointers of the static input buffers (owned
here), of the model's parameters and of every cache derived from them (prepared Winograd weights, packed GRU weights,
filterbank tables) — so the key carries the parameters' version counters and the modules' train/eval flags; the ADVSTEP_* switches and the labels'
dtype / shape; a graph is captured only when the same key shows up a second time (an adversarial-training step changes the
weights before every attack call: those calls stay eager), and capturing a workload under a new state drops the captures of
the same workload under older states (each holds a private memory pool).  Random starts are drawn OUTSIDE the graph (a fresh Philox key per call).

Eager fallback, always bit-identical: CPU op tables / checked ops (tests), launch profiling of kernels INSIDE the model
part (hip_ops.start_profile without graph_ok: bench.py's per-family brackets), ADVSTEP_ATTACK_GRAPH=0, or a failed capture."""
from __future__ import annotations

import os
import warnings
from typing import Callable, Dict, Optional, Tuple

import torch

_GRAPHS: Dict[tuple, "_Captured"] = {}
_SEEN: Dict[tuple, int] = {}
_FAILED: set = set()          # family keys (model, attack, shape, hyper-parameters) whose capture failed: they stay eager
_MAX_GRAPHS = 8


def _toggles() -> Tuple:
    """Every ADVSTEP_* environment switch, sorted: the model's forward and the step kernels read them at call time, so a
    captured kernel sequence is only valid for the values it was captured under (A/B runs flip them between calls)."""
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("ADVSTEP_")))


def enabled() -> bool:
    return os.environ.get("ADVSTEP_ATTACK_GRAPH", "1") != "0"


def fused_form() -> bool:
    """ADVSTEP_ATTACK_GRAPH=fused: rounds 2-5's single graph of two whole iterations (A/B; cannot carry event brackets)."""
    return os.environ.get("ADVSTEP_ATTACK_GRAPH", "1") == "fused"


def _state_signature(model: torch.nn.Module) -> Tuple:
    versions = tuple((p.data_ptr(), p._version) for p in model.parameters())
    buffers = tuple((b.data_ptr(), b._version) for b in model.buffers())
    flags = tuple(m.training for m in model.modules())
    return versions, buffers, flags


class _Captured:
    def __init__(self, attack, images, labels, target, step_fn, fused: bool):
        dev = images.device
        self.fused, self.step_fn = fused, step_fn
        self.images = torch.empty_like(images)
        self.adv_a = torch.empty_like(images)
        self.adv_b = torch.empty_like(images)
        self.labels = torch.empty_like(labels)
        self.target = torch.empty_like(target) if target is not None else None
        self.images.copy_(images), self.labels.copy_(labels), self.adv_a.copy_(images)
        if target is not None:
            self.target.copy_(target)

        def two_iterations():
            grad, _ = attack._input_gradient(self.adv_a, self.labels, self.target)
            step_fn(self.adv_a.detach(), grad, self.images, self.adv_b)
            grad, _ = attack._input_gradient(self.adv_b, self.labels, self.target)
            step_fn(self.adv_b.detach(), grad, self.images, self.adv_a)

        # warm-up on a side stream (torch's capture protocol): builds every lazily created cache and workspace
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            two_iterations()
        torch.cuda.current_stream(dev).wait_stream(side)
        from .. import hip_ops
        # thread-local capture mode: with N > 1 ranks the RCCL watchdog thread (event queries), and in the CLI the DataLoader's
        # pinning thread, make HIP calls of their own while this thread captures; in the default "global" mode any such call
        # invalidates the capture
        # captured on the warm-up stream: everything keyed by the launch stream (hip_ops' row-reduction workspace) was created
        # by the warm-up, OUTSIDE the capture, so nothing that outlives this object is allocated from the graph's private pool
        if fused:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                two_iterations()
        else:
            # the model part only; the gradients are the graphs' OUTPUTS: they live in the graphs' shared private pool and keep
            # their addresses as long as this object holds them.  G_b re-uses G_a's pool: the two never run at the same time
            # (one stream), so the second forward + backward's activations lie over the first's.
            self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a, stream=side, capture_error_mode="thread_local"):
                self.grad_a, _ = attack._input_gradient(self.adv_a, self.labels, self.target)
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), stream=side, capture_error_mode="thread_local"):
                self.grad_b, _ = attack._input_gradient(self.adv_b, self.labels, self.target)
        # A fused graph bakes in the ADDRESSES of hip_ops' row-reduction workspaces of (device, warm-up stream, batch shape): it
        # takes them OUT of hip_ops' table and owns them from here on — nothing hip_ops does later can free a buffer the graph
        # writes to on every replay, and no eager call on a stream that re-uses the warm-up stream's handle is ever given the same
        # buffer (one workspace, one stream: include/advstep.h).  (The split form's graphs contain no step kernel;
        # the warm-up's buffers simply leave the table with the stream.)
        self._workspaces = hip_ops.release_stream_workspaces(side.cuda_stream)

    def run(self, adv, images, labels, target, pairs: int) -> torch.Tensor:
        # (adv_a / adv_b became autograd leaves during the capture: write through detached views)
        self.images.copy_(images), self.labels.copy_(labels), self.adv_a.detach().copy_(adv)
        if target is not None:
            self.target.copy_(target)
        if self.fused:
            for _ in range(pairs):
                self.graph.replay()
            return self.adv_a.detach()
        a, b = self.adv_a.detach(), self.adv_b.detach()
        for _ in range(pairs):
            self.graph_a.replay()
            self.step_fn(a, self.grad_a, self.images, b)       # eager, on the launch stream: bracketable (hip_ops._Launch)
            self.graph_b.replay()
            self.step_fn(b, self.grad_b, self.images, a)
        return a


def run_iterations(attack, adv: torch.Tensor, images: torch.Tensor, labels: torch.Tensor, target: Optional[torch.Tensor],
                   steps: int, step_fn: Callable, hyper: Tuple) -> torch.Tensor:
    """`steps` iterations of  adv <- step_fn(adv, grad(adv), images, out)  starting from `adv`; returns the final adv
    (detached, own storage).  step_fn(adv, grad, images, out) must write `out` (a different buffer than `adv`)."""
    from .. import hip_ops
    ops = attack.ops
    fused = fused_form()
    # launch profiling: brackets around kernels of the model part need eager launches; brackets around the step kernels only
    # (start_profile(..., graph_ok=True): bench.py's timed region) go with the split form, whose steps are plain launches
    profiling_allows = hip_ops._profile is None or (hip_ops._profile_graph_ok and not fused)
    use_graph = (enabled() and ops is hip_ops and profiling_allows and adv.is_cuda and steps >= 4
                 and not torch.cuda.is_current_stream_capturing())
    done = 0
    if use_graph:
        # family: what makes two calls the same workload; state: what a capture bakes in beyond that (parameter / buffer
        # storage and versions, train/eval flags, the ADVSTEP_* switches the kernels read at call time)
        # ... and the launch stream: a capture serves ONE batch at a time (its static buffers, its activation pool), so two
        # streams with a batch in flight each (evaluation.generate_attacks) hold a capture each
        family = (id(attack.model), attack.__class__.__name__, hyper, attack._targeted, tuple(adv.shape), str(adv.device),
                  labels.dtype, tuple(labels.shape), torch.cuda.current_stream(adv.device).cuda_stream, fused)
        key = family + (_state_signature(attack.model), _toggles())
        cap = _GRAPHS.get(key)
        if cap is None and family not in _FAILED:
            _SEEN[key] = _SEEN.get(key, 0) + 1
            if len(_SEEN) > 64:
                _SEEN.clear()
            if _SEEN.get(key, 0) >= 2:
                # a capture of this family under ANOTHER state can never be replayed again once the state moved on (training
                # changed the weights, a switch was flipped): drop it now — each holds a private pool with the whole
                # forward + backward activation set
                for stale in [k for k in _GRAPHS if k[:len(family)] == family]:
                    del _GRAPHS[stale]
                for stale in [k for k in _SEEN if k[:len(family)] == family and k != key]:
                    del _SEEN[stale]
                try:
                    cap = _Captured(attack, images, labels, target, step_fn, fused)
                    if len(_GRAPHS) >= _MAX_GRAPHS:
                        _GRAPHS.pop(next(iter(_GRAPHS)))
                    _GRAPHS[key] = cap
                except Exception as exc:  # noqa: BLE001 — any capture problem means: this workload stays eager, loudly
                    _FAILED.add(family)
                    torch.cuda.synchronize()
                    warnings.warn(f"hipGraph capture of the {attack.__class__.__name__} iteration failed ({exc!r}); "
                                  "continuing with eager launches for this workload")
                    cap = None
        if cap is not None:
            adv = cap.run(adv, images, labels, target, steps // 2).detach().clone()
            done = 2 * (steps // 2)
    spare = None
    for _ in range(steps - done):
        grad, _ = attack._input_gradient(adv, labels, target)
        out = spare if spare is not None else torch.empty_like(adv)
        step_fn(adv.detach(), grad, images, out)
        spare, adv = adv.detach(), out
    return adv.detach()


def clear() -> None:
    """Drop every captured graph (tests; also releases the graphs' private memory pools)."""
    _GRAPHS.clear()
    _SEEN.clear()
    _FAILED.clear()
