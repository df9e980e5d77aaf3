"""Training-mode entry points: ``torch.autograd.Function`` wrappers around the HIP forward /
backward kernels, so ``total_loss.backward()`` + ``torch.optim.Adam`` of the reference training
loop (train_dmsr.py:56-64) work unchanged.

Gradient barriers of the reference are reproduced structurally:
  * ``weights_ins.detach()`` (render.py:22-23)  -- composite_bwd gives the ins logits no path into sigma;
  * ``h.detach()`` on the ins branch (dm_nerf.py:95) -- mlp_bwd_data does not add d(ins_feature) to dh_7;
  * ``z_samples.detach()`` (render.py:68) -- the resampled depths are computed outside autograd.

Weight gradients dW = dy . x^T (GEMMs over the M samples of the batch, both operands stored
feature-major by the kernels) and the bias row sums run in the split-K f32-MFMA kernel of
csrc/wgrad.hip (dmnerf_mlp_bwd_weights); no vendor BLAS is involved.
"""
import torch

from . import _lib
from .networks import helpers

W = 256
HW = 128

# Measurement hook (bench.py): when this is a list, the three MFMA kernels of the training step -- saved-activation
# forward, data gradient, weight gradient -- are bracketed by HIP events on the stream they are launched on and
# ``(kernel, samples M, begin, end)`` is appended per launch.  None (the default): nothing is recorded.
KERNEL_EVENTS = None
PLAN_MAX_WGS = None


# ---- run-time honesty of the opt-in f16x2 mode --------------------------------------------------------------------------------
# The split-f16 kernels saturate at 65 504 instead of overflowing (csrc/split_f16.h) and nothing at run time says so.  With
# ``DMNERF_CHECK_F16=1`` (or ``autograd.CHECK_F16 = True``, or ``args.check_f16``) every f16x2 training forward / backward is
# followed by one pass over the workspace it wrote (dmnerf_f16x2_range_flags: exactly the operands the kernels converted) that
# ORs into a sticky per-device flags word; ``check_f16x2()`` reads it back (one sync) and warns.  Off by default: no launch, no
# cost.  The inference path (networks/render.py) probes through the training forward when the check is on.
CHECK_F16 = None
F16_ACT_SATURATED, F16_GRAD_SATURATED = 1, 2          # DMNERF_F16_* (include/dmnerf_hip.h)
_f16_flags = {}


def f16_check_enabled(args=None):
    if args is not None and getattr(args, "check_f16", None) is not None:
        return bool(args.check_f16)
    if CHECK_F16 is not None:
        return bool(CHECK_F16)
    import os
    return os.environ.get("DMNERF_CHECK_F16", "0") == "1"


PROBE_SAMPLES = 262144              # f16x2_probe: samples per piece (a 2.6 GB scratch workspace)


def _f16_flags_word(device):
    key = str(device)
    if key not in _f16_flags:
        _f16_flags[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _f16_flags[key]


def _f16_range_scan(workspace, M, gradients):
    _lib.check(_lib.load().dmnerf_f16x2_range_flags(_lib.ptr(workspace), M, 1 if gradients else 0, _lib.ptr(_f16_flags_word(workspace.device)),
                                                    _lib.stream()), "dmnerf_f16x2_range_flags")


def check_f16x2(device=None, reset=True, warn=True):
    """Read the sticky f16x2 saturation flags of ``device`` (one host synchronisation) -> int (0: nothing saturated;
    ``F16_ACT_SATURATED``: an activation reached |x| >= 65 504 and was clamped by the conversion; ``F16_GRAD_SATURATED``: a scaled
    gradient did).  ``warn``: issue a ``RuntimeWarning`` naming what happened; ``reset``: clear the word."""
    import warnings
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    word = _f16_flags.get(str(device))
    if word is None:
        return 0
    flags = int(word.item())
    if reset and flags:
        word.zero_()
    if warn and flags & F16_ACT_SATURATED:
        warnings.warn("dm_nerf_amd f16x2: an activation reached |x| >= 65504 and was clamped by the f16 conversion -- results of "
                      "args.mfma_split = 'f16x2' are not f32-class for this network / input; use the default f32 kernels or 'bf16x3'",
                      RuntimeWarning, stacklevel=2)
    if warn and flags & F16_GRAD_SATURATED:
        warnings.warn("dm_nerf_amd f16x2: a scaled gradient reached |x| >= 65504 in the data-gradient pass (dynamic range of "
                      "dL/draw above ~2^11) and was clamped; use the default f32 kernels or 'bf16x3' for this step", RuntimeWarning, stacklevel=2)
    return flags


def f16x2_probe(model, rays_o, rays_d, z):
    """Inference with the check on: the same rays through the training forward (which SAVES what it converts) into a scratch
    workspace, scanned for saturation.  Doubles the cost of the call -- a diagnostic.  Runs in pieces of at most
    MAX_TRAIN_SAMPLES // S rays (the training forward's launch limit; one piece's workspace is ~2466 floats per sample), so an
    inference chunk of any size can be probed."""
    lib = _lib.load()
    N, S = z.shape
    step = max(1, min(N, MAX_TRAIN_SAMPLES // S, max(1, PROBE_SAMPLES // S)))
    raw = torch.empty(step, S, 4 + model.ins_num + 1, dtype=torch.float32, device=z.device)
    save = torch.empty(lib.dmnerf_train_save_floats(step * S), dtype=torch.float32, device=z.device)
    for s0 in range(0, N, step):
        n = min(step, N - s0)
        _lib.check(lib.dmnerf_mlp_fwd_rays_train_f16(_lib.ptr(model.blob_f16()), model.ins_num, _lib.ptr(rays_o[s0:s0 + n]),
                                                     _lib.ptr(rays_d[s0:s0 + n]), _lib.ptr(z[s0:s0 + n]), n, S, _lib.ptr(raw), _lib.ptr(save),
                                                     _lib.stream()), "dmnerf_mlp_fwd_rays_train_f16")
        _f16_range_scan(save, n * S, False)


class _timed:
    def __init__(self, tag, M):
        self.tag, self.M = tag, M

    def __enter__(self):
        if KERNEL_EVENTS is not None:
            self.b, self.e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.b.record()

    def __exit__(self, *exc):
        if KERNEL_EVENTS is not None:
            self.e.record()
            KERNEL_EVENTS.append((self.tag, self.M, self.b, self.e))
        return False


def _row_len(M):
    """Rows of the training workspace are padded to a multiple of 32 samples (csrc/layout.h::save_row_len)."""
    return (M + 31) & ~31


def _views(buf, M):
    """Feature-major [rows, row_len] COPIES of the tensors of a SaveLayout workspace (csrc/layout.h) --
    diagnostics / tests only.  On the device every tensor is block-major [block][rows][32 samples];
    padding columns hold duplicates of the last sample (activations) or exact zeros (gradients).  The
    accumulator-layout tensors (everything but the encodings) keep the memory row order 0,4,1,5,2,6,3,7
    inside each group of 8 features (csrc/layout.h::row_feature); the copies are in feature order."""
    Mp = _row_len(M)
    o = 0
    out = {}
    for name, rows, n in (("pe", 63, 1), ("de", 27, 1), ("h", W, 8), ("g1", HW, 1), ("g2", HW, 1)):
        ts = []
        for _ in range(n):
            t = buf[o:o + rows * Mp].view(Mp // 32, rows, 32).permute(1, 0, 2).reshape(rows, Mp)
            if name not in ("pe", "de"):
                rho = torch.arange(rows, device=buf.device)
                feat = (rho & ~7) | ((rho & 7) >> 1) | ((rho & 1) << 2)
                t = t[torch.argsort(feat)]
            ts.append(t)
            o += rows * Mp
        out[name] = ts[0] if n == 1 else torch.stack(ts)
    return out


_plan_cache = {}


def wgrad_plan(ins_num, M, device, max_wgs=None, split=False):
    """Device-resident split-K plan of the weight-gradient kernel for (ins_num, M): (jobs, n_jobs, outs, n_outs, part_floats).
    ``split``: False = balanced for the f32 kernel, True / "bf16x3" = for the opt-in split-bf16 kernel (csrc/wgrad_split.hip),
    "f16x2" = for the split-f16 kernel (csrc/wgrad_f16.hip); any plan is valid for any kernel."""
    import ctypes

    import numpy as np
    if max_wgs is None:
        # (PLAN_MAX_WGS: diagnostic override -- another workgroup budget gives another split of the sample axis, i.e. the SAME
        # gradients summed in another order; tests/test_gpu_convergence.py uses it to measure how far two correct f32 trainings
        # drift apart)
        max_wgs = PLAN_MAX_WGS or torch.cuda.get_device_properties(device).multi_processor_count     # one workgroup per CU
    kind = "f16x2" if split == "f16x2" else ("bf16x3" if split else None)
    key = (ins_num, M, str(device), max_wgs, kind)
    if key not in _plan_cache:
        lib = _lib.load()
        f_sizes, f_plan = {None: (lib.dmnerf_wgrad_plan_sizes, lib.dmnerf_wgrad_plan),
                           "bf16x3": (lib.dmnerf_wgrad_plan_sizes_split, lib.dmnerf_wgrad_plan_split),
                           "f16x2": (lib.dmnerf_wgrad_plan_sizes_f16, lib.dmnerf_wgrad_plan_f16)}[kind]
        jb, ob, pf = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        nj, no = ctypes.c_int(), ctypes.c_int()
        _lib.check(f_sizes(ins_num, M, max_wgs, ctypes.byref(jb), ctypes.byref(ob), ctypes.byref(pf),
                           ctypes.byref(nj), ctypes.byref(no)), "dmnerf_wgrad_plan_sizes")
        hj, ho = np.empty(jb.value, dtype=np.uint8), np.empty(ob.value, dtype=np.uint8)
        _lib.check(f_plan(ins_num, M, max_wgs, hj.ctypes.data_as(ctypes.c_void_p), jb.value,
                          ho.ctypes.data_as(ctypes.c_void_p), ob.value), "dmnerf_wgrad_plan")
        while len(_plan_cache) >= 32:                      # a training run has one or two batch sizes; ragged callers do not pile up plans
            _plan_cache.pop(next(iter(_plan_cache)))
        _plan_cache[key] = (torch.from_numpy(hj).to(device), nj.value, torch.from_numpy(ho).to(device), no.value, pf.value)
    return _plan_cache[key]


def split_flat_grads(model, flat):
    """Views of the flat gradient vector (reference parameter order) as the model's parameter tensors."""
    out, o = [], 0
    for _, p in model.named_parameters():
        n = p.numel()
        out.append(flat[o:o + n].view_as(p))
        o += n
    return out


# model -> (arena, slot): a WEAK registry (an attribute on the modules would travel with copy.deepcopy / torch.save of the model
# and tie model and arena into a reference cycle)
import weakref

_arena_of = weakref.WeakKeyDictionary()


def arena_slot(model):
    """(arena, slot index) of the GradArena this model belongs to, or None."""
    ent = _arena_of.get(model)
    if ent is None:
        return None
    arena = ent[0]()
    return None if arena is None else (arena, ent[1])


class GradArena:
    """ONE flat f32 buffer holding the gradients of several models back to back, in ``named_parameters`` order (coarse
    model first: 2 x 696 338 floats = 5.57 MB at ins_num 13).  The weight-gradient kernel already produces one flat
    vector per model; with an arena it writes that vector straight into the model's slot, autograd installs the
    per-parameter views as ``p.grad`` without copying, and the data-parallel step all-reduces ``arena.flat`` IN PLACE --
    one RCCL message, no ``cat`` before it and no 60 ``copy_`` after it (distributed.allreduce_grads).

    ``begin_step()`` marks every slot free; the first backward launch of a model in that step takes the slot, further
    launches of the same model (batches beyond DMNERF_MAX_TRAIN_SAMPLES run as several) use scratch vectors that
    autograd ADDS into the installed views, i.e. into the arena.

    The gradients of a sharded step ALIAS the arena: the next ``begin_step()`` + backward overwrites any ``p.grad`` tensor a
    caller kept from the previous step (clone it for logging / accumulation across steps).  The models refer to their arena
    weakly (``arena_slot``); the caller -- ``distributed.sharded_train_step`` keeps it in a small cache -- owns it."""

    def __init__(self, models):
        self.models = list(models)
        lib = _lib.load()
        dev = next(self.models[0].parameters()).device
        sizes = [int(lib.dmnerf_param_count(m.ins_num)) for m in self.models]
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.slots, o = [], 0
        for m, n in zip(self.models, sizes):
            self.slots.append(self.flat[o:o + n])
            _arena_of[m] = (weakref.ref(self), len(self.slots) - 1)
            o += n
        self.free = [False] * len(self.models)

    def begin_step(self):
        self.free = [True] * len(self.models)

    def hold(self):
        """No slot is handed out until the next ``begin_step()``: the backward launches write scratch vectors and autograd ADDS
        them into the installed ``p.grad`` tensors.  For steps whose gradients must ACCUMULATE into the views that are already
        installed (``zero_grad(set_to_none=False)``, several backward passes per step): a launch that took the slot would
        overwrite, in place, the very memory ``p.grad`` views, and AccumulateGrad would then add that memory onto itself."""
        self.free = [False] * len(self.models)

    def take(self, i):
        if self.free[i]:
            self.free[i] = False
            return self.slots[i]
        return None

    def resident(self):
        """True when every parameter's ``.grad`` IS its view of the arena (what autograd installs after
        ``zero_grad(set_to_none=True)`` + backward)."""
        base = self.flat.data_ptr()
        o = 0
        for m in self.models:
            for _, p in m.named_parameters():
                g = p.grad
                if g is None or g.data_ptr() != base + 4 * o or not g.is_contiguous() or g.dtype != torch.float32:
                    return False
                o += p.numel()
        return o == self.flat.numel()


_arena_cache = []          # strong references to the most recent arenas (a training run has one)


def grad_arena(models):
    """The arena shared by exactly these models (created on first use; the last few are kept alive here)."""
    models = list(models)
    cur = arena_slot(models[0])
    if cur is not None and len(cur[0].models) == len(models) and all(a is b for a, b in zip(cur[0].models, models)) \
            and cur[0].flat.device == next(models[0].parameters()).device:
        return cur[0]
    arena = GradArena(models)
    _arena_cache.append(arena)
    del _arena_cache[:-4]
    return arena


class MLPRaysFunction(torch.autograd.Function):
    """raw[N,S,4+C] = DM_NeRF(embed(o + d z) | embed(d/|d|)); parameters are inputs 5.. in state_dict order.
    ``mode``: None (default f32 kernels) | "fused" | "split" (the opt-in variants of run_network_train)."""

    @staticmethod
    def forward(ctx, model, mode, rays_o, rays_d, z, *params):
        lib = _lib.load()
        N, S = z.shape
        M = N * S
        ins_num = model.ins_num
        raw = torch.empty(N, S, 4 + ins_num + 1, dtype=torch.float32, device=z.device)
        save = torch.empty(lib.dmnerf_train_save_floats(M), dtype=torch.float32, device=z.device)
        blob = None if mode in ("split", "f16") else model.blob()  # (the f32 dgrad reads it; the split kernels have their own blobs)
        if mode == "split":
            fwd_blob, fn = model.blob_split(), lib.dmnerf_mlp_fwd_rays_train_split
        elif mode == "f16":
            fwd_blob, fn = model.blob_f16(), lib.dmnerf_mlp_fwd_rays_train_f16
        elif mode == "fused":
            fwd_blob, fn = model.blob_fused(), lib.dmnerf_mlp_fwd_rays_train_fused
        else:
            fwd_blob, fn = blob, lib.dmnerf_mlp_fwd_rays_train
        with _timed("mlp_fwd_train", M):
            _lib.check(fn(_lib.ptr(fwd_blob), ins_num, _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(z),
                          N, S, _lib.ptr(raw), _lib.ptr(save), _lib.stream()), "dmnerf_mlp_fwd_rays_train")
        ctx.check_f16 = mode == "f16" and (_check_f16_call[0] if _check_f16_call[0] is not None else f16_check_enabled())
        if ctx.check_f16:
            _f16_range_scan(save, M, False)
        ctx.model, ctx.M, ctx.save = model, M, save
        ctx.blob, ctx.flat = blob, model.flat()                                   # the weights this forward used
        ctx.blob_t = ctx.blob_ts = None
        if mode == "split":
            ctx.blob_ts = model.blob_t_split()
        elif mode == "f16":
            ctx.blob_ts = model.blob_t_f16()
        else:
            ctx.blob_t = model.blob_t()
        ctx.mode = mode
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        return (None, None, None, None, None) + _mlp_backward(ctx, g_raw)


def _grad_scale_buffer(device):
    """The 4 floats of dmnerf_grad_scale ({2^s, 2^-s, scratch, scratch}), one buffer PER BACKWARD, zeroed here by a fill on the
    launch stream (a memset node under graph capture).  Nothing is shared between backwards: two models trained on two streams do
    not race on the scale pair or on the atomic scratch words, and a kernel that died mid-run cannot leave a non-zero scratch word
    behind for the next step (the kernel's own reset of the scratch is no longer relied upon)."""
    return torch.zeros(4, dtype=torch.float32, device=device)


class _Overlap:
    """State of ``overlapped_backward``: the side stream of each device, and the backward currently in flight on it."""
    active = False
    streams = {}
    pending = None          # (done event, [tensors the side-stream kernels read that the main stream's allocator must not reuse yet])
    seen = set()            # ids of the models that already had a network backward in this pass (see _mlp_backward)


class overlapped_backward:
    """``with overlapped_backward(): loss.backward()`` -- the two levels' network backwards run CONCURRENTLY (extension; the
    reference's backward is one stream, train_dmsr.py:63).

    The coarse and the fine network are separate autograd graphs below the loss (``z_samples.detach()``, render.py:68), so
    their data-gradient and weight-gradient launches do not depend on each other.  Inside this context the first network
    backward of the pass (autograd runs the fine network's first) is issued on a side stream that waits for the main one, the
    next runs on the main stream next to it, and the main stream then waits for the side one (also when the context exits).
    At the per-rank shard of an 8-way split (384 rays) the fine launch is 576 workgroups = 2.25 rounds of the 256 CUs and the
    coarse one 192 = 0.75: together exactly three rounds instead of 3 + 1 (bench.py ``train_shard_proxy``).

    Only for a pass whose parameter gradients start as ``None`` (``zero_grad(set_to_none=True)``: autograd then INSTALLS the
    kernels' output as ``p.grad`` without reading it); anything that would read a gradient on the main stream before the join --
    accumulation into existing ``.grad`` tensors -- makes ``_mlp_backward`` fall back to the main stream for that model.  Gradient
    hooks on the parameters are the caller's responsibility (none in this package).  ``distributed.sharded_train_step`` uses it;
    plain ``loss.backward()`` of the drop-in functions stays on one stream."""

    def __init__(self, enabled=True, join_on_exit=True):
        self.enabled = bool(enabled)
        self.join_on_exit = bool(join_on_exit)     # False: leave the side-stream backward in flight (the next pass / _join_side() joins it)

    def __enter__(self):
        self.prev = _Overlap.active
        _Overlap.active = self.enabled
        _Overlap.seen = set()
        return self

    def __exit__(self, *exc):
        if self.join_on_exit:
            _join_side()
        _Overlap.active = self.prev
        _Overlap.seen = set()
        return False


def join_side():
    """Public form of ``_join_side``: anything that reads gradients on the current stream after an
    ``overlapped_backward(join_on_exit=False)`` -- an optimizer step, a gradient all-reduce -- calls this first (no-op when nothing
    is in flight)."""
    _join_side()


def _join_side():
    """The main (current) stream waits for the side-stream backward in flight, if any; its inputs may be reused after that."""
    if _Overlap.pending is not None:
        done, keep = _Overlap.pending
        torch.cuda.current_stream().wait_event(done)
        _Overlap.pending = None
        del keep


def _mlp_backward(ctx, g_raw):
    """dgrad + wgrad of one saved forward (rays or pre-embedded rows): the parameter gradients as views of one flat vector.
    Inside ``overlapped_backward`` the first call of a pass runs on the side stream (see there).

    Only the FIRST backward launch of a model in a pass may go to the side stream.  A model whose batch exceeds
    MAX_TRAIN_SAMPLES runs as several launches F3, F2, F1 (autograd calls them last chunk first); ``p.grad`` stays ``None`` until
    all of them have returned (AccumulateGrad waits for every input), so "gradients are None and nothing is pending" also holds
    at F1 -- after F3 went to the side stream and F2 joined it -- and F1 on the side stream would let the engine's input buffer
    add its output to (F3 + F2) on the main stream while the weight-gradient kernel is still writing it.  Hence the
    per-pass set of models already seen: their later launches stay on the main stream, after a join."""
    first_of_model = id(ctx.model) not in _Overlap.seen
    if _Overlap.active:
        _Overlap.seen.add(id(ctx.model))
    if (_Overlap.active and _Overlap.pending is None and first_of_model and ctx.M > 0 and g_raw.is_cuda
            and all(p.grad is None for p in ctx.model.parameters())):
        dev = g_raw.device
        side = _Overlap.streams.get(dev.index)
        if side is None:
            side = _Overlap.streams[dev.index] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        # everything the side-stream kernels READ that was allocated on the main stream stays referenced until the join: freed
        # earlier, the main stream's allocator could hand the block to a kernel that runs while they are still reading
        keep = [g_raw, ctx.save, ctx.flat, getattr(ctx, "blob", None), getattr(ctx, "blob_t", None), getattr(ctx, "blob_ts", None)]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            grads = _mlp_backward_on_stream(ctx, g_raw)
            done = torch.cuda.Event()
            done.record(side)
        _Overlap.pending = (done, keep)
        return grads
    grads = _mlp_backward_on_stream(ctx, g_raw)
    _join_side()
    return grads


def _mlp_backward_on_stream(ctx, g_raw):
    lib = _lib.load()
    model, M = ctx.model, ctx.M
    ins_num = model.ins_num
    C = ins_num + 1
    if M == 0:                                                   # an empty batch (e.g. a rank's empty shard): zero gradients
        ctx.save = None
        arena = arena_slot(model)
        flat = arena[0].take(arena[1]) if arena is not None and arena[0].flat.device == g_raw.device else None
        if flat is None:
            flat = torch.empty(lib.dmnerf_param_count(ins_num), dtype=torch.float32, device=g_raw.device)
        return tuple(split_flat_grads(model, flat.zero_()))
    g = _lib.f32(g_raw).reshape(M, 4 + C)
    dsave = torch.empty_like(ctx.save)
    Mp = _row_len(M)
    gt = torch.empty(Mp // 32, 4 + C, 32, dtype=torch.float32, device=g.device)   # d raw, block-major, written by the kernel
    split = getattr(ctx, "blob_ts", None) is not None            # opt-in: split-bf16 backward kernels (args.mfma_split)
    f16 = getattr(ctx, "mode", None) == "f16"
    scale = _grad_scale_buffer(g.device) if f16 else None         # {2^s, 2^-s}: the f16 backward runs on 2^s dL/draw (f16 range)
    with _timed("mlp_bwd_data", M):
        if f16:
            _lib.check(lib.dmnerf_grad_scale(_lib.ptr(g), g.numel(), _lib.ptr(scale), _lib.stream()), "dmnerf_grad_scale")
            _lib.check(lib.dmnerf_mlp_bwd_data_f16(_lib.ptr(ctx.blob_ts), ins_num, _lib.ptr(ctx.save), _lib.ptr(g), M,
                                                   _lib.ptr(dsave), _lib.ptr(gt), _lib.ptr(scale), _lib.stream()), "dmnerf_mlp_bwd_data_f16")
            if getattr(ctx, "check_f16", None) if getattr(ctx, "check_f16", None) is not None else f16_check_enabled():
                _f16_range_scan(dsave, M, True)
        elif split:
            _lib.check(lib.dmnerf_mlp_bwd_data_split(_lib.ptr(ctx.blob_ts), ins_num, _lib.ptr(ctx.save), _lib.ptr(g), M,
                                                     _lib.ptr(dsave), _lib.ptr(gt), _lib.stream()), "dmnerf_mlp_bwd_data_split")
        else:
            _lib.check(lib.dmnerf_mlp_bwd_data(_lib.ptr(ctx.blob), _lib.ptr(ctx.blob_t), ins_num, _lib.ptr(ctx.save), _lib.ptr(g), M,
                                               _lib.ptr(dsave), _lib.ptr(gt), _lib.stream()), "dmnerf_mlp_bwd_data")
    jobs, n_jobs, outs, n_outs, part_floats = wgrad_plan(ins_num, M, g.device, split="f16x2" if f16 else split)
    part = torch.empty(part_floats, dtype=torch.float32, device=g.device)
    flat = None
    arena = arena_slot(model)                                    # data-parallel step: write into the shared all-reduce buffer
    if arena is not None and arena[0].flat.device == g.device:
        flat = arena[0].take(arena[1])
    if flat is None:
        flat = torch.empty(lib.dmnerf_param_count(ins_num), dtype=torch.float32, device=g.device)
    with _timed("mlp_bwd_weights", M):
        if f16:
            _lib.check(lib.dmnerf_mlp_bwd_weights_f16(_lib.ptr(ctx.save), _lib.ptr(dsave), _lib.ptr(gt), M, _lib.ptr(jobs), n_jobs, _lib.ptr(outs),
                                                               n_outs, _lib.ptr(ctx.flat), ins_num, _lib.ptr(part), _lib.ptr(flat), _lib.ptr(scale),
                                                               _lib.stream()), "dmnerf_mlp_bwd_weights_f16")
        else:
            f_wgrad = lib.dmnerf_mlp_bwd_weights_split if split else lib.dmnerf_mlp_bwd_weights
            _lib.check(f_wgrad(_lib.ptr(ctx.save), _lib.ptr(dsave), _lib.ptr(gt), M, _lib.ptr(jobs), n_jobs, _lib.ptr(outs), n_outs,
                               _lib.ptr(ctx.flat), ins_num, _lib.ptr(part), _lib.ptr(flat), _lib.stream()), "dmnerf_mlp_bwd_weights")
    ctx.save = None
    return tuple(split_flat_grads(model, flat))


class MLPEmbeddedFunction(torch.autograd.Function):
    """raw[M,4+C] = DM_NeRF(x) for pre-embedded rows x [M,90] (DM_NeRF.forward called directly, dm_nerf.py:80-106);
    parameters are inputs 2.. in state_dict order.  Gradients: parameters only."""

    @staticmethod
    def forward(ctx, model, x, *params):
        lib = _lib.load()
        M = x.shape[0]
        ins_num = model.ins_num
        raw = torch.empty(M, 4 + ins_num + 1, dtype=torch.float32, device=x.device)
        save = torch.empty(lib.dmnerf_train_save_floats(M), dtype=torch.float32, device=x.device)
        _lib.check(lib.dmnerf_mlp_fwd_embedded_train(_lib.ptr(model.blob()), ins_num, _lib.ptr(x), M, _lib.ptr(raw), _lib.ptr(save),
                                                     _lib.stream()), "dmnerf_mlp_fwd_embedded_train")
        ctx.model, ctx.M, ctx.save = model, M, save
        ctx.blob, ctx.blob_t, ctx.flat = model.blob(), model.blob_t(), model.flat()
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        return (None, None) + _mlp_backward(ctx, g_raw)


class CompositeFunction(torch.autograd.Function):
    """render_train (networks/render.py:6-28) with its analytic backward."""

    @staticmethod
    def forward(ctx, raw, z, rays_d):
        lib = _lib.load()
        N, S, ch = raw.shape
        C = ch - 4
        f = dict(dtype=torch.float32, device=raw.device)
        rgb, w = torch.empty(N, 3, **f), torch.empty(N, S, **f)
        depth, ins = torch.empty(N, **f), torch.empty(N, C - 1, **f)
        _lib.check(lib.dmnerf_composite_fwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays_d), N, S, C, _lib.ptr(rgb), _lib.ptr(w),
                                            _lib.ptr(depth), _lib.ptr(ins), _lib.stream()), "dmnerf_composite_fwd")
        ctx.save_for_backward(raw, z, rays_d, ins)
        ctx.set_materialize_grads(False)                 # unused outputs (weights, depth) arrive as None, not as zero-filled tensors
        return rgb, w, depth, ins

    @staticmethod
    def backward(ctx, g_rgb, g_w, g_depth, g_ins):
        lib = _lib.load()
        raw, z, rays_d, ins = ctx.saved_tensors
        N, S, ch = raw.shape
        C = ch - 4

        def prep(g, shape):
            if g is None:
                return None
            return _lib.f32(g.expand(shape) if g.shape != torch.Size(shape) else g)
        g_rgb = prep(g_rgb, (N, 3)) if g_rgb is not None else torch.zeros(N, 3, device=raw.device)
        g_ins = prep(g_ins, (N, C - 1)) if g_ins is not None else torch.zeros(N, C - 1, device=raw.device)
        g_w, g_depth = prep(g_w, (N, S)), prep(g_depth, (N,))
        d_raw = torch.empty_like(raw)
        _lib.check(lib.dmnerf_composite_bwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays_d), _lib.ptr(ins), _lib.ptr(g_rgb), _lib.ptr(g_ins),
                                            _lib.ptr(g_depth), _lib.ptr(g_w), N, S, C, _lib.ptr(d_raw), _lib.stream()), "dmnerf_composite_bwd")
        return d_raw, None, None


class CompositePenFunction(torch.autograd.Function):
    """render_train (networks/render.py:6-28) AND the emptiness penalizer's per-ray partial sums (networks/penalizer.py:5-42, with
    this level's own depth map as its detached depth argument -- what ins_penalizer is called with, train_dmsr.py:52-57) from ONE
    pass over the ray (extension).  Extra output ``part [N,4]`` (float64): what ``dmnerf_penalizer_fwd`` returns.
    Backward: the compositing gradient plus, when a gradient arrives for ``part`` (the penalizer's normalised loss was used), the
    penalizer's gradient w.r.t. ``raw[..., 4:]`` added in the same kernel.  The gradient of ``part`` is the same 4-vector for
    every ray (the loss divides batch sums); the consumers in this package hand it over as an expanded ``[4]`` tensor."""

    @staticmethod
    def forward(ctx, raw, z, rays_d, consts):
        lib = _lib.load()
        N, S, ch = raw.shape
        C = ch - 4
        f = dict(dtype=torch.float32, device=raw.device)
        rgb, w = torch.empty(N, 3, **f), torch.empty(N, S, **f)
        depth, ins = torch.empty(N, **f), torch.empty(N, C - 1, **f)
        part = torch.empty(N, 4, dtype=torch.float64, device=raw.device)
        tol, k2w, kh = consts
        _lib.check(lib.dmnerf_composite_pen_fwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays_d), N, S, C, tol, k2w, kh, _lib.ptr(rgb), _lib.ptr(w),
                                                _lib.ptr(depth), _lib.ptr(ins), _lib.ptr(part), _lib.stream()), "dmnerf_composite_pen_fwd")
        ctx.save_for_backward(raw, z, rays_d, ins, depth)
        ctx.consts = consts
        ctx.set_materialize_grads(False)                 # unused outputs (weights, depth, part) arrive as None: no zero fills
        return rgb, w, depth, ins, part

    @staticmethod
    def backward(ctx, g_rgb, g_w, g_depth, g_ins, g_part):
        lib = _lib.load()
        raw, z, rays_d, ins, depth = ctx.saved_tensors
        N, S, ch = raw.shape
        C = ch - 4

        def prep(g, shape):
            if g is None:
                return None
            return _lib.f32(g.expand(shape) if g.shape != torch.Size(shape) else g)
        g_rgb = prep(g_rgb, (N, 3)) if g_rgb is not None else torch.zeros(N, 3, device=raw.device)
        g_ins = prep(g_ins, (N, C - 1)) if g_ins is not None else torch.zeros(N, C - 1, device=raw.device)
        g_w, g_depth = prep(g_w, (N, S)), prep(g_depth, (N,))
        d_raw = torch.empty_like(raw)
        if g_part is None or N == 0:
            _lib.check(lib.dmnerf_composite_bwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays_d), _lib.ptr(ins), _lib.ptr(g_rgb), _lib.ptr(g_ins),
                                                _lib.ptr(g_depth), _lib.ptr(g_w), N, S, C, _lib.ptr(d_raw), _lib.stream()), "dmnerf_composite_bwd")
        else:
            row = g_part[0].to(torch.float64).contiguous()          # (a view of the producer's 4 doubles: no copy)
            tol, k2w, kh = ctx.consts
            _lib.check(lib.dmnerf_composite_pen_bwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(rays_d), _lib.ptr(ins), _lib.ptr(depth), _lib.ptr(g_rgb),
                                                    _lib.ptr(g_ins), _lib.ptr(g_depth), _lib.ptr(g_w), _lib.ptr(row), N, S, C, tol, k2w, kh,
                                                    _lib.ptr(d_raw), _lib.stream()), "dmnerf_composite_pen_bwd")
        return d_raw, None, None, None


def pen_consts(args):
    """(tolerance, 2 deta_w^2, 0.4 sqrt(2 pi)) as float32 values formed like the reference (penalizer.py:7-10), or None when the
    step does not penalise (``args.penalize`` false / tolerance or deta_w unset, train_dmsr.py:51)."""
    if not getattr(args, "penalize", False) or getattr(args, "tolerance", None) is None or getattr(args, "deta_w", None) is None:
        return None
    from .networks.penalizer import _consts
    k2w, kh = _consts(args.deta_w)
    return (float(args.tolerance), k2w, kh)


def pen_partials(depth, raw, z, rays_d, consts):
    """The penalizer partial sums the fused compositing pass left for this level, or None: looked up on the ``depth`` tensor of the
    dict (an OUTPUT of the compositing node, so holding them there creates no reference cycle through the saved ``raw``), and
    only handed out when the caller is asking for exactly what was computed -- same ``raw`` / ``z`` / ``rays_d`` storage and the
    same constants."""
    info = getattr(depth, "_dmn_pen", None)
    if info is None or consts is None:
        return None
    part, p_raw, p_z, p_d, p_consts = info
    if p_consts != tuple(consts) or p_raw != raw.data_ptr() or p_z != z.data_ptr() or p_d != rays_d.data_ptr():
        return None
    return part


MAX_TRAIN_SAMPLES = 1048576          # DMNERF_MAX_TRAIN_SAMPLES (include/dmnerf_hip.h)


def _params(model):
    return [p for _, p in model.named_parameters()]


_check_f16_call = [None]            # run_network_train(check_f16=): the caller's args.check_f16 for the launches of this call


def run_network_train(model, rays_o, rays_d, z, fused=False, split=None, check_f16=None):
    """Differentiable (w.r.t. the parameters) fused points + encoding + MLP.  Opt-in variants: ``fused`` (``args.fuse_heads``)
    runs the FORWARD on the fused-heads blob (-19 % MACs; values equal up to f32 re-association, not bit-equal to the inference
    default; default f32 backward); ``split`` (``weights.split_mode(args)``: "bf16x3" | "f16x2") runs forward, data gradients and
    weight gradients on the split-operand 16-bit MFMA kernels (fused heads + six bf16 / three f16 products per f32 product:
    f32-class values).  ``check_f16`` (``f16_check_enabled(args)``): scan what the f16x2 forward AND the
    backward of these launches convert for saturation (None: the module flag / DMNERF_CHECK_F16)."""
    mode = {"bf16x3": "split", "f16x2": "f16", True: "split"}[split] if split else ("fused" if fused else None)
    _check_f16_call[0] = None if check_f16 is None else bool(check_f16)
    try:
        return _run_network_train(model, rays_o, rays_d, z, mode)
    finally:
        _check_f16_call[0] = None


def _run_network_train(model, rays_o, rays_d, z, mode):
    if not model._fused_ok():                              # another network shape: layer by layer, its own autograd Function
        from . import generic
        return generic.run_network(model, rays_o, rays_d, z, train=True)
    rays_o, rays_d, z = _lib.f32(rays_o.reshape(-1, 3)), _lib.f32(rays_d.reshape(-1, 3)), _lib.f32(z)
    _lib.require_gpu(rays_o, rays_d, z)
    N, S = z.shape
    max_rays = max(1, MAX_TRAIN_SAMPLES // S)
    if N <= max_rays:
        return MLPRaysFunction.apply(model, mode, rays_o, rays_d, z, *_params(model))
    # a training launch addresses its saved-activation workspace with 32-bit byte offsets (DMNERF_MAX_TRAIN_SAMPLES in
    # include/dmnerf_hip.h): larger batches run as several launches, each with its own workspace; autograd adds the
    # parameter gradients of the pieces (rays are independent, so this is the same sum in a different order)
    params = _params(model)
    return torch.cat([MLPRaysFunction.apply(model, mode, rays_o[s:s + max_rays], rays_d[s:s + max_rays], z[s:s + max_rays], *params)
                      for s in range(0, N, max_rays)], 0)


def render_train_train(raw, z_vals, rays_d):
    raw, z, d = _lib.f32(raw), _lib.f32(z_vals.detach()), _lib.f32(rays_d.detach())
    _lib.require_gpu(raw, z, d)
    return CompositeFunction.apply(raw, z, d)


def mlp_forward_train(model, x):
    """Differentiable (w.r.t. the parameters) ``DM_NeRF.forward`` on pre-embedded rows ``[..., 90]`` -- what a caller
    that embeds the points itself gets in training mode (the reference's train loops go through ``dm_nerf``,
    train_dmsr.py:32; mesh / third-party code calls the model directly).  The kernels produce no gradient for ``x``
    itself: an input that requires grad is refused instead of silently getting none."""
    if x.requires_grad:
        raise NotImplementedError("dm_nerf_amd: DM_NeRF.forward gives gradients for the parameters only; detach the "
                                  "embedded input (none of the reference's callers differentiates w.r.t. it)")
    model._check_supported()
    x2 = _lib.f32(x.reshape(-1, x.shape[-1]))
    _lib.require_gpu(x2)
    if x2.shape[-1] != model.input_ch_pts + model.input_ch_views:
        raise ValueError(f"DM_NeRF.forward expects {model.input_ch_pts + model.input_ch_views} input channels")
    M = x2.shape[0]
    params = _params(model)
    if M <= MAX_TRAIN_SAMPLES:
        out = MLPEmbeddedFunction.apply(model, x2, *params)
    else:                                              # as run_network_train: several launches, autograd adds the gradients
        out = torch.cat([MLPEmbeddedFunction.apply(model, x2[s:s + MAX_TRAIN_SAMPLES], *params)
                         for s in range(0, M, MAX_TRAIN_SAMPLES)], 0)
    return out.reshape(*x.shape[:-1], out.shape[-1])


def dm_nerf_train(rays, model_coarse, model_fine, z_vals_coarse, args, t_rand=None, u=None):
    """Training-mode ``dm_nerf`` (networks/render.py:31-96): same dict, differentiable w.r.t. both models."""
    rays_o, rays_d = rays
    rays_o, rays_d = _lib.f32(rays_o.reshape(-1, 3)).detach(), _lib.f32(rays_d.reshape(-1, 3)).detach()
    z_in = _lib.f32(z_vals_coarse).detach()
    _lib.require_gpu(rays_o, rays_d, z_in)
    N, S = z_in.shape
    n_imp = int(args.N_importance)
    perturb = float(args.perturb)
    from .networks.render import check_draws             # RNG order of the reference: [N,S] then [N,n_imp]; shapes validated
    t_rand, u, _ = check_draws(t_rand, u, N, S, n_imp, perturb, z_in.device)
    z_coarse = helpers.stratify(z_in, t_rand) if t_rand is not None else z_in
    from . import weights
    fused, split = bool(getattr(args, "fuse_heads", False)), weights.split_mode(args)
    chk = f16_check_enabled(args) if split == "f16x2" else None     # args.check_f16 reaches the training-side scans too
    consts = pen_consts(args)                           # the step penalises: composite + penalizer partial sums in one pass

    def composite(raw, z):
        if consts is None:
            return CompositeFunction.apply(raw, z, rays_d)
        raw, z = _lib.f32(raw), _lib.f32(z)
        rgb, w, depth, ins, part = CompositePenFunction.apply(raw, z, rays_d, consts)
        depth._dmn_pen = (part, raw.data_ptr(), z.data_ptr(), rays_d.data_ptr(), consts)       # see pen_partials
        return rgb, w, depth, ins
    raw_coarse = run_network_train(model_coarse, rays_o, rays_d, z_coarse, fused, split, check_f16=chk)
    rgb_coarse, weights_coarse, depth_coarse, ins_coarse = composite(raw_coarse, z_coarse)
    with torch.no_grad():                              # z_samples.detach()  (render.py:68)
        if n_imp == 0:                                 # sample_pdf returns [N, 0]: the fine depths are the coarse ones
            z_fine = z_coarse.clone()
        else:
            z_fine = helpers.importance_resample(z_coarse, weights_coarse.detach(), n_imp, det=(perturb == 0.), u=u)
    raw_fine = run_network_train(model_fine, rays_o, rays_d, z_fine, fused, split, check_f16=chk)
    rgb_fine, weights_fine, depth_fine, ins_fine = composite(raw_fine, z_fine)
    if getattr(args, "is_train", False) and getattr(args, "N_ins", None) is not None:
        ins_fine = ins_fine[-args.N_ins:]
        ins_coarse = ins_coarse[-args.N_ins:]
    return {'rgb_fine': rgb_fine, 'ins_fine': ins_fine, 'z_vals_fine': z_fine, 'raw_fine': raw_fine,
            'raw_coarse': raw_coarse, 'rgb_coarse': rgb_coarse, 'ins_coarse': ins_coarse,
            'z_vals_coarse': z_coarse, 'depth_fine': depth_fine, 'depth_coarse': depth_coarse}
