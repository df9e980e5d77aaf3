"""The loss tail of the optimisation step -- train_dmsr.py:33-61 -- as ONE autograd node (extension; the drop-in functions
``evaluator.img2mse`` / ``evaluator.ins_criterion`` / ``penalizer.ins_penalizer`` stay what the reference's scripts import).

    total, terms = train_losses(out, rays_d, target, labels, ins_num, args)
    total.backward()

``out`` is the dict ``dm_nerf`` returns in training mode; ``total`` is the sum the training loop forms,

    sum over (fine, coarse) of  img2mse(rgb, target) + ins_criterion(ins, labels)[0] [+ ins_penalizer(raw, z, depth, rays_d)]

and ``terms`` its six addends (``mse, criterion, penalizer`` of the fine level, then of the coarse one; detached).  Same
kernels as the drop-in functions for the object-code loss (csrc/criterion.hip, both levels per launch) and the penalizer's per-ray
sums and gradient (csrc/render_kernels.hip); the scalar arithmetic around them -- the two mean squared errors, the penalizer's
normalisation, the additions, and in the backward the upstream factors -- is two kernels (csrc/losses.hip) instead of some forty
elementwise / reduction launches of a few microseconds each.  At the 384-ray shard of an 8-way data-parallel step that tail was a
quarter of the step (profiles/r04).  Values: the criterion and penalizer terms and every gradient are bit-equal to the drop-in
functions' (same kernels, same factors); the squared-error VALUE is summed in double here (ATen: f32 tree), gradients identical.

Ray-sharded steps (``gathered=``): the squared error and the criterion are batch-global, evaluated on the all-gathered
``rgb`` / ``ins`` of both levels; the penalizer's four sums per level are all-reduced before its normalisation -- one collective
for both levels (distributed.sharded_train_step)."""
import torch

from . import _lib
from .networks.penalizer import _consts


class _TrainLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_f, rgb_c, ins_f, ins_c, raw_f, raw_c, part_f, part_c, z_f, z_c, depth_f, depth_c, rays_d, target, labels, cfg):
        lib = _lib.load()
        dev = rgb_f.device
        N = rgb_f.shape[0]
        ins_num, penalize, tol, deta_w, sharded = cfg
        C = ins_num + 1
        f32 = dict(dtype=torch.float32, device=dev)
        # object-code loss of both levels: 3 launches
        Nc = ins_f.shape[0]
        nbytes = lib.dmnerf_ins_criterion_work_bytes(Nc, ins_num)
        if nbytes < 0:
            raise ValueError(f"train_losses: unsupported N={Nc} ins_num={ins_num} (ins_num <= 128)")
        work = torch.empty(2, nbytes, dtype=torch.uint8, device=dev)
        crit = torch.empty(2, 4, **f32)
        _lib.check(lib.dmnerf_ins_criterion_fwd2(_lib.ptr(ins_f), _lib.ptr(ins_c), _lib.ptr(labels), Nc, ins_num, _lib.ptr(work[0]),
                                                 _lib.ptr(work[1]), nbytes, _lib.ptr(crit[0]), _lib.ptr(crit[1]), _lib.stream()),
                   "dmnerf_ins_criterion_fwd2")
        # emptiness penalizer: per-ray partial sums of each level, the four batch sums of both in one launch
        sums = None
        consts = None
        if penalize:
            k2w, kh = _consts(deta_w)
            consts = (float(tol), k2w, kh)
            if part_f is not None:                             # the fused compositing pass already formed the per-ray sums
                parts = [part_f, part_c]
            else:
                parts = []
                for raw, z, depth in ((raw_f, z_f, depth_f), (raw_c, z_c, depth_c)):
                    n, S, _ = raw.shape
                    part = torch.empty(n, 4, dtype=torch.float64, device=dev)
                    _lib.check(lib.dmnerf_penalizer_fwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(depth), _lib.ptr(rays_d), n, S, C,
                                                        consts[0], k2w, kh, _lib.ptr(part), _lib.stream()), "dmnerf_penalizer_fwd")
                    parts.append(part)
            sums = torch.empty(2, 4, dtype=torch.float64, device=dev)
            _lib.check(lib.dmnerf_penalizer_sums2(_lib.ptr(parts[0]), parts[0].shape[0], _lib.ptr(parts[1]), parts[1].shape[0],
                                                  _lib.ptr(sums), _lib.stream()), "dmnerf_penalizer_sums2")
            if sharded:                                        # the sums are batch-global: ONE all-reduce for both levels
                from . import distributed
                distributed.allreduce_sums(sums)
        terms = torch.empty(8, **f32)
        inv = torch.empty(4, **f32)
        _lib.check(lib.dmnerf_loss_tail_fwd(_lib.ptr(rgb_f), _lib.ptr(rgb_c), _lib.ptr(target), N, _lib.ptr(crit[0]), _lib.ptr(crit[1]),
                                            _lib.ptr(None if sums is None else sums[0]), _lib.ptr(None if sums is None else sums[1]),
                                            C, _lib.ptr(terms), _lib.ptr(inv), _lib.stream()), "dmnerf_loss_tail_fwd")
        keep_raw = penalize and part_f is None                  # (the per-ray backward kernels of this node need raw; the fused form does not)
        ctx.save_for_backward(rgb_f, rgb_c, ins_f, ins_c, raw_f if keep_raw else None, raw_c if keep_raw else None, z_f, z_c, depth_f, depth_c,
                              rays_d, target, labels, work, inv)
        ctx.cfg, ctx.consts = cfg, consts
        ctx.from_parts = (part_f.shape[0], part_c.shape[0]) if (penalize and part_f is not None) else None
        total = terms[6]
        out_terms = terms[:6]
        ctx.mark_non_differentiable(out_terms, work)
        ctx.set_materialize_grads(False)                        # (no zero-filled "gradients" for the two non-differentiable outputs)
        return total, out_terms, work

    @staticmethod
    def backward(ctx, g_total, _g_terms=None, _g_work=None):
        lib = _lib.load()
        rgb_f, rgb_c, ins_f, ins_c, raw_f, raw_c, z_f, z_c, depth_f, depth_c, rays_d, target, labels, work, inv = ctx.saved_tensors
        ins_num, penalize, tol, deta_w, sharded = ctx.cfg
        C = ins_num + 1
        dev = rgb_f.device
        N = rgb_f.shape[0]
        if g_total is None:                                     # the total was not used
            return (None,) * 16
        g = _lib.f32(g_total.reshape(1))
        d_rgb_f, d_rgb_c = torch.empty_like(rgb_f), torch.empty_like(rgb_c)
        gout = torch.empty(8, dtype=torch.float32, device=dev)
        scales = torch.empty(4, dtype=torch.float32, device=dev)
        g_part = torch.empty(2, 4, dtype=torch.float64, device=dev) if ctx.from_parts is not None else None
        _lib.check(lib.dmnerf_loss_tail_bwd(_lib.ptr(rgb_f), _lib.ptr(rgb_c), _lib.ptr(target), N, _lib.ptr(g), _lib.ptr(inv), _lib.ptr(d_rgb_f),
                                            _lib.ptr(d_rgb_c), _lib.ptr(gout), _lib.ptr(scales), _lib.ptr(g_part), _lib.stream()), "dmnerf_loss_tail_bwd")
        d_ins_f, d_ins_c = torch.empty_like(ins_f), torch.empty_like(ins_c)
        _lib.check(lib.dmnerf_ins_criterion_bwd2(_lib.ptr(ins_f), _lib.ptr(ins_c), _lib.ptr(labels), ins_f.shape[0], ins_num, _lib.ptr(work[0]),
                                                 _lib.ptr(work[1]), _lib.ptr(gout[:4]), _lib.ptr(gout[4:]), _lib.ptr(d_ins_f), _lib.ptr(d_ins_c),
                                                 _lib.stream()), "dmnerf_ins_criterion_bwd2")
        d_raw_f = d_raw_c = d_part_f = d_part_c = None
        if penalize and ctx.from_parts is not None:
            # d raw is formed inside the compositing node's backward kernel from the gradient of its partial sums
            d_part_f, d_part_c = g_part[0].expand(ctx.from_parts[0], 4), g_part[1].expand(ctx.from_parts[1], 4)
        elif penalize:
            tolf, k2w, kh = ctx.consts
            grads = []
            for lvl, (raw, z, depth) in enumerate(((raw_f, z_f, depth_f), (raw_c, z_c, depth_c))):
                n, S, _ = raw.shape
                d_raw = torch.empty_like(raw)
                _lib.check(lib.dmnerf_penalizer_bwd(_lib.ptr(raw), _lib.ptr(z), _lib.ptr(depth), _lib.ptr(rays_d), n, S, C, tolf, k2w, kh,
                                                    _lib.ptr(scales[2 * lvl:2 * lvl + 2]), _lib.ptr(d_raw), _lib.stream()), "dmnerf_penalizer_bwd")
                grads.append(d_raw)
            d_raw_f, d_raw_c = grads
        return (d_rgb_f, d_rgb_c, d_ins_f, d_ins_c, d_raw_f, d_raw_c, d_part_f, d_part_c) + (None,) * 8


def train_losses(out, rays_d, target, labels, ins_num, args, rgb_ins=None, sharded=False, check=None):
    """-> ``(total, terms[6])``, see the module docstring.  ``rgb_ins``: ``(rgb_fine, rgb_coarse, ins_fine, ins_coarse)`` to use
    instead of ``out``'s (the all-gathered batch of a ray-sharded step; ``args.N_ins`` already applied by the caller or not at
    all).  ``args.penalize`` (train_dmsr.py:51) switches the emptiness term; ``check`` as in ``evaluator.ins_criterion``: read the
    label-condition flags of both levels back (one sync) and raise like the reference."""
    if rgb_ins is None:
        rgb_ins = (out['rgb_fine'], out['rgb_coarse'], out['ins_fine'], out['ins_coarse'])
    rgb_f, rgb_c, ins_f, ins_c = (_lib.f32(t) for t in rgb_ins)
    penalize = bool(getattr(args, "penalize", False))
    target = _lib.f32(target.detach())
    lab = labels.reshape(-1).to(device=rgb_f.device, dtype=torch.int32).contiguous()
    if lab.shape[0] != ins_f.shape[0] or ins_c.shape != ins_f.shape:
        raise ValueError("train_losses: one label per row of ins_fine / ins_coarse")
    if target.shape != rgb_f.shape or rgb_c.shape != rgb_f.shape:
        raise ValueError("train_losses: target must match rgb_fine / rgb_coarse")
    if ins_f.shape[1] != ins_num:
        raise ValueError("train_losses: ins_* must be [N, ins_num]")
    raw_f, raw_c = _lib.f32(out['raw_fine']), _lib.f32(out['raw_coarse'])
    det = lambda t: _lib.f32(t.detach())
    z_f, z_c = det(out['z_vals_fine']), det(out['z_vals_coarse'])
    depth_f, depth_c = det(out['depth_fine']).reshape(-1), det(out['depth_coarse']).reshape(-1)
    rays_d = det(rays_d)
    _lib.require_gpu(rgb_f, rgb_c, ins_f, ins_c, raw_f, raw_c, z_f, z_c, depth_f, depth_c, rays_d, target, lab)
    cfg = (int(ins_num), penalize, getattr(args, "tolerance", None), getattr(args, "deta_w", None), bool(sharded))
    part_f = part_c = None
    if penalize:                                            # per-ray sums left by the fused compositing pass (both levels or neither)
        from . import autograd
        consts = autograd.pen_consts(args)
        part_f = autograd.pen_partials(out['depth_fine'], raw_f, z_f, rays_d, consts)
        part_c = autograd.pen_partials(out['depth_coarse'], raw_c, z_c, rays_d, consts)
        if part_f is None or part_c is None:
            part_f = part_c = None
    total, terms, work = _TrainLosses.apply(rgb_f, rgb_c, ins_f, ins_c, raw_f, raw_c, part_f, part_c, z_f, z_c, depth_f, depth_c, rays_d, target,
                                            lab, cfg)
    if check is None:
        import os
        check = os.environ.get("DMNERF_CHECK_LABELS", "0") == "1"
    if check:
        from .networks import evaluator as E
        off = _lib.load().dmnerf_ins_criterion_flags_offset(ins_f.shape[0], int(ins_num))
        flags = int(work[0, off:off + 4].view(torch.int32).item()) | int(work[1, off:off + 4].view(torch.int32).item())
        if flags & E.CRIT_TOO_MANY_LABELS:
            raise ValueError(f"train_losses: more than ins_num={ins_num} distinct labels in the batch (evaluator.py:21-25 raises too)")
        if flags & E.CRIT_LABEL_RANGE:
            raise ValueError(f"train_losses: a label lies outside [0, {ins_num}]")
    return total, terms
