"""Mirror of the loss side of ``networks/evaluator.py`` that sits in the training step of the reference
(train_dmsr.py:33-47): ``img2mse``, ``mse2psnr`` and the Hungarian-matched object-code loss ``ins_criterion``.

``ins_criterion`` runs entirely on the GPU stream (csrc/criterion.hip): the reference moves the cost matrix to the
host for ``scipy.optimize.linear_sum_assignment`` and syncs twice per step.  The metrics half of
the file (``calculate_ap``, ``ins_eval``) is evaluation tooling and stays with the reference, except the per-pixel
label / confidence every rendered frame needs (``ins_label_conf``), which the frame driver computes on the device.
"""
import torch

from .. import _lib

img2mse = lambda x, y: torch.mean((x - y) ** 2)                                             # evaluator.py:11
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], device=x.device))  # evaluator.py:15


class _InsCriterion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, labels, ins_num):
        lib = _lib.load()
        N = pred.shape[0]
        nbytes = lib.dmnerf_ins_criterion_work_bytes(N, ins_num)
        if nbytes < 0:
            raise ValueError(f"ins_criterion: unsupported N={N} ins_num={ins_num} (ins_num <= 128)")
        work = torch.empty(nbytes, dtype=torch.uint8, device=pred.device)
        out = torch.empty(4, dtype=torch.float32, device=pred.device)
        _lib.check(lib.dmnerf_ins_criterion_fwd(_lib.ptr(pred), _lib.ptr(labels), N, ins_num, _lib.ptr(work), nbytes, _lib.ptr(out),
                                                _lib.stream()), "dmnerf_ins_criterion_fwd")
        ctx.save_for_backward(pred, labels, work)
        ctx.ins_num = ins_num
        ctx.mark_non_differentiable(work)
        ctx.set_materialize_grads(False)                 # (no zero-filled "gradient" for the work buffer)
        return out, work

    @staticmethod
    def backward(ctx, g_out, _g_work=None):
        if g_out is None:
            return None, None, None
        pred, labels, work = ctx.saved_tensors
        grad = torch.empty_like(pred)
        g = _lib.f32(g_out)
        _lib.check(_lib.load().dmnerf_ins_criterion_bwd(_lib.ptr(pred), _lib.ptr(labels), pred.shape[0], ctx.ins_num, _lib.ptr(work),
                                                        _lib.ptr(g), _lib.ptr(grad), _lib.stream()), "dmnerf_ins_criterion_bwd")
        return grad, None, None


CRIT_TOO_MANY_LABELS, CRIT_LABEL_RANGE = 1, 2          # DMNERF_CRIT_* (include/dmnerf_hip.h)


def ins_criterion(pred_ins, gt_labels, ins_num, check=None):
    """``ins_criterion`` (networks/evaluator.py:19-37): ``pred_ins [N, ins_num]``, ``gt_labels [N]`` ->
    ``(ins_loss_sum, valid_ce, invalid_ce, valid_siou)`` as 0-dim tensors, differentiable w.r.t. ``pred_ins``.

    Same definition as the reference: rows of the cost matrices are the labels that occur (ascending), matched to
    channels by a minimum-cost assignment of ``cost_ce + cost_siou``; ``invalid_ce`` is the mean prediction of the
    unmatched channels (0 when every channel is matched, where the reference returns ``tensor([0])``).
    No host synchronisation -- which is also why two conditions on which the reference RAISES cannot raise here by
    default: more distinct labels than ``ins_num`` channels (the first ``ins_num`` are kept) and labels outside
    ``[0, ins_num]`` (they join no row).  The kernels record both in a flags word; ``check=True`` (or the environment
    variable ``DMNERF_CHECK_LABELS=1``; default off) reads it back -- one sync -- and raises ``ValueError`` like the reference.
    """
    pred = _lib.f32(pred_ins)
    _lib.require_gpu(pred)
    if pred.dim() != 2 or pred.shape[1] != int(ins_num):
        raise ValueError("ins_criterion: pred_ins must be [N, ins_num]")
    labels = gt_labels.reshape(-1).to(device=pred.device, dtype=torch.int32).contiguous()
    if labels.shape[0] != pred.shape[0]:
        raise ValueError("ins_criterion: one label per ray")
    out, work = _InsCriterion.apply(pred, labels, int(ins_num))
    if check is None:
        import os
        check = os.environ.get("DMNERF_CHECK_LABELS", "0") == "1"
    if check:
        off = _lib.load().dmnerf_ins_criterion_flags_offset(pred.shape[0], int(ins_num))
        flags = int(work[off:off + 4].view(torch.int32).item())
        if flags & CRIT_TOO_MANY_LABELS:
            raise ValueError(f"ins_criterion: more than ins_num={ins_num} distinct labels in the batch (evaluator.py:21-25 raises too)")
        if flags & CRIT_LABEL_RANGE:
            raise ValueError(f"ins_criterion: a label lies outside [0, {ins_num}]")
    return out[0], out[1], out[2], out[3]


def ins_label_conf(pred_ins):
    """The first two lines of ``ins_eval`` (networks/evaluator.py:127-137): ``pred_label = argmax(pred_ins, -1)`` (int64,
    first maximum) and ``pred_conf_mask = max(pred_ins, -1)`` for ``pred_ins [..., ins_num]``, on the device."""
    x = _lib.f32(pred_ins)
    _lib.require_gpu(x)
    C = x.shape[-1]
    flat = x.reshape(-1, C)
    label = torch.empty(flat.shape[0], dtype=torch.int64, device=x.device)
    conf = torch.empty(flat.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().dmnerf_ins_label_conf(_lib.ptr(flat), flat.shape[0], C, _lib.ptr(label), _lib.ptr(conf), _lib.stream()),
               "dmnerf_ins_label_conf")
    return label.reshape(x.shape[:-1]), conf.reshape(x.shape[:-1])
