"""``FlatAdam`` -- torch.optim.Adam for the two DM_NeRF models as TWO launches on flat buffers (extension; opt-in).

The reference builds ``torch.optim.Adam(list(model_coarse.parameters()) + list(model_fine.parameters()), lr=lrate, betas=(0.9,
0.999))`` (train_dmsr.py:124-125) and calls ``optimizer.zero_grad(); total_loss.backward(); optimizer.step()`` (:62-64), then
rewrites ``param_group['lr']`` (:68-72).  That object works unchanged on this package's models and is what ``bench.py``'s
``train`` legs time.  ``FlatAdam`` takes its place when a caller opts in:

* the parameters of all models are re-pointed at views of ONE flat f32 vector (same values, same ``nn.Parameter`` objects, same
  ``state_dict``), the moments are two more flat vectors, the gradients are the ``GradArena`` the weight-gradient kernels write
  into (autograd installs its views as ``p.grad``: no copy);
* ``step()`` = ``dmnerf_adam_step`` (one pass over the four vectors: the update of ``torch.optim.Adam`` operation for operation,
  csrc/optim.hip) + ``dmnerf_repack_train`` (one launch: for every model a fresh copy of its flat parameters, its forward blob and
  its W^T blob incl. the head product -- what ``DM_NeRF.flat() / blob() / blob_t()`` would otherwise rebuild with 8 launches on
  the next forward), installed into the models' caches;
* no host synchronisation; with ``capturable=True`` the learning rate lives on the device (``set_lr`` / writing
  ``param_groups[0]['lr']`` before ``step``), so ``GraphedTrainStep`` captures it like a capturable torch optimizer;
* ``state_dict()`` / ``load_state_dict()`` speak ``torch.optim.Adam``'s format (per-parameter ``step / exp_avg / exp_avg_sq``), so
  the reference's checkpoints (train_dmsr.py:78-86) carry over in both directions.

Against ``torch.optim.Adam`` (tests/test_gpu_optim.py): parameters within 1 ulp after 20 steps on identical gradients (every
operation is the same f32 operation; only the fused multiply-adds a compiler may or may not form inside ATen's kernels differ)."""
import ctypes

import torch

from . import _lib, autograd, weights


_f_pos_cache = {}


def _head_positions(ins_num, n_param, device):
    """Where the W^T blob keeps each element of the head product F (the inverse of the transposed pack index on its entries that
    address the F block behind the parameters): int32 [32768] on ``device``, or None if an element does not occur exactly once."""
    key = (int(ins_num), str(device))
    if key not in _f_pos_cache:
        import numpy as np
        idx = weights.pack_index_t_host(ins_num)
        where = np.nonzero(idx >= n_param)[0]
        src = idx[where] - n_param
        pos = None
        if len(where) == weights.HEAD_F_FLOATS and len(np.unique(src)) == weights.HEAD_F_FLOATS:
            inv = np.empty(weights.HEAD_F_FLOATS, dtype=np.int32)
            inv[src] = where.astype(np.int32)
            pos = torch.from_numpy(inv).to(device)
        _f_pos_cache[key] = pos
    return _f_pos_cache[key]


class FlatAdam:
    wants_arena = True           # distributed.sharded_train_step: let the backward write into the gradient arena at world 1 too

    def __init__(self, models, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        self.models = list(models)
        if not 1 <= len(self.models) <= 4:
            raise ValueError("FlatAdam: 1..4 models")
        for m in self.models:
            m._check_supported()
        lib = _lib.load()
        dev = next(self.models[0].parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("dm_nerf_amd runs on MI355X only: move the models to the GPU before building FlatAdam")
        self.sizes = [int(lib.dmnerf_param_count(m.ins_num)) for m in self.models]
        n = sum(self.sizes)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.params, self._names = [], []
        o = 0
        with torch.no_grad():
            for mi, (m, size) in enumerate(zip(self.models, self.sizes)):
                o_m = o
                for pname, p in m.named_parameters():
                    self._names.append(f"models[{mi}].{pname}")
                    if p.dtype != torch.float32 or p.device != dev:
                        raise RuntimeError("FlatAdam: f32 parameters on one GPU")
                    view = self.flat[o:o + p.numel()].view_as(p)
                    view.copy_(p)
                    p.data = view                                    # same Parameter object, storage = the flat vector
                    self.params.append(p)
                    o += p.numel()
                assert o - o_m == size
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.state2 = torch.zeros(2, dtype=torch.int64, device=dev)   # [step count, ticket]
        self.arena = autograd.grad_arena(self.models)
        self.capturable = bool(capturable)
        self.defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), capturable=self.capturable)
        self.lr_t = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self.param_groups = [dict(self.defaults, lr=(self.lr_t if self.capturable else float(lr)), params=self.params,
                                  weight_decay=0, amsgrad=False, maximize=False, fused=None, foreach=None, differentiable=False)]
        self._idx = [(weights.pack_index(m.ins_num, dev, False), weights.pack_index(m.ins_num, dev, True)) for m in self.models]
        self._n_blob = [(int(lib.dmnerf_blob_floats(m.ins_num)), int(lib.dmnerf_blob_t_floats(m.ins_num))) for m in self.models]
        self._f_pos = [_head_positions(m.ins_num, size, dev) for m, size in zip(self.models, self.sizes)]
        self._persist = None
        self.repack()                                                # the models' caches now describe the flat storage

    # -- the torch.optim.Optimizer surface the training loops use --------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        if set_to_none:
            for p in self.params:
                p.grad = None
            self.arena.begin_step()                                  # the next backward writes the arena slots in place
            return
        # Keep the gradient tensors and zero them: the next backward ACCUMULATES into them.  They are (normally) views of the arena,
        # so the arena must NOT hand its slots out again -- a backward writing a slot in place would be added onto itself by
        # AccumulateGrad (2 g) -- the launches use scratch vectors instead and autograd adds those into the zeroed views.
        have = False
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()
                have = True
        if have:
            self.arena.hold()
        else:
            self.arena.begin_step()

    def set_lr(self, lr):
        if self.capturable:
            self.lr_t.fill_(float(lr))
        else:
            self.param_groups[0]["lr"] = float(lr)

    def _grads(self):
        """The flat gradient vector: the arena when autograd installed its views (the normal case), else gathered into it.
        EVERY parameter must have a gradient: ``torch.optim.Adam`` skips a parameter whose ``.grad`` is None (no moment update, its
        own step count), which one pass over flat vectors with one step counter cannot reproduce -- so that case raises instead of
        silently updating with a zero or a stale gradient."""
        a = self.arena
        base = a.flat.data_ptr()
        if a.resident():                                             # ALL views checked (a frozen / replaced middle tensor must not
            return a.flat                                            # pass for the arena: ADVICE r05), ~60 pointer compares per step
        o = 0
        with torch.no_grad():
            for name_p, p in zip(self._names, self.params):          # gradients that came another way (accumulated, user-made)
                n = p.numel()
                if p.grad is None:
                    raise RuntimeError(f"FlatAdam.step: parameter {name_p} has no gradient.  torch.optim.Adam would skip it (and keep "
                                       "a step count of its own); FlatAdam updates all parameters in one pass -- use torch.optim.Adam "
                                       "for steps that differentiate only some of them")
                if p.grad.data_ptr() != base + 4 * o:
                    a.flat[o:o + n].copy_(p.grad.reshape(-1))
                o += n
        return a.flat

    def step(self):
        lib = _lib.load()
        autograd.join_side()                                         # a side-stream backward left in flight (join_on_exit=False) first
        g = self._grads()
        grp = self.param_groups[0]
        lr = grp["lr"]
        d_lr = None
        if torch.is_tensor(lr):
            if lr is not self.lr_t:
                self.lr_t.copy_(lr)
            d_lr, lr = self.lr_t, 0.0
        b1, b2 = grp["betas"]
        _lib.check(lib.dmnerf_adam_step(_lib.ptr(self.flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.flat.numel(),
                                        float(lr), _lib.ptr(d_lr), float(b1), float(b2), float(grp["eps"]), _lib.ptr(self.state2), _lib.stream()),
                   "dmnerf_adam_step")
        self.repack()

    def repack(self):
        """One launch: every model's flat copy, forward blob and W^T blob from the (updated) flat parameters; installed as the
        models' cached kernel-layout weights (NEW tensors each time: a backward pending across the step keeps what its forward used)."""
        lib = _lib.load()
        n_models = len(self.models)
        arr = (_lib.RepackModel * n_models)()
        keep = []
        o = 0
        for i, (m, size) in enumerate(zip(self.models, self.sizes)):
            n_blob, n_blob_t = self._n_blob[i]
            out = self._persist[i] if self._persist is not None else \
                torch.empty(size + n_blob + n_blob_t, dtype=torch.float32, device=self.flat.device)
            flat_copy, blob, blob_t = out[:size], out[size:size + n_blob], out[size + n_blob:]
            src = self.flat[o:o + size]
            arr[i] = _lib.RepackModel(src.data_ptr(), int(m.ins_num), flat_copy.data_ptr(), self._idx[i][0].data_ptr(), blob.data_ptr(),
                                      self._idx[i][1].data_ptr(), blob_t.data_ptr(), None if self._f_pos[i] is None else self._f_pos[i].data_ptr())
            keep.append((m, flat_copy, blob, blob_t))
            o += size
        _lib.check(lib.dmnerf_repack_train(arr, n_models, _lib.stream()), "dmnerf_repack_train")
        for m, flat_copy, blob, blob_t in keep:
            m.install_packed(flat_copy, blob, blob_t)

    def use_persistent_buffers(self, on=True):
        """Re-pack IN PLACE into buffers allocated once (``GraphedTrainStep``: the captured forward of replay k + 1 must read what
        the captured re-pack of replay k wrote, so the addresses cannot change) instead of into fresh tensors per step."""
        if on and self._persist is None:
            self._persist = [torch.empty(size + nb + nbt, dtype=torch.float32, device=self.flat.device)
                             for size, (nb, nbt) in zip(self.sizes, self._n_blob)]
        elif not on:
            self._persist = None

    # -- checkpoints in torch.optim.Adam's format (train_dmsr.py:78-86) ----------------------------------------------------
    def state_dict(self):
        step = self.state2[0].to(torch.float32).cpu()                # (one host synchronisation: checkpoints only)
        state, o = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            state[i] = {"step": step.clone(), "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
            o += n
        grp = {k: (float(v) if torch.is_tensor(v) else v) for k, v in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(len(self.params)))
        return {"state": state if int(step) > 0 else {}, "param_groups": [grp]}

    def load_state_dict(self, sd):
        grp = sd["param_groups"][0]
        self.param_groups[0].update({k: v for k, v in grp.items() if k in ("betas", "eps")})
        self.set_lr(float(grp["lr"]))
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.state2.zero_()
        o, steps = 0, set()
        with torch.no_grad():
            for i, p in enumerate(self.params):
                n = p.numel()
                st = sd["state"].get(i)
                if st is not None:
                    self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(int(st["step"]))
                o += n
        if len(steps) > 1:
            raise ValueError("FlatAdam.load_state_dict: the parameters carry different step counts")
        if steps:
            self.state2[0] = steps.pop()

    # -- GraphedTrainStep's warm-up protocol -------------------------------------------------------------------------------
    def snapshot(self):
        return (self.exp_avg.clone(), self.exp_avg_sq.clone(), self.state2.clone())

    def restore(self, snap):
        with torch.no_grad():
            self.exp_avg.copy_(snap[0]); self.exp_avg_sq.copy_(snap[1]); self.state2.copy_(snap[2])
