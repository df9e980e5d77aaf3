"""The optimisation step as ONE HIP graph (extension; the reference's loop is eager, train_dmsr.py:24-64).

At the shipped batch sizes the MLP kernels take 20+ ms per step and the ~100 small launches of the rest of the step
(compositing, the three losses, autograd's elementwise kernels, Adam, the weight re-packing) disappear behind them; at the
per-rank shard of an 8-GPU strong split (384 rays: 2.4 ms of MLP kernels) they ARE the step (bench.py ``train_shard_proxy``).
The library never allocates, frees or synchronises (include/dmnerf_hip.h), so the whole step -- ``dm_nerf`` forward with saved
activations, img2mse + Hungarian-matched object-code loss + emptiness penalizer on both levels, every backward kernel, the
gradient all-reduce of a sharded step, the optimizer update and the re-packing of the weight blobs -- records into a HIP graph
and replays with one launch.

    opt = torch.optim.Adam(params, lr=torch.tensor(5e-4, device=dev), capturable=True)      # (or fused=True)
    gs = GraphedTrainStep((model_coarse, model_fine), opt, args, ins_num, rays, z, target, labels)
    for it in range(...):
        loss = gs.step(rays, z, target, labels)          # device tensor; no host synchronisation
        gs.set_lr(5e-4 * 0.1 ** (it / 500000))           # train_dmsr.py:68-72, in place

Same arithmetic as the eager step, launch for launch (tests/test_gpu_driver.py: bit-equal parameters after several steps).
Shapes are fixed at construction (a new batch size needs a new graph); the jitter draws come from the device generator, whose
state the graph advances on every replay like the eager calls would."""
import torch

from . import distributed as D


class GraphedTrainStep:
    def __init__(self, models, optimizer, args, ins_num, rays, z_vals, target, labels, warmup=2, step_fn=None):
        """Warms the step up eagerly (``warmup`` iterations on the given batch: code objects, cached plans, allocator), restores
        parameters and optimizer state to what they were, then captures one step.  ``optimizer`` must be capturable
        (``capturable=True`` or ``fused=True``) with a tensor learning rate if ``set_lr`` is to be used."""
        if not rays.is_cuda:
            raise RuntimeError("GraphedTrainStep needs GPU tensors")
        for grp in optimizer.param_groups:
            if not (grp.get("capturable", False) or grp.get("fused", False)):
                raise ValueError("GraphedTrainStep: build the optimizer with capturable=True (or fused=True)")
        own = hasattr(optimizer, "use_persistent_buffers")          # dm_nerf_amd.optim.FlatAdam: snapshot / restore / in-place re-pack
        self.models, self.opt, self.args, self.ins_num = models, optimizer, args, ins_num
        self.rays, self.z = rays.detach().clone(), z_vals.detach().clone()
        self.target, self.labels = target.detach().clone(), labels.detach().clone()
        self._step_fn = step_fn or (lambda: D.sharded_train_step(self.rays, self.z, self.target, self.labels, self.models,
                                                                 self.args, self.opt, self.ins_num)[0])
        params = [p for m in models for p in m.parameters()]
        saved_p = [p.detach().clone() for p in params]
        opt_params = [p for grp in optimizer.param_groups for p in grp["params"]]
        if own:
            saved_s = optimizer.snapshot()
        else:
            fresh = all(len(optimizer.state[p]) == 0 for p in opt_params)   # a new optimizer: its state starts at zero
            saved_s = None if fresh else self._snapshot_opt()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._step_fn()                                     # (the first one creates the optimizer state)
            # back to the state before the warm-up (in place: the graph records these addresses)
            with torch.no_grad():
                for p, s in zip(params, saved_p):
                    p.copy_(s)
            if own:
                optimizer.restore(saved_s)
                optimizer.use_persistent_buffers()                  # the captured forward reads what the captured re-pack writes
                optimizer.repack()
            else:
                self._restore_opt(saved_s)
                for m in models:
                    m.invalidate_blobs()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step_fn()
        torch.cuda.synchronize()

    def _snapshot_opt(self):
        return [{k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in self.opt.state[p].items()}
                for grp in self.opt.param_groups for p in grp["params"]]

    def _restore_opt(self, snap):
        """Optimizer state back to where it was before the warm-up, in place: the snapshot, or -- for an optimizer that had no
        state yet -- zero moments and a zero step counter (what Adam creates on its first step)."""
        i = 0
        with torch.no_grad():
            for grp in self.opt.param_groups:
                for p in grp["params"]:
                    for k, v in self.opt.state[p].items():
                        if torch.is_tensor(v):
                            v.zero_() if snap is None else v.copy_(snap[i][k])
                    i += 1

    def step(self, rays=None, z_vals=None, target=None, labels=None):
        """Replay on a new batch (copied into the graph's input buffers on the current stream; ``None`` keeps the previous
        contents).  Returns the loss as a device tensor that the next ``step`` overwrites."""
        for dst, src in ((self.rays, rays), (self.z, z_vals), (self.target, target), (self.labels, labels)):
            if src is not None:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # The replay updated the parameters on the device without touching their Python-side ``_version``, which the models'
        # kernel-layout caches are keyed on (networks/dm_nerf.py ``blob``): forget them, so that an eager render between replays
        # (the periodic test render of train_dmsr.py:88-100) re-packs from the CURRENT parameters instead of reusing the copy
        # an earlier evaluation packed.  (The graph itself is unaffected: it re-packs into its own buffers on every replay.)
        if hasattr(self.opt, "use_persistent_buffers"):
            for m, out, size, (nb, _) in zip(self.opt.models, self.opt._persist, self.opt.sizes, self.opt._n_blob):
                m.install_packed(out[:size], out[size:size + nb], out[size + nb:])      # what this replay's re-pack just wrote
        else:
            for m in self.models:
                m.invalidate_blobs()
        return self.loss

    def set_lr(self, lr):
        if hasattr(self.opt, "set_lr"):
            return self.opt.set_lr(lr)
        for grp in self.opt.param_groups:
            if torch.is_tensor(grp["lr"]):
                grp["lr"].fill_(float(lr))
            else:
                raise ValueError("GraphedTrainStep.set_lr: the optimizer was not built with a tensor lr")
