"""Drop-in for the reference's ``networks/dm_nerf.py``: Embedder, get_embedder, DM_NeRF.

Same constructor arguments, parameter names and state_dict keys as the reference
(networks/dm_nerf.py:8-106), so checkpoints (train_dmsr.py:78-86) and torch.optim.Adam work
unchanged.  The arithmetic runs in libdmnerf_hip.so.
"""
import torch
import torch.nn as nn

from .. import _lib, weights


class Embedder:
    """``Embedder`` (networks/dm_nerf.py:8-38).  Only the configuration ``get_embedder`` builds
    (include_input, log sampling, sin/cos, 3 input dims) exists in the reference's configs."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        if not (kwargs.get('include_input', True) and kwargs.get('log_sampling', True)
                and kwargs.get('input_dims', 3) == 3):
            raise NotImplementedError("Embedder: only include_input=True, log_sampling=True, input_dims=3")
        self.num_freqs = int(kwargs['num_freqs'])
        if int(kwargs['max_freq_log2']) != self.num_freqs - 1:
            raise NotImplementedError("Embedder: max_freq_log2 must equal num_freqs - 1 (powers of two)")
        self.out_dim = 3 + 2 * 3 * self.num_freqs

    def embed(self, inputs):
        """``cat([x, sin(x f0), cos(x f0), ...], -1)`` (networks/dm_nerf.py:37-38)."""
        x = _lib.f32(inputs)
        _lib.require_gpu(x)
        if x.shape[-1] != 3:
            raise ValueError("Embedder.embed expects [..., 3]")
        M = x.numel() // 3
        out = torch.empty(*x.shape[:-1], self.out_dim, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().dmnerf_embed(_lib.ptr(x), M, self.num_freqs, _lib.ptr(out), _lib.stream()), "dmnerf_embed")
        return out


def get_embedder(multires, i=0):
    """``get_embedder`` (networks/dm_nerf.py:41-55)."""
    if i == -1:
        return nn.Identity(), 3
    embed_kwargs = {
        'include_input': True,
        'input_dims': 3,
        'max_freq_log2': multires - 1,
        'num_freqs': multires,
        'log_sampling': True,
        'periodic_fns': [torch.sin, torch.cos],
    }
    embedder = Embedder(**embed_kwargs)
    return embedder, embedder.out_dim


class DM_NeRF(nn.Module):
    """``DM_NeRF`` (networks/dm_nerf.py:58-106): 8x256 trunk, skip at 4, sigma / rgb / object heads.

    Parameters are ordinary ``nn.Linear`` modules with the reference's names; the fused HIP
    kernel reads a permuted copy ("blob") that is rebuilt whenever a parameter changed.
    """

    def __init__(self, D=8, W=256, input_ch_pts=3, input_ch_views=3, skips=[4], ins_num=None):
        super().__init__()
        self.D, self.W = D, W
        self.skips = skips
        self.input_ch_pts = input_ch_pts
        self.input_ch_views = input_ch_views
        self.ins_num = ins_num
        self.mlps = nn.ModuleList(
            [nn.Linear(input_ch_pts, W)] +
            [nn.Linear(W, W) if i not in skips else nn.Linear(W + input_ch_pts, W) for i in range(D - 1)])
        self.rgb_feature_linear = nn.Linear(W, W)
        self.ins_feature_linear = nn.Linear(W, W)
        self.rgb_feature_linears = nn.ModuleList([nn.Linear(W + input_ch_views, W // 2)])
        self.ins_feature_linears = nn.ModuleList([nn.Linear(W, W // 2)])
        self.density_linear = nn.Linear(W, 1)
        self.ins_linear = nn.Linear(W // 2, ins_num + 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self._blob = None
        self._blob_key = None
        self._blob_t = None
        self._blob_t_key = None
        self._blob_f = self._blob_s = self._flat = None
        self._blob_f_key = self._blob_s_key = self._blob_ts_key = self._flat_key = None

    # -- kernel-layout weights --------------------------------------------------------------
    def _fused_ok(self):
        """True for the shape every shipped config uses (D=8, W=256, skips=[4], 63+27 input channels; config.py:126-138
        defaults), which the fused register-chained kernels are specialised for.  Any other shape (args.netdepth /
        netwidth / multires*) runs layer by layer on the kernels of csrc/generic.hip (dm_nerf_amd/generic.py)."""
        return (self.D == 8 and self.W == 256 and list(self.skips) == [4]
                and self.input_ch_pts == 63 and self.input_ch_views == 27)

    def _check_supported(self):
        if not self._fused_ok():
            raise NotImplementedError(
                "the packed weight blobs exist for the fused kernels' shape only (D=8, W=256, skips=[4], 63+27 input "
                "channels); other shapes go through dm_nerf_amd.generic")

    def invalidate_blobs(self):
        """Forget every cached kernel-layout copy of the weights; the next call re-packs from the parameters.

        The caches are keyed on ``(data_ptr, tensor._version)`` of every parameter, which follows
        ``optimizer.step()``, ``load_state_dict`` and any in-place op on the parameter itself.  Updates made THROUGH
        ``.data`` (``p.data.add_(...)``, hand-written optimizers, EMA / weight-clipping code) do not bump ``_version``:
        call this after them, or the kernels keep using the stale copy."""
        self._blob_key = self._blob_t_key = self._flat_key = None
        self._blob_f_key = self._blob_s_key = self._blob_ts_key = self._blob_h_key = self._blob_th_key = None

    def install_packed(self, flat, blob, blob_t):
        """Adopt kernel-layout copies made elsewhere from the CURRENT parameters (dm_nerf_amd.optim.FlatAdam re-packs all of them
        in the launch that follows its update, csrc/optim.hip::repack_kernel): ``flat()``, ``blob()`` and ``blob_t()`` return them
        until a parameter changes again.  The opt-in layouts (fused / split) are forgotten: an update THROUGH the flat storage
        does not bump ``_version``, their keys could not tell."""
        key = tuple((p.data_ptr(), p._version) for _, p in self.named_parameters())
        self._flat, self._blob, self._blob_t = flat, blob, blob_t
        self._flat_key = self._blob_key = self._blob_t_key = key
        self._blob_f_key = self._blob_s_key = self._blob_ts_key = self._blob_h_key = self._blob_th_key = None

    def flat(self):
        """The parameters as ONE flat f32 vector in state_dict order (what the packers gather from and what the backward's
        head kernels read, csrc/heads.hip); a NEW tensor whenever a parameter changed, so a pending backward keeps the
        vector its forward used."""
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if getattr(self, "_flat", None) is None or key != self._flat_key:
            self._flat = weights.flat_params(state)
            self._flat_key = key
        return self._flat

    def blob(self):
        """Kernel-layout weights, refreshed if any parameter was updated in place or replaced (see ``invalidate_blobs``): a NEW
        tensor whenever a parameter changed, like ``flat`` -- a pending backward (two forwards around an optimizer step,
        ``retain_graph``) keeps the blob ITS forward used, so its data gradients and its weight gradients see the same weights."""
        self._check_supported()
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if self._blob is None or key != self._blob_key:
            self._blob = weights.pack_blob(state, self.ins_num, flat=self.flat())
            self._blob_key = key
        return self._blob

    def blob_fused(self):
        """Inference blob with ``rgb_feature_linear`` / ``ins_feature_linear`` folded into the hidden layers
        (``dm_nerf(..., args)`` with ``args.fuse_heads = True``; same refresh rule as ``blob``)."""
        self._check_supported()
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if getattr(self, "_blob_f", None) is None or key != self._blob_f_key:
            self._blob_f = weights.pack_blob(state, self.ins_num, fused=True)
            self._blob_f_key = key
        return self._blob_f

    def blob_split(self):
        """Split-bf16 inference blob (``args.mfma_split``; same refresh rule as ``blob``)."""
        self._check_supported()
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if getattr(self, "_blob_s", None) is None or key != self._blob_s_key:
            self._blob_s = weights.pack_blob_split(state, self.ins_num)
            self._blob_s_key = key
        return self._blob_s

    def blob_f16(self):
        """Split-f16 inference blob (``args.mfma_split = "f16x2"``; same refresh rule as ``blob``)."""
        self._check_supported()
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if getattr(self, "_blob_h", None) is None or key != self._blob_h_key:
            self._blob_h = weights.pack_blob_f16(state, self.ins_num)
            self._blob_h_key = key
        return self._blob_h

    def blob_t_f16(self):
        """Split-f16 W^T blob of the opt-in data-gradient kernel (training with ``args.mfma_split = "f16x2"``)."""
        self._check_supported()
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if getattr(self, "_blob_th", None) is None or key != self._blob_th_key:
            self._blob_th = weights.pack_blob_t_f16(self.flat(), self.ins_num)
            self._blob_th_key = key
        return self._blob_th

    def blob_t_split(self):
        """Split-bf16 W^T blob of the opt-in data-gradient kernel (training with ``args.mfma_split``)."""
        self._check_supported()
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if getattr(self, "_blob_ts", None) is None or key != self._blob_ts_key:
            self._blob_ts = weights.pack_blob_t_split(self.flat(), self.ins_num)
            self._blob_ts_key = key
        return self._blob_ts

    def blob_t(self):
        """W^T blob for the backward data-gradient kernel (same refresh rule as ``blob``)."""
        self._check_supported()
        state = dict(self.named_parameters())
        key = tuple((p.data_ptr(), p._version) for p in state.values())
        if self._blob_t is None or key != self._blob_t_key:
            self._blob_t = weights.pack_blob(state, self.ins_num, transposed=True, flat=self.flat())
            self._blob_t_key = key
        return self._blob_t

    def forward(self, x):
        """``[M, 63+27] -> [M, 4 + ins_num + 1]`` = cat[rgb, density, ins] (networks/dm_nerf.py:80-106)."""
        train = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not self._fused_ok():
            from .. import generic
            if x.requires_grad:
                raise NotImplementedError("dm_nerf_amd: DM_NeRF.forward gives gradients for the parameters only")
            return generic.mlp_embedded(self, x, train)
        if train:
            from .. import autograd
            return autograd.mlp_forward_train(self, x)
        x2 = _lib.f32(x.reshape(-1, x.shape[-1]))
        _lib.require_gpu(x2)
        if x2.shape[-1] != self.input_ch_pts + self.input_ch_views:
            raise ValueError(f"DM_NeRF.forward expects {self.input_ch_pts + self.input_ch_views} input channels")
        M = x2.shape[0]
        out = torch.empty(M, 4 + self.ins_num + 1, dtype=torch.float32, device=x2.device)
        _lib.check(_lib.load().dmnerf_mlp_fwd_embedded(_lib.ptr(self.blob()), self.ins_num, _lib.ptr(x2), M,
                                                       _lib.ptr(out), _lib.stream()), "dmnerf_mlp_fwd_embedded")
        return out.reshape(*x.shape[:-1], out.shape[-1])
