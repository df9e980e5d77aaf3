"""dm_nerf_amd -- MI355X (gfx950) implementation of DM-NeRF's ray-rendering hot path.

Drop-in mirrors of the reference's callables (vLAR-group/DM-NeRF):

    reference import                                   this package
    ------------------------------------------------   ---------------------------------------------
    from networks.render import dm_nerf, render_train  from dm_nerf_amd.networks.render import ...
    from networks.dm_nerf import DM_NeRF, get_embedder from dm_nerf_amd.networks.dm_nerf import ...
    from networks.helpers import get_rays_k, ...       from dm_nerf_amd.networks.helpers import ...
    from config import create_nerf                     from dm_nerf_amd.config import create_nerf

Everything executes in hand-written HIP kernels behind the C ABI of ``libdmnerf_hip.so``
(``include/dmnerf_hip.h``).  There is NO CPU or eager-PyTorch fallback: a missing library or a
non-GPU tensor raises.  (The directory is spelled ``dm_nerf_amd`` because ``dm-nerf_amd`` is not
an importable Python identifier.)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
