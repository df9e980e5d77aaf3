"""Training-mode (autograd) entry points.  Backward kernels are not built yet: fail loudly."""


def _todo(what):
    raise NotImplementedError(
        f"dm_nerf_amd: {what} under autograd is not implemented yet (the HIP backward kernels "
        "composite_bwd / mlp_bwd are the next rows of SURVEY.md section 8); run under torch.no_grad() "
        "for inference.  There is deliberately no eager-PyTorch fallback.")


def mlp_forward_train(model, x):
    _todo("DM_NeRF.forward")


def render_train_train(raw, z_vals, rays_d):
    _todo("render_train")


def dm_nerf_train(rays, model_coarse, model_fine, z_vals_coarse, args, t_rand=None, u=None):
    _todo("dm_nerf")
