"""Drop-in for ``config.create_nerf`` (config.py:126-138 of the reference)."""
import torch

from .networks.dm_nerf import DM_NeRF, get_embedder


def create_nerf(args):
    """Two embedders + two identical DM_NeRF models on ``args.device``; same return tuple."""
    i_embed = getattr(args, "i_embed", 0)
    position_embedder, input_ch_pos = get_embedder(getattr(args, "multires", 10), i_embed)
    view_embedder, input_ch_view = get_embedder(getattr(args, "multires_views", 4), i_embed)
    device = getattr(args, "device", None) or torch.device("cuda", torch.cuda.current_device())
    D, W = getattr(args, "netdepth", 8), getattr(args, "netwidth", 256)
    model_coarse = DM_NeRF(D, W, input_ch_pos, input_ch_view, [4], args.ins_num).to(device)
    model_fine = DM_NeRF(D, W, input_ch_pos, input_ch_view, [4], args.ins_num).to(device)
    return position_embedder, view_embedder, model_coarse, model_fine, args
