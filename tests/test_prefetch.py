"""The prefetching batch selector (dm_nerf_amd/prefetch.py): the numpy stream of the reference's training
loops reproduced from a PRIVATE RandomState by a side thread -- pinned on batches drawn by the reference's own
``get_select_full`` / ``get_select_crop`` under ``np.random.seed(0)`` (tests/golden/select_stream.npz)."""
import numpy as np
import pytest
import torch


def _stream_kw(g):
    H, W, N = [int(v) for v in g["HWN"]]
    return H, W, N, dict(i_train=g["i_train"].numpy(), n_pixels=H * W, N_train=N, seed=0)


def test_selection_stream_reproduces_the_dmsr_loop(golden):
    """img_i, the pixel set and -- every i_test iterations -- the ten test views, in the reference's order (train_dmsr.py:25,
    helpers.py:104, train_dmsr.py:92); afterwards the generator stands exactly where the reference's stands."""
    from dm_nerf_amd.prefetch import SelectionStream
    g = golden("select_stream")
    H, W, N, kw = _stream_kw(g)
    np.random.seed(12345)                                    # the global generator is NOT what the stream uses ...
    before = np.random.get_state()[1].copy()
    s = SelectionStream(i_test=np.arange(int(g["n_i_test"])), i_test_every=3, **kw)
    for i in range(6):
        sel = s.draw()
        assert sel.step == i and sel.img_i == int(g[f"full{i}_img"]) and sel.n_ins is None
        assert torch.equal(g["imgs"][sel.img_i].reshape(-1, 3)[torch.from_numpy(sel.idx)], g[f"full{i}_tc"])
        assert torch.equal(g["labs"][sel.img_i].reshape(-1)[torch.from_numpy(sel.idx)], g[f"full{i}_ti"])
        if i % 3 == 0:
            assert np.array_equal(sel.test_pick, g[f"full{i}_pick"].numpy())
        else:
            assert sel.test_pick is None
    assert s.rng.rand() == float(g["full_next_rand"])
    assert np.array_equal(np.random.get_state()[1], before)  # ... and it is left untouched


def test_selection_stream_reproduces_the_scannet_loop(golden):
    """The ScanNet draws (train_scannet.py:25, helpers.py:76,82) incl. the 30 % quota, its clamp, labelled rays last."""
    from dm_nerf_amd.prefetch import SelectionStream
    g = golden("select_stream")
    H, W, N, kw = _stream_kw(g)
    ins_indices = [g[f"ins_index{k}"].numpy() for k in range(5)]
    s = SelectionStream(ins_indices=ins_indices, crop_mask=g["crop"].numpy(), **kw)
    clamped = False
    for i in range(4):
        sel = s.draw()
        assert sel.img_i == int(g[f"crop{i}_img"]) and sel.n_ins == int(g[f"crop{i}_nins"])
        clamped |= sel.n_ins < int(N * 0.3)
        idx = torch.from_numpy(sel.idx)
        assert torch.equal(g["imgs"][sel.img_i].reshape(-1, 3)[idx], g[f"crop{i}_tc"])
        assert torch.equal(g["labs"][sel.img_i].reshape(-1)[idx[N - sel.n_ins:]], g[f"crop{i}_ti"])
    assert clamped                                            # the fixture exercises the clamp (an image with 8 labelled pixels)
    assert s.rng.rand() == float(g["crop_next_rand"])


@pytest.mark.gpu
def test_prefetcher_batches_equal_get_select_full(golden):
    """The prefetcher's device batches == the drop-in ``get_select_full`` called step by step on the global stream (which
    the golden tests pin to the reference): same pixels, targets, rays, bit for bit; endless-loop order preserved."""
    from dm_nerf_amd.networks import helpers as H_
    from dm_nerf_amd.prefetch import TrainBatchPrefetcher
    g = golden("select_stream")
    H, W, N = [int(v) for v in g["HWN"]]
    K = g["K"].numpy()
    i_train = g["i_train"].numpy()
    pf = TrainBatchPrefetcher(g["imgs"], g["labs"], g["poses"], K, i_train, N, "cuda", seed=0, i_test=np.arange(int(g["n_i_test"])),
                              i_test_every=3, depth=2, max_steps=6)
    imgs, labs, poses = g["imgs"].cuda(), g["labs"].cuda(), g["poses"].cuda()
    np.random.seed(0)
    n = 0
    for b in pf:
        img_i = np.random.choice(i_train)
        tc, ti, rays = H_.get_select_full(imgs[img_i], poses[img_i, :3, :4], K, labs[img_i], N)
        if b.step % 3 == 0:
            assert np.array_equal(b.test_pick, np.random.choice(int(g["n_i_test"]), size=[10], replace=False))
        torch.cuda.synchronize()
        assert b.step == n and b.img_i == img_i
        assert torch.equal(b.target_c, tc) and torch.equal(b.target_i, ti) and torch.equal(b.rays, rays)
        assert torch.equal(b.target_c.cpu(), g[f"full{n}_tc"])
        n += 1
    assert n == 6
    pf.close()


@pytest.mark.gpu
def test_prefetcher_scannet_form_and_full_size(golden):
    """ScanNet form (labelled rays last, N_ins per batch) through the prefetcher, and a full-size run: 640 x 480, N_train 3072,
    20 steps ahead of a consumer that never synchronises -- every batch equals its step-by-step counterpart."""
    from dm_nerf_amd.networks import helpers as H_
    from dm_nerf_amd.prefetch import TrainBatchPrefetcher
    from oracle import ref_cpu as O
    g = golden("select_stream")
    H, W, N = [int(v) for v in g["HWN"]]
    K = g["K"].numpy()
    ins_indices = [g[f"ins_index{k}"].numpy() for k in range(5)]
    pf = TrainBatchPrefetcher(g["imgs"], g["labs"], g["poses"], K, g["i_train"].numpy(), N, "cuda", seed=0, ins_indices=ins_indices,
                              crop_mask=g["crop"].numpy(), max_steps=4)
    for i, b in enumerate(pf):
        assert b.n_ins == int(g[f"crop{i}_nins"]) and b.target_i.shape == (b.n_ins,)
        assert torch.equal(b.target_c.cpu(), g[f"crop{i}_tc"]) and torch.equal(b.target_i.cpu(), g[f"crop{i}_ti"])
        assert torch.equal(b.rays[0].cpu(), g[f"crop{i}_rays"][0]) and torch.allclose(b.rays[1].cpu(), g[f"crop{i}_rays"][1], rtol=3e-7, atol=1e-7)
    pf.close()
    # full size
    Hh, Ww, Nn, n_img = 480, 640, 3072, 3
    gen = torch.Generator().manual_seed(5)
    imgs = torch.rand(n_img, Hh, Ww, 3, generator=gen)
    labs = torch.randint(0, 13, (n_img, Hh, Ww), generator=gen).to(torch.int16)
    poses = torch.stack([O.pose_spherical(40.0 * k, -65.0, 7.0) for k in range(n_img)])
    K = O.dmsr_intrinsics(Hh, Ww)
    pf = TrainBatchPrefetcher(imgs, labs, poses, K, np.arange(n_img), Nn, "cuda", seed=0, max_steps=20, depth=3)
    got = [b for b in pf]                                       # consumed without a single synchronisation
    torch.cuda.synchronize()
    assert len(got) == 20
    di, dl, dp = imgs.cuda(), labs.cuda(), poses.cuda()
    np.random.seed(0)
    for b in got:
        img_i = np.random.choice(np.arange(n_img))
        tc, ti, rays = H_.get_select_full(di[img_i], dp[img_i, :3, :4], K, dl[img_i], Nn)
        assert b.img_i == img_i and torch.equal(b.target_c, tc) and torch.equal(b.target_i, ti) and torch.equal(b.rays, rays)
    pf.close()
