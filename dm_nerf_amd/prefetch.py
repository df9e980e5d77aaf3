"""Host side of the training step, hidden behind the GPU (second half).

The reference's loop does, per iteration and on the critical path (train_dmsr.py:24-31, train_scannet.py:24-32):
``np.random.choice(i_train)``, three ``.to(device)`` copies of the image / pose / labels, a full-frame ``get_rays_k``,
``np.random.choice(H*W, N_train, replace=False)`` (a permutation of 307 200 pixels) and the gathers -- tens of
milliseconds of host work per step, which is nothing next to its seconds-long CPU step but longer than the 23-31 ms
MI355X step.  Here:

* the dataset (images, labels) is uploaded ONCE and stays resident in HBM (a few hundred MB of 288 GB); poses stay on the
  host (12 floats per step go into the ray-generation kernel's argument block, no device round trip);
* a side thread owns the selection RNG and draws, in EXACTLY the reference's order, the image index, the pixel set (and,
  every ``i_test`` iterations, the ten test views the loop picks at train_dmsr.py:92) for the next ``depth`` steps into
  pinned host buffers;
* the main thread enqueues one small asynchronous H2D copy of the indices, the ray generation for the selected pixels only
  (``dmnerf_raygen_select``) and two gathers -- no synchronisation anywhere.

The numpy stream: the reference seeds the GLOBAL legacy generator (``np.random.seed(0)``, train_dmsr.py:13) and every
draw above goes through it.  ``np.random.RandomState(seed)`` is the same Mersenne-Twister stream with the same
``choice`` algorithm, so a private instance owned by the worker thread reproduces the reference's batches bit for bit
without touching global state (tests/test_prefetch.py pins this against batches drawn by the reference's own
``get_select_full`` / ``get_select_crop``).
"""
import collections
import ctypes
import queue
import threading

import numpy as np
import torch

from . import _lib

Selection = collections.namedtuple("Selection", "step img_i idx n_ins test_pick")
Batch = collections.namedtuple("Batch", "step img_i target_c target_i rays n_ins test_pick")


class SelectionStream:
    """The host RNG draws of the reference's training loops, one ``Selection`` per iteration (numpy only, no GPU).

    DM-SR / Replica form (train_dmsr.py:24-31, helpers.py:99-111)::

        img_i = np.random.choice(i_train)
        idx   = np.random.choice(H*W, size=[N_train], replace=False)
        every i_test iterations, AFTER the step (train_dmsr.py:88-92):  np.random.choice(len(i_test), size=[10], replace=False)

    ScanNet form (``ins_indices`` and ``crop_mask`` given; train_scannet.py:24-32, helpers.py:64-95)::

        img_i   = np.random.choice(i_train)
        labeled = ins_indices[img_i][np.random.choice(len(ins_indices[img_i]), size=[N_ins], replace=False)]
        unl     = crop_indices[np.random.choice(|crop \\ labeled|, size=[N_train - N_ins], replace=False)]   # (sic, :82-84)
        idx     = concat(unl, labeled)          # labelled rays LAST (render.py:88-90)
    """

    def __init__(self, i_train, n_pixels, N_train, seed=0, i_test=None, i_test_every=None, n_test_pick=10,
                 ins_indices=None, crop_mask=None, rng=None):
        self.rng = rng if rng is not None else np.random.RandomState(seed)
        self.i_train = np.asarray(i_train)
        self.n_pixels, self.N_train = int(n_pixels), int(N_train)
        self.i_test = None if i_test is None else np.asarray(i_test)
        self.i_test_every, self.n_test_pick = i_test_every, n_test_pick
        self.ins_indices = ins_indices
        if crop_mask is not None:
            self.crop_flat = np.asarray(crop_mask).reshape(-1) == 1
            self.crop_indices = np.where(self.crop_flat)[0]
        self.scannet = ins_indices is not None
        self.step = 0

    def draw(self):
        rng, i = self.rng, self.step
        img_i = int(rng.choice(self.i_train))
        n_ins = None
        if not self.scannet:
            idx = rng.choice(self.n_pixels, size=[self.N_train], replace=False)
        else:
            ins_index = np.asarray(self.ins_indices[img_i])
            n_ins = min(int(self.N_train * 0.3), len(ins_index))
            labeled = ins_index[rng.choice(ins_index.shape[0], size=[n_ins], replace=False)]
            # |set(crop) - set(labeled)| without building two 300 k-element Python sets per step (helpers.py:81):
            # the labelled pixels are distinct, so it is |crop| minus those of them that lie inside the crop
            n_unlabeled = len(self.crop_indices) - int(self.crop_flat[labeled].sum())
            unl = self.crop_indices[rng.choice(n_unlabeled, size=[self.N_train - n_ins], replace=False)]
            idx = np.concatenate([unl, labeled])
        pick = None
        if self.i_test is not None and self.i_test_every and i % self.i_test_every == 0:
            pick = rng.choice(len(self.i_test), size=[self.n_test_pick], replace=False)
        self.step += 1
        return Selection(i, img_i, idx.astype(np.int64), n_ins, pick)


class TrainBatchPrefetcher:
    """Iterator over training batches ``(step, img_i, target_c [N,3], target_i [N or N_ins], rays [2,N,3], n_ins,
    test_pick)`` on ``device``, equal to what ``get_select_full`` / ``get_select_crop`` return for the same numpy stream.

    ``images [n,H,W,3]`` float, ``labels [n,H,W]`` integer (uploaded once), ``poses [n,3or4,4]`` (kept on the host),
    ``K`` numpy intrinsics.  ``depth`` selections are drawn ahead by a daemon thread.  ``max_steps``: stop after that many
    batches (None = endless, like the reference's 500 001-iteration loop)."""

    def __init__(self, images, labels, poses, K, i_train, N_train, device, seed=0, i_test=None, i_test_every=None,
                 n_test_pick=10, ins_indices=None, crop_mask=None, depth=3, max_steps=None):
        self.device = torch.device(device)
        self.images = torch.as_tensor(images).to(self.device, torch.float32).contiguous()
        self.labels = torch.as_tensor(labels).to(self.device).contiguous()
        n, self.H, self.W, _ = self.images.shape
        self.poses = np.ascontiguousarray(torch.as_tensor(poses).detach().cpu().numpy().astype(np.float32)[:, :3, :4])
        Kn = np.asarray(K)
        self.intr = np.array([Kn[0, 0], Kn[1, 1], Kn[0, 2], Kn[1, 2], Kn[2, 2]], dtype=np.float64).astype(np.float32)
        self.N = int(N_train)
        self.stream = SelectionStream(i_train, self.H * self.W, N_train, seed, i_test, i_test_every, n_test_pick, ins_indices, crop_mask)
        self.max_steps = max_steps
        self.depth = max(1, int(depth))
        # ring of pinned index buffers; a buffer is refilled only after the H2D copy that read it has completed
        self.n_slots = self.depth + 2
        self.pinned = [torch.empty(self.N, dtype=torch.int64).pin_memory() for _ in range(self.n_slots)]
        self.copied = [None] * self.n_slots                      # torch.cuda.Event per slot, recorded after its copy
        self.q = queue.Queue(maxsize=self.depth)
        self._stop = threading.Event()
        self._err = None
        self.thread = threading.Thread(target=self._work, name="dmnerf-batch-prefetch", daemon=True)
        self.thread.start()

    # ---- worker thread: numpy only -------------------------------------------------------------------
    def _work(self):
        try:
            k = 0
            while not self._stop.is_set() and (self.max_steps is None or k < self.max_steps):
                slot = k % self.n_slots
                ev = self.copied[slot]
                if ev is not None:
                    ev.synchronize()                             # the copy out of this pinned buffer is done
                sel = self.stream.draw()
                self.pinned[slot].numpy()[:] = sel.idx
                item = (slot, sel)
                while not self._stop.is_set():
                    try:
                        self.q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                k += 1
            self.q.put(None)
        except BaseException as e:                               # noqa: BLE001 -- surfaced in the consumer
            self._err = e
            self.q.put(None)

    # ---- consumer: enqueue-only, no synchronisation ------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration
        slot, sel = item
        idx = torch.empty(self.N, dtype=torch.int64, device=self.device)
        idx.copy_(self.pinned[slot], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.copied[slot] = ev
        rays = torch.empty(2, self.N, 3, dtype=torch.float32, device=self.device)
        c = self.poses[sel.img_i]
        _lib.check(_lib.load().dmnerf_raygen_select(self.H, self.W, self.intr.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p),
                                                    _lib.ptr(idx), self.N, _lib.ptr(rays[0]), _lib.ptr(rays[1]), _lib.stream()),
                   "dmnerf_raygen_select")
        target_c = self.images[sel.img_i].reshape(-1, 3)[idx]
        lab = self.labels[sel.img_i].reshape(-1)
        target_i = lab[idx] if sel.n_ins is None else lab[idx[self.N - sel.n_ins:]]
        return Batch(sel.step, sel.img_i, target_c, target_i, rays, sel.n_ins, sel.test_pick)

    def close(self):
        self._stop.set()
        try:
            while True:
                self.q.get_nowait()
        except queue.Empty:
            pass
        self.thread.join(timeout=5)

    def __del__(self):
        try:
            self._stop.set()
        except Exception:                                        # noqa: BLE001
            pass
