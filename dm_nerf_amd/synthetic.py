"""Synthetic DM-SR-style camera used by bench.py and the examples (no dataset ships with the repo).

Intrinsics follow datasets/loader_dmsr.py:136-137 of the reference (``[[f,0,W/2],[0,-f,H/2],[0,0,-1]]`` with
``f = W/2 / tan(camera_angle_x/2)``); poses follow tools/pose_generator.py:29-34 (``pose_spherical``)."""
import numpy as np
import torch


def dmsr_intrinsics(H=480, W=640, camera_angle_x=0.69):
    focal = .5 * W / np.tan(.5 * camera_angle_x)
    return np.array([[focal, 0, 0.5 * W], [0, -focal, 0.5 * H], [0, 0, -1]])


def pose_spherical(theta, phi, radius):
    t = np.eye(4); t[2, 3] = radius
    ph = phi / 180. * np.pi
    rx = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]])
    th = theta / 180. * np.pi
    ry = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]])
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ (ry @ rx @ t)
    return torch.from_numpy(c2w.astype(np.float32))
