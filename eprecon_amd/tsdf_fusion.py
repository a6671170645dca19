"""TSDF integration of depth frames on libeprecon_hip.so — mirror of TSDFVolumeTorch
(tools/tsdf_fusion/fusion.py:488-577 of the reference), the volume the data pipeline fuses for every sample
(datasets/transforms.py:286-297,375-387), with the same constructor and methods:

    vol = TSDFVolumeHIP(voxel_dim, origin, voxel_size, margin=3)
    vol.integrate(depth_im, cam_intr, cam_pose, obs_weight=1.)        # one frame, as the reference is driven
    vol.integrate_views(depths, intrs, poses)                           # all frames of a fragment in ONE launch
    tsdf_vol, weight_vol = vol.get_volume();  occ = vol.occupancy()

`variant="torch"` (default) reproduces TSDFVolumeTorch's arithmetic bit for bit (pinned by
tests/golden/tsdf_fusion.npz); `variant="cuda"` that of the reference's PyCUDA kernel (fusion.py:67-142).
The world->camera matrix is torch.inverse(cam_pose) on the host, the very call the reference makes (:454).
"""
import ctypes

import numpy as np
import torch

from . import _lib

VARIANTS = {"torch": 0, "cuda": 1}


class TSDFVolumeHIP:
    def __init__(self, voxel_dim, origin, voxel_size, margin=3, device=None, variant="torch"):
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if self.device.type != "cuda":
            raise _lib.EpreconError("TSDFVolumeHIP needs a GPU (no CPU fallback)")
        self._voxel_size = float(voxel_size)
        self._sdf_trunc = margin * self._voxel_size
        self._vol_dim = [int(v) for v in torch.as_tensor(voxel_dim).tolist()]
        self._vol_origin = np.asarray(torch.as_tensor(origin).detach().cpu().numpy(), np.float32).reshape(3)
        self._variant = VARIANTS[variant]
        self._occ = None
        self.reset()

    def reset(self):
        self._tsdf_vol = torch.ones(self._vol_dim, dtype=torch.float32, device=self.device)
        self._weight_vol = torch.zeros(self._vol_dim, dtype=torch.float32, device=self.device)
        self._occ = None

    def integrate_views(self, depths, intrs, poses, obs_weight=1.0, world2cam=None):
        """depths f32[V,H,W]; intrs f32[V,3,3]; poses f32[V,4,4] (camera->world) — V successive integrate() calls.
        world2cam (optional, f32[V,4,4]): use these world->camera matrices instead of torch.inverse(poses); the
        last bits of a 4x4 float inverse depend on the host's LAPACK build, so a result that must be reproduced
        bit for bit on another machine has to carry its matrices along (tests/golden/tsdf_fusion.npz does)."""
        lib = _lib.load()
        depths = depths.to(device=self.device, dtype=torch.float32).contiguous()
        v, h, w = depths.shape
        poses = torch.as_tensor(poses).detach().float().cpu().reshape(v, 4, 4)
        if self._variant == 0:
            cam = torch.inverse(poses) if world2cam is None else torch.as_tensor(world2cam).float().cpu().reshape(v, 4, 4)
        else:
            cam = poses                                                      # fusion.py:454 (torch.inverse, CPU)
        cam = np.ascontiguousarray(cam.numpy(), np.float32)
        intr = np.ascontiguousarray(torch.as_tensor(intrs).detach().float().cpu().reshape(v, 3, 3).numpy(), np.float32)
        dims = (ctypes.c_int32 * 3)(*self._vol_dim)
        self._occ = torch.empty(self._vol_dim, dtype=torch.uint8, device=self.device)
        _lib.check(lib.eprecon_tsdf_integrate_async(
            _lib.ptr(self._tsdf_vol), _lib.ptr(self._weight_vol), ctypes.cast(dims, ctypes.c_void_p),
            self._vol_origin.ctypes.data_as(ctypes.c_void_p), self._voxel_size, _lib.ptr(depths), v, h, w,
            intr.ctypes.data_as(ctypes.c_void_p), cam.ctypes.data_as(ctypes.c_void_p), float(self._sdf_trunc),
            float(obs_weight), self._variant, _lib.ptr(self._occ), _lib.current_stream()), "eprecon_tsdf_integrate_async")

    def integrate(self, depth_im, cam_intr, cam_pose, obs_weight):
        """tools/tsdf_fusion/fusion.py:551-575"""
        self.integrate_views(torch.as_tensor(depth_im)[None], torch.as_tensor(cam_intr)[None],
                             torch.as_tensor(cam_pose)[None], obs_weight)

    def get_volume(self):
        return self._tsdf_vol, self._weight_vol

    def occupancy(self):
        """(tsdf < 0.999) & (tsdf > -0.999) & (weight > 1) — datasets/transforms.py:295-297 (at least two views)"""
        if self._occ is None:
            t, w = self._tsdf_vol, self._weight_vol
            return (t < 0.999) & (t > -0.999) & (w > 1)
        return self._occ.bool()

    @property
    def sdf_trunc(self):
        return self._sdf_trunc

    @property
    def voxel_size(self):
        return self._voxel_size
