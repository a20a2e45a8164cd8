"""Forward-only SDF query for meshing — the hot half of the reference's Mesher (utils/mesher.py:33-108).

`query_points` keeps the reference's argument list and return convention (numpy arrays; `sdf_pred` is the NEGATED
decoder output, `mc_mask` says whether the point's node exists at the marching-cubes check level) but runs each
chunk as ONE launch of `shine_query_points` (csrc/shine_query.hip): no `[N,8]` index tensors, no `[N,F]` features
and no per-level host round trips.  Marching cubes itself (skimage, utils/mesher.py:262-292) and the open3d
bounding-box plumbing stay with the caller: they are outside the hot path.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .ops import _stream


def query_points_device(octree, decoder, coord, check_level=0, negate=True, query_sdf=True, query_mask=True):
    """One launch over `coord [N,3]` (device, scaled to [-1,1]); returns device tensors (sdf f32 | None, mask bool | None)."""
    t = octree._require_tables()
    coord = octree._check_coord(coord.detach())
    n = coord.shape[0]
    dev = coord.device
    sdf = torch.empty(n, dtype=torch.float32, device=dev) if query_sdf else None
    mask = torch.empty(n, dtype=torch.uint8, device=dev) if query_mask else None
    cfg = octree.step_config()
    mlp = [p.detach() for p in decoder.fused_params()] if query_sdf else None
    _lib.check(
        _lib.lib().shine_query_points(
            t.handle, C.byref(cfg), coord.data_ptr(), n, octree.feature_ptrs(), octree.row_counts(),
            _lib.ptr_array([p.data_ptr() for p in mlp]) if mlp is not None else None, int(check_level),
            1 if negate else 0, sdf.data_ptr() if sdf is not None else None,
            mask.data_ptr() if mask is not None else None, _stream(),
        ),
        "shine_query_points",
    )
    return sdf, (mask.bool() if mask is not None else None)


class Mesher:
    """The query half of utils/mesher.py's Mesher: same constructor arguments, same `query_points`."""

    def __init__(self, config, octree, geo_decoder, sem_decoder=None):
        self.config = config
        self.octree = octree
        self.geo_decoder = geo_decoder
        self.sem_decoder = sem_decoder
        self.device = config.device
        self.cur_device = self.device
        self.dtype = config.dtype
        self.world_scale = config.scale
        self.ts = 0
        self.global_transform = np.eye(4)

    def get_query_from_bbx(self, bbx, voxel_size):
        """utils/mesher.py:110-152: grid query points of a box (anything with get_min_bound()/get_max_bound(), e.g. an
        open3d AxisAlignedBoundingBox), padded, one extra layer underground; same fp32 op order, built on the device."""
        min_bound = np.asarray(bbx.get_min_bound(), dtype=np.float64).copy()
        max_bound = np.asarray(bbx.get_max_bound(), dtype=np.float64)
        voxel_num_xyz = (np.ceil((max_bound - min_bound) / voxel_size) + self.config.pad_voxel * 2).astype(np.int_)
        voxel_origin = min_bound - self.config.pad_voxel * voxel_size
        voxel_origin[2] -= voxel_size
        voxel_num_xyz[2] += 1
        dev = self.octree.hier_features[0].device if len(self.octree.hier_features) else self.device
        x = torch.arange(int(voxel_num_xyz[0]), dtype=torch.int16, device=dev)
        y = torch.arange(int(voxel_num_xyz[1]), dtype=torch.int16, device=dev)
        z = torch.arange(int(voxel_num_xyz[2]), dtype=torch.int16, device=dev)
        x, y, z = torch.meshgrid(x, y, z, indexing="ij")
        coord = torch.stack((x.flatten(), y.flatten(), z.flatten())).transpose(0, 1).float()
        coord *= voxel_size
        coord += torch.tensor(voxel_origin, dtype=self.dtype, device=dev)
        coord *= self.world_scale
        return coord, voxel_num_xyz, voxel_origin

    def query_points(self, coord, bs, query_sdf=True, query_sem=False, query_mask=True):
        """utils/mesher.py:33-108.  Returns (sdf_pred, sem_pred, mc_mask) as numpy arrays (None when not asked for)."""
        if query_sem:
            raise NotImplementedError("semantic decoding is outside the SDF hot path (SURVEY.md §8): use the reference's Mesher")
        if getattr(self.config, "time_conditioned", False):
            raise NotImplementedError("time-conditioned decoding is outside the SDF hot path")
        sample_count = coord.shape[0]
        iter_n = math.ceil(sample_count / bs)
        check_level = min(self.octree.featured_level_num, self.config.mc_vis_level) - 1
        dev = self.octree.hier_features[0].device
        with torch.no_grad():
            if iter_n > 1:  # chunked: the reference fills float64 numpy buffers (:43-53)
                sdf_pred = np.zeros(sample_count) if query_sdf else None
                mc_mask = np.zeros(sample_count) if query_mask else None
                for i in range(iter_n):
                    head, tail = i * bs, min((i + 1) * bs, sample_count)
                    sdf, mask = query_points_device(self.octree, self.geo_decoder, coord[head:tail].to(dev),
                                                    check_level, True, query_sdf, query_mask)
                    if query_sdf:
                        sdf_pred[head:tail] = sdf.cpu().numpy()
                    if query_mask:
                        mc_mask[head:tail] = mask.cpu().numpy()
            else:
                sdf, mask = query_points_device(self.octree, self.geo_decoder, coord.to(dev), check_level, True,
                                                query_sdf, query_mask)
                sdf_pred = sdf.cpu().numpy() if query_sdf else None
                mc_mask = mask.cpu().numpy() if query_mask else None
        return sdf_pred, None, mc_mask
