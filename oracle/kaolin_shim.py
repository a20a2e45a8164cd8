"""CPU restatement of the five NVIDIA-kaolin SPC ops the SHINE hot path touches.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the
product package ``shine_mapping_amd``; only ``tests/``, ``bench.py``'s
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` use it, as the checker.

kaolin is a third-party dependency of the reference that is NOT vendored under
``/root/reference`` (pinned ``kaolin==0.12.0`` in README.md:95 and
``git checkout v0.13.0`` in Dockerfile:33) and is not installable here.  Its
published semantics are restated below; the reference's own call sites are

  quantize_points            model/feature_octree.py:88,203,271
  points_to_morton           model/feature_octree.py:89,123,134,162,164,204,272
  morton_to_points           model/feature_octree.py:97
  points_to_corners          model/feature_octree.py:131
  unbatched_pointcloud_to_spc model/feature_octree.py:116

PARITY UNPINNED for these five ops: the reference holds no test or golden vector
at this boundary.  What is observable downstream (voxel membership from the
quantise formula, corner<->weight correspondence, lexicographic corner-id
order) is covered by self-consistency tests in tests/test_oracle.py; the
Morton bit order itself is an internal key (any bijection used on both the
build and the query side gives identical results).
"""
import sys
import types

import torch


def quantize_points(x: torch.Tensor, level: int) -> torch.Tensor:
    """kaolin.ops.spc.quantize_points: [-1,1] floats -> int16 voxel coords at `level`."""
    res = 2 ** level
    return torch.floor(torch.clamp(res * (x + 1.0) / 2.0, 0, res - 1.0)).short()


def points_to_morton(points: torch.Tensor) -> torch.Tensor:
    """kaolin.ops.spc.points_to_morton: int16 [..,3] -> int64; per bit triplet x is the MSB, z the LSB."""
    p = points.to(torch.int64)
    px, py, pz = p[..., 0], p[..., 1], p[..., 2]
    code = torch.zeros_like(px)
    for b in range(16):
        code |= ((pz >> b) & 1) << (3 * b)
        code |= ((py >> b) & 1) << (3 * b + 1)
        code |= ((px >> b) & 1) << (3 * b + 2)
    return code


def morton_to_points(morton: torch.Tensor) -> torch.Tensor:
    """kaolin.ops.spc.morton_to_points: inverse of points_to_morton, int16 [..,3]."""
    m = morton.to(torch.int64)
    px = torch.zeros_like(m)
    py = torch.zeros_like(m)
    pz = torch.zeros_like(m)
    for b in range(16):
        pz |= ((m >> (3 * b)) & 1) << b
        py |= ((m >> (3 * b + 1)) & 1) << b
        px |= ((m >> (3 * b + 2)) & 1) << b
    return torch.stack((px, py, pz), dim=-1).short()


def points_to_corners(points: torch.Tensor) -> torch.Tensor:
    """kaolin.ops.spc.points_to_corners: [M,3] -> [M,8,3]; corner j offset = ((j>>2)&1,(j>>1)&1,j&1)."""
    offs = torch.tensor(
        [[(j >> 2) & 1, (j >> 1) & 1, j & 1] for j in range(8)], dtype=points.dtype, device=points.device
    )
    return points.unsqueeze(-2) + offs


class _Spc:
    """Just the two attributes model/feature_octree.py:117,122 reads."""

    def __init__(self, point_hierarchies, pyramids):
        self.point_hierarchies = point_hierarchies
        self.pyramids = pyramids


def unbatched_pointcloud_to_spc(pointcloud: torch.Tensor, level: int, features=None) -> _Spc:
    """kaolin.ops.conversions.unbatched_pointcloud_to_spc.

    Quantise at `level`, unique, then every coarser level is the unique set of
    parent voxels; each level is stored in Morton order, levels concatenated
    root first.  pyramids[0][0,l] = #nodes at level l, pyramids[0][1,l] =
    exclusive prefix offset (length level+2).
    """
    leaf = quantize_points(pointcloud.contiguous(), level)
    per_level = []
    cur = leaf
    for l in range(level, -1, -1):
        m = torch.unique(points_to_morton(cur))  # sorted ascending == Morton order
        pts = morton_to_points(m)
        per_level.append(pts)
        cur = (pts >> 1) if l > 0 else pts
    per_level.reverse()
    counts = torch.tensor([p.shape[0] for p in per_level], dtype=torch.int32)
    pyramid = torch.zeros(2, level + 2, dtype=torch.int32)
    pyramid[0, : level + 1] = counts
    pyramid[1, 1:] = torch.cumsum(counts, 0)
    return _Spc(torch.cat(per_level, 0), pyramid.unsqueeze(0))


def install() -> None:
    """Register the shim as `kaolin` in sys.modules (used only to run the real reference in-container)."""
    if "kaolin" in sys.modules and not getattr(sys.modules["kaolin"], "_shine_shim", False):
        return
    kal = types.ModuleType("kaolin")
    kal._shine_shim = True
    ops = types.ModuleType("kaolin.ops")
    spc = types.ModuleType("kaolin.ops.spc")
    conv = types.ModuleType("kaolin.ops.conversions")
    render = types.ModuleType("kaolin.render")
    rspc = types.ModuleType("kaolin.render.spc")
    for fn in (quantize_points, points_to_morton, morton_to_points, points_to_corners):
        setattr(spc, fn.__name__, fn)
    conv.unbatched_pointcloud_to_spc = unbatched_pointcloud_to_spc
    ops.spc, ops.conversions = spc, conv
    render.spc = rspc
    kal.ops, kal.render = ops, render
    sys.modules.update(
        {
            "kaolin": kal,
            "kaolin.ops": ops,
            "kaolin.ops.spc": spc,
            "kaolin.ops.conversions": conv,
            "kaolin.render": render,
            "kaolin.render.spc": rspc,
        }
    )
