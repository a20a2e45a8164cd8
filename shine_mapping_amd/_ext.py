"""Loader of lib/_shine_ext.so — Tier A's autograd nodes in C++ (csrc/shine_torch_ext.cpp).

The extension holds query_feature / Decoder.sdf (one fused node) / get_gradient / sdf_bce_loss / FusedAdam.step as
torch::autograd::Node subclasses over the C ABI; the Python nodes of autograd_ops.py stay as what they fall back to (a differentiable
backward through the fused node, a driver that touches the feature tensor) and as the implementation when the extension is
switched off (SHINE_TIER_A_EXT=0) or was not built.  Same launches either way — what changes is the host time per iteration.
"""
import importlib.util
import os
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "lib", "_shine_ext.so")
_mod = None
_tried = False
_OCTREES = weakref.WeakValueDictionary()  # id(octree) -> octree, for the callbacks the C++ nodes make


def enabled() -> bool:
    return os.environ.get("SHINE_TIER_A_EXT", "1") != "0"


def module():
    """the extension module, or None (switched off / not built).  Loading it needs libshine_hip.so next to it (rpath $ORIGIN)."""
    global _mod, _tried
    if not enabled():
        return None
    if _tried:
        return _mod
    _tried = True
    if not os.path.isfile(EXT_PATH):
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    from . import _lib

    _lib.lib()  # the C ABI library first: the extension links against it
    try:
        spec = importlib.util.spec_from_file_location("_shine_ext", EXT_PATH)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        import ctypes

        if mod.config_bytes() != ctypes.sizeof(_lib.StepConfig):
            raise ImportError("built against another shine_step_config")
        mod.set_callbacks(_interp_backward, _fused_split, _read_done)
    except Exception as e:  # e.g. undefined symbols after a torch upgrade: ONE warning, then the Python nodes every time
        import warnings

        warnings.warn("shine_mapping_amd: lib/_shine_ext.so did not load (%s: %s) — falling back to the Python autograd nodes; "
                      "rebuild with `python -m shine_mapping_amd.build --force`" % (type(e).__name__, str(e)[:500]))
        return None
    _mod = mod
    return _mod


def register(octree) -> int:
    k = id(octree)
    _OCTREES[k] = octree
    return k


def _interp_backward(octree_id, g, coord, need_coord, feats):
    from .autograd_ops import OctreeInterpBackward

    octree = _OCTREES[octree_id]
    outs = OctreeInterpBackward.apply(g, coord, octree, bool(need_coord), *feats)
    return (outs[0] if need_coord else None,) + tuple(o if f.requires_grad else None for o, f in zip(outs[1:], feats))


def _fused_split(octree_id, g, coord, feats, mlp, needs):
    """a differentiable backward through the fused query_feature -> sdf node: recompute through the split, twice-differentiable
    nodes (autograd_ops._fused_split_backward's body)"""
    import torch

    from .autograd_ops import FusedMLP, OctreeInterp

    octree = _OCTREES[octree_id]
    cand = [coord] + list(feats) + list(mlp)
    if g is None:
        return (None,) * len(cand)
    with torch.enable_grad():
        feat = OctreeInterp.apply(coord, octree, *feats)
        octree.__dict__.pop("_spec_result", None)
        pred = FusedMLP.apply(feat, *mlp)
        wanted = [t for t, n in zip(cand, needs) if n and t.requires_grad]
        got = iter(torch.autograd.grad(pred, wanted, g, create_graph=True, allow_unused=True)) if wanted else iter(())
    return tuple(next(got) if (n and t.requires_grad) else None for t, n in zip(cand, needs))


def _read_done(octree_id):
    octree = _OCTREES.get(octree_id)
    if octree is not None:
        octree._tables_read_done()
