"""Zero-edit drop-in for the reference's drivers ("packages with those import paths").

    import shine_mapping_amd.dropin        # ONE line at the top of shine_batch.py / shine_incre.py

registers this package's classes under the module paths the reference imports them from
(`shine_batch.py:13-14`, `shine_incre.py:12-13`, `utils/mesher.py:11-12`, `utils/incre_learning.py`):

    model.feature_octree.FeatureOctree  ->  shine_mapping_amd.FeatureOctree
    model.decoder.Decoder               ->  shine_mapping_amd.Decoder

Only these two modules are replaced — `dataset.*` and the rest of `model.*` keep resolving to the reference's own files.  A
side effect worth having: the reference's `model/feature_octree.py` (and with it the kaolin import at `:6`) is never executed
on the training path.

Four FUNCTIONS of the reference's own `utils` modules are additionally re-bound (the drivers pick them up through
`from utils.tools import *` / `from utils.loss import *`, shine_batch.py:14-15), each falling back to the reference's original
for anything it does not cover, each with an opt-out environment variable (= "0"):

    utils.tools.setup_optimizer  ->  optim.setup_optimizer: the same Adam groups (utils/tools.py:57-83) as ONE fused launch per
                                     step instead of torch's multi-tensor Adam            SHINE_DROPIN_FUSED_OPTIMIZER
    utils.loss.sdf_bce_loss      ->  losses.sdf_bce_loss: loss and d loss / d pred in ONE launch      SHINE_DROPIN_FUSED_LOSS
    utils.tools.get_gradient     ->  losses.get_gradient: for the fused query_feature -> sdf node ONE forward-kernel launch,
                                     linked to that node so that the eikonal term's backward joins its single fused launch
                                     (lets the eikonal configurations use the fused node)            SHINE_DROPIN_FUSED_GRADIENT

    utils.incre_learning.cal_feature_importance -> incre_learning.cal_feature_importance: the chunk loop of
                                     utils/incre_learning.py:8-40 as two launches per 64 chunks.  The driver binds the name at
                                     shine_incre.py:15 (`from utils.incre_learning import cal_feature_importance`): with the
                                     drop-in imported FIRST (line 1) it picks up this one; names a module bound before the
                                     drop-in was imported are re-bound too (rebind_imported_names: `__main__` and the two
                                     drivers, only where they still hold the reference's original)   SHINE_DROPIN_FUSED_IMPORTANCE

`FeatureOctree.cal_regularization` (shine_incre.py:156) needs no re-binding: it is a method of the replaced class, and runs as one
autograd node over two launches whenever it follows a `query_feature` (autograd_ops.OctreeRegularizer).

The drivers are single-threaded, so for THEM the drop-in also lets autograd run a backward on the CALLING thread
(`torch.autograd.set_multithreading_enabled(False)`: the engine's hand-over to its device thread and back is a third of a BCE
iteration's host time at the reference's batch size — 0.32 -> 0.21 ms, profiles/r05_tier_a_bench.log).  That switch is
process-wide, so it is flipped only when the process IS one of the reference's drivers (`__main__` is shine_batch.py /
shine_incre.py) or SHINE_DROPIN_SINGLE_THREAD_BACKWARD=1 asks for it (importing the drop-in into a process with
other autograd users must not change their threading); =0 never; `uninstall()` restores what it found.

The re-binding needs `utils.tools` / `utils.loss` to be importable when this module is imported (the drivers import it from the
reference's root directory, first line); `patch_utils()` can be called again later, `status()` says what is in place.
"""
import os
import sys
import types

from .decoder import Decoder
from .feature_octree import FeatureOctree

_INSTALLED = False


def install():
    global _INSTALLED
    if _INSTALLED:
        return
    fo = types.ModuleType("model.feature_octree")
    fo.FeatureOctree = FeatureOctree
    fo.__doc__ = "shine_mapping_amd drop-in for model/feature_octree.py"
    de = types.ModuleType("model.decoder")
    de.Decoder = Decoder
    de.__doc__ = "shine_mapping_amd drop-in for model/decoder.py"
    sys.modules["model.feature_octree"] = fo
    sys.modules["model.decoder"] = de
    pkg = sys.modules.get("model")
    if pkg is None:
        try:  # the reference's own `model` package (for its other modules), if it is importable from here
            import model as pkg  # noqa: F401
        except ImportError:
            pkg = types.ModuleType("model")
            pkg.__path__ = []
            sys.modules["model"] = pkg
    pkg.feature_octree = fo
    pkg.decoder = de
    _INSTALLED = True
    if _single_thread_backward_wanted():
        import torch

        _STATUS_KEEP["autograd_multithreading"] = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)
    patch_utils()


_STATUS = {}
_STATUS_KEEP = {}


def status():
    """what patch_utils() re-bound (name -> True / the reason it did not)"""
    import torch

    return dict(_STATUS, single_thread_backward=not torch.autograd.is_multithreading_enabled())


def _on(var):
    return os.environ.get(var, "1") != "0"


DRIVER_FILES = ("shine_batch.py", "shine_incre.py")


def _single_thread_backward_wanted():
    """process-wide switch: on for the reference's own (single-threaded) drivers, or when asked for explicitly"""
    v = os.environ.get("SHINE_DROPIN_SINGLE_THREAD_BACKWARD")
    if v is not None:
        return v != "0"
    main = sys.modules.get("__main__")
    return os.path.basename(getattr(main, "__file__", "") or "") in DRIVER_FILES


_REBOUND = []  # (module, name, original) of rebind_imported_names, for uninstall()


def rebind_imported_names(triples):
    """`from utils.x import name` copies a function into the importing module: a driver (or `__main__`) that ran those imports
    BEFORE the drop-in was imported still holds the reference's originals.  triples: [(name, original, replacement)]; module
    attribute `name` is re-bound only where it still IS the original.  -> number of names re-bound"""
    mods = [m for k, m in list(sys.modules.items())
            if m is not None and (k == "__main__" or k in ("shine_batch", "shine_incre"))]
    n = 0
    for mod in mods:
        for name, orig, new in triples:
            if new is not orig and vars(mod).get(name) is orig:
                setattr(mod, name, new)
                _REBOUND.append((mod, name, orig))
                n += 1
    return n


def patch_utils():
    """Re-bind utils.tools.setup_optimizer / get_gradient and utils.loss.sdf_bce_loss (see the module docstring)."""
    import importlib

    from . import autograd_ops, losses, optim

    try:
        ut = importlib.import_module("utils.tools")
    except Exception as e:  # not importable from here (another working directory, a missing dependency of the reference)
        ut = None
        _STATUS["utils.tools"] = "not importable: %r" % (e,)
    try:
        ul = importlib.import_module("utils.loss")
    except Exception as e:
        ul = None
        _STATUS["utils.loss"] = "not importable: %r" % (e,)
    if ut is not None and not hasattr(ut, "_shine_reference"):
        ut._shine_reference = {"setup_optimizer": ut.setup_optimizer, "get_gradient": ut.get_gradient}
    if ul is not None and not hasattr(ul, "_shine_reference"):
        ul._shine_reference = {"sdf_bce_loss": ul.sdf_bce_loss}
    if ut is not None:
        ref_setup = ut._shine_reference["setup_optimizer"]
        if _on("SHINE_DROPIN_FUSED_OPTIMIZER"):
            def setup_optimizer(config, octree_feat, mlp_geo_param, mlp_sem_param, sigma_size):
                tensors = list(octree_feat) + list(mlp_geo_param or [])
                if (getattr(config, "opt_adam", True) and not getattr(config, "semantic_on", False)
                        and not getattr(config, "ray_loss", False) and tensors
                        and all(p.is_cuda and p.dtype.is_floating_point and p.element_size() == 4 for p in tensors)):
                    return optim.setup_optimizer(config, octree_feat, mlp_geo_param, mlp_sem_param, sigma_size)
                return ref_setup(config, octree_feat, mlp_geo_param, mlp_sem_param, sigma_size)

            setup_optimizer.__doc__ = "shine_mapping_amd drop-in for utils.tools.setup_optimizer (utils/tools.py:57-83)"
            ut.setup_optimizer = setup_optimizer
            _STATUS["setup_optimizer"] = True
        else:
            ut.setup_optimizer = ref_setup
            _STATUS["setup_optimizer"] = "off (SHINE_DROPIN_FUSED_OPTIMIZER=0)"
        if _on("SHINE_DROPIN_FUSED_GRADIENT"):
            ut.get_gradient = losses.get_gradient
            autograd_ops.FUSE_WITH_COORD_GRAD = True
            _STATUS["get_gradient"] = True
        else:
            ut.get_gradient = ut._shine_reference["get_gradient"]
            autograd_ops.FUSE_WITH_COORD_GRAD = False
            _STATUS["get_gradient"] = "off (SHINE_DROPIN_FUSED_GRADIENT=0)"
    try:  # (imports dataset.lidar_dataset, i.e. open3d: present wherever the drivers themselves run)
        ui = importlib.import_module("utils.incre_learning")
    except Exception as e:
        ui = None
        _STATUS["utils.incre_learning"] = "not importable: %r" % (e,)
    if ui is not None:
        if not hasattr(ui, "_shine_reference"):
            ui._shine_reference = {"cal_feature_importance": ui.cal_feature_importance}
        ref_sweep = ui._shine_reference["cal_feature_importance"]
        if _on("SHINE_DROPIN_FUSED_IMPORTANCE"):
            from . import incre_learning

            def cal_feature_importance(data, octree, mlp, sigma, bs, down_rate=1, loss_reduction='mean', loss_weight_on=False):
                if (isinstance(octree, FeatureOctree) and isinstance(mlp, Decoder) and mlp.fusable and len(octree.hier_features)
                        and octree.hier_features[0].is_cuda and octree.featured_level_num <= 4):
                    return incre_learning.cal_feature_importance(data, octree, mlp, sigma, bs, down_rate, loss_reduction,
                                                                 loss_weight_on)
                return ref_sweep(data, octree, mlp, sigma, bs, down_rate, loss_reduction, loss_weight_on)

            cal_feature_importance.__doc__ = "shine_mapping_amd drop-in for utils.incre_learning.cal_feature_importance (:8-40)"
            ui.cal_feature_importance = cal_feature_importance
            _STATUS["cal_feature_importance"] = True
        else:
            ui.cal_feature_importance = ref_sweep
            _STATUS["cal_feature_importance"] = "off (SHINE_DROPIN_FUSED_IMPORTANCE=0)"
    if ul is not None:
        if _on("SHINE_DROPIN_FUSED_LOSS"):
            ul.sdf_bce_loss = losses.sdf_bce_loss
            _STATUS["sdf_bce_loss"] = True
        else:
            ul.sdf_bce_loss = ul._shine_reference["sdf_bce_loss"]
            _STATUS["sdf_bce_loss"] = "off (SHINE_DROPIN_FUSED_LOSS=0)"
    # names the importing module copied before this ran (drop-in imported after `from utils.tools import *`, shine_incre.py:13-15)
    triples = []
    for mod in (ut, ul, ui):
        for attr, orig in getattr(mod, "_shine_reference", {}).items():
            triples.append((attr, orig, getattr(mod, attr)))
    _STATUS["names_rebound_in_loaded_drivers"] = rebind_imported_names(triples)
    return status()


def uninstall():
    global _INSTALLED
    for name in ("model.feature_octree", "model.decoder"):
        mod = sys.modules.get(name)
        if mod is not None and (mod.__doc__ or "").startswith("shine_mapping_amd drop-in"):
            del sys.modules[name]
    pkg = sys.modules.get("model")
    for attr in ("feature_octree", "decoder"):
        if pkg is not None and (getattr(getattr(pkg, attr, None), "__doc__", "") or "").startswith("shine_mapping_amd"):
            delattr(pkg, attr)
    for name in ("utils.tools", "utils.loss", "utils.incre_learning"):  # the reference's own functions back in place
        mod = sys.modules.get(name)
        for attr, fn in getattr(mod, "_shine_reference", {}).items():
            setattr(mod, attr, fn)
    while _REBOUND:
        mod, name, orig = _REBOUND.pop()
        setattr(mod, name, orig)
    from . import autograd_ops

    autograd_ops.FUSE_WITH_COORD_GRAD = False
    if "autograd_multithreading" in _STATUS_KEEP:
        import torch

        torch.autograd.set_multithreading_enabled(_STATUS_KEEP.pop("autograd_multithreading"))
    _STATUS.clear()
    _INSTALLED = False


install()
