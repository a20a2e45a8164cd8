"""Zero-edit drop-in for the reference's drivers ("packages with those import paths").

    import shine_mapping_amd.dropin        # ONE line at the top of shine_batch.py / shine_incre.py

registers this package's classes under the module paths the reference imports them from
(`shine_batch.py:13-14`, `shine_incre.py:12-13`, `utils/mesher.py:11-12`, `utils/incre_learning.py`):

    model.feature_octree.FeatureOctree  ->  shine_mapping_amd.FeatureOctree
    model.decoder.Decoder               ->  shine_mapping_amd.Decoder

Only these two modules are replaced — `utils.*`, `dataset.*` and the rest of `model.*` keep resolving to the
reference's own files.  A side effect worth having: the reference's `model/feature_octree.py` (and with it the
kaolin import at `:6`) is never executed on the training path.
"""
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
    _INSTALLED = False


install()
