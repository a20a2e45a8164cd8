"""Import the REAL reference modules from /root/reference on CPU (authoring container only).

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py and by the
container-only tests that pin oracle/shine_oracle.py against the reference's
own Python.  /root/reference does not exist on the GPU box: nothing that runs
there (``-m gpu`` tests, smoke(), bench.py) may import this module.

The reference imports packages that are absent here; none of their code runs
on the hot path, so they are satisfied with inert stubs:
  kaolin                      -> oracle/kaolin_shim.py (the 5 integer ops)
  open3d, wandb               -> utils/tools.py:15,17
  natsort, pyquaternion       -> dataset/lidar_dataset.py, utils/pose.py
  skimage.measure             -> utils/mesher.py
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("SHINE_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "feature_octree.py"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Make `model.*`, `utils.*` of the reference importable. Returns a namespace of the hot-path symbols."""
    if not available():
        raise RuntimeError("reference checkout not found at %s" % REFERENCE_ROOT)
    from . import kaolin_shim

    kaolin_shim.install()
    o3d = _stub("open3d")
    if not hasattr(o3d, "utility"):
        util = types.SimpleNamespace(random=types.SimpleNamespace(seed=lambda s: None))
        o3d.utility = util
        o3d.geometry = types.SimpleNamespace()
    _stub("wandb")
    _stub("natsort", natsorted=sorted)
    _stub("pyquaternion", Quaternion=object)
    sk = _stub("skimage")
    sk.measure = _stub("skimage.measure")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    from model.feature_octree import FeatureOctree  # noqa: E402
    from model.decoder import Decoder  # noqa: E402
    from utils.config import SHINEConfig  # noqa: E402
    from utils.loss import sdf_bce_loss  # noqa: E402
    from utils.tools import get_gradient, setup_optimizer, freeze_model  # noqa: E402
    from utils.data_sampler import dataSampler  # noqa: E402
    from utils.incre_learning import cal_feature_importance  # noqa: E402

    from utils.mesher import Mesher  # noqa: E402  (only query_points / get_query_from_bbx run; open3d/skimage are stubs)

    return types.SimpleNamespace(
        Mesher=Mesher,
        FeatureOctree=FeatureOctree,
        Decoder=Decoder,
        SHINEConfig=SHINEConfig,
        sdf_bce_loss=sdf_bce_loss,
        get_gradient=get_gradient,
        setup_optimizer=setup_optimizer,
        freeze_model=freeze_model,
        dataSampler=dataSampler,
        cal_feature_importance=cal_feature_importance,
    )
