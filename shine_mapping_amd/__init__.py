"""shine_mapping_amd — the SHINE-Mapping SDF training hot path, MI355X (gfx950) native.

Public surface mirrors the reference's modules for this path:
    FeatureOctree   model/feature_octree.py
    Decoder         model/decoder.py
    sdf_bce_loss, get_gradient          utils/loss.py, utils/tools.py
    train_step / StepOptions            the fused Tier-B step (one HIP pass) as a torch.autograd.Function
    fused_train_step                    the same launch in raw form (grads written straight into .grad)
All compute goes through libshine_hip.so (include/shine_hip.h); there is no CPU fallback.
"""
from .decoder import Decoder
from .feature_octree import FeatureOctree
from .losses import get_gradient, sdf_bce_loss
from .ops import StepOptions, forward_sdf, fused_train_step, octree_interp, train_step

__all__ = ["Decoder", "FeatureOctree", "StepOptions", "forward_sdf", "fused_train_step", "train_step", "octree_interp",
           "sdf_bce_loss", "get_gradient"]
