"""Single-env classes (batch-of-one handles); the batched env lives in `rbc_gym.vector`."""
from .rbc2D import RayleighBenardConvection2DEnv, RBCField
from .rbc3D import RayleighBenardConvection3DEnv

__all__ = ["RayleighBenardConvection2DEnv", "RayleighBenardConvection3DEnv", "RBCField"]
