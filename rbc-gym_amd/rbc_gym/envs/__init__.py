from rbc_gym.envs.rbc2D import RayleighBenardConvection2DEnv
from rbc_gym.envs.rbc3D import RayleighBenardConvection3DEnv

__all__ = ["RayleighBenardConvection2DEnv", "RayleighBenardConvection3DEnv"]
