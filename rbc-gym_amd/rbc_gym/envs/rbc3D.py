"""Placeholder for the 3D env id (reference: src/rbc_gym/envs/rbc3D.py:37-339).

The 3D stepper (BASELINE.json configs[4]) is not built yet: constructing the env fails loudly
instead of silently falling back to anything else."""
from .._gym import gym


class RayleighBenardConvection3DEnv(gym.Env):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "rbc_gym/RayleighBenardConvection3D-v0: the MI355X-native 3D stepper is not implemented in this build "
            "(2D only); see DESIGN.md 'Out of scope / next'.")
