"""rbc_gym on the MI355X-native stepper: same env ids, kwargs and spaces as the reference
(src/rbc_gym/__init__.py:4-38); the Julia/Oceananigans simulation behind them is replaced by
librbc_hip.so (include/rbc_hip.h)."""
import numpy as np

from ._gym import gym, HAVE_GYMNASIUM  # noqa: F401

register = gym.envs.registration.register if HAVE_GYMNASIUM else gym.register

register(
    id="rbc_gym/RayleighBenardConvection2D-v0",
    entry_point="rbc_gym.envs:RayleighBenardConvection2DEnv",
    vector_entry_point="rbc_gym.vector:RayleighBenardConvection2DVectorEnv",
    kwargs={
        "rayleigh_number": 10_000,
        "episode_length": 300,
        "observation_shape": (8, 48),
        "state_shape": (64, 96),
        "heater_segments": 12,
        "heater_limit": 0.75,
        "heater_duration": 1.5,
        "checkpoint": None,
        "use_gpu": False,
        "render_mode": None,
    },
)

register(
    id="rbc_gym/RayleighBenardConvection3D-v0",
    entry_point="rbc_gym.envs:RayleighBenardConvection3DEnv",
    kwargs={
        "rayleigh_number": 500,
        "prandtl_number": 0.7,
        "domain": [2, 4 * np.pi, 4 * np.pi],
        "state_shape": (16, 32, 32),
        "temperature_difference": [1, 2],
        "heater_segments": 8,
        "heater_limit": 0.9,
        "heater_duration": 0.125,
        "episode_length": 300,
        "checkpoint": None,
        "use_gpu": False,
        "render_mode": None,
    },
)
