"""rbc_gym on the MI355X-native stepper.  Importing the package registers the reference's two env ids with the
reference's default kwargs (src/rbc_gym/__init__.py:4-38); both ids additionally get a vector entry point, so
`gym.make_vec` returns the device-batched env instead of one process per env."""
from math import pi

from ._gym import gym, HAVE_GYMNASIUM  # noqa: F401

_COMMON = {"checkpoint": None, "use_gpu": False, "render_mode": None, "episode_length": 300}

ENV_SPECS = {
    "rbc_gym/RayleighBenardConvection2D-v0": {
        "entry_point": "rbc_gym.envs:RayleighBenardConvection2DEnv",
        "vector_entry_point": "rbc_gym.vector:RayleighBenardConvection2DVectorEnv",
        "kwargs": dict(_COMMON, rayleigh_number=10_000, observation_shape=(8, 48), state_shape=(64, 96), heater_segments=12,
                       heater_limit=0.75, heater_duration=1.5),
    },
    "rbc_gym/RayleighBenardConvection3D-v0": {
        "entry_point": "rbc_gym.envs:RayleighBenardConvection3DEnv",
        "vector_entry_point": "rbc_gym.vector:RayleighBenardConvection3DVectorEnv",
        "kwargs": dict(_COMMON, rayleigh_number=500, prandtl_number=0.7, domain=[2, 4 * pi, 4 * pi], state_shape=(16, 32, 32),
                       temperature_difference=[1, 2], heater_segments=8, heater_limit=0.9, heater_duration=0.125),
    },
}

_register = gym.envs.registration.register if HAVE_GYMNASIUM else gym.register
for _id, _spec in ENV_SPECS.items():
    _register(id=_id, **_spec)
