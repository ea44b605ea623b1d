"""`RayleighBenardConvection2DEnv` on the MI355X-native stepper.

Drop-in for the reference class of the same name (src/rbc_gym/envs/rbc2D.py:29-266): constructor kwargs,
attributes read by callers and wrappers, spaces, reset/step conventions and info keys are the reference's; the
Julia plugin calls (rbc2D.py:143,169,185,192,199,203-205) are calls into librbc_hip.so through a batch-of-one
`rbc_gym._native.NativeSim`.  The episode logic lives in `_common.NativeEnvBase`.
"""
from typing import Optional

import numpy as np

from .._gym import gym
from .. import _native
from ._common import NativeEnvBase, RBCField, _Window, box_stack, env_logger, pick_checkpoint_episode, temperature_image  # noqa: F401

INF = float("inf")


def colormap(value, vmin=1, vmax=2, colormap="turbo"):
    """`value` scaled to [vmin, vmax] through a matplotlib colour map, RGB bytes (rbc2D.py helper of the same name)."""
    import matplotlib
    return matplotlib.colormaps[colormap]((value - vmin) / (vmax - vmin), bytes=True)[:, :, :3]


def sim_kwargs(rayleigh_number, observation_shape, state_shape, heater_segments, heater_limit, heater_duration):
    """gym kwargs -> rbc_config fields.  The reference passes shapes to Julia reversed, (x, z) (rbc2D.py:143-153);
    the C ABI names the axes instead."""
    (nz, nx), (oz, ox) = state_shape, observation_shape
    return {"ra": float(rayleigh_number), "nx": int(nx), "nz": int(nz), "obs_nx": int(ox), "obs_nz": int(oz),
            "heaters": int(heater_segments), "heater_limit": float(heater_limit), "dt_control": float(heater_duration)}


def build_spaces(observation_shape, heater_segments, heater_limit, pressure):
    """Actions in [-1, 1] per heater segment; observations: T in [1, 2 + heater_limit], velocities (and the two
    pressure channels with pressure=True) unbounded (rbc2D.py:74-108)."""
    bounds = [(1.0, 2.0 + heater_limit)] + [(-INF, INF)] * (4 if pressure else 2)
    return gym.spaces.Box(-1, 1, shape=(heater_segments,), dtype=np.float32), box_stack(observation_shape, bounds)


class RayleighBenardConvection2DEnv(NativeEnvBase):
    def __init__(
        self,
        rayleigh_number: Optional[int] = 10_000,
        episode_length: Optional[int] = 300,
        observation_shape: Optional[list] = [8, 48],
        state_shape: Optional[list] = [64, 96],
        heater_segments: Optional[int] = 12,
        heater_limit: Optional[float] = 0.75,
        heater_duration: Optional[float] = 1.5,
        pressure: Optional[bool] = False,
        use_gpu: Optional[bool] = False,
        checkpoint: Optional[str] = None,
        render_mode: Optional[str] = None,
        device: int = 0,
        precision: str = "f64",          # "f64": the reference's Float64 arithmetic; "f32": the float32 variant
        reference_clock: str = "documented",   # "recorded": the solver-step count of the reference's recorded series
    ) -> None:
        super().__init__()
        self.ra = rayleigh_number
        self.precision = precision
        self.reference_clock = reference_clock
        self.observation_shape, self.state_shape = list(observation_shape), list(state_shape)
        self.temperature_difference = [1, 2]                      # plate temperatures fixed by the API layer (rbc_sim2D_api.jl:30-31)
        self.heater_segments, self.heater_limit, self.heater_duration = heater_segments, heater_limit, heater_duration
        self.include_pressure = pressure
        self.use_gpu = use_gpu                                    # signature compatibility: this backend is the GPU
        self.episode_steps = int(episode_length / heater_duration)
        self.action_space, self.observation_space = build_spaces(self.observation_shape, heater_segments, heater_limit, pressure)
        self._channels = 5 if pressure else 3
        self.screen_width, self.screen_height = 768, 512
        self._window = _Window((self.screen_width, self.screen_height), self.metadata["render_fps"])
        sim = _native.NativeSim(batch=1, device=device, precision=_native.PRECISIONS[precision], reference_clock=reference_clock,
                                **sim_kwargs(rayleigh_number, self.observation_shape, self.state_shape, heater_segments, heater_limit, heater_duration))
        self._setup(sim, episode_length, checkpoint, render_mode, env_logger(__name__))

    # the window handle under the reference's attribute names
    screen = property(lambda self: self._window.screen)
    clock = property(lambda self: self._window.clock)

    def _state(self):
        return self.sim.get_state(self._channels)[0]

    def _observe(self):
        return self.sim.get_obs(self._channels)[0]

    def _reward(self):
        return -float(self.sim.get_nusselt()[1][0])               # minus the Nusselt number on the sensor grid (rbc2D.py:198-200)

    def _info(self):
        (t,), (step,) = self.sim.get_info()
        (nu_state,), (nu_obs,) = self.sim.get_nusselt()
        return {"t": float(t), "step": int(step), "nusselt_state": float(nu_state), "nusselt_obs": float(nu_obs),
                "state": self._state()}

    def render(self):
        if self.render_mode is None:
            gym.logger.warn("You are calling render method without specifying any render mode. "
                            "You can specify the render_mode at initialization, ")
            return None
        image = temperature_image(self._state()[RBCField.T], 1, 2 + self.heater_limit)
        if self.render_mode == "rgb_array":
            return image.transpose(1, 0, 2)
        if self.render_mode == "human":
            self._window.show(image, (self.state_shape[1], self.state_shape[0]))
            return None
        raise ValueError(f"Unknown render mode: {self.render_mode}")
