"""`RayleighBenardConvection3DEnv` on the MI355X-native stepper.

Drop-in for the reference class (src/rbc_gym/envs/rbc3D.py:37-339): constructor kwargs, attributes, spaces,
reset/step conventions and info keys {t, step, nusselt}; its Julia calls (rbc3D.py:175,206,230,235,239-240,328-329)
become calls on a dim=3 handle of librbc_hip.so.  The observation is the full float32 state (4, Nz, Ny, Nx).
PyVista volume rendering (display only) is not reproduced.
"""
from typing import Optional

import numpy as np

from .._gym import gym
from .. import _native
from ._common import NativeEnvBase, box_stack, env_logger

INF = float("inf")


def build_spaces3d(state_shape, temperature_difference, heater_segments, heater_limit):
    """heater_segments x heater_segments actions in [-1, 1]; T bounded by the plates (+ heater_limit), velocities
    unbounded (rbc3D.py:102-142)."""
    t_cold, t_hot = temperature_difference
    obs = box_stack(state_shape, [(t_cold, t_hot + heater_limit)] + [(-INF, INF)] * 3)
    return gym.spaces.Box(-1, 1, shape=(heater_segments, heater_segments), dtype=np.float32), obs


class RayleighBenardConvection3DEnv(NativeEnvBase):
    _fields = ("b", "u", "v", "w")

    def __init__(
        self,
        rayleigh_number: Optional[int] = 2500,
        prandtl_number: Optional[float] = 0.7,
        domain: Optional[list] = [2, 4 * np.pi, 4 * np.pi],
        state_shape: Optional[list] = (16, 32, 32),
        temperature_difference: Optional[list] = [1, 2],
        heater_segments: Optional[int] = 8,
        heater_limit: Optional[float] = 0.9,
        heater_duration: Optional[float] = 0.125,
        episode_length: Optional[int] = 300,
        dt_solver: Optional[float] = 0.01,
        use_gpu: Optional[bool] = False,
        checkpoint: Optional[str] = None,
        checkpoint_idx: Optional[int] = None,
        render_mode: Optional[str] = None,
        log_dir: str = None,
        env_id: int = 0,
        device: int = 0,
        precision: str = "f64",          # "f64": the reference's Float64 arithmetic; "f32": the float32 instantiation of the 3D kernels
        reference_clock: str = "documented",   # "recorded": the solver-step count of the reference's recorded series
    ) -> None:
        super().__init__()
        self.precision = precision
        self.reference_clock = reference_clock
        self.ra, self.pr = rayleigh_number, prandtl_number
        self.domain, self.state_shape = domain, state_shape
        self.temperature_difference = temperature_difference
        self.heater_segments, self.heater_limit, self.heater_duration = heater_segments, heater_limit, heater_duration
        self.dt_solver = dt_solver
        self.use_gpu = use_gpu
        self.checkpoint_idx = checkpoint_idx
        self.action_space, self.observation_space = build_spaces3d(state_shape, temperature_difference, heater_segments, heater_limit)
        sim = _native.NativeSim3D(batch=1, device=device, shape=tuple(state_shape), domain=tuple(domain), ra=float(rayleigh_number),
                                  pr=float(prandtl_number), t_diff=tuple(temperature_difference), heaters=heater_segments,
                                  heater_limit=heater_limit, dt_control=heater_duration, dt_solver=dt_solver, precision=precision,
                                  reference_clock=reference_clock)
        self._setup(sim, episode_length, checkpoint, render_mode, env_logger(__name__, log_dir, env_id))

    def _checked_action(self, action):
        a = np.asarray(action, dtype=np.float32)
        if a.shape != self.action_space.shape:                    # preprocess_action's size check (rbc_sim3D.jl:115-117)
            raise ValueError(f"Action size does not match the number of actuators. Expected {self.action_space.shape}, got {a.shape}.")
        return a[None]

    def _observe(self):
        return self.sim.get_state()[0]

    def _reward(self):
        return -float(self.sim.get_nusselt()[0])

    def _info(self):
        (t,), (step,) = self.sim.get_info()
        return {"t": float(t), "step": int(step), "nusselt": float(self.sim.get_nusselt()[0])}

    def render(self):
        if self.render_mode is None:
            return None
        raise NotImplementedError("PyVista volume rendering of the reference (display only) is not part of this build")
