"""gym.Env for 3D Rayleigh-Benard convection on the MI355X-native stepper.

Drop-in for the reference class (src/rbc_gym/envs/rbc3D.py:37-339): same constructor kwargs,
attributes, spaces, reset/step conventions and info keys {t, step, nusselt}; the Julia calls
(rbc3D.py:175,206,230,235,239-240,328-329) become librbc_hip.so calls on a dim=3 handle.
PyVista rendering (display only) is not reproduced.
"""
import logging
import warnings
from os.path import join
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np

from .._gym import gym
from .. import _native
from ..checkpoint import read_checkpoint
from .rbc2D import pick_checkpoint_episode


def build_spaces3d(state_shape, temperature_difference, heater_segments, heater_limit):
    """rbc3D.py:102-142"""
    action_space = gym.spaces.Box(-1, 1, shape=(heater_segments, heater_segments), dtype=np.float32)
    shape = tuple(state_shape)
    lows = np.stack([np.full(shape, temperature_difference[0]), np.full(shape, -np.inf), np.full(shape, -np.inf),
                     np.full(shape, -np.inf)], axis=0).astype(np.float32)
    highs = np.stack([np.full(shape, temperature_difference[1] + heater_limit), np.full(shape, np.inf), np.full(shape, np.inf),
                      np.full(shape, np.inf)], axis=0).astype(np.float32)
    observation_space = gym.spaces.Box(lows, highs, shape=(4,) + shape, dtype=np.float32)
    return action_space, observation_space


class RayleighBenardConvection3DEnv(gym.Env):
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 10}

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
    ) -> None:
        super().__init__()
        self.closed = False
        self.use_gpu = use_gpu
        self.checkpoint = checkpoint
        self.checkpoint_idx = checkpoint_idx

        self.ra = rayleigh_number
        self.pr = prandtl_number
        self.domain = domain
        self.episode_length = episode_length
        self.dt_solver = dt_solver
        self.state_shape = state_shape
        self.temperature_difference = temperature_difference
        self.heater_segments = heater_segments
        self.heater_limit = heater_limit
        self.heater_duration = heater_duration

        self.logger = logging.getLogger(f"{__name__}.{env_id}")
        if log_dir is not None:                       # per-env file logger (rbc3D.py:84-99)
            fh = logging.FileHandler(join(log_dir, f"env_{env_id}.log"))
            fh.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
            self.logger.addHandler(fh)
        self.logger.info(f"Using Rayleigh number Ra={self.ra}")
        self.logger.info(f"Using episode length {self.episode_length} timesteps")

        self.action_space, self.observation_space = build_spaces3d(state_shape, temperature_difference, heater_segments, heater_limit)

        self.sim = _native.NativeSim3D(batch=1, device=device, shape=tuple(state_shape), domain=tuple(domain), ra=float(rayleigh_number),
                                       pr=float(prandtl_number), t_diff=tuple(temperature_difference), heaters=heater_segments,
                                       heater_limit=heater_limit, dt_control=heater_duration, dt_solver=dt_solver)
        self.render_mode = render_mode
        self._ckpt = None

    def reset(self, seed: int | None = None, options: Dict[str, Any] | None = None) -> Tuple[Any, Dict[str, Any]]:
        super().reset(seed=seed)
        seed_used = self.np_random_seed
        if self.checkpoint:
            path = Path(self.checkpoint)
            self.logger.info(f"Using checkpoint file {path.absolute()}")
            if not path.exists():
                raise FileNotFoundError(f"Checkpoint file {path} does not exist. Please provide a valid checkpoint directory.")
            if self._ckpt is None:
                self._ckpt = read_checkpoint(path)
            ck = self._ckpt
            if "v" not in ck:
                raise ValueError(f"{path}: not a 3D checkpoint (no dataset 'v')")
            idx = self.checkpoint_idx if self.checkpoint_idx is not None else pick_checkpoint_episode(ck["num_episodes"], seed_used)
            self.sim.reset_from_arrays(ck["b"][idx:idx + 1], ck["u"][idx:idx + 1], ck["v"][idx:idx + 1], ck["w"][idx:idx + 1])
        else:
            self.sim.reset(np.array([seed_used % (2**64)], dtype=np.uint64))
        self.last_action = self.action_space.sample() * 0
        return self.__get_obs(), self.__get_info()

    def step(self, action: Any = None) -> Tuple[Any, float, bool, bool, Dict[str, Any]]:
        terminated = False
        truncated = False
        if action is None:
            action = np.zeros(self.action_space.shape, dtype=np.float32)
            warnings.warn("No action provided, using zero action")
        a = np.asarray(action, dtype=np.float32)
        if a.shape != self.action_space.shape:       # preprocess_action's size check (rbc_sim3D.jl:115-117)
            raise ValueError(f"Action size does not match the number of actuators. Expected {self.action_space.shape}, got {a.shape}.")
        success = self.sim.step(a[None])
        if not success:
            self.logger.error("Simulation step failed, probably NaN values in the simulation.")
            raise RuntimeError("Error in simulation step, probably NaN values")
        self.last_obs = self.__get_obs()
        self.last_reward = self.__get_reward()
        self.last_info = self.__get_info()
        if self.last_info["t"] >= self.episode_length:
            truncated = True
        return self.last_obs, self.last_reward, terminated, truncated, self.last_info

    def __get_obs(self) -> Any:
        return self.sim.get_state()[0]

    def __get_reward(self) -> float:
        return -float(self.sim.get_nusselt()[0])

    def __get_info(self) -> dict[str, Any]:
        t, step = self.sim.get_info()
        return {"t": float(t[0]), "step": int(step[0]), "nusselt": float(self.sim.get_nusselt()[0])}

    def render(self):
        if self.render_mode is None:
            return None
        raise NotImplementedError("PyVista volume rendering of the reference (display only) is not part of this build")

    def close(self):
        if self.closed:
            return
        self.closed = True
        if getattr(self, "sim", None) is not None:   # shutdown_simulation (rbc_sim3D_api.jl:164-172)
            self.sim.close()
            self.sim = None
