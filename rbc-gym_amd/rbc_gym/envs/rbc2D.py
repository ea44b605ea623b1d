"""gym.Env for 2D Rayleigh-Benard convection on the MI355X-native stepper.

Drop-in for the reference class of the same name (src/rbc_gym/envs/rbc2D.py:29-266): same
constructor kwargs, attributes, spaces, reset/step return conventions and info keys.  Where
the reference calls the Julia plugin API through juliacall (rbc2D.py:143,169,185,192,199,
203-205) this class calls librbc_hip.so through `rbc_gym._native.NativeSim` (batch of 1).
"""
import logging
import warnings
from enum import IntEnum
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import numpy as np

from .._gym import gym
from .. import _native
from ..checkpoint import read_checkpoint


class RBCField(IntEnum):
    T = 0
    UX = 1
    UY = 2
    P = 3


def colormap(value, vmin=1, vmax=2, colormap="turbo"):
    import matplotlib
    cmap = matplotlib.colormaps[colormap]
    value = (value - vmin) / (vmax - vmin)
    return cmap(value, bytes=True)[:, :, :3]


def sim_kwargs(rayleigh_number, observation_shape, state_shape, heater_segments, heater_limit, heater_duration):
    """kwargs of initialize_simulation (rbc2D.py:143-153): shapes are passed reversed (x, z)."""
    return dict(ra=float(rayleigh_number), nx=int(state_shape[1]), nz=int(state_shape[0]),
                obs_nx=int(observation_shape[1]), obs_nz=int(observation_shape[0]),
                heaters=int(heater_segments), heater_limit=float(heater_limit), dt_control=float(heater_duration))


def build_spaces(observation_shape, heater_segments, heater_limit, pressure):
    """action / observation spaces exactly as rbc2D.py:74-108."""
    action_space = gym.spaces.Box(-1, 1, shape=(heater_segments,), dtype=np.float32)
    shape = tuple(observation_shape)
    lows = [np.ones(shape) * 1, np.ones(shape) * (-np.inf), np.ones(shape) * (-np.inf)]
    highs = [np.ones(shape) * 2 + heater_limit, np.ones(shape) * np.inf, np.ones(shape) * np.inf]
    channels = 3
    if pressure:
        channels += 2
        lows += [np.ones(shape) * (-np.inf)] * 2
        highs += [np.ones(shape) * np.inf] * 2
    observation_space = gym.spaces.Box(np.stack(lows, axis=0).astype(np.float32), np.stack(highs, axis=0).astype(np.float32),
                                       shape=(channels, shape[0], shape[1]), dtype=np.float32)
    return action_space, observation_space


def pick_checkpoint_episode(num_episodes, seed):
    """The reference draws `idx = rand(1:n)` from Julia's global RNG right after
    `Random.seed!(seed)` (rbc_sim2D_api.jl:44, rbc_sim2D.jl:176-177).  That stream cannot be
    reproduced outside Julia; this is the documented stand-in: deterministic in the seed."""
    return int(np.random.default_rng(int(seed) % (2**63)).integers(0, num_episodes))


class RayleighBenardConvection2DEnv(gym.Env):
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 10}

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
    ) -> None:
        super().__init__()
        self.closed = False
        self.use_gpu = use_gpu          # kept for signature compatibility: this backend always runs on the GPU
        self.checkpoint = checkpoint

        # Environment configuration (rbc2D.py:57-67)
        self.ra = rayleigh_number
        self.episode_length = episode_length
        self.observation_shape = list(observation_shape)
        self.state_shape = list(state_shape)
        self.temperature_difference = [1, 2]
        self.heater_segments = heater_segments
        self.heater_limit = heater_limit
        self.heater_duration = heater_duration
        self.include_pressure = pressure
        self.episode_steps = int(episode_length / heater_duration)

        self.logger = logging.getLogger(__name__)
        self.logger.info(f"Using Rayleigh number Ra={self.ra}")
        self.logger.info(f"Using episode length {self.episode_length} timesteps")

        self.action_space, self.observation_space = build_spaces(self.observation_shape, heater_segments, heater_limit, pressure)

        # the native simulation (replaces juliapkg.resolve + juliacall.newmodule + include, rbc2D.py:111-115)
        self.sim = _native.NativeSim(batch=1, device=device,
                                     **sim_kwargs(rayleigh_number, self.observation_shape, self.state_shape,
                                                  heater_segments, heater_limit, heater_duration))
        self._nch = 5 if pressure else 3
        self._ckpt = None

        # Rendering
        self.render_mode = render_mode
        self.screen_width = 768
        self.screen_height = 512
        self.screen = None
        self.clock = None

    def reset(self, seed: int | None = None, options: Dict[str, Any] | None = None) -> Tuple[Any, Dict[str, Any]]:
        super().reset(seed=seed)
        seed_used = self.np_random_seed      # quirk kept: reset(seed=None) re-uses the same seed (rbc2D.py:150)
        if self.checkpoint:
            path = Path(self.checkpoint)
            self.logger.info(f"Using checkpoint file {path.absolute()}")
            if not path.exists():
                raise FileNotFoundError(f"Checkpoint file {path} does not exist. Please provide a valid checkpoint directory.")
            if self._ckpt is None or self._ckpt[0] != str(path.absolute()):
                self._ckpt = (str(path.absolute()), read_checkpoint(path))
            ck = self._ckpt[1]
            idx = pick_checkpoint_episode(ck["num_episodes"], seed_used)
            self.sim.reset_from_arrays(ck["b"][idx:idx + 1], ck["u"][idx:idx + 1], ck["w"][idx:idx + 1])
        else:
            self.sim.reset(np.array([seed_used % (2**64)], dtype=np.uint64))
        self.last_action = self.action_space.sample() * 0
        return self.__get_obs(), self.__get_info()

    def step(self, action: Any = None) -> Tuple[Any, float, bool, bool, Dict[str, Any]]:
        terminated = False  # is always false; no terminal state
        truncated = False
        if action is None:
            action = np.zeros(self.action_space.shape, dtype=np.float32)
            warnings.warn("No action provided, using zero action")

        success = self.sim.step(np.asarray(action, dtype=np.float32).reshape(1, -1))
        if not success:
            raise RuntimeError("Error in simulation step, probably NaN values")

        self.last_obs = self.__get_obs()
        self.last_reward = self.__get_reward()
        self.last_info = self.__get_info()
        if self.last_info["t"] >= self.episode_length:
            truncated = True
        return self.last_obs, self.last_reward, terminated, truncated, self.last_info

    def __get_state(self) -> Any:
        return self.sim.get_state(self._nch)[0]

    def __get_obs(self) -> Any:
        return self.sim.get_obs(self._nch)[0]

    def __get_reward(self) -> float:
        return -float(self.sim.get_nusselt()[1][0])

    def __get_info(self) -> dict[str, Any]:
        t, step = self.sim.get_info()
        nu_state, nu_obs = self.sim.get_nusselt()
        return {"t": float(t[0]), "step": int(step[0]), "nusselt_state": float(nu_state[0]),
                "nusselt_obs": float(nu_obs[0]), "state": self.__get_state()}

    def render(self):
        if self.render_mode is None:
            gym.logger.warn("You are calling render method without specifying any render mode. "
                            "You can specify the render_mode at initialization, ")
            return
        data = self.__get_state()[RBCField.T]
        data = np.transpose(data)
        data = np.flip(data, axis=1)  # origin is top left
        data = colormap(data, vmin=1, vmax=2 + self.heater_limit)
        if self.render_mode == "rgb_array":
            return data.transpose(1, 0, 2)
        if self.render_mode == "human":
            try:
                import pygame
            except ImportError as e:      # display-only path of the reference (rbc2D.py:241-255)
                raise RuntimeError("render_mode='human' needs pygame, which is not installed") from e
            if self.screen is None:
                pygame.init()
                pygame.display.init()
                self.screen = pygame.display.set_mode((self.screen_width, self.screen_height))
                pygame.display.set_caption("Rayleigh Benard Convection")
            if self.clock is None:
                self.clock = pygame.time.Clock()
            canvas = pygame.Surface((self.state_shape[1], self.state_shape[0]))
            pygame.surfarray.blit_array(canvas, data)
            canvas = pygame.transform.scale(canvas, (self.screen_width, self.screen_height))
            self.screen.blit(canvas, (0, 0))
            pygame.event.pump()
            self.clock.tick(self.metadata["render_fps"])
            pygame.display.flip()
            return None
        raise ValueError(f"Unknown render mode: {self.render_mode}")

    def close(self):
        if self.screen is not None:
            import pygame
            pygame.display.quit()
            pygame.quit()
            self.screen = None
        if getattr(self, "sim", None) is not None:
            self.sim.close()
            self.sim = None
        self.closed = True
