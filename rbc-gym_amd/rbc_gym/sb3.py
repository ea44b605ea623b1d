"""stable-baselines3 `VecEnv` over the batched on-device env.

The reference trains with SB3: `SubprocVecEnv([make_env(i, ...) for i in range(n_envs)])` -- one process and one Julia
runtime per env (experiments/run_sarl.py:152-153, eval_sarl.py).  `make_sb3_vec_env` returns an object of SB3's own
`VecEnv` class hierarchy (so `PPO(policy, vec_env, ...)` accepts it) whose n_envs envs are ONE library handle on the GPU:

    vec_env = make_sb3_vec_env("rbc_gym/RayleighBenardConvection3D-v0", n_envs=32, rayleigh_number=10000,
                               heater_duration=0.125, normalize=dict(heater_limit=0.9, maxval=1))

SB3's conventions, which differ from gymnasium's vector API and are reproduced here:
  * `reset()` returns the observations only (infos in `reset_infos`), seeds come from `seed()`;
  * `step_async(actions)` / `step_wait()` -> (obs, rewards, dones, infos) with infos a LIST of per-env dicts;
  * SAME-STEP autoreset: an env whose episode ends is reset inside `step_wait`; the returned observation is the first of
    the new episode, the last of the old one travels in `infos[i]["terminal_observation"]`, and
    `infos[i]["TimeLimit.truncated"]` says the end was a truncation (this env never terminates: rbc2D.py:161).
stable-baselines3 is imported when the function is called, not with this module (it is not a dependency of the package).
"""
import numpy as np

from ._gym import gym


def _sb3_vecenv_base():
    from stable_baselines3.common.vec_env.base_vec_env import VecEnv       # noqa: deferred on purpose
    return VecEnv


def make_sb3_vec_env(env_id, n_envs, normalize=None, **env_kwargs):
    """-> SB3 VecEnv of `n_envs` envs of `env_id` stepped as one batch on the GPU.  `normalize`: kwargs of
    RBCNormalizeObservation (heater_limit, maxval, u_limit, eps, clip) applied by the kernel that writes the observations;
    every other keyword goes to the env (`devices=[0, ..., 7]` shards the batch over the GPUs of a node)."""
    VecEnv = _sb3_vecenv_base()
    venv = gym.make_vec(env_id, num_envs=n_envs, **env_kwargs)
    inner = venv
    if normalize is not None:
        from .wrappers import VectorRBCNormalizeObservation
        venv = VectorRBCNormalizeObservation(venv, **normalize)

    class RBCVecEnv(VecEnv):
        def __init__(self):
            # set before the base constructor runs: SB3 2.x's VecEnv.__init__ calls self.get_attr("render_mode"), which reads them
            self.venv, self.batched = venv, inner.unwrapped
            self._actions = None
            super().__init__(n_envs, venv.single_observation_space, venv.single_action_space)
            self.render_mode = getattr(inner.unwrapped, "render_mode", None)

        # -- reset / step -------------------------------------------------------------------------------------------
        def reset(self):
            seeds = getattr(self, "_seeds", None)
            seed = None
            if seeds is not None and any(s is not None for s in seeds):
                seed = [None if s is None else int(s) for s in seeds]
            obs, info = self.venv.reset(seed=seed)
            self.reset_infos = _unstack(info, self.num_envs)
            if hasattr(self, "_reset_seeds"):
                self._reset_seeds()
            if hasattr(self, "_reset_options"):
                self._reset_options()
            return obs

        def step_async(self, actions):
            self._actions = np.asarray(actions, dtype=np.float32)

        def step_wait(self):
            obs, rewards, terminated, truncated, info = self.venv.step(self._actions)
            dones = np.logical_or(terminated, truncated)
            infos = _unstack(info, self.num_envs)
            if dones.any():
                last = obs[dones].copy()
                self.batched._reset_envs(dones.astype(np.uint8))          # same-step autoreset of exactly these envs
                self.batched._autoreset[:] = False
                fresh = self.batched._observations()
                if hasattr(self.venv, "_obs"):                             # the numpy fallback of the normalisation wrapper (a fused one returns its input)
                    fresh = self.venv._obs(fresh)
                obs = np.array(obs, copy=True)
                obs[dones] = fresh[dones]
                for j, i in enumerate(np.nonzero(dones)[0]):
                    infos[i]["terminal_observation"] = last[j]
                    infos[i]["TimeLimit.truncated"] = bool(truncated[i] and not terminated[i])
            return obs, np.asarray(rewards, dtype=np.float32), dones, infos

        # -- housekeeping SB3 expects ---------------------------------------------------------------------------------
        def close(self):
            self.venv.close()

        def get_attr(self, attr_name, indices=None):
            v = getattr(self.batched, attr_name)
            return [v for _ in self._idx(indices)]

        def set_attr(self, attr_name, value, indices=None):
            setattr(self.batched, attr_name, value)

        def env_method(self, method_name, *args, indices=None, **kwargs):
            r = getattr(self.batched, method_name)(*args, **kwargs)
            return [r for _ in self._idx(indices)]

        def env_is_wrapped(self, wrapper_class, indices=None):
            return [False for _ in self._idx(indices)]

        def get_images(self):
            # the 3D batched env has no render() (PyVista display only in the reference) and render_mode may be None
            if not callable(getattr(self.batched, "render", None)) or getattr(self.batched, "render_mode", None) is None:
                return [None] * self.num_envs
            frames = self.batched.render()
            return list(frames) if frames is not None else [None] * self.num_envs

        def _idx(self, indices):
            if indices is None:
                return range(self.num_envs)
            return [indices] if isinstance(indices, int) else list(indices)

    return RBCVecEnv()


def _unstack(info, n):
    """gymnasium's dict of stacked arrays (+ `_key` masks) -> SB3's list of per-env dicts"""
    out = [dict() for _ in range(n)]
    for k, v in info.items():
        if k.startswith("_"):
            continue
        mask = info.get("_" + k)
        for i in range(n):
            if mask is None or mask[i]:
                x = v[i]
                out[i][k] = x.item() if isinstance(x, np.generic) else x
    return out
