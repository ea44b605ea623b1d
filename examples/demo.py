#!/usr/bin/env python3
"""Usage tour of the drop-in package on one MI355X: every way the reference's example/ scripts drive `rbc_gym`
(example/run_2D.py, run_vectorized.py, run_wrapped.py, run_checkpoint.py, run_3D.py, timing.py), as sub-commands.

    python examples/demo.py single [steps]          one 2D env, pressure channels, full-grid sensors, frames off-screen
    python examples/demo.py vector [envs] [steps]   gym.make_vec -> ONE device-batched env (6 or 6000 envs alike)
    python examples/demo.py wrapped [steps]         observation / reward normalisation + reward shaping wrappers
    python examples/demo.py checkpoint [file] [steps]   episodes from converged states (generated on the GPU if absent)
    python examples/demo.py three-d [time] [f32] [recorded]   the 3D env (f32: the float32 instantiation of the 3D kernels; recorded: the
                                                    solver-step count of the reference's recorded flowstats series)
    python examples/demo.py timing [iterations]     construction / reset / step latency of one env
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
import rbc_gym  # noqa: E402,F401  (registers the env ids)
from rbc_gym._gym import HAVE_GYMNASIUM, gym  # noqa: E402  (gymnasium itself when it is installed)

ENV2D, ENV3D = "rbc_gym/RayleighBenardConvection2D-v0", "rbc_gym/RayleighBenardConvection3D-v0"


def _int(args, pos, default):
    return int(args[pos]) if len(args) > pos else default


def rollout(env, steps, policy, report_every=20, keys=("nusselt_state",)):
    obs, info = env.reset()
    for n in range(steps):
        obs, reward, terminated, truncated, info = env.step(policy(env))
        if env.render_mode is not None:
            env.render()
        if n % report_every == 0 or truncated:
            extras = "  ".join(f"{k}={info[k]:.4f}" for k in keys if k in info)
            print(f"step {info['step']:4d}  t={info['t']:8.2f}  reward={reward:+.4f}  {extras}")
        if truncated:
            break
    return obs, info


def zero_action(env):
    return env.action_space.sample() * 0


def random_action(env):
    return env.action_space.sample()


def single(args):
    env = gym.make(ENV2D, render_mode="rgb_array", pressure=True, observation_shape=[64, 96])
    obs, _ = rollout(env, _int(args, 0, env.unwrapped.episode_steps), zero_action)
    print("observation", obs.shape, obs.dtype, "frame", env.render().shape)
    env.close()


def vector(args):
    n, steps = _int(args, 0, 6), _int(args, 1, 100)
    # no vectorization_mode: gymnasium then takes the id's vector_entry_point, the device-batched env ("async" would
    # make real gymnasium fork one process and one GPU handle per env, as the reference does with Julia runtimes)
    env = gym.make_vec(ENV2D, num_envs=n, render_mode="rgb_array")
    obs, info = env.reset()
    print(f"Observation shape: {obs.shape}")
    t0 = time.perf_counter()
    done = 0
    for done in range(1, steps + 1):
        obs, reward, terminated, truncated, info = env.step(env.action_space.sample())
        if truncated.any():
            break
    dt = time.perf_counter() - t0
    frames = env.render()
    print(f"{n} envs x {done} steps in {dt:.2f} s = {n * done / dt:.0f} env-steps/s; mean reward {reward.mean():+.4f}; "
          f"{len(frames)} frames {frames[0].shape}")
    env.close()


def wrapped(args):
    from rbc_gym.wrappers import RBCNormalizeObservation, RBCNormalizeReward, RBCRewardShaping
    env = gym.make(ENV2D, render_mode="rgb_array")
    env = RBCRewardShaping(RBCNormalizeReward(RBCNormalizeObservation(env, heater_limit=env.unwrapped.heater_limit)),
                           shaping_weight=0.1)
    if HAVE_GYMNASIUM:                      # the generic gymnasium wrappers the reference stacks on top
        from gymnasium.wrappers import FlattenObservation, FrameStackObservation
        env = FrameStackObservation(FlattenObservation(env), 4)
    obs, _ = rollout(env, _int(args, 0, int(env.unwrapped.episode_length)), zero_action, keys=("nusselt_obs", "cell_dist"))
    print(f"Observation shape: {obs.shape}")
    env.close()


def checkpoint(args):
    from rbc_gym.generate import generate_checkpoints_2d
    path = args[0] if args else "data/checkpoints/train/ckpt_ra10000.h5"
    if not os.path.exists(path):            # a short spin-up here; scripts/create_checkpoints_2D.py writes the t=600 sets
        path = generate_checkpoints_2d(os.path.dirname(path) or ".", ra=1e4, random_inits=4, seed=42,
                                       duration=float(os.environ.get("RBC_SPINUP", "60")))
    env = gym.make(ENV2D, checkpoint=path, render_mode="rgb_array")
    rollout(env, _int(args, 1, env.unwrapped.episode_steps), random_action)
    env.close()


def three_d(args):
    precision = "f32" if "f32" in args else "f64"
    clock = "recorded" if "recorded" in args else "documented"
    args = [a for a in args if a not in ("f32", "recorded")]
    env = gym.make(ENV3D, rayleigh_number=2500, heater_duration=0.25, episode_length=_int(args, 0, 50), precision=precision,
                   reference_clock=clock)
    obs, _ = rollout(env, 10**9, zero_action, keys=("nusselt",))
    print("observation", obs.shape)
    env.close()


def timing(args):
    n = _int(args, 0, 1000)
    t0 = time.perf_counter()
    env = gym.make(ENV2D, heater_duration=1, render_mode=None, use_gpu=True)
    env.reset()
    print(f"library load + handle + first reset: {time.perf_counter() - t0:.2f} s   (reference README: 7.68 s Julia init)")
    t0 = time.perf_counter()
    for _ in range(n):
        env.reset()
    print(f"reset: {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
    t0 = time.perf_counter()
    for _ in range(n):
        env.step(env.action_space.sample())
    print(f"step (dt=1): {(time.perf_counter() - t0) / n * 1e3:.3f} ms   (reference README: 120 ms on one CPU core)")
    env.close()


COMMANDS = {"single": single, "vector": vector, "wrapped": wrapped, "checkpoint": checkpoint, "three-d": three_d, "timing": timing}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        raise SystemExit(__doc__)
    COMMANDS[sys.argv[1]](sys.argv[2:])
