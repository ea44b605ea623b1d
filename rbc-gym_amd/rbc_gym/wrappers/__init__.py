"""Wrappers of the reference (src/rbc_gym/wrappers/): same class names, arguments and formulas.

Single-env wrappers subclass gym.ObservationWrapper / RewardWrapper / Wrapper like the reference's.
`rbc_gym.wrappers.vector` holds the batched counterparts for the device-batched vector env; there the
observation normalisation is fused into the step kernel's observation write (rbc_set_obs_normalization)."""
from .normalize import RBCNormalizeObservation, RBCNormalizeReward, normalization_bounds, reward_scale
from .shaping import RBCRewardShaping, cell_distance, cell_distances, find_peaks_min_height
from .vector import VectorRBCNormalizeObservation, VectorRBCNormalizeReward, VectorRBCRewardShaping

__all__ = ["RBCNormalizeObservation", "RBCNormalizeReward", "RBCRewardShaping",
           "VectorRBCNormalizeObservation", "VectorRBCNormalizeReward", "VectorRBCRewardShaping",
           "normalization_bounds", "reward_scale", "cell_distance", "cell_distances", "find_peaks_min_height"]
