"""iqlpref_amd: MI355X-native IQL-with-preference-reward hot path.

``iqlpref_amd.iql`` mirrors the reference's ``algorithms/offline/iql.py`` module
surface on top of libiqlhip.so (hand-written HIP for gfx950)."""
from . import _lib  # noqa: F401
from .iql import (  # noqa: F401
    EXP_ADV_MAX, LOG_STD_MAX, LOG_STD_MIN, MLP, DeterministicPolicy, GaussianPolicy,
    EnsembleQ, ImplicitQLearning, ReplayBuffer, Squeeze, TrainConfig, TwinQ, ValueFunction,
    asymmetric_l2_loss, compute_mean_std, load_config, mlp_forward_f32, normalize_states,
    set_seed, soft_update)
from .relabel import (  # noqa: F401,E402
    RewardMLP, RewardPT, cvar_stability_check, empirical_cvar, keep_mask_and_steps,
    load_mlp_reward_model, load_pt_reward_model, modify_reward, qlearning_dataset_bnn,
    qlearning_dataset_mr, qlearning_dataset_mr_ensemble, qlearning_dataset_pt, return_reward_range)
from . import custom_offline, distributed, prep  # noqa: F401,E402
from .multi import SeedGroup  # noqa: F401,E402
from .train import EpisodeLedger, build_dataset, eval_actor, policy_actions, train, wrap_env  # noqa: F401,E402
