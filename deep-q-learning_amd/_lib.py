"""ctypes binding of libdqn_hip.so (include/dqn_hip.h). There is NO fallback: if the HIP
library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DQN_HIP_LIB", os.path.join(_HERE, "libdqn_hip.so"))   # override: diagnostic builds only

# dqn_status / enums (include/dqn_hip.h)
OPT_ADAM, OPT_ADAMW = 0, 1
PREC_F32, PREC_BF16 = 0, 1
NET_ONLINE, NET_TARGET = 0, 1
ENV_SYNTHETIC, ENV_CARTPOLE = 0, 1
FLAG_NO_HANDOVER, FLAG_NO_ACTOR16, FLAG_BF16_F32_ACTOR, FLAG_BIG_ROWS, FLAG_PW_SEGMENTS, FLAG_PW_CHUNKS = 1, 2, 4, 8, 16, 32
CNN_FLAG_FC_WIDE_TILE, CNN_FLAG_NO_SIDE_STREAM, CNN_FLAG_LAYERWISE_CONV = 1, 2, 4
ABI_VERSION = 4
(BUF_PARAMS, BUF_TARGET, BUF_MU, BUF_NU, BUF_GRAD, BUF_TREE, BUF_STATES, BUF_ACTIONS, BUF_REWARDS,
 BUF_OBSERVATIONS, BUF_DONES, BUF_BATCH_IDX, BUF_BATCH_ISW, BUF_BATCH_TD, BUF_LOSS, BUF_ENV_OBS,
 BUF_ENV_ACTIONS) = range(17)


class DqnConfig(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("hidden1", C.c_int32), ("hidden2", C.c_int32),
                ("num_actions", C.c_int32), ("capacity", C.c_int64), ("use_per", C.c_int32),
                ("max_batch", C.c_int32), ("optimizer", C.c_int32), ("lr", C.c_float), ("b1", C.c_float),
                ("b2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("gamma", C.c_float),
                ("per_alpha", C.c_float), ("per_eps", C.c_float), ("per_beta", C.c_float),
                ("precision", C.c_int32), ("seed", C.c_uint64), ("world_size", C.c_int32),
                ("n_step", C.c_int32), ("flags", C.c_int32)]


_P, _I32, _I64, _U64, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float

# name -> argtypes; every function returns int except the three noted below
SIGNATURES = {
    "dqn_create": [C.POINTER(DqnConfig), C.POINTER(_P)],
    "dqn_destroy": [_P],
    "dqn_param_count": [_P, C.POINTER(_I64)],
    "dqn_set_params": [_P, C.c_int, _P, C.c_int, _P],
    "dqn_get_params": [_P, C.c_int, _P, C.c_int, _P],
    "dqn_set_opt_count": [_P, _I32, _P],
    "dqn_get_opt_count_host": [_P, C.POINTER(_I32)],
    "dqn_buffer": [_P, C.c_int, C.POINTER(_P), C.POINTER(_I64)],
    "dqn_set_schedule": [_P, _F, _F, _P],
    "dqn_set_gamma": [_P, _F],
    "dqn_replay_add": [_P, _P, _P, _P, _P, _P, _I32, _P],
    "dqn_replay_size_host": [_P, C.POINTER(_I64), C.POINTER(_I64)],
    "dqn_replay_sample_uniform": [_P, _I32, _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _P],
    "dqn_per_sample": [_P, _I32, _F, _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _P],
    "dqn_per_update": [_P, _P, _P, _I32, _P],
    "dqn_per_set": [_P, _P, _P, _I32, _P],
    "dqn_per_update_sorted": [_P, _P, _P, _I32, _P],
    "dqn_per_set_sorted": [_P, _P, _P, _I32, _P],
    "dqn_qnet_forward": [_P, C.c_int, _P, _I32, _P, _P, _P],
    "dqn_td_targets": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I32, _P, _P, _P, _P, _P],
    "dqn_q_targets": [_P, _P, _P, _P, _P, _P, _I32, _P, _P],
    "dqn_loss": [_P, _P, _P, _P, _I32, _P, _P],
    "dqn_grads": [_P, _P, _P, _P, _I32, _P, _P],
    "dqn_optimizer_step": [_P, _P],
    "dqn_train_step": [_P, _P, _P, _I32, _P],
    "dqn_update_fused": [_P, _I32, _P],
    "dqn_update_backward": [_P, _I32, _P],
    "dqn_update_apply": [_P, _I32, _P],
    "dqn_act": [_P, _P, _I32, _F, _U64, _U64, _P, _P],
    "dqn_sync_target": [_P, _P],
    "dqn_set_epsilon": [_P, _F, _P],
    "dqn_env_config": [_P, _I32, _I32, _F],
    "dqn_env_stats_host": [_P, C.POINTER(_I64), C.POINTER(_I64)],
    "dqn_env_time_feature": [_P, _I32],
    "dqn_env_reset": [_P, _P, _I32, _F, _P],
    "dqn_actor_step": [_P, _I32, _P],
    "dqn_actor_steps": [_P, _I32, _I32, _P],
    "dqn_train_iters": [_P, _I32, _I32, _I32, _I32, _P],
    "dqn_actor_backward": [_P, _I32, _I32, _I32, _P],
    "dqn_profile_begin": [_P, _P],
    "dqn_profile_end": [_P, _P, _P, _I32, _P, _I32, C.POINTER(_I32)],
    "dqn_comm_unique_id": [_P],
    "dqn_comm_init": [_P, _P, _I32, _I32],
    "dqn_allreduce_grads": [_P, _P],
    "dqn_comm_count_host": [_P, C.POINTER(_I32)],
    "dqn_device_errors_host": [_P, C.POINTER(_I64)],
    "dqn_clear_device_errors": [_P],
    "dqn_debug_withhold_handover": [_P, _I32],
    "dqn_cnn_create": [_I32, _I32, _I32, C.POINTER(_P)],
    "dqn_cnn_destroy": [_P],
    "dqn_cnn_set_flags": [_P, _I32],
    "dqn_cnn_env_reset_synth": [_P, _I32, C.c_uint64, _P],
    "dqn_cnn_env_step_synth": [_P, C.c_float, C.c_float, C.POINTER(_I64), _P],
    "dqn_per_index_advance": [_P, _I32, _P],
    "dqn_per_index_step": [_P, _I32, _I64, _I32, _P],
    "dqn_cnn_comm_init": [_P, _P, _I32, _I32],
    "dqn_cnn_comm_count_host": [_P, C.POINTER(_I32)],
    "dqn_cnn_allreduce_grads": [_P, _I32, _P],
    "dqn_cnn_param_count": [_P, C.POINTER(_I64)],
    "dqn_cnn_set_params": [_P, C.c_int, _P, C.c_int, _P],
    "dqn_cnn_forward": [_P, C.c_int, _P, _I32, _P, _P],
    "dqn_cnn_q_targets": [_P, _P, _P, _P, _P, _P, _F, _I32, _P, _P],
    "dqn_cnn_grads": [_P, _P, _P, _P, _I32, _P, _P],
    "dqn_cnn_get_buffer": [_P, C.c_int, _P, C.c_int, _P],
    "dqn_cnn_buffer": [_P, C.c_int, C.POINTER(_P), C.POINTER(_I64)],
    "dqn_cnn_set_optimizer": [_P, _I32, _F, _F, _F, _F, _F, _P],
    "dqn_cnn_optimizer_step": [_P, _F, _P],
    "dqn_cnn_train_step": [_P, _P, _P, _P, _I32, _P],
    "dqn_cnn_update": [_P, _P, _P, _P, _P, _P, _P, _F, _I32, _P, _P],
    "dqn_cnn_sync_target": [_P, _P],
    "dqn_cnn_act": [_P, _P, _I32, _F, C.c_uint64, C.c_uint64, _P, _P],
    "dqn_cnn_replay_init": [_P, _I64],
    "dqn_cnn_replay_add": [_P, _P, _P, _P, _P, _P, _I32, C.POINTER(_I64), _P],
    "dqn_cnn_replay_size_host": [_P, C.POINTER(_I64), C.POINTER(_I64)],
    "dqn_cnn_replay_gather": [_P, _P, _I32, _I32, _I32, _F, _P, _P, _P, _P, _P, _P],
    "dqn_cnn_update_replay": [_P, _P, _P, _F, _I32, _I32, _I32, _P, _P, _P],
}
OTHER = {"dqn_last_error": ([], C.c_char_p), "dqn_abi_version": ([], C.c_int),
         "dqn_default_config": ([C.POINTER(DqnConfig)], None)}

_lib = None


def load():
    """Load libdqn_hip.so and bind every symbol include/dqn_hip.h declares. Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or make -C deep-q-learning_amd/csrc). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes, fn.restype = args, C.c_int
    for name, (args, res) in OTHER.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, res
    if lib.dqn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.dqn_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


class DqnError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        raise DqnError(f"libdqn_hip error {rc}: {load().dqn_last_error().decode()}")
