"""deep-q-learning_amd: an MI355X-native (gfx950) DDDQN inner training loop that keeps the call
surface of hal9000universe/deep-q-learning's hot path (Model, ReplayBuffer/sample_batch,
generate_q_target_comp, generate_train_step, action_computation, Agent._step) on top of
hand-written HIP kernels behind a C ABI (include/dqn_hip.h). Import as `deep_q_learning_amd`.
"""
from . import _lib
from .cnn import CnnEngine
from .engine import Engine, EngineConfig

__all__ = ["Engine", "EngineConfig", "CnnEngine", "_lib"]
