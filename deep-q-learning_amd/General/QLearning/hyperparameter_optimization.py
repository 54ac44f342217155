"""ParamAgent of the reference (General/QLearning/hyperparameter_optimization.py:14-91): an Agent whose seven searched
hyper-parameters can be replaced between training runs. The bayes_opt loop around it (:94-136) is orchestration outside
the hot path and is not mirrored.

`inject` in the reference only rebinds attributes (:84-90): the jitted `compute_q_targets` closure keeps the gamma it was
built with (q_agent.py:111 runs once, in the constructor). `rebuild_closures=True` (not in the reference) re-creates the
closure so that the injected gamma takes effect; the default reproduces the reference."""
from __future__ import annotations

from .q_agent import Agent
from .q_learning_functions import generate_q_target_comp


class ParamAgent(Agent):
    def inject(self, gamma, epsilon, epsilon_decay_rate, min_epsilon, replace_frequency, batch_size, train_frequency,
               rebuild_closures=False):
        self._gamma = gamma                                                     # :84
        self._epsilon = epsilon                                                 # :85
        self._epsilon_decay_rate = epsilon_decay_rate                           # :86
        self._min_epsilon = min_epsilon                                         # :87
        self._replace_frequency = replace_frequency                             # :88
        self._batch_size = batch_size                                           # :89
        self._train_frequency = train_frequency                                 # :90
        if rebuild_closures:
            self._compute_q_targets = generate_q_target_comp(self._network, gamma, self._env)

    @property
    def max_episodes(self):
        return self._max_episodes

    @max_episodes.setter
    def max_episodes(self, value):                                              # :72-74
        self._max_episodes = value
