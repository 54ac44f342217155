"""Agent with the reference's constructor (the 23 keyword arguments of General/QLearning/q_agent.py:61-86)
and control flow (:137-222): epsilon-greedy policy, ring insert, one `_step` every `train_frequency` env
steps once `training_start` samples exist, target hard-copy every `replace_frequency` episodes, epsilon
decay per episode, stop when the 50-episode average reward exceeds `reward_to_reach`. Every numerical
operation goes through libdqn_hip.so (via the factories of q_learning_functions / ReplayBuffer).

This is the drop-in for a host-stepped (gym-style) environment; `Engine.train_iters` is the device-resident
loop for vectorised environments.
"""
from __future__ import annotations

from random import uniform
from statistics import mean

from numpy.random import randint

from ..Base.replay_buffer import ReplayBuffer, sample_batch
from ..Base.utils import generate_saving
from .q_learning_functions import action_computation, generate_q_target_comp, generate_train_step, preprocessing


class Agent:
    def __init__(self, network, params, optimizer, opt_state, env, buffer_size, obs_shape, ac_shape, gamma, epsilon,
                 epsilon_decay_rate, min_epsilon, max_episodes, max_steps, training_start, batch_size, train_frequency,
                 back_up_frequency, replace_frequency, reward_to_reach, num_actions, saving_directory,
                 monitoring=False, verbose=1):
        self._network, self._params, self._optimizer, self._opt_state = network, params, optimizer, opt_state
        self._target_params = params                                           # q_agent.py:91
        self._env = env
        self._replay_buffer = ReplayBuffer(buffer_size=buffer_size, obs_shape=obs_shape, ac_shape=ac_shape,
                                           max_batch=max(batch_size, 64))
        self._gamma, self._epsilon = gamma, epsilon
        self._epsilon_decay_rate, self._min_epsilon = epsilon_decay_rate, min_epsilon
        self._max_episodes, self._max_steps = max_episodes, max_steps
        self._training_start, self._batch_size = training_start, batch_size
        self._train_frequency, self._back_up_frequency = train_frequency, back_up_frequency
        self._replace_frequency, self._reward_to_reach = replace_frequency, reward_to_reach
        self._num_actions = num_actions
        self._reward_history = []
        self._compute_action = action_computation(network)                     # :110
        self._compute_q_targets = generate_q_target_comp(network, gamma, env)  # :111
        self._train_step = generate_train_step(optimizer, network)             # :112
        self._save_state = generate_saving(saving_directory)                   # :113
        self._monitoring, self._verbose = monitoring, verbose
        self.updates = 0

    def _average_reward(self):
        return mean(self._reward_history)                                      # :134-135

    def _policy(self, state):
        if self._epsilon < uniform(0, 1):                                      # :138
            return int(self._compute_action(self._params, state))              # :139
        return int(randint(0, self._num_actions))                              # :141

    def _step(self):
        rb = self._replay_buffer
        batch = sample_batch(rb.size, rb.states, rb.actions, rb.rewards, rb.observations, rb.dones, self._batch_size)  # :147
        states, actions, rewards, observations, dones = preprocessing(*batch)  # :154
        q_targets = self._compute_q_targets(self._params, self._target_params, states, actions, rewards,
                                            observations, dones)               # :159
        self._params, self._opt_state = self._train_step(self._params, self._opt_state, states, q_targets)   # :166
        self.updates += 1

    def _run_episode(self, step_count, episode):
        epi_reward = 0.0
        state = self._env.reset()                                              # :173
        for step in range(1, self._max_episodes + 1):                          # :174 (sic: bounded by max_episodes)
            step_count += 1
            action = self._policy(state)
            observation, reward, done, info = self._env.step(action)
            if step == self._max_steps:                                        # :179-180
                done = True
            self._replay_buffer.add(state[0], action, reward, observation[0], done)   # :182
            state = observation
            epi_reward += reward
            if self._replay_buffer.size >= self._training_start and step_count % self._train_frequency == 0:   # :186
                self._step()
            if done:
                break
        if episode % self._replace_frequency == 0:                             # :192-193
            self._target_params = self._params
        if episode % self._back_up_frequency == 0:                             # :195-196
            self._save_state(self._params, self._opt_state)
        self._epsilon = max(self._epsilon * self._epsilon_decay_rate, self._min_epsilon)    # :121
        self._reward_history.append(epi_reward)                                # :124-126
        while len(self._reward_history) > 50:
            self._reward_history.pop(0)
        return step_count

    def training(self):
        step_count = 0
        for episode in range(self._max_episodes):                              # :211
            step_count = self._run_episode(step_count, episode)
            if episode % 50 == 0 and self._verbose:
                print("Episode: {} -- Reward: {} -- Average: {}".format(episode, self._reward_history[-1],
                                                                        self._average_reward()))
            if self._average_reward() > self._reward_to_reach:                 # :219-222
                self._save_state(self._params, self._opt_state)
                return

    def evaluate(self):
        for _ in range(10):                                                    # :225-230
            state = self._env.reset()
            for _ in range(self._max_steps):
                action = int(self._compute_action(self._params, state))
                state, reward, done, info = self._env.step(action)
        return self._average_reward()
