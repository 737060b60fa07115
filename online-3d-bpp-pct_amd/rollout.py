"""Device-resident rollout edge.

The reference collects n-step rollouts through the host: `selected_leaf_node.cpu().numpy()`
(D2H), `envs.step`, `torch.tensor(1-done)` (H2D) and `PCTRolloutStorage.insert`
(train_tools.py:63-70, storage.py:4-50, tools.py:70-73).  Here the same tensors never leave
the GPU: the env takes the leaf INDEX the policy sampled (pct_step_index), reward / done stay
device tensors, and the storage below has the reference's shapes and update rules.
"""
import torch


def get_leaf_nodes(observation, internal_node_holder, leaf_node_holder):
    """tools.py:70-73: [N,(I+L+1)*9] -> ([N,I+L+1,9], leaf view [N,L,9]) (views, no copy)."""
    unify_obs = observation.reshape((observation.shape[0], -1, 9))
    leaf_nodes = unify_obs[:, internal_node_holder:internal_node_holder + leaf_node_holder, :]
    return unify_obs, leaf_nodes


class DeviceRollout(object):
    """storage.py:4-50 PCTRolloutStorage with every tensor on `device`."""

    def __init__(self, num_steps, num_processes, obs_shape, gamma, device):
        dev = torch.device(device)
        self.obs = torch.zeros(num_steps + 1, num_processes, *obs_shape, device=dev)
        self.rewards = torch.zeros(num_steps, num_processes, 1, device=dev)
        self.returns = torch.zeros(num_steps + 1, num_processes, 1, device=dev)
        self.action_log_probs = torch.zeros(num_steps, num_processes, 1, device=dev)
        self.actions = torch.zeros(num_steps, num_processes, 1, dtype=torch.long, device=dev)
        self.masks = torch.ones(num_steps + 1, num_processes, 1, device=dev)
        self.num_steps = num_steps
        self.gamma = gamma
        self.step = 0

    def insert(self, obs, actions, action_log_probs, rewards, masks):  # storage.py:33-39
        self.obs[self.step + 1].copy_(obs)
        self.actions[self.step].copy_(actions)
        self.action_log_probs[self.step].copy_(action_log_probs)
        self.rewards[self.step].copy_(rewards)
        self.masks[self.step + 1].copy_(masks)
        self.step = (self.step + 1) % self.num_steps

    def after_update(self):  # storage.py:41-43
        self.obs[0].copy_(self.obs[-1])
        self.masks[0].copy_(self.masks[-1])

    def compute_returns(self, next_value):  # storage.py:45-50
        self.returns[-1] = next_value
        for step in reversed(range(self.rewards.size(0))):
            self.returns[step] = self.returns[step + 1] * self.gamma * self.masks[step + 1] + self.rewards[step]

    def gather(self, group=None):
        """All ranks' shards as storage-shaped tensors (env axis = dim 1), over RCCL/xGMI."""
        from .sharding import gather_rollout
        out = {}
        for name in ("obs", "rewards", "returns", "action_log_probs", "actions", "masks"):
            t = getattr(self, name)
            out[name] = gather_rollout(t.transpose(0, 1).contiguous(), group).transpose(0, 1)
        return out


def collect(envs, policy, rollout, all_nodes=None):
    """train_tools.py:63-70 without host round trips.  `policy(all_nodes) -> (log_prob [N,1],
    leaf_index int64 [N,1])`; `envs` is a PctVecEnv.  Returns the last observation view."""
    I, L = envs.I, envs.Lh
    if all_nodes is None:
        all_nodes, _ = get_leaf_nodes(envs.current_obs(), I, L)
        rollout.obs[0].copy_(all_nodes)
    for _ in range(rollout.num_steps):
        with torch.no_grad():
            log_prob, idx = policy(all_nodes)
        obs, reward, mask = envs.step_device(idx)
        all_nodes, _ = get_leaf_nodes(obs, I, L)
        rollout.insert(all_nodes, idx, log_prob, reward, mask)
    return all_nodes
