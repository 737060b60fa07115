"""Device-resident rollout edge.

The reference collects n-step rollouts through the host: `selected_leaf_node.cpu().numpy()` (D2H),
`envs.step`, `torch.tensor(1-done)` (H2D) and `PCTRolloutStorage.insert`, five `copy_` calls per step
(train_tools.py:63-70, storage.py:33-39, tools.py:70-73).  Here the insert is FUSED into the transition: the
env takes the leaf INDEX the policy sampled and the transition kernel itself writes step t's new observation
into `obs[t + 1]`, its reward into `rewards[t]` and 1 - done into `masks[t + 1]` (PctVecEnv.step_into ->
pct_bind_rollout_slot + pct_step_index): one launch per step, no copy kernels, nothing leaves the GPU.  What
the policy produces (action index, log-probability) is the caller's to keep; `RolloutSlots.step_env` records
them beside the slot.  The tensor names and shapes are the trainer's contract (storage.py:5-11); the update rules
`after_update` / `compute_returns` are storage.py:41-50, restated.
"""
import torch


def get_leaf_nodes(observation, internal_node_holder, leaf_node_holder):
    """tools.py:70-73: [N,(I+L+1)*9] -> ([N,I+L+1,9], leaf view [N,L,9]) (views, no copy)."""
    unify_obs = observation.reshape((observation.shape[0], -1, 9))
    leaf_nodes = unify_obs[:, internal_node_holder:internal_node_holder + leaf_node_holder, :]
    return unify_obs, leaf_nodes


class RolloutSlots(object):
    """The trainer's rollout tensors (storage.py:5-11), every one on `device`, laid out so that slot t of each
    is a contiguous block the transition kernel can write directly."""

    def __init__(self, num_steps, num_processes, obs_shape, gamma, device):
        dev = torch.device(device)
        T, N = num_steps, num_processes
        self.obs = torch.zeros(T + 1, N, *obs_shape, device=dev)
        self.rewards = torch.zeros(T, N, 1, device=dev)
        self.returns = torch.zeros(T + 1, N, 1, device=dev)
        self.action_log_probs = torch.zeros(T, N, 1, device=dev)
        self.actions = torch.zeros(T, N, 1, dtype=torch.long, device=dev)
        self.masks = torch.ones(T + 1, N, 1, device=dev)
        self.num_steps, self.gamma, self.step = T, gamma, 0

    def begin(self, envs):
        """Slot 0 <- the env's current observation (storage.py:41-43 does this with obs[-1] after an update)."""
        self.obs[0].copy_(envs.current_obs().view_as(self.obs[0]))
        self.step = 0

    def step_env(self, envs, leaf_index, action_log_probs=None):
        """One env step written in place: obs[t+1], rewards[t], masks[t+1] by the transition kernel; the policy's
        own outputs are recorded beside them.  Returns the new observation view obs[t+1]."""
        t = self.step
        envs.step_into(leaf_index, self.obs[t + 1], self.rewards[t], self.masks[t + 1])
        if leaf_index.data_ptr() != self.actions[t].data_ptr():  # (a policy may write its choice straight into actions[t])
            self.actions[t].copy_(leaf_index.view_as(self.actions[t]))
        if action_log_probs is not None:
            self.action_log_probs[t].copy_(action_log_probs)
        self.step = (t + 1) % self.num_steps
        return self.obs[t + 1]

    def insert(self, obs, actions, action_log_probs, rewards, masks):
        """storage.py:33-39, for callers that step the env themselves (round-1 DeviceRollout pattern): five device
        copies instead of the fused write of step_env."""
        t = self.step
        self.obs[t + 1].copy_(obs.view_as(self.obs[t + 1]))
        self.actions[t].copy_(actions.view_as(self.actions[t]))
        self.action_log_probs[t].copy_(action_log_probs.view_as(self.action_log_probs[t]))
        self.rewards[t].copy_(rewards.view_as(self.rewards[t]))
        self.masks[t + 1].copy_(masks.view_as(self.masks[t + 1]))
        self.step = (t + 1) % self.num_steps

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


DeviceRollout = RolloutSlots  # round-1 name


def collect(envs, policy, rollout):
    """train_tools.py:63-70 without host round trips and without copy kernels: per step the policy and ONE
    transition launch.  `policy(all_nodes [N,I+L+1,9]) -> (log_prob [N,1], leaf_index int64 [N,1])`; `envs` is a
    PctVecEnv; if its current observation is not the rollout's current slot yet, the slot is seeded from it.
    Returns the last observation view."""
    cur = envs.current_obs()
    slot = rollout.obs[rollout.step]
    if cur.data_ptr() != slot.data_ptr():
        # the env's current observation is not this slot yet (first collection, or the env was stepped / reset outside
        # the rollout): seed the slot from it, as storage.py:41-43 / main.py's obs[0].copy_ do
        slot.copy_(cur.view_as(slot))
    all_nodes = slot
    for _ in range(rollout.num_steps):
        with torch.no_grad():
            log_prob, idx = policy(all_nodes.view(all_nodes.shape[0], -1, 9))  # tools.py:70-73 unify_obs
        all_nodes = rollout.step_env(envs, idx, log_prob)
    return all_nodes
