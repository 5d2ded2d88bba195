"""CPU stand-in for a DeviceNetwork shard, driven by the product's `parallel.ShardedStepper`: the oracle
computes the local postsynaptic range, the exchange buffer has the product's [shard][plane][stride]
layout.  Test infrastructure: lets the N > 1 protocol (what is exchanged, who updates what) run under
gloo on CPU and be compared bit for bit with the single-process oracle."""
import numpy as np
import torch

import oracle_binding as ob

NUM_PLANES = 5   # current_voltage, is_spiking, t[AMPA], t[NMDA], t[GABA]


class OracleShard:
    def __init__(self, net, rank, n_shards, stride):
        self.net, self.rank = net, rank
        nn = net.n_neurons
        self.stride, self.n_shards = stride, n_shards
        self.q0 = min(nn, rank * stride)
        self.q1 = min(nn, self.q0 + stride)
        self.buf = torch.zeros(n_shards * NUM_PLANES * stride, dtype=torch.float32)
        self._np = self.buf.numpy()

    def _plane(self, shard, plane):
        o = (shard * NUM_PLANES + plane) * self.stride
        return self._np[o:o + self.stride]

    def step_begin(self):
        n, q0, q1 = self.net, self.q0, self.q1
        if q1 > q0:
            n.inputs(q0, q1)
            n.update_neurons(q0, q1)
        m = q1 - q0
        self._plane(self.rank, 0)[:m] = n["current_voltage"][q0:q1]
        self._plane(self.rank, 1)[:m] = n["is_spiking"][q0:q1].view(np.float32)
        for k in range(3):
            self._plane(self.rank, 2 + k)[:m] = n["nt_t"][q0:q1, k]

    def apply_reward(self, reward):                               # modulators are replicated on every rank
        self.net.apply_reward(reward)

    def step_end(self):
        n = self.net
        nn = n.n_neurons
        for r in range(self.n_shards):
            if r == self.rank:
                continue
            b = min(nn, r * self.stride)
            e = min(nn, b + self.stride)
            m = e - b
            if m == 0:
                continue
            n["current_voltage"][b:e] = self._plane(r, 0)[:m]
            spk = self._plane(r, 1)[:m].view(np.uint32)
            n["is_spiking"][b:e] = spk
            n["last_firing_time"][b:e][spk != 0] = n.clock        # k_stamp_remote
            for k in range(3):
                n["nt_t"][b:e, k] = self._plane(r, 2 + k)[:m]
        n.plasticity(self.q0, self.q1)                            # owner of the column applies STDP
        n.reward_modulation(self.q0, self.q1)                     # ... and the reward-modulated update
        n.clock += 1
        if n.n_cells:
            n.spike_trains()                                      # replicated on every rank
