"""CPU stand-in for a DeviceNetwork shard, driven by the product's `parallel.ShardedStepper`: the oracle
computes the local postsynaptic range, the segments travel in the product's wire format (include/snn_amd.h:
per segment `planes` x n 32-bit words + ceil(n / 32) words of spike bits; all-gather of whole slots or halo
segments per peer).  Test infrastructure: lets the N > 1 protocol (what is exchanged, who updates what) run under
gloo on CPU and be compared bit for bit with the single-process oracle."""
import numpy as np
import torch

PLANE_V, PLANE_T0 = 0, 2


def segment_words(planes, n):
    return planes * n + (n + 31) // 32


def pack_segment(net, idx, plane_ids):
    """wire words (int32) of the neurons `idx` (global indices; entries >= n_neurons are slot padding)"""
    n = len(idx)
    nn = net.n_neurons
    real = idx < nn
    safe = np.where(real, idx, 0)
    out = np.zeros(segment_words(len(plane_ids), n), np.uint32)
    for s, pid in enumerate(plane_ids):
        vals = net["current_voltage"][safe] if pid == PLANE_V else net["nt_t"][safe, pid - PLANE_T0]
        out[s * n:(s + 1) * n] = np.where(real, np.ascontiguousarray(vals, np.float32).view(np.uint32), 0)
    spk = (net["is_spiking"][safe] != 0) & real
    bits = np.zeros(((n + 31) // 32) * 32, np.uint8)
    bits[:n] = spk
    out[len(plane_ids) * n:] = np.packbits(bits.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).ravel()
    return out.view(np.int32)


def unpack_segment(net, idx, plane_ids, words):
    """k_exchange_unpack: incoming state of the neurons `idx` into the replica, last_firing_time stamped"""
    n = len(idx)
    w = np.ascontiguousarray(words).view(np.uint32)
    real = idx < net.n_neurons
    tgt = idx[real]
    for s, pid in enumerate(plane_ids):
        vals = w[s * n:(s + 1) * n].view(np.float32)[real]
        if pid == PLANE_V:
            net["current_voltage"][tgt] = vals
        else:
            net["nt_t"][tgt, pid - PLANE_T0] = vals
    bits = np.unpackbits(w[len(plane_ids) * n:].view(np.uint8), bitorder="little")[:n][real]
    net["is_spiking"][tgt] = bits
    net["last_firing_time"][tgt[bits != 0]] = net.clock


class OracleShard:
    """mode "allgather": whole slots; mode "halo": per peer the neurons whose edges reach the peer's columns."""

    def __init__(self, net, rank, n_shards, stride, mode="allgather"):
        self.net, self.rank, self.mode = net, rank, mode
        nn = net.n_neurons
        self.stride, self.n_shards = stride, n_shards
        self.q0 = min(nn, rank * stride)
        self.q1 = min(nn, self.q0 + stride)
        self.plane_ids = []
        if net.electrical:
            self.plane_ids.append(PLANE_V)
        if net.chemical:
            self.plane_ids += [PLANE_T0 + k for k in range(3) if net["nt_flags"][:, k].any()]
        P = len(self.plane_ids)
        if mode == "allgather":
            block = segment_words(P, stride)
            self.recv = torch.zeros(n_shards * block, dtype=torch.int32)
            self.send = self.recv[rank * block:(rank + 1) * block]
            self.send_idx = [np.arange(rank * stride, (rank + 1) * stride)] * n_shards
            self.recv_idx = [np.arange(p * stride, (p + 1) * stride) for p in range(n_shards)]
            self.send_off = np.zeros(n_shards, np.uint64)
            self.recv_off = np.arange(n_shards, dtype=np.uint64) * block
        else:
            conn = net["connections"][:nn] != 0                   # [pre, post]
            bounds = [(min(nn, p * stride), min(nn, (p + 1) * stride)) for p in range(n_shards)]
            reads = lambda b, e: np.flatnonzero(conn[:, b:e].any(axis=1))      # neurons with an edge into [b, e)
            mine = reads(self.q0, self.q1)
            self.recv_idx, self.send_idx = [], []
            for p, (b, e) in enumerate(bounds):
                self.recv_idx.append(mine[(mine >= b) & (mine < e)] if p != rank else np.zeros(0, np.int64))
                theirs = reads(b, e)
                self.send_idx.append(theirs[(theirs >= self.q0) & (theirs < self.q1)] if p != rank else np.zeros(0, np.int64))
            sw = [segment_words(P, len(i)) if len(i) else 0 for i in self.send_idx]
            rw = [segment_words(P, len(i)) if len(i) else 0 for i in self.recv_idx]
            self.send = torch.zeros(sum(sw), dtype=torch.int32)
            self.recv = torch.zeros(sum(rw), dtype=torch.int32)
            self.send_off = np.concatenate([[0], np.cumsum(sw)[:-1]]).astype(np.uint64)
            self.recv_off = np.concatenate([[0], np.cumsum(rw)[:-1]]).astype(np.uint64)
        self.send_cnt = np.array([segment_words(P, len(i)) if len(i) else 0 for i in self.send_idx], np.uint64)
        self.recv_cnt = np.array([segment_words(P, len(i)) if len(i) else 0 for i in self.recv_idx], np.uint64)

    def exchange_plan(self):
        return {"mode": self.mode, "n_shards": self.n_shards, "shard_index": self.rank, "shard_stride": self.stride,
                "planes": len(self.plane_ids), "plane_id": list(self.plane_ids),
                "send_tensor": self.send, "recv_tensor": self.recv,
                "send_words": self.send.numel(), "recv_words": self.recv.numel(),
                "send_offset": self.send_off, "send_count": self.send_cnt,
                "recv_offset": self.recv_off, "recv_count": self.recv_cnt}

    def step_begin(self):
        n, q0, q1 = self.net, self.q0, self.q1
        if q1 > q0:
            n.inputs(q0, q1)
            n.update_neurons(q0, q1)
        out = self.send.numpy()
        peers = [self.rank] if self.mode == "allgather" else range(self.n_shards)
        for p in peers:
            idx = self.send_idx[p]
            if len(idx):
                o = 0 if self.mode == "allgather" else int(self.send_off[p])
                seg = pack_segment(n, np.asarray(idx), self.plane_ids)
                out[o:o + seg.size] = seg

    def apply_reward(self, reward):                               # modulators are replicated on every rank
        self.net.apply_reward(reward)

    def step_end(self):
        n = self.net
        inc = self.recv.numpy()
        for p in range(self.n_shards):
            idx = self.recv_idx[p]
            if p == self.rank or len(idx) == 0:
                continue
            o = int(self.recv_off[p])
            unpack_segment(n, np.asarray(idx), self.plane_ids, inc[o:o + int(self.recv_cnt[p])])
        n.plasticity(self.q0, self.q1)                            # owner of the column applies STDP
        n.reward_modulation(self.q0, self.q1)                     # ... and the reward-modulated update
        n.clock += 1
        if n.n_cells:
            n.spike_trains()                                      # replicated on every rank
