"""Multi-GPU stepping: one process per GPU, the postsynaptic population sharded in equal slots.

The reference has no distributed path at all; the sharding follows from the step's data
flow (backend/src/neuron/mod.rs:1077-1085, 2640-2647): every neuron's input at step t depends only on
the state S(t) of its presynaptic cells, so a rank that owns the columns W[:, shard] of the dense
synapse matrix needs, per step, exactly the other shards' exchanged planes (voltage, spike flag,
neurotransmitter concentrations) -- ONE in-place all-gather of contiguous per-rank blocks over
RCCL/xGMI (`torch.distributed`, backend "nccl"), no other collective.  Spike-train cells are
replicated (deterministic xorshift32), plasticity is applied by the owner of the column.

`ShardedStepper` is backend-agnostic on purpose: the product passes a `DeviceNetwork` shard, the CPU
tests (gloo, world_size 2) pass an oracle-backed object with the same four members.
"""


class _DeviceWords:
    """Expose a raw device allocation to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<f4",
                                         "data": (int(ptr), False), "version": 2}


def exchange_tensor(dn, device):
    """torch view of a DeviceNetwork's exchange buffer: [n_shards * words_per_neuron * stride] f32."""
    import torch
    ptr, words, n_padded = dn.exchange_buffer()
    return torch.as_tensor(_DeviceWords(ptr, words * n_padded), device=device)


def shard_geometry(n_neurons, n_shards):
    """Slot size and [begin, end) of every shard -- same rule as snn_network_finalize_shard."""
    per = -(-n_neurons // n_shards)            # ceil
    stride = max(64, -(-per // 64) * 64)       # rounded up to a wavefront
    out = []
    for r in range(n_shards):
        b = min(n_neurons, r * stride)
        out.append((b, min(n_neurons, b + stride)))
    return stride, out


class ShardedStepper:
    """Drives one shard: step_begin -> all-gather of the exchanged planes -> step_end.

    backend members used: step_begin(), step_end(); `buf` is a 1-D torch tensor over the backend's
    exchange buffer laid out [shard][plane][stride] so that shard r's block is contiguous.
    """

    def __init__(self, backend, buf, rank, world_size, group=None, sync=None, always_gather=False, stream=None):
        import torch.distributed as dist
        self.backend, self.buf, self.rank, self.world = backend, buf, rank, world_size
        self.group = group
        self.dist = dist
        assert buf.numel() % world_size == 0, "exchange buffer must split evenly over the ranks"
        self.block = buf.numel() // world_size
        self.local = buf[rank * self.block:(rank + 1) * self.block]
        self._sync = sync or (lambda: None)
        self._always = always_gather        # run the collective even at world_size 1 (path check on one GPU)
        # GPU backends: a NON-default torch.cuda.Stream that the backend has adopted (backend.set_stream); all
        # kernels, the collective's stream dependencies and `work.wait()` are ordered on it
        self.stream = stream

    def exchange(self, async_op=False):
        """all-gather the exchanged planes in place; async_op=True returns the pending work handle"""
        if self.world == 1 and not self._always:
            return None
        if self.buf.is_cuda:
            work = self.dist.all_gather_into_tensor(self.buf, self.local, group=self.group, async_op=async_op)
        else:   # gloo has no all_gather_into_tensor on every build: gather into per-rank views
            views = [self.buf[r * self.block:(r + 1) * self.block] for r in range(self.world)]
            work = self.dist.all_gather(views, self.local.clone(), group=self.group, async_op=async_op)
        if not async_op:
            self._sync()
        return work if async_op else None

    def run(self, iterations, overlap=True, rewards=None):
        """`iterations` steps.  overlap (GPU backends): the collective of step t runs on RCCL's stream while
        step t+1's synaptic-input pass over the shard's OWN presynaptic rows is already executing; only the
        rows fed by other shards wait for it (backend.step_begin_local, a no-op when it would not be valid).
        rewards: one value per step, applied to the reward-modulated lattices before that step (every rank passes
        the same sequence: the modulators are replicated)."""
        if rewards is not None and len(rewards) != int(iterations):
            raise ValueError("rewards must hold one value per step")
        self._rewards = rewards
        overlap = overlap and self.buf.is_cuda and hasattr(self.backend, "step_begin_local") and self.stream is not None
        if self.stream is not None:
            import torch
            with torch.cuda.stream(self.stream):     # collectives synchronise with the CURRENT stream: make it ours
                self._run(iterations, overlap)
        else:
            self._run(iterations, overlap)

    def _run(self, iterations, overlap):
        rewards = self._rewards
        if not overlap:
            for i in range(int(iterations)):
                if rewards is not None:
                    self.backend.apply_reward(float(rewards[i]))
                self.backend.step_begin()
                self.exchange()
                self.backend.step_end()
            return
        pending = None
        started = False
        for i in range(int(iterations)):
            self.backend.step_begin_local()
            if started:
                if pending is not None:
                    pending.wait()                 # the compute stream waits for the gather, the host does not
                self.backend.step_end()
            if rewards is not None:                # after the previous step's weight update, before this step
                self.backend.apply_reward(float(rewards[i]))
            self.backend.step_begin()
            pending = self.exchange(async_op=True)
            started = True
        if started:
            if pending is not None:
                pending.wait()
            self.backend.step_end()
