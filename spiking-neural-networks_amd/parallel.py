"""Multi-GPU stepping: one process per GPU, the postsynaptic population sharded in equal slots.

The reference has no distributed path at all; the sharding follows from the step's data
flow (backend/src/neuron/mod.rs:1077-1085, 2640-2647): every neuron's input at step t depends only on
the state S(t) of its presynaptic cells, so a rank that owns the columns W[:, shard] of the synapse matrix
needs, per step, exactly the presynaptic state of the neurons owned elsewhere that its rows read -- ONE
exchange per step, sized by the handle's exchange plan (include/snn_amd.h): per neuron 4 B of voltage (gap
junctions on), 4 B per transmitter type in use (chemical synapses on) and one bit for the spike; dense
handles all-gather whole slots in place, sparse (CSR) handles trade exactly the neurons each peer's rows
reference (halo segments, all-to-all-v).  Spike-train cells are replicated (deterministic xorshift32),
plasticity is applied by the owner of the column.

Two drivers:
  * `DeviceNetwork.run_sharded(LibraryComm(...), n)` -- the step loop and RCCL inside libsnn_amd.so (one host call);
  * `ShardedStepper` -- the same protocol with torch.distributed moving the segments.  It is backend-agnostic on
    purpose: the product passes a `DeviceNetwork` shard, the CPU tests (gloo, world_size 2) pass an oracle-backed
    object with the same members (step_begin, step_end, exchange_plan).
"""
import ctypes as C


class _DeviceWords:
    """Expose a raw device allocation to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i4",
                                         "data": (int(ptr), False), "version": 2}


def device_words(ptr, n_words, device):
    """torch int32 view of `n_words` 32-bit words of device memory owned by the library"""
    import torch
    if not n_words:
        return torch.empty(0, dtype=torch.int32, device=device)
    return torch.as_tensor(_DeviceWords(ptr, n_words), device=device)


def exchange_tensors(plan, device):
    """(send, recv) torch views of a plan's buffers; backends without device memory put tensors in the plan"""
    if "send_tensor" in plan:
        return plan["send_tensor"], plan["recv_tensor"]
    return device_words(plan["send"], plan["send_words"], device), device_words(plan["recv"], plan["recv_words"], device)


class LibraryComm:
    """An RCCL communicator created by libsnn_amd.so itself (snn_comm_unique_id / snn_comm_init_rank); the 128-byte
    id travels from rank 0 through `torch.distributed` (any backend) or a caller-supplied `broadcast(bytes) -> bytes`.
    int(comm) is the ncclComm_t for DeviceNetwork.run_sharded / exchange."""

    def __init__(self, rank, world_size, device, broadcast=None, lib=None):
        from . import _lib
        self._L = lib or _lib.load()
        self._check = lambda code: _lib.check(code, self._L)
        ident = C.create_string_buffer(128)
        raw, failure = None, None
        if rank == 0:
            try:
                self._check(self._L.snn_comm_unique_id(ident))
                raw = bytes(ident.raw)
            except Exception as e:       # noqa: BLE001 -- told to the other ranks first: they wait in the broadcast below
                failure = e
        if world_size > 1:
            if broadcast is None:
                import torch.distributed as dist
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = broadcast(raw if raw is not None else b"")
        if failure is not None:
            raise failure
        if not raw or len(raw) != 128:
            raise RuntimeError("rank 0 could not create the RCCL unique id")
        h = C.c_void_p()
        self._check(self._L.snn_comm_init_rank(C.create_string_buffer(raw, 128), world_size, rank, device, C.byref(h)))
        self._h = h.value

    def __int__(self):
        return int(self._h)

    def count(self):
        """(ncclCommCount, ncclCommUserRank) of the communicator"""
        world, rank = C.c_int(0), C.c_int(-1)
        self._check(self._L.snn_comm_count(C.c_void_p(self._h), C.byref(world), C.byref(rank)))
        return world.value, rank.value

    def close(self):
        if self._h:
            self._check(self._L.snn_comm_destroy(C.c_void_p(self._h)))
            self._h = None


def wire_halo_lists(handles):
    """Halo plans of G CSR shard handles living in ONE process (tests, single-GPU emulation): every handle's need
    lists become its peers' send lists, then all commit."""
    g = len(handles)
    for r, h in enumerate(handles):
        for p in range(g):
            if p != r:
                handles[p].halo_set_sends(r, h.halo_needs(p))
    for h in handles:
        h.halo_commit()


def connect_peers(handles):
    """The peer form (include/snn_amd.h, snn_p2p_*) between G sparse shard handles of ONE process: every handle learns where its
    values go on every peer it exchanges with, then all commit.  Halo lists must be committed (wire_halo_lists or the
    library's own exchange of the lists).  Returns False -- and connects nobody -- when some handle that exchanges anything has
    no peer form (more planes than the voltage on the wire, a dense handle); handles that exchange nothing are left out."""
    g = len(handles)
    plans = [h.exchange_plan() for h in handles]
    busy = [int(p["send_words"]) + int(p["recv_words"]) > 0 for p in plans]
    local = [None] * g
    for r, h in enumerate(handles):
        if busy[r]:
            try:
                local[r] = h.p2p_local()
            except Exception:        # noqa: BLE001 -- SnnError 12: this plan has no peer form
                return False
    for r, h in enumerate(handles):
        for p in range(g):
            if p != r and busy[r] and busy[p] and (local[p]["counts"][r] or local[r]["counts"][p]):
                h.p2p_connect(p, local[p]["recv"][0], local[p]["recv"][1], local[p]["flags"], local[p]["offsets"][r])
    for r, h in enumerate(handles):
        if busy[r]:
            h.p2p_commit()
    return True


def copy_segments(plans, tensors):
    """The exchange between G shard handles of one process: device-to-device (or host) copies of every segment
    plans[r] sends to p into plans[p]'s receive segment for r.  tensors[r] = (send, recv)."""
    g = len(plans)
    for r in range(g):
        send = tensors[r][0]
        for p in range(g):
            if p == r:
                continue
            n = int(plans[r]["send_count"][p])
            if n == 0:
                continue
            so, ro = int(plans[r]["send_offset"][p]), int(plans[p]["recv_offset"][r])
            assert n == int(plans[p]["recv_count"][r]), "send / receive segment sizes disagree"
            tensors[p][1][ro:ro + n].copy_(send[so:so + n])


class LocalExchange:
    """G shard handles of ONE process (tests and the single-GPU emulation of the multi-GPU path): the exchange is a
    set of device-to-device copies between the handles' send and receive buffers, segment by segment."""

    def __init__(self, handles, device, halo=False):
        self.handles, self.device = list(handles), device
        if halo:
            wire_halo_lists(self.handles)
        self.refresh()

    def refresh(self):
        """(re)read the plans -- after set_synapses / transmitter flags / halo_commit on the handles"""
        self.plans = [h.exchange_plan() for h in self.handles]
        self.tensors = [exchange_tensors(p, self.device) for p in self.plans]

    def exchange(self, sync=True):
        copy_segments(self.plans, self.tensors)
        if sync:
            import torch
            torch.cuda.synchronize()

    def refresh_state(self):
        """after snn_set_synapses switched a synapse kind ON: the owners' current state of the plan's planes travels once"""
        self.refresh()
        needed = [h.refresh_begin() for h in self.handles]
        if any(needed):
            assert all(needed), "the ranks disagree on whether their mirrors are current"
            self.exchange()
            for h in self.handles:
                h.refresh_end()

    def step(self, sync=True):
        for h in self.handles:
            h.step_begin()
        self.exchange(sync)
        for h in self.handles:
            h.step_end()

    def bytes_per_step(self):
        """bytes each handle receives per step"""
        return [4 * sum(int(p["recv_count"][q]) for q in range(len(self.plans)) if q != r)
                for r, p in enumerate(self.plans)]


class ThreadCollectives:
    """The collectives of the library's own sharded loop (snn_run_sharded, snn_exchange, snn_comm_exchange_halo_lists) for G
    shard handles that live in ONE process, one host thread per rank -- the single-GPU emulation of a multi-GPU run, where RCCL
    refuses two ranks on one device.  Installed process-wide through snn_set_collectives; rank r passes `comm(r)` where a
    ncclComm_t goes.  Every operation is blocking: a rank waits for the device, meets the others at a barrier, copies what its
    peers send device to device and meets them again (nobody overwrites a buffer a peer still reads).
    "Waits for the device" = for the STREAM the library posts the operation on and for the stream the copies run on, never
    torch.cuda.synchronize(): that waits for every stream of the device, the other ranks' too -- and a rank that has left the
    operation may already have launched a peer-form step that polls for THIS rank's next launch (round 6: the give-ups of the
    emulated-rank tests after a spin limit of seconds, 3 of 16 runs)."""

    def __init__(self, world_size, device, lib=None):
        import threading
        import torch
        from . import _lib
        self._lib = _lib
        self._L = lib or _lib.load()
        self._torch, self._device = torch, device
        self.world = world_size
        self._barrier = threading.Barrier(world_size)
        self._ranks = (C.c_int * world_size)(*range(world_size))         # comm(r) = the address of entry r
        self._posted = [None] * world_size                               # what rank r offers in the operation in progress
        self._local = threading.local()
        import queue
        self._mail = {(a, b): queue.Queue() for a in range(world_size) for b in range(world_size)}   # sends of a to b, in order
        self.timeout = 120.0
        self.calls = {"all_gather": 0, "groups": 0}
        self._table = _lib.Collectives(_lib.COMM_QUERY_FN(self._count), _lib.COMM_QUERY_FN(self._user_rank),
                                       _lib.ALL_GATHER_FN(self._all_gather), _lib.SEND_RECV_FN(self._send),
                                       _lib.SEND_RECV_FN(self._recv), _lib.GROUP_FN(self._group_start), _lib.GROUP_FN(self._group_end))
        _lib.check(self._L.snn_set_collectives(C.byref(self._table)), self._L)

    def comm(self, rank):
        return C.addressof(self._ranks) + rank * C.sizeof(C.c_int)

    def close(self):
        """restores RCCL's table (the library holds raw pointers to this object's thunks until then)"""
        if self._table is not None:
            self._lib.check(self._L.snn_set_collectives(None), self._L)
            self._table = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    def abort(self):
        """a rank failed: release the others from the barrier (their calls then fail too)"""
        self._barrier.abort()

    def _rank_of(self, comm):
        return (int(comm) - C.addressof(self._ranks)) // C.sizeof(C.c_int)

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException:            # noqa: BLE001 -- a failure becomes the collective's error code (and breaks the barrier)
            self._barrier.abort()
            return 1

    def _count(self, comm, out):
        out[0] = self.world
        return 0

    def _user_rank(self, comm, out):
        out[0] = self._rank_of(comm)
        return 0

    def _copy(self, dst, src, words):
        d = device_words(dst, words, self._device)
        d.copy_(device_words(src, words, self._device))

    def _wait_stream(self, stream):
        """the work the library has enqueued on `stream` (a hipStream_t) in front of the operation"""
        if stream:
            self._torch.cuda.ExternalStream(int(stream), device=self._device).synchronize()
        else:
            self._torch.cuda.default_stream(self._device).synchronize()

    def _wait_copies(self):
        self._torch.cuda.current_stream(self._device).synchronize()

    def _all_gather(self, send, recv, count, _dtype, comm, stream):
        def run():
            r = self._rank_of(comm)
            self._wait_stream(stream)
            self._posted[r] = (int(send), int(count))
            self._barrier.wait()
            for p in range(self.world):
                src, n = self._posted[p]
                assert n == count
                if n and int(recv) + 4 * p * n != src:                      # (in place: the own slot already is where it goes)
                    self._copy(int(recv) + 4 * p * n, src, n)
            self._wait_copies()
            self._barrier.wait()
            if r == 0:
                self.calls["all_gather"] += 1
        return self._guard(run)

    def _group_start(self):
        self._local.ops = []
        return 0

    def _send(self, buf, count, _dtype, peer, comm, stream):
        self._local.ops.append(("send", int(buf), int(count), int(peer), self._rank_of(comm), int(stream or 0)))
        return 0

    def _recv(self, buf, count, _dtype, peer, comm, stream):
        self._local.ops.append(("recv", int(buf), int(count), int(peer), self._rank_of(comm), int(stream or 0)))
        return 0

    def _group_end(self):
        # point to point, as RCCL's: only the ranks of a pair meet.  Every rank posts all its sends first (buffer, size and an
        # event the receiver sets once it has copied), then serves its receives, then waits until its own sends were taken.
        def run():
            import threading
            ops = self._local.ops
            self._local.ops = []
            if not ops:
                return
            r = ops[0][4]
            for stream in sorted({op[5] for op in ops}):
                self._wait_stream(stream)
            taken = []
            for kind, buf, n, peer, _, _ in ops:
                if kind == "send":
                    done = threading.Event()
                    self._mail[(r, peer)].put((buf, n, done))
                    taken.append(done)
            for kind, buf, n, peer, _, _ in ops:
                if kind == "recv":
                    src, m, done = self._mail[(peer, r)].get(timeout=self.timeout)
                    assert m == n, "send and receive sizes disagree"
                    if n:
                        self._copy(buf, src, n)
                    self._wait_copies()
                    done.set()
            for done in taken:
                if not done.wait(self.timeout):
                    raise TimeoutError("a peer never received")
            if r == 0:
                self.calls["groups"] += 1
        return self._guard(run)


class ProcessCollectives:
    """The collectives of the library's own sharded loop for ranks that are PROCESSES without RCCL between them: the bytes go
    device -> host -> torch.distributed (gloo) -> host -> device.  What it is for: the whole multi-rank path of `bench.py --gpus N`
    and of the first-contact script -- one process per rank, the agreement before a run, snn_run_sharded, the peer-form trial
    and its fall-back -- on a box with ONE GPU, where RCCL refuses two ranks on one device.  Blocking, slow, and no evidence
    about xGMI; the arithmetic and the protocol are the real ones.  Installed process-wide (snn_set_collectives); `comm()`
    stands where a ncclComm_t goes."""

    def __init__(self, dist, rank, world_size, device, lib=None):
        import torch
        from . import _lib
        self._lib, self._L = _lib, lib or _lib.load()
        self._torch, self._dist, self._device = torch, dist, device
        self.rank, self.world = rank, world_size
        self._token = C.c_int(rank)
        self._ops = []
        self.calls = {"all_gather": 0, "groups": 0}
        self._table = _lib.Collectives(_lib.COMM_QUERY_FN(self._count), _lib.COMM_QUERY_FN(self._user_rank),
                                       _lib.ALL_GATHER_FN(self._all_gather), _lib.SEND_RECV_FN(self._send),
                                       _lib.SEND_RECV_FN(self._recv), _lib.GROUP_FN(self._group_start), _lib.GROUP_FN(self._group_end))
        _lib.check(self._L.snn_set_collectives(C.byref(self._table)), self._L)

    def comm(self):
        return C.addressof(self._token)

    def __int__(self):
        return self.comm()

    def count(self):
        return self.world, self.rank

    def close(self):
        if self._table is not None:
            self._lib.check(self._L.snn_set_collectives(None), self._L)
            self._table = None

    def _guard(self, fn):
        try:
            fn()
            return 0
        except BaseException as e:       # noqa: BLE001 -- a failure becomes the collective's error code
            import sys
            print(f"[ProcessCollectives] rank {self.rank}: {e!r}", file=sys.stderr)
            return 1

    def _count(self, comm, out):
        out[0] = self.world
        return 0

    def _user_rank(self, comm, out):
        out[0] = self.rank
        return 0

    def _all_gather(self, send, recv, count, _dtype, comm, _stream):
        def run():
            torch = self._torch
            torch.cuda.synchronize()
            n = int(count)
            mine = device_words(int(send), n, self._device).cpu()
            parts = [torch.empty(n, dtype=torch.int32) for _ in range(self.world)]
            self._dist.all_gather(parts, mine)
            for p in range(self.world):
                if n and int(recv) + 4 * p * n != int(send):
                    device_words(int(recv) + 4 * p * n, n, self._device).copy_(parts[p])
            torch.cuda.synchronize()
            self.calls["all_gather"] += 1
        return self._guard(run)

    def _group_start(self):
        self._ops = []
        return 0

    def _send(self, buf, count, _dtype, peer, comm, _stream):
        self._ops.append(("send", int(buf), int(count), int(peer)))
        return 0

    def _recv(self, buf, count, _dtype, peer, comm, _stream):
        self._ops.append(("recv", int(buf), int(count), int(peer)))
        return 0

    def _group_end(self):
        def run():
            torch, dist = self._torch, self._dist
            ops, self._ops = self._ops, []
            if not ops:
                return
            torch.cuda.synchronize()
            work, landing = [], []
            for kind, buf, n, peer in ops:
                if not n:
                    continue
                if kind == "send":
                    work.append(dist.isend(device_words(buf, n, self._device).cpu(), dst=peer))
                else:
                    host = torch.empty(n, dtype=torch.int32)
                    work.append(dist.irecv(host, src=peer))
                    landing.append((buf, n, host))
            for w in work:
                w.wait()
            for buf, n, host in landing:
                device_words(buf, n, self._device).copy_(host)
            torch.cuda.synchronize()
            self.calls["groups"] += 1
        return self._guard(run)


def try_peer_form(dn, dist, rank, world, device_index, run, rebuild, trial_steps=8, sabotage_rank=None):
    """The PEER form of the sparse shard step (one launch per step, no collective per step: include/snn_amd.h, snn_p2p_*), TRIED,
    with a fall-back every rank agrees on.  `dn` has run at least one step over the collective (its halo lists are committed).
      1. every rank that exchanges anything exports IPC handles of its receive sets, maps its neighbours' and commits; if any
         rank could not, every rank switches "halo_peer" off and keeps the collective;
      2. a trial of `trial_steps` steps through `run(k)`; a rank whose polls give up (SNN_ERR_WAIT) has left its handle mid-step
         and handles cannot be rolled back across ranks, so if ANY rank failed EVERY rank unmaps what it imported, closes its
         handle and gets a fresh one from `rebuild()` (which must re-apply everything the caller had set on the old one:
         synthetic drive, history options, warm-up), with "halo_peer" off;
      3. after a trial that every rank completed, the ranks compare their clocks.
    `sabotage_rank` (tests): that rank's polls give up at once while its neighbours store late.
    Returns (handle, note); note is "taken" or "fell back to the collective: <why>"."""
    imported = []

    def connect():
        plan = dn.exchange_plan()
        busy = int(plan["send_words"]) + int(plan["recv_words"]) > 0
        mine, err = None, None
        try:
            if busy:
                loc = dn.p2p_local()
                mine = (dn.p2p_ipc_export(), [int(x) for x in loc["offsets"]], [int(x) for x in loc["counts"]])
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        everyone = [None] * world
        dist.all_gather_object(everyone, (mine, err))
        if any(e for _, e in everyone):
            return next(e for _, e in everyone if e)
        try:
            for p in range(world):
                theirs = everyone[p][0]
                if busy and p != rank and theirs is not None and (theirs[2][rank] or mine[2][p]):
                    r0, r1, fl = dn.p2p_ipc_import(theirs[0], device=device_index)
                    imported.append((r0, r1, fl))
                    dn.p2p_connect(p, r0, r1, fl, theirs[1][rank])
            if busy:
                dn.p2p_commit()
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        errs = [None] * world
        dist.all_gather_object(errs, err)
        return next((e for e in errs if e), None)

    def unmap():
        for r0, r1, fl in imported:
            try:
                dn.p2p_ipc_close(r0, r1, fl, device=device_index)
            except Exception:       # noqa: BLE001 -- the mapping goes with the process at the latest
                pass
        imported.clear()

    failed = connect()
    if failed is not None:
        dn.set_option("halo_peer", 0)
        dist.barrier()
        unmap()
        return dn, f"fell back to the collective: {failed[:200]}"
    trial = None
    if sabotage_rank is not None:
        # (tests) every rank enters the trial -- the agreement before a run is a collective -- but the sabotaged rank gives up at
        # its first unanswered poll while the others store late: its run ends with SNN_ERR_WAIT, as a peer that is too slow would
        if rank == sabotage_rank:
            dn.set_option("halo_peer_spin_limit", 1)
        else:
            dn.set_option("halo_peer_delay", 30)
            dn.set_option("halo_peer_spin_limit", 2_000_000)
    try:
        run(trial_steps)
    except Exception as e:      # noqa: BLE001
        trial = repr(e)
    trials = [None] * world
    dist.all_gather_object(trials, (trial, None if trial else int(dn.clock)))
    failed = next((t for t, _ in trials if t), None)
    if failed is None and len({c for _, c in trials}) != 1:
        failed = f"the ranks' clocks differ after the trial: {[c for _, c in trials]}"
    if failed is None:
        return dn, "taken"
    dist.barrier()                  # nobody unmaps while a neighbour's launch may still store into the sets
    unmap()
    dn.close()
    dist.barrier()
    dn = rebuild()
    dn.set_option("halo_peer", 0)
    return dn, f"fell back to the collective: {failed[:200]}"


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
    """Drives one shard with torch.distributed: step_begin -> exchange of the packed segments -> step_end.

    backend members used: step_begin(), step_end(), exchange_plan() (and step_begin_local / apply_reward when present).
    """

    def __init__(self, backend, rank, world_size, group=None, sync=None, always_exchange=False, stream=None, device=None):
        import torch.distributed as dist
        self.backend, self.rank, self.world = backend, rank, world_size
        self.group = group
        self.dist = dist
        self.device = device
        self._sync = sync or (lambda: None)
        self._always = always_exchange      # run the collective even at world_size 1 (path check on one GPU)
        # GPU backends: a NON-default torch.cuda.Stream that the backend has adopted (backend.set_stream); all
        # kernels, the collective's stream dependencies and `work.wait()` are ordered on it
        self.stream = stream
        self.refresh_plan()

    def refresh_plan(self):
        """(re)read the backend's exchange plan -- after set_synapses / transmitter flags / halo_commit"""
        self.plan = self.backend.exchange_plan()
        self.send, self.recv = exchange_tensors(self.plan, self.device)
        assert self.plan["n_shards"] == self.world and self.plan["shard_index"] == self.rank
        self.is_cuda = self.recv.is_cuda or self.send.is_cuda

    def refresh_state(self):
        """after a synapse kind was switched ON (set_synapses / transmitter flags): re-read the plan and let the owners' current
        state of its planes travel once, so that the mirrors hold what the next step reads (every rank calls it)"""
        self.refresh_plan()
        if hasattr(self.backend, "refresh_begin") and self.backend.refresh_begin():
            self.exchange()
            self.backend.refresh_end()

    def exchange(self, async_op=False):
        """move the packed segments; async_op=True returns the pending work handle"""
        if self.world == 1 and not self._always:
            return None
        plan = self.plan
        if plan["mode"] == "allgather":
            if self.recv.numel() == 0:
                return None
            if self.is_cuda:
                work = self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group, async_op=async_op)
            else:   # gloo has no all_gather_into_tensor on every build: gather into per-rank views
                block = self.send.numel()
                views = [self.recv[r * block:(r + 1) * block] for r in range(self.world)]
                work = self.dist.all_gather(views, self.send.clone(), group=self.group, async_op=async_op)
        else:
            ins = [int(x) for x in plan["send_count"]]
            outs = [int(x) for x in plan["recv_count"]]
            work = self.dist.all_to_all_single(self.recv, self.send, output_split_sizes=outs, input_split_sizes=ins,
                                               group=self.group, async_op=async_op)
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
        overlap = overlap and self.is_cuda and hasattr(self.backend, "step_begin_local") and self.stream is not None
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
