/*
 * snn_amd.h -- C ABI of the MI355X-native spiking-lattice time-stepper.
 *
 * Drop-in boundary for ONE path of NikhilMukraj/spiking-neural-networks: the
 * GPU lattice seam of the `backend` crate (all paths relative to
 * /root/reference/backend/src/):
 *
 *   LatticeGPU::from_lattice            neuron/gpu_lattices/mod.rs:496-511
 *   RunLattice for LatticeGPU           neuron/gpu_lattices/mod.rs:1081-1100
 *   LatticeNetworkGPU::from_network     neuron/gpu_lattices/mod.rs:1636-1651
 *   RunNetwork for LatticeNetworkGPU    neuron/gpu_lattices/mod.rs:3183-3212
 *   IterateAndSpikeGPU::convert_to_gpu / convert_to_cpu
 *                                       neuron/iterate_and_spike/mod.rs:3156-3189
 *   GraphToGPU / InterleavingGraphGPU   graph/mod.rs:97-107, 300-404, 579-943
 *   SpikeTrainGPU                       neuron/spike_train/mod.rs:216-255
 *   LatticeHistoryGPU                   neuron/gpu_lattices/mod.rs:178-280
 *   GPUError                            error/mod.rs:221-238
 *
 * Plain pointers and sizes only.  A handle owns device memory on ONE GPU and,
 * optionally, one contiguous shard [post_begin, post_end) of the postsynaptic
 * population (multi-GPU: one process and one handle per GPU).  Calls on one
 * handle are blocking and not re-entrant; distinct handles may be used from
 * distinct threads.  The library never frees or retains caller memory.
 *
 * Index space ("interleaved", graph/mod.rs:668-727): neurons of all neuron
 * lattices in ascending lattice id, row-major inside a lattice; spike-train
 * cells of all spike-train lattices after them, same ordering.  n_tot =
 * n_neurons + n_cells.  Spike-train cells are never postsynaptic.
 *
 * Per-cell attributes keep the reference's buffer names (struct field names,
 * `$` for nesting -- neuron/integrate_and_fire/mod.rs:729-773,
 * neuron/iterate_and_spike/mod.rs:209-243, 1369-1422, 2653) and its three
 * scalar types Float / UInt / OptionalUInt (iterate_and_spike/mod.rs:3112-3134):
 * f32, u32, and i32 with -1 == None.  Attributes with a per-type dimension
 * (neurotransmitters$*, receptors$flags) are laid out [cell * 3 + type] with
 * type AMPA = 0, NMDA = 1, GABA = 2 (iterate_and_spike/mod.rs:1323-1333), as the
 * reference's kernels index them (gpu_lattices/mod.rs:117-127).
 */
#ifndef SNN_AMD_H
#define SNN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNN_ABI_VERSION 2
#define SNN_NUM_NT_TYPES 3
/* canonical reduction chunk: presynaptic indices are summed in runs of 256 */
#define SNN_REDUCTION_CHUNK 256

typedef struct snn_network snn_network_t;

/* 1..8 mirror GPUError's variants in declaration order (error/mod.rs:221-238) */
typedef enum snn_status {
    SNN_OK = 0,
    SNN_ERR_PROGRAM_COMPILE = 1,
    SNN_ERR_KERNEL_COMPILE = 2,
    SNN_ERR_BUFFER_CREATE = 3,
    SNN_ERR_BUFFER_WRITE = 4,
    SNN_ERR_BUFFER_READ = 5,
    SNN_ERR_WAIT = 6,
    SNN_ERR_GET_DEVICE = 7,
    SNN_ERR_QUEUE = 8,
    SNN_ERR_BAD_ATTR = 9,        /* unknown attribute name / wrong scalar type */
    SNN_ERR_DIM_MISMATCH = 10,   /* GraphError::DimensionsDoNotMatch, error/mod.rs:24 */
    SNN_ERR_BAD_ARG = 11,
    SNN_ERR_BAD_STATE = 12       /* call order (e.g. add_lattice after finalize) */
} snn_status;

/* neuron models: IzhikevichNeuron integrate_and_fire/mod.rs:1159-1268,
 * LeakyIntegrateAndFireNeuron :108-215, HodgkinHuxleyNeuron hodgkin_huxley/mod.rs:49-241 */
typedef enum {
    SNN_MODEL_IZHIKEVICH = 0, SNN_MODEL_LIF = 1, SNN_MODEL_HODGKIN_HUXLEY = 2,
    /* the two models the reference's own GPU path implements (IterateAndSpikeGPU):
     * QuadraticIntegrateAndFireNeuron integrate_and_fire/mod.rs:259-917, buffers :729-773;
     * SimpleLeakyIntegrateAndFire :1523-1801 */
    SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE = 3, SNN_MODEL_SIMPLE_LIF = 4,
    /* further integrate-and-fire models of the CPU path (no reference GPU form; attribute names = field names):
     * AdaptiveLeakyIntegrateAndFireNeuron integrate_and_fire/mod.rs:918-1049 (alpha, beta, w_value + the LIF set),
     * AdaptiveExpLeakyIntegrateAndFireNeuron :1051-1155 (+ slope_factor),
     * LeakyIzhikevichNeuron :1270-1356 (the Izhikevich set + e_l) */
    SNN_MODEL_ADAPTIVE_LIF = 5, SNN_MODEL_ADAPTIVE_EXP_LIF = 6, SNN_MODEL_LEAKY_IZHIKEVICH = 7,
    /* BCMIzhikevichNeuron :1358-1518: Izhikevich + BCMActivity bookkeeping (attributes average_activity,
     * current_activity, firing_rate_clock, firing_rate_window f32; period, num_spikes u32) */
    SNN_MODEL_BCM_IZHIKEVICH = 8,
    /* The ONE neuron model generated from a neuron_builder!-style description (build_test/nb_macro) that a library
     * built with -DSNN_CUSTOM_MODEL_HEADER carries (spiking-neural-networks_amd/modelgen.py); its variables are
     * attributes under their DSL names.  snn_custom_model() names it ("" when the library has none). */
    SNN_MODEL_CUSTOM = 100
} snn_model;
/* NeurotransmitterKinetics: Approximate iterate_and_spike/mod.rs:161-205, Destexhe :122-159 */
/* DiscreteSpikeNeurotransmitter :287-317; ExponentialDecayNeurotransmitter :323-366 (attribute
 * neurotransmitters$decay_constant) */
/* SNN_NT_CUSTOM / SNN_RC_CUSTOM: the kinetics generated from a `[neurotransmitter_kinetics]` / `[receptor_kinetics]`
 * block (build_test/nb_macro/src/lib.rs:6468-6540, 6757-6826) that a library built with -DSNN_CUSTOM_MODEL_HEADER
 * carries; variables are the attributes neurotransmitters$<name> / receptors$<TYPE>$r$kinetics$<name>. */
typedef enum { SNN_NT_APPROXIMATE = 0, SNN_NT_DESTEXHE = 1, SNN_NT_DISCRETE_SPIKE = 2, SNN_NT_EXPONENTIAL_DECAY = 3,
               SNN_NT_CUSTOM = 100 } snn_nt_kinetics;
/* ReceptorKinetics: Approximate iterate_and_spike/mod.rs:427-446, Destexhe :394-425 */
/* ExponentialDecayReceptor :497-533 (attributes receptors$<T>$r$kinetics$r_max, ...$decay_constant) */
typedef enum { SNN_RC_APPROXIMATE = 0, SNN_RC_DESTEXHE = 1, SNN_RC_EXPONENTIAL_DECAY = 2, SNN_RC_CUSTOM = 100 } snn_rc_kinetics;
/* SpikeTrain: PoissonNeuron spike_train/mod.rs:259-371 (GPU generator :380-435), RateSpikeTrain :975-1031 */
/* PresetSpikeTrain :753-833 (attributes internal_clock, counter; firing times via snn_set_firing_times).
 * NeuralRefractoriness of a cell: attribute neural_refractoriness$kind (u32; 0 DeltaDirac :79-88, the default,
 * 1 ExponentialDecay :164-178), decay constant neural_refractoriness$k. */
/* BCMPoissonNeuron :835-970: the Poisson cell + the same activity attributes */
/* SNN_ST_CUSTOM: the ONE spike train generated from a `[spike_train]` block (build_test/nb_macro/src/lib.rs:4812-4905)
 * that a library built with -DSNN_CUSTOM_MODEL_HEADER carries; its variables are attributes under their DSL names.
 * Such a library may also carry a generated NeuralRefractoriness (`[neural_refractoriness]`, lib.rs:5677-5762):
 * neural_refractoriness$kind 2, variables neural_refractoriness$<name>, `decay` = neural_refractoriness$k. */
typedef enum { SNN_ST_NONE = 0, SNN_ST_POISSON = 1, SNN_ST_RATE = 2, SNN_ST_PRESET = 3, SNN_ST_BCM_POISSON = 4,
               SNN_ST_CUSTOM = 100 } snn_spike_train_model;

/* ---- construction (≙ from_lattice / from_network) -------------------------------------- */

/* device: HIP ordinal.  Errors: SNN_ERR_GET_DEVICE, SNN_ERR_QUEUE. */
int snn_network_create(int device, int neuron_model, int nt_kinetics, int receptor_kinetics,
                       int spike_train_model, snn_network_t **out);
int snn_network_destroy(snn_network_t *net);

/* Register a neuron lattice / spike-train lattice by id (LatticeNetwork::add_lattice
 * neuron/mod.rs:1663-1679, add_spike_train_lattice :1682-1699; a duplicate id is an error).  Zero-sized grids are legal. */
int snn_network_add_lattice(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols);
int snn_network_add_spike_train_lattice(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols);

/* Fix the interleaved index space and allocate device state (reference defaults) plus the
 * dense graph of this handle's postsynaptic columns.  snn_network_finalize: whole population.
 * snn_network_finalize_shard: shard `shard_index` of `n_shards` equal slots of
 * stride = round_up(ceil(n_neurons / n_shards), 64) neurons; shard r owns neurons
 * [r*stride, min(n_neurons, (r+1)*stride)).  Errors: SNN_ERR_BUFFER_CREATE. */
int snn_network_finalize(snn_network_t *net);
int snn_network_finalize_shard(snn_network_t *net, uint32_t shard_index, uint32_t n_shards);

int snn_network_sizes(const snn_network_t *net, uint32_t *n_neurons, uint32_t *n_cells,
                      uint32_t *post_begin, uint32_t *post_end);
/* first interleaved index and cell count of a lattice (InterleavingGraphGPU::calculate_index) */
int snn_network_lattice_range(const snn_network_t *net, uint32_t id, uint32_t *first, uint32_t *count);

/* ---- per-cell state (≙ convert_to_gpu / convert_to_cpu) ------------------------------- */

/* `id` selects the lattice; `count` must equal rows*cols (times 3 for per-type attributes).
 * Unknown name or wrong type: SNN_ERR_BAD_ATTR; wrong count: SNN_ERR_DIM_MISMATCH. */
int snn_set_attr_f32(snn_network_t *net, uint32_t id, const char *name, const float *src, size_t count);
int snn_get_attr_f32(snn_network_t *net, uint32_t id, const char *name, float *dst, size_t count);
int snn_set_attr_u32(snn_network_t *net, uint32_t id, const char *name, const uint32_t *src, size_t count);
int snn_get_attr_u32(snn_network_t *net, uint32_t id, const char *name, uint32_t *dst, size_t count);
int snn_set_attr_i32(snn_network_t *net, uint32_t id, const char *name, const int32_t *src, size_t count);
int snn_get_attr_i32(snn_network_t *net, uint32_t id, const char *name, int32_t *dst, size_t count);

/* ---- graph (≙ GraphToGPU::convert_to_gpu / InterleavingGraphGPU::convert_to_gpu) ------- */

/* The reference's flattened form: weights f32[n_tot*n_tot], connections u32[n_tot*n_tot],
 * element [pre*n_tot + post]; connections == 0 means None (graph/mod.rs:310-320, 729-770).
 * Columns of spike-train cells are ignored (they are never postsynaptic). */
/* A connected edge (connections != 0) whose weight is NaN is refused with SNN_ERR_BAD_ARG, here and in snn_set_graph_rows /
 * snn_set_graph_csr: NaN is how the device matrix marks an absent edge, so the reference's Some(NaN) (graph/mod.rs:204-213 --
 * an edge that turns its postsynaptic neuron's input into NaN) cannot be stored.  The dense forms report it after the rows
 * of the call have been imported with those edges absent; the sparse form before anything changes. */
int snn_set_graph_dense(snn_network_t *net, const float *weights, const uint32_t *connections, size_t n_tot);
int snn_get_graph_dense(snn_network_t *net, float *weights, uint32_t *connections, size_t n_tot);
/* Row-block form for matrices that do not fit one host buffer: presynaptic rows
 * [pre_begin, pre_begin+pre_count), n_neurons columns each ([row*n_neurons + post]). */
int snn_set_graph_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count,
                       const float *weights, const uint32_t *connections);
int snn_get_graph_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count,
                       float *weights, uint32_t *connections);
/* Device-side synthetic graph: weight(pre,post) = lo + (hi-lo)*u24(hash(seed, pre*n_neurons+post)),
 * connected unless pre == post (or always, with_diagonal != 0).  Nothing crosses PCIe. */
int snn_fill_graph_synthetic(snn_network_t *net, uint64_t seed, float lo, float hi, int with_diagonal);

/* Sparse form, CSR by LOCAL postsynaptic neuron (the reference's sparse graph, AdjacencyList
 * graph/mod.rs:974-1118, has no GPU form; needed where a dense N x N matrix cannot exist, e.g.
 * BASELINE configs[4]).  snn_network_use_csr must precede finalize; a handle is dense or CSR for life.
 * row_ptr[n_local + 1] (row_ptr[0] == 0, row_ptr[n_local] == nnz), pre_index[nnz] = interleaved
 * presynaptic index, STRICTLY ascending inside each row, weights[nnz].  Stored edges are Some(w),
 * everything else is None.  Same arithmetic as the dense form, bit for bit. */
int snn_network_use_csr(snn_network_t *net, int enable);
int snn_set_graph_csr(snn_network_t *net, const uint64_t *row_ptr, const uint32_t *pre_index,
                      const float *weights, uint64_t nnz);
/* weights in the order they were set (plasticity changes values, never the structure) */
int snn_get_graph_csr(snn_network_t *net, float *weights, uint64_t nnz);

/* ---- switches (Lattice / LatticeNetwork pub fields, neuron/mod.rs:570-586, 1577-1588) -- */

int snn_set_synapses(snn_network_t *net, int electrical_synapse, int chemical_synapse);
/* STDP of one lattice (plasticity/mod.rs:16-39) and its do_plasticity switch */
int snn_set_plasticity(snn_network_t *net, uint32_t id, float a_plus, float a_minus,
                       float tau_plus, float tau_minus, float dt, int do_plasticity);
/* update_grid_history for GridVoltageHistory (f32 snapshot per step) and the spike raster */
int snn_set_history(snn_network_t *net, int voltage_history, int spike_history);
int snn_reset_history(snn_network_t *net);
int snn_get_clock(const snn_network_t *net, uint64_t *clock);
/* internal_clock of the network as LatticeNetworkGPU::from_network hands it over (neuron/gpu_lattices/mod.rs:1630), and the
 * spike-train lattices' own clocks (SpikeTrainLattice::internal_clock, neuron/mod.rs:1318): a network that has already run
 * on the host continues at its clock -- last_firing_time values are absolute step numbers. */
int snn_set_clock(snn_network_t *net, uint64_t clock);
int snn_set_spike_train_clock(snn_network_t *net, uint32_t id, uint64_t clock);
int snn_get_spike_train_clock(snn_network_t *net, uint32_t id, uint64_t *clock);
/* reset_timing (neuron/mod.rs:405-420, 1710-1717): clock = 0, last_firing_time = None everywhere */
int snn_reset_timing(snn_network_t *net);

/* ---- the hot path (≙ run_lattice / run_lattices) --------------------------------------- */

/* Advance `iterations` time-steps.  iterations == 0, an empty network, or both synapse kinds
 * off is a successful no-op (gpu_lattices/mod.rs:1089-1091, 3196-3203).  Requires a whole
 * population handle; sharded handles are driven with the three calls below. */
int snn_run(snn_network_t *net, uint64_t iterations);

/* ---- multi-GPU: one process and one shard handle per GPU ----------------------------------------------
 * The reference has no distributed path; the sharding follows the step's data dependence (neuron/mod.rs:2640-2647:
 * every input of step t reads the state of step t - 1 only).  A shard handle owns the postsynaptic neurons
 * [post_begin, post_end), their columns of the graph and their state; per step it needs the presynaptic state of the
 * neurons owned elsewhere that its rows read.  That is ONE exchange per step, sized by the handle's EXCHANGE PLAN:
 *
 *   per neuron on the wire: 4 B current_voltage (only with electrical synapses) + 4 B per transmitter type that some
 *   NEURON of the network releases (only with chemical synapses) + 1 BIT for the spike;
 *   SNN_EXCHANGE_ALLGATHER (dense graph): every rank's whole slot, one in-place all-gather;
 *   SNN_EXCHANGE_HALO (CSR graph, after snn_halo_commit): per peer exactly the neurons that peer's rows reference,
 *   an all-to-all-v.
 *
 * Three ways to drive it:
 *   (a) snn_run_sharded: the whole step loop inside the library, RCCL called directly (one host call per run);
 *   (b) snn_step_begin -> snn_exchange (RCCL, on the handle's stream) -> snn_step_end;
 *   (c) snn_step_begin -> the caller moves the segments described by snn_exchange_plan_get / snn_exchange_peers with
 *       its own transport (torch.distributed, device-to-device copies between handles of one process) -> snn_step_end.
 * snn_step_begin computes the local neurons' step and packs the outgoing segments; snn_step_end applies the incoming
 * segments (state mirror, last_firing_time), then plasticity on the local columns, histories, clock, spike trains.
 * (After a synapse kind was switched ON between two steps, (b) and (c) start with snn_refresh_begin -> the exchange ->
 * snn_refresh_end, see below; (a) does it by itself.) */
int snn_step_begin(snn_network_t *net);
int snn_step_end(snn_network_t *net);
/* Sparse handles may shard BY LATTICE instead of by one contiguous slot of the index space: shard s owns slab s (equal
 * slabs of ceil(rows*cols / n_shards) neurons rounded up to 64) of EVERY neuron lattice.  For networks whose lattices
 * are wired position to position (BASELINE configs[4]'s ring k -> k+1) the edges between lattices then stay inside a
 * rank and only the borders of the slabs travel.  Call instead of snn_network_finalize_shard, after
 * snn_network_use_csr(net, 1).  snn_set_graph_csr then takes the rows of the OWNED neurons in ascending global order
 * (snn_shard_ranges lists them; a contiguous shard reports its one range).  Without a committed halo plan such a handle
 * trades whole ownerships with every peer (the all-gather's content as an all-to-all-v). */
int snn_network_finalize_shard_by_lattice(snn_network_t *net, uint32_t shard_index, uint32_t n_shards);
int snn_shard_ranges(const snn_network_t *net, uint32_t *begin, uint32_t *end, uint32_t capacity, uint32_t *count);
/* Optional overlap of the exchange with work that does not depend on it.  Results never depend on whether it is called.
 * Dense handles: enqueue, BEFORE the exchange of the previous step has been waited for, the part of the NEXT step's
 * synaptic-input pass that only needs the shard's own neurons as presynaptic rows; the following snn_step_begin then
 * processes the remaining rows.  A no-op when plasticity is on (STDP rewrites W in snn_step_end).
 * Sparse handles with a halo plan and no weight updates: snn_step_begin has enqueued the BORDER slices of the step (the
 * 64-row slices holding a neuron some peer reads) and written the outgoing segments; this call enqueues the INTERIOR
 * slices of the SAME step -- call it once the exchange is under way.  snn_step_end does it itself if nobody did. */
int snn_step_begin_local(snn_network_t *net);

/* A shard handle knows of the neurons owned elsewhere what past exchanges carried.  When the plan comes to need a plane that
 * was not on the wire -- gap junctions or chemical synapses switched on with snn_set_synapses between two steps -- the owners'
 * CURRENT state has to travel once before the next step: every rank calls snn_refresh_begin (packs; *needed = 0 and nothing
 * else when the mirror is current), moves the plan's segments as after snn_step_begin, and calls snn_refresh_end.
 * snn_run_sharded and snn_run_sharded_custom do this by themselves. */
int snn_refresh_begin(snn_network_t *net, int *needed);
int snn_refresh_end(snn_network_t *net);

enum { SNN_EXCHANGE_ALLGATHER = 0, SNN_EXCHANGE_HALO = 1 };
typedef struct snn_exchange_plan {
    int32_t mode;                 /* SNN_EXCHANGE_ALLGATHER | SNN_EXCHANGE_HALO */
    uint32_t n_shards, shard_index, shard_stride;
    uint32_t planes;              /* 32-bit planes per neuron on the wire (0..4) */
    uint32_t plane_id[4];         /* what plane s carries: 0 current_voltage, 2 + k transmitter type k */
    void *send, *recv;            /* device buffers of 32-bit words; all-gather: send lies inside recv (in place) */
    uint64_t send_words, recv_words;
} snn_exchange_plan;
/* The plan in force for the next step (it follows snn_set_synapses, the neurotransmitters$flags of the neuron
 * lattices and snn_halo_commit; query it after configuring and before stepping; buffers stay valid until then). */
int snn_exchange_plan_get(snn_network_t *net, snn_exchange_plan *plan);
/* Per peer p (arrays of n_shards entries, 32-bit words): the segment this handle sends to p is
 * send[send_offset[p] .. + send_words[p]), the segment it receives from p goes to recv[recv_offset[p] .. +
 * recv_words[p]).  All-gather: every peer gets the same segment (own slot) and recv_offset[p] = p * block.  A segment
 * of n neurons holds planes * n words followed by ceil(n / 32) words of spike bits. */
int snn_exchange_peers(snn_network_t *net, uint64_t *send_offset, uint64_t *send_words, uint64_t *recv_offset,
                       uint64_t *recv_words);

/* Halo plan of a CSR shard handle.  snn_set_graph_csr derives, per peer, the ascending global indices of that peer's
 * neurons the local rows read (snn_halo_needs: count, and the list when `indices` has room for it).  Every rank tells
 * every peer what it needs; snn_halo_set_sends stores what `peer` needs from this handle; snn_halo_commit (after all
 * peers' lists are in) switches the handle to SNN_EXCHANGE_HALO.  snn_run_sharded / snn_comm_exchange_halo_lists do
 * this exchange over RCCL themselves.  Without a commit the handle all-gathers whole slots (correct, larger). */
int snn_halo_needs(snn_network_t *net, uint32_t peer, uint32_t *indices, uint32_t capacity, uint32_t *count);
/* Spike-train cells are replicated on every rank (deterministic generators), but a sparse shard handle of a multi-rank
 * run advances only the cells its own rows read: snn_cells_read lists them (ascending indices into the cell population,
 * i.e. global index - n_neurons; every cell for dense, unsharded and single-shard handles).  The state of a cell that
 * is not listed stays at its last written value on this rank. */
int snn_cells_read(snn_network_t *net, uint32_t *indices, uint32_t capacity, uint32_t *count);
int snn_halo_set_sends(snn_network_t *net, uint32_t peer, const uint32_t *indices, uint32_t count);
int snn_halo_commit(snn_network_t *net);

/* RCCL, called by the library itself (librccl.so.1 is opened on first use; a process that already loaded a copy --
 * e.g. PyTorch's -- shares it).  `nccl_comm` is an ncclComm_t: the host's own, or one made by the two helpers below
 * (snn_comm_unique_id on rank 0 -> the 128 bytes travel to every rank by the host's means -> snn_comm_init_rank). */
int snn_comm_unique_id(void *id_128_bytes);
int snn_comm_init_rank(const void *id_128_bytes, int world_size, int rank, int device, void **nccl_comm);
int snn_comm_destroy(void *nccl_comm);
/* ncclCommCount / ncclCommUserRank of a communicator (rank may be null): what a benchmark line quotes as evidence that RCCL
 * itself spans the ranks. */
int snn_comm_count(void *nccl_comm, int *world_size, int *rank);
/* The collectives behind snn_run_sharded / snn_exchange / snn_comm_exchange_halo_lists, replaceable PROCESS-WIDE: a host with
 * another transport (MPI, UCX, a test harness that runs several ranks as threads of one process) keeps the library's loop with
 * its overlap of the exchange and the own-rows work.  Signatures are RCCL's (ncclCommCount, ncclCommUserRank, ncclAllGather,
 * ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd) with void * for ncclComm_t and hipStream_t and int for the data type (always
 * ncclUint32 = 3) and the result (0 = success); `comm` is whatever the host passes to snn_run_sharded.  The functions are
 * called from the thread that runs the library call and must order their work after what that thread has enqueued on `stream`
 * (as RCCL does).  NULL restores RCCL.  While the table is replaced the communicators are the host's own objects:
 * snn_comm_unique_id / _init_rank / _destroy fail with SNN_ERR_BAD_STATE (as they do when librccl is absent), and the table
 * cannot be swapped (SNN_ERR_BAD_STATE) while a thread is inside snn_run_sharded, snn_exchange or
 * snn_comm_exchange_halo_lists.  The library keeps the function pointers until the table is restored. */
typedef struct snn_collectives {
    int (*comm_count)(void *comm, int *count);
    int (*comm_user_rank)(void *comm, int *rank);
    int (*all_gather)(const void *send, void *recv, size_t count, int datatype, void *comm, void *hip_stream);
    int (*send)(const void *buf, size_t count, int datatype, int peer, void *comm, void *hip_stream);
    int (*recv)(void *buf, size_t count, int datatype, int peer, void *comm, void *hip_stream);
    int (*group_start)(void);
    int (*group_end)(void);
} snn_collectives;
int snn_set_collectives(const snn_collectives *table);

/* ---- the peer form of a library-driven run (sparse shard handles, halo exchange, voltage the only plane) ------------
 * One launch per step and no collective: the rows of a shard that a peer reads store {voltage, step tag | spike flag} as ONE
 * 8-byte granule straight into that peer's receive set (peer-mapped memory: the same process, or an IPC mapping), the rows
 * of the next step read a granule when its tag is the step's, and a done counter per peer -- stored by the last workgroup of
 * a step's launch into the neighbours' memory -- says when a set may be overwritten.  RCCL is only needed to trade the
 * addresses (or not at all: they are plain numbers).  Used by snn_run_sharded / snn_run_sharded_custom once the handle is
 * connected: snn_p2p_local on every rank -> the four numbers travel by the host's means -> snn_p2p_connect for every peer
 * this rank exchanges with -> snn_p2p_commit.  A rebuilt exchange plan (other synapse kinds, new halo lists) starts
 * unconnected again.  Failure: a value or a done counter that does not arrive within "halo_peer_spin_limit" polls ends the
 * run call with SNN_ERR_WAIT (the handle is then in the middle of a step; there is no roll-back across ranks).
 * Options: "halo_peer" [1] 0 keeps the collective on a connected handle; "halo_peer_delay" [0] (tests) n = 1 .. 64: up to n sleeps
 * of about 3 us, pseudo-random per call site and step, in front of the granule stores, the polls and the done announcements.
 * Statistic: "halo_peer_steps".
 * Status: tested with 2 - 8 shard handles of ONE process on one GPU (tests/test_gpu_halo_peer.py) and with 2 - 4 PROCESSES on
 * one GPU through the IPC pair below (tests/test_gpu_two_processes.py, forms "*_peer"); across DEVICES (peer access over
 * xGMI) it has never run -- bench.py takes it only with --peer-form. */
/* addresses (in THIS process) of the handle's two receive sets and its done counters; per shard p the granule offset and
 * count of what arrives from p (recv_offsets / recv_counts: [n_shards], may be null) */
int snn_p2p_local(snn_network_t *net, uint64_t *recv0, uint64_t *recv1, uint64_t *flags, uint64_t *recv_offsets, uint64_t *recv_counts);
/* where this handle's values go on shard `peer`: that peer's receive sets and done counters as mapped in this process, and the
 * peer's recv_offsets[this shard] */
int snn_p2p_connect(snn_network_t *net, uint32_t peer, uint64_t peer_recv0, uint64_t peer_recv1, uint64_t peer_flags,
                    uint64_t peer_recv_offset);
int snn_p2p_commit(snn_network_t *net);
/* ranks in different processes: three 64-byte IPC handles (receive set 0, set 1, done counters) of a handle; the peer opens them */
int snn_p2p_ipc_export(snn_network_t *net, void *handles_3x64_bytes);
int snn_p2p_ipc_import(int device, const void *handles_3x64_bytes, uint64_t *recv0, uint64_t *recv1, uint64_t *flags);
/* unmaps what snn_p2p_ipc_import mapped (a 0 address is skipped).  Order of a plan rebuild across ranks: every rank that holds a
 * connection to the handle whose plan changes drops it (closes its mappings, reconnects after the new snn_p2p_local numbers have
 * travelled); until a handle commits a new connection -- or is destroyed -- it keeps the receive sets and done counters of its
 * previous plan alive, so a late store of a neighbour lands in memory that still belongs to it.  Whether a run takes the peer
 * form is part of what the ranks of snn_run_sharded agree on before its first step: all of them or none
 * (SNN_ERR_BAD_STATE otherwise, instead of one rank polling granules while another posts a collective). */
int snn_p2p_ipc_close(int device, uint64_t recv0, uint64_t recv1, uint64_t flags);
/* CSR shard handles: all ranks call it once after snn_set_graph_csr; trades the need lists and commits the halo plan */
int snn_comm_exchange_halo_lists(snn_network_t *net, void *nccl_comm);
/* One exchange of the packed segments, enqueued on the handle's stream (between snn_step_begin and snn_step_end) */
int snn_exchange(snn_network_t *net, void *nccl_comm);
/* `iterations` steps of a shard handle, every rank of the communicator calling it with its own handle: per step
 * kernels -> pack -> ncclAllGather / grouped ncclSend+ncclRecv on a second stream -> unpack -> rest of the step, with
 * the next step's own-rows input pass overlapping the collective where that is valid (see snn_step_begin_local).
 * Sparse handles with a halo plan and no weight updates take two launches per step: k_step_csr over the BORDER slices
 * (the 64-row slices holding a neuron some peer reads), which writes the outgoing segments itself -> the collective,
 * overlapped by k_step_csr over the INTERIOR slices, behind which ride the spike-train cells, the mirror copy (+
 * last_firing_time stamps) of what arrived ONE STEP EARLIER and the clearing of the other set of outgoing spike bitmaps.
 * With voltage the only plane on the wire the rows gather the halo from the received segments themselves ("halo_direct",
 * two alternating sets of segments), so nothing is unpacked between the collective and the next step's rows; otherwise a
 * closing launch k_step_close unpacks first (three launches).  A plan in which nothing travels skips the collective and
 * its stream events altogether.  Blocks until the last step has finished (the mirror then holds the last arrivals).
 * Results are identical to (b) and (c) and to a single-GPU snn_run. */
int snn_run_sharded(snn_network_t *net, void *nccl_comm, uint64_t iterations);
/* The same loop with the HOST'S OWN TRANSPORT (MPI, UCX, a test harness ...) in place of RCCL: once per step, after the
 * outgoing segments have been enqueued on `hip_stream`, `exchange(user, hip_stream)` must move the segments the plan
 * describes (snn_exchange_plan_get / snn_exchange_peers) and return 0; work enqueued on `hip_stream` after it returns
 * must see the received segments (a blocking implementation synchronises the stream, moves the bytes, returns).  A
 * non-zero return stops the run with SNN_ERR_QUEUE.  No overlap of the next step's input pass with the exchange.
 * The segment pointers of snn_exchange_plan_get are fixed for such a run unless the option "halo_direct" is 2: then a
 * sparse halo run alternates between two sets of segments and the function must read the plan's pointers anew at every
 * call (counts and offsets do not change).  snn_exchange_noop is a transport that moves nothing (timing runs). */
typedef int (*snn_exchange_fn)(void *user, void *hip_stream);
int snn_run_sharded_custom(snn_network_t *net, snn_exchange_fn exchange, void *user, uint64_t iterations);
/* An snn_exchange_fn that moves nothing and returns 0: times what ONE rank's step costs without its exchange
 * (profiles/measure_c5_rank_step.py); the received state is then stale, results are not meaningful. */
int snn_exchange_noop(void *user, void *hip_stream);
/* HIP stream the handle launches on (hipStream_t), for ordering collectives against it */
int snn_stream(snn_network_t *net, void **hip_stream);
/* Adopt the caller's stream (e.g. the one its RCCL collectives are ordered against); NULL returns to the
 * handle's own stream.  With an adopted stream snn_step_begin / snn_step_end only ENQUEUE (no host
 * synchronisation): the caller orders its collective on the same stream and calls snn_synchronize (or any
 * blocking getter) when it needs results.  Without one they block until the step's kernels have finished. */
int snn_set_stream(snn_network_t *net, void *hip_stream);
int snn_synchronize(snn_network_t *net);

/* ---- histories (≙ LatticeHistoryGPU::add_from_gpu, gpu_lattices/mod.rs:215-280) -------- */

int snn_history_steps(const snn_network_t *net, uint64_t *steps);
/* [steps][rows*cols] of lattice `id`, oldest step first */
int snn_get_voltage_history(snn_network_t *net, uint32_t id, float *dst, size_t count);
int snn_get_spike_history(snn_network_t *net, uint32_t id, uint8_t *dst, size_t count);

/* BCM rule for lattice `id` instead of STDP (plasticity/mod.rs:72-116; defaults decay 0.1, average_scalar 0.1, dt 0.1):
 * same gating and edge visits as STDP (`do_update` = the neuron spiked), per visit
 * w += (post.activity * (post.activity - post.average_activity / average_scalar) * pre.activity - decay * w) * dt.
 * Needs SNN_MODEL_BCM_IZHIKEVICH (and SNN_ST_BCM_POISSON for spike-train presynaptic cells); unsharded handles.
 * snn_set_plasticity on the same lattice switches back to STDP. */
int snn_set_bcm(snn_network_t *net, uint32_t id, float decay, float average_scalar, float dt, int do_plasticity);

/* ---- reward modulation (≙ RewardModulatedLattice neuron/mod.rs:2719-3417) ---------------- */

/* Makes lattice `id` a reward-modulated lattice: its internal edges carry a TraceRSTDP (plasticity/mod.rs:126-154;
 * `weight` is the graph weight, `c` the trace, 0 initially) and are updated EVERY step by RewardModulatedSTDP
 * (:158-242; defaults dopamine 0, tau_d 20, tau_c 0.0001, a_plus/a_minus 2, tau_plus/tau_minus 4.5, dt 0.1) in the
 * deferred form (both per-step visits of an edge see the step's final last_firing_times; the reference's in-loop
 * form depends on HashSet order).  do_modulation = RewardModulatedLattice::do_modulation (neuron/mod.rs:2744).  The call makes
 * the lattice a reward-modulated lattice for good, whatever do_modulation says: it has no STDP rule of its own from then on
 * (snn_set_plasticity keeps its parameters but not its switch), rewards reach its modulator (:5287-5291), and for the
 * connection visits of OTHER lattices (snn_set_connection_kind) it is a modulated partner; do_modulation 0 only means that
 * its own weights are not updated and its neurons are not visited (:3076, :5113).  16 B per internal synapse are streamed
 * per step. */
int snn_set_reward_modulator(snn_network_t *net, uint32_t id, float dopamine, float tau_d, float tau_c, float a_plus,
                             float a_minus, float tau_plus, float tau_minus, float dt, int do_modulation);
int snn_get_dopamine(snn_network_t *net, uint32_t id, float *dopamine);
/* RewardModulator::update(reward) on every modulated lattice (plasticity/mod.rs:199-201); enqueued on the handle's
 * stream, to be followed by a step.  snn_run_with_reward = Agent::update_and_apply_reward (neuron/mod.rs:3402-3407):
 * apply the reward, then one step; snn_run alone = Agent::update / run_lattice (no reward update). */
int snn_apply_reward(snn_network_t *net, float reward);
int snn_run_with_reward(snn_network_t *net, float reward);
/* TraceRSTDP::c per edge: dense rows [pre_count][n_neurons] (a shard handle touches its own columns), or in the
 * edge order of snn_set_graph_csr. */
int snn_set_trace_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const float *traces);
int snn_get_trace_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *traces);
/* ---- connections BETWEEN lattices in a reward-modulated network (RewardModulatedLatticeNetwork, neuron/mod.rs:3419-3453) ----
 * The reference's connecting graph holds RewardModulatedConnection::{Weight, RewardModulatedWeight}.  After the neurons of a step
 * have been updated (post_neuron_update_step, :5030-5043) it visits the spiking neurons of plain lattices with do_plasticity
 * (update_weights_from_neurons_across_lattices, :4707-4802), then EVERY neuron of the reward-modulated lattices
 * (update_weights_from_neurons_across_reward_lattices, :4855-4977).  A visit of z handles, per partner o in another lattice:
 *   incoming o -> z, kind 2 (Weight):  z plain: the STDP rule of z's lattice adds its delta; z modulated: the rule of o's lattice
 *     -- the PRESYNAPTIC one -- when o sits in a plain lattice, nothing otherwise;
 *   incoming o -> z, kind 1 (RewardModulatedWeight):  ONE visit of the modulator of z's lattice (z modulated) or of o's lattice
 *     (z plain) (plasticity/mod.rs:203-237): the delta goes into TraceRSTDP::dw, every second visit folds dw into the trace c and
 *     clears it, the weight gains c * dopamine;
 *   outgoing z -> o:  the reference looks up the REVERSE connection o -> z (:4768-4771, :4929-4932), applies the rule once more
 *     to that copy with (pre = z, post = o) and stores it as z -> o: weight, trace, dw and counter of z -> o are replaced.
 * Lattices are visited in the order they were added (the reference walks a HashMap; its order between lattices is unspecified).
 * snn_set_connection_kind(pre_id, post_id, kind) tags every connection from lattice (or spike-train lattice) pre_id into
 * neuron lattice post_id (kind 0, the default: the plain LatticeNetwork's rule of snn_set_plasticity, forward lookups only).
 * dw is the `pending` matrix and the counter of the two-visit cycle the `counter` matrix, both per connection (rows as
 * snn_set_trace_rows; counters 0 / 1, one byte each; sparse handles: per stored edge in the order of snn_set_graph_csr, the
 * _csr forms below, after the graph is set -- a new sparse graph drops traces, dw and counters with the edges they belonged
 * to).  Unsharded handles, dense or sparse (k_reward_cross / k_reward_cross_csr).
 * Outside the domain where the reference's visits are defined (it unwraps None there) the next run call returns
 * SNN_ERR_BAD_STATE and snn_last_error names the case: a connection of a visited lattice (modulated, or plain with
 * do_plasticity) without a reverse connection of the same kind; kind 1 where no side has a modulator while one side is plastic,
 * or from a spike train into a plastic plain lattice; kind 2 between a plastic plain lattice and a reward-modulated one; a BCM
 * lattice on such a connection. */
int snn_set_connection_kind(snn_network_t *net, uint32_t pre_id, uint32_t post_id, int kind);
int snn_set_pending_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const float *pending);
int snn_get_pending_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *pending);
int snn_set_counter_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const uint8_t *counters);
int snn_get_counter_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, uint8_t *counters);
int snn_set_traces_csr(snn_network_t *net, const float *traces, uint64_t nnz);
int snn_get_traces_csr(snn_network_t *net, float *traces, uint64_t nnz);
int snn_set_pending_csr(snn_network_t *net, const float *pending, uint64_t nnz);
int snn_get_pending_csr(snn_network_t *net, float *pending, uint64_t nnz);
int snn_set_counters_csr(snn_network_t *net, const uint8_t *counters, uint64_t nnz);
int snn_get_counters_csr(snn_network_t *net, uint8_t *counters, uint64_t nnz);

/* PresetSpikeTrain::firing_times (spike_train/mod.rs:772) of every cell of spike-train lattice `id`: cell i (row-major)
 * fires through times[cell_ptr[i] .. cell_ptr[i+1]), cyclically; cell_ptr has rows*cols + 1 entries, starts at 0 and
 * ends at n_times.  A cell with no times never fires.  Only with SNN_ST_PRESET (else SNN_ERR_BAD_STATE). */
int snn_set_firing_times(snn_network_t *net, uint32_t id, const uint32_t *cell_ptr, const float *times, size_t n_times);

/* update_graph_history (Lattice::update_graph_history, neuron/mod.rs:572; AdjacencyMatrix::update_history
 * graph/mod.rs:278-280): one snapshot of lattice `id`'s internal weights per recorded step,
 * [steps][rows*cols][rows*cols] (presynaptic index first, absent edges 0), on the step axis of snn_history_steps.
 * enable = 1: taken AFTER the step's weight updates (a lone Lattice, neuron/mod.rs:904-910); enable = 2: BEFORE them
 * (LatticeNetwork::iterate / iterate_with_neurotransmission, neuron/mod.rs:2450-2461, 2566-2577); 0 = off.
 * Dense unsharded handles. */
int snn_set_graph_history(snn_network_t *net, uint32_t id, int enable);
int snn_get_graph_history(snn_network_t *net, uint32_t id, float *dst, size_t steps);

/* Strided capture: store the history rows (voltage, raster, reduced rows) of every `every`-th step only -- steps
 * 0, every, 2*every, ... counted from the last (re)start of the record; 1 = every step (the reference's
 * behaviour).  Changing the stride restarts the record.  Spike totals (below) always count every step. */
int snn_set_history_stride(snn_network_t *net, uint32_t every);

/* Reduced histories computed on the device, so that observing a long run does not need the T x N voltage
 * history: per step and neuron lattice the AverageVoltageHistory value (neuron/mod.rs:305-322) and the EEGHistory
 * value (:233-284; defaults reference_voltage 0.007, distance 0.8, conductivity 251), and per neuron the spike total
 * of SpikeHistory::aggregate (:331-360; accumulates until snn_reset_history).  The sums use the canonical
 * 256-chunk order (the reference adds strictly sequentially).  A shard handle reduces over whole lattices (after
 * the exchange it holds every shard's voltages); its spike totals cover the neurons it owns.  Rows share the step
 * axis of snn_history_steps; switching the average or the EEG rows on or off restarts that axis. */
int snn_set_reduced_history(snn_network_t *net, int average_voltage, int eeg, int spike_counts,
                            float reference_voltage, float distance, float conductivity);
int snn_get_average_voltage_history(snn_network_t *net, uint32_t id, float *dst, size_t steps);
int snn_get_eeg_history(snn_network_t *net, uint32_t id, float *dst, size_t steps);
int snn_get_spike_counts(snn_network_t *net, uint32_t id, uint32_t *dst, size_t count);

/* Tuning switches (results never depend on them; defaults in brackets, also settable through the environment when the
 * handle is created: SNN_AMD_<NAME in upper case>): "fused_step" [1] one-launch step for small lattices and for
 * sparse handles; "dense_close" [0] 1: streamed dense matrices on unsharded handles with gap junctions: the last workgroup of a column
 * tile of the input pass updates the tile's neurons in the same launch (k_inputs_dense_close; measured slower than input pass +
 * k_update); "resident_quarters" [1] the one-launch step of small dense networks (at most 512 presynaptic rows) spreads a chunk's 256 rows
 * over four wavefronts -- products in parallel, adds in turn (k_step_resident_q); "cells_in_step" [1] sparse electrical-only handles without weight updates: the spike-train cells advance
 * inside the step's launch; "update_packs" [1] dense shard handles: the neuron update writes the handle's own slot of
 * the all-gather buffer itself (no pack launch); "update_all_planes" [1] dense handles with chemical synapses: 1: the neuron update
 * requests the chunk partials of all planes together, 2: four wavefronts share a column's partials and warm the cache for the update
 * (k_update_wide; 3: without the warming) -- measured slower than 1, 0: plane after plane; "csr_xcd_bands" [1] the sparse step hands its row blocks to the XCDs in contiguous bands;
 * "halo_direct" [1] library-driven runs of sparse shard handles gather the halo from the received segments (0: never,
 * 2: also in snn_run_sharded_custom, see there); "defer_rstdp"
 * [1] reward-modulated weight updates riding on the next input pass; "defer_stdp" [0] 1: STDP updates riding on the next
 * input pass, 2: prepared delta vectors applied by scatter passes; "uniform_params" [1] population-wide parameter
 * values from a device table; "persistent_run" [1] all steps of an snn_run call (of 4 steps or more) on a small
 * network -- neurons, with or without Poisson / Rate cells that release no transmitter; electrical synapses only:
 * <= 4096 rows; with chemical synapses (built-in kinetics): <= 1024 rows; with STDP (electrical synapses, neurons only,
 * at most four lattices, "defer_stdp" 0): <= 1024 rows, the weights live in the workgroups' registers for the run and are
 * committed to the matrix when it has completed -- in ONE launch; "persistent_chem" [1] / "persistent_stdp" [1]: 0 keeps
 * networks with chemical synapses / with STDP on the one-launch-per-step forms
 * (snn_get_stat "persistent_run_stdp_steps": steps whose weight updates ran inside such a launch);
 * "input_shape" [0] 1 | 2 forces the 4- / 2-columns-per-lane shape of the streamed dense input pass (0: chosen by size).
 * Unknown names fail with SNN_ERR_BAD_ARG.
 *
 * Failure semantics of "persistent_run".  The one launch is a spin-wait exchange between workgroups that must all be
 * resident on the device at once.  (1) A probe launch of the same shape, once per handle and grid size, decides whether
 * they can be; if not the handle silently keeps one launch per step.  (2) If they lose sight of each other later all the
 * same (e.g. another process holds compute units with a long kernel), every waiter gives up after
 * "run_resident_spin_limit" [2^24] polls and snn_run ROLLS THE HANDLE BACK: all device state the launch may have written
 * (every per-neuron / per-cell array, the exchange buffer, spike totals, device clocks -- copied aside by one small launch
 * before every such run) is restored, the host clocks and history cursors are reset to their values at the start of the
 * call, the SAME snn_run call then takes its steps with one launch per step, and the handle stays in that mode
 * ("persistent_run" 0; snn_set_option can switch it back on).  The call returns SNN_OK with exactly the result a
 * per-step run gives; the only trace is the statistic "persistent_run_fallbacks".  A handle is therefore never left
 * half-stepped, and SNN_ERR_WAIT is only ever returned for a failed hipStreamSynchronize.  Handles that adopted a
 * caller's stream (snn_set_stream) do not take the one-launch form (its outcome is read after a host synchronisation).
 * "run_resident_fault_step" [0] is a test hook: workgroup 0 withholds the state after that step (1-based) of a launch,
 * which forces path (2); tests/test_gpu_persistent_run.py pins the rollback with it.  "run_timing" [0] makes the one-launch
 * run keep the shader-clock totals of workgroup 0's four phases (poll the voltages, barrier, the canonical sum's turns,
 * update + publish) of its last launch for snn_get_stat. */
int snn_set_option(snn_network_t *net, const char *name, int value);
/* Which step form the handle has used so far, as launch counts since creation: "persistent_run_launches" (k_run_resident:
 * many steps per launch), "persistent_run_steps" (steps those launches covered), "persistent_run_fallbacks" (launches that
 * gave up and were rolled back, see above); with option "run_timing": "run_timing_poll" / "_barrier" / "_turns" /
 * "_update" (shader clocks of workgroup 0 over the last launch) and "run_timing_steps" (its steps); "persistent_run_external_stream" (run calls of 4
 * steps or more that stayed on one launch per step ONLY because the handle runs on a caller's stream, snn_set_stream: the
 * one-launch run reads its outcome after a host synchronisation); "halo_direct_steps"
 * (steps of library-driven runs whose rows gathered the halo from the received segments); the form every step outside a
 * one-launch run took: "steps_dense_one_launch" (k_step_resident), "steps_sparse_one_launch" (k_step_csr over all rows; of those "steps_sparse_image" read the step image -- option "csr_image" [1]:
 * static weights and gap junctions only: 16-byte records of two entries and the slices' presynaptic windows staged in LDS;
 * "image_staged_slices" = slices of the current sparse graph whose window fits, "image_staged_slices_direct" = the same for the image
 * of a shard handle's direct runs, whose halo sources are words of the receive buffer),
 * "steps_sparse_split" (border + interior launches of a shard handle), "steps_dense_close" (k_inputs_dense_close),
 * "steps_two_kernel" (input pass + k_update); and
 * what happened between run calls: "shadow_refreshes" (the two shadow copies of the exchanged state were rebuilt),
 * "view_refreshes" (the spike-train cells' gap-junction values were recomputed), "history_regrows" (the history buffers
 * were reallocated).  Unknown names fail with SNN_ERR_BAD_ARG. */
int snn_get_stat(snn_network_t *net, const char *name, uint64_t *value);

/* ---- test support ------------------------------------------------------------------------ */

/* Option "verify" [0] (SNN_AMD_VERIFY=1): every snn_run call on an unsharded handle takes its steps TWICE from the same device
 * snapshot and compares the two outcomes word for word on the device -- a stepper that is not deterministic shows without any
 * oracle.  Handles with weight updates (STDP, BCM, reward modulation) are covered while their matrices -- weights, traces, dw,
 * counters -- fit a 64 MiB side copy; weight updates a previous call left deferred are applied first.  Statistics "verify_runs", "verify_mismatches", "verify_skipped" (the first one-launch run of a handle
 * lays its snapshot out anew); snn_debug_verify_report names the array, word and the two values of the last mismatch (also
 * printed to stderr).  Option "run_resident_chunk_steps" [2^20]: steps per launch of the one-launch run (a run call of more steps
 * takes several launches, each with its own rollback point); "verify_fault" (the self-check's own test hook) w + 1: word w of the
 * exchange buffer, 2^30 + w: word w of the weights, disturbed once after the second pass; "stdp_columns_form" [0] 1: the incoming-edge scatter of STDP with
 * one lane per 16-byte unit (k_stdp_columns_quads); "pinned_copies" [1] every host <-> device copy of the setters and getters goes through a page-locked buffer of the
 * handle and a memcpy on the calling thread (0: the runtime stages the caller's pageable pointer itself; 2: the 2-D copies of
 * the history and row transfers too -- run in round 6: tests/test_gpu_abi_errors.py (setters and getters with and without the buffer), the seeded tests through the trap and a
 * campaign under it, profiles/r06/README.md; the buffer is the handle's own, like everything else a call on a handle uses: calls
 * on one handle are not re-entrant); "stdp_small" [1]: dense unsharded networks of at most 1024 rows under STDP
 * take spike compaction and both weight scatters of a step in ONE launch (k_stdp_small) instead of four. */
const char *snn_debug_verify_report(snn_network_t *net);
/* restore = 0: keeps a copy of everything a later run call reads (device arrays up to 256 MiB in all, the stepper's host-side
 * cursors) in host memory; restore = 1 puts it back -- the SAME call can then be executed again from identical inputs.  Valid
 * while the handle's structure is unchanged (run calls, attribute and weight writes in between are fine). */
int snn_debug_checkpoint(snn_network_t *net, int restore);
/* Error-path test hook: the n-th allocation the library makes from this call on -- device memory, page-locked memory or a
 * host-side table -- fails once (hipErrorOutOfMemory / std::bad_alloc inside, a status code outside: no exception crosses this
 * ABI, every entry point is a function-try-block); n <= 0 disarms.  Process-wide.  *allocations_so_far (may be null) receives
 * the number of allocations made since the library was loaded, so that a test can walk n over exactly the allocations of a
 * call sequence.  Environment: SNN_AMD_FAIL_ALLOC_AT=n arms it at load time. */
int snn_debug_fail_alloc_at(int64_t n, uint64_t *allocations_so_far);
/* Stray-write hunt (tests/guard_arena.py): every host-side table the library allocates from now on -- including the temporaries
 * its getters download into before they unpack into the caller's array -- comes from `alloc(bytes, tag)` and is handed back
 * through `release(p, bytes)` (which returns 1 when p was its memory, 0 otherwise); both null restores the default.  With a
 * protected arena behind the two functions a transfer that lands in a table after the call that owned it returned faults at
 * the writing instruction.  Process-wide; keep the functions alive for the life of the process. */
typedef void *(*snn_host_alloc_fn)(size_t bytes, const char *tag);
typedef int (*snn_host_release_fn)(void *p, size_t bytes);
int snn_debug_set_host_allocator(snn_host_alloc_fn alloc, snn_host_release_fn release);

/* ---- measurement ----------------------------------------------------------------------- */

/* When enabled, every launch of the synaptic-input kernel is bracketed by HIP events on the
 * handle's stream; snn_profile_read returns the launch count and summed duration since the
 * last snn_profile_reset. */
int snn_profile_enable(snn_network_t *net, int enable);
int snn_profile_reset(snn_network_t *net);
int snn_profile_read(snn_network_t *net, uint64_t *launches, double *total_ms);
/* The same for the plasticity launches of a step (spike compaction + weight updates): steps measured, summed ms */
int snn_profile_read_plasticity(snn_network_t *net, uint64_t *steps, double *total_ms);
/* Synthetic drive -- a device-side input generator for benchmarks and load tests, OFF by default (fraction 0), not part
 * of the reference's semantics: before the step at clock t every neuron q with hash32(seed, t * n_neurons + q) <
 * fraction * 2^32 (the splitmix64 generator of snn_fill_graph_synthetic) has current_voltage set to `voltage`.  With
 * `voltage` above the spike threshold a chosen fraction of the population spikes every step, so STDP can be measured
 * under load.  Applied identically on every shard handle. */
int snn_set_synthetic_drive(snn_network_t *net, uint64_t seed, float fraction, float voltage);
/* ALGORITHMIC bytes ONE launch of the synaptic-input kernel moves (DESIGN.md "Roofline"; what the step has to move, not
 * what the counters saw): 4 B per synapse of the shard (dense); 8 B per stored synapse (sparse) + S bytes of state per OWNED
 * row when the launch is the one-launch step k_step_csr, which also is the neuron update (S by model: Izhikevich 60, leaky
 * 68, Hodgkin-Huxley 140, ...; + 44 B per live transmitter type with chemical synapses) + 28 B per spike-train cell when
 * the cells advance in that launch too; 16 B per internal synapse of a reward-modulated lattice whose weight update rides
 * on the pass (k_inputs_rstdp: weight and trace read and rewritten). */
int snn_input_kernel_bytes(const snn_network_t *net, uint64_t *bytes);

/* ---- errors ----------------------------------------------------------------------------- */

/* Message of the calling thread's last failing call ("" if none). */
const char *snn_last_error(void);
int snn_abi_version(void);
const char *snn_custom_model(void);
/* type names of the generated spike train / refractoriness this library carries ("" when it has none) */
const char *snn_custom_spike_train(void);
const char *snn_custom_refractoriness(void);
const char *snn_custom_neurotransmitter_kinetics(void);
const char *snn_custom_receptor_kinetics(void);
/* the `[receptors]` set of the generated neuron (lib.rs:7017-7600): up to three neurotransmitter types in the exchange's
 * three slots, variables receptors$<name> / receptors$<TYPE>$<name>, kinetics receptors$<TYPE>$r$kinetics$r */
const char *snn_custom_receptors(void);

/* HBM ceilings of the device with the stepper's own access shape (16 B per lane, non-temporal): GB/s of a
 * read-only stream and of a copy (read + write bytes) over `bytes` of device memory, `repeats` launches. */
int snn_probe_bandwidth(int device, uint64_t bytes, int repeats, double *read_gbps, double *copy_gbps);

/* ---- device-function probes (parity tests of the shared scalar formulas) ---------------- */

/* out[i] = f(in[i]) evaluated ON THE GPU by the same device functions the stepper uses:
 * which = 0 exp, 1 pow3, 2 pow4. */
int snn_probe_math(int device, int which, const float *in, float *out, size_t count);

/* The same over a range of binary32 BIT PATTERNS formed on the device: out[i] = f(x_i), x_i = the float with bits
 * first + i * stride (mod 2^32); which = 3 evaluates powf(x_i, y) (generated models, `x ^ n`); which = 4, 5, 6 = exp, pow3, pow4
 * through the branch-free main paths with the full function as fall-back (the form of the Hodgkin-Huxley step).  The GPU parity
 * test walks all 2^32 patterns with it. */
int snn_probe_math_bits(int device, int which, uint32_t first, uint32_t stride, float y, float *out, size_t count);

#ifdef __cplusplus
}
#endif
#endif /* SNN_AMD_H */
