// Host-side state of a handle: the snn_network struct, device allocations, the attribute registry (the reference's
// HashMap<String, BufferGPU> of IterateAndSpikeGPU::convert_to_gpu, neuron/iterate_and_spike/mod.rs:3156-3189),
// attribute transfers and the static per-column counts.  Included by snn_network.hip only (one translation unit).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/snn_amd.h"
#include "snn_kernels_csr.hpp"
#include "snn_kernels_dense_step.hpp"
#include "snn_kernels_exchange.hpp"
#include "snn_kernels_inputs.hpp"
#include "snn_kernels_misc.hpp"
#include "snn_kernels_resident.hpp"
#include "snn_kernels_update.hpp"
#include "snn_layout.hpp"

using namespace snn;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

// ---- the exception barrier of the C ABI ------------------------------------------------------------------------------------
// The reference's seam returns Result<_, GPUError> (error/mod.rs:221-238, gpu_lattices/mod.rs:1089-1091) and include/snn_amd.h
// promises "never abort": no C++ exception may leave an extern "C" entry point (it would terminate a C / Rust / ctypes host).
// EVERY exported definition in snn_network.hip is a function-try-block -- `int snn_x(...) ABI_TRY { ... } ABI_CATCH` -- whose
// handler turns whatever arrives into a status code and a message (tests/test_abi.py checks the translation unit for it):
// std::bad_alloc / std::length_error (a host-side table sized by the caller's numbers) -> SNN_ERR_BUFFER_CREATE, anything else ->
// SNN_ERR_BAD_STATE.  The handle stays destroyable: device allocations are on record in net->allocs the moment they exist.
int abi_exception(const char *entry) noexcept
{
    int code = SNN_ERR_BAD_STATE;
    const char *what = "unknown exception";
    char text[160];
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = SNN_ERR_BUFFER_CREATE; what = "host allocation failed (std::bad_alloc)";
    } catch (const std::length_error &e) {
        code = SNN_ERR_BUFFER_CREATE; snprintf(text, sizeof text, "host table too large (std::length_error: %s)", e.what()); what = text;
    } catch (const std::exception &e) {
        snprintf(text, sizeof text, "%s", e.what()); what = text;
    } catch (...) {
    }
    try {
        g_last_error.assign(entry).append(": ").append(what);
    } catch (...) {          // (the message itself could not be stored: the code still says what happened)
    }
    return code;
}
#define ABI_TRY try
#define ABI_CATCH catch (...) { return abi_exception(__func__); }
#define ABI_CATCH_PTR catch (...) { (void)abi_exception(__func__); return nullptr; }

// ---- allocation-failure hook (test support: snn_debug_fail_alloc_at, SNN_AMD_FAIL_ALLOC_AT) ----------------------------------
// Every allocation the library makes passes alloc_fault_now(): device and page-locked memory through snn_malloc / alloc_streamed /
// the host_malloc and ext_malloc wrappers below, host tables through the allocator of hvec (every std::vector of the library).
// Armed with n, the n-th allocation from then on fails -- hipErrorOutOfMemory or std::bad_alloc -- once.  Process-wide, off by
// default; a relaxed load of one word per allocation when off.
struct AllocFault {
    std::atomic<long long> countdown{0};       // > 0: armed, fails when it reaches 0
    std::atomic<unsigned long long> seen{0};   // allocations since the process started
    AllocFault() { if (const char *e = getenv("SNN_AMD_FAIL_ALLOC_AT")) countdown.store(atoll(e)); }
};
inline AllocFault &alloc_fault() { static AllocFault f; return f; }
inline bool alloc_fault_now()
{
    AllocFault &f = alloc_fault();
    f.seen.fetch_add(1, std::memory_order_relaxed);
    if (f.countdown.load(std::memory_order_relaxed) <= 0) return false;
    return f.countdown.fetch_sub(1, std::memory_order_relaxed) == 1;
}
// Test support (snn_debug_set_host_allocator): the host tables of the library -- among them the temporaries a getter downloads
// into before it unpacks into the caller's array -- can be taken from a caller-supplied allocator.  tests/guard_arena.py hands
// in its protected arena: a table then ends at an inaccessible page and becomes inaccessible for good the moment it is freed, so
// that a transfer that lands AFTER the call that owned the table returned faults at the instruction that writes it.
struct HostAllocHooks {
    void *(*alloc)(size_t bytes, const char *tag);
    int (*release)(void *p, size_t bytes);          // 1: p was the hook's (and is retired now), 0: not its memory
};
inline std::atomic<const HostAllocHooks *> &host_alloc_hooks() { static std::atomic<const HostAllocHooks *> h{nullptr}; return h; }
template <typename T>
struct HostAlloc {
    using value_type = T;
    HostAlloc() = default;
    template <typename U> HostAlloc(const HostAlloc<U> &) {}
    T *allocate(size_t n)
    {
        if (alloc_fault_now() || n > SIZE_MAX / sizeof(T)) throw std::bad_alloc();
        if (const HostAllocHooks *h = host_alloc_hooks().load(std::memory_order_acquire))
            if (void *p = h->alloc(n * sizeof(T), "libsnn_amd host table")) return static_cast<T *>(p);
        return static_cast<T *>(::operator new(n * sizeof(T)));
    }
    void deallocate(T *p, size_t n) noexcept
    {
        if (const HostAllocHooks *h = host_alloc_hooks().load(std::memory_order_acquire))
            if (h->release(p, n * sizeof(T))) return;
        ::operator delete(p);
    }
    template <typename U> bool operator==(const HostAlloc<U> &) const { return true; }
    template <typename U> bool operator!=(const HostAlloc<U> &) const { return false; }
};
template <typename T> using hvec = std::vector<T, HostAlloc<T>>;
inline hipError_t host_malloc(void **out, size_t bytes, unsigned flags)
{
    if (alloc_fault_now()) { *out = nullptr; return hipErrorOutOfMemory; }
    return hipHostMalloc(out, bytes, flags);
}
inline hipError_t ext_malloc(void **out, size_t bytes, unsigned flags)
{
    if (alloc_fault_now()) { *out = nullptr; return hipErrorOutOfMemory; }
    return hipExtMallocWithFlags(out, bytes, flags);
}

#define HIP_TRY(expr, code)                                                                      \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((code), std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

enum AttrType { T_F32 = 0, T_U32 = 1, T_I32 = 2 };
enum AttrStore { S_PLAIN = 0, S_PLAIN_K = 1, S_XPLANE = 2, S_XPLANE_K = 3 };

struct Attr {
    AttrType type;
    AttrStore store;
    void *base;        // S_PLAIN / S_PLAIN_K: device array; planes: unused
    int plane;         // S_XPLANE / S_XPLANE_K
    uint32_t pad;      // stride between types for S_PLAIN_K
    int dirties;       // 1: invalidates the static per-column counts
};

struct LatticeInfo {
    uint32_t id, rows, cols, first, count, slot;
    bool spike_train;
};

} // namespace

struct snn_network {
    int device = 0;
    hipStream_t stream = nullptr;          // the stream every launch goes to
    hipStream_t own_stream = nullptr;      // created with the handle
    bool external_stream = false;          // snn_set_stream adopted a caller's stream
    int model = 0, nt_kind = 0, rc_kind = 0, st_kind = 0;
    bool finalized = false;
    int electrical = 1, chemical = 0;
    long long clock = 0;

    hvec<LatticeInfo> lattices;      // neuron lattices, ascending id after finalize
    hvec<LatticeInfo> st_lattices;   // spike-train lattices
    hvec<long long> st_clock;        // own clocks of the spike-train lattices
    hvec<float> stdp_host;           // [n_lattices][PL_STRIDE], see plasticity_weight
    hvec<uint32_t> plast_host;       // [n_lattices]
    bool any_plasticity = false;
    std::map<uint32_t, bool> lattice_has_nt;    // lattice id -> some neurotransmitters$flags entry is set
    bool any_nt_neurons = false, any_nt_cells = false;
    std::map<uint32_t, uint32_t> lattice_nt_mask;   // lattice id -> bit k: some cell of it releases transmitter type k
    // live transmitter types of the handle (the union over its lattices), refreshed with the static counts: the dense
    // input pass is specialised on their number and leaves the partial planes of the other types untouched (zero)
    uint32_t n_live = K_TYPES, live_type[K_TYPES] = {0, 1, 2};
    uint32_t live_mask_applied = 0xFFFFFFFFu;
    // reward modulation (RewardModulatedLattice): per-lattice modulator table + per-edge trace, allocated on first use
    bool any_modulation = false;           // some lattice has do_modulation set
    bool any_modulated = false;            // some lattice is a reward-modulated lattice (modulating or paused): rewards reach its modulator
    hvec<float> rm_host;            // [n_lattices][RM_STRIDE]
    hvec<uint32_t> rm_on_host;
    float *rm_dev = nullptr;
    uint32_t *rm_on_dev = nullptr;
    float *trace = nullptr;                // dense: [n_tot][ld]; CSR: [sell_entries]
    // connections of a reward-modulated NETWORK that end in a modulated lattice (snn_set_connection_kind, k_reward_cross):
    // conn_kind [n_lattices + n_st_lattices][n_lattices], TraceRSTDP::dw per edge (`pending`, allocated on first use, layout of W),
    // TraceRSTDP::counter per post lattice
    hvec<uint8_t> conn_kind_host;
    uint8_t *conn_kind_dev = nullptr;
    bool any_conn_kind = false;
    float *pending = nullptr;
    float *edge_counter = nullptr;
    uint32_t *cross_bad = nullptr;
    bool cross_checked = false;           // the connection kinds lie where the reference defines their updates (check_reward_cross)
    // Dense handles defer the weight update of step t to the input pass of step t+1 (k_inputs_rstdp: one pass over
    // W and the traces instead of two); any host access to weights / traces / timing flushes it first.
    int defer_rstdp = 1;                   // 0: always the standalone pass (SNN_AMD_DEFER_RSTDP=0)
    bool rstdp_pending = false;
    bool reward_since_defer = false;       // a reward was applied after the deferral: use RM_DOPAMINE_BEFORE

    uint32_t nn = 0, nc = 0, n_tot = 0, n_pad = 0, c_pad = 0;
    uint32_t q0 = 0, q1 = 0, n_loc = 0, ld = 0, n_chunks = 0;
    XLayout xl{0};
    // post-population sharding (snn_network_finalize_shard): this handle owns neurons [q0, q1) = slot shard_index of
    // n_shards equal slots of shard_stride neurons
    bool sharded = false;
    uint32_t n_shards = 1, shard_index = 0, shard_stride = 0;
    // Range-set ownership (snn_network_finalize_shard_by_lattice, sparse handles): shard s owns slab s of EVERY neuron
    // lattice; local rows = the global 64-blocks that hold an owned neuron (RowMap, snn_layout.hpp)
    bool block_mode = false;
    hvec<uint32_t> lattice_slab;                         // per neuron lattice slot: neurons per slab
    hvec<std::pair<uint32_t, uint32_t>> ranges;          // owned [begin, end), ascending (every handle has them)
    uint32_t n_owned = 0;
    hvec<uint32_t> owned_local_host;                     // k-th owned neuron (ascending) -> local row
    hvec<uint32_t> local_row_host;                       // [nn] global neuron -> local row or 0xFFFFFFFF
    uint32_t *own_block_dev = nullptr, *local_row_dev = nullptr;
    unsigned long long *own_mask_dev = nullptr;
    RowMap rowmap{};
    // ---- exchange plan (snn_kernels_exchange.hpp), rebuilt by ensure_exchange_plan when x_dirty ----
    bool x_dirty = true;
    // planes (bit = plane id) whose values of the neurons owned ELSEWHERE are current in this handle's mirror: everything after
    // the attributes were written, then only what the exchange carried.  A plan that needs more (a synapse kind switched on
    // between two runs) calls for one exchange of the current state before the next step (refresh_*, snn_network_exchange.hpp)
    uint32_t mirror_mask = 0xFFFFFFFFu;
    bool x_agreed = false;                      // the ranks of the communicator compared their plans (snn_run_sharded)
    bool refresh_agreed = false;                // ... and some rank's mirror lacked a plane: all refresh before the next run's first step
    int x_mode = SNN_EXCHANGE_ALLGATHER;
    uint32_t x_planes = 0, x_plane_id[WIRE_MAX_PLANES] = {0, 0, 0, 0};
    uint64_t x_block_words = 0;                 // all-gather: words per shard slot
    uint32_t *wire = nullptr;                   // all-gather: [n_shards][block words at the largest plan]
    size_t wire_words = 0;
    // halo (sparse handles): per peer, the neurons of that peer this handle's rows read (need) and the own neurons
    // that peer reads (send); buffers and segment tables sized by the plan
    hvec<hvec<uint32_t>> halo_need, halo_send;
    bool halo_committed = false;
    uint32_t *halo_send_buf = nullptr, *halo_recv_buf = nullptr, *halo_send_idx = nullptr, *halo_recv_idx = nullptr;
    hvec<uint64_t> x_send_off, x_send_words, x_recv_off, x_recv_words;     // per peer, in words
    // device segment tables of the pack / unpack launches: {count, offset, first, list offset} per segment
    uint32_t *seg_count_dev[2] = {nullptr, nullptr}, *seg_first_dev[2] = {nullptr, nullptr};
    uint64_t *seg_offset_dev[2] = {nullptr, nullptr}, *seg_loff_dev[2] = {nullptr, nullptr};
    uint32_t seg_n[2] = {0, 0}, seg_max[2] = {0, 0};
    // one-launch sparse step on shard handles (halo mode, k_step_csr + k_step_close): the 64-row slices that hold a
    // neuron some peer reads (border) and the others (interior); per local row the outgoing-segment positions of its
    // neuron (PackTable); totals of the close launch's unpack / clear jobs
    uint32_t *csr_border_dev = nullptr, *csr_interior_dev = nullptr;
    uint32_t n_border = 0, n_interior = 0;
    uint32_t *pack_ptr_dev = nullptr, *pack_segoff_dev = nullptr, *pack_count_dev = nullptr, *pack_index_dev = nullptr;
    uint32_t recv_total = 0, send_bitmap_words = 0;
    bool send_bits_clean = true;          // every outgoing spike bitmap is zero (what the in-kernel pack ORs into)
    bool update_packed = false;           // this step's own slot of the all-gather buffer was written by k_update
    int update_packs = 1;                 // option "update_packs"
    bool resident_quarters = true;        // option "resident_quarters": the one-launch step with a chunk's rows over four wavefronts
    int update_all_planes = 1;            // option "update_all_planes": 1 all planes' partials in one thread (default), 2 / 3 the wide update (k_update_wide; measured slower)
    bool step_packed = false;             // this step's outgoing segments were written by k_step_csr itself
    bool interior_pending = false;        // the border half of this step is enqueued, the interior slices are not yet
    // Library-driven runs of such a handle (snn_run_sharded): the rows gather the halo from the received segments themselves
    // (csr_plan_direct remaps every halo index to its word of the receive buffer), so nothing is unpacked before the next
    // step's rows.  Two sets of segments alternate (hx_par = set of the step being computed): step t packs into send set
    // t % 2, the exchange fills receive set t % 2, the rows of step t + 1 read it while the exchange of step t + 1 fills the
    // other one.  The mirror copy + last_firing_time stamps of a step's arrivals ride behind the NEXT step's rows
    // (stamp_pending), the last one at the end of the run.
    uint32_t *halo_send_buf2 = nullptr, *halo_recv_buf2 = nullptr;
    uint32_t *csr_plan_direct = nullptr, *halo_word_dev = nullptr;
    bool direct_capable = false;          // the plan has the direct form (sparse, halo mode, voltage the only plane)
    bool peer_capable = false;            // the plan has the PEER form (sparse, halo mode, any planes)
    uint32_t peer_delay = 0;              // option "halo_peer_delay": injected latencies, in s_sleep(127) units (tests)
    bool direct_run = false;              // ... and the run in progress uses it
    int hx_par = 0;
    bool stamp_pending = false;
    uint64_t stat_direct_steps = 0;       // statistic "halo_direct_steps"
    bool tail_done = false;               // this step's jobs behind the rows are enqueued
    int halo_direct = 1;                  // option "halo_direct": 0 never, 1 snn_run_sharded, 2 also snn_run_sharded_custom
    // PEER form of such a run (snn_network_exchange.hpp): the border rows store {value, tag | spike} granules straight into the
    // peers' receive sets, the rows of the next step read them when their tag says so, and a done counter per peer says when a
    // set may be overwritten -- no collective, ONE launch per step.  Needs the peers' addresses (snn_p2p_connect / _commit).
    unsigned long long *p2p_recv[2] = {nullptr, nullptr};     // [recv words] granules each, fine-grained
    uint32_t *p2p_flags = nullptr;                            // [n_shards] done counters, written by the peers (fine-grained)
    uint32_t *p2p_done_blocks = nullptr;
    uint32_t *p2p_failed = nullptr;                           // host-mapped word: a poll gave up
    uint32_t *agree_words_dev = nullptr;                      // agree_on_exchange's one word per rank (kept: no hipFree in a run path)
    uint32_t agree_words_cap = 0;
    uint64_t p2p_recv_words = 0;
    struct P2pPeer { uint64_t recv[2] = {0, 0}, flags = 0, recv_offset = 0; bool set = false; };
    hvec<P2pPeer> p2p_peers;                           // per shard: where this handle's values go on that peer
    unsigned long long **p2p_dst_dev[2] = {nullptr, nullptr}; // per pack entry: the peer's granule, per set
    uint32_t *p2p_peer_dev = nullptr;                         // per pack entry: the peer
    uint32_t **p2p_signal_dev = nullptr;                      // the neighbours' flags[this shard]
    uint32_t p2p_n_signal = 0;
    bool p2p_connected = false;
    hvec<void *> p2p_retired;                          // receive sets / done counters of earlier plans, see p2p_release
    bool peer_run = false;                                    // the run in progress uses the peer form
    uint32_t p2p_epoch = 0;                                   // steps of peer-form runs done so far (tags and done counters)
    uint32_t p2p_spin_limit = 1u << 26;
    int halo_peer = 1;                                        // option "halo_peer": 0 keeps the collective even when connected
    uint64_t stat_peer_steps = 0;                             // statistic "halo_peer_steps"
    // in-library collective (snn_run_sharded): RCCL is ordered on its own stream against the compute stream
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_exchanged = nullptr;

    hvec<void *> allocs;
    std::map<void *, size_t> alloc_bytes;        // dev_alloc'ed arrays and their sizes (run_snapshot copies the small ones)
    // sparse form (CSR by local postsynaptic row); the arrays are replaced by every snn_set_graph_csr
    bool csr = false;
    uint64_t nnz = 0;
    // device: SELL-64 rows (slice_ptr / pre / w / row_len) + per-CSR-edge slot, local row and the transpose index
    uint32_t *csr_ptr = nullptr, *csr_pre = nullptr, *csr_post = nullptr, *csr_t_ptr = nullptr, *csr_t_edge = nullptr;
    uint32_t *csr_row_len = nullptr, *csr_edge_slot = nullptr;
    float *csr_w = nullptr;
    uint64_t sell_entries = 0;
    hvec<uint32_t> edge_slot_host;   // CSR edge -> SELL entry (for snn_get_graph_csr)
    float *W = nullptr;
    float *xbuf = nullptr;
    float *part_i = nullptr, *part_t = nullptr;
    uint32_t *n_in = nullptr, *tcount = nullptr;
    bool counts_dirty = true;
    NeuronArrays na{};
    CellArrays ca{};
    uint32_t *lattice_slot = nullptr;
    float *stdp_dev = nullptr;
    uint32_t *plast_dev = nullptr;
    uint32_t *spike_list = nullptr, *spike_count = nullptr;
    // sparse shard handles: the spike-train cells the local rows read (ascending cell indices); the step iterates only
    // those -- cells are replicated state, and what this rank never reads it need not advance
    hvec<uint32_t> cell_list_host;
    uint32_t *cell_list_dev = nullptr;
    uint32_t n_cells_listed = 0;
    // sparse handles: what the rows read of a cell, two copies (InputsArgs::st_view); rows read cell_view[cell_view_cur],
    // an iteration of the cells writes the other copy and flips
    uint32_t *csr_plan = nullptr;            // gather plan of the row sums (SellGraph::plan), built from csr_pre
    // the step image (snn_kernels_csr.hpp, "STEP IMAGE"): headers + window pieces and the image's plan words are built on the
    // host with the graph; the records {plan word, weight} x 2 are packed on the device when the image is first needed and again
    // after anything changed a weight (img_stale)
    uint32_t *csr_img_hdr = nullptr, *csr_plan_win = nullptr;
    uint4 *csr_img_rec = nullptr;
    uint64_t img_records = 0, img_staged_slices = 0;
    bool img_stale = true, csr_image = true;      // csr_image: option "csr_image"
    // ... and its twin for the runs of a shard handle whose rows gather the halo from the received segments (csr_plan_direct):
    // the same graph with every halo neuron's source moved to its word of the receive buffer; built with the exchange plan
    uint32_t *csr_img_hdr_direct = nullptr, *csr_plan_win_direct = nullptr;
    uint4 *csr_img_rec_direct = nullptr;
    uint64_t img_staged_slices_direct = 0;
    bool img_stale_direct = true;
    hvec<uint32_t> sell_pre_host, slice_ptr_host;     // the SELL indices as set (shard handles: the direct image is built from them)
    uint2 *cell_view[2] = {nullptr, nullptr};
    int cell_view_cur = 0;
    bool cells_stepped = false;      // this step's cells advanced inside k_step_csr (step_end skips their launch)
    int csr_xcd_bands = 1;           // option "csr_xcd_bands"
    int cells_in_step = 1;           // option "cells_in_step": 0 keeps the cells in their own launch
    // uniform-parameter tables (UniformTable, snn_layout.hpp): rescanned when attributes were set
    UniformTable *uni_neuron = nullptr, *uni_cell = nullptr;
    bool uni_dirty = true;
    int uniform_params = 1;               // 0: always read the arrays (SNN_AMD_UNIFORM_PARAMS=0)
    // many steps of a small lattice in one launch (k_run_resident): granule slots, the next free step tag, and a
    // host-visible word the kernel sets when its workgroups could not see each other
    int persistent_run = 1;               // 0: one launch per step (SNN_AMD_PERSISTENT_RUN=0)
    int persistent_chem = 1;              // option "persistent_chem": 0 keeps networks with chemical synapses on the per-step forms
    unsigned long long *run_granules = nullptr;
    unsigned long long *run_partials = nullptr;
    uint32_t run_tag = 1;
    float *run_w_out = nullptr;           // STDP inside the one-launch run: where the workgroups leave their weights (layout of W)
    int persistent_stdp = 1;              // option "persistent_stdp"
    uint64_t stat_run_stdp_steps = 0;
    uint32_t *run_failed = nullptr;       // hipHostMalloc: [0] a run gave up, [1] the co-residency probe said no
    uint32_t run_probed_grid = 0;         // grid size the probe last vouched for
    uint64_t stat_run_launches = 0, stat_run_steps = 0, stat_run_fallbacks = 0;
    uint64_t stat_run_external_stream = 0;      // run calls that kept one launch per step only because the handle runs on a caller's stream
    // which form each step took (statistics "steps_*"): k_step_resident, k_step_csr whole, k_step_csr border + interior,
    // input pass + k_update
    uint64_t stat_steps_sparse_image = 0;
    uint64_t stat_steps_dense_one_launch = 0, stat_steps_sparse_one_launch = 0, stat_steps_sparse_split = 0, stat_steps_two_kernel = 0;
    uint64_t stat_steps_dense_close = 0;         // streamed dense steps whose input pass also updated the neurons (k_inputs_dense_close)
    uint64_t stat_shadow_refreshes = 0, stat_view_refreshes = 0, stat_history_regrows = 0;
    uint32_t run_spin_limit = RUN_RESIDENT_SPIN_LIMIT;   // option "run_resident_spin_limit"
    uint32_t run_fault_step = 0;                         // option "run_resident_fault_step" (test hook, see ResidentRunArgs)
    // every small device array of the handle (all per-neuron / per-cell state, the exchange buffer and its shadows, the
    // device clocks) is copied aside in ONE launch before a one-launch run and copied back if the run gave up
    CopyEntry *snap_table = nullptr;
    uint32_t *snap_buf = nullptr;
    uint32_t snap_entries = 0, snap_max_words = 0;
    size_t snap_allocs_seen = 0;
    size_t snap_words = 0;
    hvec<CopyEntry> snap_table_host;     // the table as uploaded (names of the arrays in a "verify" report)
    size_t verify_words = 0;                    // capacity of each half of verify_buf
    uint32_t run_chunk_steps = 1u << 20;        // option "run_resident_chunk_steps" (test hook): steps per one-launch chunk
    // option "verify" (SNN_AMD_VERIFY=1; tests and campaigns): every snn_run call on a handle without weight updates takes its
    // steps TWICE from the same snapshot and compares the two outcomes on the device (k_compare_table_alt)
    int verify = 0;
    uint32_t verify_fault = 0;                  // option "verify_fault" (test hook): word + 1 of the exchange buffer (2^30 + word of the weights) to disturb once
    uint32_t *verify_buf = nullptr;             // the first outcome, laid out like snap_buf
    uint32_t *verify_report = nullptr;          // device words, see k_compare_table_alt
    int pinned_copies = 1;                      // option "pinned_copies" [1]: see copy_sync (SNN_AMD_PINNED_COPIES=0: the runtime stages pageable pointers itself)
    void *copy_stage = nullptr;                 // its page-locked staging buffer (8 MiB, allocated with the first such copy)
    uint32_t *verify_third = nullptr;           // on a mismatch: the second outcome, while a third execution decides which one repeats
    size_t verify_third_words = 0;
    char *verify_big = nullptr;                 // runs with weight updates: [the matrices at the start | after the first pass]
    size_t verify_big_bytes = 0;
    uint64_t snap_generation = 0;               // how often the snapshot table has been laid out
    uint64_t stat_verify_runs = 0, stat_verify_mismatches = 0, stat_verify_skipped = 0;
    std::string verify_text;                    // what the last mismatch was (snn_debug_verify_report)
    // snn_debug_checkpoint (test support): device arrays and the stepper's host-side cursors as they were at the call
    struct Checkpoint {
        bool valid = false;
        hvec<std::pair<void *, hvec<uint8_t>>> arrays;
        long long clock = 0;
        hvec<long long> st_clock;
        uint64_t hist_steps = 0, hist_tick = 0;
        int shadow_cur = 0, cell_view_cur = 0, persistent_run = 1;
        bool shadow_valid = false, view_dirty = true, counts_dirty = true, uni_dirty = true;
        uint32_t live_mask_applied = 0xFFFFFFFFu, n_live = K_TYPES, live_type[K_TYPES] = {0, 1, 2}, mirror_mask = 0xFFFFFFFFu;
    } checkpoint;
    unsigned long long *run_timing = nullptr;   // SNN_AMD_RUN_TIMING=1: phase clocks of k_run_resident, printed per launch
    int run_timing_opt = 0;                     // option "run_timing": collect them without printing (snn_get_stat)
    unsigned long long run_timing_last[4] = {0, 0, 0, 0};   // workgroup 0, last launch: poll, barrier, turns, update + publish
    uint32_t run_timing_steps = 0;
    int force_shape = 0;                  // 1 | 2: streamed shape of the dense input pass (SNN_AMD_INPUT_SHAPE), 0: by size
    // deferred STDP (dense handles): the update of step t is applied by the input pass of step t + 1
    // 0 (default): the scatter kernels right after the step; 1: the update of step t rides on the input pass of step
    // t + 1; 2: prepared delta vectors, applied right away by scatter passes (SNN_AMD_DEFER_STDP / "defer_stdp").
    // Measured on the quad-row matrix: the scatter kernels win at every spike rate.
    int defer_stdp = 0;
    int stdp_small = 1;                   // option "stdp_small": networks of <= 1024 rows take compaction + both scatters in ONE launch (k_stdp_small)
    int stdp_columns_form = 0;            // option "stdp_columns_form": 0 one thread per presynaptic row, 1 one lane per 16-byte unit (k_stdp_columns_quads)
    bool stdp_pending = false;
    bool stdp_pending_rows_only = false;   // ... and it is the ROW half only ("defer_stdp" 3)
    uint32_t *stdp_rowbits = nullptr;       // [n_chunks][8] one bit per presynaptic row that spiked in the step just closed
    uint32_t *stdp_flag = nullptr;
    float *stdp_dcol = nullptr, *stdp_drow = nullptr;
    uint32_t dcol_stride = 0;
    long long *st_clock_dev = nullptr;
    long long *st_clock_pinned = nullptr;   // page-locked staging of st_clock for the asynchronous upload that opens a run
    long long run_step_offset = 0;
    bool run_active = false;        // a (possibly externally driven) run is open: device clocks are ahead of st_clock
    // fused small-lattice step (k_step_resident): two shadow copies of the exchange buffer + per-tile tickets
    float *shadow[2] = {nullptr, nullptr};
    int shadow_cur = 0;
    bool shadow_valid = false;      // shadow[shadow_cur] == exchange buffer
    int fused_step = 1;             // 0: always take the two-kernel path (SNN_AMD_FUSED_STEP=0)
    int dense_close = 0;            // 1: streamed dense matrices: the last workgroup of a column tile updates its neurons (measured: slower than two kernels)
    uint32_t dense_close_max_chunks = 1u << 30;   // ... only up to this many chunks of presynaptic rows (SNN_AMD_DENSE_CLOSE_MAX_CHUNKS; experiments)
    uint32_t *tile_done = nullptr;  // k_inputs_dense_close: per column tile, the workgroups that have stored their partials (0 between launches)
    uint32_t tile_done_len = 0;
    bool view_dirty = true;         // spike-train gap-junction values must be refreshed before the next inputs
    bool local_inputs_done = false; // this step's LOCAL chunk partials are already enqueued

    std::map<std::string, Attr> neuron_attrs, cell_attrs;

    // histories
    int want_vhist = 0, want_raster = 0;
    // reduced histories: per-lattice average voltage / EEG value per step, per-neuron spike totals
    int want_avg = 0, want_eeg = 0, want_counts = 0;
    // per-lattice weight snapshots (update_graph_history): [cap][count*count] per neuron lattice slot
    hvec<int> want_whist;
    hvec<float *> whist;
    bool any_whist = false;
    float eeg_ref = 0.007f, eeg_dist = 0.8f, eeg_cond = 251.0f;     // EEGHistory defaults, neuron/mod.rs:246-255
    float *summ_avg = nullptr, *summ_eeg = nullptr;                 // [cap][n_lattices]
    uint32_t *spike_counts = nullptr, *lat_first_dev = nullptr, *lat_count_dev = nullptr;
    uint64_t hist_steps = 0, hist_cap = 0;
    hvec<hvec<float>> preset_host;   // PresetSpikeTrain firing times per cell
    float *preset_times_dev = nullptr;
    uint64_t hist_tick = 0;                // steps seen since the record was (re)started
    uint32_t hist_every = 1;               // a row is stored when hist_tick % hist_every == 0
    float *vhist = nullptr, *st_vhist = nullptr;
    unsigned long long *raster = nullptr;

    // synthetic drive (snn_set_synthetic_drive): off when drive_threshold == 0
    uint64_t drive_seed = 0;
    uint32_t drive_threshold = 0;
    float drive_voltage = 0.0f;
    // profiling of the plasticity launches (spike compaction + weight updates), same switch as below
    hvec<std::pair<hipEvent_t, hipEvent_t>> ev_pool_pl;
    size_t ev_used_pl = 0;
    uint64_t prof_launches_pl = 0;
    double prof_ms_pl = 0.0;
    // profiling of the synaptic-input kernel
    int profile = 0;
    hvec<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    hvec<int> ev_counts;     // 1: the launch closes a pass over the graph, 0: first half of a split pass
    size_t ev_used = 0;
    uint64_t prof_launches = 0;
    double prof_ms = 0.0;
};

namespace {
inline bool recording(const snn_network *net)
{
    return net->want_vhist || net->want_raster || net->want_avg || net->want_eeg || net->any_whist;
}
// does the step being computed store its history rows (strided capture: every hist_every-th step)
inline bool record_now(const snn_network *net)
{
    return recording(net) && net->hist_tick % net->hist_every == 0;
}
} // namespace

namespace {

// Big streamed arrays (the synapse matrix, the trace matrix).  SNN_AMD_CONTIGUOUS=1 asks for PHYSICALLY CONTIGUOUS memory
// first (larger page-table fragments).  Off by default: it is not a uniform gain -- C3's input pass 164.4 against 171.3 us in
// one call, 175.5 against 171.5 us in another, the reward-modulated pass 11.38 against 10.74 ms (profiles/experiments/README.md).
// (alloc_streamed, below)
// hipMalloc, and -- debugging aid, SNN_AMD_POISON=<hex word> (e.g. 7fc12345) -- every fresh allocation filled with that word
// instead of what the allocator hands out (zeros in a young process, another handle's remains in an old one): a read of
// memory nobody initialised then shows in EVERY run
inline hipError_t poison_if_asked(void *p, size_t bytes)
{
    static const char *poison = getenv("SNN_AMD_POISON");
    if (!poison || !p) return hipSuccess;
    hipError_t e = hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p), (int)strtoul(poison, nullptr, 16), bytes / 4);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
template <typename T>
inline hipError_t snn_malloc(T **out, size_t bytes)
{
    if (alloc_fault_now()) { *out = nullptr; return hipErrorOutOfMemory; }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(out), bytes);
    return e != hipSuccess ? e : poison_if_asked(*out, bytes);
}

// Host <-> device transfers of the setters / getters and the fills of fresh buffers: on the HANDLE'S OWN stream, and waited
// for.  The handle's stream is hipStreamNonBlocking -- nothing orders it against the null stream -- so a blocking null-stream
// hipMemcpy / hipMemset in front of a kernel on it is correct only as long as the runtime completes the transfer before it
// returns; with everything on one stream the order no longer rests on that (round 5: no null-stream call after finalize).
// "pinned_copies" (on by default since the end of round 5; SNN_AMD_PINNED_COPIES=0 switches it off): host <-> device copies never hand
// the runtime a pageable pointer.  The device side of the transfer goes to / from a page-locked buffer of the handle -- a DMA
// whose completion is the stream's -- and the bytes move between that buffer and the caller's memory by memcpy on the calling
// thread.  What it excludes: the runtime's own staging of pageable copies, whose host-side half runs on a runtime thread
// (campaign E, profiles/r05/README.md: under 24 worker processes on 16 cores one execution ended with two words of the oracle's
// memory overwritten by what looks like a spike-raster word, another with a voltage history that did not match its own final state).
inline hipError_t copy_sync(snn_network *net, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (net->pinned_copies && bytes && (kind == hipMemcpyDeviceToHost || kind == hipMemcpyHostToDevice)) {
        constexpr size_t STAGE = (size_t)8 << 20;
        if (!net->copy_stage) {
            const hipError_t e = host_malloc(&net->copy_stage, STAGE, hipHostMallocDefault);
            if (e != hipSuccess) return e;
        }
        for (size_t off = 0; off < bytes; off += STAGE) {
            const size_t n = std::min(STAGE, bytes - off);
            if (kind == hipMemcpyHostToDevice) memcpy(net->copy_stage, static_cast<const char *>(src) + off, n);
            hipError_t e = hipMemcpyAsync(kind == hipMemcpyHostToDevice ? static_cast<char *>(dst) + off : static_cast<char *>(net->copy_stage),
                                          kind == hipMemcpyHostToDevice ? static_cast<const char *>(net->copy_stage) : static_cast<const char *>(src) + off,
                                          n, kind, net->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(net->stream);
            if (e != hipSuccess) return e;
            if (kind == hipMemcpyDeviceToHost) memcpy(static_cast<char *>(dst) + off, net->copy_stage, n);
        }
        return hipSuccess;
    }
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, net->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(net->stream);
}
// ("pinned_copies" 2 -- not the default: written after the round's last GPU minute, so never run -- takes the 2-D copies of the
// history and row getters / setters the same way: rows packed in the page-locked buffer, moved row by row on the calling thread)
inline hipError_t copy2d_sync(snn_network *net, void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height,
                              hipMemcpyKind kind)
{
    if (net->pinned_copies >= 2 && width && height && (kind == hipMemcpyDeviceToHost || kind == hipMemcpyHostToDevice)) {
        constexpr size_t STAGE = (size_t)8 << 20;
        if (width > STAGE) return hipErrorInvalidValue;
        if (!net->copy_stage) {
            const hipError_t e = host_malloc(&net->copy_stage, STAGE, hipHostMallocDefault);
            if (e != hipSuccess) return e;
        }
        const size_t rows_per_hop = std::max<size_t>(1, STAGE / width);
        char *stage = static_cast<char *>(net->copy_stage);
        for (size_t r = 0; r < height; r += rows_per_hop) {
            const size_t rows = std::min(rows_per_hop, height - r);
            hipError_t e;
            if (kind == hipMemcpyHostToDevice) {
                for (size_t i = 0; i < rows; ++i) memcpy(stage + i * width, static_cast<const char *>(src) + (r + i) * spitch, width);
                e = hipMemcpy2DAsync(static_cast<char *>(dst) + r * dpitch, dpitch, stage, width, width, rows, kind, net->stream);
            } else {
                e = hipMemcpy2DAsync(stage, width, static_cast<const char *>(src) + r * spitch, spitch, width, rows, kind, net->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(net->stream);
            if (e != hipSuccess) return e;
            if (kind == hipMemcpyDeviceToHost)
                for (size_t i = 0; i < rows; ++i) memcpy(static_cast<char *>(dst) + (r + i) * dpitch, stage + i * width, width);
        }
        return hipSuccess;
    }
    const hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, net->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(net->stream);
}
inline hipError_t memset_sync(snn_network *net, void *dst, int value, size_t bytes)
{
    const hipError_t e = hipMemsetAsync(dst, value, bytes, net->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(net->stream);
}

hipError_t alloc_streamed(void **out, size_t bytes)
{
    static const bool contiguous = [] { const char *e = getenv("SNN_AMD_CONTIGUOUS"); return e && e[0] == '1'; }();
    if (alloc_fault_now()) { *out = nullptr; return hipErrorOutOfMemory; }
    if (contiguous && bytes >= ((size_t)256 << 20) &&
        hipExtMallocWithFlags(out, bytes, hipDeviceMallocContiguous) == hipSuccess)
        return hipSuccess;
    (void)hipGetLastError();
    return hipMalloc(out, bytes);
}

int dev_alloc(snn_network *net, void **out, size_t bytes)
{
    *out = nullptr;
    if (bytes == 0) bytes = 256;
    HIP_TRY(alloc_streamed(out, bytes), SNN_ERR_BUFFER_CREATE);
    net->allocs.push_back(*out);
    net->alloc_bytes[*out] = bytes;
    HIP_TRY(poison_if_asked(*out, bytes), SNN_ERR_BUFFER_WRITE);
    return SNN_OK;
}

template <typename T>
int dev_alloc_t(snn_network *net, T **out, size_t count)
{
    return dev_alloc(net, reinterpret_cast<void **>(out), count * sizeof(T));
}

int fill_f32(snn_network *net, float *p, size_t n, float v)
{
    if (n == 0) return SNN_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill_f32, dim3(blocks), dim3(256), 0, net->stream, p, n, v);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}
int fill_u32(snn_network *net, uint32_t *p, size_t n, uint32_t v)
{
    if (n == 0) return SNN_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, net->stream, p, n, v);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

const LatticeInfo *find_lattice(const snn_network *net, uint32_t id)
{
    for (const auto &l : net->lattices) if (l.id == id) return &l;
    for (const auto &l : net->st_lattices) if (l.id == id) return &l;
    return nullptr;
}

void reg(std::map<std::string, Attr> &m, const char *name, AttrType t, AttrStore s, void *base, int plane,
         uint32_t pad, int dirties = 0)
{
    m[name] = Attr{t, s, base, plane, pad, dirties};
}

// Allocate one f32 per-neuron array, fill with `def`, register under `name`.
int neuron_f32(snn_network *net, float **field, const char *name, float def)
{
    int rc = dev_alloc_t(net, field, net->n_pad);
    if (rc) return rc;
    rc = fill_f32(net, *field, net->n_pad, def);
    if (rc) return rc;
    if (name) reg(net->neuron_attrs, name, T_F32, S_PLAIN, *field, 0, 0);
    return SNN_OK;
}
int cell_f32(snn_network *net, float **field, const char *name, float def)
{
    int rc = dev_alloc_t(net, field, net->c_pad);
    if (rc) return rc;
    rc = fill_f32(net, *field, net->c_pad, def);
    if (rc) return rc;
    if (name) reg(net->cell_attrs, name, T_F32, S_PLAIN, *field, 0, 0);
    return SNN_OK;
}
// [3][pad] block with per-type defaults
int typed_f32(snn_network *net, float **field, uint32_t pad, float d0, float d1, float d2)
{
    int rc = dev_alloc_t(net, field, (size_t)K_TYPES * pad);
    if (rc) return rc;
    const float d[3] = {d0, d1, d2};
    for (int k = 0; k < K_TYPES; ++k) {
        rc = fill_f32(net, *field + (size_t)k * pad, pad, d[k]);
        if (rc) return rc;
    }
    return SNN_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

int build_state(snn_network *net)
{
    NeuronArrays &n = net->na;
    CellArrays &c = net->ca;
    const uint32_t np = net->n_pad, cp = net->c_pad;
    auto &A = net->neuron_attrs;
    auto &CA = net->cell_attrs;

    // uniform-parameter tables, all "not uniform" until the first scan
    TRY(dev_alloc_t(net, &net->uni_neuron, 1));
    TRY(dev_alloc_t(net, &net->uni_cell, 1));
    HIP_TRY(hipMemsetAsync(net->uni_neuron, 0, sizeof(UniformTable), net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemsetAsync(net->uni_cell, 0, sizeof(UniformTable), net->stream), SNN_ERR_BUFFER_WRITE);
    n.uni = net->uni_neuron;
    c.uni = net->uni_cell;
    // exchanged planes
    TRY(dev_alloc_t(net, &net->xbuf, (size_t)NUM_PLANES * net->xl.stride));
    HIP_TRY(hipMemsetAsync(net->xbuf, 0, (size_t)NUM_PLANES * net->xl.stride * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    if (net->sharded) {
        // all-gather wire buffer at its largest plan: 4 planes + the spike bitmap per slot
        const size_t words = (size_t)net->n_shards * ((size_t)WIRE_MAX_PLANES * net->shard_stride + net->shard_stride / 32);
        TRY(dev_alloc_t(net, &net->wire, words));
        net->wire_words = words;
        HIP_TRY(hipMemsetAsync(net->wire, 0, words * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    }
    n.xbuf = net->xbuf;
    n.xl = net->xl;
    n.n_pad = np;
    reg(A, "current_voltage", T_F32, S_XPLANE, nullptr, PLANE_V, 0);
    reg(A, "is_spiking", T_U32, S_XPLANE, nullptr, PLANE_SPIKE, 0);
    reg(A, "neurotransmitters$t", T_F32, S_XPLANE_K, nullptr, PLANE_T0, 0);

    // reference defaults: Izhikevich integrate_and_fire/mod.rs:1198-1220, LIF :149-171,
    // Hodgkin-Huxley hodgkin_huxley/mod.rs:80-98 + ion_channels/mod.rs:23-31, 205-215, 255-264, 299-307
    const bool bcm = net->model == SNN_MODEL_BCM_IZHIKEVICH;
    const bool cust = net->model == SNN_MODEL_CUSTOM;
    const bool izh = net->model == SNN_MODEL_IZHIKEVICH || bcm, lif = net->model == SNN_MODEL_LIF;
    const bool qif = net->model == SNN_MODEL_QUADRATIC_INTEGRATE_AND_FIRE, slif = net->model == SNN_MODEL_SIMPLE_LIF;
    const bool alif = net->model == SNN_MODEL_ADAPTIVE_LIF, aelif = net->model == SNN_MODEL_ADAPTIVE_EXP_LIF;
    const bool adp = alif || aelif, lizh = net->model == SNN_MODEL_LEAKY_IZHIKEVICH;
    const float v0 = cust ? custom::DEFAULT_VOLTAGE : ((lif || qif || slif || adp) ? -75.0f : -65.0f);
    TRY(fill_f32(net, net->xbuf + (size_t)PLANE_V * net->xl.stride, net->xl.stride, v0));    // initial voltage
    TRY(neuron_f32(net, &n.gap_conductance, "gap_conductance", cust ? custom::DEFAULT_GAP : (slif ? 10.0f : 7.0f)));
    TRY(neuron_f32(net, &n.dt, "dt", cust ? custom::DEFAULT_DT : (net->model == SNN_MODEL_HODGKIN_HUXLEY ? 0.01f : 0.1f)));
    TRY(neuron_f32(net, &n.c_m, "c_m", cust ? custom::DEFAULT_C_M : (net->model == SNN_MODEL_HODGKIN_HUXLEY ? 1.0f : 100.0f)));
    TRY(neuron_f32(net, &n.v_th, "v_th", (izh || lizh) ? 30.0f : ((lif || qif || slif || adp) ? -55.0f : 0.0f)));
    TRY(dev_alloc_t(net, &n.last_firing_time, np));
    HIP_TRY(hipMemsetAsync(n.last_firing_time, 0xFF, (size_t)np * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    reg(A, "last_firing_time", T_I32, S_PLAIN, n.last_firing_time, 0, 0);

    const bool izh_like = izh || lizh, lif_like = lif || adp;
    TRY(neuron_f32(net, &n.w_value, (izh_like || adp) ? "w_value" : nullptr, adp ? 0.0f : 30.0f));
    TRY(neuron_f32(net, &n.a, izh_like ? "a" : nullptr, 0.02f));
    TRY(neuron_f32(net, &n.b, izh_like ? "b" : nullptr, 0.2f));
    TRY(neuron_f32(net, &n.c, izh_like ? "c" : nullptr, -55.0f));
    TRY(neuron_f32(net, &n.d, izh_like ? "d" : nullptr, 8.0f));
    TRY(neuron_f32(net, &n.tau_m, (izh_like || lif_like || qif) ? "tau_m" : nullptr, izh ? 1.0f : (qif ? 100.0f : 10.0f)));

    TRY(neuron_f32(net, &n.v_reset, (lif_like || qif || slif) ? "v_reset" : nullptr, -75.0f));
    TRY(neuron_f32(net, &n.refractory_count, (lif_like || qif) ? "refractory_count" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.tref, (lif_like || qif) ? "tref" : nullptr, 10.0f));
    TRY(neuron_f32(net, &n.leak_constant, lif_like ? "leak_constant" : nullptr, -1.0f));
    TRY(neuron_f32(net, &n.integration_constant, (lif_like || qif) ? "integration_constant" : nullptr, 1.0f));
    // adaptive models, integrate_and_fire/mod.rs:969-996, 1105-1130
    TRY(neuron_f32(net, &n.adp_alpha, adp ? "alpha" : nullptr, 6.0f));
    TRY(neuron_f32(net, &n.adp_beta, adp ? "beta" : nullptr, 10.0f));
    TRY(neuron_f32(net, &n.slope_factor, aelif ? "slope_factor" : nullptr, 1.0f));
    // variables of the generated model: registered under their DSL names (a name such as v_th replaces the common one)
    for (int k = 0; k < CUSTOM_MAX_VARS; ++k) n.custom[k] = nullptr;
    if (cust)
        for (int k = 0; k < custom::NVARS; ++k) TRY(neuron_f32(net, &n.custom[k], custom::NAMES[k], custom::DEFAULTS[k]));
    // BCMIzhikevichNeuron's activity bookkeeping, integrate_and_fire/mod.rs:1385-1396, defaults :1425-1430
    TRY(neuron_f32(net, &n.bcm_avg, bcm ? "average_activity" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.bcm_cur, bcm ? "current_activity" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.bcm_clock, bcm ? "firing_rate_clock" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.bcm_window, bcm ? "firing_rate_window" : nullptr, 500.0f));
    TRY(dev_alloc_t(net, &n.bcm_period, np));
    TRY(fill_u32(net, n.bcm_period, np, 3));
    TRY(dev_alloc_t(net, &n.bcm_num_spikes, np));
    TRY(fill_u32(net, n.bcm_num_spikes, np, 0));
    if (bcm) {
        reg(A, "period", T_U32, S_PLAIN, n.bcm_period, 0, 0);
        reg(A, "num_spikes", T_U32, S_PLAIN, n.bcm_num_spikes, 0, 0);
    }
    // reference buffer names of the two models with a reference GPU implementation
    // (integrate_and_fire/mod.rs:729-773, 1700-1740)
    TRY(neuron_f32(net, &n.qif_alpha, qif ? "alpha" : nullptr, 1.0f));
    TRY(neuron_f32(net, &n.qif_v_c, qif ? "v_c" : nullptr, -60.0f));
    TRY(neuron_f32(net, &n.slif_g, slif ? "g" : nullptr, -0.1f));
    TRY(neuron_f32(net, &n.slif_e, slif ? "e" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.e_l, (lif_like || lizh) ? "e_l" : nullptr, lizh ? -65.0f : -75.0f));
    TRY(neuron_f32(net, &n.g_l, lif_like ? "g_l" : nullptr, 10.0f));

    const bool hh = net->model == SNN_MODEL_HODGKIN_HUXLEY;
    TRY(neuron_f32(net, &n.m_state, hh ? "na_channel$m$state" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.h_state, hh ? "na_channel$h$state" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.n_state, hh ? "k_channel$n$state" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.m_alpha, hh ? "na_channel$m$alpha" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.m_beta, hh ? "na_channel$m$beta" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.h_alpha, hh ? "na_channel$h$alpha" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.h_beta, hh ? "na_channel$h$beta" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.n_alpha, hh ? "k_channel$n$alpha" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.n_beta, hh ? "k_channel$n$beta" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.g_na, hh ? "na_channel$g_na" : nullptr, 120.0f));
    TRY(neuron_f32(net, &n.e_na, hh ? "na_channel$e_na" : nullptr, 50.0f));
    TRY(neuron_f32(net, &n.g_k, hh ? "k_channel$g_k" : nullptr, 36.0f));
    TRY(neuron_f32(net, &n.e_k, hh ? "k_channel$e_k" : nullptr, -77.0f));
    TRY(neuron_f32(net, &n.g_k_leak, hh ? "k_leak_channel$g_k_leak" : nullptr, 0.3f));
    TRY(neuron_f32(net, &n.e_k_leak, hh ? "k_leak_channel$e_k_leak" : nullptr, -55.0f));
    TRY(neuron_f32(net, &n.na_current, hh ? "na_channel$current" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.k_current, hh ? "k_channel$current" : nullptr, 0.0f));
    TRY(neuron_f32(net, &n.k_leak_current, hh ? "k_leak_channel$current" : nullptr, 0.0f));
    TRY(dev_alloc_t(net, &n.was_increasing, np));
    TRY(fill_u32(net, n.was_increasing, np, 0));
    if (hh) reg(A, "was_increasing", T_U32, S_PLAIN, n.was_increasing, 0, 0);

    // neurotransmitters (iterate_and_spike/mod.rs:136-145, 174-182) -- absent by default (flags 0)
    TRY(typed_f32(net, &n.nt_t_max, np, 1.0f, 1.0f, 1.0f));
    // clearance_constant of the Approximate kinetics / decay_constant of ExponentialDecay (:336-343) share storage
    const float nt_c = net->nt_kind == SNN_NT_EXPONENTIAL_DECAY ? 2.0f : 0.01f;
    TRY(typed_f32(net, &n.nt_clearance, np, nt_c, nt_c, nt_c));
    TRY(typed_f32(net, &n.nt_v_p, np, 2.0f, 2.0f, 2.0f));
    TRY(typed_f32(net, &n.nt_k_p, np, 5.0f, 5.0f, 5.0f));
    TRY(dev_alloc_t(net, &n.nt_flags, (size_t)K_TYPES * np));
    TRY(fill_u32(net, n.nt_flags, (size_t)K_TYPES * np, 0));
    reg(A, "neurotransmitters$t_max", T_F32, S_PLAIN_K, n.nt_t_max, 0, np);
    reg(A, "neurotransmitters$clearance_constant", T_F32, S_PLAIN_K, n.nt_clearance, 0, np);
    reg(A, "neurotransmitters$decay_constant", T_F32, S_PLAIN_K, n.nt_clearance, 0, np);
    reg(A, "neurotransmitters$v_p", T_F32, S_PLAIN_K, n.nt_v_p, 0, np);
    reg(A, "neurotransmitters$k_p", T_F32, S_PLAIN_K, n.nt_k_p, 0, np);
    reg(A, "neurotransmitters$flags", T_U32, S_PLAIN_K, n.nt_flags, 0, np, 1);
    for (int j = 0; j < CUSTOM_KINETICS_MAX_VARS; ++j) n.nt_custom[j] = n.rc_custom[j] = nullptr;
    if (net->nt_kind == SNN_NT_CUSTOM)             // generated kinetics: its variables, one value per type
        for (int j = 0; j < custom_nt::NVARS; ++j) {
            const float d = custom_nt::DEFAULTS[j];
            TRY(typed_f32(net, &n.nt_custom[j], np, d, d, d));
            reg(A, (std::string("neurotransmitters$") + custom_nt::NAMES[j]).c_str(), T_F32, S_PLAIN_K, n.nt_custom[j], 0, np);
        }

    // receptors (iterate_and_spike/mod.rs:1085-1094, 1115-1125, 1148-1157, 417-425)
    TRY(typed_f32(net, &n.rc_g, np, 1.0f, 0.6f, 1.2f));
    TRY(typed_f32(net, &n.rc_e, np, 0.0f, 0.0f, -80.0f));
    TRY(typed_f32(net, &n.rc_mg, np, 0.0f, 0.3f, 0.0f));
    TRY(typed_f32(net, &n.rc_r, np, 0.0f, 0.0f, 0.0f));
    TRY(typed_f32(net, &n.rc_alpha, np, 1.0f, 1.0f, 1.0f));
    // ExponentialDecayReceptor (:501-533): r_max lives in the alpha array, decay_constant in the beta array
    const float rc_b = net->rc_kind == SNN_RC_EXPONENTIAL_DECAY ? 2.0f : 1.0f;
    TRY(typed_f32(net, &n.rc_beta, np, rc_b, rc_b, rc_b));
    TRY(typed_f32(net, &n.rc_current, np, 0.0f, 0.0f, 0.0f));
    TRY(dev_alloc_t(net, &n.rc_flags, (size_t)K_TYPES * np));
    TRY(fill_u32(net, n.rc_flags, (size_t)K_TYPES * np, 0));
    reg(A, "receptors$flags", T_U32, S_PLAIN_K, n.rc_flags, 0, np);
    static const char *TN[3] = {"AMPA", "NMDA", "GABA"};
    for (int k = 0; k < K_TYPES; ++k) {
        const std::string p = std::string("receptors$") + TN[k];
        reg(A, (p + "_g").c_str(), T_F32, S_PLAIN, n.rc_g + (size_t)k * np, 0, 0);
        reg(A, (p + "_e").c_str(), T_F32, S_PLAIN, n.rc_e + (size_t)k * np, 0, 0);
        reg(A, (p + "_current").c_str(), T_F32, S_PLAIN, n.rc_current + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$r").c_str(), T_F32, S_PLAIN, n.rc_r + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$alpha").c_str(), T_F32, S_PLAIN, n.rc_alpha + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$beta").c_str(), T_F32, S_PLAIN, n.rc_beta + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$r_max").c_str(), T_F32, S_PLAIN, n.rc_alpha + (size_t)k * np, 0, 0);
        reg(A, (p + "$r$kinetics$decay_constant").c_str(), T_F32, S_PLAIN, n.rc_beta + (size_t)k * np, 0, 0);
    }
    reg(A, "receptors$NMDA_mg", T_F32, S_PLAIN, n.rc_mg + (size_t)1 * np, 0, 0);
    for (int j = 0; j < CUSTOM_RECEPTORS_MAX_VARS; ++j) n.rx_custom[j] = nullptr;
    if (SNN_HAVE_CUSTOM_RECEPTORS && cust && !custom_receptors::MULTI_STATE) {
        // the generated neuron's receptor set with one state per type: the kinetics of type k under the type's own name
        for (int k = 0; k < custom_receptors::NTYPES; ++k) {
            const std::string p = std::string("receptors$") + custom_receptors::NT_NAMES[k];
            reg(A, (p + "$r$kinetics$r").c_str(), T_F32, S_PLAIN, n.rc_r + (size_t)k * np, 0, 0);
            reg(A, (p + "$r$kinetics$alpha").c_str(), T_F32, S_PLAIN, n.rc_alpha + (size_t)k * np, 0, 0);
            reg(A, (p + "$r$kinetics$beta").c_str(), T_F32, S_PLAIN, n.rc_beta + (size_t)k * np, 0, 0);
        }
    }
    if (net->rc_kind == SNN_RC_CUSTOM)
        for (int j = 0; j < custom_rc::NVARS; ++j) {
            const float d = custom_rc::DEFAULTS[j];
            TRY(typed_f32(net, &n.rc_custom[j], np, d, d, d));
            for (int k = 0; k < K_TYPES; ++k)
                reg(A, (std::string("receptors$") + TN[k] + "$r$kinetics$" + custom_rc::NAMES[j]).c_str(), T_F32, S_PLAIN,
                    n.rc_custom[j] + (size_t)k * np, 0, 0);
        }
    if (SNN_HAVE_CUSTOM_RECEPTORS && cust) {
        // the set's own variables, registered LAST: a set with several states per type keeps each state's r and kinetics
        // variables among them (receptors$<Type>$<state>$kinetics$<var>), and a type of the set may carry the name of a
        // built-in type (the reference's DopaGluGABA has a GABA) -- the name then means the set's variable
        for (int j = 0; j < custom_receptors::NVARS; ++j)
            TRY(neuron_f32(net, &n.rx_custom[j], (std::string("receptors$") + custom_receptors::NAMES[j]).c_str(),
                           custom_receptors::DEFAULTS[j]));
    }

    // lattice slot per neuron + plasticity tables
    TRY(dev_alloc_t(net, &net->lattice_slot, np));
    TRY(fill_u32(net, net->lattice_slot, np, 0));
    for (const auto &l : net->lattices) TRY(fill_u32(net, net->lattice_slot + l.first, l.count, l.slot));
    const size_t nl = std::max<size_t>(1, net->lattices.size());
    {
        hvec<uint32_t> lf(nl, 0), lc(nl, 0);
        for (const auto &l : net->lattices) { lf[l.slot] = l.first; lc[l.slot] = l.count; }
        TRY(dev_alloc_t(net, &net->lat_first_dev, nl));
        TRY(dev_alloc_t(net, &net->lat_count_dev, nl));
        HIP_TRY(copy_sync(net, net->lat_first_dev, lf.data(), nl * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(copy_sync(net, net->lat_count_dev, lc.data(), nl * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        TRY(dev_alloc_t(net, &net->spike_counts, np));
        TRY(fill_u32(net, net->spike_counts, np, 0));
    }
    net->want_whist.assign(nl, 0);
    net->whist.assign(nl, nullptr);
    net->stdp_host.assign(nl * PL_STRIDE, 0.0f);
    net->plast_host.assign(nl, 0);
    for (size_t l = 0; l < nl; ++l) {   // plasticity/mod.rs:29-39
        float *s = &net->stdp_host[l * PL_STRIDE];
        s[0] = 2.0f; s[1] = 2.0f; s[2] = 4.5f; s[3] = 4.5f; s[4] = 0.1f;
        s[5] = 0.0f; s[6] = 0.1f; s[7] = 0.1f;        // STDP; BCM defaults decay 0.1, average_scalar 0.1 (plasticity/mod.rs:91-95)
    }
    TRY(dev_alloc_t(net, &net->stdp_dev, nl * PL_STRIDE));
    TRY(dev_alloc_t(net, &net->plast_dev, nl));
    HIP_TRY(hipMemcpyAsync(net->stdp_dev, net->stdp_host.data(), nl * PL_STRIDE * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemcpyAsync(net->plast_dev, net->plast_host.data(), nl * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    // RewardModulatedSTDP defaults, plasticity/mod.rs:176-189
    net->rm_host.assign(nl * RM_STRIDE, 0.0f);
    net->rm_on_host.assign(nl, 0);
    for (size_t l = 0; l < nl; ++l) {
        float *m = &net->rm_host[l * RM_STRIDE];
        m[0] = 0.0f; m[1] = 20.0f; m[2] = 0.0001f; m[3] = 2.0f; m[4] = 2.0f; m[5] = 4.5f; m[6] = 4.5f; m[7] = 0.1f;
        m[RM_DOPAMINE_BEFORE] = 0.0f;
    }
    TRY(dev_alloc_t(net, &net->rm_dev, nl * RM_STRIDE));
    TRY(dev_alloc_t(net, &net->rm_on_dev, nl));
    HIP_TRY(hipMemcpyAsync(net->rm_dev, net->rm_host.data(), nl * RM_STRIDE * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemcpyAsync(net->rm_on_dev, net->rm_on_host.data(), nl * 4, hipMemcpyHostToDevice, net->stream),
            SNN_ERR_BUFFER_WRITE);
    TRY(dev_alloc_t(net, &net->spike_list, np));
    TRY(dev_alloc_t(net, &net->spike_count, 1));
    HIP_TRY(hipMemsetAsync(net->spike_count, 0, 4, net->stream), SNN_ERR_BUFFER_WRITE);
    if (!net->csr) {
        net->dcol_stride = round_up(std::max<uint32_t>(net->n_tot, 1), 256);
        TRY(dev_alloc_t(net, &net->stdp_flag, np));
        TRY(fill_u32(net, net->stdp_flag, np, 0));
        TRY(dev_alloc_t(net, &net->stdp_dcol, (size_t)STDP_MAX_LATTICES * net->dcol_stride));
        TRY(dev_alloc_t(net, &net->stdp_drow, net->ld));
        TRY(dev_alloc_t(net, &net->stdp_rowbits, (size_t)net->n_chunks * 8));
    }

    // spike-train cells (spike_train/mod.rs:299-313, 998-1013, 50-56)
    c.c_pad = cp;
    const bool st_custom = net->st_kind == SNN_ST_CUSTOM;          // generated spike train: the description's defaults
    TRY(cell_f32(net, &c.current_voltage, "current_voltage", st_custom ? custom_st::DEFAULT_VOLTAGE : 0.0f));
    TRY(cell_f32(net, &c.v_th, "v_th", st_custom ? custom_st::DEFAULT_V_TH : 30.0f));
    TRY(cell_f32(net, &c.v_resting, "v_resting", st_custom ? custom_st::DEFAULT_V_RESTING : 0.0f));
    TRY(cell_f32(net, &c.dt, "dt", st_custom ? custom_st::DEFAULT_DT : 0.1f));
    for (int k = 0; k < CUSTOM_ST_MAX_VARS; ++k) c.custom[k] = nullptr;
    if (st_custom)
        for (int k = 0; k < custom_st::NVARS; ++k) TRY(cell_f32(net, &c.custom[k], custom_st::NAMES[k], custom_st::DEFAULTS[k]));
    // `decay` of a generated refractoriness lives in the k plane (set_decay / get_decay, nb_macro lib.rs:5738-5744)
    TRY(cell_f32(net, &c.k, "neural_refractoriness$k", SNN_HAVE_CUSTOM_REFRACTORINESS ? custom_refr::DEFAULT_DECAY : 10000.0f));
    for (int k = 0; k < CUSTOM_REFR_MAX_VARS; ++k) c.refr_custom[k] = nullptr;
    if (SNN_HAVE_CUSTOM_REFRACTORINESS) {
        reg(CA, "neural_refractoriness$decay", T_F32, S_PLAIN, c.k, 0, 0);
        for (int k = 0; k < custom_refr::NVARS; ++k) {
            const std::string name = std::string("neural_refractoriness$") + custom_refr::NAMES[k];
            TRY(cell_f32(net, &c.refr_custom[k], nullptr, custom_refr::DEFAULTS[k]));
            reg(CA, name.c_str(), T_F32, S_PLAIN, c.refr_custom[k], 0, 0);
        }
    }
    // which NeuralRefractoriness (a type parameter in the reference): 0 DeltaDirac spike_train/mod.rs:79-88,
    // 1 ExponentialDecay :164-178
    TRY(dev_alloc_t(net, &c.refractoriness, cp));
    TRY(fill_u32(net, c.refractoriness, cp, 0));
    reg(CA, "neural_refractoriness$kind", T_U32, S_PLAIN, c.refractoriness, 0, 0);
    const bool st_bcm = net->st_kind == SNN_ST_BCM_POISSON, st_poisson = net->st_kind == SNN_ST_POISSON || st_bcm;
    TRY(cell_f32(net, &c.chance_of_firing, st_poisson ? "chance_of_firing" : nullptr, 0.0f));
    // BCMPoissonNeuron's activity bookkeeping, spike_train/mod.rs:846-857, defaults :876-881
    TRY(cell_f32(net, &c.bcm_avg, st_bcm ? "average_activity" : nullptr, 0.0f));
    TRY(cell_f32(net, &c.bcm_cur, st_bcm ? "current_activity" : nullptr, 0.0f));
    TRY(cell_f32(net, &c.bcm_clock, st_bcm ? "firing_rate_clock" : nullptr, 0.0f));
    TRY(cell_f32(net, &c.bcm_window, st_bcm ? "firing_rate_window" : nullptr, 500.0f));
    TRY(dev_alloc_t(net, &c.bcm_period, cp));
    TRY(fill_u32(net, c.bcm_period, cp, 3));
    TRY(dev_alloc_t(net, &c.bcm_num_spikes, cp));
    TRY(fill_u32(net, c.bcm_num_spikes, cp, 0));
    if (st_bcm) {
        reg(CA, "period", T_U32, S_PLAIN, c.bcm_period, 0, 0);
        reg(CA, "num_spikes", T_U32, S_PLAIN, c.bcm_num_spikes, 0, 0);
    }
    TRY(cell_f32(net, &c.rate, net->st_kind == SNN_ST_RATE ? "rate" : nullptr, 0.0f));
    TRY(cell_f32(net, &c.step, net->st_kind == SNN_ST_RATE ? "step" : (net->st_kind == SNN_ST_PRESET ? "internal_clock" : nullptr), 0.0f));
    TRY(dev_alloc_t(net, &c.counter, cp));
    TRY(fill_u32(net, c.counter, cp, 0));
    if (net->st_kind == SNN_ST_PRESET) reg(CA, "counter", T_U32, S_PLAIN, c.counter, 0, 0);
    {
        // no firing times until snn_set_firing_times: every cell's list is empty
        uint32_t *ptr = nullptr;
        TRY(dev_alloc_t(net, &ptr, (size_t)cp + 1));
        TRY(fill_u32(net, ptr, (size_t)cp + 1, 0));
        c.preset_ptr = ptr;
        c.preset_times = nullptr;
        net->preset_host.assign(net->nc, {});
    }
    TRY(cell_f32(net, &c.presyn_value, nullptr, 0.0f));
    TRY(dev_alloc_t(net, &c.seed, cp));
    if (cp) {
        hipLaunchKernelGGL(k_iota_u32, dim3((cp + 255) / 256), dim3(256), 0, net->stream, c.seed, (size_t)cp, 1u);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    if (st_poisson) reg(CA, "seed", T_U32, S_PLAIN, c.seed, 0, 0);
    TRY(dev_alloc_t(net, &c.is_spiking, cp));
    TRY(fill_u32(net, c.is_spiking, cp, 0));
    reg(CA, "is_spiking", T_U32, S_PLAIN, c.is_spiking, 0, 0);
    TRY(dev_alloc_t(net, &c.last_firing_time, cp));
    HIP_TRY(hipMemsetAsync(c.last_firing_time, 0xFF, (size_t)std::max<uint32_t>(cp, 1) * 4, net->stream),
            SNN_ERR_BUFFER_WRITE);
    reg(CA, "last_firing_time", T_I32, S_PLAIN, c.last_firing_time, 0, 0);
    TRY(typed_f32(net, &c.nt_t, cp, 0.0f, 0.0f, 0.0f));
    TRY(typed_f32(net, &c.nt_t_max, cp, 1.0f, 1.0f, 1.0f));
    TRY(typed_f32(net, &c.nt_clearance, cp, nt_c, nt_c, nt_c));
    TRY(typed_f32(net, &c.nt_v_p, cp, 2.0f, 2.0f, 2.0f));
    TRY(typed_f32(net, &c.nt_k_p, cp, 5.0f, 5.0f, 5.0f));
    TRY(dev_alloc_t(net, &c.nt_flags, (size_t)K_TYPES * cp));
    TRY(fill_u32(net, c.nt_flags, (size_t)K_TYPES * cp, 0));
    reg(CA, "neurotransmitters$t", T_F32, S_PLAIN_K, c.nt_t, 0, cp);
    reg(CA, "neurotransmitters$t_max", T_F32, S_PLAIN_K, c.nt_t_max, 0, cp);
    reg(CA, "neurotransmitters$clearance_constant", T_F32, S_PLAIN_K, c.nt_clearance, 0, cp);
    reg(CA, "neurotransmitters$decay_constant", T_F32, S_PLAIN_K, c.nt_clearance, 0, cp);
    reg(CA, "neurotransmitters$v_p", T_F32, S_PLAIN_K, c.nt_v_p, 0, cp);
    reg(CA, "neurotransmitters$k_p", T_F32, S_PLAIN_K, c.nt_k_p, 0, cp);
    reg(CA, "neurotransmitters$flags", T_U32, S_PLAIN_K, c.nt_flags, 0, cp, 1);
    for (int j = 0; j < CUSTOM_KINETICS_MAX_VARS; ++j) c.nt_custom[j] = nullptr;
    if (net->nt_kind == SNN_NT_CUSTOM)
        for (int j = 0; j < custom_nt::NVARS; ++j) {
            const float d = custom_nt::DEFAULTS[j];
            TRY(typed_f32(net, &c.nt_custom[j], cp, d, d, d));
            reg(CA, (std::string("neurotransmitters$") + custom_nt::NAMES[j]).c_str(), T_F32, S_PLAIN_K, c.nt_custom[j], 0, cp);
        }
    TRY(dev_alloc_t(net, &c.lattice_slot, cp));
    TRY(fill_u32(net, c.lattice_slot, cp, 0));
    for (const auto &l : net->st_lattices)
        TRY(fill_u32(net, c.lattice_slot + (l.first - net->nn), l.count, l.slot));
    net->st_clock.assign(std::max<size_t>(1, net->st_lattices.size()), 0);
    TRY(dev_alloc_t(net, &net->st_clock_dev, net->st_clock.size()));
    HIP_TRY(host_malloc(reinterpret_cast<void **>(&net->st_clock_pinned), net->st_clock.size() * sizeof(long long), hipHostMallocDefault),
            SNN_ERR_BUFFER_CREATE);

    // graph + partials + counts
    if (net->csr) net->n_chunks = 1;     // the CSR kernel writes the finished two-level sum
    TRY(dev_alloc_t(net, &net->W, net->csr ? 0 : wcount(net->n_tot, net->ld)));
    TRY(dev_alloc_t(net, &net->part_i, (size_t)net->n_chunks * net->ld));
    TRY(dev_alloc_t(net, &net->part_t, (size_t)K_TYPES * net->n_chunks * net->ld));
    TRY(dev_alloc_t(net, &net->n_in, net->ld));
    TRY(dev_alloc_t(net, &net->tcount, (size_t)K_TYPES * net->ld));
    net->counts_dirty = true;
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}

int end_run(snn_network *net, bool keep_stdp = false);

// ---- attribute transfer ------------------------------------------------------------------------

// copy `count` 32-bit words between host and plane `plane` for global indices [first, first+count)
int xplane_copy(snn_network *net, int plane, uint32_t first, uint32_t count, void *host, bool to_device)
{
    float *dev = net->xbuf + net->xl.at(first, plane);
    if (to_device) HIP_TRY(copy_sync(net, dev, host, (size_t)count * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    else HIP_TRY(copy_sync(net, host, dev, (size_t)count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}

int attr_io(snn_network *net, uint32_t id, const char *name, AttrType type, void *host, size_t count, bool set)
{
    if (!net || !name || (!host && count)) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l) return fail(SNN_ERR_BAD_ARG, "unknown lattice id " + std::to_string(id));
    auto &table = l->spike_train ? net->cell_attrs : net->neuron_attrs;
    auto it = table.find(name);
    if (it == table.end()) return fail(SNN_ERR_BAD_ATTR, std::string("unknown attribute '") + name + "'");
    const Attr &a = it->second;
    if (a.type != type) return fail(SNN_ERR_BAD_ATTR, std::string("attribute '") + name + "' has another scalar type");
    if (set && l->spike_train && std::strcmp(name, "neural_refractoriness$kind") == 0) {
        const uint32_t top = SNN_HAVE_CUSTOM_REFRACTORINESS ? CUSTOM_REFRACTORINESS : 1u;
        const uint32_t *h = static_cast<const uint32_t *>(host);
        for (size_t i = 0; i < count; ++i)
            if (h[i] > top)
                return fail(SNN_ERR_BAD_ARG, "neural_refractoriness$kind: 0 DeltaDirac, 1 ExponentialDecay" +
                                                 std::string(top == 2 ? ", 2 the generated one" : "") + "; got " + std::to_string(h[i]));
    }
    const bool typed = (a.store == S_PLAIN_K || a.store == S_XPLANE_K);
    const size_t expect = (size_t)l->count * (typed ? K_TYPES : 1);
    if (count != expect)
        return fail(SNN_ERR_DIM_MISMATCH, std::string("attribute '") + name + "': expected " +
                                              std::to_string(expect) + " values, got " + std::to_string(count));
    if (l->count == 0) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    if (set && l->spike_train) net->view_dirty = true;
    if (set) net->shadow_valid = false;
    if (set) net->uni_dirty = true;
    const uint32_t first = l->spike_train ? l->first - net->nn : l->first;   // index inside its own arrays

    if (!typed) {
        if (a.store == S_PLAIN) {
            char *dev = static_cast<char *>(a.base) + (size_t)first * 4;
            if (set) HIP_TRY(copy_sync(net, dev, host, count * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
            else HIP_TRY(copy_sync(net, host, dev, count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
        } else {
            TRY(xplane_copy(net, a.plane, first, l->count, host, set));
        }
    } else {
        // host layout [cell*3 + k] (gpu_lattices/mod.rs:117-127) <-> device type-major planes
        hvec<uint32_t> tmp(l->count);
        uint32_t *h = static_cast<uint32_t *>(host);
        for (int k = 0; k < K_TYPES; ++k) {
            if (set) for (uint32_t i = 0; i < l->count; ++i) tmp[i] = h[(size_t)i * K_TYPES + k];
            if (a.store == S_PLAIN_K) {
                char *dev = static_cast<char *>(a.base) + ((size_t)k * a.pad + first) * 4;
                if (set) HIP_TRY(copy_sync(net, dev, tmp.data(), (size_t)l->count * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
                else HIP_TRY(copy_sync(net, tmp.data(), dev, (size_t)l->count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
            } else {
                TRY(xplane_copy(net, a.plane + k, first, l->count, tmp.data(), set));
            }
            if (!set) for (uint32_t i = 0; i < l->count; ++i) h[(size_t)i * K_TYPES + k] = tmp[i];
        }
    }
    if (set && a.dirties) net->counts_dirty = true;
    if (set && a.dirties && type == T_U32 && typed) {
        // "neurotransmitters$flags": remember which lattices release anything at all, so that networks without
        // neurotransmitters do not read three flag planes per neuron / cell and step
        const uint32_t *h = static_cast<const uint32_t *>(host);
        bool any = false;
        for (size_t i = 0; i < count && !any; ++i) any = h[i] != 0;
        net->lattice_has_nt[id] = any;
        uint32_t mask = 0;
        for (size_t i = 0; i < count; ++i) mask |= h[i] ? (1u << (i % K_TYPES)) : 0u;
        net->lattice_nt_mask[id] = mask;
        net->x_dirty = true;                       // the planes on the wire follow the transmitter types in use
        net->any_nt_neurons = net->any_nt_cells = false;
        for (const auto &kv : net->lattice_has_nt) {
            const LatticeInfo *li = find_lattice(net, kv.first);
            if (kv.second && li) (li->spike_train ? net->any_nt_cells : net->any_nt_neurons) = true;
        }
    }
    return SNN_OK;
}

// ---- per-step launches -------------------------------------------------------------------------

// The host-built half of the step image (snn_kernels_csr.hpp, "STEP IMAGE").  Per slice: the sorted set of everything its 64 rows
// gather, cut greedily into pieces -- a piece starts at the first source not yet covered and spans at most 64 LDS words of
// consecutive sources of one kind (a neuron is one word, a spike-train cell the two words of its view entry), trimmed to the last
// source it holds.  A slice that needs more than IMG_MAX_PIECES pieces stays unstaged (no pieces, plain codes).
// halo_word (shard handles, the image of direct runs): per neuron the word of the receive buffer that carries it (0xFFFFFFFF: read
// from the exchanged state); such a source has code halo_base + word, a third kind of piece next to neurons and cells.
void build_step_image_plan(const snn_network *net, const hvec<uint32_t> &slice_ptr, const hvec<uint32_t> &sell_pre, uint32_t n_slices,
                           hvec<uint32_t> &hdr, hvec<uint32_t> &plan_win, uint64_t &records, uint64_t &staged_slices,
                           const uint32_t *halo_word = nullptr)
{
    const uint32_t halo_base = net->nn + net->nc;
    auto code_of = [&](uint32_t p) { return (halo_word && p < net->nn && halo_word[p] != 0xFFFFFFFFu) ? halo_base + halo_word[p] : p; };
    auto kind_of = [&](uint32_t code) { return code < net->nn ? 0 : (code < halo_base ? 1 : 2); };
    hdr.assign((size_t)n_slices * IMG_HDR_WORDS, 0u);
    plan_win.assign(sell_pre.size(), PLAN_CODE);
    records = 0; staged_slices = 0;
    hvec<uint32_t> codes, offs;
    for (uint32_t sl = 0; sl < n_slices; ++sl) {
        const uint32_t s0 = slice_ptr[sl], s1 = slice_ptr[sl + 1], width = (s1 - s0) >> 6;
        uint32_t *h = &hdr[(size_t)sl * IMG_HDR_WORDS];
        h[0] = (uint32_t)records; h[1] = (width + 1) / 2;
        records += (uint64_t)h[1] * 64;
        if (width == 0) continue;
        codes.clear();
        for (uint32_t e = s0; e < s1; ++e)
            if (sell_pre[e] != SELL_PAD) codes.push_back(code_of(sell_pre[e]));
        std::sort(codes.begin(), codes.end());
        codes.erase(std::unique(codes.begin(), codes.end()), codes.end());
        offs.assign(codes.size(), 0u);
        uint32_t n_pieces = 0;
        bool staged = !codes.empty();
        for (size_t i = 0; i < codes.size() && staged;) {
            const uint32_t start = codes[i];
            const int kind = kind_of(start);
            const uint32_t per = kind == 1 ? 2u : 1u, span = IMG_PIECE_WORDS / per;
            if (n_pieces == IMG_MAX_PIECES) { staged = false; break; }
            size_t j = i;
            while (j < codes.size() && codes[j] - start < span && kind_of(codes[j]) == kind) {
                offs[j] = n_pieces * IMG_PIECE_WORDS + (codes[j] - start) * per;
                ++j;
            }
            h[4 + 2 * n_pieces] = start;
            h[5 + 2 * n_pieces] = (codes[j - 1] - start + 1) * per;
            ++n_pieces;
            i = j;
        }
        if (!staged) {
            n_pieces = 0;
            for (uint32_t k = 0; k < 2 * IMG_MAX_PIECES; ++k) h[4 + k] = 0u;
        }
        h[2] = n_pieces;
        staged_slices += staged ? 1 : 0;
        for (uint32_t lane = 0; lane < 64; ++lane) {
            uint32_t prev = 0;
            for (uint32_t k = 0; k < width; ++k) {
                const size_t e = s0 + lane + (size_t)k * 64;
                const uint32_t p = sell_pre[e];
                if (p == SELL_PAD) break;                   // padding only ever trails a row
                const uint32_t chunk_bit = (k == 0 || p / CHUNK != prev / CHUNK) ? 0x80000000u : 0u;
                prev = p;
                const uint32_t code = code_of(p);
                if (staged) {
                    const size_t at = std::lower_bound(codes.begin(), codes.end(), code) - codes.begin();
                    plan_win[e] = offs[at] | (kind_of(code) == 1 ? IMG_CELL_BIT : 0u) | chunk_bit;
                } else {
                    plan_win[e] = code | chunk_bit;
                }
            }
        }
    }
}

SellGraph csr_graph(const snn_network *net)
{
    SellGraph g{};
    g.slice_ptr = net->csr_ptr; g.pre = net->csr_pre; g.w = net->csr_w; g.row_len = net->csr_row_len;
    g.edge_slot = net->csr_edge_slot; g.edge_post = net->csr_post;
    g.t_ptr = net->csr_t_ptr; g.t_edge = net->csr_t_edge;
    g.n_loc = net->n_loc; g.n_slices = (net->n_loc + 63) / 64;
    g.plan = net->csr_plan; g.halo = nullptr; g.halo_base = PLAN_CODE;      // (the received segments are not read directly)
    return g;
}

// Which read-only parameters hold one value for the whole population (UniformTable): rescanned after attribute writes
int ensure_uniform_tables(snn_network *net)
{
    if (!net->uni_dirty) return SNN_OK;
    net->uni_dirty = false;
    HIP_TRY(hipMemsetAsync(net->uni_neuron, 0, sizeof(UniformTable), net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemsetAsync(net->uni_cell, 0, sizeof(UniformTable), net->stream), SNN_ERR_BUFFER_WRITE);
    if (!net->uniform_params) return SNN_OK;
    auto scan = [&](UniformTable *t, int slot, const void *arr, uint32_t n) -> int {
        if (n == 0 || !arr) return SNN_OK;
        const uint32_t *p = static_cast<const uint32_t *>(arr);
        hipLaunchKernelGGL(k_uniform_begin, dim3(1), dim3(1), 0, net->stream, t, slot, p, n);
        hipLaunchKernelGGL(k_uniform_scan, dim3(std::min<uint32_t>((n + 255) / 256, 1024)), dim3(256), 0, net->stream, t, slot, p, n);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    };
    const NeuronArrays &n = net->na;
    const CellArrays &c = net->ca;
    const struct { int slot; const void *arr; } np[] = {
        {NP_GAP, n.gap_conductance}, {NP_DT, n.dt}, {NP_C_M, n.c_m}, {NP_V_TH, n.v_th}, {NP_A, n.a}, {NP_B, n.b},
        {NP_C, n.c}, {NP_D, n.d}, {NP_TAU_M, n.tau_m}};
    for (const auto &e : np) TRY(scan(net->uni_neuron, e.slot, e.arr, net->nn));
    const struct { int slot; const void *arr; } cp[] = {
        {CP_V_TH, c.v_th}, {CP_V_RESTING, c.v_resting}, {CP_DT, c.dt}, {CP_K, c.k}, {CP_CHANCE, c.chance_of_firing},
        {CP_REFR, c.refractoriness}};
    for (const auto &e : cp) TRY(scan(net->uni_cell, e.slot, e.arr, net->nc));
    return SNN_OK;
}

int ensure_counts(snn_network *net)
{
    if (!net->counts_dirty || net->n_loc == 0) { net->counts_dirty = false; return SNN_OK; }
    {
        // live transmitter types: slots of the specialised input pass; planes of dead types must read as zero
        uint32_t mask = 0;
        for (const auto &kv : net->lattice_nt_mask) mask |= kv.second;
        if (mask != net->live_mask_applied) {
            net->n_live = 0;
            for (uint32_t k = 0; k < K_TYPES; ++k)
                if (mask >> k & 1u) net->live_type[net->n_live++] = k;
            for (uint32_t s = net->n_live, k = 0; s < K_TYPES; ++k)          // unused slots: the remaining types
                if (!(mask >> k & 1u)) net->live_type[s++] = k;
            if (net->n_live == 0) net->n_live = 1;                            // nothing released: slot 0 sums zeros
            HIP_TRY(hipMemsetAsync(net->part_t, 0, (size_t)K_TYPES * net->n_chunks * net->ld * 4, net->stream),
                    SNN_ERR_BUFFER_WRITE);
            net->live_mask_applied = mask;
        }
    }
    HIP_TRY(hipMemsetAsync(net->n_in, 0, (size_t)net->ld * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemsetAsync(net->tcount, 0, (size_t)K_TYPES * net->ld * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    if (net->csr) {
        if (net->csr_ptr) {
            CsrCountArgs a{};
            a.g = csr_graph(net);
            a.n_neurons = net->nn; a.ld = net->ld;
            a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
            a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
            a.n_in = net->n_in; a.tcount = net->tcount;
            hipLaunchKernelGGL(k_csr_count, dim3((net->n_loc + 255) / 256), dim3(256), 0, net->stream, a);
            HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        }
    } else if (net->n_tot) {
        CountArgs a{};
        a.W = net->W; a.ld = net->ld; a.n_loc = net->n_loc; a.n_neurons = net->nn; a.n_tot = net->n_tot;
        a.nt_flags = net->na.nt_flags; a.n_pad = net->n_pad;
        a.st_nt_flags = net->ca.nt_flags; a.c_pad = net->c_pad;
        a.n_in = net->n_in; a.tcount = net->tcount;
        a.rows_per_block = 256;
        dim3 grid((net->n_loc + 255) / 256, (net->n_tot + 255) / 256);
        hipLaunchKernelGGL(k_graph_count, grid, dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    net->counts_dirty = false;
    return SNN_OK;
}

} // namespace
