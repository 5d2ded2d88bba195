// Sparse form of the hot path, for connectivity that cannot be held as a dense N x N matrix (BASELINE
// configs[4]: 4 x 512^2 neurons + Poisson cells would need 4.4 TB dense).  The reference has no sparse GPU form
// (its AdjacencyList, graph/mod.rs:974-1118, is CPU only).  The C ABI takes CSR by postsynaptic neuron
// (snn_set_graph_csr); on the device the rows live in SELL-64 ("sliced ELLPACK", slice = the 64 rows of one
// wavefront): entry k of row r sits at slice_ptr[r / 64] + k * 64 + (r % 64), rows of a slice are padded to the
// slice's longest row with a sentinel index.  One thread per row then reads its k-th synapse with a fully
// coalesced load -- no staging through LDS, no barrier -- and a batch of EDGE_BATCH consecutive k has all its
// loads (index, weight, then the gathered presynaptic state) in flight together.
//
// Arithmetic = the dense kernel's canonical chunked ascending sum: a row's entries are sorted by presynaptic
// index and a partial is flushed whenever the 256-index chunk changes, so dense and sparse handles of the same
// graph produce bit-identical results.  8 B per stored synapse are streamed per step (index + weight); the
// presynaptic state is gathered from the exchanged planes, which stay L2-resident (20 B per neuron).
//
// For STDP the handle also keeps, in the caller's CSR edge order: the SELL slot and the local row of every
// edge, and the transpose index t_ptr / t_edge (edges grouped by presynaptic cell).
#pragma once
#include "snn_kernels_exchange.hpp"
#include "snn_kernels_inputs.hpp"
#include "snn_kernels_misc.hpp"
#include "snn_kernels_reward.hpp"
#include "snn_kernels_update.hpp"

namespace snn {

#ifndef SNN_CSR_PREFETCH
#define SNN_CSR_PREFETCH 0
#endif
constexpr bool CSR_PREFETCH = SNN_CSR_PREFETCH != 0;     // (A/B switch of the build: profiles/r05/README.md)
constexpr uint32_t SELL_PAD = 0xFFFFFFFFu;     // presynaptic index of a padding entry
constexpr uint32_t PLAN_CODE = 0x7FFFFFFFu;    // gather-plan word: the source code (all ones = padding), bit 31 = new chunk

// peer form: where a poll that gives up says so (see peer_give_up)
struct PeerFailure { uint32_t *word[2]; };

struct SellGraph {
    const uint32_t *slice_ptr;   // [n_slices + 1] element offsets (multiples of 64)
    const uint32_t *pre;         // [entries]
    // [entries] what the row sums read instead of `pre` (k_csr_plan): bit 31 = this entry opens a new 256-index chunk of its
    // row (the canonical flush), bits 0..30 = where the presynaptic value is gathered from: < n_neurons the exchanged
    // state of that neuron, < halo_base the spike-train cell (code - n_neurons), else word (code - halo_base) of `halo`,
    // the received segments themselves (shard handles in a library-driven run: no unpack before the rows); 0x7FFFFFFF pad
    const uint32_t *plan;
    const uint32_t *halo;
    uint32_t halo_base;
    // PEER form of a library-driven run (snn_network_exchange.hpp, "peer form"): the halo arrives as 8-byte granules
    // {value bits, tag | spike << 31} that the peers' border rows store straight into this handle's receive set; word
    // (code - halo_base) of `halo` is granule (code - halo_base) of `halo64`, read when its tag is the step's.
    const unsigned long long *halo64;
    uint32_t halo_tag, spin_limit;
    PeerFailure failed;          // set when a poll gave up
    // ... with chemical synapses a halo neuron has one granule per plane on the wire, adjacent: granule (code - halo_base) + slot;
    // halo_slot_v = slot of the voltage (0xFFFFFFFF: not on the wire), halo_slot_t[k] = slot of transmitter type k.  A
    // transmitter granule carries, in bit 30 of its tag word, whether the neuron releases that type (neurotransmitters$flags).
    uint32_t halo_slot_v, halo_slot_t[K_TYPES];
    uint32_t delay, delay_seed;  // option "halo_peer_delay": injected latency in front of the polls (peer_delay)
    float *w;                    // [entries]
    const uint32_t *row_len;     // [n_slices * 64]
    const uint32_t *edge_slot;   // [nnz] CSR edge -> SELL entry
    const uint32_t *edge_post;   // [nnz] CSR edge -> local row
    const uint32_t *t_ptr;       // [n_tot + 1]
    const uint32_t *t_edge;      // [nnz] CSR edge ids grouped by presynaptic cell
    uint32_t n_loc, n_slices;
};

struct CsrInputsArgs {
    SellGraph g;
    InputsArgs in;      // presynaptic state pointers / sizes; ld = stride of the partial rows
};

// One granule of the peer form: spins until its tag is `tag` (bounded: a peer that never arrives ends the run with an error
// instead of a hang).  System scope: the writer may be another device.
// `failed` = two words: [0] host-visible (read by the run call when the stream has drained), [1] on the device -- once set, every
// later poll of the run gives up at once (the launches already enqueued behind a failure must not each wait out the limit).
__device__ __forceinline__ void peer_give_up(uint32_t *const *failed)
{
    __hip_atomic_store(failed[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(failed[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a granule's second word: bits 0..29 the step tag, bit 30 "the neuron releases this transmitter type" (transmitter granules),
// bit 31 "the neuron spiked in the step that produced this value"
constexpr uint32_t PEER_TAG_MASK = 0x3FFFFFFFu, PEER_FLAG_BIT = 0x40000000u, PEER_SPIKE_BIT = 0x80000000u;
// Injected delay (option "halo_peer_delay", tests): up to `max_sleeps` s_sleep(127) (about 3 us each), pseudo-random per call site --
// a stand-in for the latencies of a real interconnect in front of stores, polls and done announcements
__device__ __forceinline__ void peer_delay(uint32_t max_sleeps, uint32_t seed, uint32_t salt)
{
    if (max_sleeps == 0u) return;
    const uint32_t n = hash32(seed, ((uint64_t)salt << 32) | (blockIdx.x * 4u + (threadIdx.x >> 6))) % (max_sleeps + 1u);
    for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
__device__ __forceinline__ unsigned long long peer_granule(const unsigned long long *g, uint32_t tag, uint32_t spin_limit, const PeerFailure &f)
{
    unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (((uint32_t)(x >> 32) & PEER_TAG_MASK) == tag) return x;
    if (__hip_atomic_load(f.word[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return x;     // the run has already failed
    uint32_t spins = 0;
    do {
        x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } while (((uint32_t)(x >> 32) & PEER_TAG_MASK) != tag && ++spins < spin_limit);
    if (((uint32_t)(x >> 32) & PEER_TAG_MASK) != tag) peer_give_up(f.word);
    return x;
}

// The canonical sums of one postsynaptic row (thread q of a wavefront = row q of a SELL slice).  Returns with
// `sum` / `tsum` holding the second-level sums; rows past n_loc compute on row 0 and are discarded by the caller.
// PEER: the halo arrives as granules (the peer form of a library-driven run) -- its own instantiation: the extra registers of the
// granule path cost the plain step two wavefronts of occupancy per SIMD (80 against 50 VGPRs, C5 41.8 against 38.7 us)
template <bool ELEC, bool CHEM, bool PEER = false>
__device__ __forceinline__ void csr_row_sums(const CsrInputsArgs &a, uint32_t q, float &sum, float (&tsum)[K_TYPES])
{
    // entries whose loads are in flight together (16 for the electrical form -- a row of BASELINE configs[4] in ONE batch --
    // measured slower, same box: C5 42.0 against 38.6 us per step, a G = 8 rank's step 18.9 against 18.6)
    constexpr uint32_t EDGE_BATCH = CHEM ? 4 : 8;
    const InputsArgs &in = a.in;
    const uint32_t slice = q >> 6;
    const uint32_t s0 = a.g.slice_ptr[slice];
    const uint32_t width = (a.g.slice_ptr[slice + 1] - s0) >> 6;   // wave-uniform
    const uint32_t base = s0 + (q & 63u);
    const bool row_valid = q < a.g.n_loc;
    const uint32_t qq = row_valid ? q : 0u;

    const uint32_t gq_index = in.rows.global_of(qq);           // a hole row (no edges) reads some neuron's state, unused
    const float vq = ELEC ? in.xbuf[in.xl.at(gq_index, PLANE_V)] : 0.0f;
    const float gq = ELEC ? uload(in.uni, NP_GAP, in.gap_conductance, gq_index) : 0.0f;

    if (PEER) peer_delay(a.g.delay, a.g.delay_seed, a.g.halo_tag * 4u + 3u);
    float part = 0.0f;
    float tpart[K_TYPES] = {0.0f, 0.0f, 0.0f};
    bool open = false;               // a chunk's partial is being accumulated
    sum = 0.0f;
#pragma unroll
    for (int kk = 0; kk < K_TYPES; ++kk) tsum[kk] = 0.0f;

    // (1) plan word + weight of EDGE_BATCH consecutive entries: coalesced, independent.  PREFETCH (the electrical step without
    // the peer form): the NEXT batch's words are requested before this batch's gathers -- the streamed loads of batch k + 1 are
    // in flight while the gathers of batch k come back from L2, one dependent round trip less per batch after the first
    constexpr bool PREFETCH = CSR_PREFETCH && ELEC && !CHEM && !PEER;
    auto load_batch = [&](uint32_t k0, uint32_t (&pp)[EDGE_BATCH], float (&ww)[EDGE_BATCH]) {
#pragma unroll
        for (uint32_t u = 0; u < EDGE_BATCH; ++u) {
            const uint32_t k = min(k0 + u, width - 1);         // clamped; the tail is dropped below
            pp[u] = a.g.plan[base + (size_t)k * 64];
            ww[u] = a.g.w[base + (size_t)k * 64];
            if (k0 + u >= width) pp[u] = SELL_PAD;
        }
    };
    uint32_t p_next[PREFETCH ? EDGE_BATCH : 1];
    float w_next[PREFETCH ? EDGE_BATCH : 1];
    if constexpr (PREFETCH) {
        if (width) load_batch(0, p_next, w_next);
    }
    for (uint32_t k0 = 0; k0 < width; k0 += EDGE_BATCH) {
        uint32_t p[EDGE_BATCH];
        float w[EDGE_BATCH], v[EDGE_BATCH], t[EDGE_BATCH][K_TYPES];
        uint32_t flags[EDGE_BATCH];       // bit 0 cell, bit 1 silent cell, bit 2 a granule of the peer form, bits 8.. transmitter types
        unsigned long long g64[PEER ? EDGE_BATCH : 1];
        unsigned long long t64[(PEER && CHEM) ? EDGE_BATCH : 1][K_TYPES];
        if constexpr (PREFETCH) {
#pragma unroll
            for (uint32_t u = 0; u < EDGE_BATCH; ++u) { p[u] = p_next[u]; w[u] = w_next[u]; }
            if (k0 + EDGE_BATCH < width) load_batch(k0 + EDGE_BATCH, p_next, w_next);
        } else {
            load_batch(k0, p, w);
        }
        // (2) the gathers those words address -- branch-free (the inapplicable sources are clamped to index 0)
#pragma unroll
        for (uint32_t u = 0; u < EDGE_BATCH; ++u) {
            const uint32_t code = p[u] & PLAN_CODE;
            const bool pad = code == PLAN_CODE;
            const bool is_halo = !pad && code >= a.g.halo_base;
            const bool is_cell = !pad && !is_halo && code >= in.n_neurons;
            const uint32_t pn = (pad || is_cell || is_halo) ? 0u : code;
            const uint32_t sc = is_cell ? code - in.n_neurons : 0u;
            // peer form (launch-uniform): the entry's granules are requested with the rest of the batch -- plain loads, no loop
            // here: a loop inside the gather would take the batch's loads out of flight for EVERY row -- and looked at below
            const bool granule = PEER && is_halo && a.g.halo64;
            const unsigned long long *gbase = granule ? a.g.halo64 + (code - a.g.halo_base) : a.g.halo64;
            flags[u] = (is_cell ? 1u : 0u) | (granule ? 4u : 0u);
            v[u] = 0.0f;
            if (ELEC) {
                const float *src = is_cell ? reinterpret_cast<const float *>(in.st_view + sc)
                                 : (is_halo && !granule) ? reinterpret_cast<const float *>(a.g.halo + (code - a.g.halo_base))
                                                         : in.xbuf + in.xl.at(pn, PLANE_V);
                v[u] = *src;
                if (PEER && a.g.halo64) {
                    g64[PEER ? u : 0] = 0ull;
                    if (granule) g64[PEER ? u : 0] = __hip_atomic_load(gbase + a.g.halo_slot_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                flags[u] |= (is_cell && in.st_view[sc].y) ? 2u : 0u;
            }
            if (CHEM) {
#pragma unroll
                for (int kk = 0; kk < K_TYPES; ++kk) {
                    const uint32_t *fsrc = is_cell ? in.st_nt_flags + (size_t)kk * in.c_pad + sc
                                                   : in.nt_flags + (size_t)kk * in.n_pad + pn;
                    const float *tsrc = is_cell ? in.st_nt_t + (size_t)kk * in.c_pad + sc
                                                : in.xbuf + in.xl.at(pn, PLANE_T0 + kk);
                    // (a granule entry takes flag and value from its granule below: the clamped neuron 0 is not it)
                    flags[u] |= (!granule && *fsrc != 0) ? (0x100u << kk) : 0u;
                    t[u][kk] = *tsrc;
                    if (PEER && CHEM && a.g.halo64) {
                        t64[(PEER && CHEM) ? u : 0][kk] = 0ull;
                        if (granule && a.g.halo_slot_t[kk] != 0xFFFFFFFFu)
                            t64[(PEER && CHEM) ? u : 0][kk] = __hip_atomic_load(gbase + a.g.halo_slot_t[kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
        if (PEER && a.g.halo64) {
            // peer form: a granule whose tag is not yet the step's (the neighbour's border rows are still on their way) is polled
#pragma unroll
            for (uint32_t u = 0; u < (PEER ? EDGE_BATCH : 1u); ++u) {
                if (!(flags[u] & 4u)) continue;
                const unsigned long long *gbase = a.g.halo64 + ((p[u] & PLAN_CODE) - a.g.halo_base);
                if (ELEC) {
                    if (((uint32_t)(g64[u] >> 32) & PEER_TAG_MASK) != a.g.halo_tag)
                        g64[u] = peer_granule(gbase + a.g.halo_slot_v, a.g.halo_tag, a.g.spin_limit, a.g.failed);
                    v[u] = __uint_as_float((uint32_t)g64[u]);
                }
                if (PEER && CHEM) {
#pragma unroll
                    for (int kk = 0; kk < K_TYPES; ++kk) {
                        if (a.g.halo_slot_t[kk] == 0xFFFFFFFFu) continue;          // nobody releases the type: no plane, no flag
                        unsigned long long x = t64[(PEER && CHEM) ? u : 0][kk];
                        if (((uint32_t)(x >> 32) & PEER_TAG_MASK) != a.g.halo_tag)
                            x = peer_granule(gbase + a.g.halo_slot_t[kk], a.g.halo_tag, a.g.spin_limit, a.g.failed);
                        t[u][kk] = __uint_as_float((uint32_t)x);
                        flags[u] |= ((uint32_t)(x >> 32) & PEER_FLAG_BIT) ? (0x100u << kk) : 0u;
                    }
                }
            }
        }
        // (3) the row's sum, strictly in ascending presynaptic order with the canonical chunk flush
#pragma unroll
        for (uint32_t u = 0; u < EDGE_BATCH; ++u) {
            if ((p[u] & PLAN_CODE) == PLAN_CODE) continue;       // padding only ever trails a row
            if (p[u] >> 31) {                                    // first entry of a 256-index chunk: flush the previous one
                if (open) {
                    sum += part;
#pragma unroll
                    for (int kk = 0; kk < K_TYPES; ++kk) tsum[kk] += tpart[kk];
                }
                part = 0.0f;
#pragma unroll
                for (int kk = 0; kk < K_TYPES; ++kk) tpart[kk] = 0.0f;
                open = true;
            }
            if (ELEC) {
                // gap_junction neuron/mod.rs:54-60; spike_train_gap_junction :119-137 (never fired: v_resting
                // without the conductance factor)
                const float term = (flags[u] & 1u) ? ((flags[u] & 2u) ? v[u] : gq * v[u]) : gq * (v[u] - vq);
                part += term * w[u];
            }
            if (CHEM) {
#pragma unroll
                for (int kk = 0; kk < K_TYPES; ++kk)
                    if (flags[u] & (0x100u << kk)) tpart[kk] += t[u][kk] * w[u];
            }
        }
    }
    if (open) {
        sum += part;
#pragma unroll
        for (int kk = 0; kk < K_TYPES; ++kk) tsum[kk] += tpart[kk];
    }
    // Chunks without edges contribute +0.0f partials in the dense form; x + 0.0f == x for every x this sum
    // can hold (it starts at +0.0f, so it is never -0.0f): skipping them is exact.
}

// ---- the STEP IMAGE of a sparse graph (round 6) -------------------------------------------------------------------------------
// What the one-launch step reads per row when the weights are static (no weight update of any kind) and only gap junctions are on
// -- BASELINE configs[4].  Derived from the SELL arrays, rebuilt when the graph or a weight changes (ensure_step_image):
//   * records: TWO consecutive entries of a row as one 16-byte record {plan word, weight bits, plan word, weight bits}, record j of a
//     slice's lane at rec[first + j * 64 + lane]: a wavefront's load is 1 KiB contiguous (the dense pass's shape) where the plain
//     form issues four 256-byte loads; the same 8 B per stored synapse.
//   * per slice a header of IMG_HDR_WORDS words at a fixed address (slice * IMG_HDR_WORDS): {first record, records per lane,
//     window pieces, 0} and up to IMG_MAX_PIECES pieces {source code of the first word, words <= 64}.  The pieces are the
//     run-length union of everything the slice's 64 rows gather (neighbouring rows of a lattice read neighbouring neurons:
//     configs[4] reads 12 + 1 + 1 sources per row, 5.3 distinct words per row).  The wavefront requests them with coalesced loads
//     TOGETHER with its records -- neither depends on the other -- and parks them in its 4 KiB of LDS; a staged slice's plan word
//     is the LDS word of its source (bits 0..15; bit 30: a spike-train cell, whose two view words sit side by side; bit 31: the
//     entry opens a new 256-index chunk, as in the plain plan), so the gather is a ds_read: one dependent memory round trip per
//     row instead of two per batch, 5.3 words from L2 per row instead of 14 scattered ones.  A slice whose union needs more
//     than IMG_MAX_PIECES pieces (an unstructured graph) has no pieces: its plan words are the plain codes and the rows gather
//     from global memory as before -- still through 16-byte records.
// The sums are the canonical ones (ascending, chunk flush): bit-identical to csr_row_sums and to the dense kernel.
constexpr uint32_t IMG_HDR_WORDS = 40, IMG_MAX_PIECES = 16, IMG_PIECE_WORDS = 64, IMG_CELL_BIT = 0x40000000u;
#ifndef SNN_IMG_RING
#define SNN_IMG_RING 4
#endif
constexpr uint32_t IMG_RING = SNN_IMG_RING;               // records (16 B each) a lane keeps in flight; even
constexpr uint32_t IMG_WIN_WORDS = IMG_MAX_PIECES * IMG_PIECE_WORDS;           // LDS words per wavefront
struct CsrImage {
    const uint32_t *hdr;         // [n_slices][IMG_HDR_WORDS]; null: the launch takes the plain form
    const uint4 *rec;
};

// the sums of one row over its records; STAGED: the gathers are reads of the wavefront's window
template <bool STAGED>
__device__ __forceinline__ void csr_img_sums(const InputsArgs &in, const uint4 *rec, uint32_t pairs, const uint32_t *win, const uint32_t *xv,
                                             const uint32_t *cv, const uint32_t *halo, uint32_t halo_base, float vq, float gq,
                                             uint4 (&r)[IMG_RING], float &sum)
{
    constexpr uint32_t RING = IMG_RING;
    const uint32_t last = pairs ? pairs - 1u : 0u;
    float part = 0.0f;
    bool open = false;
    sum = 0.0f;
    for (uint32_t j0 = 0; j0 < pairs; j0 += RING) {
#pragma unroll
        for (uint32_t u = 0; u < RING; u += 2) {
            // two records = four entries: their gathers together, then their adds in ascending order
            uint32_t word[4] = {r[u].x, r[u].z, r[u + 1].x, r[u + 1].z};
            const float w[4] = {__uint_as_float(r[u].y), __uint_as_float(r[u].w), __uint_as_float(r[u + 1].y), __uint_as_float(r[u + 1].w)};
            if (j0 + u >= pairs) word[0] = word[1] = PLAN_CODE;              // (a clamped load past the row: dropped)
            if (j0 + u + 1 >= pairs) word[2] = word[3] = PLAN_CODE;
            float v[4];
            uint32_t silent[4];
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                const bool is_pad = (word[e] & PLAN_CODE) == PLAN_CODE;
                if (STAGED) {
                    const uint32_t off = is_pad ? 0u : (word[e] & 0xFFFFu);
                    v[e] = __uint_as_float(win[off]);
                    silent[e] = win[off + 1u];                               // (read for every entry: branch-free; used for cells)
                } else {
                    // an unstaged slice: the plain codes, gathered from the exchanged state / the cells' view
                    const uint32_t code = word[e] & PLAN_CODE;
                    const bool is_halo = !is_pad && code >= halo_base;
                    const bool is_cell = !is_pad && !is_halo && code >= in.n_neurons;
                    const uint32_t *src = is_halo ? halo + (code - halo_base)
                                                  : is_cell ? cv + 2u * (size_t)(code - in.n_neurons) : xv + (is_pad ? 0u : code);
                    v[e] = __uint_as_float(src[0]);
                    silent[e] = is_cell ? src[1] : 0u;
                    if (!is_pad) word[e] = (word[e] & 0x80000000u) | (is_cell ? IMG_CELL_BIT : 0u);
                }
            }
            // these two register sets are free: the records RING further on (clamped: a row without them re-reads its last one)
            r[u] = rec[(size_t)min(j0 + RING + u, last) * 64];
            r[u + 1] = rec[(size_t)min(j0 + RING + u + 1, last) * 64];
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                if ((word[e] & PLAN_CODE) == PLAN_CODE) continue;            // padding only ever trails a row
                if (word[e] >> 31) {                                         // first entry of a 256-index chunk: flush the previous one
                    if (open) sum += part;
                    part = 0.0f;
                    open = true;
                }
                // gap_junction neuron/mod.rs:54-60; spike_train_gap_junction :119-137 (never fired: v_resting without the conductance)
                const float term = (word[e] & IMG_CELL_BIT) ? (silent[e] ? v[e] : gq * v[e]) : gq * (v[e] - vq);
                part += term * w[e];
            }
        }
    }
    if (open) sum += part;
}

__device__ __forceinline__ void csr_row_sums_img(const CsrInputsArgs &a, const CsrImage &im, uint32_t q, uint32_t *win, float &sum)
{
    // records in flight per lane: IMG_RING x 16 bytes (4 KiB per wavefront, 128 KiB per CU at full occupancy -- several times what
    // the memory side needs in flight); each register set is reloaded as soon as its record has been summed (a ring: the
    // stream never waits for the sums, and no gather ever waits for memory)
    constexpr uint32_t RING = IMG_RING;
    const InputsArgs &in = a.in;
    const uint32_t slice = __builtin_amdgcn_readfirstlane(q >> 6), lane = q & 63u;
    // (the header is read-only for the launch and its address wave-uniform: through the constant address space it arrives by
    // scalar loads, header and pieces in three instructions, instead of one vector load and one wait per word)
    typedef uint32_t words4 __attribute__((ext_vector_type(4)));
    typedef uint32_t words16 __attribute__((ext_vector_type(16)));
    const uint32_t *hdr_at = im.hdr + (size_t)slice * IMG_HDR_WORDS;
    const words4 head = *(const __attribute__((address_space(4))) words4 *)hdr_at;
    const words16 pc0 = *(const __attribute__((address_space(4))) words16 *)(hdr_at + 4), pc1 = *(const __attribute__((address_space(4))) words16 *)(hdr_at + 20);
    const uint32_t first = head.x, pairs = head.y, n_pieces = head.z;          // wave-uniform
    const uint32_t last = pairs ? pairs - 1u : 0u;       // (a slice without entries reads one record of slack and sums nothing)
    sum = 0.0f;
    // the records first: everything else the row needs queues behind them
    const uint4 *rec = im.rec + first + lane;
    uint4 r[RING];
#pragma unroll
    for (uint32_t u = 0; u < RING; ++u) r[u] = rec[(size_t)min(u, last) * 64];
    const bool row_valid = q < a.g.n_loc;
    const uint32_t qq = row_valid ? q : 0u;
    const uint32_t gq_index = in.rows.global_of(qq);
    const float vq = in.xbuf[in.xl.at(gq_index, PLANE_V)];
    const float gq = uload(in.uni, NP_GAP, in.gap_conductance, gq_index);
    // The window: every piece one LDS-DMA load (global_load_lds_dword) of at most 64 words -- no register holds a window word
    // (the rows' records own the register file) and there is no ds_write pass; the destination of a wave instruction is
    // base + lane * 4, which is what a piece is.  Pieces the slice does not have are 0 words long: no lane takes part.
    const uint32_t *xv = reinterpret_cast<const uint32_t *>(in.xbuf) + in.xl.at(0, PLANE_V);
    const uint32_t *cv = reinterpret_cast<const uint32_t *>(in.st_view);
#pragma unroll
    for (uint32_t u = 0; u < IMG_MAX_PIECES; ++u) {
        const uint32_t code = u < 8 ? pc0[2 * (u & 7u)] : pc1[2 * (u & 7u)], words = u < 8 ? pc0[2 * (u & 7u) + 1] : pc1[2 * (u & 7u) + 1];
        // a piece's sources are of one kind: neurons of the exchanged state, spike-train cells (two words each), or -- shard handles
        // in a direct run -- words of the received halo segments
        const uint32_t *src = code < in.n_neurons ? xv + code
                                                  : code < a.g.halo_base ? cv + 2u * (size_t)(code - in.n_neurons) : a.g.halo + (code - a.g.halo_base);
        if (lane < words)
            __builtin_amdgcn_global_load_lds(src + lane, (__attribute__((address_space(3))) uint32_t *)(win + u * IMG_PIECE_WORDS), 4, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00 | 0xC000);         // vmcnt(0): records and window have landed (expcnt / lgkmcnt left alone)
    __builtin_amdgcn_wave_barrier();                              // (one wavefront fills and reads its own window: no workgroup barrier)
    if (n_pieces) csr_img_sums<true>(in, rec, pairs, win, xv, cv, a.g.halo, a.g.halo_base, vq, gq, r, sum);
    else csr_img_sums<false>(in, rec, pairs, win, xv, cv, a.g.halo, a.g.halo_base, vq, gq, r, sum);
}

template <bool ELEC, bool CHEM>
__global__ __launch_bounds__(256) void k_inputs_csr(const CsrInputsArgs a)
{
    const InputsArgs &in = a.in;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if ((q >> 6) >= a.g.n_slices) return;                    // whole wavefront
    float sum, tsum[K_TYPES];
    csr_row_sums<ELEC, CHEM>(a, q, sum, tsum);
    if (q >= a.g.n_loc) return;
    if (ELEC) in.part_i[q] = sum;
    if (CHEM) {
#pragma unroll
        for (int kk = 0; kk < K_TYPES; ++kk) in.part_t[(size_t)kk * in.ld + q] = tsum[kk];
    }
}

// Jobs that ride behind a sparse step's rows (the last blocks of k_step_csr) or make up the step's closing launch
// (k_step_close), side by side:
//   blocks [0, cell_blocks)                 the spike-train cells this rank reads advance (k_spike_trains' body),
//   blocks [cell_blocks, +unpack_blocks)    received segments go into the mirror (and into the shadow the next step
//                                           reads), with the last_firing_time stamp of a neuron owned elsewhere,
//   the rest                                the spike bitmaps of outgoing segments are cleared for a later in-kernel pack.
struct StepCloseArgs {
    SpikeTrainArgs cells;
    WireArgs recv;                  // unpack side: segment tables of the incoming segments
    float *xbuf2;                   // second destination of the unpack (the shadow of the next step) or null
    uint32_t recv_total;            // neurons over all incoming segments
    uint32_t recv_segments;
    WireArgs send;                  // segment tables of the outgoing segments (bitmaps to clear)
    uint32_t send_segments;
    uint32_t send_bitmap_words;     // over all outgoing segments
    // peer form: the arrivals of the previous step as granules (null: words of recv.buf) and their tag
    const unsigned long long *recv64;
    uint32_t recv_tag, spin_limit;
    PeerFailure failed;
    uint32_t cell_blocks, unpack_blocks;
    __host__ __device__ uint32_t blocks() const { return cell_blocks + unpack_blocks + (send_bitmap_words + 255) / 256; }
};

template <bool PEER = true>
__device__ __forceinline__ void step_close_block(const StepCloseArgs &a, const uint32_t block)
{
    if (block < a.cell_blocks) {
        spike_train_cell(a.cells, block * 256 + threadIdx.x);
        return;
    }
    if (block < a.cell_blocks + a.unpack_blocks) {
        const uint32_t t = (block - a.cell_blocks) * 256 + threadIdx.x;
        if (t >= a.recv_total) return;
        // the segment of flat position t: list offsets are the running totals (at most n_shards - 1 segments)
        uint32_t seg = 0;
        while (seg + 1 < a.recv_segments && t >= (uint32_t)a.recv.seg_list_offset[seg + 1]) ++seg;
        const uint32_t i = t - (uint32_t)a.recv.seg_list_offset[seg];
        const uint32_t count = a.recv.seg_count[seg];
        const uint32_t g = a.recv.list[t];
        if (g >= a.recv.n_neurons) return;
        uint32_t *x = reinterpret_cast<uint32_t *>(a.recv.xbuf), *x2 = reinterpret_cast<uint32_t *>(a.xbuf2);
        if (PEER && a.recv64) {
            // one granule per neuron and plane on the wire, adjacent: value, and the spike flag in the tag word's top bit
            uint32_t spike = 0u;
            for (uint32_t s = 0; s < a.recv.planes; ++s) {
                const unsigned long long g64 = peer_granule(a.recv64 + a.recv.seg_offset[seg] + (size_t)i * a.recv.planes + s, a.recv_tag,
                                                            a.spin_limit, a.failed);
                const size_t at = a.recv.xl.at(g, (int)a.recv.plane_id[s]);
                x[at] = (uint32_t)g64;
                if (x2) x2[at] = (uint32_t)g64;
                spike = (uint32_t)(g64 >> 63);              // (every granule of the neuron carries it)
            }
            x[a.recv.xl.at(g, PLANE_SPIKE)] = spike;
            if (x2) x2[a.recv.xl.at(g, PLANE_SPIKE)] = spike;
            if (spike) a.recv.last_firing_time[g] = (int32_t)a.recv.clock;
            return;
        }
        const uint32_t *in = a.recv.buf + a.recv.seg_offset[seg];
        for (uint32_t s = 0; s < a.recv.planes; ++s) {
            const uint32_t v = in[(size_t)s * count + i];
            const size_t at = a.recv.xl.at(g, (int)a.recv.plane_id[s]);
            x[at] = v;
            if (x2) x2[at] = v;
        }
        const uint32_t spike = (in[(size_t)a.recv.planes * count + (i >> 5)] >> (i & 31u)) & 1u;
        x[a.recv.xl.at(g, PLANE_SPIKE)] = spike;
        if (x2) x2[a.recv.xl.at(g, PLANE_SPIKE)] = spike;
        if (spike) a.recv.last_firing_time[g] = (int32_t)a.recv.clock;
        return;
    }
    uint32_t t = (block - a.cell_blocks - a.unpack_blocks) * 256 + threadIdx.x;
    if (t >= a.send_bitmap_words) return;
    for (uint32_t seg = 0; seg < a.send_segments; ++seg) {
        const uint32_t count = a.send.seg_count[seg], words = (count + 31u) / 32u;
        if (t < words) { a.send.buf[a.send.seg_offset[seg] + (size_t)a.send.planes * count + t] = 0u; return; }
        t -= words;
    }
}

__global__ __launch_bounds__(256) void k_step_close(const StepCloseArgs a) { step_close_block(a, blockIdx.x); }

// peer form, start of a library-driven run: the receive set the first step reads, filled from the mirror (what earlier exchanges
// left there) as granules tagged for that step
__global__ __launch_bounds__(256) void k_peer_prefill(const WireArgs recv, uint32_t total, uint32_t segments, unsigned long long *set, uint32_t tag,
                                                      const uint32_t *nt_flags, uint32_t n_pad)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    uint32_t seg = 0;
    while (seg + 1 < segments && t >= (uint32_t)recv.seg_list_offset[seg + 1]) ++seg;
    const uint32_t i = t - (uint32_t)recv.seg_list_offset[seg];
    const uint32_t g = recv.list[t];
    if (g >= recv.n_neurons) return;
    const uint32_t *x = reinterpret_cast<const uint32_t *>(recv.xbuf);
    const uint32_t spike = x[recv.xl.at(g, PLANE_SPIKE)] ? PEER_SPIKE_BIT : 0u;
    for (uint32_t s = 0; s < recv.planes; ++s) {
        const uint32_t plane = recv.plane_id[s];
        // (neurotransmitters$flags are parameters every rank holds for every neuron)
        const uint32_t flag = (plane >= (uint32_t)PLANE_T0 && nt_flags[(size_t)(plane - PLANE_T0) * n_pad + g]) ? PEER_FLAG_BIT : 0u;
        set[recv.seg_offset[seg] + (size_t)i * recv.planes + s] = ((unsigned long long)(tag | spike | flag) << 32) | x[recv.xl.at(g, (int)plane)];
    }
}

// Inputs + neuron update of a sparse handle in ONE launch: row thread = neuron thread, so the sums never leave
// registers.  Other rows may still be gathering S(t) while a neuron writes S(t+1): as in k_step_resident the exchanged
// state is read from a shadow copy (in.xbuf = up.n.xbuf) and written to the exchange buffer and the other shadow
// (up.xout / up.xout2).
//
// Shard handles (halo exchange) run it twice per step: first over the BORDER slices -- the 64-row slices that hold a
// neuron some peer reads -- whose threads also write the neuron's wire values straight into the outgoing segments
// (PackTable: the step needs no pack launch), then, while RCCL moves those segments, over the INTERIOR slices.
// slice_list = the slices of the launch (null: all of them).
struct PackTable {
    const uint32_t *ptr;            // [n_loc + 1] entries of local row q: ptr[q] .. ptr[q + 1]; null = no packing
    const uint32_t *seg_off;        // per entry: word offset of its segment in `buf`,
    const uint32_t *seg_count;      //            neurons of that segment,
    const uint32_t *index;          //            and the neuron's position in it
    uint32_t *buf;                  // outgoing segments (wire format of snn_kernels_exchange.hpp); spike bitmaps zeroed
    uint32_t planes, plane_id[WIRE_MAX_PLANES];
    // peer form: per entry the granule of the PEER's receive set (this step's parity) that carries the neuron, and the peer;
    // a set may be overwritten once the peer has finished the step that read it: flags[peer] >= need_done (the peers' done
    // counters, stored into this handle's memory by their last workgroup)
    unsigned long long *const *dst; // [entries] or null
    const uint32_t *peer;           // [entries]
    const uint32_t *flags;          // [n_shards]
    uint32_t tag_out, need_done, spin_limit;
    PeerFailure failed;
    const uint32_t *nt_flags;       // [3][n_pad] which transmitter types a neuron releases (bit 30 of a transmitter granule's tag word)
    uint32_t n_pad;
    uint32_t delay, delay_seed;     // option "halo_peer_delay": injected latencies (peer_delay)
};

// peer form: a step's launch tells every neighbour which epoch this handle has COMPLETED (done_value: the one before the
// launch's own) by storing it into the neighbour's done counters
struct PeerSignal {
    uint32_t *const *signal;                  // [n_signal] addresses of the neighbours' flags[this shard]; null: not the peer form
    uint32_t n_signal, done_value;
    uint32_t delay, delay_seed;               // injected latency in front of the announcement (peer_delay)
};

struct CsrStepArgs {
    CsrInputsArgs c;
    UpdateArgs up;
    const uint32_t *slice_list;
    uint32_t n_listed;
    PackTable pack;
    // Jobs behind the rows (the last tail.blocks() blocks of the grid).  The spike-train cells advancing in the same launch:
    // rows read the cells' view of THIS step (InputsArgs::st_view), the cells write the next one (SpikeTrainArgs::
    // view_out) -- electrical-only handles without weight updates (nothing reads a cell's own arrays between the neuron
    // update and the cells' iteration).  Shard handles in a library-driven run whose rows gather the halo from the received
    // segments themselves (SellGraph::halo): the mirror copy + last_firing_time stamps of what arrived one step earlier,
    // and the clearing of the spike bitmaps of the other set of outgoing segments.
    StepCloseArgs tail;
    uint32_t xcd_bands;             // 1: row blocks are dealt to the XCDs in contiguous bands (see k_step_csr)
    PeerSignal peer;
    CsrImage img;                   // k_step_csr<..., IMG = true>: the step image the rows read
};
static_assert(sizeof(CsrStepArgs) <= 4096, "kernel arguments are limited to 4 KB");

template <int MODEL, bool ELEC, bool CHEM, bool PEER, bool IMG = false, bool PACK = true>
__device__ __forceinline__ void step_csr_block(const CsrStepArgs &a, uint32_t *win = nullptr)
{
    // the tail jobs come AFTER the row blocks: they fill the tail of the rows' streaming (cells at C5, one box, step time:
    // cells in their own launch 44.7 us; cell blocks first 41.5; spread evenly among the row blocks 45.6; last 39.3)
    const uint32_t row_blocks = gridDim.x - a.tail.blocks();
    if (blockIdx.x >= row_blocks) {
        step_close_block<PEER>(a.tail, blockIdx.x - row_blocks);
        return;
    }
    // Workgroups are handed to the 8 XCDs round-robin (a placement heuristic, used for speed only): XCD x takes a
    // contiguous band of the row blocks, so that the presynaptic state its rows gather (neighbouring rows of the same
    // lattice) is fetched into ONE L2 instead of all eight
    uint32_t row_block = blockIdx.x;
    if (a.xcd_bands) {
        const uint32_t x = blockIdx.x & 7u, per = row_blocks >> 3, extra = row_blocks & 7u;
        row_block = x * per + min(x, extra) + (blockIdx.x >> 3);
    }
    const uint32_t w = row_block * 4 + (threadIdx.x >> 6);  // wavefront = one SELL slice
    if (a.slice_list ? w >= a.n_listed : w >= a.c.g.n_slices) return;
    const uint32_t q = (a.slice_list ? a.slice_list[w] : w) * 64u + (threadIdx.x & 63u);
    // the row's entries of the pack table are requested BEFORE the row sums (they depend on nothing the sums produce): a
    // border launch of a few workgroups is a chain of dependent memory round trips, these two overlap with the sums'
    uint32_t pack_begin = 0, pack_end = 0, seg0_count = 0, seg0_index = 0, seg0_off = 0;
    if (PACK && a.pack.ptr && q < a.c.g.n_loc) {
        pack_begin = a.pack.ptr[q]; pack_end = a.pack.ptr[q + 1];
        if (pack_end > pack_begin) {
            seg0_count = a.pack.seg_count[pack_begin]; seg0_index = a.pack.index[pack_begin]; seg0_off = a.pack.seg_off[pack_begin];
        }
    }
    RegisterSums s;
    if constexpr (IMG) {
        s.t[0] = s.t[1] = s.t[2] = 0.0f;
        csr_row_sums_img(a.c, a.img, q, win + (threadIdx.x >> 6) * IMG_WIN_WORDS, s.i);
    } else {
        csr_row_sums<ELEC, CHEM, PEER>(a.c, q, s.i, s.t);
    }
    uint32_t spike = 0;
    float v_new = 0.0f;
    // (a range-set shard owns whole 64-blocks of the global index space only in part: the rows of the neurons it does not own
    // are holes -- no synapses, no update, k_update's rule)
    if (q < a.c.g.n_loc && a.up.rows.active(q, a.c.g.n_loc)) spike = update_neuron_at<MODEL, RegisterSums, true, CHEM>(a.up, q, s, a.up.clock, a.up.vhist_row, &v_new);
    if (a.up.spike_row) {
        // one raster word per wavefront = one aligned 64-block of the GLOBAL index space (as in k_update: a range-set shard maps
        // every slice to such a block; rows it does not own contribute 0)
        const unsigned long long word = __ballot(spike != 0);
        if ((threadIdx.x & 63u) == 0 && q < a.c.g.n_loc) {
            const uint32_t g = a.up.rows.block ? a.up.rows.block[q >> 6] * 64u : a.up.q0 + q;
            if (g < a.up.n.n_pad) a.up.spike_row[g >> 6] = word;
        }
    }
    if (PEER && pack_end > pack_begin && a.pack.dst) {
        // peer form: the voltage (from its register) and the spike flag as ONE granule per reading peer, stored into that
        // peer's receive set -- once the peer is done with the step that read the set's previous contents
        const uint32_t g = a.up.rows.global_of(q);
        const uint32_t *xo = reinterpret_cast<const uint32_t *>(a.up.xout);
        peer_delay(a.pack.delay, a.pack.delay_seed, a.pack.tag_out * 4u + 1u);
        for (uint32_t e = pack_begin; e < pack_end; ++e) {
            const uint32_t peer = a.pack.peer[e];
            if (__hip_atomic_load(a.pack.flags + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.pack.need_done &&
                !__hip_atomic_load(a.pack.failed.word[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                uint32_t spins = 0;
                while (__hip_atomic_load(a.pack.flags + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.pack.need_done &&
                       ++spins < a.pack.spin_limit) {}
                if (spins >= a.pack.spin_limit) peer_give_up(a.pack.failed.word);
            }
            // one granule per plane on the wire, adjacent in the peer's set (the voltage from its register, a transmitter
            // concentration as this thread has just written it)
            for (uint32_t pl = 0; pl < a.pack.planes; ++pl) {
                const uint32_t plane = a.pack.plane_id[pl];
                const uint32_t value = plane == (uint32_t)PLANE_V ? __float_as_uint(v_new) : xo[a.up.n.xl.at(g, (int)plane)];
                const uint32_t flag = (plane >= (uint32_t)PLANE_T0 && a.pack.nt_flags[(size_t)(plane - PLANE_T0) * a.pack.n_pad + g]) ? PEER_FLAG_BIT : 0u;
                const unsigned long long x = ((unsigned long long)(a.pack.tag_out | (spike ? PEER_SPIKE_BIT : 0u) | flag) << 32) | value;
                __hip_atomic_store(a.pack.dst[e] + pl, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    } else if (pack_end > pack_begin) {
        // the values this thread has just written, into every segment that carries the neuron (the voltage from its
        // register); the spike as ONE bit OR-ed into the segment's bitmap (zeroed after the previous exchange; spikes are rare)
        const uint32_t g = a.up.rows.global_of(q);
        const uint32_t *x = reinterpret_cast<const uint32_t *>(a.up.xout);
        for (uint32_t e = pack_begin; e < pack_end; ++e) {
            const bool first = e == pack_begin;
            const uint32_t count = first ? seg0_count : a.pack.seg_count[e], i = first ? seg0_index : a.pack.index[e];
            uint32_t *out = a.pack.buf + (first ? seg0_off : a.pack.seg_off[e]);
            for (uint32_t pl = 0; pl < a.pack.planes; ++pl)
                out[(size_t)pl * count + i] = a.pack.plane_id[pl] == PLANE_V ? __float_as_uint(v_new)
                                                                             : x[a.up.n.xl.at(g, (int)a.pack.plane_id[pl])];
            if (spike) atomicOr(out + (size_t)a.pack.planes * count + (i >> 5), 1u << (i & 31u));
        }
    }
}

// (second launch bound = wavefronts per SIMD the register allocation must leave room for: the prefetching electrical step sits
// two registers above the 64 that 8 wavefronts allow, and 16 384 wavefronts of BASELINE configs[4] are two full rounds of 8)
template <int MODEL, bool ELEC, bool CHEM, bool PEER = false>
__global__ __launch_bounds__(256, (CSR_PREFETCH && ELEC && !CHEM && !PEER) ? 8 : 1) void k_step_csr(const CsrStepArgs a)
{
    // peer form: THIS launch running means the previous one of the stream is over -- every row and the mirror job of the step
    // before have read what they had to read -- which is what the neighbours wait for before they overwrite a receive set.  One
    // thread says so.  (A counter of finished workgroups at the END of the launch was measured first: 2048 atomics on one
    // address cost 27 us of a 35 us step.)
    // (snn_layout.hpp: 46 lines of arguments in one round trip -- by the workgroups that start on a cold scalar cache; all 4 096
    // of BASELINE configs[4]'s launch doing it cost that launch 0.6 us of 33, session 12)
    if (blockIdx.x < 512u) warm_kernel_arguments<sizeof(CsrStepArgs)>();
    if (PEER && a.peer.signal && blockIdx.x == 0 && threadIdx.x == 0) {
        peer_delay(a.peer.delay, a.peer.delay_seed, a.peer.done_value * 4u + 2u);
        for (uint32_t i = 0; i < a.peer.n_signal; ++i)
            __hip_atomic_store(a.peer.signal[i], a.peer.done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    step_csr_block<MODEL, ELEC, CHEM, PEER>(a);
}

// the electrical step over the step image (static weights, no peer form): 16 KiB of LDS per workgroup, one window per wavefront.
// (No second launch bound: with a ring of four records every model but Hodgkin-Huxley allocates at most 64 registers by itself --
// eight wavefronts per SIMD, tests/test_isa_resources.py -- and forcing Hodgkin-Huxley there makes it spill.)
// PACK: the launch's rows write their wire values into outgoing segments (border slices of a shard handle) -- its own
// instantiation: the pack table's preloaded words cost the 65th register, i.e. one of eight wavefronts per SIMD
template <int MODEL, bool PACK = false>
__global__ __launch_bounds__(256) void k_step_csr_img(const CsrStepArgs a)
{
    __shared__ uint32_t win[4 * IMG_WIN_WORDS];
    if (blockIdx.x < 512u) warm_kernel_arguments<sizeof(CsrStepArgs)>();
    step_csr_block<MODEL, true, false, false, true, PACK>(a, win);
}

// the records of the step image from the SELL arrays: {plan_win[k], w[k], plan_win[k + 1], w[k + 1]} per lane and pair
__global__ __launch_bounds__(256) void k_csr_image(SellGraph g, const uint32_t *plan_win, const uint32_t *hdr, uint4 *rec)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if ((q >> 6) >= g.n_slices) return;
    const uint32_t s0 = g.slice_ptr[q >> 6];
    const uint32_t width = (g.slice_ptr[(q >> 6) + 1] - s0) >> 6;
    const uint32_t first = hdr[(size_t)(q >> 6) * IMG_HDR_WORDS];
    for (uint32_t j = 0; 2 * j < width; ++j) {
        const size_t e0 = s0 + (q & 63u) + (size_t)(2 * j) * 64, e1 = e0 + 64;
        uint4 r{plan_win[e0], __float_as_uint(g.w[e0]), PLAN_CODE, 0u};
        if (2 * j + 1 < width) { r.z = plan_win[e1]; r.w = __float_as_uint(g.w[e1]); }
        rec[first + (size_t)j * 64 + (q & 63u)] = r;
    }
}

// static counts of a sparse graph: n_in = row length, tcount[k] = entries whose presynaptic cell carries type k
struct CsrCountArgs {
    SellGraph g;
    uint32_t n_neurons, ld;
    const uint32_t *nt_flags; uint32_t n_pad;
    const uint32_t *st_nt_flags; uint32_t c_pad;
    uint32_t *n_in, *tcount;
};

__global__ __launch_bounds__(256) void k_csr_count(const CsrCountArgs a)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.g.n_loc) return;
    const uint32_t base = a.g.slice_ptr[q >> 6] + (q & 63u);
    const uint32_t len = a.g.row_len[q];
    uint32_t tc[K_TYPES] = {0, 0, 0};
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t p = a.g.pre[base + (size_t)k * 64];
#pragma unroll
        for (int kk = 0; kk < K_TYPES; ++kk) {
            const uint32_t f = (p < a.n_neurons) ? a.nt_flags[(size_t)kk * a.n_pad + p]
                                                 : a.st_nt_flags[(size_t)kk * a.c_pad + (p - a.n_neurons)];
            tc[kk] += f ? 1u : 0u;
        }
    }
    a.n_in[q] = len;
#pragma unroll
    for (int kk = 0; kk < K_TYPES; ++kk) a.tcount[(size_t)kk * a.ld + q] = tc[kk];
}

// The gather plan of the row sums (SellGraph::plan) from the SELL indices: one thread per row.  halo_word: per neuron the
// word of the receive buffer that carries its voltage (0xFFFFFFFF: read from the exchanged state), or null.
__global__ __launch_bounds__(256) void k_csr_plan(SellGraph g, uint32_t *plan, const uint32_t *halo_word, uint32_t n_neurons,
                                                  uint32_t halo_base)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if ((q >> 6) >= g.n_slices) return;
    const uint32_t s0 = g.slice_ptr[q >> 6];
    const uint32_t width = (g.slice_ptr[(q >> 6) + 1] - s0) >> 6;
    uint32_t prev = 0;
    for (uint32_t k = 0; k < width; ++k) {
        const size_t e = s0 + (q & 63u) + (size_t)k * 64;
        const uint32_t p = g.pre[e];
        uint32_t word = SELL_PAD;
        if (p != SELL_PAD) {
            uint32_t code = p;
            if (halo_word && p < n_neurons && halo_word[p] != 0xFFFFFFFFu) code = halo_base + halo_word[p];
            word = code | ((k == 0 || p / CHUNK != prev / CHUNK) ? 0x80000000u : 0u);
            prev = p;
        }
        plan[e] = word;
    }
}

// STDP on the sparse form: incoming edges of a listed local neuron are its SELL row; outgoing edges of a listed
// neuron are the transpose list of that presynaptic index (CSR edge ids -> SELL slots).
struct CsrStdpArgs {
    SellGraph g;
    StdpArgs s;
};

__global__ __launch_bounds__(64) void k_stdp_csr_in(const CsrStdpArgs a)
{
    const uint32_t count = *a.s.spike_count;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t j = a.s.spike_list[i];
        const uint32_t r = a.s.rows.local_of(j, a.s.n_loc);
        if (r == 0xFFFFFFFFu) continue;
        const float *prm = a.s.stdp + PL_STRIDE * a.s.lattice_slot[j];
        const bool bcm = prm[5] != 0.0f;
        const float post_act = bcm ? a.s.act[j] : 0.0f, post_avg = bcm ? a.s.avg[j] : 0.0f;
        const int32_t tj = a.s.last_firing_time[j];
        const uint32_t base = a.g.slice_ptr[r >> 6] + (r & 63u);
        const uint32_t len = a.g.row_len[r];
        for (uint32_t k = threadIdx.x; k < len; k += 64) {
            const size_t e = base + (size_t)k * 64;
            const uint32_t p = a.g.pre[e];
            const int32_t tp = (p < a.s.n_neurons) ? a.s.last_firing_time[p] : a.s.st_last_firing_time[p - a.s.n_neurons];
            const float pre = bcm ? ((p < a.s.n_neurons) ? a.s.act[p] : a.s.st_act[p - a.s.n_neurons]) : 0.0f;
            if (!plain_connection(a.s, p, a.s.lattice_slot[j])) continue;      // (a connection of a reward-modulated network)
            a.g.w[e] = plasticity_weight(prm, a.g.w[e], tp, tj, pre, post_act, post_avg);
        }
    }
}

__global__ __launch_bounds__(64) void k_stdp_csr_out(const CsrStdpArgs a)
{
    const uint32_t count = *a.s.spike_count;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t j = a.s.spike_list[i];
        const int32_t tj = a.s.last_firing_time[j];
        for (uint32_t t = a.g.t_ptr[j] + threadIdx.x; t < a.g.t_ptr[j + 1]; t += 64) {
            const uint32_t edge = a.g.t_edge[t];
            const uint32_t e = a.g.edge_slot[edge];
            const uint32_t gr = a.s.rows.global_of(a.g.edge_post[edge]);
            const float *prm = a.s.stdp + PL_STRIDE * a.s.lattice_slot[gr];
            const bool bcm = prm[5] != 0.0f;
            if (!plain_connection(a.s, j, a.s.lattice_slot[gr])) continue;
            a.g.w[e] = plasticity_weight(prm, a.g.w[e], tj, a.s.last_firing_time[gr], bcm ? a.s.act[j] : 0.0f,
                                         bcm ? a.s.act[gr] : 0.0f, bcm ? a.s.avg[gr] : 0.0f);
        }
    }
}

// Reward modulation on the sparse form: one thread per SELL entry (weights and traces share the entry index).
struct CsrRewardArgs {
    SellGraph g;
    float *c;                              // [entries] TraceRSTDP::c
    uint32_t q0, n_neurons;
    RowMap rows;
    const int32_t *last_firing_time;
    const uint32_t *lattice_slot;
    const float *rm;
    const uint32_t *rm_on;
    int dop;
};

__global__ __launch_bounds__(256) void k_rstdp_csr(const CsrRewardArgs a)
{
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.g.n_loc || !a.rows.active(q, a.g.n_loc)) return;
    const uint32_t gq = a.rows.global_of(q);
    const uint32_t sq = a.lattice_slot[gq];
    if (!(a.rm_on[sq] & RM_DO_MODULATION)) return;
    const float *m = a.rm + (size_t)sq * RM_STRIDE;
    const int32_t tq = a.last_firing_time[gq];
    const uint32_t base = a.g.slice_ptr[q >> 6] + (q & 63u);
    const uint32_t len = a.g.row_len[q];
    for (uint32_t k = 0; k < len; ++k) {
        const size_t e = base + (size_t)k * 64;
        const uint32_t p = a.g.pre[e];
        if (p >= a.n_neurons || a.lattice_slot[p] != sq) continue;
        float w = a.g.w[e], c = a.c[e];
        rstdp_edge(w, c, a.last_firing_time[p], tq, m, a.dop);
        a.g.w[e] = w;
        a.c[e] = c;
    }
}

// ---- connections between the lattices of a reward-modulated network on the sparse form (round 5) -------------------------------
// k_reward_cross with the pair (x < y) found from the stored edges: a thread walks the entries of one postsynaptic row; an entry
// p -> q of kind 1 / 2 between two lattices is half of the pair (min, max); its reverse is looked up by binary search in the row of
// p (rows are sorted by presynaptic index).  The pair is replayed by the thread that holds the edge FROM the higher index (y -> x),
// or, when that edge does not exist, by the one that holds x -> y.  Trace, dw and counter are per stored entry (C, P, K).
// Unsharded handles: every row is local.
struct CsrRewardCrossArgs {
    SellGraph g;
    RewardCrossArgs r;             // r.s.W is unused: the weights are g.w
};

__device__ __forceinline__ uint32_t csr_find_entry(const SellGraph &g, uint32_t row, uint32_t pre)
{
    const uint32_t base = g.slice_ptr[row >> 6] + (row & 63u), len = g.row_len[row];
    uint32_t lo = 0, hi = len;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.pre[base + (size_t)mid * 64] < pre) lo = mid + 1; else hi = mid;
    }
    return (lo < len && g.pre[base + (size_t)lo * 64] == pre) ? base + lo * 64u : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void k_reward_cross_csr(const CsrRewardCrossArgs a)
{
    const StdpArgs &s = a.r.s;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;                 // local row = neuron (unsharded)
    if (q >= a.g.n_loc || q >= s.n_neurons) return;
    const uint32_t lq = s.lattice_slot[q];
    const uint32_t base = a.g.slice_ptr[q >> 6] + (q & 63u), len = a.g.row_len[q];
    const uint32_t *spike = reinterpret_cast<const uint32_t *>(s.xbuf);
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t e = base + k * 64u;
        const uint32_t p = a.g.pre[e];
        const bool p_neuron = p < s.n_neurons;
        if (p_neuron && s.lattice_slot[p] == lq) continue;
        const uint32_t kind_e = cross_kind(s, p, lq);
        if (kind_e == 0u) continue;
        // the pair: x < y; yx = the edge y -> x, xy = the edge x -> y
        const bool e_is_yx = p > q;
        const uint32_t x = e_is_yx ? q : p, y = e_is_yx ? p : q;
        const bool y_neuron = y < s.n_neurons;
        uint32_t e_yx = e_is_yx ? e : 0xFFFFFFFFu, e_xy = e_is_yx ? 0xFFFFFFFFu : e;
        const uint32_t lx = s.lattice_slot[x], ly = y_neuron ? s.lattice_slot[y] : 0u;
        if (e_is_yx) {
            if (y_neuron) e_xy = csr_find_entry(a.g, y, x);                                  // row of y holds x -> y
            if (e_xy != 0xFFFFFFFFu && cross_kind(s, x, ly) == 0u) e_xy = 0xFFFFFFFFu;
        } else {
            e_yx = csr_find_entry(a.g, x, y);                                                // row of x holds y -> x
            if (e_yx != 0xFFFFFFFFu && cross_kind(s, y, lx) != 0u) continue;                 // that entry's thread replays the pair
            e_yx = 0xFFFFFFFFu;
        }
        CrossEdge yx{}, xy{};
        yx.exists = e_yx != 0xFFFFFFFFu; xy.exists = e_xy != 0xFFFFFFFFu;
        if (yx.exists) { yx.kind = cross_kind(s, y, lx); yx.w = a.g.w[e_yx]; }
        if (xy.exists) { xy.kind = cross_kind(s, x, ly); xy.w = a.g.w[e_xy]; }
        const bool x_is_mod = (a.r.rm_on[lx] & RM_IS_MODULATED) != 0u, y_is_mod = y_neuron && (a.r.rm_on[ly] & RM_IS_MODULATED) != 0u;
        const bool x_mod = x_is_mod && (a.r.rm_on[lx] & RM_DO_MODULATION) != 0u, y_mod = y_is_mod && (a.r.rm_on[ly] & RM_DO_MODULATION) != 0u;
        const bool x_plain = !x_is_mod && s.do_plasticity[lx] && spike[s.xl.at(x, PLANE_SPIKE)] != 0u;
        const bool y_plain = y_neuron && !y_is_mod && s.do_plasticity[ly] && spike[s.xl.at(y, PLANE_SPIKE)] != 0u;
        if (!(x_mod || x_plain || y_mod || y_plain)) continue;
        if (yx.exists) { yx.c = a.r.C[e_yx]; yx.dw = a.r.P[e_yx]; yx.k = a.r.K[e_yx]; }
        if (xy.exists) { xy.c = a.r.C[e_xy]; xy.dw = a.r.P[e_xy]; xy.k = a.r.K[e_xy]; }
        const int tx = s.last_firing_time[x];
        const int ty = y_neuron ? s.last_firing_time[y] : s.st_last_firing_time[y - s.n_neurons];
        if (x_plain) cross_visit(a.r, yx, xy, lx, ly, y_neuron, tx, ty);
        if (y_plain) cross_visit(a.r, xy, yx, ly, lx, true, ty, tx);
        if (x_mod) cross_visit(a.r, yx, xy, lx, ly, y_neuron, tx, ty);
        if (y_mod) cross_visit(a.r, xy, yx, ly, lx, true, ty, tx);
        if (yx.exists) { a.g.w[e_yx] = yx.w; a.r.C[e_yx] = yx.c; a.r.P[e_yx] = yx.dw; a.r.K[e_yx] = yx.k; }
        if (xy.exists) { a.g.w[e_xy] = xy.w; a.r.C[e_xy] = xy.c; a.r.P[e_xy] = xy.dw; a.r.K[e_xy] = xy.k; }
    }
}

// k_reward_cross_check on the sparse form (same refusal classes)
__global__ __launch_bounds__(256) void k_reward_cross_check_csr(const CsrRewardCrossArgs a)
{
    const StdpArgs &s = a.r.s;
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= a.g.n_loc || q >= s.n_neurons) return;
    const uint32_t lq = s.lattice_slot[q];
    const bool mod_q = (a.r.rm_on[lq] & RM_IS_MODULATED) != 0u, plastic_q = !mod_q && s.do_plasticity[lq] != 0u;
    const uint32_t base = a.g.slice_ptr[q >> 6] + (q & 63u), len = a.g.row_len[q];
    uint32_t bad = 0;
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t p = a.g.pre[base + k * 64u];
        const uint32_t kind = cross_kind(s, p, lq);
        if (kind == 0u) continue;
        if (plastic_q && s.stdp[PL_STRIDE * lq + 5] != 0.0f) bad = max(bad, 4u);
        if (p >= s.n_neurons) {
            if (kind == 1u && plastic_q) bad = max(bad, 2u);
            continue;
        }
        const uint32_t lp = s.lattice_slot[p];
        if (lp == lq) continue;
        const bool mod_p = (a.r.rm_on[lp] & RM_IS_MODULATED) != 0u, plastic_p = !mod_p && s.do_plasticity[lp] != 0u;
        if (plastic_p && s.stdp[PL_STRIDE * lp + 5] != 0.0f) bad = max(bad, 4u);
        if (kind == 1u && !mod_p && !mod_q && (plastic_p || plastic_q)) bad = max(bad, 2u);
        if (kind == 2u && ((plastic_p && mod_q) || (plastic_q && mod_p))) bad = max(bad, 3u);
        if ((mod_p && (a.r.rm_on[lp] & RM_DO_MODULATION)) || plastic_p) {
            if (csr_find_entry(a.g, p, q) == 0xFFFFFFFFu || s.conn_kind[(size_t)lq * s.n_lattices + lp] != kind) bad = max(bad, 1u);
        }
    }
    if (bad) atomicMax(a.r.bad, bad);
}

} // namespace snn
