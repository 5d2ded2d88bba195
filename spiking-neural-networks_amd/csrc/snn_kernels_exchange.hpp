// The per-step exchange of a shard handle (multi-GPU: one process per GPU, post-population shards).
//
// What another rank reads of a neuron is its presynaptic state S(t): current_voltage (only when gap junctions are
// on), the concentration t of every transmitter type some NEURON of the network releases (only when chemical
// synapses are on) and whether it spiked (last_firing_time of the presynaptic side of STDP).  Spike-train cells are
// replicated and never travel.  The reference has no distributed path; the data dependence is
// neuron/mod.rs:2640-2647 (every input of step t reads the state of step t - 1 only).
//
// Wire format of one SEGMENT (the values of `count` neurons travelling from one rank to another), 32-bit words:
//     [plane 0: count f32] ... [plane P-1: count f32] [spike bitmap: ceil(count / 32) u32, bit i%32 of word i/32]
// P = planes of the handle's exchange plan (0..4).  4 B per plane and neuron + ONE BIT for the spike.
//   all-gather mode (dense handles): one segment per shard = its whole slot of `stride` neurons; the segments of all
//     shards sit at shard * block_words in one buffer, filled by ONE in-place all-gather.
//   halo mode (sparse handles): one segment per peer = exactly the neurons that peer's rows reference (send list),
//     all-to-all-v between the ranks.
// k_exchange_pack fills the outgoing segments from the mirror (xbuf); k_exchange_unpack writes the incoming ones
// into the mirror at the neurons' global indices and stamps last_firing_time -- after it every kernel of the step
// addresses presynaptic state exactly as on a single GPU.
#pragma once
#include "snn_layout.hpp"

namespace snn {

constexpr int WIRE_MAX_PLANES = 1 + K_TYPES;

struct WireArgs {
    float *xbuf;                    // mirror
    XLayout xl;
    uint32_t n_neurons;
    uint32_t planes;
    uint32_t plane_id[WIRE_MAX_PLANES];   // mirror plane carried by wire plane s (PLANE_V or PLANE_T0 + k)
    uint32_t *buf;                  // outgoing (pack) / incoming (unpack) segments
    // per segment (blockIdx.y): neurons, word offset inside buf, and its neurons -- a contiguous range starting at
    // `first`, or list[list_offset ..] (global indices) when list != nullptr
    const uint32_t *seg_count;
    const uint64_t *seg_offset;
    const uint32_t *seg_first;
    const uint64_t *seg_list_offset;
    const uint32_t *list;
    uint32_t skip;                  // unpack: segment that is not applied (the handle's own slot), or 0xFFFFFFFF
    int32_t *last_firing_time;      // unpack
    long long clock;                // unpack: the step being closed
    float *xbuf2;                   // unpack: a second copy of the mirror that receives the same values, or null
};

__device__ __forceinline__ uint32_t wire_neuron(const WireArgs &a, uint32_t seg, uint32_t i)
{
    return a.list ? a.list[a.seg_list_offset[seg] + i] : a.seg_first[seg] + i;
}

__global__ __launch_bounds__(256) void k_exchange_pack(const WireArgs a)
{
    const uint32_t seg = blockIdx.y;
    const uint32_t count = a.seg_count[seg];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if ((i & ~63u) >= count) return;                        // whole wavefront past the segment
    uint32_t *out = a.buf + a.seg_offset[seg];
    bool spike = false;
    if (i < count) {
        const uint32_t g = wire_neuron(a, seg, i);
        const bool real = g < a.n_neurons;                  // padding of a slot past the population
        for (uint32_t s = 0; s < a.planes; ++s)
            out[(size_t)s * count + i] = real ? reinterpret_cast<const uint32_t *>(a.xbuf)[a.xl.at(g, (int)a.plane_id[s])] : 0u;
        spike = real && reinterpret_cast<const uint32_t *>(a.xbuf)[a.xl.at(g, PLANE_SPIKE)] != 0u;
    }
    // spike bitmap: the wavefront's ballot = two 32-bit words
    const unsigned long long word = __ballot(spike);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_words = (count + 31) / 32;
    uint32_t *bits = out + (size_t)a.planes * count;
    const uint32_t w0 = (i >> 6) * 2;
    if (lane == 0 && w0 < n_words) bits[w0] = (uint32_t)word;
    if (lane == 32 && w0 + 1 < n_words) bits[w0 + 1] = (uint32_t)(word >> 32);
}

__global__ __launch_bounds__(256) void k_exchange_unpack(const WireArgs a)
{
    const uint32_t seg = blockIdx.y;
    if (seg == a.skip) return;
    const uint32_t count = a.seg_count[seg];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t g = wire_neuron(a, seg, i);
    if (g >= a.n_neurons) return;
    const uint32_t *in = a.buf + a.seg_offset[seg];
    uint32_t *x = reinterpret_cast<uint32_t *>(a.xbuf), *x2 = reinterpret_cast<uint32_t *>(a.xbuf2);
    for (uint32_t s = 0; s < a.planes; ++s) {
        const uint32_t v = in[(size_t)s * count + i];
        x[a.xl.at(g, (int)a.plane_id[s])] = v;
        if (x2) x2[a.xl.at(g, (int)a.plane_id[s])] = v;
    }
    const uint32_t spike = (in[(size_t)a.planes * count + (i >> 5)] >> (i & 31u)) & 1u;
    x[a.xl.at(g, PLANE_SPIKE)] = spike;
    if (x2) x2[a.xl.at(g, PLANE_SPIKE)] = spike;
    if (spike) a.last_firing_time[g] = (int32_t)a.clock;   // neuron/mod.rs:2555-2557 for a neuron owned elsewhere
}

} // namespace snn
