// Exchange plan of a shard handle, pack / unpack launches, the halo lists of sparse handles, and the in-library
// collective (RCCL called directly: all-gather of whole slots or grouped send/recv of halo segments) with the
// sharded step loop snn_run_sharded.  Kernels and wire format: snn_kernels_exchange.hpp.
// Included by snn_network.hip only (one translation unit).
#pragma once
#include <atomic>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

#include "snn_network_step.hpp"

namespace {

inline uint64_t segment_words(uint32_t planes, uint64_t count) { return (uint64_t)planes * count + (count + 31) / 32; }

template <typename T>
int upload_table(snn_network *net, T **dev, const hvec<T> &host)
{
    if (*dev) { (void)hipFree(*dev); *dev = nullptr; }
    HIP_TRY(snn_malloc(dev, std::max<size_t>(host.size() * sizeof(T), 256)), SNN_ERR_BUFFER_CREATE);
    if (!host.empty())
        HIP_TRY(copy_sync(net, *dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    return SNN_OK;
}

// which shard owns neuron p
uint32_t owner_of(const snn_network *net, uint32_t p)
{
    if (!net->block_mode) return net->shard_stride ? p / net->shard_stride : 0u;
    for (const auto &l : net->lattices)
        if (p >= l.first && p < l.first + l.count) return (p - l.first) / net->lattice_slab[l.slot];
    return 0u;
}
bool owns(const snn_network *net, uint32_t p)
{
    return net->block_mode ? net->local_row_host[p] != 0xFFFFFFFFu : (p >= net->q0 && p < net->q1);
}

// A range-set shard without a committed halo plan trades whole ownerships (what the all-gather does for contiguous
// shards): it reads every neuron of every peer and sends all of its own to each.
void synthesize_full_lists(snn_network *net)
{
    net->halo_need.assign(net->n_shards, {});
    net->halo_send.assign(net->n_shards, {});
    for (uint32_t p = 0; p < net->nn; ++p) {
        const uint32_t o = owner_of(net, p);
        if (o == net->shard_index) {
            for (uint32_t s = 0; s < net->n_shards; ++s)
                if (s != net->shard_index) net->halo_send[s].push_back(p);
        } else if (o < net->n_shards) {
            net->halo_need[o].push_back(p);
        }
    }
}

// the peer form's buffers and connection belong to ONE plan: a rebuilt plan starts unconnected
int p2p_release(snn_network *net, bool final = false)
{
    // What PEERS may still address -- the two receive sets and the done counters, by committed tables or IPC mappings of the
    // old plan -- is not freed here but kept until this handle is destroyed: a neighbour that
    // announces "my previous launch is over" at the start of its next run stores into these words (include/snn_amd.h: peers
    // reconnect after a plan rebuild; until they have, their stores land in memory that is still this handle's).
    for (void *b : {(void *)net->p2p_recv[0], (void *)net->p2p_recv[1], (void *)net->p2p_flags})
        if (b) net->p2p_retired.push_back(b);
    if (final) {
        for (void *b : net->p2p_retired) (void)hipFree(b);
        net->p2p_retired.clear();
    }
    for (void *b : {(void *)net->p2p_done_blocks, (void *)net->p2p_dst_dev[0], (void *)net->p2p_dst_dev[1], (void *)net->p2p_peer_dev,
                    (void *)net->p2p_signal_dev})
        if (b) (void)hipFree(b);
    net->p2p_recv[0] = net->p2p_recv[1] = nullptr;
    net->p2p_flags = nullptr; net->p2p_done_blocks = nullptr;
    net->p2p_dst_dev[0] = net->p2p_dst_dev[1] = nullptr; net->p2p_peer_dev = nullptr; net->p2p_signal_dev = nullptr;
    net->p2p_n_signal = 0; net->p2p_recv_words = 0;
    net->p2p_peers.clear();
    net->p2p_connected = false;
    return SNN_OK;
}

// (Re)builds the plan: planes on the wire, per-peer segments, device tables.  which = 0 pack, 1 unpack.
int ensure_exchange_plan(snn_network *net)
{
    if (!net->x_dirty) return SNN_OK;
    if (!net->sharded) { net->x_dirty = false; return SNN_OK; }
    net->x_agreed = false;                                           // a rebuilt plan is compared with the peers' again
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);        // nothing in flight reads the old tables
    // planes: voltage for gap junctions; t of the types some NEURON releases for chemical synapses
    net->x_planes = 0;
    if (net->electrical) net->x_plane_id[net->x_planes++] = PLANE_V;
    if (net->chemical) {
        uint32_t mask = 0;
        for (const auto &kv : net->lattice_nt_mask) {
            const LatticeInfo *li = find_lattice(net, kv.first);
            if (li && !li->spike_train) mask |= kv.second;
        }
        for (uint32_t k = 0; k < K_TYPES; ++k)
            if (mask >> k & 1u) net->x_plane_id[net->x_planes++] = PLANE_T0 + k;
    }
    const uint32_t G = net->n_shards, me = net->shard_index, P = net->x_planes;
    net->x_send_off.assign(G, 0); net->x_send_words.assign(G, 0);
    net->x_recv_off.assign(G, 0); net->x_recv_words.assign(G, 0);
    hvec<uint32_t> cnt[2], first[2];
    hvec<uint64_t> off[2], loff[2];
    net->x_mode = ((net->csr && net->halo_committed) || net->block_mode) ? SNN_EXCHANGE_HALO : SNN_EXCHANGE_ALLGATHER;
    net->direct_capable = false;
    net->peer_capable = false;
    if (net->block_mode && !net->halo_committed) synthesize_full_lists(net);
    if (net->x_mode == SNN_EXCHANGE_ALLGATHER) {
        net->x_block_words = segment_words(P, net->shard_stride);
        // a new plan lays the slots out anew: no word of the old layout may survive where the new one expects padding zeros
        // (k_update writes only the entries of the shard's own neurons)
        if (net->wire && net->wire_words)
            HIP_TRY(hipMemsetAsync(net->wire, 0, net->wire_words * 4, net->stream), SNN_ERR_BUFFER_WRITE);
        for (uint32_t p = 0; p < G; ++p) {
            net->x_send_off[p] = 0; net->x_send_words[p] = net->x_block_words;
            net->x_recv_off[p] = (uint64_t)p * net->x_block_words; net->x_recv_words[p] = net->x_block_words;
            cnt[1].push_back(net->shard_stride); off[1].push_back((uint64_t)p * net->x_block_words);
            first[1].push_back(p * net->shard_stride); loff[1].push_back(0);
        }
        cnt[0].push_back(net->shard_stride); off[0].push_back((uint64_t)me * net->x_block_words);
        first[0].push_back(me * net->shard_stride); loff[0].push_back(0);
    } else {
        hvec<uint32_t> send_idx, recv_idx;
        uint64_t so = 0, ro = 0;
        for (uint32_t p = 0; p < G; ++p) {
            const auto &sl = net->halo_send[p];
            const auto &nl = net->halo_need[p];
            net->x_send_off[p] = so; net->x_send_words[p] = sl.empty() ? 0 : segment_words(P, sl.size());
            net->x_recv_off[p] = ro; net->x_recv_words[p] = nl.empty() ? 0 : segment_words(P, nl.size());
            if (!sl.empty()) {
                cnt[0].push_back((uint32_t)sl.size()); off[0].push_back(so); first[0].push_back(0); loff[0].push_back(send_idx.size());
                send_idx.insert(send_idx.end(), sl.begin(), sl.end());
            }
            if (!nl.empty()) {
                cnt[1].push_back((uint32_t)nl.size()); off[1].push_back(ro); first[1].push_back(0); loff[1].push_back(recv_idx.size());
                recv_idx.insert(recv_idx.end(), nl.begin(), nl.end());
            }
            so += net->x_send_words[p];
            ro += net->x_recv_words[p];
        }
        TRY(upload_table(net, &net->halo_send_idx, send_idx));
        TRY(upload_table(net, &net->halo_recv_idx, recv_idx));
        net->recv_total = (uint32_t)recv_idx.size();
        // the one-launch sparse step: border / interior slices and the per-row pack table
        const uint32_t n_slices = (net->n_loc + 63) / 64;
        hvec<uint8_t> is_border(n_slices, 0);
        hvec<uint32_t> pack_ptr(net->n_loc + 1, 0), pack_segoff, pack_count, pack_index;
        auto local_row = [&](uint32_t g) { return net->block_mode ? net->local_row_host[g] : g - net->q0; };
        for (uint32_t p = 0; p < G; ++p)
            for (uint32_t g : net->halo_send[p]) pack_ptr[local_row(g) + 1] += 1;
        for (uint32_t q = 0; q < net->n_loc; ++q) pack_ptr[q + 1] += pack_ptr[q];
        pack_segoff.resize(pack_ptr[net->n_loc]); pack_count.resize(pack_ptr[net->n_loc]); pack_index.resize(pack_ptr[net->n_loc]);
        {
            hvec<uint32_t> fill(pack_ptr.begin(), pack_ptr.end() - 1);
            net->send_bitmap_words = 0;
            for (uint32_t p = 0; p < G; ++p) {
                const auto &sl = net->halo_send[p];
                net->send_bitmap_words += (uint32_t)(sl.size() + 31) / 32;
                for (uint32_t i = 0; i < sl.size(); ++i) {
                    const uint32_t q = local_row(sl[i]), e = fill[q]++;
                    pack_segoff[e] = (uint32_t)net->x_send_off[p]; pack_count[e] = (uint32_t)sl.size(); pack_index[e] = i;
                    is_border[q >> 6] = 1;
                }
            }
        }
        hvec<uint32_t> border, interior;
        for (uint32_t sl = 0; sl < n_slices; ++sl) (is_border[sl] ? border : interior).push_back(sl);
        net->n_border = (uint32_t)border.size(); net->n_interior = (uint32_t)interior.size();
        TRY(upload_table(net, &net->csr_border_dev, border));
        TRY(upload_table(net, &net->csr_interior_dev, interior));
        TRY(upload_table(net, &net->pack_ptr_dev, pack_ptr));
        TRY(upload_table(net, &net->pack_segoff_dev, pack_segoff));
        TRY(upload_table(net, &net->pack_count_dev, pack_count));
        TRY(upload_table(net, &net->pack_index_dev, pack_index));
        net->send_bits_clean = true;                 // the send buffer is (re)created zeroed below
        for (uint32_t **b : {&net->halo_send_buf, &net->halo_recv_buf, &net->halo_send_buf2, &net->halo_recv_buf2,
                             &net->csr_plan_direct, &net->halo_word_dev})
            if (*b) { (void)hipFree(*b); *b = nullptr; }
        // the direct form (see snn_network_state.hpp): voltage is the only plane, so a halo neuron's value is ONE word
        net->direct_capable = net->csr && net->csr_pre && net->electrical && !net->chemical && P == 1 &&
                              (uint64_t)net->nn + net->nc + ro < PLAN_CODE && net->n_loc;
        // the PEER form carries every plane of the plan: one granule per neuron and plane (P adjacent granules per halo neuron
        // fit the words of its segment: P * count + ceil(count / 32) >= P * count)
        net->peer_capable = net->csr && net->csr_pre && P >= 1 && (uint64_t)net->nn + net->nc + ro < PLAN_CODE && net->n_loc;
        for (uint32_t **b : {&net->halo_send_buf, net->direct_capable ? &net->halo_send_buf2 : nullptr}) {
            if (!b) continue;
            HIP_TRY(snn_malloc(b, std::max<uint64_t>(so * 4, 256)), SNN_ERR_BUFFER_CREATE);
            HIP_TRY(memset_sync(net, *b, 0, std::max<uint64_t>(so * 4, 256)), SNN_ERR_BUFFER_WRITE);
        }
        for (uint32_t **b : {&net->halo_recv_buf, net->direct_capable ? &net->halo_recv_buf2 : nullptr}) {
            if (!b) continue;
            HIP_TRY(snn_malloc(b, std::max<uint64_t>(ro * 4, 256)), SNN_ERR_BUFFER_CREATE);
            HIP_TRY(memset_sync(net, *b, 0, std::max<uint64_t>(ro * 4, 256)), SNN_ERR_BUFFER_WRITE);
        }
        TRY(p2p_release(net));
        if (net->peer_capable && ro) {
            // the peer form's receive sets and done counters: fine-grained memory (another device may store into it while
            // kernels of this one read it), zeroed (tag 0 is never expected: epochs start at 1)
            net->p2p_recv_words = ro;
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(ext_malloc(reinterpret_cast<void **>(&net->p2p_recv[i]), std::max<uint64_t>(ro * 8, 256), hipDeviceMallocFinegrained),
                        SNN_ERR_BUFFER_CREATE);
                HIP_TRY(memset_sync(net, net->p2p_recv[i], 0, std::max<uint64_t>(ro * 8, 256)), SNN_ERR_BUFFER_WRITE);
            }
            HIP_TRY(ext_malloc(reinterpret_cast<void **>(&net->p2p_flags), std::max<size_t>((size_t)G * 4, 256), hipDeviceMallocFinegrained),
                    SNN_ERR_BUFFER_CREATE);
            HIP_TRY(memset_sync(net, net->p2p_flags, 0, std::max<size_t>((size_t)G * 4, 256)), SNN_ERR_BUFFER_WRITE);
            HIP_TRY(snn_malloc(&net->p2p_done_blocks, 256), SNN_ERR_BUFFER_CREATE);
            HIP_TRY(memset_sync(net, net->p2p_done_blocks, 0, 256), SNN_ERR_BUFFER_WRITE);
            if (!net->p2p_failed) {
                HIP_TRY(host_malloc(reinterpret_cast<void **>(&net->p2p_failed), 8, hipHostMallocMapped), SNN_ERR_BUFFER_CREATE);
                net->p2p_failed[0] = 0u;
            }
            net->p2p_peers.assign(G, snn_network::P2pPeer{});
            net->p2p_epoch = 1;
        }
        if (net->direct_capable || net->peer_capable) {
            // word of the receive buffer (direct form: plane 0 of the segment, P == 1) = first granule of the receive set (peer
            // form: P adjacent granules per neuron) that carries a halo neuron
            hvec<uint32_t> halo_word(net->nn, 0xFFFFFFFFu);
            for (uint32_t p = 0; p < G; ++p)
                for (size_t i = 0; i < net->halo_need[p].size(); ++i)
                    halo_word[net->halo_need[p][i]] = (uint32_t)(net->x_recv_off[p] + i * P);
            TRY(upload_table(net, &net->halo_word_dev, halo_word));
            HIP_TRY(snn_malloc(&net->csr_plan_direct, std::max<size_t>(net->sell_entries * 4, 256)),
                    SNN_ERR_BUFFER_CREATE);
            hipLaunchKernelGGL(k_csr_plan, dim3((n_slices * 64 + 255) / 256), dim3(256), 0, net->stream, csr_graph(net),
                               net->csr_plan_direct, net->halo_word_dev, net->nn, net->nn + net->nc);
            HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
            HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
            // the step image of direct runs (voltage the only plane on the wire: a halo neuron is ONE word of the receive buffer)
            for (void **q : {(void **)&net->csr_img_hdr_direct, (void **)&net->csr_plan_win_direct, (void **)&net->csr_img_rec_direct}) {
                if (*q) (void)hipFree(*q);
                *q = nullptr;
            }
            if (net->direct_capable && P == 1 && !net->sell_pre_host.empty()) {
                hvec<uint32_t> img_hdr, plan_win;
                uint64_t records = 0;
                build_step_image_plan(net, net->slice_ptr_host, net->sell_pre_host, n_slices, img_hdr, plan_win, records, net->img_staged_slices_direct,
                                      halo_word.data());
                TRY(upload_table(net, &net->csr_img_hdr_direct, img_hdr));
                TRY(upload_table(net, &net->csr_plan_win_direct, plan_win));
                HIP_TRY(snn_malloc(&net->csr_img_rec_direct, (size_t)records * 16 + 4096), SNN_ERR_BUFFER_CREATE);
                net->img_stale_direct = true;
            }
        }
    }
    for (int w = 0; w < 2; ++w) {
        net->seg_n[w] = (uint32_t)cnt[w].size();
        net->seg_max[w] = 0;
        for (uint32_t c : cnt[w]) net->seg_max[w] = std::max(net->seg_max[w], c);
        TRY(upload_table(net, &net->seg_count_dev[w], cnt[w]));
        TRY(upload_table(net, &net->seg_offset_dev[w], off[w]));
        TRY(upload_table(net, &net->seg_first_dev[w], first[w]));
        TRY(upload_table(net, &net->seg_loff_dev[w], loff[w]));
    }
    net->x_dirty = false;
    return SNN_OK;
}

// which = 0 outgoing, 1 incoming; set = which of the two sets of halo segments (direct runs; 0 otherwise)
WireArgs wire_args(snn_network *net, int which, int set)
{
    WireArgs a{};
    a.xbuf = net->xbuf; a.xl = net->xl; a.n_neurons = net->nn; a.planes = net->x_planes;
    for (int s = 0; s < WIRE_MAX_PLANES; ++s) a.plane_id[s] = net->x_plane_id[s];
    const bool halo = net->x_mode == SNN_EXCHANGE_HALO;
    a.buf = halo ? (which == 0 ? (set ? net->halo_send_buf2 : net->halo_send_buf) : (set ? net->halo_recv_buf2 : net->halo_recv_buf))
                 : net->wire;
    a.seg_count = net->seg_count_dev[which]; a.seg_offset = net->seg_offset_dev[which];
    a.seg_first = net->seg_first_dev[which]; a.seg_list_offset = net->seg_loff_dev[which];
    a.list = halo ? (which == 0 ? net->halo_send_idx : net->halo_recv_idx) : nullptr;
    a.skip = (!halo && which == 1) ? net->shard_index : 0xFFFFFFFFu;
    a.last_firing_time = net->na.last_firing_time;
    a.clock = net->clock;
    // sparse handles stepping with k_step_csr read S(t) from a shadow of the mirror: what arrives goes there too
    a.xbuf2 = (which == 1 && net->csr && net->shadow_valid) ? net->shadow[net->shadow_cur] : nullptr;
    return a;
}

// outgoing segments of the step just computed (after the neuron update)
int launch_exchange_pack(snn_network *net)
{
    if (!net->sharded || net->seg_n[0] == 0 || net->seg_max[0] == 0) return SNN_OK;
    if (net->step_packed) return SNN_OK;             // k_step_csr wrote the segments itself
    if (net->update_packed) { net->update_packed = false; return SNN_OK; }    // ... or k_update did (dense shard handles)
    net->send_bits_clean = false;                    // whole bitmap words, set bits included
    hipLaunchKernelGGL(k_exchange_pack, dim3((net->seg_max[0] + 255) / 256, net->seg_n[0]), dim3(256), 0, net->stream,
                       wire_args(net, 0));
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// incoming segments -> mirror + last_firing_time of the neurons owned elsewhere
int launch_exchange_unpack(snn_network *net)
{
    if (!net->sharded || net->seg_n[1] == 0 || net->seg_max[1] == 0 || net->nn == 0) return SNN_OK;
    if (net->x_mode == SNN_EXCHANGE_ALLGATHER && net->n_shards == 1) return SNN_OK;
    hipLaunchKernelGGL(k_exchange_unpack, dim3((net->seg_max[1] + 255) / 256, net->seg_n[1]), dim3(256), 0, net->stream,
                       wire_args(net, 1));
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// k_step_close: spike trains (as launch_spike_trains(net, 1, run_step_offset, clock + 1)), the unpack of the halo that
// has arrived, and the clearing of the outgoing spike bitmaps -- any subset, one launch
int launch_step_close(snn_network *net, bool cells, bool unpack)
{
    StepCloseArgs a{};
    uint32_t cell_work = 0;
    if (cells) cell_work = spike_train_args(net, a.cells, 1, net->run_step_offset, net->clock + 1);
    a.cell_blocks = (cell_work + 255) / 256;
    if (unpack && net->n_shards > 1 && net->seg_n[1] && net->recv_total) {
        a.recv = wire_args(net, 1);
        a.xbuf2 = a.recv.xbuf2;
        a.recv_total = net->recv_total;
        a.recv_segments = net->seg_n[1];
        a.unpack_blocks = (net->recv_total + 255) / 256;
    }
    const bool clear = !net->send_bits_clean && net->send_bitmap_words && !net->direct_run;   // (direct runs clear behind the rows)
    if (clear) {
        a.send = wire_args(net, 0);
        a.send_segments = net->seg_n[0];
        a.send_bitmap_words = net->send_bitmap_words;
    }
    const uint32_t blocks = a.blocks();
    if (!net->direct_run) net->send_bits_clean = true;
    if (blocks == 0) return SNN_OK;
    hipLaunchKernelGGL(k_step_close, dim3(blocks), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// ---- halo need lists of a sparse shard handle -------------------------------------------------------------------
void halo_reset(snn_network *net)
{
    net->halo_need.assign(net->n_shards, {});
    net->halo_send.assign(net->n_shards, {});
    net->halo_committed = false;
    net->x_dirty = true;
}

// per peer: the distinct neurons of that peer among the presynaptic indices of the local rows (ascending)
void halo_needs_from_rows(snn_network *net, const uint32_t *pre_index, uint64_t nnz)
{
    halo_reset(net);
    if (!net->sharded || net->n_shards < 2) return;
    hvec<uint8_t> seen(net->nn, 0);
    for (uint64_t e = 0; e < nnz; ++e) {
        const uint32_t p = pre_index[e];
        if (p < net->nn && !owns(net, p)) seen[p] = 1;
    }
    for (uint32_t p = 0; p < net->nn; ++p)
        if (seen[p]) net->halo_need[owner_of(net, p)].push_back(p);
}

// ---- RCCL, resolved at first use ---------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;                              // why `lib` is null (dlerror() read ONCE, at the failing call)
    bool replaced = false;                        // snn_set_collectives: the host's functions stand in for RCCL's
};

void rccl_resolve(Rccl &r);

Rccl &rccl_state()
{
    static Rccl r;
    static std::once_flag once;               // handles of different threads may reach for RCCL at the same time
    std::call_once(once, [] { rccl_resolve(r); });
    return r;
}
Rccl *rccl() { Rccl &r = rccl_state(); return (r.lib || r.replaced) ? &r : nullptr; }

void rccl_resolve(Rccl &r)
{
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
        const char *e = dlerror();                // dlerror() clears itself: one call, kept
        r.err = e ? e : "dlopen failed";
    }
    if (!r.lib) return;
    r.err.clear();
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(r.lib, n);
        if (!p && ok) r.err = std::string("symbol missing: ") + n;
        ok = ok && p;
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) { dlclose(r.lib); r.lib = nullptr; }
}

#define RCCL_LIB(R)                                                                               \
    Rccl *R = rccl();                                                                              \
    if (!R) return fail(SNN_ERR_BAD_STATE, std::string("librccl.so.1 could not be loaded: ") + rccl_state().err)
// the communicator's life cycle is RCCL's own: not available from a replaced table (snn_set_collectives), nor without librccl
#define RCCL_REAL(R, fn)                                                                           \
    Rccl *R = rccl();                                                                              \
    if (!R || !R->lib) return fail(SNN_ERR_BAD_STATE, std::string("librccl.so.1 could not be loaded: ") + rccl_state().err); \
    if (R->replaced) return fail(SNN_ERR_BAD_STATE, "the collectives are replaced (snn_set_collectives): communicators are the host's"); \
    if (!R->fn) return fail(SNN_ERR_BAD_STATE, "RCCL entry point missing")
// entry points that call the collectives count themselves in: snn_set_collectives refuses to swap the table under them
std::atomic<int> g_collective_users{0};
struct CollectiveUser {
    CollectiveUser() { g_collective_users.fetch_add(1); }
    ~CollectiveUser() { g_collective_users.fetch_sub(1); }
};
#define RCCL_TRY(R, expr)                                                                         \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail(SNN_ERR_QUEUE, std::string(#expr) + ": " + (R)->GetErrorString(r_)); \
    } while (0)

int comm_geometry(Rccl *R, snn_network *net, ncclComm_t comm)
{
    int world = 0, rank = -1;
    RCCL_TRY(R, R->CommCount(comm, &world));
    RCCL_TRY(R, R->CommUserRank(comm, &rank));
    if (!net->sharded || (uint32_t)world != net->n_shards || (uint32_t)rank != net->shard_index)
        return fail(SNN_ERR_BAD_STATE, "communicator size / rank do not match the handle's shard geometry");
    return SNN_OK;
}

// the collective of the packed segments on `stream`
int enqueue_exchange(Rccl *R, snn_network *net, ncclComm_t comm, hipStream_t stream)
{
    if (net->x_mode == SNN_EXCHANGE_ALLGATHER) {
        if (net->x_block_words == 0) return SNN_OK;
        RCCL_TRY(R, R->AllGather(net->wire + (size_t)net->shard_index * net->x_block_words, net->wire, net->x_block_words,
                                 ncclUint32, comm, stream));
        return SNN_OK;
    }
    RCCL_TRY(R, R->GroupStart());
    // a failing call must not leave the thread's group open (every later RCCL call of this thread would queue into it):
    // remember the first error, always close the group, then report
    ncclResult_t first = ncclSuccess;
    const char *what = "";
    for (uint32_t p = 0; p < net->n_shards && first == ncclSuccess; ++p) {
        if (p == net->shard_index) continue;
        if (net->x_send_words[p]) {
            first = R->Send((net->direct_run && net->hx_par ? net->halo_send_buf2 : net->halo_send_buf) + net->x_send_off[p],
                            net->x_send_words[p], ncclUint32, (int)p, comm, stream);
            what = "ncclSend";
        }
        if (first == ncclSuccess && net->x_recv_words[p]) {
            first = R->Recv((net->direct_run && net->hx_par ? net->halo_recv_buf2 : net->halo_recv_buf) + net->x_recv_off[p],
                            net->x_recv_words[p], ncclUint32, (int)p, comm, stream);
            what = "ncclRecv";
        }
    }
    const ncclResult_t closed = R->GroupEnd();
    if (first != ncclSuccess) return fail(SNN_ERR_QUEUE, std::string(what) + ": " + R->GetErrorString(first));
    if (closed != ncclSuccess) return fail(SNN_ERR_QUEUE, std::string("ncclGroupEnd: ") + R->GetErrorString(closed));
    return SNN_OK;
}

// The mirror of a shard handle holds, of the neurons owned elsewhere, what past exchanges carried.  When the plan asks for a
// plane that was not on the wire (gap junctions or chemical synapses switched on between two runs) the owners' CURRENT values of
// the plan's planes travel once before the next step: refresh_pack -> the exchange -> refresh_unpack.  (The spike bits travel
// with them and restate the last step's flags and stamps: idempotent.)
uint32_t plan_plane_mask(const snn_network *net)
{
    uint32_t m = 0;
    for (uint32_t s = 0; s < net->x_planes; ++s) m |= 1u << net->x_plane_id[s];
    return m;
}
bool mirror_stale(const snn_network *net)
{
    return net->sharded && net->n_shards > 1 && (plan_plane_mask(net) & ~net->mirror_mask) != 0;
}
int refresh_pack(snn_network *net)
{
    if (!net->seg_n[0] || !net->seg_max[0]) return SNN_OK;
    hipLaunchKernelGGL(k_exchange_pack, dim3((net->seg_max[0] + 255) / 256, net->seg_n[0]), dim3(256), 0, net->stream, wire_args(net, 0));
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    net->send_bits_clean = false;
    return SNN_OK;
}
int refresh_unpack(snn_network *net)
{
    if (net->seg_n[1] && net->seg_max[1] && net->nn) {
        WireArgs a = wire_args(net, 1);
        a.clock = net->clock - 1;                                 // the step whose spike flags are restated
        hipLaunchKernelGGL(k_exchange_unpack, dim3((net->seg_max[1] + 255) / 256, net->seg_n[1]), dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    net->mirror_mask |= plan_plane_mask(net);
    return SNN_OK;
}

// Opens / closes the direct form of a library-driven run (snn_network_state.hpp).  Begin: the receive set the first step reads
// is filled from the mirror (what earlier exchanges left there), both sets of outgoing bitmaps are zeroed.  End: the arrivals
// of the last step go into the mirror and the current shadow, as the closing launch of an ordinary step would have done.
int direct_begin(snn_network *net)
{
    net->direct_run = false;
    net->peer_run = false;
    if (!net->halo_direct || !csr_fast_step(net) || net->n_shards < 2 || !net->csr_plan_direct) return SNN_OK;
    const bool peer = net->halo_peer && net->peer_capable && net->p2p_connected && net->p2p_recv[0];
    if (!net->direct_capable && !peer) return SNN_OK;        // (several planes on the wire: only the peer form gathers them itself)
    uint64_t so = 0;
    for (uint32_t p = 0; p < net->n_shards; ++p) so += net->x_send_words[p];
    HIP_TRY(hipMemsetAsync(net->halo_send_buf, 0, std::max<uint64_t>(so * 4, 256), net->stream), SNN_ERR_BUFFER_WRITE);
    if (net->halo_send_buf2) HIP_TRY(hipMemsetAsync(net->halo_send_buf2, 0, std::max<uint64_t>(so * 4, 256), net->stream), SNN_ERR_BUFFER_WRITE);
    net->hx_par = 0;
    net->stamp_pending = false;
    if (peer && net->p2p_epoch > 0x3FFFFF00u) return fail(SNN_ERR_BAD_STATE, "peer form: step tags exhausted (2^30 steps): rebuild the exchange plan");
    net->peer_run = peer;
    if (net->peer_run) {
        // the peer form: the set the first step reads -- values "produced by step epoch - 1" -- from the mirror, tagged for it
        if (net->recv_total) {
            hipLaunchKernelGGL(k_peer_prefill, dim3((net->recv_total + 255) / 256), dim3(256), 0, net->stream, wire_args(net, 1), net->recv_total,
                               net->seg_n[1], net->p2p_recv[(net->p2p_epoch + 1u) & 1u], net->p2p_epoch, net->na.nt_flags, net->n_pad);
            HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        }
    } else if (net->seg_n[1] && net->seg_max[1]) {
        WireArgs a = wire_args(net, 1, /*set=*/1);                // the set step 0 reads: hx_par ^ 1
        hipLaunchKernelGGL(k_exchange_pack, dim3((net->seg_max[1] + 255) / 256, net->seg_n[1]), dim3(256), 0, net->stream, a);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    net->direct_run = true;
    return SNN_OK;
}

int direct_end(snn_network *net)
{
    if (!net->direct_run) return SNN_OK;
    net->direct_run = false;
    const bool peer = net->peer_run;
    net->peer_run = false;
    net->send_bits_clean = false;                                 // an ordinary step clears set 0's bitmaps before it packs
    if (!net->stamp_pending) return SNN_OK;
    net->stamp_pending = false;
    if (!net->seg_n[1] || !net->seg_max[1] || !net->nn) return SNN_OK;
    if (peer) {
        // the last step's arrivals (granules of set (epoch - 1) % 2, tagged epoch) into the mirror and the current shadow
        StepCloseArgs c{};
        c.recv = wire_args(net, 1);
        c.recv.clock = net->clock - 1;
        c.xbuf2 = c.recv.xbuf2;
        c.recv_total = net->recv_total; c.recv_segments = net->seg_n[1];
        c.unpack_blocks = (net->recv_total + 255) / 256;
        c.recv64 = net->p2p_recv[(net->p2p_epoch + 1u) & 1u]; c.recv_tag = net->p2p_epoch;
        c.spin_limit = net->p2p_spin_limit; c.failed = PeerFailure{{net->p2p_failed, net->p2p_done_blocks + 1}};
        if (c.unpack_blocks) hipLaunchKernelGGL(k_step_close, dim3(c.unpack_blocks), dim3(256), 0, net->stream, c);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        return SNN_OK;
    }
    WireArgs a = wire_args(net, 1, net->hx_par ^ 1);              // the set the last step's exchange filled
    a.clock = net->clock - 1;
    hipLaunchKernelGGL(k_exchange_unpack, dim3((net->seg_max[1] + 255) / 256, net->seg_n[1]), dim3(256), 0, net->stream, a);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}

// peer form: the device tables of a connected handle -- per pack entry the peer's granule (both sets) and the peer, and the
// neighbours' done-counter words this handle signals (every shard it RECEIVES from stores into this handle's sets and waits
// for this handle's counter in its own memory)
int p2p_build_tables(snn_network *net)
{
    const uint32_t G = net->n_shards, me = net->shard_index;
    auto local_row = [&](uint32_t g) { return net->block_mode ? net->local_row_host[g] : g - net->q0; };
    hvec<uint32_t> pack_ptr(net->n_loc + 1, 0);
    for (uint32_t p = 0; p < G; ++p)
        for (uint32_t g : net->halo_send[p]) pack_ptr[local_row(g) + 1] += 1;
    for (uint32_t q = 0; q < net->n_loc; ++q) pack_ptr[q + 1] += pack_ptr[q];
    const size_t entries = pack_ptr[net->n_loc];
    hvec<unsigned long long> dst[2] = {hvec<unsigned long long>(entries), hvec<unsigned long long>(entries)};
    hvec<uint32_t> peer(entries);
    hvec<uint32_t> fill(pack_ptr.begin(), pack_ptr.end() - 1);
    for (uint32_t p = 0; p < G; ++p) {
        const auto &sl = net->halo_send[p];
        if (sl.empty()) continue;
        if (!net->p2p_peers[p].set) return fail(SNN_ERR_BAD_STATE, "peer form: shard " + std::to_string(p) + " reads this shard but is not connected");
        for (uint32_t i = 0; i < sl.size(); ++i) {
            const uint32_t e = fill[local_row(sl[i])]++;            // the order ensure_exchange_plan gave the pack table
            for (int k = 0; k < 2; ++k) dst[k][e] = net->p2p_peers[p].recv[k] + 8ull * (net->p2p_peers[p].recv_offset + (uint64_t)i * net->x_planes);
            peer[e] = p;
        }
    }
    hvec<unsigned long long> signal;
    for (uint32_t p = 0; p < G; ++p) {
        if (p == me || net->halo_need[p].empty()) continue;
        if (!net->p2p_peers[p].set) return fail(SNN_ERR_BAD_STATE, "peer form: shard " + std::to_string(p) + " is read by this shard but is not connected");
        signal.push_back(net->p2p_peers[p].flags + 4ull * me);
    }
    for (int k = 0; k < 2; ++k) TRY(upload_table(net, reinterpret_cast<unsigned long long **>(&net->p2p_dst_dev[k]), dst[k]));
    TRY(upload_table(net, &net->p2p_peer_dev, peer));
    TRY(upload_table(net, reinterpret_cast<unsigned long long **>(&net->p2p_signal_dev), signal));
    net->p2p_n_signal = (uint32_t)signal.size();
    net->p2p_connected = true;
    net->x_agreed = false;                         // whether a run takes the peer form is part of what the ranks agree on
    // Receive sets of earlier plans stay allocated until the handle is destroyed (p2p_release(final)): THIS handle's commit says
    // nothing about its peers -- one that has not run its own connect / commit yet still holds tables or IPC mappings into them
    // and would store its next granules or done counters into freed, possibly reallocated memory.  A plan rebuild is rare and a
    // receive set small (8 B per halo neuron), so nothing is gained by freeing earlier.
    return SNN_OK;
}

int ensure_comm_objects(snn_network *net)
{
    if (!net->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&net->comm_stream, hipStreamNonBlocking), SNN_ERR_QUEUE);
    if (!net->ev_packed) HIP_TRY(hipEventCreateWithFlags(&net->ev_packed, hipEventDisableTiming), SNN_ERR_QUEUE);
    if (!net->ev_exchanged) HIP_TRY(hipEventCreateWithFlags(&net->ev_exchanged, hipEventDisableTiming), SNN_ERR_QUEUE);
    return SNN_OK;
}

} // namespace
