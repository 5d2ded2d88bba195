// C ABI of the MI355X-native spiking-lattice stepper (include/snn_amd.h): the extern "C" entry points.  The one
// translation unit of libsnn_amd.so: snn_network_state.hpp holds the handle (index space, device allocations,
// attribute registry), snn_network_step.hpp the kernel launches and the step loop, snn_kernels_*.hpp the kernels.
#include "snn_network_state.hpp"
#include "snn_network_step.hpp"
#include "snn_network_exchange.hpp"

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int snn_abi_version(void) ABI_TRY { return SNN_ABI_VERSION; } ABI_CATCH
const char *snn_custom_model(void) ABI_TRY { return custom::TYPE_NAME; } ABI_CATCH_PTR
const char *snn_custom_spike_train(void) ABI_TRY { return custom_st::TYPE_NAME; } ABI_CATCH_PTR
const char *snn_custom_refractoriness(void) ABI_TRY { return custom_refr::TYPE_NAME; } ABI_CATCH_PTR
const char *snn_custom_neurotransmitter_kinetics(void) ABI_TRY { return custom_nt::TYPE_NAME; } ABI_CATCH_PTR
const char *snn_custom_receptor_kinetics(void) ABI_TRY { return custom_rc::TYPE_NAME; } ABI_CATCH_PTR
const char *snn_custom_receptors(void) ABI_TRY { return custom_receptors::TYPE_NAME; } ABI_CATCH_PTR
const char *snn_last_error(void) ABI_TRY { return g_last_error.c_str(); } ABI_CATCH_PTR

int snn_network_create(int device, int neuron_model, int nt_kinetics, int receptor_kinetics,
                       int spike_train_model, snn_network_t **out) ABI_TRY
{
    if (!out) return fail(SNN_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    const bool custom_ok = SNN_HAVE_CUSTOM_NEURON && neuron_model == SNN_MODEL_CUSTOM;
    const bool custom_st_ok = SNN_HAVE_CUSTOM_SPIKE_TRAIN && spike_train_model == SNN_ST_CUSTOM;
    const bool custom_nt_ok = SNN_HAVE_CUSTOM_NT && nt_kinetics == SNN_NT_CUSTOM;
    const bool custom_rc_ok = SNN_HAVE_CUSTOM_RC && receptor_kinetics == SNN_RC_CUSTOM;
    if (((neuron_model < 0 || neuron_model > 8) && !custom_ok) || ((nt_kinetics < 0 || nt_kinetics > 3) && !custom_nt_ok) ||
        ((receptor_kinetics < 0 || receptor_kinetics > 2) && !custom_rc_ok) ||
        ((spike_train_model < 0 || spike_train_model > 4) && !custom_st_ok))
        return fail(SNN_ERR_BAD_ARG, "unknown model / kinetics selector");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SNN_ERR_GET_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(SNN_ERR_GET_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    snn_network *net = new snn_network();
    net->device = device;
    if (const char *e = getenv("SNN_AMD_FUSED_STEP")) net->fused_step = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_DENSE_CLOSE")) net->dense_close = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_DENSE_CLOSE_MAX_CHUNKS")) net->dense_close_max_chunks = (uint32_t)strtoul(e, nullptr, 10);
    if (const char *e = getenv("SNN_AMD_PINNED_COPIES")) net->pinned_copies = (e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
    if (const char *e = getenv("SNN_AMD_CSR_XCD_BANDS")) net->csr_xcd_bands = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_CSR_IMAGE")) net->csr_image = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_RESIDENT_QUARTERS")) net->resident_quarters = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_HALO_DIRECT")) net->halo_direct = (e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1;
    if (const char *e = getenv("SNN_AMD_UPDATE_PACKS")) net->update_packs = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_UPDATE_ALL_PLANES")) net->update_all_planes = (e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1;
    if (const char *e = getenv("SNN_AMD_CELLS_IN_STEP")) net->cells_in_step = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_DEFER_RSTDP")) net->defer_rstdp = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_DEFER_STDP")) net->defer_stdp = (e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1;
    if (const char *e = getenv("SNN_AMD_UNIFORM_PARAMS")) net->uniform_params = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_PERSISTENT_RUN")) net->persistent_run = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_PERSISTENT_CHEM")) net->persistent_chem = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_PERSISTENT_STDP")) net->persistent_stdp = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_HALO_PEER")) net->halo_peer = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_STDP_COLUMNS_FORM")) net->stdp_columns_form = (e[0] == '1') ? 1 : 0;
    if (const char *e = getenv("SNN_AMD_STDP_SMALL")) net->stdp_small = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_VERIFY")) net->verify = (e[0] != '0');
    if (const char *e = getenv("SNN_AMD_INPUT_SHAPE")) net->force_shape = (e[0] == '1') ? 1 : ((e[0] == '2') ? 2 : 0);
    net->model = neuron_model; net->nt_kind = nt_kinetics; net->rc_kind = receptor_kinetics;
    net->st_kind = spike_train_model;
    if (hipStreamCreateWithFlags(&net->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete net;
        return fail(SNN_ERR_QUEUE, "hipStreamCreate failed");
    }
    net->stream = net->own_stream;
    *out = net;
    return SNN_OK;
}
ABI_CATCH

int snn_network_destroy(snn_network_t *net) ABI_TRY
{
    if (!net) return SNN_OK;
    (void)hipSetDevice(net->device);
    if (net->stream) (void)hipStreamSynchronize(net->stream);
    for (void *p : net->allocs) (void)hipFree(p);
    for (void *p : {(void *)net->csr_ptr, (void *)net->csr_pre, (void *)net->csr_post, (void *)net->csr_t_ptr,
                    (void *)net->csr_t_edge, (void *)net->csr_w, (void *)net->csr_row_len, (void *)net->csr_edge_slot,
                    (void *)net->csr_plan, (void *)net->csr_img_hdr, (void *)net->csr_plan_win, (void *)net->csr_img_rec,
                    (void *)net->csr_img_hdr_direct, (void *)net->csr_plan_win_direct, (void *)net->csr_img_rec_direct})
        if (p) (void)hipFree(p);
    if (net->vhist) (void)hipFree(net->vhist);
    if (net->st_vhist) (void)hipFree(net->st_vhist);
    if (net->raster) (void)hipFree(net->raster);
    if (net->preset_times_dev) (void)hipFree(net->preset_times_dev);
    if (net->trace) (void)hipFree(net->trace);
    if (net->pending) (void)hipFree(net->pending);
    if (net->edge_counter) (void)hipFree(net->edge_counter);
    if (net->cross_bad) (void)hipFree(net->cross_bad);
    if (net->conn_kind_dev) (void)hipFree(net->conn_kind_dev);
    if (net->run_failed) (void)hipHostFree(net->run_failed);
    if (net->copy_stage) (void)hipHostFree(net->copy_stage);
    if (net->st_clock_pinned) (void)hipHostFree(net->st_clock_pinned);
    if (net->verify_report) (void)hipFree(net->verify_report);
    if (net->verify_big) (void)hipFree(net->verify_big);
    if (net->verify_third) (void)hipFree(net->verify_third);
    for (float *b : net->whist) if (b) (void)hipFree(b);
    if (net->summ_avg) (void)hipFree(net->summ_avg);
    if (net->summ_eeg) (void)hipFree(net->summ_eeg);
    for (auto &e : net->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : net->ev_pool_pl) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (net->cell_list_dev) (void)hipFree(net->cell_list_dev);
    for (void *p : {(void *)net->halo_send_buf2, (void *)net->halo_recv_buf2, (void *)net->csr_plan_direct, (void *)net->halo_word_dev})
        if (p) (void)hipFree(p);
    for (void *p : {(void *)net->halo_send_buf, (void *)net->halo_recv_buf, (void *)net->halo_send_idx, (void *)net->halo_recv_idx,
                    (void *)net->seg_count_dev[0], (void *)net->seg_count_dev[1], (void *)net->seg_first_dev[0],
                    (void *)net->seg_first_dev[1], (void *)net->seg_offset_dev[0], (void *)net->seg_offset_dev[1],
                    (void *)net->seg_loff_dev[0], (void *)net->seg_loff_dev[1], (void *)net->csr_border_dev,
                    (void *)net->csr_interior_dev, (void *)net->pack_ptr_dev, (void *)net->pack_segoff_dev,
                    (void *)net->pack_count_dev, (void *)net->pack_index_dev})
        if (p) (void)hipFree(p);
    (void)p2p_release(net, /*final=*/true);
    if (net->p2p_failed) (void)hipHostFree(net->p2p_failed);
    if (net->agree_words_dev) (void)hipFree(net->agree_words_dev);
    if (net->comm_stream) { (void)hipStreamSynchronize(net->comm_stream); (void)hipStreamDestroy(net->comm_stream); }
    if (net->ev_packed) (void)hipEventDestroy(net->ev_packed);
    if (net->ev_exchanged) (void)hipEventDestroy(net->ev_exchanged);
    if (net->own_stream) (void)hipStreamDestroy(net->own_stream);
    delete net;
    return SNN_OK;
}
ABI_CATCH

static int add_lattice_impl(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols, bool st)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (net->finalized) return fail(SNN_ERR_BAD_STATE, "lattices cannot be added after finalize");
    if (find_lattice(net, id))   // LatticeNetworkError::GraphIDAlreadyPresent, neuron/mod.rs:1669-1671
        return fail(SNN_ERR_BAD_ARG, "lattice id " + std::to_string(id) + " already present");
    if (st && net->st_kind == SNN_ST_NONE) return fail(SNN_ERR_BAD_STATE, "network was created without a spike-train model");
    if ((uint64_t)rows * cols > 0x7FFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "lattice too large");
    LatticeInfo l{id, rows, cols, 0, rows * cols, 0, st};
    (st ? net->st_lattices : net->lattices).push_back(l);
    return SNN_OK;
}

int snn_network_add_lattice(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols) ABI_TRY
{
    return add_lattice_impl(net, id, rows, cols, false);
}
ABI_CATCH
int snn_network_add_spike_train_lattice(snn_network_t *net, uint32_t id, uint32_t rows, uint32_t cols) ABI_TRY
{
    return add_lattice_impl(net, id, rows, cols, true);
}
ABI_CATCH

// kind: 0 whole population, 1 contiguous shard [post_begin, post_end), 2 by lattice (slab shard_index of every lattice)
static int finalize_impl(snn_network_t *net, int kind, uint32_t post_begin, uint32_t post_end, uint32_t n_shards,
                         uint32_t stride, uint32_t shard_index)
{
    const bool whole = kind == 0;
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (net->finalized) return fail(SNN_ERR_BAD_STATE, "already finalized");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    auto by_id = [](const LatticeInfo &a, const LatticeInfo &b) { return a.id < b.id; };
    std::sort(net->lattices.begin(), net->lattices.end(), by_id);
    std::sort(net->st_lattices.begin(), net->st_lattices.end(), by_id);
    uint64_t off = 0;
    uint32_t slot = 0;
    for (auto &l : net->lattices) { l.first = (uint32_t)off; l.slot = slot++; off += l.count; }
    net->nn = (uint32_t)off;
    slot = 0;
    for (auto &l : net->st_lattices) { l.first = (uint32_t)off; l.slot = slot++; off += l.count; }
    if (off > 0x7FFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "network too large");
    net->n_tot = (uint32_t)off;
    net->nc = net->n_tot - net->nn;
    net->n_pad = std::max<uint32_t>(256, round_up(net->nn, 256));
    net->c_pad = std::max<uint32_t>(256, round_up(net->nc, 256));
    if (whole) {
        net->q0 = 0; net->q1 = net->nn;
    } else if (kind == 1) {
        if (post_begin > post_end || post_end > net->nn) return fail(SNN_ERR_DIM_MISMATCH, "shard range outside the population");
        if (stride == 0 || stride % 64 != 0 || n_shards == 0 || (uint64_t)stride * n_shards < net->nn)
            return fail(SNN_ERR_BAD_ARG, "shard stride must be a multiple of 64 covering the population");
        net->q0 = post_begin; net->q1 = post_end;
        net->sharded = true;
        net->n_shards = n_shards; net->shard_stride = stride; net->shard_index = shard_index;
        net->n_pad = std::max<uint32_t>(net->n_pad, round_up(stride * n_shards, 256));
    } else {
        net->q0 = 0; net->q1 = 0;
        net->sharded = true;
        net->n_shards = n_shards; net->shard_stride = 0; net->shard_index = shard_index;
    }
    net->xl = XLayout{net->n_pad};
    net->n_loc = net->q1 - net->q0;
    net->ranges.clear();
    if (kind != 2) {
        if (net->q1 > net->q0) net->ranges.emplace_back(net->q0, net->q1);
        net->n_owned = net->n_loc;
        net->rowmap = RowMap{net->q0, nullptr, nullptr, nullptr};
    } else {
        // slab `shard_index` of every neuron lattice; local rows = the global 64-blocks holding an owned neuron
        net->block_mode = true;
        net->lattice_slab.assign(net->lattices.size(), 64);
        net->local_row_host.assign(net->n_pad, 0xFFFFFFFFu);
        hvec<uint32_t> blocks;
        hvec<unsigned long long> masks;
        net->n_owned = 0;
        for (const auto &l : net->lattices) {
            const uint32_t slab = std::max<uint32_t>(64, round_up((l.count + n_shards - 1) / n_shards, 64));
            net->lattice_slab[l.slot] = slab;
            const uint64_t b = std::min<uint64_t>(l.count, (uint64_t)shard_index * slab), e = std::min<uint64_t>(l.count, b + slab);
            if (e <= b) continue;
            const uint32_t gb = l.first + (uint32_t)b, ge = l.first + (uint32_t)e;
            if (!net->ranges.empty() && net->ranges.back().second == gb) net->ranges.back().second = ge;
            else net->ranges.emplace_back(gb, ge);
            for (uint32_t q = gb; q < ge; ++q) {
                if (blocks.empty() || blocks.back() != (q >> 6)) { blocks.push_back(q >> 6); masks.push_back(0ull); }
                masks.back() |= 1ull << (q & 63u);
                net->local_row_host[q] = (uint32_t)(blocks.size() - 1) * 64u + (q & 63u);
                net->owned_local_host.push_back(net->local_row_host[q]);
                ++net->n_owned;
            }
        }
        if (blocks.empty()) { blocks.push_back(0); masks.push_back(0ull); }
        net->q0 = 0; net->q1 = 0;
        net->n_loc = (uint32_t)blocks.size() * 64u;
        HIP_TRY(snn_malloc(&net->own_block_dev, blocks.size() * 4), SNN_ERR_BUFFER_CREATE);
        HIP_TRY(snn_malloc(&net->own_mask_dev, masks.size() * 8), SNN_ERR_BUFFER_CREATE);
        HIP_TRY(snn_malloc(&net->local_row_dev, (size_t)net->n_pad * 4), SNN_ERR_BUFFER_CREATE);
        HIP_TRY(copy_sync(net, net->own_block_dev, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(copy_sync(net, net->own_mask_dev, masks.data(), masks.size() * 8, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(copy_sync(net, net->local_row_dev, net->local_row_host.data(), (size_t)net->n_pad * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        net->allocs.push_back(net->own_block_dev);
        net->allocs.push_back(net->own_mask_dev);
        net->allocs.push_back(net->local_row_dev);
        net->rowmap = RowMap{0, net->own_block_dev, net->own_mask_dev, net->local_row_dev};
    }
    net->ld = std::max<uint32_t>(64, round_up(net->n_loc, 64));
    // A row stride that is a multiple of 4 KiB puts the same columns of consecutive rows on the same HBM
    // channels; 256 B of padding per row de-aligns them (measured at 256x256, same process: +2 % bandwidth).
    if (net->ld % 1024 == 0) net->ld += 64;
    net->n_chunks = (net->n_tot + CHUNK - 1) / CHUNK;
    int rc = build_state(net);
    if (rc) return rc;
    rc = choose_matrix_placement(net);
    if (rc) return rc;
    net->finalized = true;
    return SNN_OK;
}

int snn_network_finalize(snn_network_t *net) ABI_TRY { return finalize_impl(net, 0, 0, 0, 1, 0, 0); } ABI_CATCH

int snn_network_finalize_shard_by_lattice(snn_network_t *net, uint32_t shard_index, uint32_t n_shards) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (n_shards == 0 || shard_index >= n_shards) return fail(SNN_ERR_BAD_ARG, "shard_index must be < n_shards");
    if (!net->csr) return fail(SNN_ERR_BAD_STATE, "sharding by lattice needs a sparse handle: call snn_network_use_csr first");
    return finalize_impl(net, 2, 0, 0, n_shards, 64, shard_index);
}
ABI_CATCH

int snn_shard_ranges(const snn_network_t *net, uint32_t *begin, uint32_t *end, uint32_t capacity, uint32_t *count) ABI_TRY
{
    if (!net || !count) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    *count = (uint32_t)net->ranges.size();
    if (begin && end && capacity >= net->ranges.size())
        for (size_t i = 0; i < net->ranges.size(); ++i) { begin[i] = net->ranges[i].first; end[i] = net->ranges[i].second; }
    return SNN_OK;
}
ABI_CATCH

int snn_network_finalize_shard(snn_network_t *net, uint32_t shard_index, uint32_t n_shards) ABI_TRY
{
    // equal slots: stride = ceil(n_neurons / n_shards) rounded up to a wavefront (64 neurons); shard r owns
    // neurons [r*stride, min(n, (r+1)*stride)) -- trailing shards may be short or empty
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (n_shards == 0 || shard_index >= n_shards) return fail(SNN_ERR_BAD_ARG, "shard_index must be < n_shards");
    uint64_t nn = 0;
    for (const auto &l : net->lattices) nn += l.count;
    const uint32_t stride = std::max<uint32_t>(64, round_up((uint32_t)((nn + n_shards - 1) / n_shards), 64));
    const uint32_t begin = (uint32_t)std::min<uint64_t>(nn, (uint64_t)shard_index * stride);
    const uint32_t end = (uint32_t)std::min<uint64_t>(nn, (uint64_t)begin + stride);
    return finalize_impl(net, 1, begin, end, n_shards, stride, shard_index);
}
ABI_CATCH

int snn_network_sizes(const snn_network_t *net, uint32_t *n_neurons, uint32_t *n_cells, uint32_t *post_begin,
                      uint32_t *post_end) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (n_neurons) *n_neurons = net->nn;
    if (n_cells) *n_cells = net->nc;
    if (post_begin) *post_begin = net->q0;
    if (post_end) *post_end = net->q1;
    return SNN_OK;
}
ABI_CATCH

int snn_network_lattice_range(const snn_network_t *net, uint32_t id, uint32_t *first, uint32_t *count) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l) return fail(SNN_ERR_BAD_ARG, "unknown lattice id");
    if (first) *first = l->first;
    if (count) *count = l->count;
    return SNN_OK;
}
ABI_CATCH

int snn_set_attr_f32(snn_network_t *net, uint32_t id, const char *name, const float *src, size_t count) ABI_TRY
{ return attr_io(net, id, name, T_F32, const_cast<float *>(src), count, true); }
ABI_CATCH
int snn_get_attr_f32(snn_network_t *net, uint32_t id, const char *name, float *dst, size_t count) ABI_TRY
{ return attr_io(net, id, name, T_F32, dst, count, false); }
ABI_CATCH
int snn_set_attr_u32(snn_network_t *net, uint32_t id, const char *name, const uint32_t *src, size_t count) ABI_TRY
{ return attr_io(net, id, name, T_U32, const_cast<uint32_t *>(src), count, true); }
ABI_CATCH
int snn_get_attr_u32(snn_network_t *net, uint32_t id, const char *name, uint32_t *dst, size_t count) ABI_TRY
{ return attr_io(net, id, name, T_U32, dst, count, false); }
ABI_CATCH
int snn_set_attr_i32(snn_network_t *net, uint32_t id, const char *name, const int32_t *src, size_t count) ABI_TRY
{ return attr_io(net, id, name, T_I32, const_cast<int32_t *>(src), count, true); }
ABI_CATCH
int snn_get_attr_i32(snn_network_t *net, uint32_t id, const char *name, int32_t *dst, size_t count) ABI_TRY
{ return attr_io(net, id, name, T_I32, dst, count, false); }
ABI_CATCH

int snn_set_graph_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const float *weights,
                       const uint32_t *connections) ABI_TRY
{
    if (net) net->cross_checked = false;
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    return graph_rows_io(net, pre_begin, pre_count, const_cast<float *>(weights),
                         const_cast<uint32_t *>(connections), net->nn, true);
}
ABI_CATCH
int snn_get_graph_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *weights, uint32_t *connections) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    return graph_rows_io(net, pre_begin, pre_count, weights, connections, net->nn, false);
}
ABI_CATCH
int snn_set_graph_dense(snn_network_t *net, const float *weights, const uint32_t *connections, size_t n_tot) ABI_TRY
{
    if (net) net->cross_checked = false;
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (n_tot != net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "graph size does not match the network");
    return graph_rows_io(net, 0, net->n_tot, const_cast<float *>(weights), const_cast<uint32_t *>(connections),
                         net->n_tot, true);
}
ABI_CATCH
int snn_get_graph_dense(snn_network_t *net, float *weights, uint32_t *connections, size_t n_tot) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (n_tot != net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "graph size does not match the network");
    return graph_rows_io(net, 0, net->n_tot, weights, connections, net->n_tot, false);
}
ABI_CATCH

int snn_fill_graph_synthetic(snn_network_t *net, uint64_t seed, float lo, float hi, int with_diagonal) ABI_TRY
{
    if (net) net->cross_checked = false;
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) return fail(SNN_ERR_BAD_STATE, "the synthetic dense graph needs a dense handle");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));                     // a deferred reward-modulated update belongs to the OLD weights: apply it first
    if (net->trace)                        // traces of the replaced edges do not carry over
        HIP_TRY(hipMemsetAsync(net->trace, 0, std::max<size_t>(wcount(net->n_tot, net->ld), 64) * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    if (net->n_tot && net->ld) {
        const unsigned gy = std::min<uint32_t>(net->n_tot, 4096);
        hipLaunchKernelGGL(k_graph_synthetic, dim3((net->ld + 255) / 256, gy), dim3(256), 0, net->stream, net->W,
                           net->ld, net->n_loc, net->q0, net->nn, net->n_tot, seed, lo, hi, with_diagonal);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    }
    net->counts_dirty = true;
    return SNN_OK;
}
ABI_CATCH

int snn_network_use_csr(snn_network_t *net, int enable) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (net->finalized) return fail(SNN_ERR_BAD_STATE, "the graph form is fixed at finalize");
    net->csr = enable != 0;
    return SNN_OK;
}
ABI_CATCH

namespace { int ensure_traces(snn_network *net); int ensure_pending(snn_network *net); }

static int set_graph_csr_impl(snn_network_t *net, const uint64_t *row_ptr, const uint32_t *pre_index, const float *weights,
                              uint64_t nnz)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a dense graph: call snn_network_use_csr before finalize");
    if (!row_ptr || (nnz && (!pre_index || !weights))) return fail(SNN_ERR_BAD_ARG, "null graph pointer");
    if (nnz >= 0xFFFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "more than 2^32-1 stored synapses per handle");
    if ((uint64_t)net->nn + net->nc >= PLAN_CODE) return fail(SNN_ERR_DIM_MISMATCH, "the gather plan addresses fewer than 2^31-1 presynaptic rows");
    // the caller's rows are the OWNED neurons in ascending order; a range-set shard places them on its local rows
    // (global 64-blocks, holes in between keep length 0)
    const uint32_t n_loc = net->n_loc, n_rows = net->n_owned;
    auto local = [&](uint32_t k) { return net->block_mode ? net->owned_local_host[k] : k; };
    if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return fail(SNN_ERR_DIM_MISMATCH, "row_ptr must run from 0 to nnz");
    const uint32_t n_slices = (n_loc + 63) / 64;
    hvec<uint32_t> slice_ptr((size_t)n_slices + 1, 0), row_len((size_t)n_slices * 64, 0), post(nnz),
        t_ptr((size_t)net->n_tot + 1, 0), t_edge(nnz), edge_slot(nnz);
    for (uint32_t k = 0; k < n_rows; ++k) {
        const uint32_t q = local(k);
        if (row_ptr[k + 1] < row_ptr[k]) return fail(SNN_ERR_DIM_MISMATCH, "row_ptr is not monotone");
        row_len[q] = (uint32_t)(row_ptr[k + 1] - row_ptr[k]);
        for (uint64_t e = row_ptr[k]; e < row_ptr[k + 1]; ++e) {
            if (pre_index[e] >= net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "presynaptic index out of range");
            if (e > row_ptr[k] && pre_index[e] <= pre_index[e - 1])
                return fail(SNN_ERR_BAD_ARG, "presynaptic indices of a row must be strictly ascending");
            if (weights[e] != weights[e])          // (the dense form cannot hold such an edge: the two forms refuse alike)
                return fail(SNN_ERR_BAD_ARG, "stored edge " + std::to_string(e) + " (pre " + std::to_string(pre_index[e]) + ") carries a NaN weight");
            post[e] = q;
            ++t_ptr[pre_index[e] + 1];
        }
    }
    // SELL-64: every slice (64 rows) is padded to its longest row
    uint64_t entries = 0;
    for (uint32_t sl = 0; sl < n_slices; ++sl) {
        uint32_t width = 0;
        for (uint32_t r = sl * 64; r < sl * 64 + 64; ++r) width = std::max(width, row_len[r]);
        slice_ptr[sl] = (uint32_t)entries;
        entries += (uint64_t)width * 64;
        if (entries >= 0xFFFFFFFFull) return fail(SNN_ERR_DIM_MISMATCH, "sparse graph too large for 32-bit slots");
    }
    slice_ptr[n_slices] = (uint32_t)entries;
    hvec<uint32_t> sell_pre(entries, SELL_PAD);
    hvec<float> sell_w(entries, 0.0f);
    for (uint32_t k = 0; k < n_rows; ++k) {
        const uint32_t q = local(k);
        const uint32_t base = slice_ptr[q >> 6] + (q & 63u);
        for (uint64_t e = row_ptr[k]; e < row_ptr[k + 1]; ++e) {
            const uint32_t slot = base + (uint32_t)(e - row_ptr[k]) * 64;
            sell_pre[slot] = pre_index[e];
            sell_w[slot] = weights[e];
            edge_slot[e] = slot;
        }
    }
    for (size_t p = 0; p < net->n_tot; ++p) t_ptr[p + 1] += t_ptr[p];
    {
        hvec<uint32_t> fill(t_ptr.begin(), t_ptr.end() - 1);
        for (uint64_t e = 0; e < nnz; ++e) t_edge[fill[pre_index[e]]++] = (uint32_t)e;
    }
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    for (void **p : {(void **)&net->csr_ptr, (void **)&net->csr_pre, (void **)&net->csr_post, (void **)&net->csr_t_ptr,
                     (void **)&net->csr_t_edge, (void **)&net->csr_w, (void **)&net->csr_row_len,
                     (void **)&net->csr_edge_slot, (void **)&net->csr_plan, (void **)&net->csr_img_hdr, (void **)&net->csr_plan_win,
                     (void **)&net->csr_img_rec}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        HIP_TRY(snn_malloc(dst, std::max<size_t>(bytes, 256)), SNN_ERR_BUFFER_CREATE);
        if (bytes) HIP_TRY(copy_sync(net, *dst, src, bytes, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
        return SNN_OK;
    };
    TRY(up((void **)&net->csr_ptr, slice_ptr.data(), slice_ptr.size() * 4));
    TRY(up((void **)&net->csr_pre, sell_pre.data(), entries * 4));
    TRY(up((void **)&net->csr_w, sell_w.data(), entries * 4));
    TRY(up((void **)&net->csr_row_len, row_len.data(), row_len.size() * 4));
    TRY(up((void **)&net->csr_edge_slot, edge_slot.data(), nnz * 4));
    TRY(up((void **)&net->csr_post, post.data(), nnz * 4));
    TRY(up((void **)&net->csr_t_ptr, t_ptr.data(), t_ptr.size() * 4));
    TRY(up((void **)&net->csr_t_edge, t_edge.data(), nnz * 4));
    HIP_TRY(snn_malloc(&net->csr_plan, std::max<size_t>(entries * 4, 256)), SNN_ERR_BUFFER_CREATE);
    {
        // the step image's host-built half: slice headers with the window pieces, and the plan words that go with them
        hvec<uint32_t> img_hdr, plan_win;
        build_step_image_plan(net, slice_ptr, sell_pre, n_slices, img_hdr, plan_win, net->img_records, net->img_staged_slices);
        TRY(up((void **)&net->csr_img_hdr, img_hdr.data(), img_hdr.size() * 4));
        TRY(up((void **)&net->csr_plan_win, plan_win.data(), plan_win.size() * 4));
        HIP_TRY(snn_malloc(&net->csr_img_rec, (size_t)net->img_records * 16 + 4096), SNN_ERR_BUFFER_CREATE);    // (+ a wavefront's load of slack)
        net->img_stale = true;
    }
    if (net->trace) { (void)hipFree(net->trace); net->trace = nullptr; }      // traces belong to the replaced edges
    for (float **m : {&net->pending, &net->edge_counter})                     // ... and so do dw and the counters of its connections
        if (*m) { (void)hipFree(*m); *m = nullptr; }
    net->cross_checked = false;
    net->nnz = nnz;
    net->sell_entries = entries;
    if (n_slices) {
        hipLaunchKernelGGL(k_csr_plan, dim3((n_slices * 64 + 255) / 256), dim3(256), 0, net->stream, csr_graph(net),
                           net->csr_plan, (const uint32_t *)nullptr, net->nn, PLAN_CODE);
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    }
    net->edge_slot_host.swap(edge_slot);
    if (net->sharded && net->n_shards > 1) { net->sell_pre_host.swap(sell_pre); net->slice_ptr_host.swap(slice_ptr); }
    else { net->sell_pre_host.clear(); net->slice_ptr_host.clear(); }
    net->img_stale_direct = true;
    net->counts_dirty = true;
    halo_needs_from_rows(net, pre_index, nnz);      // the new rows decide what is read from the other shards
    if (net->sharded && net->n_shards > 1 && net->nc) {
        // ... and which spike-train cells this rank reads at all
        hvec<uint8_t> seen(net->nc, 0);
        for (uint64_t e = 0; e < nnz; ++e)
            if (pre_index[e] >= net->nn) seen[pre_index[e] - net->nn] = 1;
        net->cell_list_host.clear();
        for (uint32_t s = 0; s < net->nc; ++s)
            if (seen[s]) net->cell_list_host.push_back(s);
        TRY(upload_table(net, &net->cell_list_dev, net->cell_list_host));
        net->n_cells_listed = (uint32_t)net->cell_list_host.size();
        net->view_dirty = true;
    }
    // a reward-modulated handle keeps modulating: zeroed traces for the new edges (the old ones went with their edges)
    if (net->any_modulation || net->any_conn_kind) TRY(ensure_traces(net));
    if (net->any_conn_kind) TRY(ensure_pending(net));
    return SNN_OK;
}

int snn_set_graph_csr(snn_network_t *net, const uint64_t *row_ptr, const uint32_t *pre_index, const float *weights,
                      uint64_t nnz) ABI_TRY
{
    try {
        return set_graph_csr_impl(net, row_ptr, pre_index, weights, nnz);
    } catch (const std::bad_alloc &) {       // the host-side index arrays are sized by nnz: never let it cross the C ABI
        return fail(SNN_ERR_BUFFER_CREATE, "out of host memory while building the sparse graph");
    }
}
ABI_CATCH

int snn_get_graph_csr(snn_network_t *net, float *weights, uint64_t nnz) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a dense graph");
    if (nnz != net->nnz) return fail(SNN_ERR_DIM_MISMATCH, "nnz does not match the stored graph");
    if (nnz == 0) return SNN_OK;
    if (!weights) return fail(SNN_ERR_BAD_ARG, "weights is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    hvec<float> sell((size_t)net->sell_entries);
    HIP_TRY(copy_sync(net, sell.data(), net->csr_w, sell.size() * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    for (uint64_t e = 0; e < nnz; ++e) weights[e] = sell[net->edge_slot_host[e]];
    return SNN_OK;
}
ABI_CATCH

int snn_set_synapses(snn_network_t *net, int electrical_synapse, int chemical_synapse) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    net->electrical = electrical_synapse ? 1 : 0;
    net->chemical = chemical_synapse ? 1 : 0;
    net->x_dirty = true;                 // which planes travel between shards follows the synapse kinds
    return SNN_OK;
}
ABI_CATCH

int snn_set_plasticity(snn_network_t *net, uint32_t id, float a_plus, float a_minus, float tau_plus,
                       float tau_minus, float dt, int do_plasticity) ABI_TRY
{
    if (net) net->cross_checked = false;
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "plasticity belongs to neuron lattices");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    float *s = &net->stdp_host[(size_t)l->slot * PL_STRIDE];
    s[0] = a_plus; s[1] = a_minus; s[2] = tau_plus; s[3] = tau_minus; s[4] = dt; s[5] = 0.0f;
    // (a reward-modulated lattice keeps the parameters -- a partner's visit may borrow them -- but has no rule of its own)
    net->plast_host[l->slot] = (do_plasticity && !(l->slot < net->rm_on_host.size() && (net->rm_on_host[l->slot] & RM_IS_MODULATED))) ? 1u : 0u;
    net->any_plasticity = false;
    for (uint32_t p : net->plast_host) net->any_plasticity |= (p != 0);
    TRY(end_run(net));
    HIP_TRY(copy_sync(net, net->stdp_dev, net->stdp_host.data(), net->stdp_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(copy_sync(net, net->plast_dev, net->plast_host.data(), net->plast_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    return SNN_OK;
}
ABI_CATCH

int snn_set_bcm(snn_network_t *net, uint32_t id, float decay, float average_scalar, float dt, int do_plasticity) ABI_TRY
{
    if (net) net->cross_checked = false;
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "plasticity belongs to neuron lattices");
    if (net->model != SNN_MODEL_BCM_IZHIKEVICH)
        return fail(SNN_ERR_BAD_STATE, "the BCM rule needs neurons with BCMActivity (SNN_MODEL_BCM_IZHIKEVICH)");
    if (net->sharded) return fail(SNN_ERR_BAD_STATE, "the BCM rule is not available on shard handles (activities are not exchanged)");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    float *s = &net->stdp_host[(size_t)l->slot * PL_STRIDE];
    s[4] = dt; s[5] = 1.0f; s[6] = decay; s[7] = average_scalar;
    // (a reward-modulated lattice keeps the parameters -- a partner's visit may borrow them -- but has no rule of its own)
    net->plast_host[l->slot] = (do_plasticity && !(l->slot < net->rm_on_host.size() && (net->rm_on_host[l->slot] & RM_IS_MODULATED))) ? 1u : 0u;
    net->any_plasticity = false;
    for (uint32_t p : net->plast_host) net->any_plasticity |= (p != 0);
    TRY(end_run(net));
    HIP_TRY(copy_sync(net, net->stdp_dev, net->stdp_host.data(), net->stdp_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(copy_sync(net, net->plast_dev, net->plast_host.data(), net->plast_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    return SNN_OK;
}
ABI_CATCH

namespace {
size_t trace_elems(const snn_network *net) { return net->csr ? (size_t)net->sell_entries : wcount(net->n_tot, net->ld); }

// W and a (zeroed) trace candidate read and written back in place, index-aligned: the four streams of k_inputs_rstdp, no bit
// changed.  Average of two passes after one warm pass, in ms.
int time_rw_pass(snn_network *net, float *buf, size_t n4, float *ms)
{
    *ms = 0.0f;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(SNN_ERR_QUEUE, "hipEventCreate failed");
    const unsigned blocks = 256 * 16;
    auto pass = [&]() { hipLaunchKernelGGL(k_probe_rw_pair, dim3(blocks), dim3(256), 0, net->stream, (const probe_v4f *)net->W, (probe_v4f *)net->W, (const probe_v4f *)buf, (probe_v4f *)buf, n4); };
    pass();
    int rc = SNN_OK;
    if (hipGetLastError() != hipSuccess) rc = fail(SNN_ERR_QUEUE, "placement probe launch failed");
    if (rc == SNN_OK && hipEventRecord(e0, net->stream) != hipSuccess) rc = fail(SNN_ERR_QUEUE, "hipEventRecord failed");
    pass(); pass();
    if (rc == SNN_OK && (hipEventRecord(e1, net->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                         hipEventElapsedTime(ms, e0, e1) != hipSuccess))
        rc = fail(SNN_ERR_WAIT, "placement timing failed");
    *ms *= 0.5f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int ensure_traces(snn_network *net)
{
    if (net->trace) return SNN_OK;
    const size_t n = std::max<size_t>(trace_elems(net), 64);
    HIP_TRY(alloc_streamed(reinterpret_cast<void **>(&net->trace), n * 4), SNN_ERR_BUFFER_CREATE);
    HIP_TRY(hipMemsetAsync(net->trace, 0, n * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    // The trace matrix of a dense handle is rewritten every step (k_inputs_rstdp): like the synapse matrix its HBM placement
    // decides a few per cent of the pass (choose_matrix_placement) -- up to three more candidates, each held until the
    // choice is made, timed next to W with the pass's own four streams (k_probe_rw_pair); the fastest stays.
    if (!net->csr && n * 4 >= ((size_t)1 << 30)) {
        float best = 0.0f;
        int rc = time_rw_pass(net, net->trace, n / 4, &best);
        hvec<void *> losers;
        for (int cand = 0; cand < 3 && rc == SNN_OK; ++cand) {
            size_t free_b = 0, total_b = 0;
            void *b = nullptr;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < n * 4 + (n >> 0) || alloc_streamed(&b, n * 4) != hipSuccess) break;
            float ms = 0.0f;
            if (hipMemsetAsync(b, 0, n * 4, net->stream) != hipSuccess) { losers.push_back(b); break; }
            rc = time_rw_pass(net, static_cast<float *>(b), n / 4, &ms);
            if (getenv("SNN_DEBUG_PLACEMENT"))
                fprintf(stderr, "[snn] trace placement (%zu vectors): held %p %.3f ms, candidate %p %.3f ms\n", n / 4, (void *)net->trace, best, b, ms);
            if (rc == SNN_OK && ms < best * 0.99f) {
                losers.push_back(net->trace);
                net->trace = static_cast<float *>(b);
                best = ms;
            } else {
                losers.push_back(b);
            }
        }
        for (void *p : losers) (void)hipFree(p);
        if (rc) return rc;
    }
    return SNN_OK;
}
} // namespace

int snn_set_reward_modulator(snn_network_t *net, uint32_t id, float dopamine, float tau_d, float tau_c, float a_plus,
                             float a_minus, float tau_plus, float tau_minus, float dt, int do_modulation) ABI_TRY
{
    if (net) net->cross_checked = false;
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "reward modulation belongs to neuron lattices");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const size_t nl = net->rm_on_host.size();
    // dopamine of the other lattices evolves on the device: refresh the host copy before rewriting the table
    HIP_TRY(copy_sync(net, net->rm_host.data(), net->rm_dev, nl * RM_STRIDE * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    float *m = &net->rm_host[(size_t)l->slot * RM_STRIDE];
    m[0] = dopamine; m[1] = tau_d; m[2] = tau_c; m[3] = a_plus; m[4] = a_minus; m[5] = tau_plus; m[6] = tau_minus; m[7] = dt;
    m[RM_DOPAMINE_BEFORE] = dopamine;
    // bit 0 do_modulation, bit 1 "this is a reward-modulated lattice" (snn_kernels_reward.hpp): the call makes the lattice one for good
    net->rm_on_host[l->slot] = (do_modulation ? RM_DO_MODULATION : 0u) | RM_IS_MODULATED;
    net->any_modulation = net->any_modulated = false;
    for (uint32_t v : net->rm_on_host) { net->any_modulation |= (v & RM_DO_MODULATION) != 0; net->any_modulated |= (v & RM_IS_MODULATED) != 0; }
    {
        // a RewardModulatedLattice has no STDP rule of its own, whether or not it is modulating
        net->plast_host[l->slot] = 0;
        net->any_plasticity = false;
        for (uint32_t p : net->plast_host) net->any_plasticity |= (p != 0);
        HIP_TRY(copy_sync(net, net->plast_dev, net->plast_host.data(), net->plast_host.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    }
    if (do_modulation) {
        if (net->csr && !net->csr_ptr) return fail(SNN_ERR_BAD_STATE, "set the sparse graph before enabling reward modulation");
        TRY(ensure_traces(net));
    }
    HIP_TRY(copy_sync(net, net->rm_dev, net->rm_host.data(), nl * RM_STRIDE * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(copy_sync(net, net->rm_on_dev, net->rm_on_host.data(), nl * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    hipLaunchKernelGGL(k_modulator_update, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, net->stream, net->rm_dev,
                       net->rm_on_dev, (uint32_t)nl, 0.0f, 1);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}
ABI_CATCH

int snn_get_dopamine(snn_network_t *net, uint32_t id, float *dopamine) ABI_TRY
{
    if (!net || !dopamine) return fail(SNN_ERR_BAD_ARG, "null pointer");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "reward modulation belongs to neuron lattices");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    HIP_TRY(copy_sync(net, dopamine, net->rm_dev + (size_t)l->slot * RM_STRIDE, 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}
ABI_CATCH

int snn_apply_reward(snn_network_t *net, float reward) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->any_modulated) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    // a deferred weight update keeps ONE earlier dopamine value: a second reward before the next step applies it first
    if (net->rstdp_pending && net->reward_since_defer) TRY(flush_rstdp(net));
    if (net->rstdp_pending) net->reward_since_defer = true;
    const size_t nl = net->rm_on_host.size();
    hipLaunchKernelGGL(k_modulator_update, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, net->stream, net->rm_dev,
                       net->rm_on_dev, (uint32_t)nl, reward, 0);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    return SNN_OK;
}
ABI_CATCH

int snn_run_with_reward(snn_network_t *net, float reward) ABI_TRY
{
    // run_lattice_with_reward does nothing at all with both synapse kinds off (neuron/mod.rs:3250-3257)
    if (net && net->finalized && !net->electrical && !net->chemical) return SNN_OK;
    int rc = snn_apply_reward(net, reward);
    return rc ? rc : snn_run(net, 1);
}
ABI_CATCH

// TraceRSTDP::c of the edges in presynaptic rows [pre_begin, pre_begin + pre_count), row-major [pre_count][n_neurons];
// a shard handle reads / writes its own columns only.
namespace {
// TraceRSTDP::dw and ::counter of the connections of a reward-modulated network (k_reward_cross): allocated with the first such
// connection, both in the layout of W (the counter as 0.0 / 1.0)
int ensure_pending(snn_network *net)
{
    if (net->pending) return SNN_OK;
    const size_t n = std::max<size_t>(trace_elems(net), 64);           // dense: the layout of W; sparse: one word per stored entry
    HIP_TRY(snn_malloc(&net->pending, n * 4), SNN_ERR_BUFFER_CREATE);
    HIP_TRY(snn_malloc(&net->edge_counter, n * 4), SNN_ERR_BUFFER_CREATE);
    if (!net->cross_bad) HIP_TRY(snn_malloc(&net->cross_bad, 256), SNN_ERR_BUFFER_CREATE);
    HIP_TRY(hipMemsetAsync(net->pending, 0, n * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemsetAsync(net->edge_counter, 0, n * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}
} // namespace

static int trace_rows_io(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *traces, bool set, int plane = 0)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a CSR graph: use snn_set_traces_csr / snn_get_traces_csr");
    if ((uint64_t)pre_begin + pre_count > net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "row range exceeds n_tot");
    if (pre_count == 0 || net->nn == 0 || net->n_loc == 0) return SNN_OK;
    if (!traces) return fail(SNN_ERR_BAD_ARG, "traces is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_traces(net));
    if (plane) TRY(ensure_pending(net));
    float *matrix = plane == 2 ? net->edge_counter : plane == 1 ? net->pending : net->trace;
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    // through a row-major staging block of <= 64 MiB and <= 32768 rows per hop (the matrix is in quad-row order)
    float *host = traces + net->q0;
    const uint32_t hop = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(pre_count, 32768), ((size_t)64 << 20) / ((size_t)net->n_loc * 4)));
    float *stage = nullptr;
    HIP_TRY(snn_malloc(&stage, (size_t)hop * net->n_loc * 4), SNN_ERR_BUFFER_CREATE);
    int rc = SNN_OK;
    for (uint32_t r = 0; r < pre_count && rc == SNN_OK; r += hop) {
        const uint32_t rows = std::min(hop, pre_count - r);
        const dim3 grid((net->n_loc + 255) / 256, rows);
        if (set) {
            if (copy2d_sync(net, stage, (size_t)net->n_loc * 4, host + (size_t)r * net->nn, (size_t)net->nn * 4, (size_t)net->n_loc * 4, rows,
                            hipMemcpyHostToDevice) != hipSuccess) { rc = fail(SNN_ERR_BUFFER_WRITE, "trace upload failed"); break; }
            hipLaunchKernelGGL(k_rows_staging, grid, dim3(256), 0, net->stream, matrix, net->ld, net->n_loc, pre_begin + r, rows, stage, 1);
            if (hipStreamSynchronize(net->stream) != hipSuccess) rc = fail(SNN_ERR_WAIT, "trace upload wait failed");
        } else {
            hipLaunchKernelGGL(k_rows_staging, grid, dim3(256), 0, net->stream, matrix, net->ld, net->n_loc, pre_begin + r, rows, stage, 0);
            if (hipStreamSynchronize(net->stream) != hipSuccess) { rc = fail(SNN_ERR_WAIT, "trace download wait failed"); break; }
            if (copy2d_sync(net, host + (size_t)r * net->nn, (size_t)net->nn * 4, stage, (size_t)net->n_loc * 4, (size_t)net->n_loc * 4, rows,
                            hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SNN_ERR_BUFFER_READ, "trace download failed");
        }
    }
    (void)hipFree(stage);
    return rc;
}
int snn_set_trace_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const float *traces) ABI_TRY
{ return trace_rows_io(net, pre_begin, pre_count, const_cast<float *>(traces), true); }
ABI_CATCH
int snn_get_trace_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *traces) ABI_TRY
{ return trace_rows_io(net, pre_begin, pre_count, traces, false); }
ABI_CATCH
int snn_set_pending_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const float *pending) ABI_TRY
{ return trace_rows_io(net, pre_begin, pre_count, const_cast<float *>(pending), true, 1); }
ABI_CATCH
int snn_get_pending_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, float *pending) ABI_TRY
{ return trace_rows_io(net, pre_begin, pre_count, pending, false, 1); }
ABI_CATCH
// TraceRSTDP::counter of the same connections (0 / 1), one byte each
int snn_set_counter_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, const uint8_t *counters) ABI_TRY
{
    if (!net || !counters) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a CSR graph");
    if ((uint64_t)pre_begin + pre_count > net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "row range exceeds n_tot");
    try {
        hvec<float> rows((size_t)pre_count * net->nn);
        for (size_t i = 0; i < rows.size(); ++i) rows[i] = counters[i] ? 1.0f : 0.0f;
        return trace_rows_io(net, pre_begin, pre_count, rows.data(), true, 2);
    } catch (const std::bad_alloc &) {
        return fail(SNN_ERR_BUFFER_CREATE, "out of host memory for the counter rows");
    }
}
ABI_CATCH
int snn_get_counter_rows(snn_network_t *net, uint32_t pre_begin, uint32_t pre_count, uint8_t *counters) ABI_TRY
{
    if (!net || !counters) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a CSR graph");
    if ((uint64_t)pre_begin + pre_count > net->n_tot) return fail(SNN_ERR_DIM_MISMATCH, "row range exceeds n_tot");
    try {
        hvec<float> rows((size_t)pre_count * net->nn, 0.0f);
        TRY(trace_rows_io(net, pre_begin, pre_count, rows.data(), false, 2));
        for (size_t i = 0; i < rows.size(); ++i) counters[i] = rows[i] != 0.0f ? 1 : 0;
    } catch (const std::bad_alloc &) {
        return fail(SNN_ERR_BUFFER_CREATE, "out of host memory for the counter rows");
    }
    return SNN_OK;
}
ABI_CATCH

int snn_set_connection_kind(snn_network_t *net, uint32_t pre_id, uint32_t post_id, int kind) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (kind < 0 || kind > 2) return fail(SNN_ERR_BAD_ARG, "kind: 0 a plain network's connection, 1 reward-modulated weights, 2 plain weights of a reward-modulated network");
    const LatticeInfo *pre = find_lattice(net, pre_id), *post = find_lattice(net, post_id);
    if (!pre || !post || post->spike_train) return fail(SNN_ERR_BAD_ARG, "connections end in neuron lattices");
    if (pre_id == post_id) return fail(SNN_ERR_BAD_ARG, "a lattice's own edges follow its own rule (snn_set_plasticity / snn_set_reward_modulator)");
    if (net->sharded && net->n_shards > 1) return fail(SNN_ERR_BAD_STATE, "connections of a reward-modulated network: unsharded handles");
    if (net->csr && !net->csr_ptr) return fail(SNN_ERR_BAD_STATE, "set the sparse graph before tagging its connections");
    const size_t nl = net->lattices.size(), ns = net->st_lattices.size();
    if (nl > 64) return fail(SNN_ERR_BAD_STATE, "connections of a reward-modulated network: at most 64 neuron lattices");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    if (net->conn_kind_host.empty()) {
        net->conn_kind_host.assign((nl + ns) * nl, 0);
        HIP_TRY(snn_malloc(&net->conn_kind_dev, std::max<size_t>((nl + ns) * nl, 256)), SNN_ERR_BUFFER_CREATE);
    }
    const size_t source = pre->spike_train ? nl + pre->slot : pre->slot;
    net->conn_kind_host[source * nl + post->slot] = (uint8_t)kind;
    net->any_conn_kind = false;
    net->cross_checked = false;
    for (uint8_t k : net->conn_kind_host) net->any_conn_kind |= k != 0;
    HIP_TRY(copy_sync(net, net->conn_kind_dev, net->conn_kind_host.data(), net->conn_kind_host.size(), hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    if (kind) {
        TRY(ensure_traces(net));
        TRY(ensure_pending(net));
    }
    return SNN_OK;
}
ABI_CATCH

static int traces_csr_io(snn_network_t *net, float *traces, uint64_t nnz, bool set, int plane = 0)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->csr) return fail(SNN_ERR_BAD_STATE, "handle holds a dense graph");
    if (nnz != net->nnz) return fail(SNN_ERR_DIM_MISMATCH, "nnz does not match the stored graph");
    if (nnz == 0) return SNN_OK;
    if (!traces) return fail(SNN_ERR_BAD_ARG, "traces is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_traces(net));
    if (plane) TRY(ensure_pending(net));
    float *array = plane == 2 ? net->edge_counter : plane == 1 ? net->pending : net->trace;
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    hvec<float> sell((size_t)net->sell_entries);
    HIP_TRY(copy_sync(net, sell.data(), array, sell.size() * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    if (set) {
        for (uint64_t e = 0; e < nnz; ++e) sell[net->edge_slot_host[e]] = traces[e];
        HIP_TRY(copy_sync(net, array, sell.data(), sell.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    } else {
        for (uint64_t e = 0; e < nnz; ++e) traces[e] = sell[net->edge_slot_host[e]];
    }
    return SNN_OK;
}
int snn_set_traces_csr(snn_network_t *net, const float *traces, uint64_t nnz) ABI_TRY
{ return traces_csr_io(net, const_cast<float *>(traces), nnz, true); }
ABI_CATCH
int snn_get_traces_csr(snn_network_t *net, float *traces, uint64_t nnz) ABI_TRY
{ return traces_csr_io(net, traces, nnz, false); }
ABI_CATCH
int snn_set_pending_csr(snn_network_t *net, const float *pending, uint64_t nnz) ABI_TRY
{ return traces_csr_io(net, const_cast<float *>(pending), nnz, true, 1); }
ABI_CATCH
int snn_get_pending_csr(snn_network_t *net, float *pending, uint64_t nnz) ABI_TRY
{ return traces_csr_io(net, pending, nnz, false, 1); }
ABI_CATCH
int snn_set_counters_csr(snn_network_t *net, const uint8_t *counters, uint64_t nnz) ABI_TRY
{
    if (!net || (nnz && !counters)) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (net->finalized && net->csr && nnz != net->nnz) return fail(SNN_ERR_DIM_MISMATCH, "nnz does not match the stored graph");
    try {
        hvec<float> v(nnz);
        for (uint64_t e = 0; e < nnz; ++e) v[e] = counters[e] ? 1.0f : 0.0f;
        return traces_csr_io(net, v.data(), nnz, true, 2);
    } catch (const std::bad_alloc &) {
        return fail(SNN_ERR_BUFFER_CREATE, "out of host memory for the counters");
    }
}
ABI_CATCH
int snn_get_counters_csr(snn_network_t *net, uint8_t *counters, uint64_t nnz) ABI_TRY
{
    if (!net || (nnz && !counters)) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (net->finalized && net->csr && nnz != net->nnz) return fail(SNN_ERR_DIM_MISMATCH, "nnz does not match the stored graph");
    try {
        hvec<float> v(nnz, 0.0f);
        TRY(traces_csr_io(net, v.data(), nnz, false, 2));
        for (uint64_t e = 0; e < nnz; ++e) counters[e] = v[e] != 0.0f ? 1 : 0;
    } catch (const std::bad_alloc &) {
        return fail(SNN_ERR_BUFFER_CREATE, "out of host memory for the counters");
    }
    return SNN_OK;
}
ABI_CATCH

int snn_set_history(snn_network_t *net, int voltage_history, int spike_history) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if ((voltage_history != 0) != (net->want_vhist != 0) || (spike_history != 0) != (net->want_raster != 0)) {
        // switching what is recorded restarts the record so that all rows cover the same steps
        net->hist_steps = 0; net->hist_tick = 0;
    }
    net->want_vhist = voltage_history ? 1 : 0;
    net->want_raster = spike_history ? 1 : 0;
    return SNN_OK;
}
ABI_CATCH

int snn_reset_history(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    net->hist_steps = 0; net->hist_tick = 0;
    if (net->finalized && net->spike_counts) {
        HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
        TRY(end_run(net));
        HIP_TRY(memset_sync(net, net->spike_counts, 0, (size_t)net->n_pad * 4), SNN_ERR_BUFFER_WRITE);
    }
    return SNN_OK;
}
ABI_CATCH

int snn_set_firing_times(snn_network_t *net, uint32_t id, const uint32_t *cell_ptr, const float *times, size_t n_times) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->st_kind != SNN_ST_PRESET) return fail(SNN_ERR_BAD_STATE, "the spike-train model is not SNN_ST_PRESET");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || !l->spike_train) return fail(SNN_ERR_BAD_ARG, "no such spike-train lattice");
    if (l->count == 0) return SNN_OK;
    if (!cell_ptr || (n_times && !times)) return fail(SNN_ERR_BAD_ARG, "null pointer");
    if (cell_ptr[0] != 0 || cell_ptr[l->count] != n_times) return fail(SNN_ERR_DIM_MISMATCH, "cell_ptr must run from 0 to n_times");
    for (uint32_t i = 0; i < l->count; ++i)
        if (cell_ptr[i] > cell_ptr[i + 1]) return fail(SNN_ERR_BAD_ARG, "cell_ptr must be non-decreasing");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const uint32_t c0 = l->first - net->nn;
    for (uint32_t i = 0; i < l->count; ++i) net->preset_host[c0 + i].assign(times + cell_ptr[i], times + cell_ptr[i + 1]);
    hvec<uint32_t> ptr((size_t)net->c_pad + 1, 0);
    hvec<float> flat;
    for (uint32_t s = 0; s < net->nc; ++s) {
        ptr[s] = (uint32_t)flat.size();
        flat.insert(flat.end(), net->preset_host[s].begin(), net->preset_host[s].end());
    }
    for (size_t s = net->nc; s <= net->c_pad; ++s) ptr[s] = (uint32_t)flat.size();
    float *nt = nullptr;
    HIP_TRY(snn_malloc(&nt, std::max<size_t>(256, flat.size() * 4)), SNN_ERR_BUFFER_CREATE);
    if (!flat.empty()) HIP_TRY(copy_sync(net, nt, flat.data(), flat.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(copy_sync(net, const_cast<uint32_t *>(net->ca.preset_ptr), ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice),
            SNN_ERR_BUFFER_WRITE);
    if (net->preset_times_dev) (void)hipFree(net->preset_times_dev);
    net->preset_times_dev = nt;
    net->ca.preset_times = nt;
    return SNN_OK;
}
ABI_CATCH

int snn_set_graph_history(snn_network_t *net, uint32_t id, int enable) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "graph histories belong to neuron lattices");
    if (net->csr || net->sharded) return fail(SNN_ERR_BAD_STATE, "graph histories need a dense, unsharded handle");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    if (enable < 0 || enable > 2) return fail(SNN_ERR_BAD_ARG, "enable: 0 off, 1 after the weight updates, 2 before them");
    if ((enable != 0) != (net->want_whist[l->slot] != 0)) { net->hist_steps = 0; net->hist_tick = 0; }
    net->want_whist[l->slot] = enable;
    net->any_whist = false;
    for (int v : net->want_whist) net->any_whist |= (v != 0);
    return SNN_OK;
}
ABI_CATCH

int snn_get_graph_history(snn_network_t *net, uint32_t id, float *dst, size_t steps) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "graph histories belong to neuron lattices");
    if (!net->want_whist[l->slot]) return fail(SNN_ERR_BAD_STATE, "graph history is off for this lattice");
    if (steps != net->hist_steps) return fail(SNN_ERR_DIM_MISMATCH, "history size mismatch");
    if (steps == 0 || l->count == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    HIP_TRY(copy_sync(net, dst, net->whist[l->slot], steps * (size_t)l->count * l->count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}
ABI_CATCH

int snn_set_history_stride(snn_network_t *net, uint32_t every) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (every == 0) return fail(SNN_ERR_BAD_ARG, "stride must be at least 1");
    if (every != net->hist_every) { net->hist_steps = 0; net->hist_tick = 0; }
    net->hist_every = every;
    return SNN_OK;
}
ABI_CATCH

int snn_set_reduced_history(snn_network_t *net, int average_voltage, int eeg, int spike_counts,
                            float reference_voltage, float distance, float conductivity) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if ((average_voltage != 0) != (net->want_avg != 0) || (eeg != 0) != (net->want_eeg != 0))
        net->hist_steps = 0, net->hist_tick = 0;      // all recorded rows must cover the same steps
    net->want_avg = average_voltage ? 1 : 0;
    net->want_eeg = eeg ? 1 : 0;
    net->want_counts = spike_counts ? 1 : 0;
    net->eeg_ref = reference_voltage; net->eeg_dist = distance; net->eeg_cond = conductivity;
    return SNN_OK;
}
ABI_CATCH

static int get_summary(snn_network_t *net, uint32_t id, float *dst, size_t steps, bool eeg)
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "reduced histories exist for neuron lattices");
    if (!(eeg ? net->want_eeg : net->want_avg)) return fail(SNN_ERR_BAD_STATE, "this reduced history is off");
    if (steps != net->hist_steps) return fail(SNN_ERR_DIM_MISMATCH, "history size mismatch");
    if (steps == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const size_t nl = net->lattices.size();
    const float *src = (eeg ? net->summ_eeg : net->summ_avg) + l->slot;
    HIP_TRY(copy2d_sync(net, dst, 4, src, nl * 4, 4, steps, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}

int snn_get_average_voltage_history(snn_network_t *net, uint32_t id, float *dst, size_t steps) ABI_TRY
{ return get_summary(net, id, dst, steps, false); }
ABI_CATCH
int snn_get_eeg_history(snn_network_t *net, uint32_t id, float *dst, size_t steps) ABI_TRY
{ return get_summary(net, id, dst, steps, true); }
ABI_CATCH

int snn_get_spike_counts(snn_network_t *net, uint32_t id, uint32_t *dst, size_t count) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "spike counts exist for neuron lattices");
    if (count != l->count) return fail(SNN_ERR_DIM_MISMATCH, "count must equal rows*cols");
    if (count == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    HIP_TRY(copy_sync(net, dst, net->spike_counts + l->first, count * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    return SNN_OK;
}
ABI_CATCH

int snn_get_clock(const snn_network_t *net, uint64_t *clock) ABI_TRY
{
    if (!net || !clock) return fail(SNN_ERR_BAD_ARG, "null argument");
    *clock = (uint64_t)net->clock;
    return SNN_OK;
}
ABI_CATCH

int snn_reset_timing(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    net->clock = 0;
    net->view_dirty = true;
    for (auto &c : net->st_clock) c = 0;
    HIP_TRY(hipMemsetAsync(net->na.last_firing_time, 0xFF, (size_t)net->n_pad * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipMemsetAsync(net->ca.last_firing_time, 0xFF, (size_t)net->c_pad * 4, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}
ABI_CATCH

int snn_set_clock(snn_network_t *net, uint64_t clock) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (clock > 0x7FFFFFFFull) return fail(SNN_ERR_BAD_ARG, "the clock must fit the firing times' 31 bits");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    net->clock = (long long)clock;
    net->view_dirty = true;            // the cells' gap-junction values are functions of the clock
    return SNN_OK;
}
ABI_CATCH

int snn_set_spike_train_clock(snn_network_t *net, uint32_t id, uint64_t clock) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || !l->spike_train) return fail(SNN_ERR_BAD_ARG, "no such spike-train lattice");
    if (clock > 0x7FFFFFFFull) return fail(SNN_ERR_BAD_ARG, "the clock must fit the firing times' 31 bits");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    net->st_clock[l->slot] = (long long)clock;
    return SNN_OK;
}
ABI_CATCH

int snn_get_spike_train_clock(snn_network_t *net, uint32_t id, uint64_t *clock) ABI_TRY
{
    if (!net || !clock) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || !l->spike_train) return fail(SNN_ERR_BAD_ARG, "no such spike-train lattice");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    *clock = (uint64_t)net->st_clock[l->slot];
    return SNN_OK;
}
ABI_CATCH

const char *snn_debug_verify_report(snn_network_t *net) ABI_TRY { return net ? net->verify_text.c_str() : ""; } ABI_CATCH_PTR

// Test support for the exception barrier and the error paths (snn_network_state.hpp, alloc_fault_now): the n-th allocation the
// library makes from now on -- device, page-locked or a host table -- fails once (n <= 0 disarms); process-wide.
int snn_debug_fail_alloc_at(int64_t n, uint64_t *allocations_so_far) ABI_TRY
{
    AllocFault &f = alloc_fault();
    if (allocations_so_far) *allocations_so_far = f.seen.load(std::memory_order_relaxed);
    f.countdown.store(n > 0 ? n : 0, std::memory_order_relaxed);
    return SNN_OK;
}
ABI_CATCH

// Test support for the hunt after stray host writes (snn_network_state.hpp, HostAllocHooks): from now on every host table of the
// library comes from `alloc` and goes back through `release`; both null: the default allocator again (tables taken from the
// hooks before that still go back to them: keep the functions alive).  Process-wide.
int snn_debug_set_host_allocator(snn_host_alloc_fn alloc, snn_host_release_fn release) ABI_TRY
{
    static HostAllocHooks hooks;                   // (one set per process: a later call replaces the functions in place)
    if ((alloc == nullptr) != (release == nullptr)) return fail(SNN_ERR_BAD_ARG, "both functions or none");
    if (!alloc) { host_alloc_hooks().store(nullptr, std::memory_order_release); return SNN_OK; }
    hooks.alloc = alloc; hooks.release = release;
    host_alloc_hooks().store(&hooks, std::memory_order_release);
    return SNN_OK;
}
ABI_CATCH

// Test support (tests/checkpoint.py): everything a later run call reads -- every device array of the handle up to 256 MiB in
// all, the sparse weights, traces, and the host-side cursors of the stepper -- kept in host memory; restore puts it back.
// Valid between calls that leave the handle's STRUCTURE alone (run calls, attribute and weight writes): a restore after a
// structural call (a new sparse graph, a rebuilt exchange plan, histories switched) fails with SNN_ERR_BAD_STATE.
int snn_debug_checkpoint(snn_network_t *net, int restore) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    if (restore) net->img_stale = net->img_stale_direct = true;          // the sparse weights come back: the step image's records are rebuilt
    auto &cp = net->checkpoint;
    hvec<std::pair<void *, size_t>> arrays;
    for (const auto &kv : net->alloc_bytes)
        if (kv.first != (void *)net->snap_buf && kv.first != (void *)net->snap_table && kv.first != (void *)net->verify_buf &&
            kv.first != (void *)net->run_granules && kv.first != (void *)net->run_partials && kv.first != (void *)net->run_timing)
            arrays.emplace_back(kv.first, kv.second);
    if (net->csr_w) arrays.emplace_back(net->csr_w, (size_t)net->sell_entries * 4);
    if (net->trace) arrays.emplace_back(net->trace, std::max<size_t>(trace_elems(net), 64) * 4);
    for (void *m : {(void *)net->pending, (void *)net->edge_counter})
        if (m) arrays.emplace_back(m, std::max<size_t>(trace_elems(net), 64) * 4);
    size_t total = 0;
    for (const auto &a : arrays) total += a.second;
    if (total > ((size_t)256 << 20)) return fail(SNN_ERR_BAD_STATE, "checkpoints are for small handles (256 MiB of device state at most)");
    if (!restore) {
        cp.arrays.clear();
        for (const auto &a : arrays) {
            cp.arrays.emplace_back(a.first, hvec<uint8_t>(a.second));
            HIP_TRY(hipMemcpyAsync(cp.arrays.back().second.data(), a.first, a.second, hipMemcpyDeviceToHost, net->stream), SNN_ERR_BUFFER_READ);
        }
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
        cp.clock = net->clock; cp.st_clock = net->st_clock; cp.hist_steps = net->hist_steps; cp.hist_tick = net->hist_tick;
        cp.shadow_cur = net->shadow_cur; cp.shadow_valid = net->shadow_valid; cp.cell_view_cur = net->cell_view_cur;
        cp.view_dirty = net->view_dirty; cp.counts_dirty = net->counts_dirty; cp.uni_dirty = net->uni_dirty;
        cp.live_mask_applied = net->live_mask_applied; cp.n_live = net->n_live;
        for (int k = 0; k < K_TYPES; ++k) cp.live_type[k] = net->live_type[k];
        cp.persistent_run = net->persistent_run; cp.mirror_mask = net->mirror_mask;
        cp.valid = true;
        return SNN_OK;
    }
    if (!cp.valid) return fail(SNN_ERR_BAD_STATE, "no checkpoint was taken");
    // every array of the checkpoint must still be the handle's (arrays allocated since -- shadows, views, granules -- are caches
    // whose validity flags are put back below)
    std::map<void *, size_t> current(arrays.begin(), arrays.end());
    for (const auto &a : cp.arrays) {
        const auto it = current.find(a.first);
        if (it == current.end() || it->second != a.second.size())
            return fail(SNN_ERR_BAD_STATE, "the handle's structure changed since the checkpoint");
    }
    for (const auto &a : cp.arrays)
        HIP_TRY(hipMemcpyAsync(a.first, a.second.data(), a.second.size(), hipMemcpyHostToDevice, net->stream), SNN_ERR_BUFFER_WRITE);
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    net->clock = cp.clock; net->st_clock = cp.st_clock; net->hist_steps = cp.hist_steps; net->hist_tick = cp.hist_tick;
    net->shadow_cur = cp.shadow_cur; net->shadow_valid = cp.shadow_valid; net->cell_view_cur = cp.cell_view_cur;
    net->view_dirty = cp.view_dirty; net->counts_dirty = cp.counts_dirty; net->uni_dirty = cp.uni_dirty;
    net->live_mask_applied = cp.live_mask_applied; net->n_live = cp.n_live;
    for (int k = 0; k < K_TYPES; ++k) net->live_type[k] = cp.live_type[k];
    net->persistent_run = cp.persistent_run; net->mirror_mask = cp.mirror_mask;
    net->stdp_pending = false; net->stdp_pending_rows_only = false; net->rstdp_pending = false; net->reward_since_defer = false;
    net->cells_stepped = false; net->local_inputs_done = false; net->run_tag = 1;
    if (net->run_granules) {
        HIP_TRY(hipMemsetAsync(net->run_granules, 0, RUN_GRANULE_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(hipMemsetAsync(net->run_partials, 0, RUN_PARTIAL_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    }
    return SNN_OK;
}
ABI_CATCH

extern "C++" {
namespace {
// host-side cursors a run moves: what a rollback (or the second pass of "verify") puts back
struct RunCursors {
    long long clock, run_step_offset;
    uint64_t hist_steps, hist_tick, launches, steps, stdp_steps;
    size_t ev_used;
};
RunCursors run_cursors(const snn_network *net)
{
    return {net->clock, net->run_step_offset, net->hist_steps, net->hist_tick, net->stat_run_launches, net->stat_run_steps,
            net->stat_run_stdp_steps, net->ev_used};
}
void restore_cursors(snn_network *net, const RunCursors &c)
{
    net->clock = c.clock; net->run_step_offset = c.run_step_offset;
    net->hist_steps = c.hist_steps; net->hist_tick = c.hist_tick;
    net->stat_run_launches = c.launches; net->stat_run_steps = c.steps; net->stat_run_stdp_steps = c.stdp_steps;
    net->ev_used = c.ev_used;
}

// The steps of a run call, between begin_run and end_run.
int run_steps(snn_network *net, uint64_t iterations)
{
    uint64_t it = 0;
    if (iterations >= 4 && net->external_stream && run_resident_shape(net)) net->stat_run_external_stream += 1;
    // below 4 steps the launch's fixed cost (seed, weights into registers) shows
    while (iterations - it >= 4 && run_resident_applies(net)) {
        // The launch is a spin-wait all-to-all between workgroups that must all be resident.  A probe vouches for that
        // once per handle; should they lose sight of each other later all the same (device shared with a long kernel),
        // the waiters give up, the handle is put back exactly where the CHUNK started -- device state from a snapshot taken
        // in one launch, host cursors from `saved` -- and the remaining steps are taken with one launch per step, which this
        // handle then keeps.  The caller sees a slower call, never a half-stepped network.  The unit of the rollback is the
        // chunk (at most run_chunk_steps steps): what earlier chunks of the call committed -- the weights of STDP inside
        // the run among it -- is never replayed.
        const uint64_t chunk = std::min<uint64_t>(iterations - it, std::max<uint32_t>(4u, net->run_chunk_steps));
        const RunCursors saved = run_cursors(net);
        TRY(run_snapshot(net, /*restore=*/false));
        TRY(launch_run_resident(net, chunk, it));
        if (!net->persistent_run) break;              // the co-residency probe said no, nothing was stepped
        HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
        if (net->run_failed && net->run_failed[0]) {
            net->run_failed[0] = 0u;
            TRY(run_snapshot(net, /*restore=*/true));
            HIP_TRY(hipMemsetAsync(net->run_granules, 0, RUN_GRANULE_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
            HIP_TRY(hipMemsetAsync(net->run_partials, 0, RUN_PARTIAL_WORDS * 8, net->stream), SNN_ERR_BUFFER_WRITE);
            net->run_tag = 1;
            restore_cursors(net, saved);
            net->shadow_valid = false;
            net->view_dirty = true;
            net->persistent_run = 0;
            net->stat_run_fallbacks += 1;
            break;
        }
        it += chunk;
    }
    for (; it < iterations; ++it) {
        if (net->nn) TRY(step_begin(net));
        TRY(step_end(net));
        if (net->profile && (net->ev_used >= 8192 || net->ev_used_pl >= 8192)) TRY(collect_profile(net));
    }
    return SNN_OK;
}

// "verify": the matrices a run with weight updates rewrites (the snapshot table holds the small arrays only): synapse matrix or
// sparse weights, traces, dw, counters, the weights a one-launch run with STDP leaves behind
hvec<std::pair<void *, size_t>> verify_matrices(const snn_network *net)
{
    hvec<std::pair<void *, size_t>> m;
    if (!net->any_plasticity && !net->any_modulation && !net->any_conn_kind) return m;
    if (net->csr) { if (net->csr_w && net->sell_entries) m.emplace_back(net->csr_w, (size_t)net->sell_entries * 4); }
    else if (net->W) m.emplace_back(net->W, wcount(net->n_tot, net->ld) * 4);
    const size_t edges = std::max<size_t>(net->csr ? (size_t)net->sell_entries : wcount(net->n_tot, net->ld), 64) * 4;
    for (void *a : {(void *)net->trace, (void *)net->pending, (void *)net->edge_counter})
        if (a) m.emplace_back(a, edges);
    return m;
}

// "verify": what may be stepped twice from one snapshot -- nothing measured, the handle's own stream, unsharded; with weight updates
// only while the matrices fit a side buffer (64 MiB: the networks of the randomized tests)
bool verify_applies(const snn_network *net)
{
    if (!net->verify || net->profile || net->external_stream || net->sharded || !net->nn) return false;
    size_t bytes = 0;
    for (const auto &m : verify_matrices(net)) bytes += m.second;
    return bytes <= ((size_t)64 << 20);
}

// name of the array a snapshot entry covers, for the report of a "verify" mismatch
std::string describe_array(const snn_network *net, const void *ptr, uint32_t word)
{
    const char *p = static_cast<const char *>(ptr);
    auto inside = [&](const void *base, size_t bytes) { return base && p >= (const char *)base && p < (const char *)base + bytes; };
    const size_t plane = (size_t)net->xl.stride * 4;
    if (inside(net->xbuf, plane * NUM_PLANES)) return "exchange buffer, plane " + std::to_string(word / net->xl.stride) + ", neuron " + std::to_string(word % net->xl.stride);
    for (int i = 0; i < 2; ++i)
        if (inside(net->shadow[i], plane * NUM_PLANES)) return "shadow " + std::to_string(i) + ", plane " + std::to_string(word / net->xl.stride) + ", neuron " + std::to_string(word % net->xl.stride);
    for (int i = 0; i < 2; ++i)
        if (inside(net->cell_view[i], (size_t)net->c_pad * 8)) return "cell view " + std::to_string(i) + ", word " + std::to_string(word);
    const struct { const void *base; const char *name; } known[] = {
        {net->part_i, "part_i"}, {net->part_t, "part_t"}, {net->n_in, "n_in"}, {net->tcount, "tcount"}, {net->W, "W"},
        {net->spike_counts, "spike_counts"}, {net->spike_count, "spike_count"}, {net->st_clock_dev, "st_clock_dev"},
        {net->uni_neuron, "uniform table (neurons)"}, {net->uni_cell, "uniform table (cells)"}, {net->ca.presyn_value, "cells: presyn_value"},
        {net->ca.seed, "cells: seed"}, {net->ca.step, "cells: step"}, {net->ca.counter, "cells: counter"}, {net->lattice_slot, "lattice_slot"},
        {net->csr_w, "sparse weights"}, {net->trace, "traces"}, {net->pending, "dw of reward-modulated connections"},
        {net->edge_counter, "counters of reward-modulated connections"}};
    for (const auto &k : known)
        if (k.base == ptr) return std::string(k.name) + ", word " + std::to_string(word);
    for (const auto *table : {&net->neuron_attrs, &net->cell_attrs})
        for (const auto &kv : *table) {
            const Attr &a = kv.second;
            if (!a.base) continue;
            const uint32_t pad = table == &net->neuron_attrs ? net->n_pad : net->c_pad;
            const size_t bytes = (size_t)pad * 4 * ((a.store == S_PLAIN_K) ? K_TYPES : 1);
            if (inside(a.base, bytes))
                return std::string(table == &net->neuron_attrs ? "neurons: " : "cells: ") + kv.first + ", word " +
                       std::to_string(word + (uint32_t)((p - (const char *)a.base) / 4));
        }
    char buf[64];
    snprintf(buf, sizeof buf, "array at %p, word %u", ptr, word);
    return buf;
}
} // namespace
} // extern "C++"

int snn_run(snn_network_t *net, uint64_t iterations) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->n_shards > 1) return fail(SNN_ERR_BAD_STATE, "sharded handles are stepped with snn_step_begin/end or snn_run_sharded");
    if (iterations == 0 || net->n_tot == 0) return SNN_OK;          // gpu_lattices/mod.rs:1089-1091, 3196-3203
    if (!net->electrical && !net->chemical) return SNN_OK;           // neuron/mod.rs:1217, 2672
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(begin_run(net, iterations));
    if (!verify_applies(net)) {
        TRY(run_steps(net, iterations));
        return end_run(net, /*keep_stdp=*/true);
    }
    // ---- "verify": the same steps twice from the same snapshot, the outcomes compared on the device -------------------
    // (weight updates a previous call deferred are applied first: the delta vectors they read are rewritten by the steps below)
    TRY(flush_rstdp(net));
    TRY(flush_stdp(net));
    TRY(run_snapshot(net, /*restore=*/false));                    // (builds the table; its own copy of S(t) is not used here)
    if (!net->verify_buf || net->verify_words < net->snap_words) {
        if (net->verify_buf) {
            (void)hipFree(net->verify_buf);
            net->alloc_bytes.erase(net->verify_buf);
            net->allocs.erase(std::remove(net->allocs.begin(), net->allocs.end(), (void *)net->verify_buf), net->allocs.end());
            net->verify_buf = nullptr;
        }
        net->verify_words = net->snap_words + net->snap_words / 4 + 1024;
        TRY(dev_alloc_t(net, &net->verify_buf, 2 * net->verify_words));
        TRY(run_snapshot(net, /*restore=*/false));                // the handle has allocated: the table is laid out anew
    }
    if (!net->verify_report) HIP_TRY(snn_malloc(&net->verify_report, 256), SNN_ERR_BUFFER_CREATE);
    if (!net->snap_entries || net->snap_words > net->verify_words) {
        TRY(run_steps(net, iterations));
        return end_run(net, /*keep_stdp=*/true);
    }
    const CopyEntry *table = net->snap_table;
    const uint32_t *base = net->snap_buf;
    const uint64_t generation = net->snap_generation;
    const dim3 grid(std::max(1u, std::min(16u, (net->snap_max_words + 1023u) / 1024u)), net->snap_entries);
    uint32_t *start = net->verify_buf, *first = net->verify_buf + net->verify_words;
    const RunCursors c0 = run_cursors(net);
    const int shadow_cur = net->shadow_cur, view_cur = net->cell_view_cur;
    const bool shadow_valid = net->shadow_valid;
    // the matrices (runs with weight updates): [start state | first outcome], one after the other in a side buffer
    const auto matrices = verify_matrices(net);
    size_t big = 0;
    for (const auto &m : matrices) big += m.second;
    if (big > net->verify_big_bytes) {
        if (net->verify_big) (void)hipFree(net->verify_big);
        net->verify_big = nullptr; net->verify_big_bytes = 0;
        HIP_TRY(snn_malloc(&net->verify_big, 2 * big), SNN_ERR_BUFFER_CREATE);
        net->verify_big_bytes = big;
    }
    auto matrices_copy = [&](int half, bool restore) -> int {
        size_t off = (size_t)half * net->verify_big_bytes;
        if (restore) net->img_stale = net->img_stale_direct = true;
        for (const auto &m : matrices) {
            void *side = net->verify_big + off;
            HIP_TRY(hipMemcpyAsync(restore ? m.first : side, restore ? side : m.first, m.second, hipMemcpyDeviceToDevice, net->stream), SNN_ERR_BUFFER_WRITE);
            off += m.second;
        }
        return SNN_OK;
    };
    hipLaunchKernelGGL(k_copy_table_alt, grid, dim3(256), 0, net->stream, table, base, start, 0);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    TRY(matrices_copy(0, false));
    const uint64_t gave_up_0 = net->stat_run_fallbacks;
    TRY(run_steps(net, iterations));
    // what the first pass left of the caches (valid?, which copy) and how it stepped: compared only where both passes agree
    const bool shadow_valid_1 = net->shadow_valid, view_dirty_1 = net->view_dirty;
    const int shadow_cur_1 = net->shadow_cur, view_cur_1 = net->cell_view_cur;
    const uint64_t launches_1 = net->stat_run_launches - c0.launches, gave_up_1 = net->stat_run_fallbacks;
    if (net->snap_generation != generation) {
        // the run allocated and laid the table out anew (first one-launch run of a handle): nothing to compare with this time
        net->stat_verify_skipped += 1;
        return end_run(net, /*keep_stdp=*/true);
    }
    // (a deferred weight update still pending at the end of the first pass belongs to its outcome: applied before the copy)
    TRY(flush_rstdp(net));
    TRY(flush_stdp(net));
    hipLaunchKernelGGL(k_copy_table_alt, grid, dim3(256), 0, net->stream, table, base, first, 0);
    hipLaunchKernelGGL(k_copy_table_alt, grid, dim3(256), 0, net->stream, table, base, start, 1);
    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
    TRY(matrices_copy(1, false));
    TRY(matrices_copy(0, true));
    restore_cursors(net, c0);
    net->shadow_cur = shadow_cur; net->shadow_valid = shadow_valid; net->cell_view_cur = view_cur;
    net->cells_stepped = false; net->local_inputs_done = false;
    TRY(run_steps(net, iterations));
    TRY(flush_rstdp(net));
    TRY(flush_stdp(net));
    net->stat_verify_runs += 1;
    if (net->snap_generation == generation) {
        uint32_t report[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        HIP_TRY(hipMemsetAsync(net->verify_report, 0, 32, net->stream), SNN_ERR_BUFFER_WRITE);
        if (net->verify_fault) {          // option "verify_fault" (test hook): the second outcome is not the first
            // (values from 2^30: that word of the first matrix -- the weights -- of a handle with weight updates)
            const bool in_matrix = net->verify_fault >= (1u << 30) && !matrices.empty();
            hipLaunchKernelGGL(k_flip_bit, dim3(1), dim3(1), 0, net->stream,
                               in_matrix ? static_cast<uint32_t *>(matrices[0].first) : reinterpret_cast<uint32_t *>(net->xbuf),
                               in_matrix ? (size_t)(net->verify_fault - (1u << 30)) : (size_t)(net->verify_fault - 1));
            net->verify_fault = 0;
        }
        // left out of the comparison: the chunk partials (scratch of the two-kernel step only), and the shadows / cell views
        // unless both passes left the same copy valid -- a pass that fell back from the one-launch run to one launch per step
        // leaves them in another state than a pass that did not, and the validity flags (kept from this pass) say so
        SkipSet skip{};
        auto leave_out = [&](const void *array) {
            for (size_t k = 0; array && k < net->snap_table_host.size() && skip.n < 8; ++k)
                if ((const void *)net->snap_table_host[k].src == array) skip.entry[skip.n++] = (uint32_t)k + 1u;
        };
        leave_out(net->part_i); leave_out(net->part_t);
        if (!(shadow_valid_1 && net->shadow_valid && shadow_cur_1 == net->shadow_cur)) { leave_out(net->shadow[0]); leave_out(net->shadow[1]); }
        if (!(!view_dirty_1 && !net->view_dirty && view_cur_1 == net->cell_view_cur)) { leave_out(net->cell_view[0]); leave_out(net->cell_view[1]); }
        hipLaunchKernelGGL(k_compare_table_alt, grid, dim3(256), 0, net->stream, table, base, first, net->verify_report, skip);
        {
            // the matrices of the two outcomes, word for word (entry numbers past the table's: 2^20 + matrix index)
            size_t off = net->verify_big_bytes;
            uint32_t k = 0;
            for (const auto &m : matrices) {
                hipLaunchKernelGGL(k_compare_words, dim3(std::min<size_t>(1024, (m.second / 4 + 255) / 256)), dim3(256), 0, net->stream,
                                   reinterpret_cast<const uint32_t *>(net->verify_big + off), static_cast<const uint32_t *>(m.first), m.second / 4,
                                   (1u << 20) + k, net->verify_report);
                off += m.second;
                ++k;
            }
        }
        HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
        HIP_TRY(copy_sync(net, report, net->verify_report, 32, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
        if (report[0]) {
            const uint32_t n = report[0], e = report[1], w = report[2];
            char vals[96];
            snprintf(vals, sizeof vals, "first pass 0x%08x (%g), second pass 0x%08x (%g)", report[3],
                     (double)__builtin_bit_cast(float, report[3]), report[4], (double)__builtin_bit_cast(float, report[4]));
            const void *arr = (e >= 1 && e <= net->snap_table_host.size()) ? (const void *)net->snap_table_host[e - 1].src
                            : (e >= (1u << 20) && e - (1u << 20) < matrices.size()) ? matrices[e - (1u << 20)].first : nullptr;
            net->verify_text = "run of " + std::to_string(iterations) + " steps ending at clock " + std::to_string(net->clock) + ": " +
                               std::to_string(n) + " words differ between two executions from the same state; e.g. " +
                               describe_array(net, arr, w) + ": " + vals;
            net->verify_text += "; first pass: " + std::to_string(launches_1) + " one-launch launches, " + std::to_string(gave_up_1 - gave_up_0) +
                                " gave up; second pass: " + std::to_string(net->stat_run_launches - c0.launches) + " one-launch launches, " +
                                std::to_string(net->stat_run_fallbacks - gave_up_1) + " gave up";
            net->stat_verify_mismatches += 1;
            // Which of the two repeats?  A THIRD execution from the same start (handles without weight updates): the handle keeps its
            // outcome.
            if (matrices.empty()) {
                if (net->verify_third_words < net->verify_words) {
                    if (net->verify_third) (void)hipFree(net->verify_third);
                    net->verify_third = nullptr; net->verify_third_words = 0;
                    HIP_TRY(snn_malloc(&net->verify_third, net->verify_words * 4), SNN_ERR_BUFFER_CREATE);
                    net->verify_third_words = net->verify_words;
                }
                hipLaunchKernelGGL(k_copy_table_alt, grid, dim3(256), 0, net->stream, table, base, net->verify_third, 0);
                hipLaunchKernelGGL(k_copy_table_alt, grid, dim3(256), 0, net->stream, table, base, start, 1);
                HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
                const bool shadow_valid_2 = net->shadow_valid, view_dirty_2 = net->view_dirty;
                const int shadow_cur_2 = net->shadow_cur, view_cur_2 = net->cell_view_cur;
                restore_cursors(net, c0);
                net->shadow_cur = shadow_cur; net->shadow_valid = shadow_valid; net->cell_view_cur = view_cur;
                net->cells_stepped = false; net->local_inputs_done = false;
                TRY(run_steps(net, iterations));
                uint32_t differ[2] = {0u, 0u};
                for (int against = 0; against < 2; ++against) {
                    SkipSet sk{};
                    auto out = [&](const void *array) {
                        for (size_t k = 0; array && k < net->snap_table_host.size() && sk.n < 8; ++k)
                            if ((const void *)net->snap_table_host[k].src == array) sk.entry[sk.n++] = (uint32_t)k + 1u;
                    };
                    out(net->part_i); out(net->part_t);
                    const bool sv = against == 0 ? shadow_valid_1 : shadow_valid_2, vd = against == 0 ? view_dirty_1 : view_dirty_2;
                    const int sc = against == 0 ? shadow_cur_1 : shadow_cur_2, vc = against == 0 ? view_cur_1 : view_cur_2;
                    if (!(sv && net->shadow_valid && sc == net->shadow_cur)) { out(net->shadow[0]); out(net->shadow[1]); }
                    if (!(!vd && !net->view_dirty && vc == net->cell_view_cur)) { out(net->cell_view[0]); out(net->cell_view[1]); }
                    HIP_TRY(hipMemsetAsync(net->verify_report, 0, 32, net->stream), SNN_ERR_BUFFER_WRITE);
                    hipLaunchKernelGGL(k_compare_table_alt, grid, dim3(256), 0, net->stream, table, base,
                                       against == 0 ? first : net->verify_third, net->verify_report, sk);
                    HIP_TRY(hipGetLastError(), SNN_ERR_QUEUE);
                    uint32_t r3[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    HIP_TRY(copy_sync(net, r3, net->verify_report, 32, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
                    differ[against] = r3[0];
                }
                net->verify_text += "; a third execution differs from the first in " + std::to_string(differ[0]) + " words, from the second in " +
                                    std::to_string(differ[1]) + (differ[0] && !differ[1] ? ": the FIRST execution was the odd one"
                                                                 : !differ[0] && differ[1] ? ": the SECOND execution was the odd one"
                                                                 : differ[0] && differ[1] ? ": no two executions agree" : "");
            }
            fprintf(stderr, "[snn verify] MISMATCH %s\n", net->verify_text.c_str());
            if (const char *path = getenv("SNN_AMD_VERIFY_LOG")) {
                if (FILE *f = fopen(path, "a")) { fprintf(f, "%s\n", net->verify_text.c_str()); fclose(f); }
            }
        }
    } else {
        net->stat_verify_skipped += 1;
    }
    return end_run(net, /*keep_stdp=*/true);
}
ABI_CATCH

int snn_step_begin_local(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->electrical && !net->chemical) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(begin_run(net, 1));
    // sparse shard handles: the INTERIOR slices of the step snn_step_begin opened (its border slices and the outgoing
    // segments are enqueued; nothing an interior row reads or writes travels) -- call it once the exchange is under way
    if (net->interior_pending) return step_interior(net);
    // Only when nothing of the previous step is still pending for these chunks: STDP rewrites W in
    // snn_step_end and needs the gathered spikes first, so with plasticity on the split is not taken.
    if (net->nn && !net->csr && !net->any_plasticity && !net->any_modulation && !net->drive_threshold && !net->local_inputs_done) {
        TRY(launch_inputs(net, INPUTS_LOCAL));
        net->local_inputs_done = true;
    }
    return SNN_OK;
}
ABI_CATCH

int snn_step_begin(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->electrical && !net->chemical) return SNN_OK;           // neuron/mod.rs:1217, 2672
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(begin_run(net, 1));
    if (net->nn) TRY(step_begin(net));
    TRY(launch_exchange_pack(net));
    if (!net->external_stream) HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}
ABI_CATCH

int snn_step_end(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->electrical && !net->chemical) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    if (!net->run_active) return fail(SNN_ERR_BAD_STATE, "snn_step_end without snn_step_begin");
    TRY(step_end(net));
    if (!net->external_stream) HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}
ABI_CATCH

int snn_refresh_begin(snn_network_t *net, int *needed) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized || !net->sharded) return fail(SNN_ERR_BAD_STATE, "not a finalized shard handle");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_exchange_plan(net));
    const bool stale = mirror_stale(net);
    if (needed) *needed = stale ? 1 : 0;
    if (!stale) return SNN_OK;
    TRY(refresh_pack(net));
    if (!net->external_stream) HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}
ABI_CATCH

int snn_refresh_end(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized || !net->sharded) return fail(SNN_ERR_BAD_STATE, "not a finalized shard handle");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(refresh_unpack(net));
    if (!net->external_stream) HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}
ABI_CATCH

int snn_exchange_plan_get(snn_network_t *net, snn_exchange_plan *plan) ABI_TRY
{
    if (!net || !plan) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->sharded) return fail(SNN_ERR_BAD_STATE, "not a shard handle (snn_network_finalize_shard)");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(ensure_exchange_plan(net));
    *plan = snn_exchange_plan{};
    plan->mode = net->x_mode;
    plan->n_shards = net->n_shards; plan->shard_index = net->shard_index; plan->shard_stride = net->shard_stride;
    plan->planes = net->x_planes;
    for (uint32_t s = 0; s < 4; ++s) plan->plane_id[s] = s < net->x_planes ? net->x_plane_id[s] : 0;
    if (net->x_mode == SNN_EXCHANGE_ALLGATHER) {
        plan->recv = net->wire;
        plan->send = net->wire + (size_t)net->shard_index * net->x_block_words;
        plan->send_words = net->x_block_words;
        plan->recv_words = net->x_block_words * net->n_shards;
    } else {
        // (inside a direct run, i.e. from the exchange function of snn_run_sharded_custom: the sets of the current step)
        plan->send = (net->direct_run && net->hx_par) ? net->halo_send_buf2 : net->halo_send_buf;
        plan->recv = (net->direct_run && net->hx_par) ? net->halo_recv_buf2 : net->halo_recv_buf;
        for (uint32_t p = 0; p < net->n_shards; ++p) {
            plan->send_words += net->x_send_words[p];
            plan->recv_words += net->x_recv_words[p];
        }
    }
    return SNN_OK;
}
ABI_CATCH

int snn_exchange_peers(snn_network_t *net, uint64_t *send_offset, uint64_t *send_words, uint64_t *recv_offset,
                       uint64_t *recv_words) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized || !net->sharded) return fail(SNN_ERR_BAD_STATE, "not a finalized shard handle");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(ensure_exchange_plan(net));
    for (uint32_t p = 0; p < net->n_shards; ++p) {
        if (send_offset) send_offset[p] = net->x_send_off[p];
        if (send_words) send_words[p] = net->x_send_words[p];
        if (recv_offset) recv_offset[p] = net->x_recv_off[p];
        if (recv_words) recv_words[p] = net->x_recv_words[p];
    }
    return SNN_OK;
}
ABI_CATCH

int snn_halo_needs(snn_network_t *net, uint32_t peer, uint32_t *indices, uint32_t capacity, uint32_t *count) ABI_TRY
{
    if (!net || !count) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized || !net->sharded || !net->csr) return fail(SNN_ERR_BAD_STATE, "halo plans belong to finalized CSR shard handles");
    if (peer >= net->n_shards) return fail(SNN_ERR_BAD_ARG, "peer out of range");
    if (net->halo_need.size() != net->n_shards) halo_reset(net);
    const auto &l = net->halo_need[peer];
    *count = (uint32_t)l.size();
    if (indices && capacity >= l.size() && !l.empty()) std::memcpy(indices, l.data(), l.size() * 4);
    return SNN_OK;
}
ABI_CATCH

int snn_halo_set_sends(snn_network_t *net, uint32_t peer, const uint32_t *indices, uint32_t count) ABI_TRY
{
    if (!net || (count && !indices)) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized || !net->sharded || !net->csr) return fail(SNN_ERR_BAD_STATE, "halo plans belong to finalized CSR shard handles");
    if (peer >= net->n_shards || peer == net->shard_index) return fail(SNN_ERR_BAD_ARG, "peer must be another shard");
    if (net->halo_send.size() != net->n_shards) halo_reset(net);
    for (uint32_t i = 0; i < count; ++i)
        if (indices[i] >= net->nn || !owns(net, indices[i]))
            return fail(SNN_ERR_DIM_MISMATCH, "a send list may only name neurons this handle owns");
    TRY(end_run(net));
    net->halo_send[peer].assign(indices, indices + count);
    net->halo_committed = false;
    net->x_dirty = true;
    return SNN_OK;
}
ABI_CATCH

int snn_cells_read(snn_network_t *net, uint32_t *indices, uint32_t capacity, uint32_t *count) ABI_TRY
{
    if (!net || !count) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (!net->cell_list_dev) {                       // every cell
        *count = net->nc;
        if (indices && capacity >= net->nc)
            for (uint32_t s = 0; s < net->nc; ++s) indices[s] = s;
        return SNN_OK;
    }
    *count = net->n_cells_listed;
    if (indices && capacity >= net->n_cells_listed && net->n_cells_listed)
        std::memcpy(indices, net->cell_list_host.data(), (size_t)net->n_cells_listed * 4);
    return SNN_OK;
}
ABI_CATCH

int snn_halo_commit(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized || !net->sharded || !net->csr) return fail(SNN_ERR_BAD_STATE, "halo plans belong to finalized CSR shard handles");
    if (net->halo_need.size() != net->n_shards) halo_reset(net);
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    net->halo_committed = true;
    net->x_dirty = true;
    return ensure_exchange_plan(net);
}
ABI_CATCH

int snn_set_collectives(const snn_collectives *table) ABI_TRY
{
    Rccl &r = rccl_state();
    static const Rccl resolved = r;               // what dlopen / dlsym found (possibly nothing), for the way back
    if (g_collective_users.load() != 0)
        return fail(SNN_ERR_BAD_STATE, "the collective table cannot change while a library-driven run or a list exchange is in progress");
    if (!table) {
        r = resolved;
        return SNN_OK;
    }
    if (!table->comm_count || !table->comm_user_rank || !table->all_gather || !table->send || !table->recv ||
        !table->group_start || !table->group_end)
        return fail(SNN_ERR_BAD_ARG, "every entry of the table must be set");
    // RCCL's own types at the call sites: an enum result and data type (int-sized), ncclComm_t and hipStream_t pointers
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(table->comm_count);
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(table->comm_user_rank);
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(table->all_gather);
    r.Send = reinterpret_cast<decltype(r.Send)>(table->send);
    r.Recv = reinterpret_cast<decltype(r.Recv)>(table->recv);
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(table->group_start);
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(table->group_end);
    r.GetErrorString = [](ncclResult_t) -> const char * { return "the host's collective reported a failure"; };
    r.replaced = true;
    return SNN_OK;
}
ABI_CATCH

int snn_comm_unique_id(void *id_128_bytes) ABI_TRY
{
    if (!id_128_bytes) return fail(SNN_ERR_BAD_ARG, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    RCCL_REAL(R, GetUniqueId);
    ncclUniqueId id;
    RCCL_TRY(R, R->GetUniqueId(&id));
    std::memcpy(id_128_bytes, &id, sizeof id);
    return SNN_OK;
}
ABI_CATCH

int snn_comm_init_rank(const void *id_128_bytes, int world_size, int rank, int device, void **nccl_comm) ABI_TRY
{
    if (!id_128_bytes || !nccl_comm) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(SNN_ERR_BAD_ARG, "rank must be in [0, world_size)");
    RCCL_REAL(R, CommInitRank);
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    ncclUniqueId id;
    std::memcpy(&id, id_128_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    RCCL_TRY(R, R->CommInitRank(&comm, world_size, id, rank));
    *nccl_comm = comm;
    return SNN_OK;
}
ABI_CATCH

int snn_comm_destroy(void *nccl_comm) ABI_TRY
{
    if (!nccl_comm) return SNN_OK;
    RCCL_REAL(R, CommDestroy);
    RCCL_TRY(R, R->CommDestroy(static_cast<ncclComm_t>(nccl_comm)));
    return SNN_OK;
}
ABI_CATCH

int snn_comm_count(void *nccl_comm, int *world_size, int *rank) ABI_TRY
{
    if (!nccl_comm || !world_size) return fail(SNN_ERR_BAD_ARG, "null argument");
    RCCL_LIB(R);
    if (!R->CommCount || !R->CommUserRank) return fail(SNN_ERR_BAD_STATE, "the collective table has no communicator queries");
    RCCL_TRY(R, R->CommCount(static_cast<ncclComm_t>(nccl_comm), world_size));
    if (rank) RCCL_TRY(R, R->CommUserRank(static_cast<ncclComm_t>(nccl_comm), rank));
    return SNN_OK;
}
ABI_CATCH

int snn_comm_exchange_halo_lists(snn_network_t *net, void *nccl_comm) ABI_TRY
{
    if (!net || !nccl_comm) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized || !net->sharded || !net->csr) return fail(SNN_ERR_BAD_STATE, "halo plans belong to finalized CSR shard handles");
    RCCL_LIB(R);
    CollectiveUser in_use;
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(comm_geometry(R, net, comm));
    TRY(end_run(net));
    if (net->halo_need.size() != net->n_shards) halo_reset(net);
    const uint32_t G = net->n_shards, me = net->shard_index;
    // (1) counts: row `me` of a G x G matrix, all-gathered; (2) the lists themselves, grouped send / recv
    hvec<uint32_t> counts((size_t)G * G, 0);
    for (uint32_t p = 0; p < G; ++p) counts[(size_t)me * G + p] = (uint32_t)net->halo_need[p].size();
    uint32_t *d_counts = nullptr;
    HIP_TRY(snn_malloc(&d_counts, counts.size() * 4), SNN_ERR_BUFFER_CREATE);
    int rc = SNN_OK;
    uint32_t *d_need = nullptr, *d_send = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_counts); if (d_need) (void)hipFree(d_need); if (d_send) (void)hipFree(d_send); };
#define HALO_STEP(expr) do { rc = (expr); if (rc) { cleanup(); return rc; } } while (0)
    auto hip_ok = [&](hipError_t e, int code, const char *what) { return e == hipSuccess ? SNN_OK : fail(code, std::string(what) + ": " + hipGetErrorString(e)); };
    auto nccl_ok = [&](ncclResult_t r, const char *what) { return r == ncclSuccess ? SNN_OK : fail(SNN_ERR_QUEUE, std::string(what) + ": " + R->GetErrorString(r)); };
    HALO_STEP(hip_ok(copy_sync(net, d_counts, counts.data(), counts.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE, "counts upload"));
    HALO_STEP(nccl_ok(R->AllGather(d_counts + (size_t)me * G, d_counts, G, ncclUint32, comm, net->stream), "ncclAllGather(counts)"));
    HALO_STEP(hip_ok(hipStreamSynchronize(net->stream), SNN_ERR_WAIT, "counts wait"));
    HALO_STEP(hip_ok(copy_sync(net, counts.data(), d_counts, counts.size() * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ, "counts download"));
    hvec<uint64_t> need_off(G + 1, 0), send_off(G + 1, 0);
    for (uint32_t p = 0; p < G; ++p) {
        need_off[p + 1] = need_off[p] + counts[(size_t)me * G + p];        // what I ask of p
        send_off[p + 1] = send_off[p] + counts[(size_t)p * G + me];        // what p asks of me
    }
    hvec<uint32_t> need_flat(need_off[G]), send_flat(send_off[G]);
    for (uint32_t p = 0; p < G; ++p) std::copy(net->halo_need[p].begin(), net->halo_need[p].end(), need_flat.begin() + need_off[p]);
    HALO_STEP(hip_ok(snn_malloc(&d_need, std::max<size_t>(need_flat.size() * 4, 256)), SNN_ERR_BUFFER_CREATE, "hipMalloc"));
    HALO_STEP(hip_ok(snn_malloc(&d_send, std::max<size_t>(send_flat.size() * 4, 256)), SNN_ERR_BUFFER_CREATE, "hipMalloc"));
    if (!need_flat.empty())
        HALO_STEP(hip_ok(copy_sync(net, d_need, need_flat.data(), need_flat.size() * 4, hipMemcpyHostToDevice), SNN_ERR_BUFFER_WRITE, "lists upload"));
    HALO_STEP(nccl_ok(R->GroupStart(), "ncclGroupStart"));
    {
        // a failure inside the group still closes it (a dangling group would swallow every later RCCL call of this thread)
        int first = SNN_OK;
        for (uint32_t p = 0; p < G && !first; ++p) {
            if (p == me) continue;
            if (need_off[p + 1] > need_off[p])
                first = nccl_ok(R->Send(d_need + need_off[p], need_off[p + 1] - need_off[p], ncclUint32, (int)p, comm, net->stream), "ncclSend(list)");
            if (!first && send_off[p + 1] > send_off[p])
                first = nccl_ok(R->Recv(d_send + send_off[p], send_off[p + 1] - send_off[p], ncclUint32, (int)p, comm, net->stream), "ncclRecv(list)");
        }
        const ncclResult_t closed = R->GroupEnd();
        HALO_STEP(first);
        HALO_STEP(nccl_ok(closed, "ncclGroupEnd"));
    }
    HALO_STEP(hip_ok(hipStreamSynchronize(net->stream), SNN_ERR_WAIT, "lists wait"));
    if (!send_flat.empty())
        HALO_STEP(hip_ok(copy_sync(net, send_flat.data(), d_send, send_flat.size() * 4, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ, "lists download"));
#undef HALO_STEP
    cleanup();
    for (uint32_t p = 0; p < G; ++p) {
        if (p == me) continue;
        TRY(snn_halo_set_sends(net, p, send_flat.data() + send_off[p], (uint32_t)(send_off[p + 1] - send_off[p])));
    }
    return snn_halo_commit(net);
}
ABI_CATCH

int snn_p2p_local(snn_network_t *net, uint64_t *recv0, uint64_t *recv1, uint64_t *flags, uint64_t *recv_offsets, uint64_t *recv_counts) ABI_TRY
{
    if (!net || !recv0 || !recv1 || !flags) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized || !net->sharded || !net->csr) return fail(SNN_ERR_BAD_STATE, "the peer form belongs to finalized CSR shard handles");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_exchange_plan(net));
    if (!net->peer_capable || !net->p2p_recv[0])
        return fail(SNN_ERR_BAD_STATE, "the peer form needs a committed halo plan in which something travels");
    *recv0 = reinterpret_cast<uint64_t>(net->p2p_recv[0]);
    *recv1 = reinterpret_cast<uint64_t>(net->p2p_recv[1]);
    *flags = reinterpret_cast<uint64_t>(net->p2p_flags);
    for (uint32_t p = 0; p < net->n_shards; ++p) {
        if (recv_offsets) recv_offsets[p] = net->x_recv_off[p];
        if (recv_counts) recv_counts[p] = net->halo_need[p].size();
    }
    return SNN_OK;
}
ABI_CATCH

int snn_p2p_connect(snn_network_t *net, uint32_t peer, uint64_t peer_recv0, uint64_t peer_recv1, uint64_t peer_flags, uint64_t peer_recv_offset) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized || !net->sharded || !net->csr) return fail(SNN_ERR_BAD_STATE, "the peer form belongs to finalized CSR shard handles");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_exchange_plan(net));
    if (!net->peer_capable || !net->p2p_recv[0]) return fail(SNN_ERR_BAD_STATE, "the handle's exchange plan has no peer form");
    if (peer >= net->n_shards) return fail(SNN_ERR_BAD_ARG, "peer out of range");
    if (!peer_recv0 || !peer_recv1 || !peer_flags) return fail(SNN_ERR_BAD_ARG, "null peer address");
    snn_network::P2pPeer &pp = net->p2p_peers[peer];
    pp.recv[0] = peer_recv0; pp.recv[1] = peer_recv1; pp.flags = peer_flags; pp.recv_offset = peer_recv_offset; pp.set = true;
    net->p2p_connected = false;                   // until snn_p2p_commit
    net->x_agreed = false;
    return SNN_OK;
}
ABI_CATCH

int snn_p2p_commit(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized || !net->sharded || !net->csr) return fail(SNN_ERR_BAD_STATE, "the peer form belongs to finalized CSR shard handles");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    TRY(ensure_exchange_plan(net));
    if (!net->peer_capable || !net->p2p_recv[0]) return fail(SNN_ERR_BAD_STATE, "the handle's exchange plan has no peer form");
    return p2p_build_tables(net);
}
ABI_CATCH

int snn_p2p_ipc_export(snn_network_t *net, void *handles_3x64_bytes) ABI_TRY
{
    if (!net || !handles_3x64_bytes) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->p2p_recv[0]) return fail(SNN_ERR_BAD_STATE, "the handle's exchange plan has no peer form");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    hipIpcMemHandle_t *h = static_cast<hipIpcMemHandle_t *>(handles_3x64_bytes);
    HIP_TRY(hipIpcGetMemHandle(&h[0], net->p2p_recv[0]), SNN_ERR_BUFFER_CREATE);
    HIP_TRY(hipIpcGetMemHandle(&h[1], net->p2p_recv[1]), SNN_ERR_BUFFER_CREATE);
    HIP_TRY(hipIpcGetMemHandle(&h[2], net->p2p_flags), SNN_ERR_BUFFER_CREATE);
    return SNN_OK;
}
ABI_CATCH

int snn_p2p_ipc_import(int device, const void *handles_3x64_bytes, uint64_t *recv0, uint64_t *recv1, uint64_t *flags) ABI_TRY
{
    if (!handles_3x64_bytes || !recv0 || !recv1 || !flags) return fail(SNN_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    const hipIpcMemHandle_t *h = static_cast<const hipIpcMemHandle_t *>(handles_3x64_bytes);
    void *p[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; ++i) {
        const hipError_t e = hipIpcOpenMemHandle(&p[i], h[i], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipIpcCloseMemHandle(p[j]);       // nothing half-opened is left behind
            return fail(SNN_ERR_BUFFER_CREATE, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
        }
    }
    *recv0 = reinterpret_cast<uint64_t>(p[0]); *recv1 = reinterpret_cast<uint64_t>(p[1]); *flags = reinterpret_cast<uint64_t>(p[2]);
    return SNN_OK;
}
ABI_CATCH

int snn_p2p_ipc_close(int device, uint64_t recv0, uint64_t recv1, uint64_t flags) ABI_TRY
{
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    int rc = SNN_OK;
    for (uint64_t a : {recv0, recv1, flags}) {
        if (!a) continue;
        const hipError_t e = hipIpcCloseMemHandle(reinterpret_cast<void *>(a));
        if (e != hipSuccess && rc == SNN_OK) rc = fail(SNN_ERR_BAD_ARG, std::string("hipIpcCloseMemHandle: ") + hipGetErrorString(e));
    }
    return rc;
}
ABI_CATCH

int snn_exchange(snn_network_t *net, void *nccl_comm) ABI_TRY
{
    if (!net || !nccl_comm) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized || !net->sharded) return fail(SNN_ERR_BAD_STATE, "not a finalized shard handle");
    RCCL_LIB(R);
    CollectiveUser in_use;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(ensure_exchange_plan(net));
    TRY(comm_geometry(R, net, static_cast<ncclComm_t>(nccl_comm)));
    TRY(enqueue_exchange(R, net, static_cast<ncclComm_t>(nccl_comm), net->stream));
    if (!net->external_stream) HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    return SNN_OK;
}
ABI_CATCH

int snn_exchange_noop(void *, void *) ABI_TRY { return 0; } ABI_CATCH

// peer form: a poll that gave up (a peer that never stored, or never finished) has left the handle in the middle of a step
static int p2p_outcome(snn_network *net)
{
    if (net->p2p_failed && net->p2p_failed[0]) {
        net->p2p_failed[0] = 0u;
        if (net->p2p_done_blocks) (void)hipMemsetAsync(net->p2p_done_blocks, 0, 256, net->stream);     // block counter + device abort word
        return fail(SNN_ERR_WAIT, "peer form: a neighbour's values or its done counter did not arrive within the spin limit; the handle is mid-step");
    }
    return SNN_OK;
}

int snn_run_sharded_custom(snn_network_t *net, snn_exchange_fn exchange, void *user, uint64_t iterations) ABI_TRY
{
    if (!net || !exchange) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized || !net->sharded) return fail(SNN_ERR_BAD_STATE, "not a finalized shard handle");
    if (iterations == 0 || net->n_tot == 0) return SNN_OK;
    if (!net->electrical && !net->chemical) return SNN_OK;
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(begin_run(net, iterations));
    // the direct form of a sparse halo run (rows gather from the received segments): with the do-nothing transport, or when
    // the caller's function reads the segment pointers of every step from snn_exchange_plan_get ("halo_direct" 2)
    if (mirror_stale(net)) {
        TRY(refresh_pack(net));
        if (exchange(user, net->stream) != 0) return fail(SNN_ERR_QUEUE, "the caller's exchange function failed");
        TRY(refresh_unpack(net));
    }
    if (net->halo_direct == 2 || exchange == &snn_exchange_noop) TRY(direct_begin(net));
    int rc = SNN_OK;
    for (uint64_t it = 0; it < iterations && rc == SNN_OK; ++it) {
        if (net->nn) rc = step_begin(net);
        if (!rc) rc = launch_exchange_pack(net);
        if (!rc) rc = step_interior(net);         // sparse handles: interior slices are enqueued before the host-side exchange
        if (!rc && exchange(user, net->stream) != 0) rc = fail(SNN_ERR_QUEUE, "the caller's exchange function failed");
        if (!rc) rc = step_end(net);
        if (!rc && net->profile && (net->ev_used >= 8192 || net->ev_used_pl >= 8192)) rc = collect_profile(net);
    }
    const int rc_end = direct_end(net);
    if (rc) return rc;
    TRY(rc_end);
    TRY(end_run(net, /*keep_stdp=*/true));
    return p2p_outcome(net);
}
ABI_CATCH

// The ranks of a communicator agree on HOW they exchange before the first step of a run: a rank that decided from its
// own state alone (no graph yet, lists committed by hand while the peers' are not, other synapse kinds) would skip or
// mismatch the collective its peers block in -- a hang instead of an error.  One word per rank is all-gathered: graph
// form, whether the halo lists are committed, and (after the plan is built) the planes on the wire.
static int agree_on_exchange(Rccl *R, snn_network *net, ncclComm_t comm, void *nccl_comm)
{
    const uint32_t G = net->n_shards, me = net->shard_index;
    // (the handle's for good: hipFree waits for EVERY stream of the device -- with the ranks emulated as threads of one process a
    // rank that freed these words here waited for a neighbour's first peer-form launch, which was polling for this rank's: a
    // give-up after the spin limit, 3 of 16 runs of the emulated-rank tests in round 6.  And a run path has no business
    // synchronising the device.)
    if (net->agree_words_dev && net->agree_words_cap < G) { (void)hipFree(net->agree_words_dev); net->agree_words_dev = nullptr; }
    if (!net->agree_words_dev) {
        HIP_TRY(snn_malloc(&net->agree_words_dev, std::max<size_t>((size_t)G * 4, 256)), SNN_ERR_BUFFER_CREATE);
        net->agree_words_cap = G;
    }
    uint32_t *const d_words = net->agree_words_dev;
    hvec<uint32_t> words(G, 0);
    auto gather = [&](uint32_t mine) -> int {
        words.assign(G, 0);
        words[me] = mine;
        if (copy_sync(net, d_words, words.data(), (size_t)G * 4, hipMemcpyHostToDevice) != hipSuccess) return fail(SNN_ERR_BUFFER_WRITE, "agreement upload");
        const ncclResult_t r = R->AllGather(d_words + me, d_words, 1, ncclUint32, comm, net->stream);
        if (r != ncclSuccess) return fail(SNN_ERR_QUEUE, std::string("ncclAllGather(agreement): ") + R->GetErrorString(r));
        if (hipStreamSynchronize(net->stream) != hipSuccess) return fail(SNN_ERR_WAIT, "agreement wait");
        if (copy_sync(net, words.data(), d_words, (size_t)G * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(SNN_ERR_BUFFER_READ, "agreement download");
        return SNN_OK;
    };
    int rc = gather((net->csr ? 1u : 0u) | (net->halo_committed ? 2u : 0u) | (net->block_mode ? 4u : 0u));
    bool all_committed = true;
    for (uint32_t p = 0; p < G && !rc; ++p) {
        if ((words[p] & 5u) != (words[me] & 5u))
            rc = fail(SNN_ERR_BAD_STATE, "the ranks of this communicator hold different graph forms (dense / CSR / by-lattice shards)");
        all_committed = all_committed && (words[p] & 2u);
    }
    // sparse handles: unless EVERY rank has its lists committed, every rank trades them now (a rank without rows asks for nothing)
    if (!rc && net->csr && !all_committed) rc = snn_comm_exchange_halo_lists(net, nccl_comm);
    if (!rc) rc = ensure_exchange_plan(net);
    if (!rc) {
        uint32_t mask = (uint32_t)net->x_mode << 8;
        for (uint32_t s = 0; s < net->x_planes; ++s) mask |= 1u << net->x_plane_id[s];
        // bit 30: this rank will step in the PEER form (connected, committed, "halo_peer" on).  A rank that is not, next to one
        // that is, would post ncclSend / ncclRecv nobody answers while the other polls granules nobody stores.
        const bool peer_form = net->halo_peer && net->p2p_connected && net->p2p_recv[0] && net->peer_capable && net->halo_direct && net->csr_plan_direct &&
                               csr_fast_step(net);
        mask |= peer_form ? 0x40000000u : 0u;
        // bit 29: nothing travels to or from this rank (a lone shard, an empty one, rows that read no neighbour): it posts no
        // collective and polls no granule either way, so its vote on the peer form does not count
        bool exchanges = net->x_mode == SNN_EXCHANGE_ALLGATHER;
        for (uint32_t p = 0; p < G && !exchanges; ++p) exchanges = net->x_send_words[p] || net->x_recv_words[p];
        const uint32_t idle = exchanges ? 0u : 0x20000000u;
        // bit 31: this rank's mirror lacks a plane of the plan (mirror_stale).  A plane can only go missing when the plan
        // grows -- which is when this agreement runs -- and whether it is missing is rank-local history (host-driven steps,
        // attributes written on some ranks only): if ANY rank is stale, EVERY rank sends its current state once before the
        // first step (idempotent), instead of one rank entering a collective its peers never post.
        rc = gather(mask | idle | (mirror_stale(net) ? 0x80000000u : 0u));
        bool any_stale = false;
        for (uint32_t p = 0; p < G && !rc; ++p) {
            any_stale = any_stale || (words[p] >> 31) != 0u;
            if ((words[p] & 0x1FFFFFFFu) != (mask & 0x1FFFFFFFu))
                rc = fail(SNN_ERR_BAD_STATE, "the ranks of this communicator disagree on the exchange (synapse kinds / transmitter types / mode)");
            else if (!idle && !(words[p] & 0x20000000u) && (words[p] & 0x40000000u) != (mask & 0x40000000u))
                rc = fail(SNN_ERR_BAD_STATE, "the ranks of this communicator disagree on the peer form: rank " + std::to_string(p) +
                                                 ((words[p] & 0x40000000u) ? " is connected (snn_p2p_commit, \"halo_peer\" 1), this rank is not"
                                                                           : " is not connected, this rank is") + "; every rank or none");
        }
        if (!rc) net->refresh_agreed = any_stale;
    }
    if (!rc) net->x_agreed = true;
    return rc;
}

int snn_run_sharded(snn_network_t *net, void *nccl_comm, uint64_t iterations) ABI_TRY
{
    if (!net || !nccl_comm) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized || !net->sharded) return fail(SNN_ERR_BAD_STATE, "not a finalized shard handle");
    if (iterations == 0 || net->n_tot == 0) return SNN_OK;
    if (!net->electrical && !net->chemical) return SNN_OK;
    RCCL_LIB(R);
    CollectiveUser in_use;
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(comm_geometry(R, net, comm));
    // (SNN_AMD_ALWAYS_AGREE: the agreement also at world size 1 -- lets a one-GPU test walk the code every rank runs)
    if ((net->n_shards > 1 || getenv("SNN_AMD_ALWAYS_AGREE")) && (!net->x_agreed || net->x_dirty)) {
        TRY(end_run(net));
        TRY(agree_on_exchange(R, net, comm, nccl_comm));
    }
    TRY(begin_run(net, iterations));
    TRY(ensure_comm_objects(net));
    // The own-rows part of step t + 1's input pass does not read what the exchange of step t delivers: it is enqueued
    // before the compute stream waits for the collective (dense handles, no weight updates pending in step_end).
    const bool split = !net->csr && !net->any_plasticity && !net->any_modulation && !net->drive_threshold && net->n_shards > 1;
    // a halo plan in which nothing travels (a lone shard, or peers that read nothing of each other): no collective, no
    // cross-stream events -- the step is its kernels
    bool travels = net->x_mode == SNN_EXCHANGE_ALLGATHER;
    for (uint32_t p = 0; p < net->n_shards && !travels; ++p) travels = net->x_send_words[p] || net->x_recv_words[p];
    if (mirror_stale(net) || net->refresh_agreed) {
        // a plane the plan needs was not on the wire so far (here, or on some other rank: agree_on_exchange): the owners'
        // current state travels once before the first step
        net->refresh_agreed = false;
        TRY(refresh_pack(net));
        if (travels) {
            HIP_TRY(hipEventRecord(net->ev_packed, net->stream), SNN_ERR_QUEUE);
            HIP_TRY(hipStreamWaitEvent(net->comm_stream, net->ev_packed, 0), SNN_ERR_QUEUE);
            TRY(enqueue_exchange(R, net, comm, net->comm_stream));
            HIP_TRY(hipEventRecord(net->ev_exchanged, net->comm_stream), SNN_ERR_QUEUE);
            HIP_TRY(hipStreamWaitEvent(net->stream, net->ev_exchanged, 0), SNN_ERR_QUEUE);
        }
        TRY(refresh_unpack(net));
    }
    if (travels) TRY(direct_begin(net));          // sparse halo runs: the rows gather from the received segments themselves
    int rc = SNN_OK;
    auto hip_step = [&](hipError_t e, int code) { if (!rc && e != hipSuccess) rc = fail(code, hipGetErrorString(e)); };
    for (uint64_t it = 0; it < iterations && rc == SNN_OK; ++it) {
        if (net->nn) rc = step_begin(net);
        if (!rc) rc = launch_exchange_pack(net);
        if (!rc && travels && !net->peer_run) {          // (peer form: the border rows stored into the peers themselves)
            hip_step(hipEventRecord(net->ev_packed, net->stream), SNN_ERR_QUEUE);
            hip_step(hipStreamWaitEvent(net->comm_stream, net->ev_packed, 0), SNN_ERR_QUEUE);
            if (!rc) rc = enqueue_exchange(R, net, comm, net->comm_stream);
            hip_step(hipEventRecord(net->ev_exchanged, net->comm_stream), SNN_ERR_QUEUE);
        }
        if (!rc) rc = step_interior(net);         // sparse handles: the interior slices run while the halo travels
        if (!rc && split && it + 1 < iterations && net->nn && !net->local_inputs_done) {
            // step_end below advances the clock and the spike trains; the LOCAL chunks read neither
            rc = launch_inputs(net, INPUTS_LOCAL);
            net->local_inputs_done = true;
        }
        if (!rc && travels && !net->peer_run) hip_step(hipStreamWaitEvent(net->stream, net->ev_exchanged, 0), SNN_ERR_QUEUE);
        if (!rc) rc = step_end(net);
        if (!rc && net->profile && (net->ev_used >= 8192 || net->ev_used_pl >= 8192)) rc = collect_profile(net);
    }
    const int rc_end = direct_end(net);
    if (rc) return rc;
    TRY(rc_end);
    TRY(end_run(net, /*keep_stdp=*/true));
    return p2p_outcome(net);
}
ABI_CATCH

int snn_set_stream(snn_network_t *net, void *hip_stream) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    if (hip_stream) {
        net->stream = static_cast<hipStream_t>(hip_stream);
        net->external_stream = true;
    } else {
        net->stream = net->own_stream;
        net->external_stream = false;
    }
    return SNN_OK;
}
ABI_CATCH

int snn_synchronize(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    return end_run(net);
}
ABI_CATCH

int snn_stream(snn_network_t *net, void **hip_stream) ABI_TRY
{
    if (!net || !hip_stream) return fail(SNN_ERR_BAD_ARG, "null argument");
    *hip_stream = net->stream;
    return SNN_OK;
}
ABI_CATCH

int snn_history_steps(const snn_network_t *net, uint64_t *steps) ABI_TRY
{
    if (!net || !steps) return fail(SNN_ERR_BAD_ARG, "null argument");
    *steps = net->hist_steps;
    return SNN_OK;
}
ABI_CATCH

int snn_get_voltage_history(snn_network_t *net, uint32_t id, float *dst, size_t count) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l) return fail(SNN_ERR_BAD_ARG, "unknown lattice id");
    if (!net->want_vhist) return fail(SNN_ERR_BAD_STATE, "voltage history is off");
    if (count != net->hist_steps * l->count) return fail(SNN_ERR_DIM_MISMATCH, "history size mismatch");
    if (count == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const float *src = l->spike_train ? net->st_vhist + (l->first - net->nn) : net->vhist + l->first;
    const size_t pitch = (size_t)(l->spike_train ? net->c_pad : net->n_pad) * 4;
    HIP_TRY(copy2d_sync(net, dst, (size_t)l->count * 4, src, pitch, (size_t)l->count * 4, net->hist_steps, hipMemcpyDeviceToHost),
            SNN_ERR_BUFFER_READ);
    return SNN_OK;
}
ABI_CATCH

int snn_get_spike_history(snn_network_t *net, uint32_t id, uint8_t *dst, size_t count) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    const LatticeInfo *l = find_lattice(net, id);
    if (!l || l->spike_train) return fail(SNN_ERR_BAD_ARG, "spike history exists for neuron lattices");
    if (!net->want_raster) return fail(SNN_ERR_BAD_STATE, "spike history is off");
    if (count != net->hist_steps * l->count) return fail(SNN_ERR_DIM_MISMATCH, "history size mismatch");
    if (count == 0) return SNN_OK;
    if (!dst) return fail(SNN_ERR_BAD_ARG, "dst is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(end_run(net));
    const size_t words = net->n_pad / 64;
    hvec<unsigned long long> host(net->hist_steps * words);
    HIP_TRY(copy_sync(net, host.data(), net->raster, host.size() * 8, hipMemcpyDeviceToHost), SNN_ERR_BUFFER_READ);
    for (uint64_t s = 0; s < net->hist_steps; ++s)
        for (uint32_t i = 0; i < l->count; ++i) {
            const uint32_t q = l->first + i;
            dst[s * l->count + i] = (uint8_t)((host[s * words + (q >> 6)] >> (q & 63)) & 1ull);
        }
    return SNN_OK;
}
ABI_CATCH

int snn_set_option(snn_network_t *net, const char *name, int value) ABI_TRY
{
    if (!net || !name) return fail(SNN_ERR_BAD_ARG, "null argument");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    if (net->finalized) TRY(end_run(net));            // pending deferred updates belong to the old setting
    const std::string n(name);
    if (n == "fused_step") net->fused_step = value != 0;
    else if (n == "pinned_copies") net->pinned_copies = (value >= 0 && value <= 2) ? value : 1;
    else if (n == "dense_close") net->dense_close = value != 0;
    else if (n == "cells_in_step") net->cells_in_step = value != 0;
    else if (n == "update_packs") net->update_packs = value != 0;
    else if (n == "update_all_planes") net->update_all_planes = (value >= 0 && value <= 3) ? value : 1;
    else if (n == "persistent_stdp") net->persistent_stdp = value != 0;
    else if (n == "halo_direct") net->halo_direct = (value >= 0 && value <= 2) ? value : 1;
    else if (n == "halo_peer") { net->halo_peer = value != 0; net->x_agreed = false; }
    else if (n == "halo_peer_delay") net->peer_delay = (uint32_t)std::max(0, std::min(value, 64));
    else if (n == "halo_peer_spin_limit") net->p2p_spin_limit = value > 0 ? (uint32_t)value : (1u << 26);
    else if (n == "csr_xcd_bands") net->csr_xcd_bands = value != 0;
    else if (n == "csr_image") net->csr_image = value != 0;
    else if (n == "resident_quarters") net->resident_quarters = value != 0;
    else if (n == "defer_rstdp") net->defer_rstdp = value != 0;
    else if (n == "defer_stdp") net->defer_stdp = (value >= 0 && value <= 3) ? value : 1;
    else if (n == "uniform_params") { net->uniform_params = value != 0; net->uni_dirty = true; }
    else if (n == "persistent_run") { net->persistent_run = value != 0; net->run_probed_grid = 0; }
    else if (n == "persistent_chem") net->persistent_chem = value != 0;
    else if (n == "run_resident_spin_limit") net->run_spin_limit = value > 0 ? (uint32_t)std::min<long long>(value, 0x7FFFFFFF) : RUN_RESIDENT_SPIN_LIMIT;
    else if (n == "run_resident_fault_step") net->run_fault_step = (uint32_t)std::max<long long>(value, 0);
    else if (n == "run_timing") net->run_timing_opt = value != 0;
    else if (n == "input_shape") net->force_shape = (value == 1 || value == 2) ? value : 0;
    else if (n == "stdp_columns_form") net->stdp_columns_form = value == 1 ? 1 : 0;
    else if (n == "stdp_small") net->stdp_small = value != 0;
    else if (n == "verify") net->verify = value != 0;
    else if (n == "verify_fault") net->verify_fault = value > 0 ? (uint32_t)value : 0u;
    else if (n == "run_resident_chunk_steps") net->run_chunk_steps = value >= 4 ? (uint32_t)std::min<long long>(value, 1 << 20) : (1u << 20);
    else return fail(SNN_ERR_BAD_ARG, "unknown option '" + n + "'");
    net->shadow_valid = false;
    return SNN_OK;
}
ABI_CATCH

int snn_get_stat(snn_network_t *net, const char *name, uint64_t *value) ABI_TRY
{
    if (!net || !name || !value) return fail(SNN_ERR_BAD_ARG, "null argument");
    const std::string n(name);
    if (n == "persistent_run_launches") *value = net->stat_run_launches;
    else if (n == "persistent_run_steps") *value = net->stat_run_steps;
    else if (n == "persistent_run_stdp_steps") *value = net->stat_run_stdp_steps;
    else if (n == "persistent_run_fallbacks") *value = net->stat_run_fallbacks;
    else if (n == "halo_direct_steps") *value = net->stat_direct_steps;
    else if (n == "halo_peer_steps") *value = net->stat_peer_steps;
    else if (n == "persistent_run_external_stream") *value = net->stat_run_external_stream;
    else if (n == "steps_dense_one_launch") *value = net->stat_steps_dense_one_launch;
    else if (n == "steps_dense_close") *value = net->stat_steps_dense_close;
    else if (n == "steps_sparse_one_launch") *value = net->stat_steps_sparse_one_launch;
    else if (n == "steps_sparse_image") *value = net->stat_steps_sparse_image;
    else if (n == "image_staged_slices") *value = net->img_staged_slices;
    else if (n == "image_staged_slices_direct") *value = net->img_staged_slices_direct;
    else if (n == "steps_sparse_split") *value = net->stat_steps_sparse_split;
    else if (n == "steps_two_kernel") *value = net->stat_steps_two_kernel;
    else if (n == "shadow_refreshes") *value = net->stat_shadow_refreshes;
    else if (n == "view_refreshes") *value = net->stat_view_refreshes;
    else if (n == "history_regrows") *value = net->stat_history_regrows;
    else if (n == "verify_runs") *value = net->stat_verify_runs;
    else if (n == "verify_mismatches") *value = net->stat_verify_mismatches;
    else if (n == "verify_skipped") *value = net->stat_verify_skipped;
    else if (n == "run_timing_poll") *value = net->run_timing_last[0];
    else if (n == "run_timing_barrier") *value = net->run_timing_last[1];
    else if (n == "run_timing_turns") *value = net->run_timing_last[2];
    else if (n == "run_timing_update") *value = net->run_timing_last[3];
    else if (n == "run_timing_steps") *value = net->run_timing_steps;
    else return fail(SNN_ERR_BAD_ARG, "unknown statistic '" + n + "'");
    return SNN_OK;
}
ABI_CATCH

int snn_profile_enable(snn_network_t *net, int enable) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    net->profile = enable ? 1 : 0;
    return SNN_OK;
}
ABI_CATCH
int snn_profile_reset(snn_network_t *net) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    HIP_TRY(hipStreamSynchronize(net->stream), SNN_ERR_WAIT);
    net->ev_used = 0; net->prof_launches = 0; net->prof_ms = 0.0;
    net->ev_used_pl = 0; net->prof_launches_pl = 0; net->prof_ms_pl = 0.0;
    return SNN_OK;
}
ABI_CATCH
int snn_profile_read(snn_network_t *net, uint64_t *launches, double *total_ms) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(collect_profile(net));
    if (launches) *launches = net->prof_launches;
    if (total_ms) *total_ms = net->prof_ms;
    return SNN_OK;
}
ABI_CATCH
int snn_profile_read_plasticity(snn_network_t *net, uint64_t *steps, double *total_ms) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    TRY(collect_profile(net));
    if (steps) *steps = net->prof_launches_pl;
    if (total_ms) *total_ms = net->prof_ms_pl;
    return SNN_OK;
}
ABI_CATCH

int snn_set_synthetic_drive(snn_network_t *net, uint64_t seed, float fraction, float voltage) ABI_TRY
{
    if (!net) return fail(SNN_ERR_BAD_ARG, "net is null");
    if (!(fraction >= 0.0f && fraction <= 1.0f)) return fail(SNN_ERR_BAD_ARG, "fraction must be in [0, 1]");
    HIP_TRY(hipSetDevice(net->device), SNN_ERR_GET_DEVICE);
    if (net->finalized) TRY(end_run(net));
    net->drive_seed = seed;
    net->drive_threshold = (uint32_t)std::min<double>(4294967295.0, (double)fraction * 4294967296.0);
    net->drive_voltage = voltage;
    net->shadow_valid = false;
    return SNN_OK;
}
ABI_CATCH

int snn_input_kernel_bytes(const snn_network_t *net, uint64_t *bytes) ABI_TRY
{
    if (!net || !bytes) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (!net->finalized) return fail(SNN_ERR_BAD_STATE, "network not finalized");
    if (net->csr) {
        // sparse: index + weight of every stored synapse; the one-launch step (k_step_csr) also is the neuron update of its
        // OWNED rows (the hole rows of a range-set shard move nothing): + S bytes of state per neuron ("8 B x nnz +
        // S x N"), S by model -- the words update_neuron reads and writes per step: state read + written, parameters and the
        // averager's count read, spike flag and raster bit written; Izhikevich 60 B as SURVEY 8d counts it -- plus, with
        // chemical synapses, 44 B per live transmitter type (t read + written, two kinetics parameters, receptor r read +
        // written, g, e, current, two flag words).  ... and, when the spike-train cells advance in the same launch: + 28 B per
        // cell (seed read + written, voltage and spike flag written, last firing time read, the 8-byte view entry written).
        // ALGORITHMIC bytes: what the step has to move, not what the counters saw.
        static const uint32_t S_MODEL[8] = {60, 68, 140, 64, 44, 80, 88, 64};
        uint64_t per_neuron = net->model == SNN_MODEL_CUSTOM ? (uint64_t)8 * custom::NVARS + 28
                                                            : S_MODEL[net->model == SNN_MODEL_BCM_IZHIKEVICH ? 0 : net->model & 7];
        if (net->model == SNN_MODEL_BCM_IZHIKEVICH) per_neuron += 36;       // activities, window clock, period, spike count
        if (net->chemical) per_neuron += (uint64_t)44 * net->n_live;
        *bytes = (uint64_t)8 * net->nnz + (fused_csr_step_applies(net) ? per_neuron * net->n_owned : 0u);
        if (fused_csr_step_applies(net) && cells_ride_allowed(net))
            *bytes += (uint64_t)28 * (net->cell_list_dev ? net->n_cells_listed : net->nc);
        return SNN_OK;
    }
    uint64_t b = (uint64_t)4 * net->n_tot * net->n_loc;                    // dense: every weight of the shard, read once
    if (net->any_modulation && net->defer_rstdp) {
        // k_inputs_rstdp: the internal edges of a reward-modulated lattice are read AND rewritten, weight and trace --
        // 16 B per synapse instead of 4
        for (const auto &l : net->lattices) {
            if (l.slot >= net->rm_on_host.size() || !(net->rm_on_host[l.slot] & RM_DO_MODULATION)) continue;
            const uint32_t c0 = std::max(l.first, net->q0), c1 = std::min(l.first + l.count, net->q1);
            if (c1 > c0) b += (uint64_t)12 * l.count * (c1 - c0);
        }
    }
    *bytes = b;
    return SNN_OK;
}
ABI_CATCH

int snn_probe_bandwidth(int device, uint64_t bytes, int repeats, double *read_gbps, double *copy_gbps) ABI_TRY
{
    if (!read_gbps || !copy_gbps) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (bytes < (1u << 20) || repeats <= 0) return fail(SNN_ERR_BAD_ARG, "need >= 1 MiB and >= 1 repeat");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    const size_t n4 = bytes / 16;
    void *a = nullptr, *b = nullptr;
    float *sink = nullptr;
    HIP_TRY(snn_malloc(&a, n4 * 16), SNN_ERR_BUFFER_CREATE);
    if (snn_malloc(&b, n4 * 16) != hipSuccess || snn_malloc(&sink, 256) != hipSuccess) {
        (void)hipFree(a);
        if (b) (void)hipFree(b);
        return fail(SNN_ERR_BUFFER_CREATE, "probe allocation failed");
    }
    int rc = SNN_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipMemset(a, 0, n4 * 16) != hipSuccess || hipMemset(b, 0, n4 * 16) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        rc = fail(SNN_ERR_QUEUE, "probe setup failed");
    auto timed = [&](bool copy, double *out) {
        const unsigned blocks = 256 * 8;     // 8 workgroups per CU, grid-stride
        for (int warm = 0; warm < 2; ++warm) {
            if (copy) hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, (probe_v4f *)b, n4);
            else hipLaunchKernelGGL(k_probe_read, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, n4, sink);
        }
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < repeats; ++r) {
            if (copy) hipLaunchKernelGGL(k_probe_copy, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, (probe_v4f *)b, n4);
            else hipLaunchKernelGGL(k_probe_read, dim3(blocks), dim3(256), 0, 0, (const probe_v4f *)a, n4, sink);
        }
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(SNN_ERR_WAIT, "probe kernel failed"); return; }
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *out = (double)(copy ? 2 : 1) * (double)(n4 * 16) * repeats / (ms * 1e-3) / 1e9;
    };
    if (rc == SNN_OK) timed(false, read_gbps);
    if (rc == SNN_OK) timed(true, copy_gbps);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(sink);
    return rc;
}
ABI_CATCH

int snn_probe_math(int device, int which, const float *in, float *out, size_t count) ABI_TRY
{
    if (!in || !out) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (which < 0 || which > 2) return fail(SNN_ERR_BAD_ARG, "unknown function selector");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    if (count == 0) return SNN_OK;
    float *di = nullptr, *dout = nullptr;
    HIP_TRY(snn_malloc(&di, count * 4), SNN_ERR_BUFFER_CREATE);
    if (snn_malloc(&dout, count * 4) != hipSuccess) { (void)hipFree(di); return fail(SNN_ERR_BUFFER_CREATE, "hipMalloc failed"); }
    int rc = SNN_OK;
    if (hipMemcpy(di, in, count * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SNN_ERR_BUFFER_WRITE, "upload failed");
    if (rc == SNN_OK) {
        hipLaunchKernelGGL(k_probe_math, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, which, di, dout, count);
        if (hipDeviceSynchronize() != hipSuccess) rc = fail(SNN_ERR_WAIT, "probe kernel failed");
    }
    if (rc == SNN_OK && hipMemcpy(out, dout, count * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SNN_ERR_BUFFER_READ, "download failed");
    (void)hipFree(di);
    (void)hipFree(dout);
    return rc;
}
ABI_CATCH

int snn_probe_math_bits(int device, int which, uint32_t first, uint32_t stride, float y, float *out, size_t count) ABI_TRY
{
    if (!out) return fail(SNN_ERR_BAD_ARG, "null argument");
    if (which < 0 || which > 6) return fail(SNN_ERR_BAD_ARG, "unknown function selector");
    HIP_TRY(hipSetDevice(device), SNN_ERR_GET_DEVICE);
    if (count == 0) return SNN_OK;
    float *dout = nullptr;
    HIP_TRY(snn_malloc(&dout, count * 4), SNN_ERR_BUFFER_CREATE);
    int rc = SNN_OK;
    hipLaunchKernelGGL(k_probe_math_bits, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, which, first, stride, y, dout, count);
    if (hipDeviceSynchronize() != hipSuccess) rc = fail(SNN_ERR_WAIT, "probe kernel failed");
    if (rc == SNN_OK && hipMemcpy(out, dout, count * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SNN_ERR_BUFFER_READ, "download failed");
    (void)hipFree(dout);
    return rc;
}
ABI_CATCH

} // extern "C"
