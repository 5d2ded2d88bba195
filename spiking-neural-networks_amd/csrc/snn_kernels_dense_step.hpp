// k_inputs_dense_close -- the streamed dense step in ONE launch (round 5).
//
// The two-kernel step of a streamed matrix is k_inputs_dense (chunk partials of every column) + k_update (second level of
// the canonical sum, the neuron's step).  At BASELINE configs[2] (128 x 128 Hodgkin-Huxley + AMPA, a 1 GiB matrix) the pass
// takes 161 us, the update 9.8 us and the two kernel boundaries another 7 -- a tenth of the step spent outside the kernel
// that is bound by HBM.  Here the workgroups of a column tile count themselves off as they finish (one agent-scope atomic
// per workgroup); the LAST one of a tile finds every chunk partial of its columns in memory and runs the update of those
// columns itself, with the code of k_update, while the other tiles still stream.
//
// Visibility inside the launch: the workgroups of a tile sit on different XCDs, whose L2s do not see each other's dirty lines.
// The partials therefore leave as agent-scope relaxed atomic stores (write-through) and the closing workgroup reads them with
// agent-scope relaxed atomic loads (not served from a stale line) -- the granule traffic of k_run_resident.  NO fence: the first
// form of this kernel released and acquired with __threadfence(), i.e. an L2 write-back and an L2 invalidate per wavefront, and
// the pass took 680 instead of 161 us (profiles/r05/c3_close_with_fences_kernel_stats.csv).  Everything else the update touches was
// written by an earlier launch.  The pass reads S(t) from a shadow of the
// exchange buffer and the update writes S(t+1) to the buffer and to the other shadow (as k_step_resident does): a tile that
// closes early must not disturb the presynaptic values the other tiles still stage.
//
// Order of the sums: unchanged -- partials per chunk in memory, combined in ascending chunk order by ONE thread per column
// (combine_tile_columns).  Bit-identical to the two-kernel step (tests/test_gpu_dense_close.py).
#pragma once
#include "snn_kernels_inputs.hpp"
#include "snn_kernels_update.hpp"

namespace snn {

struct DenseStepArgs {
    InputsArgs in;              // in.xbuf = the shadow holding S(t)
    UpdateArgs up;              // up.n.xbuf = the same shadow; up.xout = exchange buffer; up.xout2 = other shadow
    uint32_t *tile_done;        // [column tiles] workgroups of the tile that have stored their partials; 0 between launches
};

// Second-level sums of the VEC columns a thread of the closing workgroup owns (columns THREADS apart), every plane the step needs:
// the loads of a batch of ALL its columns are in flight together, the adds of each column stay strictly ascending from 0.0f (the
// order of combine_partials).  One closing workgroup per tile is all the parallelism the tail has; column after column it would
// wait VEC x planes x n_chunks / 16 dependent round trips (64 at BASELINE configs[1]), this way planes x n_chunks / 16.
template <int VEC, int THREADS>
__device__ __forceinline__ void combine_tile_columns(const UpdateArgs &a, uint32_t ql0, RegisterSums (&out)[VEC])
{
    constexpr uint32_t B = 16;
    uint32_t col[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) col[j] = min(ql0 + (uint32_t)j * THREADS, a.ld - 1u);      // (columns past the row: value unused)
#pragma unroll
    for (int pl = 0; pl < 1 + K_TYPES; ++pl) {
        const bool on = pl == 0 ? a.electrical != 0 : (a.chemical != 0 && ((a.live_mask >> (pl - 1)) & 1u) != 0u);   // launch-uniform
        float sum[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum[j] = 0.0f;
        if (on) {
            const float *plane = pl == 0 ? a.part_i : a.part_t + (size_t)(pl - 1) * a.n_chunks * a.ld;
            uint32_t c = 0;
            for (; c + B <= a.n_chunks; c += B) {
                float v[VEC][B];
#pragma unroll
                for (uint32_t u = 0; u < B; ++u) {
                    const float *row = plane + (size_t)(c + u) * a.ld;       // wave-uniform base + the lane's 32-bit column offset
#pragma unroll
                    for (int j = 0; j < VEC; ++j) v[j][u] = __hip_atomic_load(row + col[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j)
#pragma unroll
                    for (uint32_t u = 0; u < B; ++u) sum[j] += v[j][u];
            }
            for (; c < a.n_chunks; ++c) {
                const float *row = plane + (size_t)c * a.ld;
#pragma unroll
                for (int j = 0; j < VEC; ++j) sum[j] += __hip_atomic_load(row + col[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (pl == 0) out[j].i = sum[j];
            else out[j].t[pl - 1] = sum[j];
        }
    }
}

template <int MODEL, bool ELEC, bool CHEM, int STREAM, int NT>
__global__ __launch_bounds__(InputsShape<STREAM>::THREADS, STREAM == 1 ? 3 : 4) void k_inputs_dense_close(const DenseStepArgs a)
{
    using S = InputsShape<STREAM>;
    __shared__ uint32_t s_last;
    inputs_dense_pass<ELEC, CHEM, STREAM, NT, 0, /*AGENT_STORES=*/true>(a.in);

    // every thread's partials have been acknowledged by the memory side before the workgroup is counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t tile = (blockIdx.x + blockIdx.y) % gridDim.x;          // as inputs_dense_pass maps it
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(a.tile_done + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = before + 1u == gridDim.y;
        if (last) __hip_atomic_store(a.tile_done + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        s_last = last ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u) return;

    // ---- the tile's columns: second level of the canonical sum + the neuron's step (k_update's code) ----
    const UpdateArgs &u = a.up;
    RegisterSums sums[S::VEC];
    combine_tile_columns<S::VEC, S::THREADS>(u, tile * S::TILE + threadIdx.x, sums);
#pragma unroll 1
    for (int j = 0; j < S::VEC; ++j) {
        const uint32_t ql = tile * S::TILE + (uint32_t)j * S::THREADS + threadIdx.x;
        if (tile * S::TILE + (uint32_t)j * S::THREADS >= u.ld) break;     // (workgroup-uniform: the raster's ballot below is whole)
        uint32_t spike = 0u;
        RegisterSums mine = sums[0];                 // (selected, not indexed: the sums stay in registers)
#pragma unroll
        for (int k = 1; k < S::VEC; ++k)
            if (j == k) mine = sums[k];
        if (ql < u.n_loc) spike = update_neuron_at<MODEL>(u, ql, mine, u.clock, u.vhist_row);
        if (u.spike_row) {
            const unsigned long long word = __ballot(spike != 0);
            if ((threadIdx.x & 63u) == 0u && ql < u.ld && u.q0 + ql < u.n.n_pad) u.spike_row[(u.q0 + ql) >> 6] = word;
        }
    }
}

} // namespace snn
