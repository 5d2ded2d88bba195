// Device-memory layout shared by the kernels and the host-side handle.
//
// HBM layout (per handle = per GPU):
//   W        dense synapse matrix of the local postsynaptic shard in QUAD-ROW order: the presynaptic rows are stored in
//            groups of 4; element (p, q) sits at ((p / 4) * ld + q) * 4 + p % 4, ld = round_up(n_local, 64) -- the 16
//            bytes at ((p / 4) * ld + q) * 16 are rows 4g .. 4g+3 of column q ("unit").  A lane that owns a column
//            reads 4 consecutive rows of it with one dwordx4, a wavefront (64 adjacent columns) 1 KiB contiguous, and
//            a column -- what STDP rewrites when a neuron spikes -- holds 16 contiguous bytes per row group, so its
//            update dirties 4x fewer 128-byte lines than in a row-major matrix (widx / wcount below).
//            An absent edge (None in the reference's Vec<Vec<Option<f32>>>, graph/mod.rs:139-150)
//            is stored as a quiet NaN, so `connections` needs no second matrix and the averager
//            count n_in[post] is precomputed once per graph upload.  4 B per synapse, read once
//            per step.
//   xbuf     the state every kernel reads of presynaptic neurons (the "mirror"), neuron-major planes indexed by
//            the GLOBAL neuron index: [plane][n_pad] 32-bit words; planes: 0 current_voltage (f32),
//            1 is_spiking (u32), 2..4 neurotransmitter concentration t of type AMPA/NMDA/GABA.  A shard handle
//            keeps the entries of its own neurons current itself and refreshes the remote entries it reads
//            from the per-step exchange (wire buffers, snn_kernels_exchange.hpp).
//   SoA      one array per reference struct field, length n_neurons padded to 256; per-type
//            attributes are stored type-major [3][n_pad] so that lanes = consecutive neurons
//            stay coalesced.
//   partials [n_chunks][ld] f32 per accumulated quantity: the fixed-order two-level sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snn {

constexpr int K_TYPES = 3;
constexpr int CHUNK = 256;        // canonical reduction chunk (SNN_REDUCTION_CHUNK)
constexpr int NUM_PLANES = 5;
enum Plane { PLANE_V = 0, PLANE_SPIKE = 1, PLANE_T0 = 2 };

// Dense matrix addressing (quad-row order, see the head of this file): word offset of element (row p, local column q)
__host__ __device__ __forceinline__ size_t widx(uint32_t p, uint32_t q, size_t ld)
{
    return (((size_t)(p >> 2)) * ld + q) * 4 + (p & 3u);
}
// words of a matrix with n_rows rows (rows are padded to whole groups of 4; padding rows hold the absent-edge NaN) plus
// the slack the streaming kernels may read past the last row group (one column tile of units, never used)
constexpr size_t WMATRIX_SLACK = 1024 * 4;
__host__ __device__ __forceinline__ size_t wcount(uint32_t n_rows, size_t ld)
{
    return (size_t)((n_rows + 3u) >> 2) * ld * 4 + WMATRIX_SLACK;
}

// Mirror addressing: global neuron index -> word offset inside xbuf.
struct XLayout {
    uint32_t stride;       // words per plane (= n_pad)
    __host__ __device__ __forceinline__ size_t at(uint32_t neuron, int plane) const
    {
        return (size_t)plane * stride + neuron;
    }
};

// Local row <-> global neuron of a handle.  A whole-population handle and a contiguous shard own [q0, q0 + n_loc):
// row l is neuron q0 + l.  A shard that owns a SET of ranges (the same slab of every lattice, sparse handles only)
// lays its rows out as the list of global 64-neuron blocks that hold at least one owned neuron: row l is neuron
// block[l / 64] * 64 + l % 64 and is active when bit l % 64 of mask[l / 64] is set -- a wavefront still maps to one
// aligned 64-block of the global index space (ballot word = raster word), rows of neurons owned elsewhere are holes.
struct RowMap {
    uint32_t q0;
    const uint32_t *block;               // null: contiguous
    const unsigned long long *mask;
    const uint32_t *local_row;           // [n_pad] global neuron -> local row, 0xFFFFFFFF when not owned (block form only)
    __device__ __forceinline__ uint32_t global_of(uint32_t l) const { return block ? block[l >> 6] * 64u + (l & 63u) : q0 + l; }
    __device__ __forceinline__ bool active(uint32_t l, uint32_t n_loc) const
    {
        return block ? ((mask[l >> 6] >> (l & 63u)) & 1ull) != 0ull : l < n_loc;
    }
    __device__ __forceinline__ uint32_t local_of(uint32_t q, uint32_t n_loc) const      // 0xFFFFFFFF: not a local neuron
    {
        if (block) return local_row[q];
        return (q >= q0 && q < q0 + n_loc) ? q - q0 : 0xFFFFFFFFu;
    }
};

// Per-cell PARAMETERS that hold one value for the whole population (the usual case: a lattice is populated from one base
// neuron) need not be streamed from HBM per cell and step.  The handle scans the read-only parameter arrays whenever
// attributes were set (k_uniform_scan) and keeps {is uniform, the value's bits} per parameter in a small device table;
// kernels read the table entry (one wave-uniform load) and fall back to the array otherwise.  Sparse / large
// populations gain: BASELINE configs[4] streams 28 B less per neuron and 24 B less per Poisson cell and step.
enum NeuronParam { NP_GAP = 0, NP_DT, NP_C_M, NP_V_TH, NP_A, NP_B, NP_C, NP_D, NP_TAU_M, NP_COUNT };
enum CellParam { CP_V_TH = 0, CP_V_RESTING, CP_DT, CP_K, CP_CHANCE, CP_REFR, CP_COUNT };
struct UniformTable {
    uint32_t flag[16];
    uint32_t bits[16];
};
// (The table is written by an earlier launch and its address is wave-uniform: read through the constant address space its
// entries arrive by SCALAR loads -- no vector-memory round trip in front of the load they decide about, and none at all when the
// parameter is uniform.  Round 6: as vector loads the flags of a step's parameters were a dependent trip of their own.)
typedef const __attribute__((address_space(4))) UniformTable *UniformTableK;
__device__ __forceinline__ float uload(const UniformTable *u, int slot, const float *arr, uint32_t i)
{
    const UniformTableK t = (UniformTableK)u;
    return t->flag[slot] ? __uint_as_float(t->bits[slot]) : arr[i];
}
__device__ __forceinline__ uint32_t uload(const UniformTable *u, int slot, const uint32_t *arr, uint32_t i)
{
    const UniformTableK t = (UniformTableK)u;
    return t->flag[slot] ? t->bits[slot] : arr[i];
}
// (the table entry by a vector load, as before round 6: the one-launch run, whose register budget is the weights', spills two
// registers when the entries of its prologue arrive in scalar registers)
__device__ __forceinline__ float uload_vector(const UniformTable *u, int slot, const float *arr, uint32_t i)
{
    return u->flag[slot] ? __uint_as_float(u->bits[slot]) : arr[i];
}
__device__ __forceinline__ uint32_t uload_vector(const UniformTable *u, int slot, const uint32_t *arr, uint32_t i)
{
    return u->flag[slot] ? u->bits[slot] : arr[i];
}

// A launch's arguments live in memory (the kernarg segment) and reach a wavefront by scalar loads, one 64-byte line at a time,
// WHERE the compiler needs a field: a kernel with 1.6 KB of pointers and far more of them than scalar registers (the one-launch
// steps: 200 spilled SGPRs) walks its lines in a chain of load-wait rounds, each a miss of the scalar cache at the start of a
// launch -- 6 to 25 rounds before the first vector load of k_step_resident_q (round 6, session 10: 6 500 shader clocks to the
// first ballot with transmitters, 2 500 without, whatever the order of the vector loads).  warm_kernel_arguments<BYTES>() asks
// for one word of every line at once and waits once: the rounds that follow hit.
template <uint32_t BYTES>
__device__ __forceinline__ void warm_kernel_arguments()
{
    const auto args = __builtin_amdgcn_kernarg_segment_ptr();
    // lines in fours (what lies behind the arguments for up to three lines more are the launch's implicit arguments, 256 bytes);
    // ONE statement per size, loads and wait together: the four scratch registers are written when a load returns, not where it
    // is issued -- nothing of the compiler's may sit between
    constexpr uint32_t LINES = ((BYTES + 63u) / 64u + 3u) & ~3u;
    static_assert(LINES >= 4u && LINES <= 64u, "extend the list below");
    uint32_t t0, t1, t2, t3;
    if constexpr (LINES == 4u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 8u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 12u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 16u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 20u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 24u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 28u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 32u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 36u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 40u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_load_dword %0, %4, 0x900\n\ts_load_dword %1, %4, 0x940\n\ts_load_dword %2, %4, 0x980\n\ts_load_dword %3, %4, 0x9c0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 44u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_load_dword %0, %4, 0x900\n\ts_load_dword %1, %4, 0x940\n\ts_load_dword %2, %4, 0x980\n\ts_load_dword %3, %4, 0x9c0\n\ts_load_dword %0, %4, 0xa00\n\ts_load_dword %1, %4, 0xa40\n\ts_load_dword %2, %4, 0xa80\n\ts_load_dword %3, %4, 0xac0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 48u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_load_dword %0, %4, 0x900\n\ts_load_dword %1, %4, 0x940\n\ts_load_dword %2, %4, 0x980\n\ts_load_dword %3, %4, 0x9c0\n\ts_load_dword %0, %4, 0xa00\n\ts_load_dword %1, %4, 0xa40\n\ts_load_dword %2, %4, 0xa80\n\ts_load_dword %3, %4, 0xac0\n\ts_load_dword %0, %4, 0xb00\n\ts_load_dword %1, %4, 0xb40\n\ts_load_dword %2, %4, 0xb80\n\ts_load_dword %3, %4, 0xbc0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 52u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_load_dword %0, %4, 0x900\n\ts_load_dword %1, %4, 0x940\n\ts_load_dword %2, %4, 0x980\n\ts_load_dword %3, %4, 0x9c0\n\ts_load_dword %0, %4, 0xa00\n\ts_load_dword %1, %4, 0xa40\n\ts_load_dword %2, %4, 0xa80\n\ts_load_dword %3, %4, 0xac0\n\ts_load_dword %0, %4, 0xb00\n\ts_load_dword %1, %4, 0xb40\n\ts_load_dword %2, %4, 0xb80\n\ts_load_dword %3, %4, 0xbc0\n\ts_load_dword %0, %4, 0xc00\n\ts_load_dword %1, %4, 0xc40\n\ts_load_dword %2, %4, 0xc80\n\ts_load_dword %3, %4, 0xcc0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 56u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_load_dword %0, %4, 0x900\n\ts_load_dword %1, %4, 0x940\n\ts_load_dword %2, %4, 0x980\n\ts_load_dword %3, %4, 0x9c0\n\ts_load_dword %0, %4, 0xa00\n\ts_load_dword %1, %4, 0xa40\n\ts_load_dword %2, %4, 0xa80\n\ts_load_dword %3, %4, 0xac0\n\ts_load_dword %0, %4, 0xb00\n\ts_load_dword %1, %4, 0xb40\n\ts_load_dword %2, %4, 0xb80\n\ts_load_dword %3, %4, 0xbc0\n\ts_load_dword %0, %4, 0xc00\n\ts_load_dword %1, %4, 0xc40\n\ts_load_dword %2, %4, 0xc80\n\ts_load_dword %3, %4, 0xcc0\n\ts_load_dword %0, %4, 0xd00\n\ts_load_dword %1, %4, 0xd40\n\ts_load_dword %2, %4, 0xd80\n\ts_load_dword %3, %4, 0xdc0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 60u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_load_dword %0, %4, 0x900\n\ts_load_dword %1, %4, 0x940\n\ts_load_dword %2, %4, 0x980\n\ts_load_dword %3, %4, 0x9c0\n\ts_load_dword %0, %4, 0xa00\n\ts_load_dword %1, %4, 0xa40\n\ts_load_dword %2, %4, 0xa80\n\ts_load_dword %3, %4, 0xac0\n\ts_load_dword %0, %4, 0xb00\n\ts_load_dword %1, %4, 0xb40\n\ts_load_dword %2, %4, 0xb80\n\ts_load_dword %3, %4, 0xbc0\n\ts_load_dword %0, %4, 0xc00\n\ts_load_dword %1, %4, 0xc40\n\ts_load_dword %2, %4, 0xc80\n\ts_load_dword %3, %4, 0xcc0\n\ts_load_dword %0, %4, 0xd00\n\ts_load_dword %1, %4, 0xd40\n\ts_load_dword %2, %4, 0xd80\n\ts_load_dword %3, %4, 0xdc0\n\ts_load_dword %0, %4, 0xe00\n\ts_load_dword %1, %4, 0xe40\n\ts_load_dword %2, %4, 0xe80\n\ts_load_dword %3, %4, 0xec0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    else if constexpr (LINES == 64u)
        asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\ts_load_dword %1, %4, 0x140\n\ts_load_dword %2, %4, 0x180\n\ts_load_dword %3, %4, 0x1c0\n\ts_load_dword %0, %4, 0x200\n\ts_load_dword %1, %4, 0x240\n\ts_load_dword %2, %4, 0x280\n\ts_load_dword %3, %4, 0x2c0\n\ts_load_dword %0, %4, 0x300\n\ts_load_dword %1, %4, 0x340\n\ts_load_dword %2, %4, 0x380\n\ts_load_dword %3, %4, 0x3c0\n\ts_load_dword %0, %4, 0x400\n\ts_load_dword %1, %4, 0x440\n\ts_load_dword %2, %4, 0x480\n\ts_load_dword %3, %4, 0x4c0\n\ts_load_dword %0, %4, 0x500\n\ts_load_dword %1, %4, 0x540\n\ts_load_dword %2, %4, 0x580\n\ts_load_dword %3, %4, 0x5c0\n\ts_load_dword %0, %4, 0x600\n\ts_load_dword %1, %4, 0x640\n\ts_load_dword %2, %4, 0x680\n\ts_load_dword %3, %4, 0x6c0\n\ts_load_dword %0, %4, 0x700\n\ts_load_dword %1, %4, 0x740\n\ts_load_dword %2, %4, 0x780\n\ts_load_dword %3, %4, 0x7c0\n\ts_load_dword %0, %4, 0x800\n\ts_load_dword %1, %4, 0x840\n\ts_load_dword %2, %4, 0x880\n\ts_load_dword %3, %4, 0x8c0\n\ts_load_dword %0, %4, 0x900\n\ts_load_dword %1, %4, 0x940\n\ts_load_dword %2, %4, 0x980\n\ts_load_dword %3, %4, 0x9c0\n\ts_load_dword %0, %4, 0xa00\n\ts_load_dword %1, %4, 0xa40\n\ts_load_dword %2, %4, 0xa80\n\ts_load_dword %3, %4, 0xac0\n\ts_load_dword %0, %4, 0xb00\n\ts_load_dword %1, %4, 0xb40\n\ts_load_dword %2, %4, 0xb80\n\ts_load_dword %3, %4, 0xbc0\n\ts_load_dword %0, %4, 0xc00\n\ts_load_dword %1, %4, 0xc40\n\ts_load_dword %2, %4, 0xc80\n\ts_load_dword %3, %4, 0xcc0\n\ts_load_dword %0, %4, 0xd00\n\ts_load_dword %1, %4, 0xd40\n\ts_load_dword %2, %4, 0xd80\n\ts_load_dword %3, %4, 0xdc0\n\ts_load_dword %0, %4, 0xe00\n\ts_load_dword %1, %4, 0xe40\n\ts_load_dword %2, %4, 0xe80\n\ts_load_dword %3, %4, 0xec0\n\ts_load_dword %0, %4, 0xf00\n\ts_load_dword %1, %4, 0xf40\n\ts_load_dword %2, %4, 0xf80\n\ts_load_dword %3, %4, 0xfc0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(args) : "memory");
    (void)t0; (void)t1; (void)t2; (void)t3;
}

// Pointers the per-neuron update kernels need.  All arrays are device memory.
struct NeuronArrays {
    // exchanged planes
    float *xbuf;
    XLayout xl;
    const UniformTable *uni;         // NeuronParam slots
    // common
    float *gap_conductance, *dt, *c_m, *v_th;
    int32_t *last_firing_time;
    // Izhikevich / LIF
    float *w_value, *a, *b, *c, *d, *tau_m;
    float *v_reset, *refractory_count, *tref, *leak_constant, *integration_constant, *e_l, *g_l;
    // quadratic integrate-and-fire / simple leaky integrate-and-fire
    float *qif_alpha, *qif_v_c, *slif_g, *slif_e;
    // adaptive (exponential) leaky integrate-and-fire
    float *adp_alpha, *adp_beta, *slope_factor;
    // variables of the generated model (snn_custom_model.hpp), one array each
    float *custom[32];
    // BCMActivity bookkeeping (BCMIzhikevichNeuron): average / current activity, window clock and length, period, spike count
    float *bcm_avg, *bcm_cur, *bcm_clock, *bcm_window;
    uint32_t *bcm_period, *bcm_num_spikes;
    // Hodgkin-Huxley
    float *m_state, *h_state, *n_state;
    float *m_alpha, *m_beta, *h_alpha, *h_beta, *n_alpha, *n_beta;
    float *g_na, *e_na, *g_k, *e_k, *g_k_leak, *e_k_leak;
    float *na_current, *k_current, *k_leak_current;
    uint32_t *was_increasing;
    // neurotransmitters [3][n_pad] (t lives in xbuf planes 2..4)
    float *nt_t_max, *nt_clearance, *nt_v_p, *nt_k_p;
    float *nt_custom[8], *rc_custom[8];   // variables of generated kinetics (custom_nt / custom_rc), [3][n_pad] each
    float *rx_custom[32];                 // variables of a generated receptor set (custom_receptors), [n_pad] each
    uint32_t *nt_flags;
    // receptors [3][n_pad]
    float *rc_g, *rc_e, *rc_mg, *rc_r, *rc_alpha, *rc_beta, *rc_current;
    uint32_t *rc_flags;
    uint32_t n_pad;
};

// Spike-train cells (presynaptic only), length n_cells padded to 256.
struct CellArrays {
    const UniformTable *uni;         // CellParam slots
    float *current_voltage, *v_th, *v_resting, *dt, *k;
    float *chance_of_firing, *rate, *step;
    uint32_t *seed, *is_spiking;
    uint32_t *refractoriness;   // NeuralRefractoriness: 0 DeltaDirac, 1 ExponentialDecay, 2 generated (custom_refr)
    float *custom[16];          // variables of a generated spike train (custom_st)
    float *refr_custom[8];      // variables of a generated refractoriness other than `decay` (= k)
    // BCMActivity bookkeeping (BCMPoissonNeuron)
    float *bcm_avg, *bcm_cur, *bcm_clock, *bcm_window;
    uint32_t *bcm_period, *bcm_num_spikes;
    // PresetSpikeTrain: firing times of cell s = preset_times[preset_ptr[s] .. preset_ptr[s + 1]); `step` holds
    // its internal_clock
    uint32_t *counter;
    const uint32_t *preset_ptr;
    const float *preset_times;
    int32_t *last_firing_time;
    float *nt_t, *nt_t_max, *nt_clearance, *nt_v_p, *nt_k_p;   // [3][c_pad]
    float *nt_custom[8];        // variables of generated neurotransmitter kinetics, [3][c_pad] each
    uint32_t *nt_flags;
    uint32_t *lattice_slot;     // [c_pad] -> spike-train lattice slot
    float *presyn_value;        // per-step presynaptic gap-junction value (see k_spike_train_view)
    uint32_t c_pad;
};

} // namespace snn
