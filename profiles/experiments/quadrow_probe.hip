// EXPERIMENT (not product code): does a "quad-row" layout of the dense synapse matrix stream as fast as the row-major
// one?  Motivation ("STDP under load"): in the row-major matrix a COLUMN is one 4-byte word per
// 128-byte line, so the STDP update of a spiking neuron's incoming edges dirties n_tot lines.  If the 4 rows of a row
// group are stored next to each other per column -- element (p, q) at ((p / 4) * n_cols + q) * 4 + p % 4 -- a column
// holds 16 contiguous bytes per row group (4x fewer lines touched), a lane that owns ONE column reads 4 consecutive rows
// with one dwordx4, and a wavefront's load is still 1 KiB contiguous.  This program times both access shapes with the
// arithmetic of k_inputs_dense's plain path (absent-edge test, g * (v_pre - v_post) * w, sequential sum per column).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o quadrow_probe quadrow_probe.hip && ./quadrow_probe [side]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 256;

__device__ __forceinline__ float acc_if_edge(float acc, float term, float w) { return (w == w) ? acc + term * w : acc; }

// A: row-major, lane = 4 adjacent columns of one row per load (the product's streaming shape)
__global__ __launch_bounds__(256) void k_rowmajor(const float *W, size_t ld, const float *v, uint32_t n, float *part)
{
    __shared__ float s_val[CHUNK];
    const uint32_t chunk = blockIdx.y, p0 = chunk * CHUNK, tid = threadIdx.x;
    const uint32_t tile = (blockIdx.x + blockIdx.y) % gridDim.x;
    const uint32_t q = tile * 1024 + tid * 4;
    const float *wrow = W + (size_t)p0 * ld + q;
    constexpr int B = 8;
    float wa[B][4], wb[B][4];
#pragma unroll
    for (int u = 0; u < B; ++u) { const v4f x = __builtin_nontemporal_load((const v4f *)(wrow + (size_t)u * ld)); wa[u][0] = x.x; wa[u][1] = x.y; wa[u][2] = x.z; wa[u][3] = x.w; }
    s_val[tid] = v[p0 + tid];
    __syncthreads();
    float vq[4], acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) vq[j] = v[q + j];
    const float g = 10.0f;
    uint32_t r = 0;
    auto body = [&](uint32_t row, const float (&w)[4]) {
        const float vp = s_val[row];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc_if_edge(acc[j], g * (vp - vq[j]), w[j]);
    };
    for (; r + 3 * B <= CHUNK; r += 2 * B) {
#pragma unroll
        for (int u = 0; u < B; ++u) { const v4f x = __builtin_nontemporal_load((const v4f *)(wrow + (size_t)(r + B + u) * ld)); wb[u][0] = x.x; wb[u][1] = x.y; wb[u][2] = x.z; wb[u][3] = x.w; }
#pragma unroll
        for (int u = 0; u < B; ++u) body(r + u, wa[u]);
#pragma unroll
        for (int u = 0; u < B; ++u) { const v4f x = __builtin_nontemporal_load((const v4f *)(wrow + (size_t)(r + 2 * B + u) * ld)); wa[u][0] = x.x; wa[u][1] = x.y; wa[u][2] = x.z; wa[u][3] = x.w; }
#pragma unroll
        for (int u = 0; u < B; ++u) body(r + B + u, wb[u]);
    }
#pragma unroll
    for (int u = 0; u < B; ++u) body(r + u, wa[u]);
    r += B;
    for (; r < CHUNK; r += B) {
#pragma unroll
        for (int u = 0; u < B; ++u) { const v4f x = __builtin_nontemporal_load((const v4f *)(wrow + (size_t)(r + u) * ld)); wb[u][0] = x.x; wb[u][1] = x.y; wb[u][2] = x.z; wb[u][3] = x.w; }
#pragma unroll
        for (int u = 0; u < B; ++u) body(r + u, wb[u]);
    }
    float *dst = part + (size_t)chunk * n + q;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = acc[j];
}

// B: quad-row layout, lane = ONE column, one dwordx4 = 4 consecutive rows of it; a workgroup walks COLS_PER_WG / 256
// sub-tiles of 256 columns for its chunk (the staged presynaptic values serve all of them)
template <int SUBTILES>
__global__ __launch_bounds__(256) void k_quadrow(const float *W, size_t ncols_padded, const float *v, uint32_t n, float *part)
{
    __shared__ float s_val[CHUNK];
    const uint32_t chunk = blockIdx.y, p0 = chunk * CHUNK, tid = threadIdx.x;
    const uint32_t tile = (blockIdx.x + blockIdx.y) % gridDim.x;
    s_val[tid] = v[p0 + tid];
    __syncthreads();
    const float g = 10.0f;
    constexpr int B = 8;                                     // row groups in flight = 32 rows
    constexpr int GROUPS = CHUNK / 4;
#pragma unroll 1
    for (int t = 0; t < SUBTILES; ++t) {
        const uint32_t q = (tile * SUBTILES + t) * 256 + tid;
        const v4f *col = (const v4f *)W + (size_t)(p0 / 4) * ncols_padded + q;      // unit (group, q)
        const float vq = v[q];
        float acc = 0.0f;
        v4f wa[B], wb[B];
#pragma unroll
        for (int u = 0; u < B; ++u) wa[u] = __builtin_nontemporal_load(col + (size_t)u * ncols_padded);
        int gidx = 0;
        auto body = [&](int grp, const v4f &w) {
            const float *sv = &s_val[grp * 4];
            acc = acc_if_edge(acc, g * (sv[0] - vq), w.x);
            acc = acc_if_edge(acc, g * (sv[1] - vq), w.y);
            acc = acc_if_edge(acc, g * (sv[2] - vq), w.z);
            acc = acc_if_edge(acc, g * (sv[3] - vq), w.w);
        };
        for (; gidx + 3 * B <= GROUPS; gidx += 2 * B) {
#pragma unroll
            for (int u = 0; u < B; ++u) wb[u] = __builtin_nontemporal_load(col + (size_t)(gidx + B + u) * ncols_padded);
#pragma unroll
            for (int u = 0; u < B; ++u) body(gidx + u, wa[u]);
#pragma unroll
            for (int u = 0; u < B; ++u) wa[u] = __builtin_nontemporal_load(col + (size_t)(gidx + 2 * B + u) * ncols_padded);
#pragma unroll
            for (int u = 0; u < B; ++u) body(gidx + B + u, wb[u]);
        }
#pragma unroll
        for (int u = 0; u < B; ++u) body(gidx + u, wa[u]);
        gidx += B;
        for (; gidx < GROUPS; gidx += B) {
#pragma unroll
            for (int u = 0; u < B; ++u) wb[u] = __builtin_nontemporal_load(col + (size_t)(gidx + u) * ncols_padded);
#pragma unroll
            for (int u = 0; u < B; ++u) body(gidx + u, wb[u]);
        }
        part[(size_t)chunk * n + q] = acc;
    }
}

__global__ void k_fill(float *p, size_t n, float v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }

int main(int argc, char **argv)
{
    const uint32_t side = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 256;
    const uint32_t n = side * side;                          // multiple of 1024 for the sides used (128, 192?, 256)
    if (n % 1024) { std::fprintf(stderr, "side*side must be a multiple of 1024\n"); return 1; }
    const size_t ld = (size_t)n + 64;                        // the product's de-aligned row stride
    float *W = nullptr, *v = nullptr, *part = nullptr;
    CK(hipMalloc((void **)&W, ld * n * sizeof(float)));
    CK(hipMalloc((void **)&v, (size_t)n * sizeof(float)));
    CK(hipMalloc((void **)&part, (size_t)(n / CHUNK) * n * sizeof(float)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, W, ld * n, 1.0f);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, v, (size_t)n, -60.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = 4.0 * (double)n * n / 1e9;
    const int reps = 12;
    auto timeit = [&](const char *name, auto launch) -> int {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms / reps, gb / (ms / reps) * 1e3);
        return 0;
    };
    const dim3 grid(n / 1024, n / CHUNK);
    for (int round = 0; round < 2; ++round) {
        if (timeit("row-major, 4 cols/lane", [&] { hipLaunchKernelGGL(k_rowmajor, grid, dim3(256), 0, 0, W, ld, v, n, part); })) return 1;
        if (timeit("quad-row, 1 col x 4 rows", [&] { hipLaunchKernelGGL(k_quadrow<4>, grid, dim3(256), 0, 0, W, ld, v, n, part); })) return 1;
        if (timeit("quad-row, 2 sub-tiles/WG", [&] { hipLaunchKernelGGL(k_quadrow<2>, dim3(n / 512, n / CHUNK), dim3(256), 0, 0, W, ld, v, n, part); })) return 1;
    }
    return 0;
}
