// Experiment: what does one "turn" of k_run_resident cost -- 64 rows of  sum += (gq * (vp - vq)) * w  for one wavefront per
// SIMD (the others idle), voltages broadcast from LDS, weights in registers -- and which part of it: the dependent adds, the
// packed products, the LDS reads?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o probe chain_turn_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(256) void k_turn(const float *wsrc, float *out, unsigned long long *clocks, int reps)
{
    __shared__ __attribute__((aligned(16))) float s_v[64];
    const uint32_t lane = threadIdx.x & 63u;
    if (threadIdx.x < 64) s_v[threadIdx.x] = 0.25f * threadIdx.x;
    float w[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) w[r] = wsrc[r * 64 + lane];
    __syncthreads();
    const float vq = 0.5f * lane, gq = 1.25f;
    const v2f vq2 = {vq, vq}, gq2 = {gq, gq};
    float acc = 0.0f;
    uint32_t zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    const v4f *pre = reinterpret_cast<const v4f *>(s_v + zero);
    const unsigned long long t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
        if (VARIANT == 0) {              // the 64 dependent adds alone
#pragma unroll
            for (int r = 0; r < 64; ++r) acc += w[r];
        } else if (VARIANT == 1) {       // packed products formed inside the chain (the kernel's loop)
#pragma unroll
            for (int r = 0; r < 64; r += 4) {
                const v4f vp = pre[r >> 2];
                const v2f p0 = (gq2 * (v2f{vp.x, vp.y} - vq2)) * v2f{w[r], w[r + 1]};
                const v2f p1 = (gq2 * (v2f{vp.z, vp.w} - vq2)) * v2f{w[r + 2], w[r + 3]};
                acc += p0.x; acc += p0.y; acc += p1.x; acc += p1.y;
            }
        } else if (VARIANT == 2) {       // scalar products inside the chain
#pragma unroll
            for (int r = 0; r < 64; r += 4) {
                const v4f vp = pre[r >> 2];
                const float e[4] = {vp.x, vp.y, vp.z, vp.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += (gq * (e[k] - vq)) * w[r + k];
            }
        } else if (VARIANT == 3) {       // packed products only (no chain): throughput of the product part
            v2f s2 = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 64; r += 4) {
                const v4f vp = pre[r >> 2];
                const v2f p0 = (gq2 * (v2f{vp.x, vp.y} - vq2)) * v2f{w[r], w[r + 1]};
                const v2f p1 = (gq2 * (v2f{vp.z, vp.w} - vq2)) * v2f{w[r + 2], w[r + 3]};
                s2 = s2 + p0 * p1;       // one packed op per 4 rows to keep them alive
            }
            acc += s2.x + s2.y;
        } else if (VARIANT == 5 || VARIANT == 6) {       // products of 16 (5) / 32 (6) rows stage by stage (independent instructions back to back), then their adds
            constexpr int B = VARIANT == 5 ? 16 : 32;
#pragma unroll
            for (int r0 = 0; r0 < 64; r0 += B) {
                v4f vp[B / 4];
                v2f d[B / 2];
#pragma unroll
                for (int k = 0; k < B / 4; ++k) vp[k] = pre[(r0 >> 2) + k];
#pragma unroll
                for (int k = 0; k < B / 4; ++k) { d[2 * k] = v2f{vp[k].x, vp[k].y} - vq2; d[2 * k + 1] = v2f{vp[k].z, vp[k].w} - vq2; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < B / 2; ++k) d[k] = gq2 * d[k];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < B / 2; ++k) d[k] = d[k] * v2f{w[r0 + 2 * k], w[r0 + 2 * k + 1]};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < B / 2; ++k) { acc += d[k].x; acc += d[k].y; }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (VARIANT == 4) {       // 2 independent half chains (NOT the canonical order): is it latency or issue?
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int r = 0; r < 32; ++r) { a0 += w[r]; a1 += w[32 + r]; }
            acc += a0 + a1;
        }
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

int main()
{
    float *w, *out; unsigned long long *clk;
    CHECK(hipMalloc(&w, 64 * 64 * 4)); CHECK(hipMalloc(&out, 16 * 256 * 4)); CHECK(hipMalloc(&clk, 16 * 8));
    CHECK(hipMemset(w, 0, 64 * 64 * 4));
    const int reps = 1000;
    const char *names[] = {"64 dependent adds", "packed products in the chain (kernel loop)", "scalar products in the chain",
                           "packed products only", "two independent 32-add chains", "staged products of 16 rows, then adds",
                           "staged products of 32 rows, then adds"};
    for (int v = 0; v < 7; ++v) {
        for (int run = 0; run < 2; ++run) {
            switch (v) {
            case 0: hipLaunchKernelGGL(k_turn<0>, dim3(16), dim3(256), 0, 0, w, out, clk, reps); break;
            case 1: hipLaunchKernelGGL(k_turn<1>, dim3(16), dim3(256), 0, 0, w, out, clk, reps); break;
            case 2: hipLaunchKernelGGL(k_turn<2>, dim3(16), dim3(256), 0, 0, w, out, clk, reps); break;
            case 3: hipLaunchKernelGGL(k_turn<3>, dim3(16), dim3(256), 0, 0, w, out, clk, reps); break;
            case 4: hipLaunchKernelGGL(k_turn<4>, dim3(16), dim3(256), 0, 0, w, out, clk, reps); break;
            case 5: hipLaunchKernelGGL(k_turn<5>, dim3(16), dim3(256), 0, 0, w, out, clk, reps); break;
            default: hipLaunchKernelGGL(k_turn<6>, dim3(16), dim3(256), 0, 0, w, out, clk, reps); break;
            }
            CHECK(hipDeviceSynchronize());
        }
        unsigned long long c; CHECK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
        printf("%-44s %.0f clocks per 64 rows\n", names[v], (double)c / reps);
    }
    return 0;
}
