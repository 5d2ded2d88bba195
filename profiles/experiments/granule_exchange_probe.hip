// Experiment: what does one all-to-all hand-off of a small state vector cost INSIDE a launch on MI355X?
// A persistent multi-step kernel for the 32x32 lattice (BASELINE configs[0]) needs, per step, every workgroup to see
// every neuron's new voltage.  Here G workgroups each publish `own` 8-byte {value, step} granules per step (agent-scope
// relaxed stores = write-through `sc1`) into the parity slot of the step and then poll-read ALL n granules (agent-scope
// relaxed loads) until every tag equals the step.  Prints microseconds per step.
//   hipcc --offload-arch=gfx950 -O3 -o granule_exchange_probe granule_exchange_probe.hip && ./granule_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_exchange(unsigned long long *slots, uint32_t n, uint32_t own, uint32_t steps, float *out, uint32_t *failed)
{
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, b = blockIdx.x;
    float acc = 0.0f;
    for (uint32_t t = 1; t <= steps; ++t) {
        unsigned long long *slot = slots + (size_t)(t & 1u) * n;
        if (tid < own) {
            const float v = (float)(b * own + tid) + acc * 1e-30f;
            const unsigned long long g = ((unsigned long long)t << 32) | __float_as_uint(v);
            __hip_atomic_store(slot + b * own + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (uint32_t i = tid; i < n; i += nthr) {
            unsigned long long g;
            uint32_t spins = 0;
            do {
                g = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (++spins > (1u << 22)) { *failed = 1; break; }
            } while ((uint32_t)(g >> 32) != t);
            acc += __uint_as_float((uint32_t)g);
        }
        __syncthreads();
    }
    if (tid == 0) out[b] = acc;
}

int main()
{
    unsigned long long *slots; float *out; uint32_t *failed;
    const uint32_t steps = 2000;
    CHECK(hipMalloc(&slots, 2 * 8192 * 8)); CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&failed, 4));
    struct Case { uint32_t groups, threads, n; } cases[] = {
        {16, 1024, 1024}, {16, 256, 1024}, {32, 512, 1024}, {64, 256, 1024}, {64, 256, 4096}, {32, 512, 4096}, {4, 256, 256}, {1, 256, 64},
    };
    for (const Case &c : cases) {
        CHECK(hipMemset(slots, 0, 2 * 8192 * 8)); CHECK(hipMemset(failed, 0, 4));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(slots, 0, 2 * 8192 * 8));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_exchange, dim3(c.groups), dim3(c.threads), 0, 0, slots, c.n, c.n / c.groups, steps, out, failed);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t f; CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
            printf("groups %3u x %4u threads, %4u granules: %.3f us per step%s\n", c.groups, c.threads, c.n, ms * 1000.0f / steps, f ? "  (SPIN LIMIT HIT)" : "");
        }
    }
    return 0;
}
