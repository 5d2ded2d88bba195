// Does a kernel on a NON-BLOCKING stream, launched right after a blocking hipMemcpy(H2D, pageable) on the null stream, always
// see the copied data?  usage: h2d_race <seconds> <words>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
__global__ void check(const unsigned *d, unsigned n, unsigned want, unsigned *err, unsigned *first)
{
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
        if (d[i] != want) { if (atomicAdd(err, 1u) == 0u) { first[0] = want; first[1] = d[i]; first[2] = i; } }
}
int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const unsigned n = argc > 2 ? (unsigned)atoi(argv[2]) : 1024u;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;      // 0: hipMemcpy on the null stream; 1: hipMemcpyAsync on the kernel's stream + sync
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *d, *err, *first;
    hipMalloc(&d, n * 4); hipMalloc(&err, 4); hipMalloc(&first, 12);
    hipMemset(err, 0, 4); hipMemset(first, 0, 12); hipMemset(d, 0, n * 4);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long iters = 0;
    for (unsigned v = 1;; ++v) {
        for (unsigned i = 0; i < n; ++i) h[i] = v;
        if (mode == 0) { hipStreamSynchronize(s); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); }   // (the library's setters: stream sync, then the blocking copy)
        else { hipMemcpyAsync(d, h.data(), n * 4, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
        hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, s, d, n, v, err, first);
        ++iters;
        if ((v & 255u) == 0u) {
            hipStreamSynchronize(s);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) break;
        }
    }
    hipStreamSynchronize(s);
    unsigned e = 0, f[3] = {0, 0, 0};
    hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); hipMemcpy(f, first, 12, hipMemcpyDeviceToHost);
    printf("mode %d words %u iterations %llu stale_words %u first: want %u got %u at %u\n", mode, n, iters, e, f[0], f[1], f[2]);
    return 0;
}
