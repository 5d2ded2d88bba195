// Does a pageable host <-> device copy on a non-blocking stream complete when hipStreamSynchronize returns?  (test infrastructure)
//
// The five unexplained mismatches of rounds 3 - 5 all look like HOST memory of the test process holding bytes it should not:
// two words of a spike raster inside an oracle array, a voltage history that did not match its own final state.  Every getter of
// the library is `hipMemcpyAsync(pageable pointer, ..., stream)` + `hipStreamSynchronize(stream)` -- for the raster into a
// temporary the getter frees before it returns.  If the host-side half of such a copy could land AFTER the synchronisation
// returned (a staging thread of the runtime that has not been scheduled yet on an oversubscribed host), the bytes would land in
// memory that has been freed and handed out again: exactly that picture.  This program asks the runtime directly, at a rate no
// test campaign reaches:
//   D2H:  device buffer = pattern(i); copy into a fresh malloc'ed buffer; synchronise; (1) EARLY: is every word pattern(i)?
//         free it; malloc + zero a buffer of the same size (the allocator hands the same block back); keep it for `lag`
//         iterations; (2) LATE: is it still zero then?
//   H2D:  host buffer = pattern(i); copy to the device; synchronise; overwrite the host buffer at once; later read the device
//         buffer back through page-locked memory: (3) does it hold pattern(i)?
// T threads, each with its own non-blocking stream; run several processes at once for campaign E's load.
//   late_copy_probe <seconds> <threads> [max_bytes]      -> one JSON line; exit code 1 when any event was seen
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

static std::atomic<unsigned long long> g_iterations{0}, g_early{0}, g_late{0}, g_h2d{0};
static std::atomic<bool> g_stop{false};

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

struct Kept { uint32_t *p; size_t words; unsigned long long born; uint32_t pattern; };

static void worker(int id, size_t max_bytes)
{
    CK(hipSetDevice(0));
    hipStream_t stream;
    CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    uint32_t *dev = nullptr, *dev2 = nullptr, *pinned = nullptr;
    CK(hipMalloc(reinterpret_cast<void **>(&dev), max_bytes));
    CK(hipMalloc(reinterpret_cast<void **>(&dev2), max_bytes));
    CK(hipHostMalloc(reinterpret_cast<void **>(&pinned), max_bytes, hipHostMallocDefault));
    const unsigned lag = 32;
    std::vector<Kept> kept;
    uint32_t rng = 0x9E3779B9u * (uint32_t)(id + 1);
    for (unsigned long long i = 1; !g_stop.load(std::memory_order_relaxed); ++i) {
        rng = mix(rng + (uint32_t)i);
        const size_t words = 16 + rng % (max_bytes / 4 - 16);
        const uint32_t pattern = 0x80000000u | ((uint32_t)id << 24) | (uint32_t)(i & 0xFFFFFF);      // never 0
        // ---- device -> pageable host ----
        CK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dev), (int)pattern, words, stream));
        uint32_t *host = static_cast<uint32_t *>(std::malloc(words * 4));
        CK(hipMemcpyAsync(host, dev, words * 4, hipMemcpyDeviceToHost, stream));
        CK(hipStreamSynchronize(stream));
        for (size_t k = 0; k < words; ++k)
            if (host[k] != pattern) {
                if (g_early.fetch_add(1) < 8)
                    std::fprintf(stderr, "EARLY RETURN: thread %d iteration %llu word %zu of %zu holds %08x, not %08x\n", id, i, k, words, host[k], pattern);
                break;
            }
        std::free(host);
        uint32_t *again = static_cast<uint32_t *>(std::malloc(words * 4));      // (the allocator hands the block back)
        std::memset(again, 0, words * 4);
        kept.push_back({again, words, i, pattern});
        if (kept.size() > lag) {
            const Kept old = kept.front();
            kept.erase(kept.begin());
            for (size_t k = 0; k < old.words; ++k)
                if (old.p[k] != 0u) {
                    if (g_late.fetch_add(1) < 8)
                        std::fprintf(stderr, "LATE WRITE: thread %d, buffer zeroed in iteration %llu (same block as that iteration's download: %s), word %zu holds %08x "
                                             "(iteration's pattern %08x) %llu iterations later\n", id, old.born, old.p == host ? "yes" : "not known", k, old.p[k], old.pattern, i - old.born);
                    break;
                }
            std::free(old.p);
        }
        // ---- pageable host -> device ----
        uint32_t *src = static_cast<uint32_t *>(std::malloc(words * 4));
        for (size_t k = 0; k < words; ++k) src[k] = pattern ^ 0x55555555u;
        CK(hipMemcpyAsync(dev2, src, words * 4, hipMemcpyHostToDevice, stream));
        CK(hipStreamSynchronize(stream));
        std::memset(src, 0xA5, words * 4);                                        // what MALLOC_PERTURB_ does to a freed block
        std::free(src);
        if ((i & 7) == 0) {                                                         // (the read-back through page-locked memory costs a trip)
            CK(hipMemcpyAsync(pinned, dev2, words * 4, hipMemcpyDeviceToHost, stream));
            CK(hipStreamSynchronize(stream));
            for (size_t k = 0; k < words; ++k)
                if (pinned[k] != (pattern ^ 0x55555555u)) {
                    if (g_h2d.fetch_add(1) < 8)
                        std::fprintf(stderr, "LATE READ: thread %d iteration %llu: device word %zu holds %08x, the source held %08x when the copy was synchronised\n",
                                     id, i, k, pinned[k], pattern ^ 0x55555555u);
                    break;
                }
        }
        g_iterations.fetch_add(1, std::memory_order_relaxed);
    }
    for (auto &k : kept) std::free(k.p);
    (void)hipFree(dev); (void)hipFree(dev2); (void)hipHostFree(pinned); (void)hipStreamDestroy(stream);
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? std::atof(argv[1]) : 10.0;
    const int threads = argc > 2 ? std::atoi(argv[2]) : 4;
    const size_t max_bytes = argc > 3 ? (size_t)std::atoll(argv[3]) : (size_t)256 << 10;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t, max_bytes);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    g_stop.store(true);
    for (auto &t : pool) t.join();
    std::printf("{\"seconds\": %.1f, \"threads\": %d, \"max_bytes\": %zu, \"iterations\": %llu, \"early_returns\": %llu, \"late_writes\": %llu, \"late_reads\": %llu}\n",
                seconds, threads, max_bytes, g_iterations.load(), g_early.load(), g_late.load(), g_h2d.load());
    return (g_early.load() || g_late.load() || g_h2d.load()) ? 1 : 0;
}
