// A native-thread host of the C ABI (include/snn_amd.h: "distinct handles may be used from distinct threads"; the reference's
// objects are Send but not Sync, backend/src/neuron/gpu_lattices/mod.rs:327-350).  Four std::threads, one handle each -- a dense
// lattice with histories, a sparse handle with Rate cells, a plastic network (STDP), a handle of a library that carries a generated
// neuron model -- each making 200 rounds of run / get / set calls.  The digest of everything a thread read must equal the digest of
// the same workload run alone on the main thread, in every one of three concurrent repetitions; each thread also provokes an error
// and must find ITS OWN message in snn_last_error (thread-local).  Raw C ABI through dlopen: two libraries with the same symbols.
//   threads_host_test <libsnn_amd.so> <libsnn_amd_<generated>.so>
#include <dlfcn.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/snn_amd.h"

struct Api {
    void *so = nullptr;
    decltype(&snn_network_create) create;
    decltype(&snn_network_destroy) destroy;
    decltype(&snn_network_add_lattice) add_lattice;
    decltype(&snn_network_add_spike_train_lattice) add_spike_train_lattice;
    decltype(&snn_network_finalize) finalize;
    decltype(&snn_network_use_csr) use_csr;
    decltype(&snn_set_attr_f32) set_f32;
    decltype(&snn_get_attr_f32) get_f32;
    decltype(&snn_get_attr_i32) get_i32;
    decltype(&snn_set_graph_rows) set_graph_rows;
    decltype(&snn_get_graph_rows) get_graph_rows;
    decltype(&snn_set_graph_csr) set_graph_csr;
    decltype(&snn_get_graph_csr) get_graph_csr;
    decltype(&snn_set_synapses) set_synapses;
    decltype(&snn_set_plasticity) set_plasticity;
    decltype(&snn_set_history) set_history;
    decltype(&snn_reset_history) reset_history;
    decltype(&snn_get_voltage_history) get_voltage_history;
    decltype(&snn_get_spike_history) get_spike_history;
    decltype(&snn_history_steps) history_steps;
    decltype(&snn_run) run;
    decltype(&snn_last_error) last_error;
    bool load(const char *path)
    {
        so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!so) { std::fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return false; }
#define SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(so, #name)); if (!field) { std::fprintf(stderr, "missing %s\n", #name); return false; }
        SYM(create, snn_network_create) SYM(destroy, snn_network_destroy) SYM(add_lattice, snn_network_add_lattice)
        SYM(add_spike_train_lattice, snn_network_add_spike_train_lattice) SYM(finalize, snn_network_finalize) SYM(use_csr, snn_network_use_csr)
        SYM(set_f32, snn_set_attr_f32) SYM(get_f32, snn_get_attr_f32) SYM(get_i32, snn_get_attr_i32) SYM(set_graph_rows, snn_set_graph_rows)
        SYM(get_graph_rows, snn_get_graph_rows) SYM(set_graph_csr, snn_set_graph_csr) SYM(get_graph_csr, snn_get_graph_csr)
        SYM(set_synapses, snn_set_synapses) SYM(set_plasticity, snn_set_plasticity) SYM(set_history, snn_set_history)
        SYM(reset_history, snn_reset_history) SYM(get_voltage_history, snn_get_voltage_history) SYM(get_spike_history, snn_get_spike_history)
        SYM(history_steps, snn_history_steps) SYM(run, snn_run) SYM(last_error, snn_last_error)
#undef SYM
        return true;
    }
};

struct Digest {
    uint64_t h = 1469598103934665603ull;
    void add(const void *p, size_t n) { const unsigned char *b = static_cast<const unsigned char *>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } }
};

static uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
static float unit(uint32_t seed, uint32_t i) { return (float)(mix(seed * 0x9E3779B9u + i) >> 8) * (1.0f / 16777216.0f); }

#define CHECK(expr) do { int rc_ = (expr); if (rc_ != SNN_OK) { std::snprintf(err, 256, "%s -> %d: %s", #expr, rc_, api.last_error()); return false; } } while (0)

enum Kind { DENSE = 0, SPARSE = 1, PLASTIC = 2, GENERATED = 3 };

// one workload, start to finish, on the calling thread; `digest` covers everything read back
static bool workload(const Api &api, Kind kind, int rounds, uint64_t *digest, char *err)
{
    const uint32_t rows = kind == DENSE ? 12 : kind == SPARSE ? 20 : kind == PLASTIC ? 9 : 8, cols = kind == SPARSE ? 20 : rows + 1;
    const uint32_t n = rows * cols, cells = kind == SPARSE ? 16 : 0, n_tot = n + cells, seed = 17 + (uint32_t)kind;
    snn_network_t *net = nullptr;
    uint64_t nnz = 0;
    CHECK(api.create(0, kind == GENERATED ? SNN_MODEL_CUSTOM : SNN_MODEL_IZHIKEVICH, 0, 0, cells ? SNN_ST_RATE : SNN_ST_NONE, &net));
    struct Closer { const Api &a; snn_network_t *n; ~Closer() { a.destroy(n); } } closer{api, net};
    CHECK(api.add_lattice(net, 0, rows, cols));
    if (cells) CHECK(api.add_spike_train_lattice(net, 1, 4, 4));
    if (kind == SPARSE) CHECK(api.use_csr(net, 1));
    CHECK(api.finalize(net));
    std::vector<float> v(n), g(n, 10.0f);
    for (uint32_t i = 0; i < n; ++i) v[i] = -65.0f + 95.0f * unit(seed, i);
    CHECK(api.set_f32(net, 0, "current_voltage", v.data(), n));
    CHECK(api.set_f32(net, 0, "gap_conductance", g.data(), n));
    if (cells) { std::vector<float> rate(cells); for (uint32_t i = 0; i < cells; ++i) rate[i] = 0.3f + 0.1f * (float)(i % 5); CHECK(api.set_f32(net, 1, "rate", rate.data(), cells)); }
    if (kind == SPARSE) {
        std::vector<uint64_t> ptr(n + 1, 0);
        std::vector<uint32_t> pre;
        std::vector<float> w;
        for (uint32_t q = 0; q < n; ++q) {                                   // neighbours at distance <= 2 in the flat index + one cell
            for (uint32_t p = q >= 2 ? q - 2 : 0; p <= q + 2 && p < n; ++p)
                if (p != q) { pre.push_back(p); w.push_back(0.5f + unit(seed + 1, q * 8 + (p + 2 - q))); }
            if (q % 7 == 0) { pre.push_back(n + (q / 7) % cells); w.push_back(1.5f); }
            ptr[q + 1] = pre.size();
        }
        nnz = pre.size();
        CHECK(api.set_graph_csr(net, ptr.data(), pre.data(), w.data(), nnz));
    } else {
        std::vector<float> w((size_t)n_tot * n);
        std::vector<uint32_t> c((size_t)n_tot * n);
        for (uint32_t p = 0; p < n_tot; ++p)
            for (uint32_t q = 0; q < n; ++q) { c[(size_t)p * n + q] = (p != q && mix(seed * 31 + p * 1031 + q) % 4 != 0) ? 1u : 0u; w[(size_t)p * n + q] = 0.2f + unit(seed + 2, p * n + q); }
        CHECK(api.set_graph_rows(net, 0, n_tot, w.data(), c.data()));
    }
    CHECK(api.set_synapses(net, 1, 0));
    if (kind == PLASTIC) CHECK(api.set_plasticity(net, 0, 2.0f, 2.0f, 4.5f, 4.5f, 0.1f, 1));
    if (kind == DENSE) CHECK(api.set_history(net, 1, 1));
    Digest d;
    std::vector<float> hist, wrow;
    std::vector<uint8_t> spikes;
    std::vector<int32_t> lft(n);
    std::vector<uint32_t> crow;
    for (int r = 0; r < rounds; ++r) {
        CHECK(api.run(net, 1 + (uint64_t)(mix(seed + r) % 7)));
        CHECK(api.get_f32(net, 0, "current_voltage", v.data(), n));
        d.add(v.data(), n * 4);
        if (r % 5 == 0) { CHECK(api.get_i32(net, 0, "last_firing_time", lft.data(), n)); d.add(lft.data(), n * 4); }
        if (kind == DENSE && r % 10 == 9) {
            uint64_t steps = 0;
            CHECK(api.history_steps(net, &steps));
            hist.resize(steps * n); spikes.resize(steps * n);
            CHECK(api.get_voltage_history(net, 0, hist.data(), hist.size()));
            CHECK(api.get_spike_history(net, 0, spikes.data(), spikes.size()));
            d.add(hist.data(), hist.size() * 4); d.add(spikes.data(), spikes.size());
            CHECK(api.reset_history(net));
        }
        if (kind == PLASTIC && r % 8 == 7) {
            wrow.resize((size_t)n * n); crow.resize((size_t)n * n);
            CHECK(api.get_graph_rows(net, 0, n, wrow.data(), crow.data()));
            for (size_t i = 0; i < wrow.size(); ++i) if (crow[i]) d.add(&wrow[i], 4);
        }
        if (kind == SPARSE && r % 16 == 15) {
            wrow.resize(nnz);
            CHECK(api.get_graph_csr(net, wrow.data(), nnz));
            d.add(wrow.data(), nnz * 4);
        }
        // a host write between run calls: a few neurons get a new voltage
        for (uint32_t k = 0; k < 3; ++k) v[mix(seed * 7 + r * 3 + k) % n] = -60.0f + 80.0f * unit(seed + 3, r * 3 + k);
        CHECK(api.set_f32(net, 0, "current_voltage", v.data(), n));
    }
    // the provoked error: this thread's own message
    std::string name = "no_such_field_of_workload_" + std::to_string((int)kind);
    if (api.set_f32(net, 0, name.c_str(), v.data(), n) != SNN_ERR_BAD_ATTR) { std::snprintf(err, 256, "bad attribute accepted"); return false; }
    if (!std::strstr(api.last_error(), name.c_str())) { std::snprintf(err, 256, "snn_last_error is not this thread's: \"%s\"", api.last_error()); return false; }
    *digest = d.h;
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s libsnn_amd.so libsnn_amd_<generated>.so [rounds]\n", argv[0]); return 64; }
    const int rounds = argc > 3 ? std::atoi(argv[3]) : 200;
    Api base, generated;
    if (!base.load(argv[1]) || !generated.load(argv[2])) return 65;
    const Api *api_of[4] = {&base, &base, &base, &generated};
    uint64_t alone[4];
    char err[256];
    for (int k = 0; k < 4; ++k)
        if (!workload(*api_of[k], (Kind)k, rounds, &alone[k], err)) { std::fprintf(stderr, "workload %d alone: %s\n", k, err); return 1; }
    int bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        uint64_t got[4] = {0, 0, 0, 0};
        char errs[4][256] = {{0}, {0}, {0}, {0}};
        std::atomic<int> failed{0};
        std::vector<std::thread> threads;
        for (int k = 0; k < 4; ++k)
            threads.emplace_back([&, k] { if (!workload(*api_of[k], (Kind)k, rounds, &got[k], errs[k])) failed.fetch_add(1); });
        for (auto &t : threads) t.join();
        for (int k = 0; k < 4; ++k) {
            if (errs[k][0]) { std::fprintf(stderr, "repetition %d, thread %d: %s\n", rep, k, errs[k]); ++bad; }
            else if (got[k] != alone[k]) { std::fprintf(stderr, "repetition %d, thread %d: digest %016llx, alone %016llx\n", rep, k, (unsigned long long)got[k], (unsigned long long)alone[k]); ++bad; }
        }
    }
    std::printf("{\"rounds\": %d, \"threads\": 4, \"repetitions\": 3, \"mismatches\": %d, \"digests\": [\"%016llx\", \"%016llx\", \"%016llx\", \"%016llx\"]}\n", rounds, bad,
                (unsigned long long)alone[0], (unsigned long long)alone[1], (unsigned long long)alone[2], (unsigned long long)alone[3]);
    return bad ? 2 : 0;
}
