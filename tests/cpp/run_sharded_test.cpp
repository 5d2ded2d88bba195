// A C/C++ host driving the multi-GPU path through the C ABI alone (include/snn_amd.h): the library creates the RCCL
// communicator (snn_comm_unique_id / snn_comm_init_rank), the handle is a shard handle (snn_network_finalize_shard)
// and the whole step loop runs inside snn_run_sharded -- what a Rust `extern "C"` caller of the reference's
// LatticeNetworkGPU::run_lattices seam (neuron/gpu_lattices/mod.rs:3183-3212) would do, one process per GPU.  Here
// world size = 1 (one GPU); argv: <out dir> <dense|csr> <steps>.  Writes the final state as raw arrays that
// tests/test_gpu_run_sharded.py compares with the oracle and with an unsharded snn_run.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/snn_amd.h"

#define CHECK(x)                                                                        \
    do {                                                                                \
        int rc_ = (x);                                                                  \
        if (rc_ != SNN_OK) {                                                            \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, snn_last_error());          \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static float v_init(uint32_t i) { return -65.0f + 6.0f * (float)((i * 7) % 19); }
static float weight(uint32_t p, uint32_t q) { return 0.5f + 0.0625f * (float)((p * 3 + q * 5) % 16); }
static bool edge(uint32_t p, uint32_t q) { return p != q && (p * 31 + q * 17) % 5 != 0; }

template <typename T>
static void dump(const std::string &path, const std::vector<T> &v)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    std::fwrite(v.data(), sizeof(T), v.size(), f);
    std::fclose(f);
}

int main(int argc, char **argv)
{
    const std::string out = argc > 1 ? argv[1] : ".";
    const bool csr = argc > 2 && std::string(argv[2]) == "csr";
    const uint64_t steps = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 200;
    const uint32_t rows = 9, cols = 11, n = rows * cols;

    unsigned char id[128];
    void *comm = nullptr;
    CHECK(snn_comm_unique_id(id));
    CHECK(snn_comm_init_rank(id, /*world*/ 1, /*rank*/ 0, /*device*/ 0, &comm));

    snn_network_t *net = nullptr;
    CHECK(snn_network_create(0, SNN_MODEL_IZHIKEVICH, 0, 0, 0, &net));
    CHECK(snn_network_add_lattice(net, 0, rows, cols));
    if (csr) CHECK(snn_network_use_csr(net, 1));
    CHECK(snn_network_finalize_shard(net, 0, 1));

    std::vector<float> v(n), g(n, 10.0f);
    for (uint32_t i = 0; i < n; ++i) v[i] = v_init(i);
    CHECK(snn_set_attr_f32(net, 0, "current_voltage", v.data(), n));
    CHECK(snn_set_attr_f32(net, 0, "gap_conductance", g.data(), n));
    if (csr) {
        std::vector<uint64_t> ptr(n + 1, 0);
        std::vector<uint32_t> pre;
        std::vector<float> w;
        for (uint32_t q = 0; q < n; ++q) {
            for (uint32_t p = 0; p < n; ++p)
                if (edge(p, q)) { pre.push_back(p); w.push_back(weight(p, q)); }
            ptr[q + 1] = pre.size();
        }
        CHECK(snn_set_graph_csr(net, ptr.data(), pre.data(), w.data(), pre.size()));
    } else {
        std::vector<float> w((size_t)n * n, 0.0f);
        std::vector<uint32_t> c((size_t)n * n, 0);
        for (uint32_t p = 0; p < n; ++p)
            for (uint32_t q = 0; q < n; ++q)
                if (edge(p, q)) { c[(size_t)p * n + q] = 1; w[(size_t)p * n + q] = weight(p, q); }
        CHECK(snn_set_graph_dense(net, w.data(), c.data(), n));
    }
    CHECK(snn_set_plasticity(net, 0, 2.0f, 2.0f, 4.5f, 4.5f, 0.1f, 1));

    snn_exchange_plan plan;
    CHECK(snn_exchange_plan_get(net, &plan));
    if (plan.n_shards != 1 || plan.planes != 1 || plan.plane_id[0] != 0 || plan.send_words != plan.shard_stride + plan.shard_stride / 32) {
        std::fprintf(stderr, "unexpected exchange plan: planes %u, send_words %llu, stride %u\n", plan.planes,
                     (unsigned long long)plan.send_words, plan.shard_stride);
        return 1;
    }
    CHECK(snn_run_sharded(net, comm, steps / 2));
    CHECK(snn_run_sharded(net, comm, steps - steps / 2));         // resumes
    // the three-call form with the library's collective in between
    for (int i = 0; i < 10; ++i) {
        CHECK(snn_step_begin(net));
        CHECK(snn_exchange(net, comm));
        CHECK(snn_step_end(net));
    }
    uint64_t clock = 0;
    CHECK(snn_get_clock(net, &clock));
    if (clock != steps + 10) { std::fprintf(stderr, "clock %llu\n", (unsigned long long)clock); return 1; }

    std::vector<float> wv(n);
    std::vector<int32_t> lft(n);
    CHECK(snn_get_attr_f32(net, 0, "current_voltage", v.data(), n));
    CHECK(snn_get_attr_f32(net, 0, "w_value", wv.data(), n));
    CHECK(snn_get_attr_i32(net, 0, "last_firing_time", lft.data(), n));
    dump(out + "/v.f32", v);
    dump(out + "/w_value.f32", wv);
    dump(out + "/lft.i32", lft);
    CHECK(snn_network_destroy(net));
    CHECK(snn_comm_destroy(comm));
    std::puts("ok");
    return 0;
}
