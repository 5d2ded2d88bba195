/*
 * oracle/snn_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference's lattice time-stepper hot
 * path, NikhilMukraj/spiking-neural-networks `backend` crate:
 *   RunLattice::run_lattice      backend/src/neuron/mod.rs:1199-1220
 *   RunNetwork::run_lattices     backend/src/neuron/mod.rs:2654-2675
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (spiking-neural-networks_amd/) never does.
 *
 * PARITY STATUS -- bit level: UNPINNED BY THE REFERENCE.  The reference is a
 * Rust crate (no rustc/cargo in this image), holds no golden vectors, seeds
 * every test from thread_rng, and sums synaptic inputs in HashSet iteration
 * order (mod.rs:710-720), i.e. its own CPU results are not bit-reproducible
 * run to run.  What IS pinned (tests/test_oracle_reference_pins.py): the
 * reference's known-answer tests for this path -- RateSpikeTrain spike
 * positions/counts (backend/tests/rate_spike_train.rs:28-72), AdjacencyMatrix
 * None-vs-Some semantics (graph/mod.rs:113-137), zero-size no-ops
 * (tests/size_zero_cases.rs), Poisson->neuron spike-count behaviour
 * (tests/spike_train_neuron_interaction.rs:91-203), interleaved index
 * placement (tests/interleaving_graph_conversion.rs) -- plus hand-derived
 * single-step values of every formula cited below.  The committed golden
 * vectors (tests/golden/ *.npz) are NOT written by this file: a second,
 * independently written numpy restatement (tests/numpy_net.py, host libm for
 * exp / powf) writes them, and this oracle is held to them
 * (tests/test_golden_oracle.py) as the device is (tests/test_gpu_golden.py);
 * tests/test_numpy_twin_randomized.py compares the two restatements on random
 * networks of every built-in model.
 *
 * Canonical choices where the reference leaves the result unspecified
 * (DESIGN.md "Canonical semantics"):
 *  - synaptic sums: presynaptic index space cut into chunks of SNN_O_CHUNK
 *    consecutive indices; inside a chunk strictly ascending sequential f32
 *    adds starting from 0.0f; chunk partials then added in ascending chunk
 *    order starting from 0.0f.
 *  - plasticity: the deferred (LatticeNetwork) form for single lattices too
 *    (mod.rs:2573-2576); the inline single-Lattice form (mod.rs:968-970)
 *    depends on a randomised HashSet order.
 *  - Poisson spike trains: the reference's GPU generator (xorshift32,
 *    spike_train/mod.rs:380-388, 411-435) with explicit seeds; the CPU form
 *    draws from an unseedable thread_rng (spike_train/mod.rs:354).
 *  - exp / powf: see snn_oracle_math.h.
 *
 * Index space: neurons of all lattices first (ascending lattice id, row-major
 * inside a lattice), spike-train cells after them -- the interleaved order of
 * InterleavingGraphGPU, backend/src/graph/mod.rs:668-727.  Spike-train cells
 * are never postsynaptic (mod.rs:1852-1854), so the weight matrix is stored
 * n_tot rows (pre) x n_neurons columns (post), row-major.
 */
#ifndef SNN_ORACLE_H
#define SNN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNN_O_K      3     /* AMPA=0, NMDA=1, GABA=2  (iterate_and_spike/mod.rs:1323-1333) */
#define SNN_O_CHUNK  256   /* canonical reduction chunk (presynaptic indices) */

enum { SNN_O_IZHIKEVICH = 0, SNN_O_LIF = 1, SNN_O_HH = 2, SNN_O_QIF = 3, SNN_O_SIMPLE_LIF = 4,
       SNN_O_ADAPTIVE_LIF = 5, SNN_O_ADAPTIVE_EXP_LIF = 6, SNN_O_LEAKY_IZHIKEVICH = 7,
       SNN_O_BCM_IZHIKEVICH = 8, SNN_O_CUSTOM = 100 };
enum { SNN_O_NT_APPROX = 0, SNN_O_NT_DESTEXHE = 1, SNN_O_NT_DISCRETE_SPIKE = 2, SNN_O_NT_EXPONENTIAL_DECAY = 3,
       SNN_O_NT_CUSTOM = 100 };
enum { SNN_O_RC_APPROX = 0, SNN_O_RC_DESTEXHE = 1, SNN_O_RC_EXPONENTIAL_DECAY = 2, SNN_O_RC_CUSTOM = 100 };
enum { SNN_O_ST_NONE = 0, SNN_O_ST_POISSON = 1, SNN_O_ST_RATE = 2, SNN_O_ST_PRESET = 3, SNN_O_ST_BCM_POISSON = 4,
       SNN_O_ST_CUSTOM = 100 };

typedef struct snn_o_net {
    /* ---- sizes / switches ---- */
    uint32_t n_neurons;      /* postsynaptic-capable cells */
    uint32_t n_cells;        /* spike-train cells (presynaptic only) */
    int32_t  model;          /* SNN_O_IZHIKEVICH | LIF | HH */
    int32_t  nt_kind;        /* neurotransmitter kinetics of neurons AND cells */
    int32_t  rc_kind;        /* receptor kinetics */
    int32_t  st_kind;        /* spike-train model */
    int32_t  electrical;     /* electrical_synapse  mod.rs:574 */
    int32_t  chemical;       /* chemical_synapse    mod.rs:576 */
    int64_t  clock;          /* internal_clock      mod.rs:586 */

    /* ---- neuron SoA, length n_neurons (names = reference struct fields) ---- */
    float    *current_voltage, *gap_conductance, *dt, *c_m, *v_th;
    uint32_t *is_spiking;
    int32_t  *last_firing_time;            /* -1 == None */
    /* Izhikevich (integrate_and_fire/mod.rs:1159-1194); tau_m shared with LIF */
    float    *w_value, *a, *b, *c, *d, *tau_m;
    /* LIF (integrate_and_fire/mod.rs:108-147) */
    float    *v_reset, *refractory_count, *tref, *leak_constant,
             *integration_constant, *e_l, *g_l;
    /* Hodgkin-Huxley (hodgkin_huxley/mod.rs:49-79, ion_channels/mod.rs) */
    float    *m_state, *h_state, *n_state;      /* na_channel.m/h, k_channel.n */
    float    *m_alpha, *m_beta, *h_alpha, *h_beta, *n_alpha, *n_beta;
    float    *g_na, *e_na, *g_k, *e_k, *g_k_leak, *e_k_leak;
    float    *na_current, *k_current, *k_leak_current;
    uint32_t *was_increasing;

    /* ---- neurotransmitters of neurons, [n_neurons * K], index n*K+k ---- */
    float    *nt_t, *nt_t_max, *nt_clearance, *nt_v_p, *nt_k_p;
    uint32_t *nt_flags;
    /* ---- receptors of neurons, [n_neurons * K] (mg: only k=NMDA is read) ---- */
    float    *rc_g, *rc_e, *rc_mg, *rc_r, *rc_alpha, *rc_beta, *rc_current;
    uint32_t *rc_flags;

    /* ---- spike-train cells, length n_cells ---- */
    float    *st_current_voltage, *st_v_th, *st_v_resting, *st_dt, *st_k;
    float    *st_chance_of_firing;         /* Poisson */
    float    *st_rate, *st_step;           /* Rate    */
    uint32_t *st_seed;                     /* Poisson xorshift32 state */
    uint32_t *st_is_spiking;
    int32_t  *st_last_firing_time;
    float    *st_nt_t, *st_nt_t_max, *st_nt_clearance, *st_nt_v_p, *st_nt_k_p; /* [n_cells*K] */
    uint32_t *st_nt_flags;
    uint32_t *st_lattice;                  /* [n_cells] -> spike-train lattice slot */
    uint32_t n_st_lattices;
    int64_t  *st_clock;                    /* [n_st_lattices] own internal clocks mod.rs:1391 */

    /* ---- graph: dense, row-major [n_tot][n_neurons]; conn==0 <=> None ---- */
    float    *weights;
    uint8_t  *connections;

    /* ---- lattices (plasticity is per lattice, mod.rs:578-580) ---- */
    uint32_t *lattice;                     /* [n_neurons] -> lattice slot */
    uint32_t n_lattices;
    float    *stdp_a_plus, *stdp_a_minus, *stdp_tau_plus, *stdp_tau_minus, *stdp_dt; /* [n_lattices] */
    uint32_t *do_plasticity;               /* [n_lattices] */

    /* ---- optional per-step outputs (NULL = off) ---- */
    float    *voltage_history;             /* [iterations][n_neurons]  GridVoltageHistory */
    uint8_t  *spike_history;               /* [iterations][n_neurons]  SpikeHistory */
    float    *st_voltage_history;          /* [iterations][n_cells] */

    /* ---- scratch supplied by the caller ---- */
    float    *input_current;               /* [n_neurons] */
    float    *input_t;                     /* [n_neurons*K] */
    float    *input_count;                 /* [n_neurons*K] (#pres that carry type k) */
    int32_t  n_threads;                    /* >1: OpenMP over postsynaptic neurons (≙ rayon par_, mod.rs:775-790) */
    /* column window of `weights`/`connections` (bounded CPU-baseline samples of a large matrix):
     * the arrays hold columns [w_col0, w_col0 + w_ld) only; w_ld == 0 means the full n_neurons. */
    uint32_t w_col0, w_ld;
    /* QuadraticIntegrateAndFireNeuron (integrate_and_fire/mod.rs:259-322; shares v_reset, refractory_count, tref,
     * integration_constant, tau_m with LIF) and SimpleLeakyIntegrateAndFire (:1523-1575; shares v_reset) -- the two
     * models the reference's own GPU path implements */
    float    *qif_alpha, *qif_v_c, *slif_g, *slif_e;
    /* Reduced per-lattice histories (AverageVoltageHistory neuron/mod.rs:305-322, EEGHistory :233-284) and
     * SpikeHistory::aggregate (:331-360).  Lattice l owns neurons [lattice_first[l], +lattice_count[l]).  Sums use
     * the canonical chunked order (chunks of SNN_O_CHUNK consecutive neurons of the lattice); the reference sums
     * strictly sequentially -- deterministic, but not parallelisable, see DESIGN.md. */
    uint32_t *lattice_first, *lattice_count;   /* [n_lattices] */
    float    *avg_history, *eeg_history;       /* [iterations][n_lattices] or NULL */
    float    eeg_reference_voltage, eeg_distance, eeg_conductivity;
    uint32_t *spike_counts;                    /* [n_neurons] accumulated over the run, or NULL */
    /* AdaptiveLeakyIntegrateAndFireNeuron (integrate_and_fire/mod.rs:918-1049) and
     * AdaptiveExpLeakyIntegrateAndFireNeuron (:1051-1155): alpha, beta (+ slope_factor) next to the LIF arrays and
     * w_value; LeakyIzhikevichNeuron (:1270-1356) uses the Izhikevich arrays + e_l */
    float    *adp_alpha, *adp_beta, *slope_factor;
    /* PresetSpikeTrain (spike_train/mod.rs:753-833): firing times of cell s are
     * st_firing_times[st_firing_ptr[s] .. st_firing_ptr[s+1]); st_step holds its internal_clock.
     * Further kinetics reuse arrays: ExponentialDecayNeurotransmitter's decay_constant lives in nt_clearance /
     * st_nt_clearance, ExponentialDecayReceptor's r_max in rc_alpha and its decay_constant in rc_beta. */
    uint32_t *st_firing_ptr;                   /* [n_cells + 1] */
    float    *st_firing_times;
    uint32_t *st_counter;                      /* [n_cells] */
    /* Reward modulation of a lattice's internal edges: RewardModulatedLattice (neuron/mod.rs:2719-3417) with
     * RewardModulatedSTDP + TraceRSTDP (plasticity/mod.rs:126-242), in the deferred form (see snn_o_reward_modulation).
     * rm_* are per lattice; `traces` holds TraceRSTDP::c per edge in the layout of `weights` (TraceRSTDP::weight IS
     * the entry of `weights`; counter and dw are 0 at every step boundary in the deferred form).  `rewards[it]` is
     * applied before step `it` of snn_o_run (Agent::update_and_apply_reward, :3402-3407); NULL = run without reward. */
    float    *traces;
    uint32_t *rm_do_modulation;
    float    *rm_dopamine, *rm_tau_d, *rm_tau_c, *rm_a_plus, *rm_a_minus, *rm_tau_plus, *rm_tau_minus, *rm_dt;
    const float *rewards;
    /* NeuralRefractoriness of each spike-train cell: 0 DeltaDiracRefractoriness (spike_train/mod.rs:79-88),
     * 1 ExponentialDecayRefractoriness (:164-178); NULL = all DeltaDirac */
    uint32_t *st_refractoriness;
    /* BCM (plasticity/mod.rs:72-116): plasticity_kind[l] 0 = STDP, 1 = BCM with bcm_* parameters; the activity state of
     * BCMIzhikevichNeuron (integrate_and_fire/mod.rs:1358-1518) per neuron and of BCMPoissonNeuron
     * (spike_train/mod.rs:835-970) per cell */
    uint32_t *plasticity_kind;                         /* [n_lattices] */
    float    *bcm_decay, *bcm_average_scalar, *bcm_dt; /* [n_lattices] */
    float    *bcm_average_activity, *bcm_current_activity, *bcm_clock, *bcm_window;   /* [n_neurons] */
    uint32_t *bcm_period, *bcm_num_spikes;                                             /* [n_neurons] */
    float    *st_bcm_average_activity, *st_bcm_current_activity, *st_bcm_clock, *st_bcm_window;   /* [n_cells] */
    uint32_t *st_bcm_period, *st_bcm_num_spikes;                                                   /* [n_cells] */
    /* SNN_O_CUSTOM: a neuron model given as a small stack program (tests/modelgen_ref.py compiles it from the same
     * description the product turns into HIP; semantics of the reference's neuron_builder! output,
     * build_test/nb_macro/src/lib.rs:2259-2345).  custom_code = int32 words, three sections starting at
     * custom_section[0..2] (on_iteration, spike_detection, on_spike), each ended by OP_END; custom_vars[k] is the
     * k-th model variable, [n_neurons]. */
    const int32_t *custom_code;
    const float   *custom_consts;
    uint32_t custom_section[3];
    uint32_t custom_nvars;
    float    *custom_vars;                     /* [custom_nvars][n_neurons] */
    /* SNN_O_ST_CUSTOM: a generated spike train (nb_macro lib.rs:4812-4905) as a stack program -- one section, the
     * on_iteration; slots 0 current_voltage, 1 is_spiking (1.0 / 0.0), 2 dt, 3 v_resting, 4 v_th, 5.. variables. */
    const int32_t *st_custom_code;
    const float   *st_custom_consts;
    uint32_t st_custom_nvars;
    float    *st_custom_vars;                  /* [st_custom_nvars][n_cells] */
    /* st_refractoriness == 2: a generated NeuralRefractoriness::get_effect (lib.rs:5677-5762) as ONE expression;
     * slots 0 time_difference, 1 v_th, 2 dt, 3 v_resting, 4 decay (= st_k), 5.. variables. */
    const int32_t *refr_code;
    const float   *refr_consts;
    uint32_t refr_nvars;
    float    *refr_vars;                       /* [refr_nvars][n_cells] */
    /* nt_kind == SNN_O_NT_CUSTOM: generated NeurotransmitterKinetics::apply_t_change (lib.rs:6468-6540) as a stack
     * program; slots 0 t, 1 is_spiking, 2 dt, 3 voltage, 4 unused, 5.. variables (one value per cell and type, indexed
     * like nt_t).  rc_kind == SNN_O_RC_CUSTOM: generated ReceptorKinetics::apply_r_change (lib.rs:6757-6826); slots
     * 0 r, 1 t, 2 dt, 5.. variables (indexed like rc_r). */
    const int32_t *nt_code;
    const float   *nt_consts;
    uint32_t nt_nvars;
    float    *nt_custom_vars;                  /* [nt_nvars][n_neurons * 3] */
    float    *st_nt_custom_vars;               /* [nt_nvars][n_cells * 3] */
    const int32_t *rc_code;
    const float   *rc_consts;
    uint32_t rc_nvars;
    float    *rc_custom_vars;                  /* [rc_nvars][n_neurons * 3] */
    /* SNN_O_CUSTOM with an on_electrochemical_iteration (lib.rs:2280-2316): a fourth section of custom_code that
     * replaces the default chemical step (receptor update, on_iteration, v -= currents, transmitter update) */
    uint32_t custom_has_chem, custom_chem_section;
    /* the generated neuron's [receptors] set (lib.rs:7017-7600): rx_ntypes (0 = the ionotropic AMPA/NMDA/GABA set)
     * neurotransmitter types in slots 0.., one program section per type (slots 0 v, 1 r, 5.. the set's variables),
     * rx_current_index[k] = the variable holding type k's current or -1 */
    const int32_t *rx_code;
    const float   *rx_consts;
    uint32_t rx_ntypes, rx_nvars;
    uint32_t rx_section[3];
    int32_t  rx_current_index[3];
    float    *rx_vars;                         /* [rx_nvars][n_neurons] */
    /* several receptor states per type (`receptors: a, b`): rx_multi != 0, and rx_kin_section[k] = the program that
     * applies the receptor kinetics to every state of type k (slots 2 dt, 3 t, 5.. the set's variables, which then hold
     * the states' r and kinetics variables); rc_r / rc_kind are not used by such a set */
    uint32_t rx_multi;
    uint32_t rx_kin_section[3];
    /* RewardModulatedLatticeNetwork: connections BETWEEN lattices that carry a RewardModulatedConnection
     * (update_weights_from_neurons_across_lattices / _across_reward_lattices, neuron/mod.rs:4707-4977; see snn_o_reward_cross).
     * conn_kind[source * n_lattices + post lattice], source = the lattice slot of a presynaptic neuron or n_lattices + the
     * spike-train lattice slot of a cell: 0 = an edge of a plain LatticeNetwork (the default), 1 =
     * RewardModulatedConnection::RewardModulatedWeight, 2 = RewardModulatedConnection::Weight.  `pending` = TraceRSTDP::dw and
     * `edge_counter` = TraceRSTDP::counter per edge, both in the layout of `weights`.  NULL: no such connection. */
    uint8_t  *conn_kind;
    float    *pending;
    uint8_t  *edge_counter;
    /* Which MAP of the reference's network holds a lattice (RewardModulatedLatticeNetwork::lattices vs
     * ::reward_modulated_lattices, neuron/mod.rs:3419-3453) is one thing, RewardModulatedLattice::do_modulation (:2744) another:
     * a modulated lattice whose do_modulation is false is never visited (:5113) and updates no weight of its own (:3076), but
     * as a PARTNER of another lattice's visit it is still a modulated lattice (:4729, :4869, :4937), it has no STDP rule, and its
     * modulator still takes every reward (:5287-5291).  rm_is_modulated[l] != 0 or rm_do_modulation[l] != 0: lattice l is a
     * reward-modulated lattice.  NULL: the lattices with do_modulation set are. */
    uint32_t *rm_is_modulated;
} snn_o_net;

/* Step 1 of SURVEY §8(g): electrical + chemical inputs for every neuron from state S(t). */
void snn_o_inputs(snn_o_net *net);
/* Same, restricted to postsynaptic columns [q0, q1) (bounded CPU-baseline sample / column checks). */
void snn_o_inputs_range(snn_o_net *net, uint32_t q0, uint32_t q1);
/* Step 2: advance every neuron once with the inputs in net->input_*; stamps last_firing_time. */
void snn_o_update_neurons(snn_o_net *net);
void snn_o_update_neurons_range(snn_o_net *net, uint32_t q0, uint32_t q1);
/* Step 3: deferred STDP for every neuron that spiked in this step. */
void snn_o_plasticity(snn_o_net *net);
void snn_o_apply_reward(snn_o_net *n, float reward);
void snn_o_reward_modulation(snn_o_net *n);
void snn_o_reward_cross(snn_o_net *n);
/* 0 = every connection of kind 1 / 2 lies where the reference defines it; 1..4: see snn_o_reward_cross */
int snn_o_reward_cross_check(const snn_o_net *n);
void snn_o_reward_modulation_cols(snn_o_net *n, uint32_t c0, uint32_t c1);
void snn_o_plasticity_cols(snn_o_net *net, uint32_t c0, uint32_t c1);
/* Step 6: iterate every spike-train cell once. */
void snn_o_spike_trains(snn_o_net *net);
/* Step 1 for posts [q0, q1) over a sparse graph in CSR-by-post form (ascending presynaptic indices per row): the same
 * canonical chunked order as the dense routine, bit-identical to it on the same graph; and the whole loop over it
 * (no plasticity) -- BASELINE configs[4] at full size */
void snn_o_inputs_csr(snn_o_net *net, const uint64_t *row_ptr, const uint32_t *pre, const float *w,
                      uint32_t q0, uint32_t q1);
void snn_o_run_csr(snn_o_net *net, const uint64_t *row_ptr, const uint32_t *pre, const float *w, uint64_t iterations);
/* Whole loop (steps 1-6) `iterations` times, filling the optional histories. */
void snn_o_run(snn_o_net *net, uint64_t iterations);
/* Step 1 for [q0, q1) arranged for all-core memory bandwidth (bench.py's cpu_baseline); same results bit for bit */
void snn_o_inputs_tiled(snn_o_net *n, uint32_t q0, uint32_t q1, uint32_t block);
void snn_o_fill_graph_window_blocked(float *weights, uint8_t *connections, uint32_t n_tot, uint32_t n_neurons,
                                     uint32_t col0, uint32_t ncols, uint32_t block, uint64_t seed, float lo, float hi,
                                     int with_diagonal, int n_threads);

/* Single-formula entry points (known-answer tests, GPU device-function parity). */
float snn_o_expf_export(float x);
float snn_o_pow3f_export(float x);
float snn_o_pow4f_export(float x);
float snn_o_tanhf_export(float x);
float snn_o_sinhf_export(float x);
float snn_o_coshf_export(float x);
float snn_o_powif_export(float x, int n);
float snn_o_sinf_export(float x);
float snn_o_cosf_export(float x);
float snn_o_tanf_export(float x);
float snn_o_powf_export(float x, float y);
void snn_o_math_bits(int which, uint32_t first, uint32_t stride, uint64_t count, float y, float *out);
float snn_o_stdp_delta(int32_t t_pre, int32_t t_post, float a_plus, float a_minus,
                       float tau_plus, float tau_minus, float dt);
float snn_o_exponential_decay_effect(int64_t timestep, int32_t last_firing_time,
                                     float v_th, float v_resting, float k, float dt);
float snn_o_delta_dirac_effect(int64_t timestep, int32_t last_firing_time,
                               float v_th, float v_resting, float k, float dt);
uint32_t snn_o_xorshift32(uint32_t seed);

/* Synthetic data shared by tests and bench: counter-based, order independent. */
uint32_t snn_o_hash32(uint64_t seed, uint64_t index);
float    snn_o_uniform(uint64_t seed, uint64_t index, float lo, float hi);
/* weights[p][q] = U[lo,hi) from (seed, p*n_neurons+q); conn = (p != q) unless with_diagonal */
void snn_o_fill_graph(float *weights, uint8_t *connections, uint32_t n_tot, uint32_t n_neurons,
                      uint64_t seed, float lo, float hi, int with_diagonal);

void snn_o_fill_graph_window(float *weights, uint8_t *connections, uint32_t n_tot, uint32_t n_neurons,
                             uint32_t col0, uint32_t ncols, uint64_t seed, float lo, float hi,
                             int with_diagonal);

#ifdef __cplusplus
}
#endif
#endif /* SNN_ORACLE_H */
