// Hook for ONE generated description: spiking-neural-networks_amd/modelgen.py turns blocks of the reference's
// neuron_builder! DSL (build_test/nb_macro) into a header with, per block, a variable table and the block's code as
// device functions:
//   [neuron] (+ [ion_channel]s)   -> namespace custom       = neuron model SNN_MODEL_CUSTOM
//   [spike_train]                 -> namespace custom_st    = spike-train model SNN_ST_CUSTOM
//   [neural_refractoriness]       -> namespace custom_refr  = neural_refractoriness$kind 2
//   [neurotransmitter_kinetics]   -> namespace custom_nt    = neurotransmitter kinetics SNN_NT_CUSTOM
//   [receptor_kinetics]           -> namespace custom_rc    = receptor kinetics SNN_RC_CUSTOM
//   [receptors] (of the [neuron]) -> namespace custom_receptors: the neuron's receptor set in place of AMPA/NMDA/GABA
// A library compiled with -DSNN_CUSTOM_MODEL_HEADER="\"generated/<name>.hpp\"" carries them next to the built-in
// models.  Without the define (or for a block the description does not have) the hook is an empty stub and the
// selector is refused.
#pragma once
#include "snn_math.hpp"

#ifdef SNN_CUSTOM_MODEL_HEADER
#include SNN_CUSTOM_MODEL_HEADER
#define SNN_HAVE_CUSTOM_MODEL 1            // a generated library: two-kernel step only (shorter compile)
#else
#define SNN_HAVE_CUSTOM_MODEL 0
#endif
#ifndef SNN_HAVE_CUSTOM_NEURON
#define SNN_HAVE_CUSTOM_NEURON 0
namespace snn {
namespace custom {
constexpr int NVARS = 0;
constexpr int NSTORE = 1;
static const char *const TYPE_NAME = "";
static const char *const NAMES[NSTORE] = {""};
static const float DEFAULTS[NSTORE] = {0.0f};
constexpr float DEFAULT_VOLTAGE = 0.0f, DEFAULT_DT = 0.1f, DEFAULT_C_M = 1.0f, DEFAULT_GAP = 10.0f;
__device__ __forceinline__ void on_iteration(float &, float (&)[NSTORE], float, float, float, float) {}
__device__ __forceinline__ bool spike_detection(float, float (&)[NSTORE], float, float, float, float) { return false; }
__device__ __forceinline__ void on_spike(float &, float (&)[NSTORE], float, float, float, float) {}
constexpr bool HAS_ELECTROCHEMICAL = false;
template <class Chem>
__device__ __forceinline__ void on_electrochemical_iteration(float &, float (&)[NSTORE], float, float, float, float, Chem &) {}
} // namespace custom
} // namespace snn
#endif
#ifndef SNN_HAVE_CUSTOM_SPIKE_TRAIN
#define SNN_HAVE_CUSTOM_SPIKE_TRAIN 0
namespace snn {
namespace custom_st {
constexpr int NVARS = 0;
constexpr int NSTORE = 1;
static const char *const TYPE_NAME = "";
static const char *const NAMES[NSTORE] = {""};
static const float DEFAULTS[NSTORE] = {0.0f};
constexpr float DEFAULT_VOLTAGE = 0.0f, DEFAULT_DT = 0.1f, DEFAULT_V_RESTING = 0.0f, DEFAULT_V_TH = 30.0f;
__device__ __forceinline__ void on_iteration(float &, bool &, float (&)[NSTORE], float, float, float) {}
} // namespace custom_st
} // namespace snn
#endif
#ifndef SNN_HAVE_CUSTOM_REFRACTORINESS
#define SNN_HAVE_CUSTOM_REFRACTORINESS 0
namespace snn {
namespace custom_refr {
constexpr int NVARS = 0;
constexpr int NSTORE = 1;
static const char *const TYPE_NAME = "";
static const char *const NAMES[NSTORE] = {""};
static const float DEFAULTS[NSTORE] = {0.0f};
constexpr float DEFAULT_DECAY = 10000.0f;
__device__ __forceinline__ float effect(float, float, float v_resting, float, float, const float (&)[NSTORE]) { return v_resting; }
} // namespace custom_refr
} // namespace snn
#endif

#ifndef SNN_HAVE_CUSTOM_NT
#define SNN_HAVE_CUSTOM_NT 0
namespace snn {
namespace custom_nt {
constexpr int NVARS = 0;
constexpr int NSTORE = 1;
static const char *const TYPE_NAME = "";
static const char *const NAMES[NSTORE] = {""};
static const float DEFAULTS[NSTORE] = {0.0f};
__device__ __forceinline__ void apply(float &, float (&)[NSTORE], float, bool, float) {}
} // namespace custom_nt
} // namespace snn
#endif
#ifndef SNN_HAVE_CUSTOM_RC
#define SNN_HAVE_CUSTOM_RC 0
namespace snn {
namespace custom_rc {
constexpr int NVARS = 0;
constexpr int NSTORE = 1;
static const char *const TYPE_NAME = "";
static const char *const NAMES[NSTORE] = {""};
static const float DEFAULTS[NSTORE] = {0.0f};
__device__ __forceinline__ void apply(float &, float (&)[NSTORE], float, float) {}
} // namespace custom_rc
} // namespace snn
#endif

#ifndef SNN_HAVE_CUSTOM_RECEPTORS
#define SNN_HAVE_CUSTOM_RECEPTORS 0
namespace snn {
namespace custom_receptors {
static const char *const TYPE_NAME = "";
constexpr int NTYPES = 0;
static const char *const NT_NAMES[3] = {"", "", ""};
constexpr int NVARS = 0;
constexpr int NSTORE = 1;
static const char *const NAMES[NSTORE] = {""};
static const float DEFAULTS[NSTORE] = {0.0f};
constexpr int CURRENT_INDEX[3] = {-1, -1, -1};
constexpr bool MULTI_STATE = false;
__device__ __forceinline__ void update_kinetics(int, float, float, float (&)[NSTORE]) {}
__device__ __forceinline__ void iterate(int, float, float, float (&)[NSTORE]) {}
} // namespace custom_receptors
} // namespace snn
#endif

namespace snn {
constexpr int CUSTOM_MODEL = 100;        // SNN_MODEL_CUSTOM
constexpr int CUSTOM_SPIKE_TRAIN = 100;  // SNN_ST_CUSTOM
constexpr uint32_t CUSTOM_REFRACTORINESS = 2;
constexpr int CUSTOM_KINETICS = 100;     // SNN_NT_CUSTOM / SNN_RC_CUSTOM
constexpr int CUSTOM_MAX_VARS = 32, CUSTOM_ST_MAX_VARS = 16, CUSTOM_REFR_MAX_VARS = 8, CUSTOM_KINETICS_MAX_VARS = 8;
static_assert(custom_nt::NVARS <= CUSTOM_KINETICS_MAX_VARS, "too many variables in the generated neurotransmitter kinetics");
static_assert(custom_rc::NVARS <= CUSTOM_KINETICS_MAX_VARS, "too many variables in the generated receptor kinetics");
constexpr int CUSTOM_RECEPTORS_MAX_VARS = 32;
static_assert(custom_receptors::NVARS <= CUSTOM_RECEPTORS_MAX_VARS, "too many variables in the generated receptor set");
static_assert(custom::NVARS <= CUSTOM_MAX_VARS, "too many variables in the generated neuron model");
static_assert(custom_st::NVARS <= CUSTOM_ST_MAX_VARS, "too many variables in the generated spike train");
static_assert(custom_refr::NVARS <= CUSTOM_REFR_MAX_VARS, "too many variables in the generated refractoriness");
} // namespace snn
