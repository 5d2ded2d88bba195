// Hook for ONE generated neuron model (model id SNN_MODEL_CUSTOM): spiking-neural-networks_amd/modelgen.py turns a
// neuron description in the reference's neuron_builder! DSL (build_test/nb_macro) into a header with the model's
// variable table and its on_iteration / spike_detection / on_spike as device functions; a library compiled with
// -DSNN_CUSTOM_MODEL_HEADER="\"generated/<name>.hpp\"" carries that model next to the built-in ones.  Without the
// define the hook is empty and SNN_MODEL_CUSTOM is refused.
#pragma once
#include "snn_math.hpp"

#ifdef SNN_CUSTOM_MODEL_HEADER
#include SNN_CUSTOM_MODEL_HEADER
#define SNN_HAVE_CUSTOM_MODEL 1
#else
#define SNN_HAVE_CUSTOM_MODEL 0
namespace snn {
namespace custom {
constexpr int NVARS = 0;
constexpr int NSTORE = 1;
static const char *const TYPE_NAME = "";
static const char *const NAMES[NSTORE] = {""};
static const float DEFAULTS[NSTORE] = {0.0f};
constexpr float DEFAULT_VOLTAGE = 0.0f, DEFAULT_DT = 0.1f, DEFAULT_C_M = 1.0f, DEFAULT_GAP = 10.0f;
__device__ __forceinline__ void on_iteration(float &, float (&)[NSTORE], float, float, float, float) {}
__device__ __forceinline__ bool spike_detection(float, const float (&)[NSTORE], float, float, float, float) { return false; }
__device__ __forceinline__ void on_spike(float &, float (&)[NSTORE], float, float, float, float) {}
} // namespace custom
} // namespace snn
#endif

namespace snn {
constexpr int CUSTOM_MODEL = 100;        // SNN_MODEL_CUSTOM
constexpr int CUSTOM_MAX_VARS = 32;
static_assert(custom::NVARS <= CUSTOM_MAX_VARS, "too many variables in the generated model");
} // namespace snn
