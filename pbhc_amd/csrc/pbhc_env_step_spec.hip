// pbhc_env_step_spec.hip — ONE config-specialised instance of the fused env-step kernel.
//
// Built on request (pbhc_amd/specialise.py, or `make spec` for the fixture configs) as its own small shared object:
//   hipcc ... -DPBHC_STATIC_CFG='"<key>.h"' -DPBHC_SPEC_MODE=<0|1> -shared -o <key>.so pbhc_env_step_spec.hip
// where <key>.h holds `static constexpr PbhcEnvConfig kStaticCfg = {...}`: the env's config as pbhc_env_create finalised it
// (pbhc_env_get_config), pointers nulled.  Every config scalar of k_env_step then folds into the instruction stream: the switches that
// are off disappear, loop bounds (dofs, bodies, feet, terms) are literals, feature offsets become LDS immediates and the ~100 scalar
// loads of the generic kernel go away (14.0 k -> 8.6 k static instructions for the 23-DoF walk config).  Pointers, env count, seed and the
// reference yaw still come from the run-time config, so one object serves every env count.
// pbhc_env_attach_specialised (pbhc_kernels.hip) compares pbhc_spec_config() with the env's own config field by field before it
// accepts the kernel: a specialised kernel can never run against a config it was not built from.
#include <hip/hip_runtime.h>

#include "../../include/pbhc_hip.h"
#include "pbhc_math.h"

#ifndef PBHC_STATIC_CFG
#error "pbhc_env_step_spec.hip needs -DPBHC_STATIC_CFG=<header with kStaticCfg>"
#endif
#ifndef PBHC_SPEC_MODE
#error "pbhc_env_step_spec.hip needs -DPBHC_SPEC_MODE=<tracking_mode of the config>"
#endif

using namespace pbhc;

#include "pbhc_env_step.h"

static_assert(kStaticCfg.tracking_mode == PBHC_SPEC_MODE, "PBHC_SPEC_MODE must equal the config's tracking_mode");

extern "C" {
int pbhc_spec_abi_version(void) { return PBHC_ABI_VERSION; }
int pbhc_spec_mode(void) { return PBHC_SPEC_MODE; }
const PbhcEnvConfig* pbhc_spec_config(void) { return &kStaticCfg; }
const void* pbhc_spec_kernel(void) { return (const void*)k_env_step<PBHC_SPEC_MODE>; }
// this build's LDS plan (step_lds_plan: no compact maps when the rows are unrolled runs, the history block out of the feature row)
int pbhc_spec_lds_stride(void) { return step_lds_plan(kStaticCfg, obs_runs_complete(kStaticCfg)).stride; }
int pbhc_spec_lds_bytes(void) { return step_lds_plan(kStaticCfg, obs_runs_complete(kStaticCfg)).bytes; }
// 1: this build reads the history rows 16 bytes per lane (the launch then requires them 16-byte aligned with a pitch that is a multiple of 4 floats)
int pbhc_spec_hist_wide(void) { return step_lds_plan(kStaticCfg, obs_runs_complete(kStaticCfg)).hist_in_bodies; }
#ifdef PBHC_STAMPS          // diagnostic builds only (PBHC_SPEC_DEFINES=-DPBHC_STAMPS, tools/kernel_probe.py): this object's own stamp buffers
int pbhc_spec_read_stamps(unsigned long long* out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return PBHC_EHIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (n < 64 ? n : 64)) == hipSuccess ? PBHC_OK : PBHC_EHIP;
}
int pbhc_spec_read_wg_times(unsigned long long* out, int num_workgroups) {
  if (hipDeviceSynchronize() != hipSuccess) return PBHC_EHIP;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_times), sizeof(unsigned long long) * 2 * (num_workgroups < 4096 ? num_workgroups : 4096)) == hipSuccess ? PBHC_OK : PBHC_EHIP;
}
#endif
}
