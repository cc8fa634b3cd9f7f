// Thread-local error string + ABI version for libparadis_hip.
#include <stdarg.h>
#include <stdlib.h>
#include "common.h"

static thread_local char g_err[512] = "";

void paradis_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

bool paradis_deterministic() {
  static const int on = [] {
    const char* e = getenv("PARADIS_DETERMINISTIC");
    return (e && e[0] && e[0] != '0') ? 1 : 0;
  }();
  return on != 0;
}

extern "C" const char* paradis_last_error(void) { return g_err; }
extern "C" int paradis_abi_version(void) { return 9; }   // 9: round 6, bf16-STORED tensors of the bf16-mixed mode: paradis_pw_gemm_fwd16 / _dgrad16 / _wgrad16, paradis_bias_grads16 (additions, no signature changed; paradis_pw_gemm_wgrad_ws_bytes shrank by 4 KiB); 8: round-5 additions (paradis_pw_gemm_wgrad_slabs, ...); 7: paradis_dwconv_geo_dgrad_add, paradis_dwconv_geo_bwd, paradis_pw_gemm_split_weights_pair, paradis_pw_gemm_fwd_gated, paradis_gated_blend_bwd_out; 6: advection workspace sized per flags (strip schedule); 5: amax side outputs removed (4: GEMM scheme arguments + paradis_amax_partials)
