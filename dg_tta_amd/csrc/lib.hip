// Library-level entry points: version and thread-local error message.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void dgtta_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dgtta_version(void) { return 60000; /* 6.0.0: dgtta_softdice_bwd_t, dgtta_seghead_warp_bwd_g16 (logit gradient in the storage type); 5.1.0: feature-space window accumulation (dgtta_feature_*); 5.0.0: dice_ce_fwd / bwd (pre-training loss), argmax_rows for > 112 classes, laboratory switches out of the product build */ }
extern "C" const char *dgtta_last_error(void) { return g_err; }

// ---------------------------------------------------------------- environment switches (snapshot, see common.h)
#include <mutex>
#include <stdlib.h>

static std::atomic<const DgttaSwitches *> g_switches{nullptr};
static std::once_flag g_switches_once;


// strict: a switch is the single character the documentation names; anything else ('off', 'no', '7x', ...) counts as unset
static int env_char(const char *name) {
  const char *v = getenv(name);
  return (v && v[0] && !v[1] && v[0] >= '0' && v[0] <= '9') ? (int)(unsigned char)v[0] : -1;
}

static const DgttaSwitches *read_switches() {
  DgttaSwitches *s = new DgttaSwitches;       // snapshots are immutable and never freed (a handful of bytes per reload)
  s->conv_rows = env_char("DGTTA_CONV_ROWS");
  s->conv_s2 = env_char("DGTTA_CONV_S2");
  s->convt_gemm = env_char("DGTTA_CONVT_GEMM");
  s->rows_order = env_char("DGTTA_ROWS_ORDER");
  s->wgrad_upw = env_char("DGTTA_WGRAD_UPW");
  s->wgrad_xcd = env_char("DGTTA_WGRAD_XCD");
  s->in_nt = env_char("DGTTA_IN_NT");
  s->in_nt = s->in_nt == '0' ? '0' : '1';       // default on: streaming loads / stores in the InstanceNorm apply passes
  s->dgrad_s2_allcls = env_char("DGTTA_DGRAD_S2_ALLCLS");
  s->wgrad_tr = env_char("DGTTA_WGRAD_TR");
  s->wgrad_tr8 = env_char("DGTTA_WGRAD_TR8");
  s->wgrad_s2_onepass = env_char("DGTTA_WGRAD_S2_ONEPASS");
  s->convt_wgrad_onepass = env_char("DGTTA_CONVT_WGRAD_ONEPASS");
  s->in_gstats = env_char("DGTTA_IN_GSTATS");
  s->softdice16 = env_char("DGTTA_SOFTDICE16");
  s->conv_ring = env_char("DGTTA_CONV_RING");
  s->wgrad_ring = env_char("DGTTA_WGRAD_RING");
  s->ha_mfma = env_char("DGTTA_HA_MFMA");
  s->wgrad_f32_split = env_char("DGTTA_WGRAD_F32_SPLIT");
  s->feature_head_mfma = env_char("DGTTA_FEATURE_HEAD_MFMA");
  s->headwarp_mfma = env_char("DGTTA_HEADWARP_MFMA");
  s->wgrad_flat = env_char("DGTTA_WGRAD_FLAT");
  s->wgrad_reduce_taps = env_char("DGTTA_WGRAD_REDUCE_TAPS");
  // product switches select between kernels of equal results only: values outside a switch's documented set are ignored
  if (s->conv_ring != '0' && s->conv_ring != '1' && s->conv_ring != '3') s->conv_ring = -1;
  if (s->wgrad_ring != '0' && s->wgrad_ring != '1' && s->wgrad_ring != '4' && s->wgrad_ring != '5' && s->wgrad_ring != '6') s->wgrad_ring = -1;
  if (s->convt_gemm != '0' && s->convt_gemm != '1') s->convt_gemm = -1;
  s->rows_abl = s->rows_var = s->ring_nt = s->ring_abl = s->wgrad_ring_lab = s->ha_abl = s->warp_abl = s->convt_gemm_abl = -1;
  s->ncu = 0;
#ifdef DGTTA_DIAG
  s->rows_abl = env_char("DGTTA_ROWS_ABL");
  s->convt_gemm_abl = env_char("DGTTA_CONVT_GEMM_ABL");      // '2' / '3': the transposed-conv GEMM without stores / without MFMAs
  s->rows_var = env_char("DGTTA_ROWS_VAR");
  s->ring_nt = env_char("DGTTA_RING_NT");
  s->ring_abl = env_char("DGTTA_RING_ABL");
  s->wgrad_ring_lab = env_char("DGTTA_WGRAD_RING_CLK");      // '6': cycle stamps behind the slabs (profiles/tools/wring_clock.py)
  s->ha_abl = env_char("DGTTA_HA_ABL");
  s->warp_abl = env_char("DGTTA_WARP_ABL");
  if (const char *v = getenv("DGTTA_NCU")) s->ncu = atoi(v);
#endif
  return s;
}

const DgttaSwitches &dgtta_switches() {
  std::call_once(g_switches_once, [] { g_switches.store(read_switches(), std::memory_order_release); });
  return *g_switches.load(std::memory_order_acquire);
}

extern "C" int dgtta_reload_env(void) {
  (void)dgtta_switches();
  g_switches.store(read_switches(), std::memory_order_release);
  return DGTTA_OK;
}
