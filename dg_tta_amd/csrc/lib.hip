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

extern "C" int dgtta_version(void) { return 40000; /* 4.0.0: accumulator storage type (window_accumulate_t, seghead_window_accumulate_t, logits_chunk_f64_t), argmax_rows */ }
extern "C" const char *dgtta_last_error(void) { return g_err; }

// ---------------------------------------------------------------- environment switches (snapshot, see common.h)
#include <mutex>
#include <stdlib.h>

static std::atomic<const DgttaSwitches *> g_switches{nullptr};
static std::once_flag g_switches_once;

static int env_char(const char *name) {
  const char *v = getenv(name);
  return (v && v[0]) ? (int)(unsigned char)v[0] : -1;
}

static const DgttaSwitches *read_switches() {
  DgttaSwitches *s = new DgttaSwitches;       // snapshots are immutable and never freed (a handful of bytes per reload)
  s->conv_rows = env_char("DGTTA_CONV_ROWS");
  s->conv_variant = env_char("DGTTA_CONV_VARIANT");
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
  s->conv_abl = env_char("DGTTA_CONV_ABL");
  s->rows_abl = env_char("DGTTA_ROWS_ABL");
  s->rows_var = env_char("DGTTA_ROWS_VAR");
  s->warp_coop = env_char("DGTTA_WARP_COOP");
  s->warp_nt = env_char("DGTTA_WARP_NT");
  s->warp_xcd = env_char("DGTTA_WARP_XCD");
  s->in_gstats = env_char("DGTTA_IN_GSTATS");
  s->softdice16 = env_char("DGTTA_SOFTDICE16");
  s->wgrad_abl = env_char("DGTTA_WGRAD_ABL");
  s->conv_ring = env_char("DGTTA_CONV_RING");
  s->ring_nt = env_char("DGTTA_RING_NT");
  s->ring_abl = env_char("DGTTA_RING_ABL");
  s->wgrad_ring = env_char("DGTTA_WGRAD_RING");
  s->ha_abl = env_char("DGTTA_HA_ABL");
  s->ha_mfma = env_char("DGTTA_HA_MFMA");
  s->warp_abl = env_char("DGTTA_WARP_ABL");
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
