// snx_configure / snx_config_get: the only way into the library's process-wide switches (config.h).
#include <string.h>

#include "common.h"
#include "config.h"
#include "snx.h"

SnxConfig g_snx_cfg;

namespace {
struct Key { const char* name; int SnxConfig::*field; int lo, hi; };
const Key KEYS[] = {
    {"nt256", &SnxConfig::nt256, 0, 2},
    {"nt256_min_m", &SnxConfig::nt256_min_m, 1, 1 << 30},
    {"nt256_coldeal", &SnxConfig::nt256_coldeal, 0, 1},
    {"nt256_rev", &SnxConfig::nt256_rev, 0, 1},
    {"tn256", &SnxConfig::tn256, 0, 1},
    {"tn256_min_m", &SnxConfig::tn256_min_m, 1, 1 << 30},
    {"dec256", &SnxConfig::dec256, 0, 1},
    {"dec256_min_t", &SnxConfig::dec256_min_t, 1, 1 << 30},
    {"bwd_overlap", &SnxConfig::bwd_overlap, 0, 1},
    {"side_prio", &SnxConfig::side_prio, 0, 1},
    {"attn_streaming", &SnxConfig::attn_streaming, 0, 1},
    {"attn_bwd_onepass", &SnxConfig::attn_bwd_onepass, 0, 1},
    {"attn_interleave", &SnxConfig::attn_interleave, 0, 1},
    {"splade_dh_panels", &SnxConfig::splade_dh_panels, 0, 64},
    {"splade_dw_last", &SnxConfig::splade_dw_last, 0, 2},
    {"f32_gemm64", &SnxConfig::f32_gemm64, 0, 1},
    {"f32_attn_rows", &SnxConfig::f32_attn_rows, 0, 1},
    {"wcache_per_tensor", &SnxConfig::wcache_per_tensor, 0, 1},
    {"resid_in_ln", &SnxConfig::resid_in_ln, 0, 1},
    {"det_reduce", &SnxConfig::det_reduce, 0, 1},
    {"stream_nt", &SnxConfig::stream_nt, 0, 511},
    {"nt_pipe", &SnxConfig::nt_pipe, 0, 2},
    {"nt_pipe_min_m", &SnxConfig::nt_pipe_min_m, 1, 1 << 30},
#ifdef SNX_DIAG
    {"gemm_cg", &SnxConfig::gemm_cg, -1, 64},
    {"gemm_dbg", &SnxConfig::gemm_dbg, 0, 3},
    {"gemm_mid", &SnxConfig::gemm_mid, 0, 31},
    {"tn_splits", &SnxConfig::tn_splits, 0, 64},
    {"nt256_cg", &SnxConfig::nt256_cg, -1, 64},
    {"nt256_dbg", &SnxConfig::nt256_dbg, 0, 31},
    {"nt256_force", &SnxConfig::nt256_force, 0, 31},
    {"tn256_tail_pct", &SnxConfig::tn256_tail_pct, 0, 100},
    {"tn256_dbg", &SnxConfig::tn256_dbg, 0, 7},
#endif
};
const Key* find(const char* key) {
  if (!key) return nullptr;
  for (const Key& k : KEYS)
    if (!strcmp(k.name, key)) return &k;
  return nullptr;
}
}  // namespace

extern "C" int snx_configure(const char* key, int32_t value) {
  const Key* k = find(key);
  if (!k || value < k->lo || value > k->hi) return SNX_E_ARG;
  g_snx_cfg.*(k->field) = value;
  return SNX_OK;
}

extern "C" int snx_config_get(const char* key, int32_t* value) {
  const Key* k = find(key);
  if (!k || !value) return SNX_E_ARG;
  *value = g_snx_cfg.*(k->field);
  return SNX_OK;
}
