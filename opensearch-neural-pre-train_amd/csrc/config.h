// Process-wide configuration of libsnx.so.  The library reads NO environment variable: every choice below is made through
// `snx_configure(key, value)` (include/snx.h) -- by the Python binding from its own SNX_* variables at load time
// (snx/_lib.py), by tests and tools directly.  The second block exists in diagnostics builds only (-DSNX_DIAG): those
// switches change which kernel variant or schedule a concluded experiment used; the product build compiles their
// defaults in.
#pragma once

struct SnxConfig {
  int nt256 = 1;              // 256x256 persistent NT GEMM: 0 off, 1 default shape policy, 2 every eligible shape
  int nt256_min_m = 8192;     //   ... from this many rows on
  int nt256_coldeal = 1;      //   ... leftover 64-row units dealt along column runs: one short tile per workgroup (0: in tile
                              //   order, two short tiles for three workgroups in eight; A/B).  Same bits either way.
  int nt256_rev = 0;          //   ... 1: the dX GEMMs with K >= 3 N walk every super-block's row panels from the last to the first
                              //   (their A operand's last-written rows are the ones still in the Infinity Cache)
  int tn256 = 1;              // 256x256 persistent weight-gradient GEMM (0: the 128x128 kernel everywhere)
  int tn256_min_m = 8192;
  int dec256 = 1;             // 256x192 persistent decoder + SPLADE kernel (0: the 128x128 kernel)
  int dec256_min_t = 2048;
  int bwd_overlap = 1;        // weight-gradient GEMMs of the backward on the internal side stream
  int side_prio = 1;          //   ... which has the lowest stream priority
  int attn_streaming = 0;     // 1: tile-by-tile attention kernels for every sequence length
  int attn_bwd_onepass = 1;   // one-pass attention backward for sequences of <= 256 tokens (0: dQ + dK/dV pair)
  int attn_interleave = 0;    // 1: units of the sequence groups interleaved in proportion (attention_common.h); default 0 =
                              // group by group, longest first -- measured level in the step (45.08 against 45.01 ms, ABAB)
  int splade_dh_panels = 64;  // vocabulary panels of the routed decoder backward's dHd gather (0: one wave per row); rounds 3-5: 16,
                              // round 6 with the activation half first and nt bucket lists: 64 (3.07 / 3.12 / 3.21 ms at 64 / 32 / 16)
  int splade_dw_last = 2;     // routed decoder backward's weight half AFTER its activation half, gradient rows non-temporal: the dHd
                              // gather finds W_E where the decoder forward left it (44.17 against 44.27 ms, ABAB; 0: before);
                              // 2: also the bucket lists (read once) through non-temporal loads
  int f32_gemm64 = 0;         // fp32 path: the 64x64 GEMM tile for every shape
  int f32_attn_rows = 0;      // fp32 path: wave-per-(token, head) attention forward
  int wcache_per_tensor = 0;  // bf16 weight cache refreshed one launch per tensor
  int resid_in_ln = 1;        // forward: Wo GEMMs store bf16, the residual add happens inside the following LayerNorm
                              // (0: in the GEMM's fp32 epilogue; same bits, 0.17 ms per micro-step slower)
  int nt_pipe = 2;            // GeGLU-backward GEMM on the helper-wave kernel (gemm_nt_pipe.hip): 2 = plain du stores (default:
                              // the next two GEMMs read du), 1 = non-temporal du stores (faster alone, equal in the step),
                              // 0 = gemm.hip's 128x128 kernel
  int nt_pipe_min_m = 4096;
  int stream_nt = 271;         // non-temporal accesses of streams whose bytes have no reader soon (bitmask): 1 LayerNorm forward's
                              // loads of h and y, 2 its store of h_out, 4 LayerNorm backward's loads of the saved h and of dy,
                              // 8 the GeGLU-forward GEMM's stores of the saved u, 16 the weight-gradient GEMM's operand LDS-DMA,
                              // 32 the attention backward's loads of q, k, v,
                              // dO, 64 the GeGLU-backward GEMM's loads of the saved u and the attention forward's loads of q, k, v, 128 LayerNorm backward's load + store of the fp32 gradient stream dh, 256 the weight-gradient
                              // GEMM's ordered reduce (slab loads, gradient-tile read-modify-write).  Default 271 = 15 + 256: 15 measured
                              // 43.72 against 44.22 ms per micro-step (three ABA rounds on one box; the NT GEMM classes gain
                              // 0.45 ms: their operands stay cached); 16 costs 0.3 ms, 32 and 64 measured level
  int det_reduce = 1;         // weight gradients reduced in a FIXED order (partial slabs in the caller's workspace + an
                              // ordered reduction; bit-reproducible).  0: float atomics in arrival order (rounds 1-4; A/B)
#ifdef SNX_DIAG
  int gemm_cg = -1;           // column-group width of the 128x128 NT tile order (-1: cost model)
  int gemm_dbg = 0;           // 1: L2-resident operands, 2: no epilogue
  int gemm_mid = (1 << 1) | (1 << 4);   // epilogues (bitmask over EPI) on the mid-step-barrier main loop
  int tn_splits = 0;          // token splits of the 128x128 weight-gradient GEMM (0: fill model)
  int nt256_cg = -1;
  int nt256_dbg = 0;          // 1 no write-back, 2 L2-resident operands, 4 write-back without stores, 16 no deep request
  int nt256_force = 0;        // epilogues (bitmask) that take every eligible shape
  int tn256_tail_pct = 95;
  int tn256_dbg = 0;          // 1 no atomics, 2 no DMA, 4 L2-resident operands
#endif
};
extern SnxConfig g_snx_cfg;

#ifdef SNX_DIAG
#define SNX_DIAG_CFG(field, dflt) (g_snx_cfg.field)
#else
#define SNX_DIAG_CFG(field, dflt) (dflt)
#endif
