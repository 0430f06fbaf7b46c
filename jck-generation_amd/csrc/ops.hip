// C-ABI launchers for the per-op entry points of include/jckgan.h (host side; kernels in *.hpp).
#include "ops_internal.hpp"
#include "thin.hpp"
#include "bnres.hpp"

#include <cstdlib>
#include <cstring>
#include <string>

static thread_local std::string g_err;
void jck_set_error(const std::string& s) { g_err = s; }
extern "C" const char* jck_last_error(void) { return g_err.c_str(); }
// hash of csrc/ + include/jckgan.h at build time (hipgan/build.py passes it; the loader refuses a binary whose answer differs from
// the sources beside it); 100 for a hand-made build without the define
#ifndef JCK_BUILD_ID
#define JCK_BUILD_ID 100
#endif
extern "C" int jck_version(void) { return JCK_BUILD_ID; }
extern "C" int jck_pad_rows(int c) { return c <= 16 ? 16 : (c <= 64 ? 64 : (c + 127) / 128 * 128); }
extern "C" int jck_pad_chan(int c) { return c == 3 ? 4 : c; }

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// ---------------------------------------------------------------------------------------------------------
// kernel-selection knobs: defaults are the measured-best choices; each can be preset with an environment variable of the
// same name (JCK_<KEY>) or changed at run time with jck_tune("<key>", value) - which is what lets one process A/B two
// variants on the same device and lets a test force a variant at a small shape.
// ---------------------------------------------------------------------------------------------------------
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static int g_igemm_dma = env_int("JCK_IGEMM_DMA", 1);        // LDS-DMA gather-GEMM for bf16 tiles
static int g_igemm_ws = env_int("JCK_IGEMM_WS", 1);          // wave-specialised 128x64 tiles when < 512 tiles of 128x128
static int g_igemm_256 = env_int("JCK_IGEMM_256", 250);      // minimum number of 128x256 tiles to take that kernel (0: never)
static int g_igemm_eff = env_int("JCK_IGEMM_EFF", 1);        // prefer 128x128 tiles where 128x256 tiles leave the last round of workgroups half empty
static int g_igemm_128 = env_int("JCK_IGEMM_128", 200);      // minimum number of 128x128 tiles to take the persistent kernel at that tile (0: never)
static int g_igemm_persist = env_int("JCK_IGEMM_PERSIST", 7);   // persistent wave-specialised gather-GEMMs (igemm.hpp); bit 0: 128x256, 1: 128x64, 2: 64x128 tiles
// s_setprio 1 for the loader waves of the wave-specialised kernels: the younger half of a workgroup loses the issue arbitration
// (MI355X guide, "Two waves per SIMD", item 4), and the kernels are bound by how fast the loaders issue their LDS-DMA pieces -
// +0.3..3 % per gather-GEMM (tools/mb2.py igemm_prio 0 1 ...)
static int g_igemm_dbg = 0;            // JCK_DIAG builds only: timing-experiment variant of the persistent gather-GEMM
static int g_igemm_prio = env_int("JCK_IGEMM_PRIO", 1);      // in the step: neutral (1.905 vs 1.907 ms)
static int g_wgrad_prio = env_int("JCK_WGRAD_PRIO", 0);
static int g_stat_accum = env_int("JCK_STAT_ACCUM", 1);         // forward statistics accumulated per workgroup (persistent kernels, *_grouped calls)
static int g_bn_unr = env_int("JCK_BN_UNR", 2);
static int g_bn_res_mb = env_int("JCK_BN_RES_MB", 120);         // multi-group passes: resident form above this many MB of (g_a, y)
static int g_bn_fuse = env_int("JCK_BN_FUSE", 1);               // forward BatchNorm finalize + apply as one launch (bn_fwd_fused_kernel) ...
static int g_bn_fuse_wgs = env_int("JCK_BN_FUSE_WGS", 256);     // ... on about this many workgroups (each re-reads its slice's rows)
static int g_bn_fuse_rows = env_int("JCK_BN_FUSE_ROWS", 320);   // ... while a group has at most this many statistics rows
static int g_igemm_dma_ksplit = env_int("JCK_IGEMM_DMA_KSPLIT", 1);
static int g_bn_bwd_fuse = env_int("JCK_BN_BWD_FUSE", 1);       // three-launch BatchNorm backward as two: the apply sums its slice's partial rows itself (bn_bwd_apply_fused_kernel) ...
static int g_bn_bwd_fuse_wgs = env_int("JCK_BN_BWD_FUSE_WGS", 256);   // ... on about this many workgroups
static int g_bn_res_small_mb = env_int("JCK_BN_RES_SMALL_MB", 0);   // multi-group passes: resident form also at or below this many MB (launch-latency-bound layers)
static int g_bn_res = env_int("JCK_BN_RES", 1);                  // resident one-launch BatchNorm backward (bnres.hpp); 0: reduce + sums + apply, 2: whenever it fits                  // rows in flight per thread in bn_bwd_reduce (1, 2, 4)
static int g_thin = env_int("JCK_THIN", 1);                  // streaming kernels for the image-side layers
static int g_wgrad_gt = env_int("JCK_WGRAD_GT", 1);          // 2: 256-column weight-gradient tile (measured: no gain)
static int g_wgrad_wgs = env_int("JCK_WGRAD_WGS", 256);      // split-K target workgroups
static int g_wgrad_small_wgs = env_int("JCK_WGRAD_SMALL_WGS", 512);
static int g_wgrad_stamp = env_int("JCK_WGRAD_STAMP", 0);
static int g_wgrad_ws = env_int("JCK_WGRAD_WS", 1);
static int g_wgrad_dma = env_int("JCK_WGRAD_DMA", 1);
static int g_wgrad_dbg = 0;             // JCK_DIAG builds only: timing-experiment variant of the weight-gradient kernel (wgrad.hpp WDBG)
static int g_wgrad_pipe = env_int("JCK_WGRAD_PIPE", 1);      // software-pipelined consumer waves of the wave-specialised weight gradient
extern "C" int jck_tune(const char* key, int value) {
  struct { const char* k; int* p; } tab[] = {{"igemm_dma", &g_igemm_dma}, {"igemm_ws", &g_igemm_ws}, {"igemm_256", &g_igemm_256}, {"igemm_128", &g_igemm_128}, {"igemm_eff", &g_igemm_eff}, {"igemm_persist", &g_igemm_persist}, {"igemm_dbg", &g_igemm_dbg}, {"igemm_prio", &g_igemm_prio}, {"wgrad_prio", &g_wgrad_prio},  {"stat_accum", &g_stat_accum}, {"bn_unr", &g_bn_unr}, {"bn_res", &g_bn_res}, {"bn_res_mb", &g_bn_res_mb}, {"bn_fuse", &g_bn_fuse}, {"bn_fuse_rows", &g_bn_fuse_rows}, {"bn_fuse_wgs", &g_bn_fuse_wgs}, {"igemm_dma_ksplit", &g_igemm_dma_ksplit}, {"bn_bwd_fuse", &g_bn_bwd_fuse}, {"bn_bwd_fuse_wgs", &g_bn_bwd_fuse_wgs}, {"bn_res_small_mb", &g_bn_res_small_mb},
                                              {"thin", &g_thin}, {"wgrad_gt", &g_wgrad_gt}, {"wgrad_wgs", &g_wgrad_wgs},
                                              {"wgrad_small_wgs", &g_wgrad_small_wgs}, {"wgrad_stamp", &g_wgrad_stamp},
                                              {"wgrad_ws", &g_wgrad_ws}, {"wgrad_dma", &g_wgrad_dma}, {"wgrad_pipe", &g_wgrad_pipe}, {"wgrad_dbg", &g_wgrad_dbg}};
  for (auto& t : tab)
    if (key && !strcmp(t.k, key)) { *t.p = value; return JCK_OK; }
  JCK_FAIL(JCK_E_ARG, std::string("jck_tune: unknown key ") + (key ? key : "(null)"));
}

#include <hip/hip_ext.h>
// launch whose completion hands a tensor to another stream: `ev` (may be null) is completed by the dispatch packet itself
// (hipExtLaunchKernel's stop event) - what a hipEventRecord behind the launch would do with a marker packet of its own, which
// costs the launch stream ~6-7 us of idle time per record on this runtime.  Every kernel argument must be passed explicitly
// (the extended launch checks the count).
#define LAUNCH_EV(kernel, grid, block, shmem, stream, ev, ...)                                                      \
  do {                                                                                                              \
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, (hipEvent_t) nullptr, (hipEvent_t)(ev), 0u, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                       \
  } while (0)
#define DISPATCH_T(prec, CALL)                                  \
  do {                                                          \
    if ((prec) == JCK_PREC_BF16) { typedef bf16_t T; CALL; }    \
    else if ((prec) == JCK_PREC_F32) { typedef float T; CALL; } \
    else JCK_FAIL(JCK_E_ARG, "bad prec");                       \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// optional per-launch timing of the MFMA kernels with HIP events on the launch stream (bench.py's roofline
// leg).  Off by default: zero cost in the timed region.
// ---------------------------------------------------------------------------------------------------------
#include <vector>
namespace {
// a launch is priced in algorithmic FLOPs (MFMA kernels) or algorithmic bytes (the streaming BatchNorm kernels: bytes > 0)
struct ProfRec { int variant; double flops, bytes; hipStream_t st; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
const char* const PROF_NAMES[] = {"igemm<bf16,128,128>", "igemm<bf16,128,64>", "igemm<bf16,64,128,img>", "igemm<bf16,64,128>",
                                  "igemm<bf16,16,256>",  "igemm<f32,128,128>", "igemm<f32,128,64>",      "igemm<f32,64,128,img>",
                                  "igemm<f32,64,128>",   "igemm<f32,16,256>",  "wgrad<bf16,128,128>",    "wgrad<bf16,128,64>",
                                  "wgrad<bf16,64,64,img>", "wgrad<bf16,64,64>", "wgrad<f32,128,128>",    "wgrad<f32,128,64>",
                                  "wgrad<f32,64,64,img>", "wgrad<f32,64,64>",  "img_down<bf16>",         "img_up<bf16>",
                                  "igemm<bf16,128,256>",  "wgrad<bf16,256,128>", "bn_act_fwd",           "bn_bwd_resident",
                                  "bn_bwd_3launch"};
#define PROF_BN_ACT_FWD 22
#define PROF_BN_BWD_RES 23
#define PROF_BN_BWD_3L 24
struct ProfScope {
  ProfRec r; bool on; hipStream_t st;
  ProfScope(int variant, double flops, hipStream_t s, double bytes = 0.0) : on(g_prof_on), st(s) {
    if (!on) return;
    r.variant = variant; r.flops = flops; r.bytes = bytes; r.st = s;
    (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.e1, st);
    g_prof.push_back(r);
  }
};
}  // namespace
bool jck_prof_is_on() { return g_prof_on; }
extern "C" int jck_prof_enable(int on) {
  g_prof_on = on != 0;
  return JCK_OK;
}
// Synchronises the recorded events, accumulates per (kernel variant, HIP stream the launches ran on): count, total ms, total
// algorithmic FLOPs, total algorithmic bytes (streaming kernels).  A variant launched on two streams comes back as two rows.
// Returns the number of rows written (<= cap).  name_out[i] points at a static string.
extern "C" int jck_prof_collect(int cap, const char** name_out, int* count_out, double* ms_out, double* flops_out, double* bytes_out,
                                void** stream_out) {
  constexpr int NV = sizeof(PROF_NAMES) / sizeof(PROF_NAMES[0]);
  struct Row { int variant; hipStream_t st; int cnt; double ms, fl, by; };
  std::vector<Row> rows;
  for (auto& r : g_prof) {
    (void)hipEventSynchronize(r.e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.e0, r.e1);
    if (r.variant >= 0 && r.variant < NV) {
      Row* q = nullptr;
      for (auto& x : rows) if (x.variant == r.variant && x.st == r.st) { q = &x; break; }
      if (!q) { rows.push_back(Row{r.variant, r.st, 0, 0.0, 0.0, 0.0}); q = &rows.back(); }
      q->cnt++; q->ms += t; q->fl += r.flops; q->by += r.bytes;
    }
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  int n = 0;
  for (auto& x : rows) {
    if (n >= cap) break;
    name_out[n] = PROF_NAMES[x.variant]; count_out[n] = x.cnt; ms_out[n] = x.ms; flops_out[n] = x.fl;
    if (bytes_out) bytes_out[n] = x.by;
    if (stream_out) stream_out[n] = (void*)x.st;
    ++n;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------------------
// gather-GEMM dispatch
// ---------------------------------------------------------------------------------------------------------
template <class P, int BCH, int BPIX, int NSUB>
static int launch_igemm_t(const IgemmParams& p, int nch_pad, int phases, hipStream_t st, int* slots) {
  typedef IgemmCfg<P, BCH, BPIX> C;
  constexpr int variant = (P::IS_F32 ? 5 : 0) + (BCH == 128 ? (BPIX == 128 ? 0 : 1) : (BCH == 64 ? (NSUB == 2 ? 2 : 3) : 4));
  ProfScope prof(variant, p.flops, st);
  auto kern = igemm_kernel<P, BCH, BPIX, NSUB, 2>;
  static bool attr_done = false;
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_done = true;
  }
  dim3 grid(cdiv(p.M, BPIX), nch_pad / BCH, phases);
  IgemmParams q = p;
  q.gx = grid.x; q.gy = grid.y; q.gz = grid.z;
  if (q.stats) {
    if (q.cstat % BCH != 0 && BCH % q.cstat != 0) JCK_FAIL(JCK_E_ARG, "igemm: stats channel count incompatible with the tile");
    q.ytiles_per_cset = std::max(1, q.cstat / BCH);
    if (slots) *slots = (int)(grid.x * grid.z * (grid.y / q.ytiles_per_cset) * C::WPIX);
  }
  hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(256), C::LDS_BYTES, st, q);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

template <int BCH, int BPIX, int NSTG, bool WS = false, int NCW = 4>
static int launch_igemm_dma(const IgemmParams& p, int nch_pad, int phases, hipStream_t st, int* slots) {
  constexpr int LDSB = NSTG * (BCH + BPIX) * IG_BK * 2;
  constexpr int variant = BPIX == 256 ? 20 : BCH == 64 ? 3 : (BPIX == 128 ? 0 : 1);
  ProfScope prof(variant, p.flops, st);
  auto kern = igemm_dma_kernel<BCH, BPIX, NSTG, WS, NCW>;
  static bool attr_done = false;
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    attr_done = true;
  }
  dim3 grid(cdiv(p.M, BPIX), nch_pad / BCH, phases);
  IgemmParams q = p;
  q.gx = grid.x; q.gy = grid.y; q.gz = grid.z;
  if (q.stats) {
    q.ytiles_per_cset = std::max(1, q.cstat / BCH);
    if (slots) *slots = (int)(grid.x * grid.z * (grid.y / q.ytiles_per_cset) * IgemmCfg<PrecBf16, BCH, BPIX, NCW>::WPIX);
  }
  hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(WS ? (NCW + 4) * 64 : 256), LDSB, st, q);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// persistent wave-specialised form: at most `cap` workgroups (what the chip holds at this tile's LDS footprint) walk the tiles
template <int BCH, int BPIX, int NCW>
static int launch_igemm_dma_persist(const IgemmParams& p, int nch_pad, int phases, hipStream_t st, int* slots) {
  // three stages + the per-wave BatchNorm lane values and their arrival counters (igemm_wg_row)
  constexpr int LDSB = 3 * (BCH + BPIX) * IG_BK * 2 + NCW * (BCH >= 128 ? 2 : 1) * 256 + 64;
  constexpr int variant = BCH == 64 ? 3 : BPIX == 256 ? 20 : BPIX == 128 ? 0 : 1;
  ProfScope prof(variant, p.flops, st);
  // JCK_DIAG builds (hipgan/build.py with JCK_DIAG=1 in the environment) carry the timing-experiment variants of igemm.hpp - part of
  // the gather skipped, no loads / MFMAs / LDS reads / epilogue / stores / statistics (wrong results; jck_tune("igemm_dbg", code)):
  // the ablation table of DESIGN.md section 7 comes from them
#ifdef JCK_DIAG
#define JCK_DBG_VARIANT(code) g_igemm_dbg == code ? igemm_dma_persist_kernel<BCH, BPIX, NCW, code>
  auto kern = g_igemm_dbg == 2 ? igemm_dma_persist_kernel<BCH, BPIX, NCW, 2> : g_igemm_dbg == 4 ? igemm_dma_persist_kernel<BCH, BPIX, NCW, (BPIX >= 128 ? 4 : 2)>
              : JCK_DBG_VARIANT(101) : JCK_DBG_VARIANT(102) : JCK_DBG_VARIANT(103) : JCK_DBG_VARIANT(104) : JCK_DBG_VARIANT(105)
              : JCK_DBG_VARIANT(106) : JCK_DBG_VARIANT(107) : JCK_DBG_VARIANT(108) : igemm_dma_persist_kernel<BCH, BPIX, NCW>;
#undef JCK_DBG_VARIANT
#else
  auto kern = igemm_dma_persist_kernel<BCH, BPIX, NCW>;
#endif
  static const void* attr_done[16] = {};                             // the variants of this tile that have their LDS attribute set
  {
    int i = 0;
    while (i < 16 && attr_done[i] && attr_done[i] != reinterpret_cast<const void*>(kern)) ++i;
    if (i < 16 && !attr_done[i]) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done[i] = reinterpret_cast<const void*>(kern);
    }
  }
  dim3 grid(cdiv(p.M, BPIX), nch_pad / BCH, phases);
  IgemmParams q = p;
  q.gx = grid.x; q.gy = grid.y; q.gz = grid.z;
  q.loader_prio = g_igemm_prio;
  const int ntiles = (int)(grid.x * grid.y * grid.z);
  const int cap = 256 * (160 * 1024 / LDSB);                         // 256 CUs x workgroups that fit their LDS
  const int nwg = std::min(ntiles, cap);
  if (q.stats && q.stat_accum) {
    // accumulated rows need several tiles of ONE channel tile per workgroup and tiles inside one group (igemm.hpp)
    const int gyy = (int)grid.y;
    if (!(g_stat_accum && ntiles >= cap && ntiles % 8 == 0 && (ntiles / 8) % gyy == 0 && (cap / 8) % gyy == 0 && q.cstat == nch_pad &&
          (q.bn_group_rows == 0 || q.bn_group_rows % BPIX == 0)))
      q.stat_accum = 0;
  }
  if (q.stats && !q.stat_accum) {
    q.ytiles_per_cset = std::max(1, q.cstat / BCH);
    if (slots) *slots = (int)(grid.x * grid.z * (grid.y / q.ytiles_per_cset));            // one row per tile (igemm_wg_row)
  }
  if (q.stats && q.stat_accum) {    // accumulated forward statistics: rows [group][nwg / gy][2][cstat], one per workgroup (igemm.hpp)
    const int groups = q.bn_group_rows > 0 ? (q.M + q.bn_group_rows - 1) / q.bn_group_rows : 1;
    if (slots) *slots = groups * (nwg / (int)grid.y);
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3((NCW + 4) * 64), LDSB, st, q);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

template <class P>
static int launch_igemm_p(const IgemmParams& p, int nch_pad, int phases, int nsub, hipStream_t st, int* slots) {
  // bf16 tiles with >= 128 channel rows run on the LDS-DMA kernel with 2 LDS stages (64 / 48 KB -> 2-3 workgroups per CU,
  // which hide each other's load latency): 128x128 tiles while that still gives >= 512 workgroups, else 128x64.
  // Measured on MI355X at B=256 (tools/micro.py, us): down2 36.2 -> 29.7, down3 43.0 -> 30.3, down4 65.8 -> 40.3,
  // up2 47.2 -> 30.8, up3 34.5 -> 28.8; 3-4 stages at one workgroup per CU are slower.  JCK_IGEMM_DMA=0 disables.
  const int use_dma = g_igemm_dma;
  // (split-K plain GEMMs - CGAN's Linear(8392,256) - take the non-persistent LDS-DMA kernels: round 5, JCK_IGEMM_DMA_KSPLIT=0 the register-staged one)
  if (use_dma && !P::IS_F32 && nsub == 1 && nch_pad % 128 == 0 && (p.ksplit <= 1 || (g_igemm_dma_ksplit && p.act_row_elems)) && !p.rows_are_phases) {
    const long long wgs = (long long)cdiv(p.M, 128) * (nch_pad / 128) * phases;
    // wave-specialised variant for the launches that would run 128x64 tiles (< 512 tiles of 128x128: one or two workgroups
    // per CU); JCK_IGEMM_WS=0 disables.  128x128 WS (one workgroup per CU) and WS for the 64-channel tile measured slower.
    const int ws_mode = g_igemm_ws;
    // 128 x 256 tiles (8 consumer + 4 loader waves, one workgroup per CU) when that still gives about one tile per CU and a
    // tile cannot straddle two BatchNorm groups: groups are multiples of 8 images, so 8 * OH*OW must be a multiple of 256.
    // JCK_IGEMM_256 = minimum number of such tiles (0 disables).
    const int min256 = g_igemm_256;
    const long long wgs256 = (long long)cdiv(p.M, 256) * (nch_pad / 128) * phases;
    // persistent kernels: plain bf16 conv / dgrad launches only (their epilogue has no bias, tanh, fp32 or split-K output)
    int persist = (p.bias || p.epi || p.out_f32 || p.rows_are_phases || p.out_split_stride) ? 0 : g_igemm_persist;
    // a tile must not straddle two BatchNorm groups: groups are multiples of 8 images (8 * OH*OW % 256 == 0)
    const bool groups_ok = !p.stats || p.logOHW >= 5;
    // ... and the 256 persistent workgroups are not left half idle in their last round: 384 tiles are 1.5 rounds, the same layer in
    // 128 x 128 tiles is 3 full ones (round 5: D.conv3's forward at 3 x 256 images 60.6 -> 55.3 us)
    const auto round_eff = [](long long t) { return (double)t / (double)(((t + 255) / 256) * 256); };
    const bool prefer128 = g_igemm_eff && g_igemm_128 > 0 && wgs >= g_igemm_128 && round_eff(wgs256) < 0.85 && round_eff(wgs) > round_eff(wgs256) + 0.1;
    if (min256 > 0 && wgs256 >= min256 && !prefer128 && !p.act_row_elems && groups_ok && p.M % 256 == 0)
      return (persist & 1) ? launch_igemm_dma_persist<128, 256, 8>(p, nch_pad, phases, st, slots)
                           : launch_igemm_dma<128, 256, 3, true, 8>(p, nch_pad, phases, st, slots);
    // 128 x 128 tiles on the persistent kernel (4 consumer + 4 loader waves, one workgroup per CU) when that gives about one tile
    // per CU and 128 x 256 tiles would leave half the chip idle: batch-256 layers with 8x8 outputs (round 5; JCK_IGEMM_128 = minimum
    // number of such tiles, 0 disables)
    if (g_igemm_128 > 0 && wgs >= g_igemm_128 && (persist & 1) && !p.act_row_elems && (!p.stats || p.logOHW >= 4) && p.M % 128 == 0)
      return launch_igemm_dma_persist<128, 128, 4>(p, nch_pad, phases, st, slots);
    if (wgs >= 512) return launch_igemm_dma<128, 128, 2>(p, nch_pad, phases, st, slots);
    if (ws_mode && (persist & 2) && !p.act_row_elems) return launch_igemm_dma_persist<128, 64, 4>(p, nch_pad, phases, st, slots);
    if (ws_mode) return launch_igemm_dma<128, 64, 3, true>(p, nch_pad, phases, st, slots);
    return launch_igemm_dma<128, 64, 2>(p, nch_pad, phases, st, slots);
  }
  if (nch_pad % 128 == 0) {
    if (nsub != 1) JCK_FAIL(JCK_E_ARG, "igemm: 4-channel gather with >=128 output rows unsupported");
    // keep >= ~256 workgroups in flight: halve the pixel tile for small pixel counts
    const long long wgs = (long long)cdiv(p.M, 128) * (nch_pad / 128) * phases;
    if (wgs >= 256) return launch_igemm_t<P, 128, 128, 1>(p, nch_pad, phases, st, slots);
    return launch_igemm_t<P, 128, 64, 1>(p, nch_pad, phases, st, slots);
  }
  if (nch_pad == 64) {
    if (use_dma && !P::IS_F32 && nsub == 1 && p.ksplit <= 1 && !p.rows_are_phases) {
      const long long t128 = (long long)cdiv(p.M, 128) * phases;
      if ((g_igemm_persist & 4) && !p.act_row_elems && !p.bias && !p.epi && !p.out_f32 && !p.rows_are_phases && !p.out_split_stride)
        return launch_igemm_dma_persist<64, 128, 4>(p, nch_pad, phases, st, slots);
      return launch_igemm_dma<64, 128, 2>(p, nch_pad, phases, st, slots);
    }
    if (nsub == 2) return launch_igemm_t<P, 64, 128, 2>(p, nch_pad, phases, st, slots);
    return launch_igemm_t<P, 64, 128, 1>(p, nch_pad, phases, st, slots);
  }
  if (nch_pad == 16) {
    if (nsub != 1) JCK_FAIL(JCK_E_ARG, "igemm: 4->4 channel product unsupported");
    return launch_igemm_t<P, 16, 256, 1>(p, nch_pad, phases, st, slots);
  }
  JCK_FAIL(JCK_E_ARG, "igemm: unsupported padded row count " + std::to_string(nch_pad));
}

static int launch_wgrad_reduce(const float* ws, int Z, int CsRows, int ncols, int Cs, int Cb, int logCbPad, float* grad,
                               int accumulate, hipStream_t st);
int launch_igemm(int prec, const IgemmParams& p0, int nch_pad, int phases, int nsub, hipStream_t st, int* slots) {
  IgemmParams p = p0;
  for (int zz = 0; zz < 4; ++zz)
    for (int t = 0; t < 16; ++t) p.tap[zz][t] = ((int)p.dy[zz][t] << 16) | ((int)p.dx[zz][t] & 0xffff);
  const long long esz = prec == JCK_PREC_F32 ? 4 : 2;
  if (nsub == 1 && p.logC < 6 && !p.act_row_elems) JCK_FAIL(JCK_E_ARG, "igemm: the gathered tensor needs >= 64 channels (or exactly 4)");
  {
    // extent of the gathered tensor: rows (n, oy, ox) span N = M / (OH*OW) images of H x W x C
    const long long nimg = ((long long)p.M + (1ll << p.logOHW) - 1) >> p.logOHW;
    const long long ab = p.act_row_elems ? (long long)p.M * p.act_row_elems * esz : nimg * p.H * p.W * (1ll << p.logC) * esz;
    const long long wb = (long long)(p.ksplit > 1 ? 1 : phases) * (p.w_phase_stride ? p.w_phase_stride : (long long)nch_pad * p.K) * esz;
    if (ab >= (1ll << 31) || wb >= (1ll << 31)) JCK_FAIL(JCK_E_ARG, "igemm: operand exceeds 2 GiB (32-bit buffer offsets)");
    p.act_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
  }
  if (p.K % IG_BK != 0) JCK_FAIL(JCK_E_ARG, "igemm: K must be a multiple of 64, got " + std::to_string(p.K));
  if (p.M <= 0) JCK_FAIL(JCK_E_ARG, "igemm: empty problem");
  if (p.stats && !slots) JCK_FAIL(JCK_E_ARG, "igemm: stats requested without a slot-count output");
  if (prec == JCK_PREC_BF16) return launch_igemm_p<PrecBf16>(p, nch_pad, phases, nsub, st, slots);
  if (prec == JCK_PREC_F32) return launch_igemm_p<PrecF32>(p, nch_pad, phases, nsub, st, slots);
  JCK_FAIL(JCK_E_ARG, "bad prec");
}

// rows of partial statistics a launch may write: one per (tile, wave) = at most one per 32 pixels, or - the persistent kernels'
// accumulated BatchNorm-backward rows - [<= 4 groups][<= 512 workgroups / channel tiles][<= 4 pixel waves] <= 4096
extern "C" size_t jck_stats_floats(long long pixels, int C, int nyrep) {
  return (size_t)std::max<long long>(pixels / 32 + 16, 4096) * (size_t)std::max(1, nyrep) * 2 * (size_t)C;
}
extern "C" size_t jck_packed_bytes(int prec, long long elems) { return (size_t)elems * (prec == JCK_PREC_F32 ? 4 : 2); }

// image-side layers on the streaming kernels of thin.hpp (bf16, 64 channels on the wide side, row length % 16 == 0)
#define g_use_thin g_thin
#define IMG_GPW 8
static int launch_img_down(const void* x, const void* w, void* out, float* stats, int* slots, int N, int Hb, int Wb, double flops,
                           hipStream_t st) {
  ImgDownParams q = {};
  const int OH = Hb / 2, OW = Wb / 2;
  q.x = x; q.w = w; q.out = out; q.stats = stats;
  q.ngroups = N * OH * (OW / 16); q.H = Hb; q.W = Wb; q.logOH = ilog2(OH); q.logG = ilog2(OW / 16);
  q.x_bytes = (unsigned)((long long)N * Hb * Wb * 4 * 2);
  const int grid = cdiv(q.ngroups, 4 * IMG_GPW);       // 8 groups per wave: 2 / 4 / 16 measured 20.0 / 14.7 / 13.2 us against 13.6
  if (stats) {
    if (!slots) JCK_FAIL(JCK_E_ARG, "conv_down: stats requested without a slot-count output");
    *slots = grid;
  }
  ProfScope prof(18, flops, st);
  hipLaunchKernelGGL(img_down_kernel<IMG_GPW>, dim3(grid), dim3(256), 0, st, q);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
#define IMG_UP_R 8
static int launch_img_up(const void* a, const void* w, void* out, int epi_tanh, int N, int Hs, int Ws, double flops, hipStream_t st,
                         const void* mul_t = nullptr, float mul_scale = 1.f, hipEvent_t done = nullptr) {
  ImgUpParams q = {};
  q.a = a; q.w = w; q.out = out; q.epi_tanh = epi_tanh; q.mul_t = mul_t; q.mul_scale = mul_scale;
  q.nunits = N * (Hs / IMG_UP_R) * (Ws / 16); q.Hs = Hs; q.Ws = Ws; q.logYB = ilog2(Hs / IMG_UP_R); q.logG = ilog2(Ws / 16);
  q.a_bytes = (unsigned)((long long)N * Hs * Ws * 64 * 2);
  ProfScope prof(19, flops, st);
  LAUNCH_EV(img_up_kernel<IMG_UP_R>, dim3(cdiv(q.nunits, 4)), dim3(256), 0, st, done, q);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

static int conv_down_impl(int prec, const void* big, const void* w, void* small_out, float* stats, int* stats_slots,
                          int N, int Hb, int Wb, int Cb, int Cs, void* stream, int fwd_group_images = 0) {
  const int cbp = jck_pad_chan(Cb);
  if (!is_pow2(cbp) || !is_pow2(Hb) || !is_pow2(Wb) || Hb < 2 || Wb < 2 || Cs % 4 != 0)
    JCK_FAIL(JCK_E_ARG, "conv_down: shapes must be powers of two (Hb,Wb,Cb) and Cs % 4 == 0");
  if ((long long)N * Hb * Wb * cbp >= (1ll << 31)) JCK_FAIL(JCK_E_ARG, "conv_down: tensor exceeds 2^31 elements");
  IgemmParams p = {};
  p.act = big; p.w = w; p.out = small_out; p.stats = stats;
  const int OH = Hb / 2, OW = Wb / 2;
  p.M = N * OH * OW; p.NchStore = Cs; p.logC = ilog2(cbp); p.K = 16 << p.logC;
  p.H = Hb; p.W = Wb; p.logOW = ilog2(OW); p.logOHW = ilog2(OH * OW); p.sy = p.sx = 2; p.ntaps = 16;
  for (int t = 0; t < 16; ++t) { p.dy[0][t] = (signed char)(t / 4 - 1); p.dx[0][t] = (signed char)(t % 4 - 1); }
  p.osN = (long long)OH * OW * Cs; p.osY = OW * Cs; p.osX = Cs; p.obase[0] = 0;
  p.cstat = Cs; p.ytiles_per_cset = 1; p.epi = 0; p.w_phase_stride = 0;
  if (stats && !is_pow2(Cs)) JCK_FAIL(JCK_E_ARG, "conv_down: BN statistics need a power-of-two channel count");
  p.flops = 2.0 * p.M * Cs * 16.0 * Cb;
  if (stats && fwd_group_images > 0) { p.bn_group_rows = fwd_group_images * OH * OW; p.stat_accum = 1; }
  if (g_use_thin && prec == JCK_PREC_BF16 && cbp == 4 && Cs == 64 && OW % 16 == 0 && is_pow2(OH))
    return launch_img_down(big, w, small_out, stats, stats_slots, N, Hb, Wb, p.flops, (hipStream_t)stream);
  return launch_igemm(prec, p, jck_pad_rows(Cs), 1, cbp == 4 ? 2 : 1, (hipStream_t)stream, stats_slots);
}
extern "C" int jck_conv_down(int prec, const void* big, const void* w, void* small_out, float* stats, int* stats_slots,
                             int N, int Hb, int Wb, int Cb, int Cs, void* stream) {
  return conv_down_impl(prec, big, w, small_out, stats, stats_slots, N, Hb, Wb, Cb, Cs, stream);
}
// Forward statistics per BatchNorm group of `group_images` images (N % group_images == 0): *stats_slots rows, the first
// *stats_slots / (N / group_images) of them belong to group 0, and so on - the layout jck_bn_finalize_grouped reads.  Large
// launches on the persistent kernels write one row per (workgroup, group) instead of one per (tile, wave).
extern "C" int jck_conv_down_grouped(int prec, const void* big, const void* w, void* small_out, float* stats, int* stats_slots,
                                     int N, int Hb, int Wb, int Cb, int Cs, int group_images, void* stream) {
  if (group_images < 1 || N % group_images) JCK_FAIL(JCK_E_ARG, "conv_down_grouped: N must be a multiple of group_images >= 1");
  return conv_down_impl(prec, big, w, small_out, stats, stats_slots, N, Hb, Wb, Cb, Cs, stream, group_images);
}
static int conv_up_impl(int prec, const void* small_in, const void* w, void* big_out, float* stats, int* stats_slots,
                        int epi_tanh, int N, int Hs, int Ws, int Cs, int Cb, void* stream, int fwd_group_images = 0) {
  const int cbp = jck_pad_chan(Cb);
  if (!is_pow2(Cs) || Cs < 16 || !is_pow2(Hs) || !is_pow2(Ws) || cbp % 4 != 0)
    JCK_FAIL(JCK_E_ARG, "conv_up: shapes must be powers of two (Hs,Ws,Cs>=16)");
  if ((long long)N * Hs * Ws * 4 * cbp >= (1ll << 31)) JCK_FAIL(JCK_E_ARG, "conv_up: tensor exceeds 2^31 elements");
  IgemmParams p = {};
  p.act = small_in; p.w = w; p.out = big_out; p.stats = stats;
  p.M = N * Hs * Ws; p.NchStore = cbp; p.logC = ilog2(Cs); p.K = 4 << p.logC;
  p.H = Hs; p.W = Ws; p.logOW = ilog2(Ws); p.logOHW = ilog2(Hs * Ws); p.sy = p.sx = 1; p.ntaps = 4;
  if (cbp == 4) {
    // 3/4-channel output: one launch, the four output parities are the 16 MFMA rows, 9 input offsets as taps; every
    // workgroup then writes whole contiguous output rows instead of interleaved 8-byte pixels
    if (Cs % 64) JCK_FAIL(JCK_E_ARG, "conv_up: Cs % 64 != 0 for a <=4-channel output");
    if (stats) JCK_FAIL(JCK_E_ARG, "conv_up: statistics are not provided for <=4-channel outputs");
    p.ntaps = 9; p.K = 9 * Cs; p.NchStore = 16; p.rows_are_phases = 1;
    for (int t = 0; t < 9; ++t) { p.dy[0][t] = (signed char)(t / 3 - 1); p.dx[0][t] = (signed char)(t % 3 - 1); }
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) p.obase[ph * 2 + pw] = (ph * 2 * Ws + pw) * cbp;
    p.osN = (long long)4 * Hs * Ws * cbp; p.osY = 2 * 2 * Ws * cbp; p.osX = 2 * cbp;
    p.cstat = 4; p.ytiles_per_cset = 1; p.epi = epi_tanh ? 1 : 0; p.w_phase_stride = 0;
    p.flops = 2.0 * p.M * 4.0 * Cb * 4.0 * Cs;
    if (g_use_thin && prec == JCK_PREC_BF16 && Cs == 64 && Ws % 16 == 0 && Hs % IMG_UP_R == 0)
      return launch_img_up(small_in, w, big_out, epi_tanh ? 1 : 0, N, Hs, Ws, p.flops, (hipStream_t)stream);
    return launch_igemm(prec, p, 16, 1, 1, (hipStream_t)stream, nullptr);
  }
  static const int DI[2][2] = {{0, -1}, {1, 0}};          // input offset of tap th for output parity ph
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      const int z = ph * 2 + pw;
      for (int t = 0; t < 4; ++t) { p.dy[z][t] = (signed char)DI[ph][t >> 1]; p.dx[z][t] = (signed char)DI[pw][t & 1]; }
      p.obase[z] = (ph * 2 * Ws + pw) * cbp;
    }
  p.osN = (long long)4 * Hs * Ws * cbp; p.osY = 2 * 2 * Ws * cbp; p.osX = 2 * cbp;
  p.cstat = cbp; p.ytiles_per_cset = 1; p.epi = epi_tanh ? 1 : 0;
  if (stats && !is_pow2(cbp)) JCK_FAIL(JCK_E_ARG, "conv_up: BN statistics need a power-of-two channel count");
  const int rows = jck_pad_rows(Cb);
  p.w_phase_stride = (long long)rows * p.K;
  if (p.K % 64 != 0) JCK_FAIL(JCK_E_ARG, "conv_up: 4*Cs must be a multiple of 64");
  p.flops = 2.0 * p.M * 4.0 * Cb * 4.0 * Cs;
  if (stats && fwd_group_images > 0) { p.bn_group_rows = fwd_group_images * Hs * Ws; p.stat_accum = 1; }
  return launch_igemm(prec, p, rows, 4, 1, (hipStream_t)stream, stats_slots);
}
// D.conv1's input gradient with the tanh + instance-noise-mix backward of G's output in its epilogue (thin.hpp: ImgUpParams::mul_t):
// out = scale * bf16(convT(small_in)) * (1 - tanh_y^2), bit for bit jck_conv_up followed by tanh_bwd_ev.  *fused = false (and
// nothing launched) when the layer does not run on the image-side streaming kernel: the caller then issues the two launches.
int conv_up_tanh_bwd_ev(int prec, const void* small_in, const void* w, const void* tanh_y, float scale, void* out, int N, int Hs, int Ws,
                        int Cs, int Cb, hipStream_t stream, hipEvent_t done, bool* fused) {
  *fused = g_use_thin && prec == JCK_PREC_BF16 && jck_pad_chan(Cb) == 4 && Cs == 64 && Ws % 16 == 0 && Hs % IMG_UP_R == 0 &&
           is_pow2(Hs) && is_pow2(Ws) && (long long)N * Hs * Ws * 16 < (1ll << 31);
  if (!*fused) return JCK_OK;
  return launch_img_up(small_in, w, out, 0, N, Hs, Ws, 2.0 * N * Hs * Ws * 4.0 * Cb * 4.0 * Cs, stream, tanh_y, scale, done);
}
extern "C" int jck_conv_up(int prec, const void* small_in, const void* w, void* big_out, float* stats, int* stats_slots,
                           int epi_tanh, int N, int Hs, int Ws, int Cs, int Cb, void* stream) {
  return conv_up_impl(prec, small_in, w, big_out, stats, stats_slots, epi_tanh, N, Hs, Ws, Cs, Cb, stream);
}
extern "C" int jck_conv_up_grouped(int prec, const void* small_in, const void* w, void* big_out, float* stats, int* stats_slots,
                                   int N, int Hs, int Ws, int Cs, int Cb, int group_images, void* stream) {
  if (group_images < 1 || N % group_images) JCK_FAIL(JCK_E_ARG, "conv_up_grouped: N must be a multiple of group_images >= 1");
  return conv_up_impl(prec, small_in, w, big_out, stats, stats_slots, 0, N, Hs, Ws, Cs, Cb, stream, group_images);
}
static int g1_fwd_impl(int prec, const void* z, const void* w, void* out, float* stats, int* stats_slots, int B,
                       int CiPad, int Co, void* stream) {
  if (!is_pow2(CiPad) || CiPad < 64 || !is_pow2(Co) || (16 * Co) % 128 != 0)
    JCK_FAIL(JCK_E_ARG, "g1_fwd: CiPad must be a power of two >= 64, Co a power of two");
  IgemmParams p = {};
  p.act = z; p.w = w; p.out = out; p.stats = stats;
  p.M = B; p.NchStore = 16 * Co; p.logC = ilog2(CiPad); p.K = CiPad;
  p.H = 1; p.W = 1; p.logOW = 0; p.logOHW = 0; p.sy = p.sx = 1; p.ntaps = 1;
  p.dy[0][0] = 0; p.dx[0][0] = 0;
  p.osN = (long long)16 * Co; p.osY = 0; p.osX = 0; p.obase[0] = 0;
  p.cstat = Co; p.ytiles_per_cset = 1; p.epi = 0; p.w_phase_stride = 0;
  if (Co < 128) JCK_FAIL(JCK_E_ARG, "g1_fwd: Co must be >= 128");
  p.flops = 2.0 * B * 16.0 * Co * CiPad;
  return launch_igemm(prec, p, 16 * Co, 1, 1, (hipStream_t)stream, stats_slots);
}
extern "C" int jck_g1_fwd(int prec, const void* z, const void* w, void* out, float* stats, int* stats_slots, int B,
                          int CiPad, int Co, void* stream) {
  return g1_fwd_impl(prec, z, w, out, stats, stats_slots, B, CiPad, Co, stream);
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------
struct WgradPlan { int BG, BS, gx, gy, Z, mchunk, CsRows, ncols; size_t ws; };

// wide: 256-column gathered tile of the LDS-DMA kernel (bf16, Cs % 128 == 0, ncols % 256 == 0); JCK_WGRAD_GT=1 disables
static bool wgrad_wide_shape(int ncols, int Cs) { return g_wgrad_gt == 2 && ncols % 256 == 0 && Cs % 128 == 0; }
static WgradPlan plan_wgrad(long long Mtot, int ncols, int Cs, bool wide = false) {
  WgradPlan pl;
  pl.ncols = ncols;
  pl.BG = wide ? 256 : (ncols % 128 == 0) ? 128 : 64;
  pl.BS = (Cs >= 128) ? 128 : 64;
  pl.gx = cdiv(ncols, pl.BG);
  pl.gy = cdiv(Cs, pl.BS);
  pl.CsRows = pl.gy * pl.BS;
  const int tiles = pl.gx * pl.gy;
  const int target = g_wgrad_wgs, target_small = g_wgrad_small_wgs;
  long long Z = std::max(1, (tiles >= 4 ? target : target_small) / tiles);
  const long long maxZ = std::max(1ll, Mtot / (WG_BKP * 4));
  Z = std::min(Z, maxZ);
  long long mchunk = (Mtot + Z - 1) / Z;
  mchunk = (mchunk + 63) / 64 * 64;            // multiple of both k-step sizes (32 register-staged, 64 LDS-DMA)
  Z = (Mtot + mchunk - 1) / mchunk;
  pl.Z = (int)Z; pl.mchunk = (int)mchunk;
  pl.ws = (size_t)Z * pl.CsRows * ncols * sizeof(float);
  return pl;
}

template <class P, int BG, int BS, int NSUB>
static int launch_wgrad_t(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
  constexpr int variant = 10 + (P::IS_F32 ? 4 : 0) + (BG == 128 ? (BS == 128 ? 0 : 1) : (NSUB == 2 ? 2 : 3));
  ProfScope prof(variant, p.flops, st);
  constexpr int LDSB = WgradCfg<P, BG, BS>::LDS_BYTES;
  auto kern = wgrad_kernel<P, BG, BS, NSUB>;
  static bool attr_done = false;
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(pl.gx, pl.gy, pl.Z), dim3(256), LDSB, st, p);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// LDS-DMA weight gradient: wave-specialised (4 loader + 4 consumer waves, 3 stages = 96 KB) by default; JCK_WGRAD_WS=0 selects
// the 4-wave, 2-stage form (48.4 vs 33.7 us at B=256 on the isolated product; 3-4 stages or 8 symmetric waves: within 7 %).
// JCK_WGRAD_STAMP=1 (development) launches the instrumented twin read back by jck_debug_wgrad_stamps.
template <int NSTG, bool STAMP, bool WS, int GT = 1, bool PIPE = false, int WDBG = 0>
static int launch_wgrad_dma_t(const WgradParams& q, int grid, hipStream_t st) {
  constexpr int LDSB = NSTG * (GT + 1) * WGD_BKP * 256;
  static bool attr_done = false;
  if (!attr_done) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma_kernel<NSTG, 4, STAMP, WS, GT, PIPE, WDBG>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    attr_done = true;
  }
  hipLaunchKernelGGL((wgrad_dma_kernel<NSTG, 4, STAMP, WS, GT, PIPE, WDBG>), dim3(grid), dim3(WS ? (4 + 4 * GT) * 64 : 256), LDSB, st, q);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
static int launch_wgrad_dma(const WgradParams& p, const WgradPlan& pl, hipStream_t st) {
  ProfScope prof(pl.BG == 256 ? 21 : 10, p.flops, st);
  WgradParams q = p;
  q.gx = pl.gx; q.gy = pl.gy; q.gz = pl.Z;
  q.loader_prio = g_wgrad_prio;
  const int grid = pl.gx * pl.gy * pl.Z;
  const int stamp = g_wgrad_stamp, wsp = g_wgrad_ws;
  if (pl.BG == 256) return launch_wgrad_dma_t<3, false, true, 2>(q, grid, st);
#ifdef JCK_DIAG
  if (g_wgrad_dbg == 1) return launch_wgrad_dma_t<3, false, true, 1, true, 1>(q, grid, st);
  if (g_wgrad_dbg == 2) return launch_wgrad_dma_t<3, false, true, 1, true, 2>(q, grid, st);
  if (g_wgrad_dbg == 3) return launch_wgrad_dma_t<3, false, true, 1, true, 3>(q, grid, st);
  if (g_wgrad_dbg == 4) return launch_wgrad_dma_t<3, false, true, 1, true, 4>(q, grid, st);
#endif
  if (wsp && !stamp && g_wgrad_pipe) return launch_wgrad_dma_t<3, false, true, 1, true>(q, grid, st);
  if (wsp) return stamp ? launch_wgrad_dma_t<3, true, true>(q, grid, st) : launch_wgrad_dma_t<3, false, true>(q, grid, st);
  return stamp ? launch_wgrad_dma_t<2, true, false>(q, grid, st) : launch_wgrad_dma_t<2, false, false>(q, grid, st);
}

template <class P>
static int launch_wgrad_p(const WgradParams& p, const WgradPlan& pl, int nsub, hipStream_t st) {
  const int use_dma = g_wgrad_dma;
  if (use_dma && p.big_bytes && p.s_bytes && !P::IS_F32 && (pl.BG == 128 || pl.BG == 256) && pl.BS == 128 && nsub == 1 && !p.big_row_elems && p.logCb >= 6 && p.logCb < 30 &&
      pl.mchunk % WGD_BKP == 0 && p.logOW <= 6 &&
      ((1 << p.logOHW) <= WGD_BKP || p.H == p.sy * ((1 << p.logOHW) >> p.logOW)))     // constant 64-pixel address step (wgrad.hpp)
    return launch_wgrad_dma(p, pl, st);
  if (pl.BG == 256) JCK_FAIL(JCK_E_ARG, "wgrad: the 256-column plan is for the LDS-DMA kernel only");
  if (pl.BG == 128 && pl.BS == 128 && nsub == 1) return launch_wgrad_t<P, 128, 128, 1>(p, pl, st);
  if (pl.BG == 128 && pl.BS == 64 && nsub == 1) return launch_wgrad_t<P, 128, 64, 1>(p, pl, st);
  if (pl.BG == 64 && pl.BS == 64 && nsub == 2) return launch_wgrad_t<P, 64, 64, 2>(p, pl, st);
  if (pl.BG == 64 && pl.BS == 64 && nsub == 1) return launch_wgrad_t<P, 64, 64, 1>(p, pl, st);
  JCK_FAIL(JCK_E_ARG, "wgrad: unsupported tile plan");
}

static int run_wgrad(int prec, WgradParams& p, const WgradPlan& pl, int nsub, float* ws, size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < pl.ws) JCK_FAIL(JCK_E_WS, "wgrad: workspace too small: need " + std::to_string(pl.ws));
  p.part = ws; p.CsRows = pl.CsRows; p.ncols = pl.ncols; p.mchunk = pl.mchunk;
  if (prec == JCK_PREC_BF16) return launch_wgrad_p<PrecBf16>(p, pl, nsub, st);
  if (prec == JCK_PREC_F32) return launch_wgrad_p<PrecF32>(p, pl, nsub, st);
  JCK_FAIL(JCK_E_ARG, "bad prec");
}

static int launch_wgrad_reduce(const float* ws, int Z, int CsRows, int ncols, int Cs, int Cb, int logCbPad, float* grad,
                               int accumulate, hipStream_t st) {
  if (Cb % 64 == 0 && (1 << logCbPad) == Cb) {
    // few workgroups and many slabs (the tap-reuse plan): four slab groups per workgroup
    if (Z >= 16 && (long long)(Cb / 64) * Cs <= 1024)
      hipLaunchKernelGGL(wgrad_reduce16_kernel<4>, dim3(Cb / 64, Cs), dim3(1024), 0, st, ws, Z, CsRows, ncols, Cb, logCbPad, grad, accumulate);
    else
      hipLaunchKernelGGL(wgrad_reduce16_kernel<1>, dim3(Cb / 64, Cs), dim3(256), 0, st, ws, Z, CsRows, ncols, Cb, logCbPad, grad, accumulate);
  } else if (logCbPad == 2 && ncols == 64) {
    hipLaunchKernelGGL(wgrad_reduce_img_kernel, dim3(Cs), dim3(256), 0, st, ws, Z, CsRows, Cb, grad, accumulate);
  } else {
    const long long total = (long long)Cs * Cb * 16;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, st, ws,
                       Z, CsRows, ncols, Cs, Cb, logCbPad, 16, grad, accumulate);
  }
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// the 256-column plan applies when the launch takes the LDS-DMA kernel (bf16, >= 64 gathered channels, 128-row S tiles)
static bool conv_wgrad_wide(int prec, int cbp, int Cs) {
  return g_wgrad_dma && prec == JCK_PREC_BF16 && cbp >= 64 && wgrad_wide_shape(16 * cbp, Cs);
}
extern "C" size_t jck_conv_wgrad_ws_bytes(int N, int Hb, int Wb, int Cb, int Cs) {
  const long long M = (long long)N * (Hb / 2) * (Wb / 2);
  const int cbp = jck_pad_chan(Cb);
  // the larger of the two plans, whatever the knobs say right now (a workspace outlives a jck_tune call)
  const bool wide_shape = cbp >= 64 && (16 * cbp) % 256 == 0 && Cs % 128 == 0;
  return std::max(plan_wgrad(M, 16 * cbp, Cs).ws, wide_shape ? plan_wgrad(M, 16 * cbp, Cs, true).ws : (size_t)0);
}

extern "C" int jck_conv_wgrad(int prec, const void* small_side, const void* big_side, float* ws, size_t ws_bytes,
                              float* grad, int accumulate, int N, int Hb, int Wb, int Cb, int Cs, void* stream) {
  const int cbp = jck_pad_chan(Cb);
  if (!is_pow2(cbp) || !is_pow2(Hb) || !is_pow2(Wb) || Cs % 8 != 0) JCK_FAIL(JCK_E_ARG, "conv_wgrad: bad shape");
  const int OH = Hb / 2, OW = Wb / 2;
  WgradParams p = {};
  p.sside = small_side; p.big = big_side; p.Mtot = N * OH * OW; p.CsStride = Cs; p.logCb = ilog2(cbp);
  p.H = Hb; p.W = Wb; p.logOW = ilog2(OW); p.logOHW = ilog2(OH * OW); p.sy = p.sx = 2; p.ntaps = 16;
  for (int t = 0; t < 16; ++t) { p.dy[t] = (signed char)(t / 4 - 1); p.dx[t] = (signed char)(t % 4 - 1); }
  p.flops = 2.0 * p.Mtot * Cs * 16.0 * Cb;
  {   // operand sizes for the buffer descriptors of the LDS-DMA kernels (32-bit byte offsets: < 2 GiB each)
    const long long esz = prec == JCK_PREC_F32 ? 4 : 2;
    const long long bb = (long long)N * Hb * Wb * cbp * esz, sbytes = (long long)p.Mtot * Cs * esz;
    if (bb < (1ll << 31) && sbytes < (1ll << 31)) { p.big_bytes = (unsigned)bb; p.s_bytes = (unsigned)sbytes; }
  }
  const WgradPlan pl = plan_wgrad(p.Mtot, 16 * cbp, Cs, conv_wgrad_wide(prec, cbp, Cs));
  JCK_TRY(run_wgrad(prec, p, pl, cbp == 4 ? 2 : 1, ws, ws_bytes, (hipStream_t)stream));
  JCK_TRY(launch_wgrad_reduce(ws, pl.Z, pl.CsRows, pl.ncols, Cs, Cb, p.logCb, grad, accumulate, (hipStream_t)stream));
  return JCK_OK;
}

extern "C" size_t jck_g1_wgrad_ws_bytes(int B, int CiPad, int Co) { return plan_wgrad(B, 16 * Co, CiPad).ws; }

extern "C" int jck_g1_wgrad(int prec, const void* z, const void* dy, float* ws, size_t ws_bytes, float* grad,
                            int accumulate, int B, int Ci, int CiPad, int Co, void* stream) {
  if (!is_pow2(Co) || CiPad % 64 != 0) JCK_FAIL(JCK_E_ARG, "g1_wgrad: bad shape");
  WgradParams p = {};
  p.sside = z; p.big = dy; p.Mtot = B; p.CsStride = CiPad; p.logCb = ilog2(Co);
  p.H = 4; p.W = 4; p.logOW = 0; p.logOHW = 0; p.sy = p.sx = 1; p.ntaps = 16;
  for (int t = 0; t < 16; ++t) { p.dy[t] = (signed char)(t / 4); p.dx[t] = (signed char)(t % 4); }
  const WgradPlan pl = plan_wgrad(B, 16 * Co, CiPad);
  p.flops = 2.0 * B * Ci * 16.0 * Co;
  {
    const long long esz = prec == JCK_PREC_F32 ? 4 : 2;
    p.big_bytes = (unsigned)((long long)B * 16 * Co * esz); p.s_bytes = (unsigned)((long long)B * CiPad * esz);
  }
  int rc = run_wgrad(prec, p, pl, 1, ws, ws_bytes, (hipStream_t)stream);
  if (rc) return rc;
  JCK_TRY(launch_wgrad_reduce(ws, pl.Z, pl.CsRows, pl.ncols, Ci, Co, p.logCb, grad, accumulate, (hipStream_t)stream));
  return JCK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------------------
static unsigned ew_grid(long long n, int per_block = 256) { return (unsigned)std::max<long long>(1, std::min<long long>((n + per_block - 1) / per_block, 8192)); }

extern "C" int jck_pack_down(int prec, const float* w, int Cs, int Cb, void* wp, void* stream) {
  const int cbp = jck_pad_chan(Cb), rows = jck_pad_rows(Cs);
  if (!is_pow2(cbp)) JCK_FAIL(JCK_E_ARG, "pack_down: Cb must be 3 or a power of two");
  const long long total = (long long)rows * 16 * cbp;
  DISPATCH_T(prec, hipLaunchKernelGGL(pack_down_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, Cs, Cb,
                                      rows, ilog2(cbp), (T*)wp));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_pack_up(int prec, const float* w, int Cs, int Cb, void* wp, void* stream) {
  if (Cb <= 4) {     // thin outputs: four parities as rows of one operand (see jck_conv_up)
    if (Cs % 64) JCK_FAIL(JCK_E_ARG, "pack_up: Cs % 64 != 0 for a <=4-channel output");
    const long long total16 = 16ll * 9 * Cs;
    DISPATCH_T(prec, hipLaunchKernelGGL(pack_up16_kernel<T>, dim3(ew_grid(total16)), dim3(256), 0, (hipStream_t)stream, w, Cs, Cb,
                                        (T*)wp));
    HIPCHK(hipGetLastError());
    return JCK_OK;
  }
  const int rows = jck_pad_rows(Cb);
  const long long total = 4ll * rows * 4 * Cs;
  DISPATCH_T(prec, hipLaunchKernelGGL(pack_up_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, Cs, Cb, rows,
                                      (T*)wp));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_pack_g1(int prec, const float* w, int Ci, int Co, int CiPad, void* wp, void* stream) {
  const long long total = 16ll * Co * CiPad;
  DISPATCH_T(prec, hipLaunchKernelGGL(pack_g1_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, Ci, Co, CiPad,
                                      (T*)wp));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_pack_head(const float* w, int C, float* wp, void* stream) {
  hipLaunchKernelGGL(pack_head_kernel, dim3(cdiv(16 * C, 256)), dim3(256), 0, (hipStream_t)stream, w, C, wp);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm
// ---------------------------------------------------------------------------------------------------------
extern "C" int jck_bn_finalize(const float* stats, int slots, float count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* nbt, float momentum, float eps, float* aux,
                               int C, void* stream) {
  if (slots < 1) JCK_FAIL(JCK_E_ARG, "bn_finalize: slots must be >= 1");
  if (C % 4) JCK_FAIL(JCK_E_ARG, "bn_finalize: C % 4 != 0");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 4), dim3(256), 0, (hipStream_t)stream, stats, slots, count, gamma, beta,
                     running_mean, running_var, (long long*)nbt, momentum, eps, aux, C);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}


extern "C" int jck_bn_act_fwd(int prec, const void* y, const float* aux, float slope, void* a, long long rows, int C,
                              void* stream) {
  if (!is_pow2(C) || C < 8) JCK_FAIL(JCK_E_ARG, "bn_act_fwd: C must be a power of two >= 8");
  const long long total8 = rows * C / 8;
  ProfScope prof(PROF_BN_ACT_FWD, 0.0, (hipStream_t)stream, 2.0 * rows * C * (prec == JCK_PREC_F32 ? 4 : 2));
  DISPATCH_T(prec, hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                      (const T*)y, aux, slope, (T*)a, total8, C));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// finalize + apply in one launch (ew.hpp: bn_fwd_fused_kernel).  Taken while the statistics rows are few (every workgroup sums the rows of
// its channel slice itself); *fused = false and nothing launched otherwise - the caller then issues jck_bn_finalize* + jck_bn_act_fwd*.
int bn_fwd_fused(int prec, const void* y, const float* stats, int slots_per_group, float count, const float* gamma, const float* beta,
                 float eps, float slope, void* a, float* aux, float* stat_out, float* running_mean, float* running_var, int64_t* nbt,
                 float momentum, long long rows_per_group, int C, int groups, long long out_row, long long out_pitch, hipStream_t stream,
                 bool* fused) {
  *fused = g_bn_fuse && C >= 64 && C % 64 == 0 && is_pow2(C) && slots_per_group >= 1 && slots_per_group <= g_bn_fuse_rows && groups >= 1 &&
           (!out_pitch || (out_row >= 8 && !(out_row & (out_row - 1)) && out_pitch >= out_row && out_pitch % 8 == 0));
  if (!*fused) return JCK_OK;
  // ~512 workgroups: each re-reads slots x 512 bytes of rows (L2-resident), so more of them cost more than they hide
  const int nsl = C / 64;
  const long long per = std::max<long long>(1, g_bn_fuse_wgs / ((long long)nsl * groups));
  const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((rows_per_group + 31) / 32, per));
  ProfScope prof(PROF_BN_ACT_FWD, 0.0, stream, 2.0 * groups * rows_per_group * C * (prec == JCK_PREC_F32 ? 4 : 2));
  DISPATCH_T(prec, hipLaunchKernelGGL(bn_fwd_fused_kernel<T>, dim3(gx, nsl, groups), dim3(BNF_THREADS), 0, stream, (const T*)y, stats,
                                      slots_per_group, count, gamma, beta, eps, slope, (T*)a, aux, stat_out, running_mean, running_var,
                                      (long long*)nbt, momentum, rows_per_group, C, out_pitch ? ilog2((int)out_row) : 0, out_pitch));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// C-ABI form: falls back to the two launches itself where the fused form does not apply
extern "C" int jck_bn_fwd(int prec, const void* y, const float* stats, int slots_per_group, float count, const float* gamma, const float* beta,
                          float eps, float slope, void* a, float* aux, float* stat_out, float* running_mean, float* running_var, int64_t* nbt,
                          float momentum, long long rows_per_group, int C, int groups, void* stream) {
  if ((running_mean || nbt) && groups != 1) JCK_FAIL(JCK_E_ARG, "bn_fwd: running statistics are updated in place for ONE group only");
  bool fused = false;
  JCK_TRY(bn_fwd_fused(prec, y, stats, slots_per_group, count, gamma, beta, eps, slope, a, aux, stat_out, running_mean, running_var, nbt, momentum,
                       rows_per_group, C, groups, 0, 0, (hipStream_t)stream, &fused));
  if (fused) return JCK_OK;
  if (slots_per_group < 1 || groups < 1 || C % 4) JCK_FAIL(JCK_E_ARG, "bn_fwd: slots and groups must be >= 1, C % 4 == 0");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 4, groups), dim3(256), 0, (hipStream_t)stream, stats, slots_per_group, count, gamma, beta,
                     running_mean, running_var, (long long*)nbt, momentum, eps, aux, C, stat_out);
  HIPCHK(hipGetLastError());
  return bn_act_fwd_pitched(prec, y, aux, slope, a, rows_per_group, C, groups, 0, 0, (hipStream_t)stream);
}

extern "C" size_t jck_bn_bwd_ws_floats(int C) { return (size_t)(2 + 2 * BN_BWD_MAX_BLOCKS) * C; }
// workgroups of the backward reduction (see bn_bwd_reduce_kernel for the measurement behind the cap)
static int bn_bwd_blocks(long long rows, int rstep, int groups) {
  (void)groups;
  return (int)std::max<long long>(1, std::min<long long>((rows + rstep * 4 - 1) / (rstep * 4), BN_BWD_MAX_BLOCKS));
}

static int bn_act_bwd_grouped_ev(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y,
                                 float* dgamma, float* dbeta, long long rows_per_group, int C, int groups, int grad_groups, hipStream_t st,
                                 hipEvent_t done);
extern "C" int jck_bn_act_bwd(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums,
                              void* g_y, float* dgamma, float* dbeta, long long rows, int C, void* stream) {
  return bn_act_bwd_grouped_ev(prec, g_a, y, aux, slope, sums, g_y, dgamma, dbeta, rows, C, 1, 1, (hipStream_t)stream, nullptr);
}

// Grouped forms: `groups` independent BatchNorm batches stored back to back ([groups][rows][C] tensors, [groups][slots]
// statistics slots, [groups][4C] aux, [groups][jck_bn_bwd_ws_floats(C)] backward workspace).  They serve D passes that
// share weights and went through ONE conv launch (train/dcgan_trainer.py:162,173,118 run D on three batches with the
// same weights); each group is normalised with its own batch statistics exactly as the separate passes are.
extern "C" int jck_bn_finalize_grouped(const float* stats, int slots_per_group, float count, const float* gamma, const float* beta,
                                       float eps, float* aux, float* stat_out, int C, int groups, void* stream) {
  if (slots_per_group < 1 || groups < 1) JCK_FAIL(JCK_E_ARG, "bn_finalize_grouped: slots and groups must be >= 1");
  if (C % 4) JCK_FAIL(JCK_E_ARG, "bn_finalize_grouped: C % 4 != 0");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 4, groups), dim3(256), 0, (hipStream_t)stream, stats, slots_per_group, count, gamma,
                     beta, (float*)nullptr, (float*)nullptr, (long long*)nullptr, 0.f, eps, aux, C, stat_out);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_bn_act_fwd_grouped(int prec, const void* y, const float* aux, float slope, void* a, long long rows_per_group,
                                      int C, int groups, void* stream) {
  return bn_act_fwd_pitched(prec, y, aux, slope, a, rows_per_group, C, groups, 0, 0, (hipStream_t)stream);
}
// out_pitch > 0: every out_row (a power of two, >= 8) elements of the output start out_pitch elements apart (ew.hpp)
int bn_act_fwd_pitched(int prec, const void* y, const float* aux, float slope, void* a, long long rows_per_group, int C, int groups,
                       long long out_row, long long out_pitch, hipStream_t stream) {
  if (!is_pow2(C) || C < 8 || groups < 1) JCK_FAIL(JCK_E_ARG, "bn_act_fwd_grouped: C must be a power of two >= 8");
  if (out_pitch && (out_row < 8 || (out_row & (out_row - 1)) || out_pitch < out_row || out_pitch % 8))
    JCK_FAIL(JCK_E_ARG, "bn_act_fwd: a pitched output needs rows of a power of two >= 8 elements, pitch >= row, pitch % 8 == 0");
  const long long total8 = rows_per_group * C / 8;
  ProfScope prof(PROF_BN_ACT_FWD, 0.0, stream, 2.0 * groups * rows_per_group * C * (prec == JCK_PREC_F32 ? 4 : 2));
  DISPATCH_T(prec, hipLaunchKernelGGL(bn_act_fwd_kernel<T>, dim3(ew_grid(total8), groups), dim3(256), 0, stream,
                                      (const T*)y, aux, slope, (T*)a, total8, C, out_pitch ? ilog2((int)out_row) : 0, out_pitch));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_bn_act_bwd_grouped(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums,
                                      void* g_y, float* dgamma, float* dbeta, long long rows_per_group, int C, int groups,
                                      int grad_groups, void* stream) {
  return bn_act_bwd_grouped_ev(prec, g_a, y, aux, slope, sums, g_y, dgamma, dbeta, rows_per_group, C, groups, grad_groups, (hipStream_t)stream, nullptr);
}
// (`done`: completed by the launch that writes g_y)
static int bn_act_bwd_grouped_ev(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y,
                                 float* dgamma, float* dbeta, long long rows_per_group, int C, int groups, int grad_groups, hipStream_t stream,
                                 hipEvent_t done) {
  if (!is_pow2(C) || C < 8 || C > 2048 || groups < 1) JCK_FAIL(JCK_E_ARG, "bn_act_bwd_grouped: C must be a power of two in [8, 2048]");
  const long long rows = rows_per_group;
  const int rstep = 256 / (C / 8);
  if (rstep < 1) JCK_FAIL(JCK_E_ARG, "bn_act_bwd_grouped: C too large");
  // algorithmic bytes of the backward: read g_a and y once, write g_y (what the resident form moves; this form reads twice)
  ProfScope prof(PROF_BN_BWD_3L, 0.0, (hipStream_t)stream, 3.0 * groups * rows * C * (prec == JCK_PREC_F32 ? 4 : 2));
  const int blocks = bn_bwd_blocks(rows, rstep, groups);
  const long long gstride = (long long)jck_bn_bwd_ws_floats(C);
  float* partial = sums + 2 * C;
  if (g_bn_unr >= 4) { DISPATCH_T(prec, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 4>), dim3(blocks, groups), dim3(256), 2 * C * rstep * sizeof(float),
                                      (hipStream_t)stream, (const T*)g_a, (const T*)y, aux, slope, partial, rows, C, gstride)); }
  else if (g_bn_unr == 2) { DISPATCH_T(prec, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 2>), dim3(blocks, groups), dim3(256), 2 * C * rstep * sizeof(float),
                                      (hipStream_t)stream, (const T*)g_a, (const T*)y, aux, slope, partial, rows, C, gstride)); }
  else { DISPATCH_T(prec, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), dim3(blocks, groups), dim3(256), 2 * C * rstep * sizeof(float),
                                      (hipStream_t)stream, (const T*)g_a, (const T*)y, aux, slope, partial, rows, C, gstride)); }
  HIPCHK(hipGetLastError());
  if (g_bn_bwd_fuse && C >= 64 && C % 64 == 0 && (prec == JCK_PREC_BF16 || g_bn_bwd_fuse > 1)) {
    // two launches: every workgroup of the apply sums the partial rows of its 64-channel slice itself (ew.hpp).  The fast path's
    // form: the fp32 parity path keeps the three launches and with them the summation order its step tolerances were measured
    // with (a free-running second step amplifies a last-bit change of the sums to 1e-3 of D(G(z))); bn_bwd_fuse = 2 takes it there too
    const int nsl = C / 64;
    const long long per = std::max<long long>(1, g_bn_bwd_fuse_wgs / ((long long)nsl * groups));
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((rows + 31) / 32, per));
    DISPATCH_T(prec, LAUNCH_EV(bn_bwd_apply_fused_kernel<T>, dim3(gx, nsl, groups), dim3(BNF_THREADS), 0, (hipStream_t)stream, done,
                               (const T*)g_a, (const T*)y, (const float*)aux, (const float*)partial, blocks, sums, dgamma, dbeta, slope,
                               1.0f / (float)rows, (T*)g_y, rows, C, gstride, grad_groups));
    HIPCHK(hipGetLastError());
    return JCK_OK;
  }
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(C / 4), dim3(256), 0, (hipStream_t)stream, partial, blocks, sums, dgamma, dbeta, C,
                     gstride, groups, grad_groups);
  HIPCHK(hipGetLastError());
  const long long total8 = rows * C / 8;
  DISPATCH_T(prec, LAUNCH_EV(bn_bwd_apply_kernel<T>, dim3(ew_grid(total8), groups), dim3(256), 0, (hipStream_t)stream, done,
                             (const T*)g_a, (const T*)y, (const float*)aux, (const float*)sums, slope, 1.0f / (float)rows, (T*)g_y, total8, C, gstride));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// Resident form (bnres.hpp): one launch per layer, every tensor byte read once, the groups one after the other.  Falls back to
// the three-launch form above when the layer does not fit the register file of the chip (or is not bf16, or JCK_BN_RES=0).
static int bnres_cus() {
  static int cus = [] {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
    return pr.multiProcessorCount;
  }();
  return cus;
}
static unsigned long long* g_bnres_stamps = nullptr;
extern "C" int jck_debug_bnres_stamps(void* buf) { g_bnres_stamps = (unsigned long long*)buf; return JCK_OK; }   // [256][8] u64, development aid
extern "C" size_t jck_grid_sync_bytes(void) { return BNRES_SYNC_BYTES; }
// 1 if a grid barrier of a resident launch timed out since the state was last zeroed (synchronises the device)
const unsigned* jck_grid_sync_error_word(const void* sync_ws) { return sync_ws ? (const unsigned*)sync_ws + BNRES_W_ERR * 32 : nullptr; }
extern "C" int jck_grid_sync_error(const void* sync_ws) {
  unsigned err = 0;
  if (!sync_ws) return 0;
  if (hipMemcpy(&err, (const unsigned*)sync_ws + BNRES_W_ERR * 32, sizeof(err), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return err != 0;
}
// nb workgroups (all co-resident), nsl channel slices, chunks per thread; 0 = does not fit
static int bnres_plan(int prec, long long rows, int C, int groups, int* nb_out, int* nsl_out) {
  if (!g_bn_res || prec != JCK_PREC_BF16 || !is_pow2(C) || C < 64 || groups < 1 || rows < 1) return 0;
  // Several groups in one pass (the batched D pass): the three-launch form's second read of (g_a, y) comes out of the
  // Infinity Cache while all groups fit it, and then it is as fast or faster in the step (measured, DESIGN.md section 5.3:
  // D.conv2's layer at 3 x 256 images 60 vs 70 us alone, the step 1.744 vs 1.733 ms); the resident form wins where they do not
  // fit (D.conv1's layer: 3 x 2 x 33.5 MB, or 2 x 2 x 33.5 MB for its loss groups alone).  bn_res = 2 takes the resident form whenever it fits the registers.
  if (g_bn_res == 1 && groups > 1 && (long long)groups * rows * C * 4 <= ((long long)g_bn_res_mb << 20) &&
      (long long)groups * rows * C * 4 > ((long long)g_bn_res_small_mb << 20)) return 0;
  const int nsl = C / 64;
  int nb = std::min(bnres_cus(), 256);
  nb -= nb % std::max(nsl, 8);
  if (nb < nsl || nb < 8) return 0;
  if ((long long)groups * rows * C * 2 >= (1ll << 40)) return 0;
  const long long per_iter = (long long)(nb / nsl) * BNRES_ROWS;
  const long long need = (rows + per_iter - 1) / per_iter;
  if (need > 16) return 0;
  *nb_out = nb; *nsl_out = nsl;
  return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : 16;
}
extern "C" int jck_bn_act_bwd_res(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y,
                                  float* dgamma, float* dbeta, long long rows_per_group, int C, int groups, int grad_groups,
                                  void* sync_ws, void* stream) {
  return bn_act_bwd_res_ev(prec, g_a, y, aux, slope, sums, g_y, dgamma, dbeta, rows_per_group, C, groups, grad_groups, sync_ws,
                           (hipStream_t)stream, nullptr);
}
int bn_act_bwd_res_ev(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y, float* dgamma,
                      float* dbeta, long long rows_per_group, int C, int groups, int grad_groups, void* sync_ws, hipStream_t stream,
                      hipEvent_t done) {
  int nb = 0, nsl = 0;
  const int nch = sync_ws ? bnres_plan(prec, rows_per_group, C, groups, &nb, &nsl) : 0;
  if (!nch) return bn_act_bwd_grouped_ev(prec, g_a, y, aux, slope, sums, g_y, dgamma, dbeta, rows_per_group, C, groups, grad_groups, stream, done);
  BnResParams p;
  p.ga = (const bf16_t*)g_a; p.y = (const bf16_t*)y; p.gy = (bf16_t*)g_y; p.aux = aux;
  p.sums = sums; p.sums_stride = (long long)jck_bn_bwd_ws_floats(C);
  p.dgamma = dgamma; p.dbeta = dbeta; p.sync = (unsigned*)sync_ws; p.stamps = g_bnres_stamps;
  p.rows = rows_per_group; p.C = C; p.groups = groups; p.grad_groups = grad_groups; p.nb = nb; p.nsl = nsl;
  p.slope = slope; p.inv_count = 1.0f / (float)rows_per_group;
  const dim3 grid(nb), block(BNRES_THREADS);
  ProfScope prof(PROF_BN_BWD_RES, 0.0, stream, 3.0 * groups * rows_per_group * C * 2);
  // groups resident together (one barrier for all of them) while their chunks fit the register file
  const int ng = (groups >= 3 && 3 * nch <= 16) ? 3 : (groups >= 2 && 2 * nch <= 16) ? 2 : 1;
#define BNRES_CASE(NCH_, NG_) case NCH_ * 4 + NG_: LAUNCH_EV((bn_bwd_res_kernel<NCH_, NG_>), grid, block, 0, stream, done, p); break
  switch (nch * 4 + ng) {
    BNRES_CASE(1, 1); BNRES_CASE(2, 1); BNRES_CASE(4, 1); BNRES_CASE(8, 1); BNRES_CASE(16, 1);
    BNRES_CASE(1, 2); BNRES_CASE(2, 2); BNRES_CASE(4, 2); BNRES_CASE(8, 2);
    BNRES_CASE(1, 3); BNRES_CASE(2, 3); BNRES_CASE(4, 3);
    default: JCK_FAIL(JCK_E_ARG, "bn_act_bwd_res: no kernel for this plan");
  }
#undef BNRES_CASE
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// images, heads
// ---------------------------------------------------------------------------------------------------------
extern "C" int jck_resize_norm(const float* in, float* out, int N, int C, int H, int W, int OH, int OW, float pre_scale,
                               float pre_shift, const float* mean, const float* stdv, void* stream) {
  if (!in || !out || !mean || !stdv || N < 1 || C < 1 || H < 1 || W < 1 || OH < 1 || OW < 1) JCK_FAIL(JCK_E_ARG, "resize_norm: bad arguments");
  hipLaunchKernelGGL(resize_norm_kernel, dim3(ew_grid((long long)N * C * OH * OW)), dim3(256), 0, (hipStream_t)stream, in, out, N, C, H,
                     W, OH, OW, pre_scale, pre_shift, mean, stdv);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_img_prep_u8(int prec, const unsigned char* data, const int64_t* idx, const float* noise, float keep, float mix,
                               void* out_nhwc4, float* out_nchw, int B, int Hs, int Ws, void* stream) {
  if (!data || B < 1 || Hs < 1 || Ws < 1) JCK_FAIL(JCK_E_ARG, "img_prep_u8: bad arguments");
  if (!out_nhwc4 && !out_nchw) return JCK_OK;
  DISPATCH_T(prec, hipLaunchKernelGGL(img_prep_u8_kernel<T>, dim3(ew_grid((long long)B * 4 * Hs * Ws)), dim3(256), 0,
                                      (hipStream_t)stream, data, (const long long*)idx, noise, keep, mix, (T*)out_nhwc4, out_nchw, B,
                                      Hs, Ws));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// in-kernel Philox noise instead of an uploaded noise tensor (ew.hpp: pixel_normals); rng = device uint32[4]
extern "C" int jck_img_prep_rng(int prec, const float* img, const unsigned* rng, int tensor_id, float keep, float mix, void* out, int N,
                                int HW, void* stream) {
  if (!rng) JCK_FAIL(JCK_E_ARG, "img_prep_rng: rng is NULL");
  DISPATCH_T(prec, hipLaunchKernelGGL(img_prep_kernel<T>, dim3(ew_grid((long long)N * HW)), dim3(256), 0, (hipStream_t)stream, img,
                                      (const float*)nullptr, keep, mix, (T*)out, N, HW, rng, (unsigned)tensor_id));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_img_prep_u8_rng(int prec, const unsigned char* data, const int64_t* idx, const unsigned* rng, int tensor_id,
                                   float keep, float mix, void* out_nhwc4, int B, int Hs, int Ws, void* stream) {
  if (!data || !rng || !out_nhwc4 || B < 1) JCK_FAIL(JCK_E_ARG, "img_prep_u8_rng: bad arguments");
  DISPATCH_T(prec, hipLaunchKernelGGL(img_prep_u8_kernel<T>, dim3(ew_grid((long long)B * 4 * Hs * Ws)), dim3(256), 0, (hipStream_t)stream,
                                      data, (const long long*)idx, (const float*)nullptr, keep, mix, (T*)out_nhwc4, (float*)nullptr, B,
                                      Hs, Ws, rng, (unsigned)tensor_id));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_axpy_noise_rng(int prec, const void* x, const unsigned* rng, int tensor_id, float keep, float mix, void* out, int N,
                                  int HW, void* stream) {
  if (!rng) JCK_FAIL(JCK_E_ARG, "axpy_noise_rng: rng is NULL");
  DISPATCH_T(prec, hipLaunchKernelGGL(axpy_noise_kernel<T>, dim3(ew_grid((long long)N * HW)), dim3(256), 0, (hipStream_t)stream,
                                      (const T*)x, (const float*)nullptr, keep, mix, (T*)out, N, HW, rng, (unsigned)tensor_id));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_img_prep(int prec, const float* img, const float* noise, float keep, float mix, void* out, int N, int HW,
                            void* stream) {
  DISPATCH_T(prec, hipLaunchKernelGGL(img_prep_kernel<T>, dim3(ew_grid((long long)N * HW)), dim3(256), 0,
                                      (hipStream_t)stream, img, noise, keep, mix, (T*)out, N, HW));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_nhwc4_to_nchw(int prec, const void* in, float* out, int N, int HW, void* stream) {
  DISPATCH_T(prec, hipLaunchKernelGGL(nhwc4_to_nchw_kernel<T>, dim3(ew_grid((long long)N * HW)), dim3(256), 0,
                                      (hipStream_t)stream, (const T*)in, out, N, HW));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_axpy_noise(int prec, const void* x, const float* noise, float keep, float mix, void* out, int N, int HW,
                              void* stream) {
  DISPATCH_T(prec, hipLaunchKernelGGL(axpy_noise_kernel<T>, dim3(ew_grid((long long)N * HW)), dim3(256), 0,
                                      (hipStream_t)stream, (const T*)x, noise, keep, mix, (T*)out, N, HW));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// jck_axpy_noise (noise given) / jck_axpy_noise_rng (rng given) followed by jck_interp(real, out, alpha) -> xhat as ONE launch
extern "C" int jck_mix_interp(int prec, const void* x, const float* noise, const unsigned* rng, int tensor_id, float keep, float mix,
                              void* out, const void* real, const float* alpha, void* xhat, int N, int HW, void* stream) {
  if (!real || !alpha || !xhat) JCK_FAIL(JCK_E_ARG, "mix_interp: real / alpha / xhat is NULL");
  if ((noise != nullptr) == (rng != nullptr)) JCK_FAIL(JCK_E_ARG, "mix_interp: exactly one of noise and rng");
  DISPATCH_T(prec, hipLaunchKernelGGL(axpy_noise_kernel<T>, dim3(ew_grid((long long)N * HW)), dim3(256), 0, (hipStream_t)stream,
                                      (const T*)x, noise, keep, mix, (T*)out, N, HW, rng, (unsigned)tensor_id, (const T*)real, alpha, (T*)xhat));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_interp(int prec, const void* a, const void* b, const float* alpha, void* out, int N, int HW, void* stream) {
  DISPATCH_T(prec, hipLaunchKernelGGL(interp_kernel<T>, dim3(ew_grid((long long)N * HW)), dim3(256), 0, (hipStream_t)stream,
                                      (const T*)a, (const T*)b, alpha, (T*)out, N, HW));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_gp_norm(int prec, const void* g, int N, int HW, float* scal, int slot, int scal_ld, float* norms, void* stream) {
  if (scal && slot >= 0 && scal_ld < N) JCK_FAIL(JCK_E_ARG, "gp_norm: scal_ld < N");
  DISPATCH_T(prec, hipLaunchKernelGGL(gp_norm_kernel<T>, dim3(N), dim3(256), 0, (hipStream_t)stream, (const T*)g, HW, scal,
                                      slot, scal_ld, norms));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_tanh_bwd(int prec, const void* g, const void* y, float scale, void* out, long long numel, void* stream) {
  return tanh_bwd_ev(prec, g, y, scale, out, numel, (hipStream_t)stream, nullptr);
}
int tanh_bwd_ev(int prec, const void* g, const void* y, float scale, void* out, long long numel, hipStream_t stream, hipEvent_t done) {
  if (numel % 4) JCK_FAIL(JCK_E_ARG, "tanh_bwd: numel % 4 != 0");
  DISPATCH_T(prec, LAUNCH_EV(tanh_bwd_kernel<T>, dim3(ew_grid(numel / 4)), dim3(256), 0, stream, done,
                             (const T*)g, (const T*)y, scale, (T*)out, numel / 4));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// G (<= 4) batches of B rows stacked in a4 / prob / ds, each with its own target, mode and scalar slots - one launch
extern "C" int jck_head_fwd_grouped(int prec, const void* a4, const float* wp, const float* bias, int B, int K, int G,
                                    const float* targets, const int* modes, float* prob, float* ds, float* scal,
                                    const int* slot_loss, const int* slot_p, int scal_ld, void* stream) {
  return head_fwd_grouped_ev(prec, a4, wp, bias, B, K, G, targets, modes, prob, ds, scal, slot_loss, slot_p, scal_ld, nullptr,
                             (hipStream_t)stream, nullptr);
}
// g_out (optional): the rows' input gradient ds[n] * wp[k] from the same launch; done (optional): completed by the launch
int head_fwd_grouped_ev(int prec, const void* a4, const float* wp, const float* bias, int B, int K, int G, const float* targets,
                        const int* modes, float* prob, float* ds, float* scal, const int* slot_loss, const int* slot_p, int scal_ld,
                        void* g_out, hipStream_t stream, hipEvent_t done) {
  if (K % 8) JCK_FAIL(JCK_E_ARG, "head_fwd: K % 8 != 0");
  if (G < 1 || G > 4 || B < 1) JCK_FAIL(JCK_E_ARG, "head_fwd: 1..4 groups of >= 1 rows");
  HeadGroups hg = {};
  hg.rows_per_group = B;
  for (int g = 0; g < G; ++g) {
    hg.target[g] = targets[g]; hg.mode[g] = modes[g]; hg.slot_loss[g] = slot_loss[g]; hg.slot_p[g] = slot_p[g];
    if ((slot_loss[g] >= 0 || slot_p[g] >= 0) && (!scal || scal_ld < B)) JCK_FAIL(JCK_E_ARG, "head_fwd: scalar slots need scal with scal_ld >= B");
  }
  DISPATCH_T(prec, LAUNCH_EV(head_fwd_kernel<T>, dim3(G * B), dim3(256), 0, stream, done, (const T*)a4, wp, K, bias,
                             hg, 1.0f / (float)B, prob, ds, scal, scal_ld, (T*)g_out));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_head_fwd(int prec, const void* a4, const float* wp, const float* bias, int B, int K, float target, int mode,
                            float* prob, float* ds, float* scal, int slot_loss, int slot_p, int scal_ld, void* stream) {
  return jck_head_fwd_grouped(prec, a4, wp, bias, B, K, 1, &target, &mode, prob, ds, scal, &slot_loss, &slot_p, scal_ld, stream);
}
// jck_linear_finish + jck_head_fwd_grouped + the input-gradient half of jck_head_bwd + jck_dropout of CGAN's head in one launch
// (ew.hpp: cg_head_mid_kernel); N = 256 columns; rows = G * B; mask required
int cg_head_mid(int prec, const float* slab, int ksplit, const float* bias1, const float* mask, float scale, void* h, void* hd, const float* w2,
                const float* bias2, int B, int G, const float* targets, const int* modes, float* prob, float* ds, float* scal,
                const int* slot_loss, const int* slot_p, int scal_ld, void* g_hd, void* g_h, hipStream_t stream) {
  if (G < 1 || G > 4 || B < 1 || !mask || !slab || ksplit < 1) JCK_FAIL(JCK_E_ARG, "cg_head_mid: 1..4 groups of >= 1 rows, a dropout mask, split-K slabs");
  HeadGroups hg = {};
  hg.rows_per_group = B;
  for (int g = 0; g < G; ++g) {
    hg.target[g] = targets[g]; hg.mode[g] = modes[g]; hg.slot_loss[g] = slot_loss[g]; hg.slot_p[g] = slot_p[g];
    if ((slot_loss[g] >= 0 || slot_p[g] >= 0) && (!scal || scal_ld < B)) JCK_FAIL(JCK_E_ARG, "cg_head_mid: scalar slots need scal with scal_ld >= B");
  }
  const long long rows = (long long)G * B;
  DISPATCH_T(prec, hipLaunchKernelGGL(cg_head_mid_kernel<T>, dim3((unsigned)rows), dim3(256), 0, stream, slab, ksplit, rows * 256, bias1, mask, scale,
                                      (T*)h, (T*)hd, w2, bias2, hg, 1.0f / (float)B, prob, ds, scal, scal_ld, (T*)g_hd, (T*)g_h));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
#define HEAD_NS 8          /* partial rows of jck_head_bwd */
#define HEAD_CONV_NS 16    /* partial rows of jck_head_bwd_conv */
extern "C" size_t jck_head_bwd_ws_floats(int K) { return (size_t)HEAD_CONV_NS * K; }
extern "C" int jck_head_bwd(int prec, const float* ds, const float* wp, const void* a4, int B, int K, void* g_a4, float* dwp,
                            int accumulate, float* ws, void* stream) {
  if (K % 8) JCK_FAIL(JCK_E_ARG, "head_bwd: K % 8 != 0");
  if (dwp && !ws) JCK_FAIL(JCK_E_ARG, "head_bwd: the weight gradient needs a workspace of jck_head_bwd_ws_floats(K) floats");
  if (g_a4) {
    const long long total8 = (long long)B * K / 8;
    DISPATCH_T(prec, hipLaunchKernelGGL(head_dgrad_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, ds, wp,
                                        K, (T*)g_a4, total8));
    HIPCHK(hipGetLastError());
  }
  if (dwp) {
    DISPATCH_T(prec, hipLaunchKernelGGL(head_wgrad_kernel<T>, dim3(cdiv(K / 8, 64), HEAD_NS), dim3(256), 0, (hipStream_t)stream, ds,
                                        (const T*)a4, B, K, ws));
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(head_part_reduce_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, ws, HEAD_NS, K, 0, dwp, accumulate);
    HIPCHK(hipGetLastError());
  }
  return JCK_OK;
}
extern "C" int jck_head_bwd_conv(int prec, const float* ds, const float* wp, const void* a4, int B, int C, void* g_a4,
                                 float* grad, float* ws, void* stream) {
  return jck_head_bwd_conv2(prec, ds, wp, a4, B, 0, C, g_a4, grad, ws, stream);
}
// jck_head_bwd_conv over B rows and, in the same launch, the input gradient alone for the B_more rows behind them (ds, a4, g_a4 hold
// B + B_more rows; the weight gradient sums the first B only) - the loss groups and the penalty group of the batched D pass
extern "C" int jck_head_bwd_conv2(int prec, const float* ds, const float* wp, const void* a4, int B, int B_more, int C, void* g_a4,
                                  float* grad, float* ws, void* stream) {
  return head_bwd_conv2_ev(prec, ds, wp, a4, B, B_more, C, g_a4, grad, ws, (hipStream_t)stream, nullptr, nullptr);
}
// side / handover (both or neither): the ordered sum of the weight-gradient partial rows - wanted by the optimiser only - runs on
// `side` behind `handover`, which the launch that writes the rows completes itself
int head_bwd_conv2_ev(int prec, const float* ds, const float* wp, const void* a4, int B, int B_more, int C, void* g_a4, float* grad, float* ws,
                      hipStream_t stream, hipStream_t side, hipEvent_t handover) {
  if (C % 8) JCK_FAIL(JCK_E_ARG, "head_bwd_conv: C % 8 != 0");
  if ((side != nullptr) != (handover != nullptr)) JCK_FAIL(JCK_E_ARG, "head_bwd_conv: side stream and hand-over event go together");
  if (B_more < 0 || (B_more > 0 && !g_a4)) JCK_FAIL(JCK_E_ARG, "head_bwd_conv2: the extra rows produce an input gradient only");
  if (!g_a4 && !grad) return JCK_OK;
  if (grad && !ws) JCK_FAIL(JCK_E_ARG, "head_bwd_conv: the weight gradient needs a workspace of jck_head_bwd_ws_floats(16*C) floats");
  const int K = 16 * C;
  if (!g_a4 && grad && side) {
    // only the weight gradient is wanted (the rows' input gradient came out of head_fwd_grouped_ev, which completed `handover`):
    // the partial rows and their sum both run on `side`
    HIPCHK(hipStreamWaitEvent(side, handover, 0));
    DISPATCH_T(prec, hipLaunchKernelGGL(head_bwd_fused_kernel<T>, dim3(cdiv(K / 8, 64), HEAD_CONV_NS), dim3(256), 0, side, ds, wp,
                                        (const T*)a4, B, K, C, (T*)nullptr, ws, 0));
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(head_part_reduce_kernel, dim3(cdiv(K, 256)), dim3(256), 0, side, ws, HEAD_CONV_NS, K, C, grad, 1);
    HIPCHK(hipGetLastError());
    return JCK_OK;
  }
  hipEvent_t ev = grad ? handover : nullptr;
  DISPATCH_T(prec, LAUNCH_EV(head_bwd_fused_kernel<T>, dim3(cdiv(K / 8, 64), HEAD_CONV_NS), dim3(256), 0, stream, ev, ds, wp,
                             (const T*)a4, B, K, C, (T*)g_a4, grad ? ws : nullptr, B_more));
  HIPCHK(hipGetLastError());
  if (grad) {
    hipStream_t rs = stream;
    if (ev) { HIPCHK(hipStreamWaitEvent(side, ev, 0)); rs = side; }
    hipLaunchKernelGGL(head_part_reduce_kernel, dim3(cdiv(K, 256)), dim3(256), 0, rs, ws, HEAD_CONV_NS, K, C, grad, 1);
    HIPCHK(hipGetLastError());
  }
  return JCK_OK;
}
extern "C" int jck_head_unpack_grad(const float* dwp, int C, float* grad, int accumulate, void* stream) {
  return launch_wgrad_reduce(dwp, 1, 1, 16 * C, 1, C, ilog2(C), grad, accumulate, (hipStream_t)stream);
}

extern "C" int jck_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                        double eps, int step, float grad_scale, void* stream) {
  if (step < 1) JCK_FAIL(JCK_E_ARG, "adam: step is 1-based");
  // scalar preparation in double exactly as torch.optim.Adam does it in Python, then one cast to float
  const double bc1 = 1.0 - std::pow(beta1, step), bc2 = 1.0 - std::pow(beta2, step);
  const float step_size = (float)(lr / bc1), bc2s = (float)std::sqrt(bc2);
  const int vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(vec ? (n + 3) / 4 : n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, bc2s, grad_scale, (const float*)nullptr, vec);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// the same update with {step_size, bc2_sqrt} read from device memory: jck_adam_set_step writes them (same host arithmetic)
// ... and (rz / ralpha / rmasks, each optional) the step's small random inputs, drawn by the same launch (ew.hpp: adam_hp_kernel)
int jck_adam_set_step(float* hp, double lr, double beta1, double beta2, int step, unsigned long long seed, hipStream_t st, float* rz,
                      long long nz, float* ralpha, long long nalpha, float* rmasks, long long nmask, float keep_p, float* zero,
                      long long nzero, float* zbig0, long long nzbig0, float* zbig1, long long nzbig1, void* zpad, int zd, int zp, int zpad_f32) {
  if (step < 1) JCK_FAIL(JCK_E_ARG, "adam: step is 1-based");
  if (((uintptr_t)zbig0 | (uintptr_t)zbig1) & 15 || (nzbig0 | nzbig1) & 3) JCK_FAIL(JCK_E_ARG, "set_step: large zero ranges must be 16-byte aligned, counts % 4 == 0");
  const double bc1 = 1.0 - std::pow(beta1, step), bc2 = 1.0 - std::pow(beta2, step);
  StepRng r = {rz, rz ? nz : 0, ralpha, ralpha ? nalpha : 0, rmasks, rmasks ? nmask : 0, keep_p, zero, zero ? nzero : 0,
               {zbig0, zbig1}, {zbig0 ? nzbig0 : 0, zbig1 ? nzbig1 : 0}, (rz && zd > 0 && zp >= zd) ? zpad : nullptr, zd, zp, zpad_f32};
  const long long quads = std::max((r.nz + 3) / 4 + (r.nalpha + 3) / 4 + (r.nmask + 3) / 4, std::max(r.nzbig[0], r.nzbig[1]) / 16);
  const unsigned blocks = (unsigned)std::max<long long>(1, std::min<long long>((quads + 255) / 256, 1024));
  hipLaunchKernelGGL(adam_hp_kernel, dim3(blocks), dim3(256), 0, st, hp, (float)(lr / bc1), (float)std::sqrt(bc2), (unsigned)seed,
                     (unsigned)(seed >> 32), (unsigned)step, r);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// test / binding access to the per-step draws: fills z [nz] ~ N(0,1), alpha [nalpha] ~ U[0,1), masks [nmask] ~ Bernoulli(keep_p)
// exactly as jck_engine_set_step does for step `step` and noise seed `seed`; hp: device float[8] scratch
extern "C" int jck_step_rng(float* hp, int step, unsigned long long seed, float* z, long long nz, float* alpha, long long nalpha,
                            float* masks, long long nmask, float keep_p, void* stream) {
  if (!hp) JCK_FAIL(JCK_E_ARG, "step_rng: hp scratch (8 floats) is required");
  return jck_adam_set_step(hp, 2e-4, 0.5, 0.999, step, seed, (hipStream_t)stream, z, nz, alpha, nalpha, masks, nmask, keep_p);
}
int jck_adam_hp(float* p, const float* g, float* m, float* v, long long n, double beta1, double beta2, double eps,
                float grad_scale, const float* hp, hipStream_t st, float* zero, long long nzero, const unsigned* skip_if) {
  const int vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (zero && (((uintptr_t)zero & 15) || (nzero & 3))) JCK_FAIL(JCK_E_ARG, "adam: the zero range must be 16-byte aligned, count % 4 == 0");
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(vec ? (n + 3) / 4 : n)), dim3(256), 0, st, p, g, m, v, n, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps, 0.f, 1.f, grad_scale, hp, vec, zero, zero ? nzero / 4 : 0, skip_if);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// debug probe: what ds_read_b64_tr_b16 returns for the addressing wgrad.hpp uses (pins the hardware
// semantics the weight-gradient kernel relies on; exercised by tests/test_ops_gpu.py)
// ---------------------------------------------------------------------------------------------------------
__global__ void debug_tr_kernel(const bf16_t* __restrict__ in, int ld, bf16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* t = reinterpret_cast<bf16_t*>(smem_raw);
  for (int i = threadIdx.x; i < 32 * ld; i += 64) t[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x;
  const int trow = (lane >> 4) * 8 + ((lane & 15) >> 2), tcol = (lane & 3) * 4;
  short4v a = lds_tr4(t + trow * ld + tcol), b = lds_tr4(t + (trow + 4) * ld + tcol);
  for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = (bf16_t)a[j]; out[lane * 8 + 4 + j] = (bf16_t)b[j]; }
}
extern "C" int jck_debug_tr_read(const void* in, int ld, void* out, void* stream) {
  if (ld % 4 || ld < 16) JCK_FAIL(JCK_E_ARG, "ld must be a multiple of 4 and >= 16");
  hipLaunchKernelGGL(debug_tr_kernel, dim3(1), dim3(64), 32 * ld * 2, (hipStream_t)stream, (const bf16_t*)in, ld, (bf16_t*)out);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// CGAN: Linear layers, label embedding, concat, dropout, second-order terms of the gradient penalty
// ---------------------------------------------------------------------------------------------------------
extern "C" int jck_pack_linear(int prec, const float* w, int N, int K, int rows, int cols, int transpose, int permC, int permHW,
                               void* wp, void* stream) {
  const long long total = (long long)rows * cols;
  DISPATCH_T(prec, hipLaunchKernelGGL(pack_linear_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, N, K, rows,
                                      cols, transpose, permC, permHW, (T*)wp));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// jck_pack_linear(transpose = 0) into wp0[rows0][cols0] and jck_pack_linear(transpose = 1) into wp1[rows1][cols1], one launch
int pack_linear_pair(int prec, const float* w, int N, int K, int rows0, int cols0, void* wp0, int rows1, int cols1, void* wp1, int permC,
                     int permHW, hipStream_t stream) {
  const long long total = std::max((long long)rows0 * cols0, (long long)rows1 * cols1);
  DISPATCH_T(prec, hipLaunchKernelGGL(pack_linear_pair_kernel<T>, dim3(ew_grid(total), 2), dim3(256), 0, stream, w, N, K, rows0, cols0,
                                      (T*)wp0, rows1, cols1, (T*)wp1, permC, permHW));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// out[B][NStore] (T, or fp32 slabs [ksplit][B][NStore] when ksplit > 1) = x[B][Kpad] * wp[rows][Kpad]^T (+ bias)
extern "C" int jck_linear_fwd(int prec, const void* x, const void* wp, const float* bias, void* out, int B, int Kpad, int N,
                              int NStore, int ksplit, void* stream) {
  if (Kpad % 64 || NStore % 4) JCK_FAIL(JCK_E_ARG, "linear_fwd: Kpad % 64 or NStore % 4");
  const int rows = jck_pad_rows(N);
  if (rows % 128) JCK_FAIL(JCK_E_ARG, "linear_fwd: N must be >= 65");
  IgemmParams p = {};
  p.act = x; p.w = wp; p.out = out; p.stats = nullptr;
  p.M = B; p.NchStore = std::min(NStore, rows); p.K = Kpad; p.logC = 30; p.H = 1; p.W = 1; p.logOW = 0; p.logOHW = 0;
  p.sy = p.sx = 1; p.ntaps = 1; p.act_row_elems = Kpad; p.bias = ksplit > 1 ? nullptr : bias;
  p.osN = NStore; p.cstat = 4; p.ytiles_per_cset = 1;
  int phases = 1;
  if (ksplit > 1) {
    const int nk = Kpad / 64;
    if (nk % ksplit) JCK_FAIL(JCK_E_ARG, "linear_fwd: k-steps not divisible by ksplit");
    p.ksplit = ksplit; p.ksteps = nk / ksplit; p.out_split_stride = (long long)B * NStore; p.out_f32 = 1;
    phases = ksplit;
  }
  p.flops = 2.0 * B * N * (double)Kpad;
  return launch_igemm(prec, p, rows, phases, 1, (hipStream_t)stream, nullptr);
}

extern "C" size_t jck_linear_wgrad_ws_bytes(int B, int Kpad, int N) { return plan_wgrad(B, Kpad, N).ws; }
// gradp[N][Kpad] fp32 (+)= gy[B][N]^T * x[B][Kpad]     (our column order; see jck_unperm_linear_grad)
extern "C" int jck_linear_wgrad(int prec, const void* gy, int ldgy, const void* x, int Kpad, float* ws, size_t ws_bytes,
                                float* gradp, int accumulate, int B, int N, void* stream) {
  if (Kpad % 64 || ldgy % 8) JCK_FAIL(JCK_E_ARG, "linear_wgrad: bad leading dimensions");
  WgradParams p = {};
  p.sside = gy; p.big = x; p.Mtot = B; p.CsStride = ldgy; p.logCb = 30; p.H = 1; p.W = 1; p.logOW = 0; p.logOHW = 0;
  p.sy = p.sx = 1; p.ntaps = 1; p.dy[0] = 0; p.dx[0] = 0; p.big_row_elems = Kpad;
  const WgradPlan pl = plan_wgrad(B, Kpad, N);
  p.flops = 2.0 * B * N * (double)Kpad;
  int rc = run_wgrad(prec, p, pl, 1, ws, ws_bytes, (hipStream_t)stream);
  if (rc) return rc;
  const long long total = (long long)N * Kpad;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, ws, pl.Z, pl.CsRows, pl.ncols, N, Kpad, 0, 1, gradp, accumulate);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_unperm_linear_grad(const float* gp, int N, int K, int ldp, int permC, int permHW, float* grad, int accumulate,
                                      void* stream) {
  const long long total = (long long)N * K;
  hipLaunchKernelGGL(unperm_linear_grad_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, gp, N, K, ldp, permC,
                     permHW, grad, accumulate);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_linear_finish(int prec, const float* slab, int Z, const float* bias, const float* mask, float scale, void* h,
                                 void* hd, int B, int N, void* stream) {
  DISPATCH_T(prec, hipLaunchKernelGGL(linear_finish_kernel<T>, dim3(cdiv(B * N, 256)), dim3(256), 0, (hipStream_t)stream, slab, Z,
                                      (long long)B * N, bias, mask, scale, (T*)h, (T*)hd, B, N));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// `_tiled`: B rows that are `label_period`-row batches stacked on top of each other, all labelled by the same [label_period][NI]
// tensor (the real | fake | penalty groups of one CGAN step share the batch's labels, train/cgan_trainer.py:181-203)
extern "C" int jck_label_embed_fwd_tiled(int prec, const int64_t* labels, const float* W, const float* b, float slope, int B, int NI,
                                         int NO, void* cbuf, int ld, int col0, float* pre, int label_period, void* stream) {
  if (label_period < 0 || (label_period > 0 && B % label_period)) JCK_FAIL(JCK_E_ARG, "label_embed: rows are not a multiple of the label period");
  DISPATCH_T(prec, hipLaunchKernelGGL(label_embed_fwd_kernel<T>, dim3(B), dim3(256), (size_t)NI * sizeof(float), (hipStream_t)stream,
                                      (const long long*)labels, W, b, slope, B, NI, NO, (T*)cbuf, ld, col0, pre, label_period));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_label_embed_fwd(int prec, const int64_t* labels, const float* W, const float* b, float slope, int B, int NI,
                                   int NO, void* cbuf, int ld, int col0, float* pre, void* stream) {
  return jck_label_embed_fwd_tiled(prec, labels, W, b, slope, B, NI, NO, cbuf, ld, col0, pre, 0, stream);
}
extern "C" int jck_label_embed_bwd_tiled(int prec, const void* gc, int ld, int col0, const float* pre, const int64_t* labels, float slope,
                                         int B, int NI, int NO, float* dW, float* db, int label_period, void* stream) {
  if (label_period < 0 || (label_period > 0 && B % label_period)) JCK_FAIL(JCK_E_ARG, "label_embed: rows are not a multiple of the label period");
  DISPATCH_T(prec, hipLaunchKernelGGL(label_embed_bwd_kernel<T>, dim3(NI + cdiv(NO, 16)), dim3(256), (size_t)B * (sizeof(float) + sizeof(int)), (hipStream_t)stream,
                                      (const T*)gc, ld, col0, pre, (const long long*)labels, slope, B, NI, NO, dW, db, label_period));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_label_embed_bwd(int prec, const void* gc, int ld, int col0, const float* pre, const int64_t* labels, float slope,
                                   int B, int NI, int NO, float* dW, float* db, void* stream) {
  return jck_label_embed_bwd_tiled(prec, gc, ld, col0, pre, labels, slope, B, NI, NO, dW, db, 0, stream);
}
extern "C" int jck_concat_rows(int prec, const void* a4, int K0, void* cbuf, int ld, int B, void* stream) {
  if (K0 % 8 || ld % 8) JCK_FAIL(JCK_E_ARG, "concat_rows: K0 % 8 or ld % 8");
  const long long total8 = (long long)B * K0 / 8;
  DISPATCH_T(prec, hipLaunchKernelGGL(concat_rows_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, (const T*)a4,
                                      K0, (T*)cbuf, ld, total8));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_split_rows(int prec, const void* gc, int ld, int K0, void* ga4, int B, void* stream) {
  if (K0 % 8 || ld % 8) JCK_FAIL(JCK_E_ARG, "split_rows: K0 % 8 or ld % 8");
  const long long total8 = (long long)B * K0 / 8;
  DISPATCH_T(prec, hipLaunchKernelGGL(split_rows_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, (const T*)gc, ld,
                                      K0, (T*)ga4, total8));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_dropout(int prec, const void* x, const float* mask, float scale, void* y, long long n, void* stream) {
  DISPATCH_T(prec, hipLaunchKernelGGL(dropout_kernel<T>, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, mask, scale,
                                      (T*)y, n));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_colsum(int prec, const void* g, int B, int N, int ld, float* db, void* stream) {
  DISPATCH_T(prec, hipLaunchKernelGGL(colsum_kernel<T>, dim3(cdiv(N, COLSUM_COLS)), dim3(256), 0, (hipStream_t)stream, (const T*)g, B, N, ld, db));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_gp_grad(int prec, const void* g, const float* norms, float coef, int N, int HW, void* u, void* stream) {
  const long long total4 = (long long)N * HW;
  DISPATCH_T(prec, hipLaunchKernelGGL(gp_grad_kernel<T>, dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, (const T*)g, norms,
                                      coef, HW * 4, (T*)u, total4));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// ws: float[B + jck_head_bwd_ws_floats(K)] (pq[n] = p(1-p), then the partial rows of the dw2 sum)
extern "C" int jck_gp_head2(int prec, const void* ughd, const float* w2, const float* prob, int B, int K, float* rs, float* dw2,
                            float* ws, void* stream) {
  return gp_head2_ev(prec, ughd, w2, prob, B, K, rs, dw2, ws, (hipStream_t)stream, nullptr, nullptr);
}
// side / handover (both or neither): the dw2 sum - wanted by the optimiser only - runs on `side` behind `handover`, which the
// launch that writes rs and pq completes itself
int gp_head2_ev(int prec, const void* ughd, const float* w2, const float* prob, int B, int K, float* rs, float* dw2, float* ws,
                hipStream_t stream, hipStream_t side, hipEvent_t handover) {
  if (K % 8 || !ws) JCK_FAIL(JCK_E_ARG, "gp_head2: K % 8 != 0 or no workspace");
  if ((side != nullptr) != (handover != nullptr)) JCK_FAIL(JCK_E_ARG, "gp_head2: side stream and hand-over event go together");
  DISPATCH_T(prec, LAUNCH_EV(gp_head2_kernel<T>, dim3(B), dim3(256), 0, stream, handover, (const T*)ughd, w2, prob, B, K, rs, ws));
  HIPCHK(hipGetLastError());
  hipStream_t gs = stream;
  if (side) { HIPCHK(hipStreamWaitEvent(side, handover, 0)); gs = side; }
  // dw2[j] += sum_n pq[n] * ughd[n][j]
  float* part = ws + (B + 63) / 64 * 64;
  DISPATCH_T(prec, hipLaunchKernelGGL(head_wgrad_kernel<T>, dim3(cdiv(K / 8, 64), HEAD_NS), dim3(256), 0, gs, ws, (const T*)ughd, B, K, part));
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(head_part_reduce_kernel, dim3(cdiv(K, 256)), dim3(256), 0, gs, part, HEAD_NS, K, 0, dw2, 1);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// jck_linear_finish(bias = NULL, h = NULL) + gp_head2_ev + the input-gradient half of jck_head_bwd(ds = rs) + jck_dropout in one launch
// (ew.hpp: cg_gp_head_mid_kernel; 256 columns), then the dw2 sum as in gp_head2_ev (on `side` behind `handover` when given)
int gp_head_mid_ev(int prec, const float* slab, int ksplit, const float* mask, float scale, void* ughd, const float* w2, const float* prob, int B,
                   float* rs, float* dw2, float* ws, void* g_hd, void* g_h, hipStream_t stream, hipStream_t side, hipEvent_t handover) {
  if (!ws || !mask || !slab || ksplit < 1) JCK_FAIL(JCK_E_ARG, "gp_head_mid: workspace, dropout mask and split-K slabs are required");
  if ((side != nullptr) != (handover != nullptr)) JCK_FAIL(JCK_E_ARG, "gp_head_mid: side stream and hand-over event go together");
  const int K = 256;
  DISPATCH_T(prec, LAUNCH_EV(cg_gp_head_mid_kernel<T>, dim3(B), dim3(256), 0, stream, handover, slab, ksplit, (long long)B * K, mask, scale,
                             (T*)ughd, w2, prob, rs, ws, (T*)g_hd, (T*)g_h));
  HIPCHK(hipGetLastError());
  hipStream_t gs = stream;
  if (side) { HIPCHK(hipStreamWaitEvent(side, handover, 0)); gs = side; }
  float* part = ws + (B + 63) / 64 * 64;
  DISPATCH_T(prec, hipLaunchKernelGGL(head_wgrad_kernel<T>, dim3(cdiv(K / 8, 64), HEAD_NS), dim3(256), 0, gs, ws, (const T*)ughd, B, K, part));
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(head_part_reduce_kernel, dim3(cdiv(K, 256)), dim3(256), 0, gs, part, HEAD_NS, K, 0, dw2, 1);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// v-chain step of the penalty's double backward at one BatchNorm layer.  ws: jck_bn2_ws_floats(C) floats; on return
// ws[0..3C) = {sum v, sum v*xhat, sum v*gy} (keep it for jck_bn2_reverse).  u may alias v.
extern "C" size_t jck_bn2_ws_floats(int C) { return (size_t)(4 + 4 * BN_BWD_MAX_BLOCKS) * C; }
extern "C" int jck_bn2_vchain(int prec, const void* v, const void* y, const void* gy, const float* aux, const float* s1,
                              const float* gamma, float slope, float* ws, void* u, void* xdir, float* dgamma, long long rows, int C,
                              void* stream) {
  if (!is_pow2(C) || C < 8 || C > 2048) JCK_FAIL(JCK_E_ARG, "bn2_vchain: C must be a power of two in [8, 2048]");
  const int rstep = 256 / (C / 8);
  const int blocks = bn_bwd_blocks(rows, rstep, 1);
  float* partial = ws + 4 * C;
  DISPATCH_T(prec, hipLaunchKernelGGL((bn2_reduce_kernel<T, 1>), dim3(blocks), dim3(256), 3 * C * rstep * sizeof(float),
                                      (hipStream_t)stream, (const T*)v, (const T*)y, (const T*)gy, aux, slope, partial, rows, C));
  HIPCHK(hipGetLastError());
  if (g_bn_bwd_fuse && C >= 64 && C % 64 == 0 && (prec == JCK_PREC_BF16 || g_bn_bwd_fuse > 1)) {
    // two launches: the apply sums the partial rows of its own channel slice (ew.hpp: bn2_vchain_apply_fused_kernel)
    const int nsl = C / 64;
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((rows + 31) / 32, std::max(1, g_bn_bwd_fuse_wgs / nsl)));
    DISPATCH_T(prec, hipLaunchKernelGGL(bn2_vchain_apply_fused_kernel<T>, dim3(gx, nsl), dim3(256), 0, (hipStream_t)stream, (const T*)v, (const T*)y,
                                        (const T*)gy, aux, s1, (const float*)partial, blocks, ws, gamma, dgamma, slope, 1.0f / (float)rows,
                                        (T*)u, (T*)xdir, rows, C));
    HIPCHK(hipGetLastError());
    return JCK_OK;
  }
  hipLaunchKernelGGL(bn2_sums_kernel, dim3(3 * C / 4), dim3(256), 0, (hipStream_t)stream, partial, blocks, 3, C, ws, dgamma ? 1 : 0, gamma,
                     dgamma, (float*)nullptr);
  HIPCHK(hipGetLastError());
  const long long total8 = rows * C / 8;
  DISPATCH_T(prec, hipLaunchKernelGGL(bn2_vchain_apply_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, (const T*)v,
                                      (const T*)y, (const T*)gy, aux, s1, ws, slope, 1.0f / (float)rows, (T*)u, (T*)xdir, total8, C));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
// reverse-sweep BatchNorm backward with the penalty's extra inputs (xdir, sum v*gy from the v-chain at vsums[2C..3C)).
extern "C" int jck_bn2_reverse(int prec, const void* ua, const void* y, const void* xdir, const float* aux, const float* gamma,
                               const float* vsums, float slope, float* ws, void* uy, float* dgamma, float* dbeta, long long rows,
                               int C, void* stream) {
  if (!is_pow2(C) || C < 8 || C > 2048) JCK_FAIL(JCK_E_ARG, "bn2_reverse: C must be a power of two in [8, 2048]");
  const int rstep = 256 / (C / 8);
  const int blocks = bn_bwd_blocks(rows, rstep, 1);
  float* partial = ws + 4 * C;
  DISPATCH_T(prec, hipLaunchKernelGGL((bn2_reduce_kernel<T, 2>), dim3(blocks), dim3(256), 4 * C * rstep * sizeof(float),
                                      (hipStream_t)stream, (const T*)ua, (const T*)y, (const T*)xdir, aux, slope, partial, rows, C));
  HIPCHK(hipGetLastError());
  if (g_bn_bwd_fuse && C >= 64 && C % 64 == 0 && (prec == JCK_PREC_BF16 || g_bn_bwd_fuse > 1)) {
    const int nsl = C / 64;
    const unsigned gx = (unsigned)std::max<long long>(1, std::min<long long>((rows + 31) / 32, std::max(1, g_bn_bwd_fuse_wgs / nsl)));
    DISPATCH_T(prec, hipLaunchKernelGGL(bn2_reverse_apply_fused_kernel<T>, dim3(gx, nsl), dim3(256), 0, (hipStream_t)stream, (const T*)ua,
                                        (const T*)y, (const T*)xdir, aux, gamma, (const float*)partial, blocks, ws, vsums + 2 * C,
                                        (dgamma && dbeta) ? dgamma : nullptr, (dgamma && dbeta) ? dbeta : nullptr, slope,
                                        1.0f / (float)rows, (T*)uy, rows, C));
    HIPCHK(hipGetLastError());
    return JCK_OK;
  }
  hipLaunchKernelGGL(bn2_sums_kernel, dim3(4 * C / 4), dim3(256), 0, (hipStream_t)stream, partial, blocks, 4, C, ws,
                     (dgamma && dbeta) ? 2 : 0, (const float*)nullptr, dgamma, dbeta);
  HIPCHK(hipGetLastError());
  const long long total8 = rows * C / 8;
  DISPATCH_T(prec, hipLaunchKernelGGL(bn2_reverse_apply_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                      (const T*)ua, (const T*)y, (const T*)xdir, aux, gamma, ws, vsums + 2 * C, slope,
                                      1.0f / (float)rows, (T*)uy, total8, C));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

extern "C" int jck_sum_vec(const float* x, int n, float* out, void* stream) {
  hipLaunchKernelGGL(sum_vec_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
extern "C" int jck_cgan_z(int prec, const float* z, const int64_t* labels, int B, int NZ, int NL, int CiPad, void* out, void* stream) {
  const long long total = (long long)B * CiPad;
  DISPATCH_T(prec, hipLaunchKernelGGL(cgan_z_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, z,
                                      (const long long*)labels, B, NZ, NL, CiPad, (T*)out));
  HIPCHK(hipGetLastError());
  return JCK_OK;
}

// development probe: copies the per-wave stamp totals of the last stamped wgrad_dma launch (see wgrad.hpp) to the host
extern "C" int jck_debug_wgrad_stamps(unsigned long long* out, int n) {
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgd_stamps), (size_t)n * sizeof(unsigned long long)));
  return JCK_OK;
}
