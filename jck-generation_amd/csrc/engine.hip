// Whole-step engine: the reference's training iteration (train/dcgan_trainer.py:155-189) as one native
// schedule of kernel launches on a HIP stream.  The host makes one call per phase; nothing in here
// allocates, synchronises or reads back - scalars stay on the device until the trainer logs them.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ops_internal.hpp"
#define JCK_BATCHED_DEFAULT 3
#define TT (e->T)        /* the engine's channel plan; `e` is the engine in every function below (this inside its members) */

namespace {

struct TensorInfo { const char* name; int kind; long long offset, numel; int shape[4]; };

struct NetLayout {
  std::vector<TensorInfo> t;
  long long n_params = 0, n_bn = 0;
};

constexpr long long ALIGN_F = 64;   // floats

void add_param(NetLayout& L, const char* name, int a, int b, int c, int d) {
  TensorInfo ti{name, 0, L.n_params, (long long)a * b * c * d, {a, b, c, d}};
  L.t.push_back(ti);
  L.n_params = (L.n_params + ti.numel + ALIGN_F - 1) / ALIGN_F * ALIGN_F;
}
void add_bn(NetLayout& L, const char* wn, const char* bn, const char* rm, const char* rv, int c) {
  add_param(L, wn, c, 1, 1, 1);
  add_param(L, bn, c, 1, 1, 1);
  TensorInfo m{rm, 1, L.n_bn, c, {c, 1, 1, 1}};
  L.t.push_back(m);
  L.n_bn += (c + ALIGN_F - 1) / ALIGN_F * ALIGN_F;
  TensorInfo v{rv, 2, L.n_bn, c, {c, 1, 1, 1}};
  L.t.push_back(v);
  L.n_bn += (c + ALIGN_F - 1) / ALIGN_F * ALIGN_F;
}

// Channel plan.  S = 64 is the reference's topology (model/DCGAN.py:10-26, :42-58): NS = 4 stride-2 stages, 512 channels at
// the 4x4 end.  S = 128 (BASELINE.json configs[4]; no reference behaviour exists - SURVEY section 8d) adds ONE stage at the
// deep end, the usual way a DCGAN is grown: D 3 -> 64 -> 128 -> 256 -> 512 -> 1024 -> 1, G 100 -> 1024 -> ... -> 64 -> 3,
// so every layer keeps >= 64 channels on its wide side and the image-side layers stay 3 <-> 64.
#define JCK_MAX_STAGES 5
struct Topo {
  int S, NS, HW;                       // image size, stride-2 stages, S*S
  int D_CS[JCK_MAX_STAGES], D_CB[JCK_MAX_STAGES], D_HB[JCK_MAX_STAGES];
  int G_CS[JCK_MAX_STAGES], G_CB[JCK_MAX_STAGES], G_HS[JCK_MAX_STAGES];     // G.conv2 .. G.conv(NS+1)
  int G_C1, FEAT;                      // channels at the 4x4 end; D's flattened top features 16*G_C1
};
Topo make_topo(int S) {
  Topo T = {};
  T.S = S; T.NS = S == 128 ? 5 : 4; T.HW = S * S;
  T.G_C1 = 64 << (T.NS - 1);
  T.FEAT = 16 * T.G_C1;
  for (int i = 0; i < T.NS; ++i) {
    T.D_CS[i] = 64 << i; T.D_CB[i] = i == 0 ? 3 : 32 << i; T.D_HB[i] = S >> i;
    T.G_CS[i] = T.G_C1 >> i; T.G_CB[i] = i == T.NS - 1 ? 3 : T.G_C1 >> (i + 1); T.G_HS[i] = 4 << i;
  }
  return T;
}
const int N_CLASS = 100, EMB = 200, L1_OUT = 256, L1_FEAT = 8192, L1_K = L1_FEAT + EMB, L1_KPAD = 8448;   // model/CGAN.py:83,104
// split-K of Linear(8392, 256): 132 k-steps of 64 over L1_KSPLIT workgroups per tile (a divisor of 132; JCK_L1_KSPLIT to A/B)
static const int L1_KSPLIT = [] { const char* v = getenv("JCK_L1_KSPLIT"); const int k = v ? atoi(v) : 12; return (k > 0 && 132 % k == 0) ? k : 12; }();
inline int z_dim(int family) { return family == 1 ? 200 : 100; }      // model/CGAN.py:132: ConvTranspose2d(200, 512)
inline int z_pad(int family) { return family == 1 ? 256 : 128; }

static const char* const CWN[6] = {"conv1.weight", "conv2.weight", "conv3.weight", "conv4.weight", "conv5.weight", "conv6.weight"};
static const char* const NWN[5] = {"norm1.weight", "norm2.weight", "norm3.weight", "norm4.weight", "norm5.weight"};
static const char* const NBN[5] = {"norm1.bias", "norm2.bias", "norm3.bias", "norm4.bias", "norm5.bias"};
static const char* const RMN[5] = {"norm1.running_mean", "norm2.running_mean", "norm3.running_mean", "norm4.running_mean", "norm5.running_mean"};
static const char* const RVN[5] = {"norm1.running_var", "norm2.running_var", "norm3.running_var", "norm4.running_var", "norm5.running_var"};

NetLayout make_layout(int family, int net, int S = 64) {
  NetLayout L;
  const Topo T = make_topo(S);
  const char* const* CW = CWN; const char* const* NW = NWN; const char* const* NB = NBN; const char* const* RM = RMN; const char* const* RV = RVN;
  if (net == 0) {
    add_param(L, CW[0], z_dim(family), T.G_C1, 4, 4);
    add_bn(L, NW[0], NB[0], RM[0], RV[0], T.G_C1);
    for (int i = 0; i < T.NS; ++i) {
      add_param(L, CW[i + 1], T.G_CS[i], T.G_CB[i], 4, 4);
      if (i < T.NS - 1) add_bn(L, NW[i + 1], NB[i + 1], RM[i + 1], RV[i + 1], T.G_CB[i]);
    }
  } else {
    if (family == 1) {                                  // model/CGAN.py:83: registered first
      add_param(L, "label_embedding.weight", EMB, N_CLASS, 1, 1);
      add_param(L, "label_embedding.bias", EMB, 1, 1, 1);
    }
    for (int i = 0; i < T.NS; ++i) {
      add_param(L, CW[i], T.D_CS[i], T.D_CB[i], 4, 4);
      add_bn(L, NW[i], NB[i], RM[i], RV[i], T.D_CS[i]);
    }
    if (family == 0) {
      add_param(L, CW[T.NS], 1, T.G_C1, 4, 4);
    } else {                                            // model/CGAN.py:104,106
      add_param(L, "linear1.weight", L1_OUT, L1_K, 1, 1);
      add_param(L, "linear1.bias", L1_OUT, 1, 1, 1);
      add_param(L, "linear2.weight", 1, L1_OUT, 1, 1);
      add_param(L, "linear2.bias", 1, 1, 1, 1);
    }
  }
  return L;
}

const TensorInfo* find(const NetLayout& L, const char* name) {
  for (auto& t : L.t) if (!strcmp(t.name, name)) return &t;
  return nullptr;
}

struct Carver {
  size_t off = 0;
  unsigned char* base = nullptr;
  template <typename T> T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off = (off + count * sizeof(T) + 255) / 256 * 256;
    return p;
  }
};

struct BnBuf { float *stats, *sums, *aux; int slots; };

}  // namespace

// the engine's events order kernels of ONE device across its streams: no system-scope fence (cache write-back for the host's sake)
// when they are recorded
static unsigned jck_event_flags() { return hipEventDisableTiming | hipEventDisableSystemFence; }
struct jck_engine {
  int family, prec, B;
  Topo T;
  size_t esz;
  NetLayout LG, LD;
  size_t ws_bytes = 0;
  bool bound = false;
  // arenas
  float *gp = nullptr, *gg = nullptr, *gm = nullptr, *gv = nullptr, *gbn = nullptr;
  float *dp = nullptr, *dg = nullptr, *dm = nullptr, *dv = nullptr, *dbn = nullptr;
  int64_t *gnbt = nullptr, *dnbt = nullptr;
  // packed weights
  void *d_down[JCK_MAX_STAGES], *d_up[JCK_MAX_STAGES];
  float *d_head_wp, *d_head_dwp;
  void *g1_w, *g_up[JCK_MAX_STAGES], *g_down[JCK_MAX_STAGES];
  // activations: three B-image D sets for the per-pass schedules (each pass that may run concurrently has its own)
  struct DSet { void *y[JCK_MAX_STAGES], *a[JCK_MAX_STAGES], *g[JCK_MAX_STAGES], *gx; BnBuf bn[JCK_MAX_STAGES]; float *prob, *ds, *norms; } dset[3];      // 0: D(fake) and the G-phase pass, 1: penalty pass, 2: D(real) (may overlap the previous step's G phase)
  void **d_y = dset[0].y, **d_a = dset[0].a, **d_g = dset[0].g;
  void*& d_gx = dset[0].gx;
  BnBuf* d_bn = dset[0].bn;
  float*& prob = dset[0].prob; float*& ds = dset[0].ds; float*& norms = dset[0].norms;
  // batched D passes (DCGAN): up to 3 batches that share D's weights go through ONE launch per layer, BatchNorm grouped
  struct BSet { void *y[JCK_MAX_STAGES], *a[JCK_MAX_STAGES], *g[JCK_MAX_STAGES]; float *stats[JCK_MAX_STAGES], *aux[JCK_MAX_STAGES], *sums[JCK_MAX_STAGES]; float *prob, *ds; } bset;
  int batched = 0;                      // 0: one D pass per batch (batch % 8 != 0, JCK_BATCHED=0), 3: [real | fake | penalty] as ONE 3B pass
  bool gp_done = false;
  long long real_fwd_step = -1;         // step whose D(real) forward already ran (PHASE_D_REAL_FWD), -1: none
  int head_row0 = 0;                    // CGAN: first row of the head state (h_drop, cbuf, pre_e) the pass in hand READS (2B: penalty group of a batched head)
  int head_wrow0 = 0;                   //       first row of the head buffers it WRITES (g_hd, g_h, gc)
  float* d_rs[JCK_MAX_STAGES];                       // deferred BatchNorm running-stat records of D: [step parity][pass 0..3][2*C] per layer
  int parity = 0;                       // step & 1: selects the scalar accumulators and the BN records of the step in flight
  // side streams: A = weight gradients beside the dgrad chain, B = G forward beside D(real), C = penalty pass beside D(fake)
  hipStream_t sA = nullptr, sB = nullptr, sC = nullptr;
  hipEvent_t evW[JCK_MAX_STAGES] = {}, evWdone = nullptr, evWmid = nullptr, evHead = nullptr, evTail = nullptr, ev0 = nullptr, evF = nullptr, evReal = nullptr, evGP = nullptr;
  bool overlap = true, gp_inflight = false, defer_join = true;
  bool tail_on_side = false, pre_on_side = false;       // PHASE_D_LOSS_A's tail marker sits on sA; PHASE_D_REAL_FWD ran on sA (evReal)
  bool join_pending = false, mid_recorded = false;    // PHASE_LAZY_JOIN: evWdone (and evWmid) recorded on sA, not yet waited for
  // cross-stream hand-overs of the backward: the producing launch completes the event itself (the `done` argument of
  // bn_act_bwd_res_ev / tanh_bwd_ev: hipExtLaunchKernel's stop event) instead of a
  // hipEventRecord behind it; JCK_EXT_EVENTS=0 restores the records
  bool ext_events = true;
  // resident one-launch BatchNorm backward (bnres.hpp): grid-barrier state (zeroed at bind); JCK_BN_RES=0 switches it off
  unsigned* gsync = nullptr; bool bn_res = true;
  bool cbuf_direct = true;              // CGAN: D's last BatchNorm+LeakyReLU writes straight into the head's concat buffer (JCK_CBUF_DIRECT=0: a copy)
  bool head_fuse = true;                // CGAN: Linear finish + Linear(256,1)/sigmoid/BCE + its input gradient + Dropout backward as one launch; DCGAN: conv5's
                                        // forward writes its input gradient too (JCK_HEAD_FUSE=0: separate launches)
  bool real_side = true;                // batched D pass: D(real)'s forward beside G's forward on the weight-gradient stream (JCK_REAL_SIDE=0: one 3B forward)
  bool fold_zero = true;                // zero_grad() of both networks inside neighbouring launches (JCK_FOLD_ZERO=0: memsets)
  bool fuse_tanh = true;                // G's loss pass: tanh backward in the epilogue of D.conv1's input gradient (JCK_FUSE_TANH=0: a launch of its own)
  void *g_z, *g_y[JCK_MAX_STAGES], *g_a[JCK_MAX_STAGES], *g_gr[JCK_MAX_STAGES], *fake_raw, *fake, *g_raw;
  void *real_noisy, *xhat;
  // small buffers
  unsigned char* zero_d; size_t zero_d_bytes;      // stats + sums of D's 4 layers
  unsigned char* zero_g; size_t zero_g_bytes;
  BnBuf g_bn[JCK_MAX_STAGES];
  float *acc, *scal_out;                // current-parity views into acc2 / scal2
  float *acc2, *scal2;                  // acc2: [2 parities][8 rows][acc_ld] per-image scalar table (summed by the step tail)
  int acc_ld = 0;
  float *head_ws, *gp2_ws;              // partial rows of the head weight gradients (deterministic two-stage sums)
  float* hp2;                           // [2 parities][8]: {step_size, bc2_sqrt, -, -, noise seed lo, hi, step, 0} of the step in flight (jck_engine_set_step)
  unsigned long long noise_seed = 0x6a636b67616e0001ull;      // in-kernel instance noise (jck_engine_set_noise_seed)
  float *rz[2] = {nullptr, nullptr}, *ralpha[2] = {nullptr, nullptr}, *rmask[2] = {nullptr, nullptr};
  int hp_step[2] = {0, 0};              // which step's scalars each parity holds (checked by the optimiser phases) ...
  float hp_lr[2] = {-1.f, -1.f};        // ... and at which learning rate they were computed (a scheduler may change it between steps)
  bool hp_holds(int step, float lr) const { return hp_step[step & 1] == step && hp_lr[step & 1] == lr; }
  int acc_clean_step = -1;              // the step whose accumulator rows jck_engine_set_step has just cleared (consumed by its first D phase)
  // zero_grad() without a launch of its own: D's gradient arena (+ CGAN's permuted Linear gradient) is cleared by the set-step
  // launch of the step's first D phase, G's by D's Adam launch; the step whose arena is clean (consumed by the phase that would memset)
  int dg_clean_step = -1, gg_clean_step = -1;
  bool capturing = false;
  float* g1_ws = nullptr; size_t g1_ws_bytes = 0;
  float* wg_ws; size_t wg_ws_bytes;
  long long gz_step = -1;               // the step whose set-step launch wrote its own z into g_z (G.conv1's operand) as well
  // family 1 (CGAN): Linear head, label path, second-order penalty buffers
  void *l1_w, *l1_wT;                 // packed linear1: [256][8448] and transposed [8448][256]
  void *cbuf, *cbuf2;                 // [B][8448] concat(flatten(a4), e) ; [u4 | 0]
  void *h_pre, *h_drop, *g_h, *g_hd, *gh_b1, *ughd, *gc;    // [B][256] x6, [B][8448]
  float *pre_e, *l1_slab, *gw1p, *rs, *prob_gp;
  void *d_v[JCK_MAX_STAGES], *d_xdir[JCK_MAX_STAGES], *d_u0;
  float* bn2_ws[JCK_MAX_STAGES]; float* bn2_ws_rev;
  const int64_t* cur_labels = nullptr;

  void carve(unsigned char* base) {
    jck_engine* e = this;
    Carver c; c.base = base;
    auto bytes = [&](size_t n) { return n * esz; };
    for (int i = 0; i < TT.NS; ++i) {
      d_down[i] = c.take<unsigned char>(bytes((size_t)jck_pad_rows(TT.D_CS[i]) * 16 * jck_pad_chan(TT.D_CB[i])));
      d_up[i] = c.take<unsigned char>(bytes((size_t)4 * jck_pad_rows(TT.D_CB[i]) * 4 * TT.D_CS[i]));
    }
    d_head_wp = c.take<float>(TT.FEAT); d_head_dwp = c.take<float>(TT.FEAT);
    g1_w = c.take<unsigned char>(bytes((size_t)16 * TT.G_C1 * z_pad(family)));
    for (int i = 0; i < TT.NS; ++i) {
      g_up[i] = c.take<unsigned char>(bytes((size_t)4 * jck_pad_rows(TT.G_CB[i]) * 4 * TT.G_CS[i]));
      g_down[i] = c.take<unsigned char>(bytes((size_t)jck_pad_rows(TT.G_CS[i]) * 16 * jck_pad_chan(TT.G_CB[i])));
    }
    const size_t img = (size_t)B * TT.HW * 4;
    for (int sI = 0; sI < 3; ++sI) {
      DSet& D = dset[sI];
      for (int i = 0; i < TT.NS; ++i) {
        const size_t n = (size_t)B * (TT.D_HB[i] / 2) * (TT.D_HB[i] / 2) * TT.D_CS[i];
        D.y[i] = c.take<unsigned char>(bytes(n)); D.a[i] = c.take<unsigned char>(bytes(n)); D.g[i] = c.take<unsigned char>(bytes(n));
        D.bn[i].sums = c.take<float>(jck_bn_bwd_ws_floats(TT.D_CS[i]));
        D.bn[i].aux = c.take<float>(4 * TT.D_CS[i]);
        D.bn[i].stats = c.take<float>(jck_stats_floats((long long)B * (TT.D_HB[i] / 2) * (TT.D_HB[i] / 2), TT.D_CS[i], 1));
      }
      D.gx = c.take<unsigned char>(bytes(img));
      D.prob = c.take<float>(B); D.ds = c.take<float>(B); D.norms = c.take<float>(B);
    }
    for (int i = 0; i < TT.NS; ++i) d_rs[i] = c.take<float>(2 * 4 * 2 * TT.D_CS[i]);
    g_z = c.take<unsigned char>(bytes((size_t)B * z_pad(family)));
    // G layer i (0..3): output of conv(i+1) = [B, h, h, C] with (h, C) = (4,512), (8,256), (16,128), (32,64)
    for (int i = 0; i < TT.NS; ++i) {
      const int h = 4 << i, C = TT.G_C1 >> i;
      const size_t n = (size_t)B * h * h * C;
      g_y[i] = c.take<unsigned char>(bytes(n)); g_a[i] = c.take<unsigned char>(bytes(n)); g_gr[i] = c.take<unsigned char>(bytes(n));
    }
    fake_raw = c.take<unsigned char>(bytes(img)); g_raw = c.take<unsigned char>(bytes(img));
    // real_noisy | fake | xhat are consecutive (img bytes % 256 == 0): the batched D pass reads them as one 3B-image tensor
    real_noisy = c.take<unsigned char>(bytes(img)); fake = c.take<unsigned char>(bytes(img)); xhat = c.take<unsigned char>(bytes(img));
    if (batched) {
      for (int i = 0; i < TT.NS; ++i) {
        const size_t n = (size_t)3 * B * (TT.D_HB[i] / 2) * (TT.D_HB[i] / 2) * TT.D_CS[i];
        bset.y[i] = c.take<unsigned char>(bytes(n)); bset.a[i] = c.take<unsigned char>(bytes(n)); bset.g[i] = c.take<unsigned char>(bytes(n));
        bset.sums[i] = c.take<float>(3 * jck_bn_bwd_ws_floats(TT.D_CS[i]));
        bset.aux[i] = c.take<float>(3 * 4 * TT.D_CS[i]);
        bset.stats[i] = c.take<float>(jck_stats_floats((long long)3 * B * (TT.D_HB[i] / 2) * (TT.D_HB[i] / 2), TT.D_CS[i], 1));
      }
      bset.prob = c.take<float>(3 * B); bset.ds = c.take<float>(3 * B);
    }
    // zeroed-per-pass regions
    {
      size_t start = c.off;
      zero_d = nullptr; zero_d_bytes = 0;
      start = c.off;
      zero_g = base ? base + start : nullptr;
      for (int i = 0; i < TT.NS; ++i) g_bn[i].sums = c.take<float>(jck_bn_bwd_ws_floats(TT.G_C1 >> i));
      zero_g_bytes = c.off - start;
      for (int i = 0; i < TT.NS; ++i) {
        const int C = TT.G_C1 >> i, h = 4 << i;
        g_bn[i].aux = c.take<float>(4 * C);
        g_bn[i].stats = c.take<float>(jck_stats_floats((long long)B * h * h, C, i == 0 ? 16 : 1));
      }
    }
    acc_ld = (B + 63) / 64 * 64;
    acc2 = c.take<float>((size_t)2 * 8 * acc_ld); scal2 = c.take<float>(16);
    acc = acc2; scal_out = scal2;
    hp2 = c.take<float>(16);
    // the step's small random inputs when the caller hands over none, per step parity (jck_engine_set_step draws them)
    for (int q = 0; q < 2; ++q) {
      rz[q] = c.take<float>((size_t)B * 100); ralpha[q] = c.take<float>((size_t)(B + 3) / 4 * 4);
      rmask[q] = family == 1 ? c.take<float>((size_t)4 * B * L1_OUT) : nullptr;
    }
    head_ws = c.take<float>(jck_head_bwd_ws_floats(TT.FEAT));
    gp2_ws = c.take<float>((size_t)acc_ld + jck_head_bwd_ws_floats(L1_OUT));
    gsync = c.take<unsigned>(jck_grid_sync_bytes() / sizeof(unsigned));
    size_t w = 0;
    for (int i = 0; i < TT.NS; ++i) {
      w = std::max(w, jck_conv_wgrad_ws_bytes(B, TT.D_HB[i], TT.D_HB[i], TT.D_CB[i], TT.D_CS[i]));
      w = std::max(w, jck_conv_wgrad_ws_bytes(2 * B, TT.D_HB[i], TT.D_HB[i], TT.D_CB[i], TT.D_CS[i]));
      w = std::max(w, jck_conv_wgrad_ws_bytes(B, TT.G_HS[i] * 2, TT.G_HS[i] * 2, TT.G_CB[i], TT.G_CS[i]));
    }
    w = std::max(w, jck_g1_wgrad_ws_bytes(B, z_pad(family), TT.G_C1));
    if (family == 1) w = std::max(std::max(w, jck_linear_wgrad_ws_bytes(B, L1_KPAD, L1_OUT)), jck_linear_wgrad_ws_bytes(2 * B, L1_KPAD, L1_OUT));
    wg_ws_bytes = w;
    wg_ws = c.take<float>(w / 4);
    // G.conv1's weight gradient runs on the MAIN stream at the very end of G's backward (it needs the last BatchNorm backward,
    // which is on that stream): its own split-K workspace, because the side stream may still be using wg_ws
    g1_ws_bytes = jck_g1_wgrad_ws_bytes(B, z_pad(family), TT.G_C1);
    g1_ws = c.take<float>(g1_ws_bytes / 4);
    if (family == 1) {
      l1_w = c.take<unsigned char>(bytes((size_t)L1_OUT * L1_KPAD)); l1_wT = c.take<unsigned char>(bytes((size_t)L1_KPAD * L1_OUT));
      // head buffers: HR batches of rows - the batched schedule runs the label / Linear / Dropout head of the real | fake |
      // penalty groups as ONE 3B-row pass (cg_head_batched)
      const size_t HR = batched ? 3 : 1;
      cbuf = c.take<unsigned char>(bytes(HR * B * L1_KPAD)); cbuf2 = c.take<unsigned char>(bytes((size_t)B * L1_KPAD));
      h_pre = c.take<unsigned char>(bytes(HR * B * L1_OUT)); h_drop = c.take<unsigned char>(bytes(HR * B * L1_OUT));
      g_h = c.take<unsigned char>(bytes(HR * B * L1_OUT)); g_hd = c.take<unsigned char>(bytes(HR * B * L1_OUT));
      gh_b1 = c.take<unsigned char>(bytes((size_t)B * L1_OUT)); ughd = c.take<unsigned char>(bytes((size_t)B * L1_OUT));
      gc = c.take<unsigned char>(bytes(HR * B * L1_KPAD));
      pre_e = c.take<float>(HR * B * EMB); l1_slab = c.take<float>((size_t)L1_KSPLIT * HR * B * L1_OUT);
      gw1p = c.take<float>((size_t)L1_OUT * L1_KPAD); rs = c.take<float>(B); prob_gp = c.take<float>(B);
      for (int i = 0; i < TT.NS; ++i) {
        const size_t n = (size_t)B * (TT.D_HB[i] / 2) * (TT.D_HB[i] / 2) * TT.D_CS[i];
        d_v[i] = c.take<unsigned char>(bytes(n)); d_xdir[i] = c.take<unsigned char>(bytes(n));
        bn2_ws[i] = c.take<float>(jck_bn2_ws_floats(TT.D_CS[i]));
      }
      bn2_ws_rev = c.take<float>(jck_bn2_ws_floats(512));
      d_u0 = c.take<unsigned char>(bytes(img));
    }
    ws_bytes = c.off;
  }

  float* P(const NetLayout& L, float* arena, const char* name) const { return arena + find(L, name)->offset; }
  const unsigned* rng() const { return reinterpret_cast<const unsigned*>(hp2 + 8 * parity + 4); }     // this step's Philox words
  // fake = 0.9 * G(z) + 0.1 * N(0,1) (:171): uploaded noise, or drawn in the kernel when the step carries none
  int mix_fake_noise(const jck_step_inputs* in, int B, hipStream_t st) {
    if (!in->noise_fake) return jck_axpy_noise_rng(prec, fake_raw, rng(), 1, 0.9f, 0.1f, fake, B, T.HW, st);
    return jck_axpy_noise(prec, fake_raw, in->noise_fake, 0.9f, 0.1f, fake, B, T.HW, st);
  }
  // ... and the penalty's interpolate (:111-113) in the same launch
  int mix_fake_noise_interp(const jck_step_inputs* in, int B, hipStream_t st) {
    return jck_mix_interp(prec, fake_raw, in->noise_fake, in->noise_fake ? nullptr : rng(), 1, 0.9f, 0.1f, fake, real_noisy, in->alpha, xhat,
                          B, T.HW, st);
  }
};


extern "C" int jck_engine_create(jck_engine** out, int family, int prec, int batch) {
  return jck_engine_create_sized(out, family, prec, batch, 64);
}
// image_size 64 = the reference's topology; 128 = one more stride-2 stage (DCGAN only; BASELINE.json configs[4])
extern "C" int jck_engine_create_sized(jck_engine** out, int family, int prec, int batch, int image_size) {
  if (!out) JCK_FAIL(JCK_E_ARG, "null out");
  if (image_size != 64 && image_size != 128) JCK_FAIL(JCK_E_ARG, "image_size must be 64 or 128");
  if (image_size != 64 && family != 0) JCK_FAIL(JCK_E_ARG, "the 128x128 topology exists for DCGAN only (CGAN's Linear(8392,256) fixes 64x64)");
  if (family != 0 && family != 1) JCK_FAIL(JCK_E_ARG, "family must be 0 (DCGAN) or 1 (CGAN)");
  if (prec != JCK_PREC_BF16 && prec != JCK_PREC_F32) JCK_FAIL(JCK_E_ARG, "bad prec");
  if (batch < 1 || batch > 8192) JCK_FAIL(JCK_E_ARG, "batch must be in [1, 8192]");
  jck_engine* e = new jck_engine();
  e->family = family; e->prec = prec; e->B = batch; e->esz = prec == JCK_PREC_BF16 ? 2 : 4;
  e->T = make_topo(image_size);
  e->LG = make_layout(family, 0, image_size); e->LD = make_layout(family, 1, image_size);
  e->overlap = !(getenv("JCK_OVERLAP") && atoi(getenv("JCK_OVERLAP")) == 0);
  // batched D passes need whole tiles per group: 16*B rows at the last layer, tiles of up to 128 rows
  e->batched = getenv("JCK_BATCHED") ? atoi(getenv("JCK_BATCHED")) : JCK_BATCHED_DEFAULT;
  if (!e->overlap || batch % 8 != 0 || e->batched != 3) e->batched = 0;
  e->carve(nullptr);
  e->ext_events = !(getenv("JCK_EXT_EVENTS") && atoi(getenv("JCK_EXT_EVENTS")) == 0);
  e->bn_res = !(getenv("JCK_BN_RES") && atoi(getenv("JCK_BN_RES")) == 0);
  e->cbuf_direct = family == 1 && !(getenv("JCK_CBUF_DIRECT") && atoi(getenv("JCK_CBUF_DIRECT")) == 0);
  e->head_fuse = !(getenv("JCK_HEAD_FUSE") && atoi(getenv("JCK_HEAD_FUSE")) == 0);
  e->real_side = !(getenv("JCK_REAL_SIDE") && atoi(getenv("JCK_REAL_SIDE")) == 0);
  e->fold_zero = !(getenv("JCK_FOLD_ZERO") && atoi(getenv("JCK_FOLD_ZERO")) == 0);
  e->fuse_tanh = !(getenv("JCK_FUSE_TANH") && atoi(getenv("JCK_FUSE_TANH")) == 0);
  if (e->overlap) {
    hipStream_t* ss[3] = {&e->sA, &e->sB, &e->sC};
    for (auto pp : ss) HIPCHK(hipStreamCreateWithFlags(pp, hipStreamNonBlocking));      // queue priorities measured neutral (r1-r4)
    hipEvent_t* ev[8 + JCK_MAX_STAGES] = {&e->evWdone, &e->evWmid, &e->evHead, &e->evTail, &e->ev0, &e->evF, &e->evReal, &e->evGP};
    for (int i = 0; i < JCK_MAX_STAGES; ++i) ev[8 + i] = &e->evW[i];
    for (auto p : ev) HIPCHK(hipEventCreateWithFlags(p, jck_event_flags()));
  }
  *out = e;
  return JCK_OK;
}
extern "C" void jck_engine_destroy(jck_engine* e) {
  if (!e) return;
  if (e->overlap) {
    hipStream_t ss[3] = {e->sA, e->sB, e->sC};
    for (auto p : ss) if (p) { (void)hipStreamSynchronize(p); (void)hipStreamDestroy(p); }
    hipEvent_t ev[8 + JCK_MAX_STAGES] = {e->evWdone, e->evWmid, e->evHead, e->evTail, e->ev0, e->evF, e->evReal, e->evGP};
    for (int i = 0; i < JCK_MAX_STAGES; ++i) ev[8 + i] = e->evW[i];
    for (auto p : ev) if (p) (void)hipEventDestroy(p);
  }
  delete e;
}
extern "C" int jck_engine_num_tensors(int family, int net) { return (int)make_layout(family, net).t.size(); }
extern "C" int jck_engine_tensor_info(int family, int net, int idx, char* name, int name_cap, int* kind, long long* offset,
                                      long long* numel, int* shape4) {
  NetLayout L = make_layout(family, net);
  if (idx < 0 || idx >= (int)L.t.size()) JCK_FAIL(JCK_E_ARG, "index out of range");
  const TensorInfo& t = L.t[idx];
  if (name && name_cap > 0) { strncpy(name, t.name, name_cap - 1); name[name_cap - 1] = 0; }
  if (kind) *kind = t.kind;
  if (offset) *offset = t.offset;
  if (numel) *numel = t.numel;
  if (shape4) for (int q_ = 0; q_ < 4; ++q_) shape4[q_] = t.shape[q_];
  return JCK_OK;
}
extern "C" long long jck_engine_arena_numel(int family, int net, int which) {
  NetLayout L = make_layout(family, net);
  return which == 0 ? L.n_params : L.n_bn;
}
// the same three queries for a created engine (its own image size)
extern "C" int jck_engine_num_tensors_of(const jck_engine* e, int net) { return e ? (int)(net == 0 ? e->LG : e->LD).t.size() : 0; }
extern "C" int jck_engine_tensor_info_of(const jck_engine* e, int net, int idx, char* name, int name_cap, int* kind, long long* offset,
                                         long long* numel, int* shape4) {
  if (!e) JCK_FAIL(JCK_E_ARG, "null engine");
  const NetLayout& L = net == 0 ? e->LG : e->LD;
  if (idx < 0 || idx >= (int)L.t.size()) JCK_FAIL(JCK_E_ARG, "index out of range");
  const TensorInfo& t = L.t[idx];
  if (name && name_cap > 0) { strncpy(name, t.name, name_cap - 1); name[name_cap - 1] = 0; }
  if (kind) *kind = t.kind;
  if (offset) *offset = t.offset;
  if (numel) *numel = t.numel;
  if (shape4) for (int q_ = 0; q_ < 4; ++q_) shape4[q_] = t.shape[q_];
  return JCK_OK;
}
extern "C" long long jck_engine_arena_numel_of(const jck_engine* e, int net, int which) {
  if (!e) return 0;
  const NetLayout& L = net == 0 ? e->LG : e->LD;
  return which == 0 ? L.n_params : L.n_bn;
}
extern "C" int jck_engine_image_size(const jck_engine* e) { return e ? e->T.S : 0; }
extern "C" size_t jck_engine_workspace_bytes(const jck_engine* e) { return e ? e->ws_bytes : 0; }

extern "C" int jck_engine_bind(jck_engine* e, void* workspace, size_t ws_bytes, float* g_params, float* g_grads, float* g_m,
                               float* g_v, float* g_bn, int64_t* g_nbt, float* d_params, float* d_grads, float* d_m,
                               float* d_v, float* d_bn, int64_t* d_nbt) {
  if (!e || !workspace) JCK_FAIL(JCK_E_ARG, "null engine / workspace");
  if (ws_bytes < e->ws_bytes) JCK_FAIL(JCK_E_WS, "workspace too small: need " + std::to_string(e->ws_bytes));
  if (((uintptr_t)workspace) % 256) JCK_FAIL(JCK_E_ARG, "workspace must be 256-byte aligned");
  e->carve(reinterpret_cast<unsigned char*>(workspace));
  // the repack kernel writes only real (channel, channel) pairs: the padding rows / channels of the packed operands must
  // read as zero -> clear that region once (from the first packed operand up to the first activation buffer)
  {
    unsigned char* p0 = reinterpret_cast<unsigned char*>(e->d_down[0]);
    unsigned char* p1 = reinterpret_cast<unsigned char*>(e->dset[0].y[0]);
    HIPCHK(hipMemset(p0, 0, (size_t)(p1 - p0)));
  }
  HIPCHK(hipMemset(e->gsync, 0, jck_grid_sync_bytes()));
  e->gp = g_params; e->gg = g_grads; e->gm = g_m; e->gv = g_v; e->gbn = g_bn; e->gnbt = g_nbt;
  e->dp = d_params; e->dg = d_grads; e->dm = d_m; e->dv = d_v; e->dbn = d_bn; e->dnbt = d_nbt;
  e->bound = true;
  return JCK_OK;
}

// all conv operands of one network in one launch (ew.hpp: pack_multi_kernel); the two Linear operands of CGAN's D follow
// tail (optional): the end-of-step job rides in the same launch (ew.hpp: pack_tail_kernel)
static int repack_convs(jck_engine* e, int net, void* stream, const TailJobs* tail = nullptr, int tail_x = 0) {
  PackJobs jobs = {};
  int n = 0, chunk = 0;
  auto add = [&](int kind, const float* w, void* wp, long long total, int a, int b, int c) {
    PackJob& J = jobs.j[n];
    J.w = w; J.wp = wp; J.total = total; J.kind = kind; J.a = a; J.b = b; J.c = c;
    jobs.first_chunk[n] = chunk;
    chunk += (int)((total + PACK_CHUNK - 1) / PACK_CHUNK);
    ++n;
  };
  // totals are (output channel, input channel) pairs: one thread each (ew.hpp: pack_multi_kernel)
  auto add_down = [&](const float* w, int Cs, int Cb, void* wp) { add(0, w, wp, (long long)Cs * Cb, Cs, Cb, ilog2(jck_pad_chan(Cb))); };
  auto add_up = [&](const float* w, int Cs, int Cb, void* wp) {
    if (Cb <= 4) add(2, w, wp, (long long)Cs * Cb, Cs, Cb, 0);
    else add(1, w, wp, (long long)Cs * Cb, Cs, Cb, jck_pad_rows(Cb));
  };
  if (net == 1) {
    for (int i = 0; i < TT.NS; ++i) {
      const float* w = e->P(e->LD, e->dp, CWN[i]);
      add_down(w, TT.D_CS[i], TT.D_CB[i], e->d_down[i]);
      add_up(w, TT.D_CS[i], TT.D_CB[i], e->d_up[i]);
    }
    if (e->family == 0) add(4, e->P(e->LD, e->dp, CWN[TT.NS]), e->d_head_wp, TT.G_C1, TT.G_C1, 0, 0);
  } else {
    add(3, e->P(e->LG, e->gp, CWN[0]), e->g1_w, (long long)z_dim(e->family) * TT.G_C1, z_dim(e->family), TT.G_C1, z_pad(e->family));
    for (int i = 0; i < TT.NS; ++i) {
      const float* w = e->P(e->LG, e->gp, CWN[i + 1]);
      add_up(w, TT.G_CS[i], TT.G_CB[i], e->g_up[i]);
      add_down(w, TT.G_CS[i], TT.G_CB[i], e->g_down[i]);
    }
  }
  jobs.first_chunk[n] = chunk;
  jobs.n = n;
  if (tail) {
    const int nblk = chunk + tail_x * (tail->nl + 1);
    if (e->prec == JCK_PREC_BF16) hipLaunchKernelGGL(pack_tail_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, jobs, chunk, *tail, tail_x);
    else hipLaunchKernelGGL(pack_tail_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, jobs, chunk, *tail, tail_x);
  } else if (e->prec == JCK_PREC_BF16) hipLaunchKernelGGL(pack_multi_kernel<bf16_t>, dim3(chunk), dim3(256), 0, (hipStream_t)stream, jobs);
  else hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(chunk), dim3(256), 0, (hipStream_t)stream, jobs);
  HIPCHK(hipGetLastError());
  return JCK_OK;
}
static int repack_linear(jck_engine* e, void* stream) {
  const float* w1 = e->P(e->LD, e->dp, "linear1.weight");
  return pack_linear_pair(e->prec, w1, L1_OUT, L1_K, L1_OUT, L1_KPAD, e->l1_w, L1_KPAD, L1_OUT, e->l1_wT, 512, 16, (hipStream_t)stream);
}
extern "C" int jck_engine_repack(jck_engine* e, int net, void* stream) {
  if (!e || !e->bound) JCK_FAIL(JCK_E_ARG, "engine not bound");
  JCK_TRY(repack_convs(e, net, stream));
  if (net == 1 && e->family == 1) JCK_TRY(repack_linear(e, stream));
  return JCK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// building blocks
// ---------------------------------------------------------------------------------------------------------
static const float BN_MOM = 0.1f, BN_EPS = 1e-5f, LRELU = 0.2f;


typedef jck_engine::DSet DSet;

// conv stack of D on activation set `D`; BatchNorm running statistics are NOT touched here: (mean, unbiased var) go to the
// deferred record of `pass` (0 real, 1 fake, 2 penalty, 3 G phase) and are applied in that order at the end of the step,
// which keeps the result bitwise independent of how the passes overlap on streams.
// to_cbuf (CGAN): the last layer's activation goes to rows [0, B) of the head's concat buffer instead of D.a (cg_head_forward then
// has nothing to copy)
static int d_convs_forward(jck_engine* e, DSet& D, const void* x_in, int B, int pass, hipStream_t st, bool to_cbuf = false) {
  const void* in = x_in;
  for (int i = 0; i < TT.NS; ++i) {
    const int hb = TT.D_HB[i], cs = TT.D_CS[i];
    const long long rows = (long long)B * (hb / 2) * (hb / 2);
    JCK_TRY(jck_conv_down_grouped(e->prec, in, e->d_down[i], D.y[i], D.bn[i].stats, &D.bn[i].slots, B, hb, hb, TT.D_CB[i], cs, B, st));
    const bool cb = to_cbuf && i == TT.NS - 1;
    bool fused = false;                           // finalize + apply as one launch where the statistics rows are few (ops.hip)
    JCK_TRY(bn_fwd_fused(e->prec, D.y[i], D.bn[i].stats, D.bn[i].slots, (float)rows, e->P(e->LD, e->dp, NWN[i]), e->P(e->LD, e->dp, NBN[i]), BN_EPS,
                         LRELU, cb ? e->cbuf : D.a[i], D.bn[i].aux, e->d_rs[i] + ((size_t)e->parity * 4 + pass) * 2 * cs, nullptr, nullptr, nullptr,
                         BN_MOM, rows, cs, 1, cb ? TT.FEAT : 0, cb ? L1_KPAD : 0, st, &fused));
    if (!fused) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cs / 4), dim3(256), 0, st, D.bn[i].stats, D.bn[i].slots, (float)rows,
                       e->P(e->LD, e->dp, NWN[i]), e->P(e->LD, e->dp, NBN[i]), (float*)nullptr, (float*)nullptr,
                       (long long*)nullptr, BN_MOM, BN_EPS, D.bn[i].aux, cs, e->d_rs[i] + ((size_t)e->parity * 4 + pass) * 2 * cs);
    HIPCHK(hipGetLastError());
    if (cb) JCK_TRY(bn_act_fwd_pitched(e->prec, D.y[i], D.bn[i].aux, LRELU, e->cbuf, rows, cs, 1, TT.FEAT, L1_KPAD, st));
    else JCK_TRY(jck_bn_act_fwd(e->prec, D.y[i], D.bn[i].aux, LRELU, D.a[i], rows, cs, st));
    }
    in = D.a[i];
  }
  return JCK_OK;
}

// D forward up to (not including) the sigmoid head.  family 1: concat + Linear(8392,256) + Dropout (model/CGAN.py:117-122)
// CGAN head up to the dropped-out hidden layer from the conv features a4 [B][8192] (model/CGAN.py:111-121): concat with the
// label embedding, Linear(8392,256), Dropout.  Leaves cbuf, pre_e, h_pre, h_drop for the matching d_head_backward.
// rows: B, or 3B for the batched head (label_period = B: the three groups share the batch's labels; drop_mask = their three
// [B][256] masks back to back)
// finish = false: stops at the split-K slabs of Linear(8392,256); cg_head_mid (one launch) takes it from there
static int cg_head_forward(jck_engine* e, const void* a4, int rows, const float* drop_mask, hipStream_t st, int label_period = 0, bool finish = true) {
  if (!e->cur_labels || !drop_mask) JCK_FAIL(JCK_E_ARG, "CGAN pass needs labels and a dropout mask");
  if (a4) JCK_TRY(jck_concat_rows(e->prec, a4, TT.FEAT, e->cbuf, L1_KPAD, rows, st));      // null: the conv stack wrote the rows itself
  JCK_TRY(jck_label_embed_fwd_tiled(e->prec, e->cur_labels, e->P(e->LD, e->dp, "label_embedding.weight"),
                                    e->P(e->LD, e->dp, "label_embedding.bias"), LRELU, rows, N_CLASS, EMB, e->cbuf, L1_KPAD, TT.FEAT,
                                    e->pre_e, label_period, st));
  JCK_TRY(jck_linear_fwd(e->prec, e->cbuf, e->l1_w, nullptr, e->l1_slab, rows, L1_KPAD, L1_OUT, L1_OUT, L1_KSPLIT, st));
  if (!finish) return JCK_OK;
  return jck_linear_finish(e->prec, e->l1_slab, L1_KSPLIT, e->P(e->LD, e->dp, "linear1.bias"), drop_mask, 1.0f / 0.75f, e->h_pre,
                           e->h_drop, rows, L1_OUT, st);
}
// The head from the split-K slabs to the gradient at the Linear(8392,256) output, rows [0, G * B) of the head buffers: h_pre, h_drop,
// prob / ds / scalar slots per group, g_hd, g_h (ops.hip: cg_head_mid)
static int cg_head_middle(jck_engine* e, int B, int G, const float* targets, const int* modes, const int* slot_loss, const int* slot_p,
                          float* prob, float* ds, const float* drop_mask, hipStream_t st) {
  return cg_head_mid(e->prec, e->l1_slab, L1_KSPLIT, e->P(e->LD, e->dp, "linear1.bias"), drop_mask, 1.0f / 0.75f, e->h_pre, e->h_drop,
                     e->P(e->LD, e->dp, "linear2.weight"), e->P(e->LD, e->dp, "linear2.bias"), B, G, targets, modes, prob, ds, e->acc, slot_loss,
                     slot_p, e->acc_ld, e->g_hd, e->g_h, st);
}

static int d_forward(jck_engine* e, DSet& D, const void* x_in, int B, int pass, const float* drop_mask, hipStream_t st, bool head_finish = true) {
  const bool direct = e->family == 1 && e->cbuf_direct;
  JCK_TRY(d_convs_forward(e, D, x_in, B, pass, st, direct));
  if (e->family == 1) return cg_head_forward(e, direct ? nullptr : e->d_a[TT.NS - 1], B, drop_mask, st, 0, head_finish);
  return JCK_OK;
}

// sigmoid head + loss (mode 0) or + d(sum p)/dlogit (mode 1); fills e->prob / e->ds
// g_out (family 0, optional): conv5's input gradient ds[n] * w from the same launch
static int d_head(jck_engine* e, DSet& D, int B, float target, int mode, int slot_loss, int slot_p, hipStream_t st, void* g_out = nullptr) {
  if (e->family == 0)
    return head_fwd_grouped_ev(e->prec, D.a[TT.NS - 1], e->d_head_wp, nullptr, B, TT.FEAT, 1, &target, &mode, D.prob, D.ds, e->acc, &slot_loss,
                               &slot_p, e->acc_ld, g_out, st, nullptr);
  return jck_head_fwd(e->prec, e->h_drop, e->P(e->LD, e->dp, "linear2.weight"), e->P(e->LD, e->dp, "linear2.bias"), B, L1_OUT, target,
                      mode, D.prob, D.ds, e->acc, slot_loss, slot_p, e->acc_ld, st);
}

// head backward from ds (device float[B]) down to the gradient w.r.t. a4 in e->d_g[3] (or `ga4_out`).
// side (family 1, optional): the parameter gradients run there (see cg_head_backward_batched)
// mid_done: g_hd and g_h are in place (cg_head_middle)
static int d_head_backward(jck_engine* e, DSet& D, const float* ds, int B, bool want_wgrad, const float* drop_mask, void* ga4_out, hipStream_t st,
                           hipStream_t side = nullptr, bool mid_done = false) {
  if (e->family == 0)       // mid_done: the input gradient is in place (d_head with g_out)
    return jck_head_bwd_conv(e->prec, ds, e->d_head_wp, D.a[TT.NS - 1], B, TT.G_C1, mid_done ? nullptr : ga4_out,
                             want_wgrad ? e->P(e->LD, e->dg, CWN[TT.NS]) : nullptr, e->head_ws, st);
  // linear2 + sigmoid: g_hd = ds * w2, dW2 += sum ds * h_drop, db2 += sum ds.  Head buffers from row e->head_row0 on (the
  // penalty group of a batched head sits at rows [2B, 3B))
  const size_t r0 = (size_t)e->head_row0, w0 = (size_t)e->head_wrow0;
  auto hb = [&](void* p, size_t row, size_t per_row) { return (void*)((unsigned char*)p + row * per_row * e->esz); };
  void *h_drop = hb(e->h_drop, r0, L1_OUT), *cbuf = hb(e->cbuf, r0, L1_KPAD);
  void *g_hd = hb(e->g_hd, w0, L1_OUT), *g_h = hb(e->g_h, w0, L1_OUT), *gc = hb(e->gc, w0, L1_KPAD);
  const float* pre_e = e->pre_e + r0 * EMB;
  hipStream_t ws = st;
  auto fork = [&](int k) -> int {                     // `ws` work enqueued from here on starts behind everything on st so far
    if (!side || !want_wgrad) return JCK_OK;
    HIPCHK(hipEventRecord(e->evW[k], st));
    HIPCHK(hipStreamWaitEvent(side, e->evW[k], 0));
    ws = side;
    return JCK_OK;
  };
  const float* w2 = e->P(e->LD, e->dp, "linear2.weight");
  if (!mid_done) JCK_TRY(jck_head_bwd(e->prec, ds, w2, h_drop, B, L1_OUT, g_hd, nullptr, 1, e->head_ws, st));
  if (want_wgrad) {
    JCK_TRY(fork(0));
    JCK_TRY(jck_head_bwd(e->prec, ds, w2, h_drop, B, L1_OUT, nullptr, e->P(e->LD, e->dg, "linear2.weight"), 1, e->head_ws, ws));
    JCK_TRY(jck_sum_vec(ds, B, e->P(e->LD, e->dg, "linear2.bias"), ws));
  }
  if (!mid_done) JCK_TRY(jck_dropout(e->prec, g_hd, drop_mask, 1.0f / 0.75f, g_h, (long long)B * L1_OUT, st));
  if (want_wgrad) {
    JCK_TRY(fork(1));
    JCK_TRY(jck_linear_wgrad(e->prec, g_h, L1_OUT, cbuf, L1_KPAD, e->wg_ws, e->wg_ws_bytes, e->gw1p, 1, B, L1_OUT, ws));
    JCK_TRY(jck_colsum(e->prec, g_h, B, L1_OUT, L1_OUT, e->P(e->LD, e->dg, "linear1.bias"), ws));
  }
  JCK_TRY(jck_linear_fwd(e->prec, g_h, e->l1_wT, nullptr, gc, B, L1_OUT, L1_KPAD, L1_KPAD, 1, st));
  if (want_wgrad) {
    JCK_TRY(fork(2));
    JCK_TRY(jck_label_embed_bwd(e->prec, gc, L1_KPAD, TT.FEAT, pre_e, e->cur_labels, LRELU, B, N_CLASS, EMB,
                                e->P(e->LD, e->dg, "label_embedding.weight"), e->P(e->LD, e->dg, "label_embedding.bias"), ws));
  }
  return jck_split_rows(e->prec, gc, L1_KPAD, TT.FEAT, ga4_out, B, st);
}

// The head backward of the real | fake | penalty groups as ONE pass over 3B rows (forward: cg_head_forward(rows = 3B)): input
// gradients for all three, parameter gradients from the first two (the penalty's come from its double backward, PHASE_D_GP).
// 11 launches instead of 26; the penalty group's rows [2B, 3B) stay behind as gp_double_backward expects them.
// side (optional): the parameter gradients of the head (linear2, linear1, the label embedding: 8 launches, ~80 us that only the
// optimiser waits for) run on the weight-gradient stream - idle at this point of the step - beside the chain that carries the input
// gradient down to the conv stack; evHead marks their end on that stream (PHASE_D_GP's main-stream writers of the same gradients
// wait for it).
static int cg_head_backward_batched(jck_engine* e, const float* ds, int B, const float* drop_mask, void* ga4_out, hipStream_t st,
                                    hipStream_t side = nullptr, bool mid_done = false) {
  const int R3 = 3 * B, R2 = 2 * B;
  const float* w2 = e->P(e->LD, e->dp, "linear2.weight");
  hipStream_t ws = st;
  auto fork = [&](int k) -> int {                     // `ws` work enqueued from here on starts behind everything on st so far
    if (!side) return JCK_OK;
    HIPCHK(hipEventRecord(e->evW[k], st));
    HIPCHK(hipStreamWaitEvent(side, e->evW[k], 0));
    ws = side;
    return JCK_OK;
  };
  if (!mid_done) JCK_TRY(jck_head_bwd(e->prec, ds, w2, e->h_drop, R3, L1_OUT, e->g_hd, nullptr, 1, e->head_ws, st));
  if (!mid_done) {        // (with the middle in one launch both forks start at the same point: one hand-over)
    JCK_TRY(fork(0));
    JCK_TRY(jck_head_bwd(e->prec, ds, w2, e->h_drop, R2, L1_OUT, nullptr, e->P(e->LD, e->dg, "linear2.weight"), 1, e->head_ws, ws));
    JCK_TRY(jck_sum_vec(ds, R2, e->P(e->LD, e->dg, "linear2.bias"), ws));
    JCK_TRY(jck_dropout(e->prec, e->g_hd, drop_mask, 1.0f / 0.75f, e->g_h, (long long)R3 * L1_OUT, st));
  }
  JCK_TRY(fork(1));
  if (mid_done) {
    JCK_TRY(jck_head_bwd(e->prec, ds, w2, e->h_drop, R2, L1_OUT, nullptr, e->P(e->LD, e->dg, "linear2.weight"), 1, e->head_ws, ws));
    JCK_TRY(jck_sum_vec(ds, R2, e->P(e->LD, e->dg, "linear2.bias"), ws));
  }
  JCK_TRY(jck_linear_wgrad(e->prec, e->g_h, L1_OUT, e->cbuf, L1_KPAD, e->wg_ws, e->wg_ws_bytes, e->gw1p, 1, R2, L1_OUT, ws));
  JCK_TRY(jck_colsum(e->prec, e->g_h, R2, L1_OUT, L1_OUT, e->P(e->LD, e->dg, "linear1.bias"), ws));
  JCK_TRY(jck_linear_fwd(e->prec, e->g_h, e->l1_wT, nullptr, e->gc, R3, L1_OUT, L1_KPAD, L1_KPAD, 1, st));
  JCK_TRY(fork(2));
  JCK_TRY(jck_label_embed_bwd_tiled(e->prec, e->gc, L1_KPAD, TT.FEAT, e->pre_e, e->cur_labels, LRELU, R2, N_CLASS, EMB,
                                    e->P(e->LD, e->dg, "label_embedding.weight"), e->P(e->LD, e->dg, "label_embedding.bias"), B, ws));
  if (side) HIPCHK(hipEventRecord(e->evHead, side));
  return jck_split_rows(e->prec, e->gc, L1_KPAD, TT.FEAT, ga4_out, R3, st);
}

// D backward on set `D`.  With `side` != nullptr the weight-gradient products run on that stream beside the dgrad chain
// (both only READ gy_i and the saved activations); the main stream waits for them before returning.
// resident = false: the pass runs beside another pass of the step on a second stream.  Two resident BatchNorm launches in flight
// at once would share the engine's barrier words and could each hold CUs the other's missing workgroups need: such a pass takes
// the three-launch form.
// xe (G's loss pass): the input gradient is wanted only as the operand of the tanh + noise-mix backward of G's output - that
// product rides in the epilogue of the launch that computes it (conv_up_tanh_bwd_ev) when the layer runs on the image-side kernel
struct XgradEpi { const void* tanh_y; float scale; void* out; hipEvent_t done; bool fused; };
static int d_backward(jck_engine* e, DSet& D, const void* x_in, int B, bool want_wgrad, bool want_xgrad, const float* drop_mask,
                      hipStream_t st, hipStream_t side, bool join = true, bool resident = true, XgradEpi* xe = nullptr, bool head_mid_done = false) {
  JCK_TRY(d_head_backward(e, D, D.ds, B, want_wgrad, drop_mask, D.g[TT.NS - 1], st, nullptr, head_mid_done));
  const bool par = want_wgrad && side != nullptr;
  for (int i = TT.NS - 1; i >= 0; --i) {
    const int hb = TT.D_HB[i], cs = TT.D_CS[i], cb = TT.D_CB[i];
    const long long rows = (long long)B * (hb / 2) * (hb / 2);
    float* dgam = want_wgrad ? e->P(e->LD, e->dg, NWN[i]) : nullptr;
    float* dbet = want_wgrad ? e->P(e->LD, e->dg, NBN[i]) : nullptr;
    // the launch that writes g_y hands it to the weight-gradient stream by completing evW[i] itself
    hipEvent_t done = par && e->ext_events ? e->evW[i] : nullptr;
    JCK_TRY(bn_act_bwd_res_ev(e->prec, D.g[i], D.y[i], D.bn[i].aux, LRELU, D.bn[i].sums, D.g[i], dgam, dbet, rows, cs, 1, 1,
                              e->bn_res && resident ? e->gsync : nullptr, st, done));
    const void* big = i == 0 ? x_in : D.a[i - 1];
    if (want_wgrad) {
      hipStream_t ws = st;
      if (par) { if (!done) HIPCHK(hipEventRecord(e->evW[i], st)); HIPCHK(hipStreamWaitEvent(side, e->evW[i], 0)); ws = side; }
      JCK_TRY(jck_conv_wgrad(e->prec, D.g[i], big, e->wg_ws, e->wg_ws_bytes, e->P(e->LD, e->dg, CWN[i]), 1, B, hb, hb, cb, cs, ws));
    }
    if (i > 0)
      JCK_TRY(jck_conv_up(e->prec, D.g[i], e->d_up[i], D.g[i - 1], nullptr, nullptr, 0, B, hb / 2, hb / 2, cs, cb, st));
    else if (want_xgrad) {
      if (xe) JCK_TRY(conv_up_tanh_bwd_ev(e->prec, D.g[0], e->d_up[0], xe->tanh_y, xe->scale, xe->out, B, hb / 2, hb / 2, cs, cb, st, xe->done, &xe->fused));
      if (!xe || !xe->fused)
        JCK_TRY(jck_conv_up(e->prec, D.g[0], e->d_up[0], D.gx, nullptr, nullptr, 0, B, hb / 2, hb / 2, cs, cb, st));
    }
  }
  // join = false: the caller's NEXT d_backward/g_backward with a side stream (or its own join) orders the main stream
  // behind these weight gradients - nothing on the main stream reads them before the optimiser step
  if (par && join) { HIPCHK(hipEventRecord(e->evWdone, side)); HIPCHK(hipStreamWaitEvent(st, e->evWdone, 0)); }
  return JCK_OK;
}

// D forward + backward on G batches stored back to back from `x_in` (real_noisy | fake | xhat are consecutive): ONE launch
// per layer and direction instead of G, BatchNorm statistics per group (= per batch, as in the separate passes,
// train/dcgan_trainer.py:162,173,118).  Groups [0, G-1) are loss passes (targets / accumulator slots given per group) and
// contribute weight gradients; the LAST group is the penalty pass (head mode 1, gradient w.r.t. its input image -> dset[0].gx,
// norms -> dset[0].norms).  Group g writes BatchNorm record pass0 + g.  Weight gradients run on `side`.
// forward of the conv stack for groups [g0, g0 + n) of the batched set (x_in = first image of group g0); own slot range
static int d_batched_forward(jck_engine* e, const void* x_in, int B, int g0, int n, int pass0, hipStream_t st, bool to_cbuf = false) {
  auto& S = e->bset;
  const size_t esz = e->esz;
  auto at = [&](void* p, size_t elems) { return (void*)((unsigned char*)p + elems * esz); };
  const void* in = x_in;
  for (int i = 0; i < TT.NS; ++i) {
    const int hb = TT.D_HB[i], cs = TT.D_CS[i];
    const long long rows = (long long)B * (hb / 2) * (hb / 2);
    // statistic slots of group g0 start at the g0/3 point of the buffer (sized for 3B pixels at one slot per 32 pixels)
    float* stats = S.stats[i] + (size_t)g0 * (jck_stats_floats((long long)3 * B * (hb / 2) * (hb / 2), cs, 1) / 3 / (2 * cs)) * (2 * cs);
    int slots = 0;
    JCK_TRY(jck_conv_down_grouped(e->prec, in, e->d_down[i], at(S.y[i], (size_t)g0 * rows * cs), stats, &slots, n * B, hb, hb, TT.D_CB[i], cs, B, st));
    if (slots % n) JCK_FAIL(JCK_E_ARG, "batched D pass: statistic slots do not split by group");
    const bool cb = to_cbuf && i == TT.NS - 1;      // CGAN: rows [g0 * B, (g0 + n) * B) of the head's concat buffer
    bool fused = false;
    JCK_TRY(bn_fwd_fused(e->prec, at(S.y[i], (size_t)g0 * rows * cs), stats, slots / n, (float)rows, e->P(e->LD, e->dp, NWN[i]), e->P(e->LD, e->dp, NBN[i]),
                         BN_EPS, LRELU, cb ? at(e->cbuf, (size_t)g0 * B * L1_KPAD) : at(S.a[i], (size_t)g0 * rows * cs), S.aux[i] + (size_t)g0 * 4 * cs,
                         e->d_rs[i] + ((size_t)e->parity * 4 + pass0 + g0) * 2 * cs, nullptr, nullptr, nullptr, BN_MOM, rows, cs, n, cb ? TT.FEAT : 0,
                         cb ? L1_KPAD : 0, st, &fused));
    if (!fused) {
    JCK_TRY(jck_bn_finalize_grouped(stats, slots / n, (float)rows, e->P(e->LD, e->dp, NWN[i]), e->P(e->LD, e->dp, NBN[i]),
                                    BN_EPS, S.aux[i] + (size_t)g0 * 4 * cs, e->d_rs[i] + ((size_t)e->parity * 4 + pass0 + g0) * 2 * cs, cs, n, st));
    if (cb)
      JCK_TRY(bn_act_fwd_pitched(e->prec, at(S.y[i], (size_t)g0 * rows * cs), S.aux[i] + (size_t)g0 * 4 * cs, LRELU,
                                 at(e->cbuf, (size_t)g0 * B * L1_KPAD), rows, cs, n, TT.FEAT, L1_KPAD, st));
    else
      JCK_TRY(jck_bn_act_fwd_grouped(e->prec, at(S.y[i], (size_t)g0 * rows * cs), S.aux[i] + (size_t)g0 * 4 * cs, LRELU,
                                     at(S.a[i], (size_t)g0 * rows * cs), rows, cs, n, st));
    }
    in = at(S.a[i], (size_t)g0 * rows * cs);
  }
  return JCK_OK;
}

// Backward of the conv stack for G groups whose gradients w.r.t. a4 are in bset.g[3]: BatchNorm backward grouped, ONE dgrad
// launch per layer over all groups, ONE weight-gradient launch per layer over the first gw groups (on `side` when given),
// and - xgrad_last - the gradient w.r.t. the input image of the LAST group -> dset[0].gx.  Joins `side` before returning.
// part 0: the whole pass.  part 1 / 2 (data parallel, PHASE_D_LOSS_A / _B): up to and including the weight gradient of the LAST
// layer (joined into `st`: the tail of D's gradient arena - conv4.weight, norm4.*, conv5.weight - is then final in `st` order and
// its all-reduce can start) / everything after it.
static int d_batched_backward(jck_engine* e, const void* x_in, int B, int G, int gw, bool xgrad_last, hipStream_t st, hipStream_t side,
                              bool with_gp_norm = false, int part = 0, bool lazy_join = false) {
  auto& S = e->bset;
  const size_t esz = e->esz;
  auto at = [&](void* p, size_t elems) { return (void*)((unsigned char*)p + elems * esz); };
  for (int i = TT.NS - 1; i >= 0; --i) {
    const int hb = TT.D_HB[i], cs = TT.D_CS[i], cb = TT.D_CB[i];
    const long long rows = (long long)B * (hb / 2) * (hb / 2);
    const bool resume = part == 2 && i == TT.NS - 1;      // part 2 starts at this layer's dgrad
    // The bottom layer's weight gradient is the last product of the pass and nothing but the penalty's image gradient runs
    // beside it (37 us of the main stream waiting for it in round 3).  Its operand is the gradient of the LOSS groups only:
    // their BatchNorm backward goes first, the weight gradient starts behind it on the second stream, and the penalty
    // group's BatchNorm backward (a launch of its own - in the resident form the groups of this layer go one after the other
    // anyway) runs beside it.
    // (the split does not depend on the stream layout: a captured, one-stream step is the same arithmetic launch for launch)
    const bool split0 = !resume && i == 0 && e->bn_res && e->prec == JCK_PREC_BF16 && gw > 0 && gw < G;
    hipEvent_t done = side && e->ext_events && !resume ? e->evW[i] : nullptr;    // the launch that writes g_y completes evW[i] itself
    if (!resume)
      JCK_TRY(bn_act_bwd_res_ev(e->prec, S.g[i], S.y[i], S.aux[i], LRELU, S.sums[i], S.g[i], e->P(e->LD, e->dg, NWN[i]),
                                e->P(e->LD, e->dg, NBN[i]), rows, cs, split0 ? gw : G, gw, e->bn_res ? e->gsync : nullptr, st, done));
    const void* big = i == 0 ? x_in : S.a[i - 1];
    if (!resume) {
      hipStream_t ws = st;
      if (side) { if (!done) HIPCHK(hipEventRecord(e->evW[i], st)); HIPCHK(hipStreamWaitEvent(side, e->evW[i], 0)); ws = side; }
      JCK_TRY(jck_conv_wgrad(e->prec, S.g[i], big, e->wg_ws, e->wg_ws_bytes, e->P(e->LD, e->dg, CWN[i]), 1, gw * B, hb, hb, cb, cs, ws));
    }
    if (split0)
      JCK_TRY(bn_act_bwd_res_ev(e->prec, at(S.g[i], (size_t)gw * rows * cs), at(S.y[i], (size_t)gw * rows * cs), S.aux[i] + (size_t)gw * 4 * cs, LRELU,
                                S.sums[i] + (size_t)gw * jck_bn_bwd_ws_floats(cs), at(S.g[i], (size_t)gw * rows * cs), nullptr, nullptr, rows, cs,
                                G - gw, 0, e->gsync, st, nullptr));
    if (part == 1) {
      // evTail: the tail of D's gradient arena is final behind it (jck_engine_order_after_tail); lazy_join: st does not wait
      HIPCHK(hipEventRecord(e->evTail, side ? side : st));
      e->tail_on_side = side != nullptr;
      if (side && !lazy_join) HIPCHK(hipStreamWaitEvent(st, e->evTail, 0));
      return JCK_OK;
    }
    if (i > 0)
      JCK_TRY(jck_conv_up(e->prec, S.g[i], e->d_up[i], S.g[i - 1], nullptr, nullptr, 0, G * B, hb / 2, hb / 2, cs, cb, st));
    else if (xgrad_last)
      JCK_TRY(jck_conv_up(e->prec, at(S.g[0], (size_t)(G - 1) * rows * cs), e->d_up[0], e->d_gx, nullptr, nullptr, 0, B, hb / 2, hb / 2, cs, cb, st));
  }
  // the penalty's norm does not need the weight gradients: it runs while the side stream finishes the last of them
  if (with_gp_norm) JCK_TRY(jck_gp_norm(e->prec, e->d_gx, B, TT.HW, e->acc, 6, e->acc_ld, e->norms, st));
  if (side) {
    HIPCHK(hipEventRecord(e->evWdone, side));
    if (lazy_join) { e->join_pending = true; e->mid_recorded = false; }      // JCK_PHASE_LAZY_JOIN: whoever comes next waits
    else HIPCHK(hipStreamWaitEvent(st, e->evWdone, 0));
  }
  return JCK_OK;
}

// Heads + backward of G batches stored back to back from `x_in` (real_noisy | fake | xhat are consecutive) after
// d_batched_forward has filled every group: ONE launch per layer and direction instead of G, BatchNorm statistics per
// group (= per batch, as in the separate passes, train/dcgan_trainer.py:162,173,118).  Groups [0, G-1) are loss passes
// (targets / accumulator slots given per group) and contribute weight gradients; the LAST group is the penalty pass (head
// mode 1, gradient w.r.t. its input image -> dset[0].gx, norms -> dset[0].norms).  Weight gradients run on `side`.
static int d_batched_pass(jck_engine* e, const void* x_in, int B, int G, int pass0, const float* targets, const int* slot_loss,
                          const int* slot_p, hipStream_t st, hipStream_t side, bool forward_done = false, int part = 0, bool lazy_tail = false) {
  auto& S = e->bset;
  const size_t esz = e->esz;
  const int gw = G - 1;
  if (part == 2) return d_batched_backward(e, x_in, B, G, gw, true, st, side, true, 2);
  auto at = [&](void* p, size_t elems) { return (void*)((unsigned char*)p + elems * esz); };
  if (!forward_done) JCK_TRY(d_batched_forward(e, x_in, B, 0, G, pass0, st));
  {
    float tg[4] = {0.f, 0.f, 0.f, 0.f};
    int md[4] = {0, 0, 0, 0}, sl[4] = {-1, -1, -1, -1}, sp[4] = {-1, -1, -1, -1};
    for (int g = 0; g < G; ++g) {
      const bool pen = g == G - 1;
      tg[g] = pen ? 0.f : targets[g]; md[g] = pen ? 1 : 0; sl[g] = pen ? -1 : slot_loss[g]; sp[g] = pen ? -1 : slot_p[g];
    }
    // conv5's forward also writes the rows' input gradient and completes the hand-over event itself: the weight gradient (partial
    // rows + their ordered sum) is then all the weight-gradient stream's (JCK_HEAD_FUSE=0: head_bwd_fused_kernel does both on st)
    JCK_TRY(head_fwd_grouped_ev(e->prec, S.a[TT.NS - 1], e->d_head_wp, nullptr, B, TT.FEAT, G, tg, md, S.prob, S.ds, e->acc, sl, sp, e->acc_ld,
                                e->head_fuse ? S.g[TT.NS - 1] : nullptr, st, e->head_fuse && side && e->ext_events ? e->evHead : nullptr));
  }
  // loss groups (input + weight gradient) and the penalty group (input gradient only), one launch
  // (the ordered sum of conv5's weight-gradient rows is wanted by the optimiser only: weight-gradient stream, 10 us off the main one)
  static const bool head_side = !(getenv("JCK_HEAD_SIDE") && atoi(getenv("JCK_HEAD_SIDE")) == 0);
  const bool hs = side && e->ext_events && head_side;
  if (e->head_fuse) {
    const bool hf = side && e->ext_events;
    JCK_TRY(head_bwd_conv2_ev(e->prec, S.ds, e->d_head_wp, S.a[TT.NS - 1], gw * B, 0, TT.G_C1, nullptr, e->P(e->LD, e->dg, CWN[TT.NS]),
                              e->head_ws, st, hf ? side : nullptr, hf ? e->evHead : nullptr));
  } else
    JCK_TRY(head_bwd_conv2_ev(e->prec, S.ds, e->d_head_wp, S.a[TT.NS - 1], gw * B, B, TT.G_C1, S.g[TT.NS - 1], e->P(e->LD, e->dg, CWN[TT.NS]),
                              e->head_ws, st, hs ? side : nullptr, hs ? e->evHead : nullptr));
  JCK_TRY(d_batched_backward(e, x_in, B, G, gw, true, st, side, true, part, part == 1 && lazy_tail));
  return JCK_OK;
}

// The back-propagated gradient penalty (train/cgan_trainer.py:200-203).  Precondition: the GP forward and first backward
// have run (d_forward(xhat), d_head(mode 1), d_backward(wgrad=false, xgrad=true)), so d_g[i] = gy_i, d_bn[i].sums = first-
// backward sums, g_h = gradient at the Linear(8392,256) output, d_gx = g_x, norms = ||g_x[n]||.  Accumulates
// lambda * dGP/dtheta into D's gradient arena.  Derivation + fp64 check: tests/test_gp_double_backward_math.py.
// `side` (optional): the weight-gradient products run there beside the v-chain / reverse sweep; they share the split-K
// workspace, so everything that uses it is ordered on that one stream, and the main stream joins it before the reverse sweep
// (which overwrites the v_i the v-chain products read and runs the Linear weight gradient on the main stream) and at the end.
// where the penalty pass left its forward / first-backward tensors: activation set 0 (per-pass schedule) or the last group
// of the batched set
struct GpSrc { const void *y[JCK_MAX_STAGES], *a[JCK_MAX_STAGES], *g[JCK_MAX_STAGES]; const float *aux[JCK_MAX_STAGES], *sums[JCK_MAX_STAGES], *prob; };
static GpSrc gp_src_dset0(jck_engine* e) {
  GpSrc r;
  for (int i = 0; i < TT.NS; ++i) { r.y[i] = e->d_y[i]; r.a[i] = e->d_a[i]; r.g[i] = e->d_g[i]; r.aux[i] = e->d_bn[i].aux; r.sums[i] = e->d_bn[i].sums; }
  r.prob = e->prob;
  return r;
}
static GpSrc gp_src_group(jck_engine* e, int g, int B) {
  GpSrc r;
  auto& S = e->bset;
  for (int i = 0; i < TT.NS; ++i) {
    const size_t off = (size_t)g * B * (TT.D_HB[i] / 2) * (TT.D_HB[i] / 2) * TT.D_CS[i] * e->esz;
    r.y[i] = (const unsigned char*)S.y[i] + off; r.a[i] = (const unsigned char*)S.a[i] + off; r.g[i] = (const unsigned char*)S.g[i] + off;
    r.aux[i] = S.aux[i] + (size_t)g * 4 * TT.D_CS[i];
    r.sums[i] = S.sums[i] + (size_t)g * jck_bn_bwd_ws_floats(TT.D_CS[i]);
  }
  r.prob = S.prob + (size_t)g * B;
  return r;
}

static int gp_double_backward(jck_engine* e, const GpSrc& P, const void* xhat, int B, float lambda, const float* drop_mask, hipStream_t st,
                              hipStream_t side = nullptr, bool lazy_join = false) {
  const size_t esz = e->esz;
  auto fork = [&](int k) -> hipStream_t {            // work enqueued on the returned stream starts after everything on st so far
    if (!side) return st;
    (void)hipEventRecord(e->evW[k], st);
    (void)hipStreamWaitEvent(side, e->evW[k], 0);
    return side;
  };
  auto join = [&]() {
    if (!side) return;
    (void)hipEventRecord(e->evWdone, side);
    (void)hipStreamWaitEvent(st, e->evWdone, 0);
  };
  // the first backward's gradient at the Linear(8392,256) output and the penalty pass's probabilities: read in place after a
  // batched head (its penalty rows [2B, 3B) are not written again: the head backward below writes rows [0, B)), copied aside
  // otherwise (the per-pass head backward below overwrites g_h)
  // the head's parameter gradients of PHASE_D_LOSS may still be queued on the weight-gradient stream (cg_head_backward_batched, no
  // join in between under JCK_PHASE_LAZY_JOIN): this pass adds to the same tensors from the main stream
  if (side) (void)hipStreamWaitEvent(st, e->evHead, 0);
  const void* gh1 = e->gh_b1;
  const float* prob1 = e->prob_gp;
  if (e->head_row0 > 0) {
    gh1 = (const unsigned char*)e->g_h + (size_t)e->head_row0 * L1_OUT * esz;
    prob1 = P.prob;
  } else {
    HIPCHK(hipMemcpyAsync(e->gh_b1, e->g_h, (size_t)B * L1_OUT * esz, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(e->prob_gp, P.prob, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  JCK_TRY(jck_gp_grad(e->prec, e->d_gx, e->norms, 2.0f * lambda / (float)B, B, TT.HW, e->d_u0, st));
  // ---- v-chain: adjoint of the first backward, swept forward through D
  const void* u = e->d_u0;
  for (int i = 0; i < TT.NS; ++i) {
    const int hb = TT.D_HB[i], cs = TT.D_CS[i], cb = TT.D_CB[i];
    const long long rows = (long long)B * (hb / 2) * (hb / 2);
    // g_{a_{i-1}} = convT(gy_i; W_i):  dW_i += wgrad(gy_i, u_{i-1}),  v_i = conv(u_{i-1}; W_i)
    JCK_TRY(jck_conv_wgrad(e->prec, P.g[i], u, e->wg_ws, e->wg_ws_bytes, e->P(e->LD, e->dg, CWN[i]), 1, B, hb, hb, cb, cs, fork(i)));
    JCK_TRY(jck_conv_down(e->prec, u, e->d_down[i], e->d_v[i], nullptr, nullptr, B, hb, hb, cb, cs, st));
    JCK_TRY(jck_bn2_vchain(e->prec, e->d_v[i], P.y[i], P.g[i], P.aux[i], P.sums[i], e->P(e->LD, e->dp, NWN[i]),
                           LRELU, e->bn2_ws[i], e->d_v[i], e->d_xdir[i], e->P(e->LD, e->dg, NWN[i]), rows, cs, st));
    u = e->d_v[i];
  }
  // head: gc[:, :8192] = gh W1 -> adj(gh) = [u4 | 0] W1^T, dW1 += gh^T [u4 | 0]; gh = gh' * m/(1-p); gh' = ds w2; ds = p(1-p)
  JCK_TRY(jck_concat_rows(e->prec, e->d_v[TT.NS - 1], TT.FEAT, e->cbuf2, L1_KPAD, B, st));          // tail columns of cbuf2 stay zero
  JCK_TRY(jck_linear_wgrad(e->prec, gh1, L1_OUT, e->cbuf2, L1_KPAD, e->wg_ws, e->wg_ws_bytes, e->gw1p, 1, B, L1_OUT, fork(0)));
  // What the reverse sweep must wait for are the v-chain's products (they read the v_i it overwrites): the weight-gradient stream is
  // marked HERE, behind the last of them - not after the head's parameter gradients, which are forked to it below and which nothing
  // on the main stream waits for before the optimiser (JCK_GP_NARROW_JOIN=0: the join covers them too)
  static const bool narrow_join = !(getenv("JCK_GP_NARROW_JOIN") && atoi(getenv("JCK_GP_NARROW_JOIN")) == 0);
  const bool narrow = side && narrow_join;
  if (narrow) (void)hipEventRecord(e->evGP, side);
  JCK_TRY(jck_linear_fwd(e->prec, e->cbuf2, e->l1_w, nullptr, e->l1_slab, B, L1_KPAD, L1_OUT, L1_OUT, L1_KSPLIT, st));
  const bool fuse = e->head_fuse && L1_OUT == 256;    // the next four launches as one (ops.hip: gp_head_mid_ev)
  if (!fuse) JCK_TRY(jck_linear_finish(e->prec, e->l1_slab, L1_KSPLIT, nullptr, drop_mask, 1.0f / 0.75f, nullptr, e->ughd, B, L1_OUT, st));
  // (the head's parameter gradients - the dw2 sum here, linear2 / linear1 / the label embedding in d_head_backward below: ~75 us of
  // launches that only the optimiser waits for - go to the weight-gradient stream; JCK_HEAD_SIDE=0 keeps them on the main one)
  static const bool head_side = !(getenv("JCK_HEAD_SIDE") && atoi(getenv("JCK_HEAD_SIDE")) == 0);
  hipStream_t hs = head_side && e->ext_events ? side : nullptr;
  if (fuse) {
    // g_hd / g_h rows [head_wrow0, + B): not the rows the v-chain's Linear weight gradient reads (the penalty group's, or the copy)
    const size_t w0 = (size_t)e->head_wrow0 * L1_OUT * esz;
    JCK_TRY(gp_head_mid_ev(e->prec, e->l1_slab, L1_KSPLIT, drop_mask, 1.0f / 0.75f, e->ughd, e->P(e->LD, e->dp, "linear2.weight"), prob1, B, e->rs,
                           e->P(e->LD, e->dg, "linear2.weight"), e->gp2_ws, (unsigned char*)e->g_hd + w0, (unsigned char*)e->g_h + w0, st, hs,
                           hs ? e->evHead : nullptr));
  } else
    JCK_TRY(gp_head2_ev(e->prec, e->ughd, e->P(e->LD, e->dp, "linear2.weight"), prob1, B, L1_OUT, e->rs,
                        e->P(e->LD, e->dg, "linear2.weight"), e->gp2_ws, st, hs, hs ? e->evHead : nullptr));
  // ---- reverse sweep through the forward pass from the logit adjoint rs, with the extra BatchNorm inputs
  if (narrow) (void)hipStreamWaitEvent(st, e->evGP, 0); else join();
  JCK_TRY(d_head_backward(e, e->dset[0], e->rs, B, true, drop_mask, e->d_v[TT.NS - 1], st, hs, fuse));
  // the last sum into the permuted Linear gradient is enqueued: back to the reference's layout, on the stream that holds it
  JCK_TRY(jck_unperm_linear_grad(e->gw1p, L1_OUT, L1_K, L1_KPAD, 512, 16, e->P(e->LD, e->dg, "linear1.weight"), 1, hs ? hs : st));
  for (int i = TT.NS - 1; i >= 0; --i) {
    const int hb = TT.D_HB[i], cs = TT.D_CS[i], cb = TT.D_CB[i];
    const long long rows = (long long)B * (hb / 2) * (hb / 2);
    JCK_TRY(jck_bn2_reverse(e->prec, e->d_v[i], P.y[i], e->d_xdir[i], P.aux[i], e->P(e->LD, e->dp, NWN[i]), e->bn2_ws[i],
                            LRELU, e->bn2_ws_rev, e->d_v[i], e->P(e->LD, e->dg, NWN[i]), e->P(e->LD, e->dg, NBN[i]), rows, cs, st));
    const void* big = i == 0 ? xhat : P.a[i - 1];
    JCK_TRY(jck_conv_wgrad(e->prec, e->d_v[i], big, e->wg_ws, e->wg_ws_bytes, e->P(e->LD, e->dg, CWN[i]), 1, B, hb, hb, cb, cs, fork(i)));
    // everything but the bottom conv weight's gradient is final on the side stream here (lazy join: the optimiser phase starts
    // behind this point and takes that one tensor last)
    if (side && lazy_join && i == 1) { (void)hipEventRecord(e->evWmid, side); e->mid_recorded = true; }
    if (i > 0)
      JCK_TRY(jck_conv_up(e->prec, e->d_v[i], e->d_up[i], e->d_v[i - 1], nullptr, nullptr, 0, B, hb / 2, hb / 2, cs, cb, st));
  }
  if (side && lazy_join) { (void)hipEventRecord(e->evWdone, side); e->join_pending = true; }
  else join();
  return JCK_OK;
}

template <typename T>
static void launch_pad_rows(const float* z, int B, int zd, int zp, void* out, hipStream_t st) {
  const long long total = (long long)B * zp;
  hipLaunchKernelGGL(pad_rows_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, B, zd, zp, (T*)out);
}

static int g_forward(jck_engine* e, const float* z, const int64_t* labels, int B, hipStream_t st, bool z_in_place = false) {
  const int zp = z_pad(e->family);
  if (e->family == 1) {
    if (!labels) JCK_FAIL(JCK_E_ARG, "CGAN generator needs labels");
    JCK_TRY(jck_cgan_z(e->prec, z, labels, B, 100, N_CLASS, zp, e->g_z, st));            // model/CGAN.py:154-155
  } else if (!z_in_place) {                 // (z_in_place: the step's set-step launch drew this z and wrote the operand rows itself)
    e->gz_step = -1;                        // the operand rows now hold somebody else's z
    if (e->prec == JCK_PREC_BF16) launch_pad_rows<bf16_t>(z, B, 100, zp, e->g_z, st); else launch_pad_rows<float>(z, B, 100, zp, e->g_z, st);
    HIPCHK(hipGetLastError());
  }
  JCK_TRY(jck_g1_fwd(e->prec, e->g_z, e->g1_w, e->g_y[0], e->g_bn[0].stats, &e->g_bn[0].slots, B, zp, TT.G_C1, st));
  for (int i = 0; i < TT.NS; ++i) {
    const int h = 4 << i, C = TT.G_C1 >> i;
    const long long rows = (long long)B * h * h;
    bool fused = false;
    JCK_TRY(bn_fwd_fused(e->prec, e->g_y[i], e->g_bn[i].stats, e->g_bn[i].slots, (float)rows, e->P(e->LG, e->gp, NWN[i]), e->P(e->LG, e->gp, NBN[i]), BN_EPS,
                         0.f, e->g_a[i], e->g_bn[i].aux, nullptr, e->gbn + find(e->LG, RMN[i])->offset, e->gbn + find(e->LG, RVN[i])->offset,
                         (int64_t*)(e->gnbt + i), BN_MOM, rows, C, 1, 0, 0, st, &fused));
    if (!fused) {
    JCK_TRY(jck_bn_finalize(e->g_bn[i].stats, e->g_bn[i].slots, (float)rows, e->P(e->LG, e->gp, NWN[i]), e->P(e->LG, e->gp, NBN[i]),
                            e->gbn + find(e->LG, RMN[i])->offset, e->gbn + find(e->LG, RVN[i])->offset,
                            e->gnbt + i, BN_MOM, BN_EPS, e->g_bn[i].aux, C, st));
    JCK_TRY(jck_bn_act_fwd(e->prec, e->g_y[i], e->g_bn[i].aux, 0.f, e->g_a[i], rows, C, st));
    }
    if (i < TT.NS - 1)
      JCK_TRY(jck_conv_up_grouped(e->prec, e->g_a[i], e->g_up[i], e->g_y[i + 1], e->g_bn[i + 1].stats, &e->g_bn[i + 1].slots, B, h, h, TT.G_CS[i], TT.G_CB[i], B, st));
    else                                             // last ConvTranspose + tanh -> the image
      JCK_TRY(jck_conv_up(e->prec, e->g_a[i], e->g_up[i], e->fake_raw, nullptr, nullptr, 1, B, h, h, TT.G_CS[i], TT.G_CB[i], st));
  }
  return JCK_OK;
}

// g_fake = gradient w.r.t. the noisy fake image (NHWC4); fills G's grads arena (accumulating)
// raw_ready: g_raw was written by the launch that computed g_fake (XgradEpi), which also completed the first stage's event
static int g_backward(jck_engine* e, const void* g_fake, int B, hipStream_t st, hipStream_t side, bool raw_ready = false) {
  // every launch that writes a gradient the weight-gradient stream reads (tanh backward, then the BatchNorm backward of each
  // stage) completes that stage's event itself
  const bool ext = side && e->ext_events;
  if (!raw_ready)
    JCK_TRY(tanh_bwd_ev(e->prec, g_fake, e->fake_raw, 0.9f, e->g_raw, (long long)B * TT.HW * 4, st, ext ? e->evW[TT.NS - 1] : nullptr));
  const void* gbig = e->g_raw;       // gradient w.r.t. the output of conv(i+2)
  for (int i = TT.NS - 1; i >= 0; --i) {     // stage i: conv(i+2): small = g_a[i] (C = TT.G_CS[i]), big side has TT.G_CB[i] channels
    const int hs = TT.G_HS[i], cs = TT.G_CS[i], cb = TT.G_CB[i];
    hipStream_t ws = st;
    if (side) { if (!ext) HIPCHK(hipEventRecord(e->evW[i], st)); HIPCHK(hipStreamWaitEvent(side, e->evW[i], 0)); ws = side; }
    JCK_TRY(jck_conv_wgrad(e->prec, e->g_a[i], gbig, e->wg_ws, e->wg_ws_bytes, e->P(e->LG, e->gg, CWN[i + 1]), 1, B,
                           2 * hs, 2 * hs, cb, cs, ws));
    const long long rows = (long long)B * hs * hs;
    JCK_TRY(jck_conv_down(e->prec, gbig, e->g_down[i], e->g_gr[i], nullptr, nullptr, B, 2 * hs, 2 * hs, cb, cs, st));
    // g_gr[i] is what stage i - 1's weight gradient reads
    JCK_TRY(bn_act_bwd_res_ev(e->prec, e->g_gr[i], e->g_y[i], e->g_bn[i].aux, 0.f, e->g_bn[i].sums, e->g_gr[i],
                              e->P(e->LG, e->gg, NWN[i]), e->P(e->LG, e->gg, NBN[i]), rows, cs, 1, 1, e->bn_res ? e->gsync : nullptr, st,
                              ext && i > 0 ? e->evW[i - 1] : nullptr));
    gbig = e->g_gr[i];
  }
  // G.conv1's weight gradient depends on the last kernel of the main stream: it runs there, behind it, with its own
  // workspace (forking it to the side stream cost a ~12 us cross-stream hand-over at the very end of the step, twice)
  JCK_TRY(jck_g1_wgrad(e->prec, e->g_z, e->g_gr[0], e->g1_ws, e->g1_ws_bytes, e->P(e->LG, e->gg, CWN[0]), 1, B, z_dim(e->family),
                       z_pad(e->family), TT.G_C1, st));
  if (side) { HIPCHK(hipEventRecord(e->evWdone, side)); HIPCHK(hipStreamWaitEvent(st, e->evWdone, 0)); }
  return JCK_OK;
}

// the real batch -> NHWC4 with instance noise (:160): from an fp32 NCHW tensor, or gathered from the device-resident uint8
// dataset with the input transform applied on the fly
static int prep_real(jck_engine* e, const jck_step_inputs* in, int B, hipStream_t st) {
  if (in->real_u8) {
    if (TT.S != 64) JCK_FAIL(JCK_E_ARG, "the device-resident uint8 pipeline is the reference's Resize(64) of 32x32 images: 64x64 engines only");
    if (!in->noise_real)
      return jck_img_prep_u8_rng(e->prec, in->real_u8, in->real_idx, e->rng(), 0, 0.9f, 0.1f, e->real_noisy, B, 32, 32, st);
    return jck_img_prep_u8(e->prec, in->real_u8, in->real_idx, in->noise_real, 0.9f, 0.1f, e->real_noisy, nullptr, B, 32, 32, st);
  }
  if (!in->noise_real) return jck_img_prep_rng(e->prec, in->real_nchw, e->rng(), 0, 0.9f, 0.1f, e->real_noisy, B, TT.HW, st);
  return jck_img_prep(e->prec, in->real_nchw, in->noise_real, 0.9f, 0.1f, e->real_noisy, B, TT.HW, st);
}

// the step's accumulator rows (losses, probabilities, penalty norms): cleared by the launch that wrote the step's scalars
// (jck_engine_set_step) when that has just run for this step, else - captured steps, a phase called twice - by a memset
// An optimiser phase called with another learning rate than the step's loss phases (separate G and D rates): only the Adam
// scalars of the step's slot are rewritten - NOT the accumulator rows (the step's losses are in them, the tail of G_STEP reads
// them) and not the step's random inputs (jck_engine_set_step here logged loss_d = loss_g = gp = 0).
static int refresh_adam_scalars(jck_engine* e, int step, float lr, hipStream_t st) {
  const int q = step & 1;
  JCK_TRY(jck_adam_set_step(e->hp2 + 8 * q, (double)lr, 0.5, 0.999, step, e->noise_seed, st));
  e->hp_step[q] = step;
  e->hp_lr[q] = lr;
  return JCK_OK;
}
static int set_step_impl(jck_engine* e, int step, float lr, void* stream, bool zero_d);
// D.zero_grad() (:155): cleared by the step's set-step launch when that has just run for this step (zero_d), else by a memset
static int zero_d_grads(jck_engine* e, int step, hipStream_t st) {
  const bool clean = e->dg_clean_step == step && !e->capturing;
  e->dg_clean_step = -1;
  if (clean) return JCK_OK;
  HIPCHK(hipMemsetAsync(e->dg, 0, e->LD.n_params * sizeof(float), st));
  if (e->family == 1) HIPCHK(hipMemsetAsync(e->gw1p, 0, (size_t)L1_OUT * L1_KPAD * sizeof(float), st));
  return JCK_OK;
}
static int clear_acc(jck_engine* e, int step, hipStream_t st) {
  if (e->acc_clean_step == step && !e->capturing) { e->acc_clean_step = -1; return JCK_OK; }
  HIPCHK(hipMemsetAsync(e->acc, 0, (size_t)8 * e->acc_ld * sizeof(float), st));
  return JCK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// phases
// ---------------------------------------------------------------------------------------------------------
static int phase_impl(jck_engine* e, int phase, const jck_step_inputs* in_, void* stream);
extern "C" int jck_engine_phase(jck_engine* e, int phase, const jck_step_inputs* in_, void* stream) {
  // JCK_PHASE_NO_RESIDENT: no grid-barrier launch in this call (a collective may be holding CUs: include/jckgan.h)
  const bool keep_res = e ? e->bn_res : false;
  if (e && (phase & JCK_PHASE_NO_RESIDENT)) e->bn_res = false;
  const int rc = phase_impl(e, phase & ~JCK_PHASE_NO_RESIDENT, in_, stream);
  if (e) e->bn_res = keep_res;
  if (rc != JCK_OK && e) {
    // a phase that failed half way leaves no promise behind: the next D phase clears its arenas itself and joins the
    // weight-gradient stream before anything else (state armed before a failing call must not outlive it)
    e->acc_clean_step = e->dg_clean_step = e->gg_clean_step = -1;
    if (e->join_pending && e->sA && stream && !e->capturing) (void)hipStreamWaitEvent((hipStream_t)stream, e->evWdone, 0);
    e->join_pending = e->mid_recorded = false;
  }
  return rc;
}
static int phase_impl(jck_engine* e, int phase, const jck_step_inputs* in_, void* stream) {
  if (!e || !e->bound || !in_) JCK_FAIL(JCK_E_ARG, "engine not bound / null inputs");
  // z, alpha and (CGAN) the Dropout keep masks left NULL: the engine's own draws for this step (jck_engine_set_step)
  jck_step_inputs loc = *in_;
  {
    const int q = loc.step & 1;
    if (!loc.z) loc.z = e->rz[q];
    if (!loc.alpha) loc.alpha = e->ralpha[q];
    if (e->family == 1 && !loc.drop_mask[0] && !loc.drop_mask[1] && !loc.drop_mask[2] && !loc.drop_mask[3])
      for (int i = 0; i < 4; ++i) loc.drop_mask[i] = e->rmask[q] + (size_t)i * e->B * L1_OUT;
  }
  const jck_step_inputs* in = &loc;
  hipStream_t st = (hipStream_t)stream;
  const int B = e->B, HW = TT.HW;
  const bool cg = e->family == 1;
  const bool lazy = (phase & JCK_PHASE_LAZY_JOIN) && cg && !e->capturing;
  const bool lazy_tail = (phase & JCK_PHASE_LAZY_JOIN) && !cg && (phase & 0xff) == JCK_PHASE_D_LOSS_A;
  phase &= ~JCK_PHASE_LAZY_JOIN;
  // a join left open by the phase before (JCK_PHASE_LAZY_JOIN): the penalty's double backward keeps using the weight-gradient
  // stream in order and the optimiser phase closes it tensor by tensor; anything else waits here
  if (e->join_pending && !(phase == JCK_PHASE_D_STEP || (phase == JCK_PHASE_D_GP && cg && e->gp_done))) {
    HIPCHK(hipStreamWaitEvent(st, e->evWdone, 0));
    e->join_pending = e->mid_recorded = false;
  }
  if (cg) {
    if (!in->labels) JCK_FAIL(JCK_E_ARG, "CGAN phases need labels");
    e->cur_labels = in->labels;
  }
  DSet& D0 = e->dset[0];
  DSet& D1 = e->dset[1];
  DSet& DR = cg ? e->dset[0] : e->dset[2];          // D(real): own set so it may run beside the previous step's G phase
  e->parity = in->step & 1;
  // eager callers that did not call jck_engine_set_step: the step's scalars / Philox words are written by its first phase
  // (a phase called with another learning rate than the step's earlier phases - separate G and D rates - rewrites the Adam
  // scalars only: the step's accumulator rows and random inputs are live)
  if (!e->capturing) {
    if (e->hp_step[in->step & 1] != in->step)
      // (PHASE_D_REAL_FWD ahead of its step: D's gradients of the step before were consumed by its PHASE_D_STEP, and nothing writes
      // them until this step's D pass - the set-step launch clears them here as well, instead of a memset in front of that pass)
      JCK_TRY(set_step_impl(e, in->step, in->lr, st, (phase & 0xff) == JCK_PHASE_D_LOSS || (phase & 0xff) == JCK_PHASE_D_LOSS_A || (phase & 0xff) == JCK_PHASE_D_REAL ||
                                                    (phase & 0xff) == JCK_PHASE_D_REAL_FWD));
    else if (e->hp_lr[in->step & 1] != in->lr) JCK_TRY(refresh_adam_scalars(e, in->step, in->lr, st));
  }
  e->acc = e->acc2 + (size_t)8 * e->acc_ld * e->parity;
  e->scal_out = e->scal2 + 8 * e->parity;
  // stream overlap (DCGAN): A = wgrads, B = G forward beside D(real), C = penalty pass beside D(fake).  CGAN keeps the penalty
  // on the main stream (it produces gradients and shares the head buffers).
  // weight gradients beside the dgrad chain on a second stream: +8 % for DCGAN in round 1, +1.5 % now; CGAN +1.4 % (2.886 vs
  // 2.929 ms eager, round 3).  JCK_WGRAD_SIDE=0 keeps them on the main stream.  A captured step keeps everything on one stream
  // either way (below).
  static const bool wgrad_side = !(getenv("JCK_WGRAD_SIDE") && atoi(getenv("JCK_WGRAD_SIDE")) == 0);
  // Under a stream capture everything stays on the capturing stream: a hipGraph with parallel branches makes the ROCm 7.2
  // runtime keep per-graph side streams, costs ~7 us of host time per node at launch, ran slower than the linear graph for
  // CGAN, and its hipGraphLaunch reads past the end of the graph's stream pool whenever two of those streams share the launch
  // stream's hardware queue (hip::Graph::UpdateStreams; cause and frame in DESIGN.md section 5.6; jck_engine_capture_end refuses
  // any non-linear graph).  Same kernels, same order per stream as the eager schedule - bitwise the same results.
  const bool par = e->overlap && !e->capturing;
  hipStream_t sA = (par && wgrad_side) ? e->sA : nullptr;
  const bool ov_g = par, ov_gp = par && !cg;
  auto penalty_pass = [&](DSet& D, hipStream_t s) -> int {                                      // :110-127, 178
    JCK_TRY(jck_interp(e->prec, e->real_noisy, e->fake, in->alpha, e->xhat, B, HW, s));
    JCK_TRY(d_forward(e, D, e->xhat, B, 2, in->drop_mask[2], s));
    JCK_TRY(d_head(e, D, B, 0.f, 1, -1, -1, s));
    // DCGAN's per-pass schedule runs this pass on its own stream beside D(fake): three-launch BatchNorm backward (d_backward),
    // whatever the stream layout of this call (a captured one-stream step stays launch for launch the eager one)
    JCK_TRY(d_backward(e, D, e->xhat, B, false, true, in->drop_mask[2], s, nullptr, true, cg));
    JCK_TRY(jck_gp_norm(e->prec, D.gx, B, HW, e->acc, 6, e->acc_ld, D.norms, s));
    return JCK_OK;
  };
  switch (phase) {
    case JCK_PHASE_D_LOSS_A:
    case JCK_PHASE_D_LOSS_B:
      if (cg || e->batched != 3 || e->capturing) JCK_FAIL(JCK_E_ARG, "PHASE_D_LOSS_A / _B: only with the batched DCGAN schedule, outside a capture");
      [[fallthrough]];
    case JCK_PHASE_D_LOSS:
      if (cg && e->batched) {
        // CGAN: the real, fake and penalty passes (:181-203) share D's weights -> their conv stacks run as ONE 3B pass with
        // grouped BatchNorm.  The label / Linear / Dropout head runs per batch, forward and backward back to back, so the
        // head buffers serve all three; the penalty group goes last and leaves them as its double backward (PHASE_D_GP)
        // expects them.
        if ((!in->real_nchw && !in->real_u8) || !in->z || !in->alpha) JCK_FAIL(JCK_E_ARG, "PHASE_D_LOSS needs real_nchw (or real_u8), z and alpha");
        JCK_TRY(clear_acc(e, in->step, st));
        JCK_TRY(zero_d_grads(e, in->step, st));
        // the three dropout masks back to back (hipgan/engine.py hands them over that way): the head runs once over 3B rows
        const bool head3 = in->drop_mask[0] && in->drop_mask[1] == in->drop_mask[0] + (size_t)B * L1_OUT &&
                           in->drop_mask[2] == in->drop_mask[0] + (size_t)2 * B * L1_OUT;
        const bool direct = head3 && e->cbuf_direct;
        // D(real)'s conv stack beside G's forward, [fake | penalty] as a 2B forward behind it (see the DCGAN branch below)
        const bool split = e->real_side, beside = split && sA;
        hipStream_t sr = beside ? sA : st;
        if (beside) { HIPCHK(hipEventRecord(e->ev0, st)); HIPCHK(hipStreamWaitEvent(sA, e->ev0, 0)); }
        JCK_TRY(prep_real(e, in, B, sr));
        if (split) JCK_TRY(d_batched_forward(e, e->real_noisy, B, 0, 1, 0, sr, direct));
        if (beside) HIPCHK(hipEventRecord(e->evReal, sA));
        JCK_TRY(g_forward(e, in->z, in->labels, B, st, in->z == e->rz[in->step & 1] && e->gz_step == (long long)in->step));
        if (beside) HIPCHK(hipStreamWaitEvent(st, e->evReal, 0));
        JCK_TRY(e->mix_fake_noise_interp(in, B, st));     // :171, :111-113
        if (split) JCK_TRY(d_batched_forward(e, e->fake, B, 1, 2, 0, st, direct));
        else JCK_TRY(d_batched_forward(e, e->real_noisy, B, 0, 3, 0, st, direct));
        auto& S = e->bset;
        const float tg[2] = {0.9f, 0.1f};
        e->head_row0 = 0;
        const bool fuse = head3 && e->head_fuse;
        if (head3) JCK_TRY(cg_head_forward(e, direct ? nullptr : S.a[TT.NS - 1], 3 * B, in->drop_mask[0], st, B, !fuse));
        if (head3) {
          const float tg3[3] = {tg[0], tg[1], 0.f};
          const int md3[3] = {0, 0, 1}, sl3[3] = {0, 1, -1}, sp3[3] = {3, 4, -1};
          if (fuse) JCK_TRY(cg_head_middle(e, B, 3, tg3, md3, sl3, sp3, S.prob, S.ds, in->drop_mask[0], st));
          else
            JCK_TRY(jck_head_fwd_grouped(e->prec, e->h_drop, e->P(e->LD, e->dp, "linear2.weight"), e->P(e->LD, e->dp, "linear2.bias"), B, L1_OUT,
                                         3, tg3, md3, S.prob, S.ds, e->acc, sl3, sp3, e->acc_ld, st));
        }
        for (int g = 0; g < 3 && !head3; ++g) {
          const bool pen = g == 2;
          void* a4 = (unsigned char*)S.a[TT.NS - 1] + (size_t)g * B * TT.FEAT * e->esz;
          void* g4 = (unsigned char*)S.g[TT.NS - 1] + (size_t)g * B * TT.FEAT * e->esz;
          if (!head3) JCK_TRY(cg_head_forward(e, a4, B, in->drop_mask[g], st));
          const void* hd = (const unsigned char*)e->h_drop + (head3 ? (size_t)g * B * L1_OUT * e->esz : 0);
          JCK_TRY(jck_head_fwd(e->prec, hd, e->P(e->LD, e->dp, "linear2.weight"), e->P(e->LD, e->dp, "linear2.bias"), B, L1_OUT,
                               pen ? 0.f : tg[g], pen ? 1 : 0, S.prob + g * B, S.ds + g * B, e->acc, pen ? -1 : g, pen ? -1 : 3 + g, e->acc_ld, st));
          if (!head3) JCK_TRY(d_head_backward(e, e->dset[0], S.ds + g * B, B, !pen, in->drop_mask[g], g4, st));
        }
        if (head3) {
          static const bool head_side = !(getenv("JCK_HEAD_SIDE") && atoi(getenv("JCK_HEAD_SIDE")) == 0);
          JCK_TRY(cg_head_backward_batched(e, S.ds, B, in->drop_mask[0], S.g[TT.NS - 1], st, head_side ? sA : nullptr, fuse));
          e->head_row0 = 2 * B;                        // where PHASE_D_GP finds the penalty group's head state
          e->head_wrow0 = 0;
        }
        JCK_TRY(d_batched_backward(e, e->real_noisy, B, 3, 2, true, st, sA, true, 0, lazy));
        e->gp_done = true;
        return JCK_OK;
      }
      if (!cg && e->batched == 3) {                   // [real | fake | penalty] as one 3B pass after G's forward
        const float tgB[2] = {0.9f, 0.1f};
        const int slB[2] = {0, 1}, spB[2] = {3, 4};
        if (phase == JCK_PHASE_D_LOSS_B) {            // second half of a split pass (after the early all-reduce has started)
          JCK_TRY(d_batched_pass(e, e->real_noisy, B, 3, 0, tgB, slB, spB, st, sA, true, 2));
          e->gp_done = true;
          return JCK_OK;
        }
        const int part = phase == JCK_PHASE_D_LOSS_A ? 1 : 0;
        if ((!in->real_nchw && !in->real_u8) || !in->z || !in->alpha) JCK_FAIL(JCK_E_ARG, "PHASE_D_LOSS needs real_nchw (or real_u8), z and alpha");
        // D(real)'s forward of this step may already have run under the previous step's G all-reduce (PHASE_D_REAL_FWD)
        const bool pre = e->real_fwd_step == (long long)in->step;
        e->real_fwd_step = -1;
        JCK_TRY(clear_acc(e, in->step, st));
        JCK_TRY(zero_d_grads(e, in->step, st));                                                   // D.zero_grad()  :155
        // D(real)'s forward needs nothing of G, and G's forward is a chain of small launches that leaves most of the chip idle: the
        // real batch's input transform and conv stack run on the weight-gradient stream (idle at the start of a step) beside it,
        // [fake | penalty] follow as one 2B forward - the split PHASE_D_REAL_FWD makes across steps; the arithmetic of the 3B forward
        // (a batch's BatchNorm statistics may be split into partial sums differently: summation order only)
        // (1.632 -> 1.609 ms; JCK_REAL_SIDE=0: one 3B forward behind G's)
        // (the split does not depend on the stream layout: a captured, one-stream step is the same arithmetic launch for launch)
        const bool split = pre || e->real_side;
        const bool beside = !pre && e->real_side && sA;
        const bool pre_side = pre && e->pre_on_side;    // PHASE_D_REAL_FWD ran on the weight-gradient stream: joined behind G's forward
        e->pre_on_side = false;
        hipStream_t sr = beside ? sA : st;
        if (beside) { HIPCHK(hipEventRecord(e->ev0, st)); HIPCHK(hipStreamWaitEvent(sA, e->ev0, 0)); }
        if (!pre) JCK_TRY(prep_real(e, in, B, sr));                                               // :160
        if (!pre && split) JCK_TRY(d_batched_forward(e, e->real_noisy, B, 0, 1, 0, sr));          // :162
        if (beside) HIPCHK(hipEventRecord(e->evReal, sA));
        JCK_TRY(g_forward(e, in->z, in->labels, B, st, in->z == e->rz[in->step & 1] && e->gz_step == (long long)in->step));                                          // :168-169
        if (beside || pre_side) HIPCHK(hipStreamWaitEvent(st, e->evReal, 0));
        JCK_TRY(e->mix_fake_noise_interp(in, B, st));                                             // :171, :111-113
        const float tg[2] = {0.9f, 0.1f};
        const int sl[2] = {0, 1}, sp[2] = {3, 4};
        if (split) JCK_TRY(d_batched_forward(e, e->fake, B, 1, 2, 0, st));                        // :173, 118 as one 2B forward
        JCK_TRY(d_batched_pass(e, e->real_noisy, B, 3, 0, tg, sl, sp, st, sA, split, part, lazy_tail));      // :162-176, 178
        if (part == 0) e->gp_done = true;
        return JCK_OK;
      }
      if (phase != JCK_PHASE_D_LOSS) JCK_FAIL(JCK_E_ARG, "PHASE_D_LOSS_A / _B: only with the batched DCGAN schedule");
      [[fallthrough]];
    case JCK_PHASE_D_REAL:
    case JCK_PHASE_D_FAKE: {
      if (phase != JCK_PHASE_D_FAKE) {              // ---- D on the real batch (:155-165); independent of G
        if (!in->real_nchw && !in->real_u8) JCK_FAIL(JCK_E_ARG, "PHASE_D_REAL needs real_nchw (or real_u8)");
        JCK_TRY(clear_acc(e, in->step, st));
        JCK_TRY(zero_d_grads(e, in->step, st));                                                 // D.zero_grad()  :155
      }
      hipStream_t sG = st;
      if (phase == JCK_PHASE_D_LOSS || phase == JCK_PHASE_D_FAKE) {   // G forward: beside D(real) when both are in this call
        if (!in->z) JCK_FAIL(JCK_E_ARG, "G forward needs z");
        if (ov_gp && !in->alpha) JCK_FAIL(JCK_E_ARG, "alpha is needed here (the penalty pass starts inside this phase)");
        if (ov_g) { HIPCHK(hipEventRecord(e->ev0, st)); HIPCHK(hipStreamWaitEvent(e->sB, e->ev0, 0)); sG = e->sB; }
        JCK_TRY(g_forward(e, in->z, in->labels, B, sG, in->z == e->rz[in->step & 1] && e->gz_step == (long long)in->step));                                          // :168-169
        JCK_TRY(e->mix_fake_noise(in, B, sG));   // :171
        if (ov_g) HIPCHK(hipEventRecord(e->evF, sG));
      }
      if (phase != JCK_PHASE_D_FAKE) {
        JCK_TRY(prep_real(e, in, B, st));   // :160
        if (ov_gp && phase == JCK_PHASE_D_LOSS) HIPCHK(hipEventRecord(e->evReal, st));           // penalty pass needs only this
        JCK_TRY(d_forward(e, DR, e->real_noisy, B, 0, in->drop_mask[0], st));                      // :162
        JCK_TRY(d_head(e, DR, B, 0.9f, 0, 0, 3, st));                                             // :163,165
        // D(real)'s weight gradients stay in flight on sA while D(fake) starts (own activation set); D(fake)'s backward
        // queues behind them on sA and joins.  PHASE_D_REAL alone (pipeline mode) and CGAN (shared set) join here.
        JCK_TRY(d_backward(e, DR, e->real_noisy, B, true, false, in->drop_mask[0], st, sA,
                           phase == JCK_PHASE_D_REAL || cg || !e->defer_join));                                     // :164 (cgan :203)
        if (phase == JCK_PHASE_D_REAL) return JCK_OK;
      }
      // ---- D on the fake batch (:170-176) with the penalty pass (:178) beside it
      if (ov_g) HIPCHK(hipStreamWaitEvent(st, e->evF, 0));
      if (ov_gp) {                                   // penalty pass on its own stream and activation set
        if (phase == JCK_PHASE_D_FAKE) HIPCHK(hipEventRecord(e->evReal, st));    // real_noisy came from an earlier call on st
        HIPCHK(hipStreamWaitEvent(e->sC, e->evF, 0));
        HIPCHK(hipStreamWaitEvent(e->sC, e->evReal, 0));
        JCK_TRY(penalty_pass(D1, e->sC));
        HIPCHK(hipEventRecord(e->evGP, e->sC));
        e->gp_inflight = true;
      }
      JCK_TRY(d_forward(e, D0, e->fake, B, 1, in->drop_mask[1], st));                             // :173
      JCK_TRY(d_head(e, D0, B, 0.1f, 0, 1, 4, st));                                               // :174,176
      JCK_TRY(d_backward(e, D0, e->fake, B, true, false, in->drop_mask[1], st, sA));              // :175
      return JCK_OK;
    }
    case JCK_PHASE_D_REAL_FWD: {
      // forward half of D(real) of step in->step, ahead of its PHASE_D_LOSS: touches only real_noisy and group 0 of the batched
      // set (activations, statistic slots, scale/shift table) and the BatchNorm record of (parity of in->step, pass 0) - nothing
      // the G phase of the step before it reads or writes
      if (cg || e->batched != 3 || e->capturing) JCK_FAIL(JCK_E_ARG, "PHASE_D_REAL_FWD: only with the batched DCGAN schedule, outside a capture");
      if (!in->real_nchw && !in->real_u8) JCK_FAIL(JCK_E_ARG, "PHASE_D_REAL_FWD needs real_nchw (or real_u8)");
      // on the weight-gradient stream when there is one (idle here: G's backward has joined it): the caller's stream then only
      // carries the wait for G's all-reduce and Adam(G), and the rest of this forward runs beside the next step's G forward
      hipStream_t sr = (e->real_side && sA) ? sA : st;
      if (sr != st) { HIPCHK(hipEventRecord(e->ev0, st)); HIPCHK(hipStreamWaitEvent(sr, e->ev0, 0)); }
      JCK_TRY(prep_real(e, in, B, sr));                                                           // :160
      JCK_TRY(d_batched_forward(e, e->real_noisy, B, 0, 1, 0, sr));                               // :162
      if (sr != st) HIPCHK(hipEventRecord(e->evReal, sr));
      e->pre_on_side = sr != st;
      e->real_fwd_step = in->step;
      return JCK_OK;
    }
    case JCK_PHASE_GP_ONLY: {
      // The stand-alone gradient penalty of the module path (train/dcgan_trainer.py:110-127, train/cgan_trainer.py:114-131 called
      // outside this engine's step): real_nchw = real_data, noise_real = fake_data - both [B,3,S,S] fp32, taken as they are -, alpha
      // (CGAN: labels, drop_mask[2]).  Leaves the per-image gradient norms in "norms" (the caller forms mean((n - 1)^2)) and, CGAN,
      // d(penalty)/d(theta_D) with lambda = 1 in D's gradient arena, which is cleared first: the double backward in closed form.
      if (!in->real_nchw || !in->noise_real || !in_->alpha) JCK_FAIL(JCK_E_ARG, "PHASE_GP_ONLY needs real_nchw (real), noise_real (fake) and alpha");
      if (e->capturing) JCK_FAIL(JCK_E_ARG, "PHASE_GP_ONLY: not inside a capture");
      if (cg && !in_->drop_mask[2]) JCK_FAIL(JCK_E_ARG, "PHASE_GP_ONLY (CGAN) needs drop_mask[2]");
      HIPCHK(hipMemsetAsync(e->acc, 0, (size_t)8 * e->acc_ld * sizeof(float), st));
      HIPCHK(hipMemsetAsync(e->dg, 0, e->LD.n_params * sizeof(float), st));
      if (cg) HIPCHK(hipMemsetAsync(e->gw1p, 0, (size_t)L1_OUT * L1_KPAD * sizeof(float), st));
      e->acc_clean_step = e->dg_clean_step = -1;
      e->gp_done = e->gp_inflight = false;
      e->head_row0 = e->head_wrow0 = 0;
      JCK_TRY(jck_img_prep(e->prec, in->real_nchw, nullptr, 1.0f, 0.0f, e->real_noisy, B, HW, st));
      JCK_TRY(jck_img_prep(e->prec, in->noise_real, nullptr, 1.0f, 0.0f, e->fake, B, HW, st));
      JCK_TRY(penalty_pass(D0, st));
      if (cg) JCK_TRY(gp_double_backward(e, gp_src_dset0(e), e->xhat, B, 1.0f, in->drop_mask[2], st, nullptr));
      return JCK_OK;
    }
    case JCK_PHASE_D_GP: {
      if (!in->alpha) JCK_FAIL(JCK_E_ARG, "PHASE_D_GP needs alpha");
      if (e->gp_done && cg) {                         // forward and first backward ran as group 2 of the batched pass
        e->gp_done = false;
        JCK_TRY(gp_double_backward(e, gp_src_group(e, 2, B), e->xhat, B, 10.0f, in->drop_mask[2], st, sA, lazy));
        if (!(lazy && sA)) e->join_pending = e->mid_recorded = false;      // it joined the stream itself
        e->head_row0 = 0;
        return JCK_OK;
      }
      if (e->gp_done) { e->gp_done = false; return JCK_OK; }    // computed inside the batched pass of PHASE_D_LOSS
      if (e->gp_inflight) {                           // started in PHASE_D_LOSS: just join
        HIPCHK(hipStreamWaitEvent(st, e->evGP, 0));
        e->gp_inflight = false;
        return JCK_OK;
      }
      JCK_TRY(penalty_pass(D0, st));
      if (cg) {                                      // CGAN back-propagates the penalty (train/cgan_trainer.py:200-203)
        JCK_TRY(gp_double_backward(e, gp_src_dset0(e), e->xhat, B, 10.0f, in->drop_mask[2], st, sA));
      }
      return JCK_OK;
    }
    case JCK_PHASE_D_STEP: {                                                                      // :180
      if (e->gp_inflight) JCK_FAIL(JCK_E_ARG, "PHASE_D_GP must be called before PHASE_D_STEP");
      if (!e->hp_holds(in->step, in->lr) && !e->capturing) JCK_TRY(refresh_adam_scalars(e, in->step, in->lr, st));   // (eager callers)
      const float* hp = e->hp2 + 8 * e->parity;
      // G.zero_grad() (:182) rides on D's Adam launch: G's gradient arena is dead between G's optimiser step and PHASE_G_LOSS
      float* zg = (e->fold_zero && !e->capturing && e->LG.n_params % 4 == 0) ? e->gg : nullptr;
      if (zg) e->gg_clean_step = in->step;
      if (e->join_pending) {
        // the weight-gradient stream is still on its last products: Adam and the Linear repack for everything behind the bottom
        // conv weight in the arena now, [label embedding | conv1.weight] and the conv repack once that stream is through
        const TensorInfo* c1 = find(e->LD, CWN[0]);
        const long long cut = c1->offset + c1->numel;
        HIPCHK(hipStreamWaitEvent(st, e->mid_recorded ? e->evWmid : e->evWdone, 0));
        JCK_TRY(jck_adam_hp(e->dp + cut, e->dg + cut, e->dm + cut, e->dv + cut, e->LD.n_params - cut, 0.5, 0.999, 1e-8, in->grad_scale, hp, st,
                            zg, e->LG.n_params, jck_grid_sync_error_word(e->gsync)));
        JCK_TRY(repack_linear(e, st));
        if (e->mid_recorded) HIPCHK(hipStreamWaitEvent(st, e->evWdone, 0));
        e->join_pending = e->mid_recorded = false;
        JCK_TRY(jck_adam_hp(e->dp, e->dg, e->dm, e->dv, cut, 0.5, 0.999, 1e-8, in->grad_scale, hp, st, nullptr, 0, jck_grid_sync_error_word(e->gsync)));
        return repack_convs(e, 1, st);
      }
      JCK_TRY(jck_adam_hp(e->dp, e->dg, e->dm, e->dv, e->LD.n_params, 0.5, 0.999, 1e-8, in->grad_scale, hp, st, zg, e->LG.n_params,
                          jck_grid_sync_error_word(e->gsync)));
      return jck_engine_repack(e, 1, st);
    }
    case JCK_PHASE_G_LOSS: {                                                                      // :182-188
      if (!(e->gg_clean_step == in->step && !e->capturing)) HIPCHK(hipMemsetAsync(e->gg, 0, e->LG.n_params * sizeof(float), st));
      e->gg_clean_step = -1;
      const bool fuse = e->head_fuse;                 // CGAN: the middle of the head as one launch (cg_head_middle); DCGAN: conv5's forward
                                                      // writes its input gradient too
      JCK_TRY(d_forward(e, D0, e->fake, B, 3, in->drop_mask[3], st, !fuse));
      if (fuse && !cg) JCK_TRY(d_head(e, D0, B, 0.9f, 0, 2, 5, st, D0.g[TT.NS - 1]));
      else if (fuse) {
        const float tg1 = 0.9f;
        const int md1 = 0, sl1 = 2, sp1 = 5;
        JCK_TRY(cg_head_middle(e, B, 1, &tg1, &md1, &sl1, &sp1, D0.prob, D0.ds, in->drop_mask[3], st));
      } else JCK_TRY(d_head(e, D0, B, 0.9f, 0, 2, 5, st));
      // D's own weight gradients from this pass are dead (zeroed at :155 before they are read): skipped
      XgradEpi xe = {e->fake_raw, 0.9f, e->g_raw, (sA && e->ext_events) ? e->evW[TT.NS - 1] : nullptr, false};
      JCK_TRY(d_backward(e, D0, e->fake, B, false, true, in->drop_mask[3], st, nullptr, true, true, e->fuse_tanh ? &xe : nullptr, fuse));
      JCK_TRY(g_backward(e, D0.gx, B, st, sA, xe.fused));
      return JCK_OK;
    }
    case JCK_PHASE_G_STEP: {                                                                      // :189
      if (!e->hp_holds(in->step, in->lr) && !e->capturing) JCK_TRY(refresh_adam_scalars(e, in->step, in->lr, st));
      JCK_TRY(jck_adam_hp(e->gp, e->gg, e->gm, e->gv, e->LG.n_params, 0.5, 0.999, 1e-8, in->grad_scale, e->hp2 + 8 * e->parity, st, nullptr, 0,
                          jck_grid_sync_error_word(e->gsync)));
      {   // G's repack, the four D passes' BatchNorm records in the reference's order and the logged scalars: one launch
        TailJobs t = {};
        for (int i = 0; i < TT.NS; ++i) {
          const int cs = TT.D_CS[i];
          t.l[i].rec = e->d_rs[i] + (size_t)e->parity * 4 * 2 * cs;
          t.l[i].rm = e->dbn + find(e->LD, RMN[i])->offset;
          t.l[i].rv = e->dbn + find(e->LD, RVN[i])->offset;
          t.l[i].nbt = (long long*)(e->dnbt + i);
          t.l[i].C = cs;
        }
        t.nl = TT.NS; t.npass = 4; t.momentum = BN_MOM; t.acc = e->acc; t.acc_ld = e->acc_ld; t.B = B; t.invB = 1.0f / (float)B; t.lambda_gp = 10.0f; t.out = e->scal_out;
        JCK_TRY(repack_convs(e, 0, st, &t, cdiv(TT.G_C1, 256)));
      }
      return JCK_OK;
    }
  }
  JCK_FAIL(JCK_E_ARG, "unknown phase");
}

// Per-step optimiser scalars (Adam bias corrections of step `step` at learning rate `lr`, computed on the host in double as
// torch.optim.Adam does) -> device memory of parity step & 1.  Call once per step BEFORE its phases when the phases are
// replayed from a captured graph (the graph bakes every kernel argument); eager callers may skip it - the optimiser phases
// then write the scalars themselves.
// zero_d: the caller is the step's first D-loss phase - the same launch clears D's gradient arena (D.zero_grad(), :155)
static int set_step_impl(jck_engine* e, int step, float lr, void* stream, bool zero_d) {
  if (!e || !e->bound) JCK_FAIL(JCK_E_ARG, "engine not bound");
  if (e->capturing) JCK_FAIL(JCK_E_ARG, "set_step inside a graph capture would bake one step's scalars into the graph");
  const int q = step & 1;
  static const bool fold_z = !(getenv("JCK_FOLD_Z") && atoi(getenv("JCK_FOLD_Z")) == 0);      // (=0: pad_rows_kernel in front of G's forward)
  zero_d = zero_d && e->fold_zero && e->LD.n_params % 4 == 0;
  JCK_TRY(jck_adam_set_step(e->hp2 + 8 * q, (double)lr, 0.5, 0.999, step, e->noise_seed, (hipStream_t)stream, e->rz[q], (long long)e->B * 100,
                            e->ralpha[q], e->B, e->rmask[q], e->rmask[q] ? (long long)4 * e->B * L1_OUT : 0, 0.75f,
                            e->acc2 + (size_t)8 * e->acc_ld * q, (long long)8 * e->acc_ld,
                            zero_d ? e->dg : nullptr, e->LD.n_params, zero_d && e->family == 1 ? e->gw1p : nullptr, (long long)L1_OUT * L1_KPAD,
                            // DCGAN: the drawn z goes straight into G.conv1's operand rows too (no pad_rows launch in front of G's forward
                            // when the step uses the engine's own z)
                            e->family == 0 && e->fold_zero && fold_z ? e->g_z : nullptr, 100, z_pad(e->family), e->prec == JCK_PREC_F32 ? 1 : 0));
  e->gz_step = (e->family == 0 && e->fold_zero && fold_z) ? step : -1;
  if (zero_d) e->dg_clean_step = step;
  e->acc_clean_step = step;
  e->hp_step[step & 1] = step;
  e->hp_lr[step & 1] = lr;
  return JCK_OK;
}
extern "C" int jck_engine_set_step(jck_engine* e, int step, float lr, void* stream) { return set_step_impl(e, step, lr, stream, false); }
// Seed of the in-kernel instance noise (steps whose jck_step_inputs carry no noise tensors draw 0.1*N(0,1) inside the image
// kernels: Philox4x32-10 keyed by this seed, counter = pixel | tensor | optimiser step).  Data-parallel ranks pass seed + rank.
extern "C" int jck_engine_set_noise_seed(jck_engine* e, unsigned long long seed) {
  if (!e) JCK_FAIL(JCK_E_ARG, "null engine");
  e->noise_seed = seed;
  e->hp_step[0] = e->hp_step[1] = 0;      // the next step rewrites the device copy
  e->hp_lr[0] = e->hp_lr[1] = -1.f;
  return JCK_OK;
}

// hipGraph capture of phases: begin -> jck_engine_phase(...) x n on the same stream -> end gives an executable graph that
// replays exactly those launches (incl. the side-stream forks and joins, which become graph dependencies).  Everything a
// phase reads through jck_step_inputs pointers is baked: the caller keeps those buffers at fixed addresses, refreshes
// their contents before each launch, keeps one graph per step parity (scalar / BatchNorm-record buffers alternate) and
// calls jck_engine_set_step first.  `stream` must not be the legacy default stream.
extern "C" int jck_engine_capture_begin(jck_engine* e, void* stream) {
  if (!e || !e->bound) JCK_FAIL(JCK_E_ARG, "engine not bound");
  if (!stream) JCK_FAIL(JCK_E_ARG, "cannot capture on the default stream");
  if (jck_prof_is_on()) JCK_FAIL(JCK_E_ARG, "per-launch profiling (jck_prof_enable) records timing events: not capturable");
  // relaxed: a call that is "unsafe" during a capture (a free, a stream destroy from a garbage-collected object, ...) issued
  // by this thread must not invalidate the graph being captured
  HIPCHK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
  e->capturing = true;
  return JCK_OK;
}
extern "C" int jck_engine_capture_end(jck_engine* e, void* stream, void** graph_exec) {
  if (!e || !graph_exec) JCK_FAIL(JCK_E_ARG, "null argument");
  e->capturing = false;
  hipGraph_t g = nullptr;
  HIPCHK(hipStreamEndCapture((hipStream_t)stream, &g));
  // Only LINEAR graphs are instantiated (every node at most one successor, one root).  ROCm 7.2's hip::GraphExec creates
  // max_streams_ internal streams for a graph with parallel branches and hip::Graph::UpdateStreams (called by every
  // hipGraphLaunch of such a graph) skips each of them that shares the launch stream's HARDWARE queue - with no bound on the
  // index: when two or more of the internal streams sit on the launch stream's queue (streams are dealt onto a small pool of
  // hardware queues, so this depends on every stream the process ever created) it reads past the end of the vector and
  // dereferences garbage - the SIGSEGV of round 2.  A graph whose width is 1 never enters that code.
  {
    size_t nn = 0, ne = 0;
    hipError_t q = hipGraphGetNodes(g, nullptr, &nn);
    if (q == hipSuccess) q = hipGraphGetEdges(g, nullptr, nullptr, &ne);
    bool linear = q == hipSuccess && nn > 0 && ne == nn - 1;
    if (linear && ne > 0) {
      std::vector<hipGraphNode_t> from(ne), to(ne);
      q = hipGraphGetEdges(g, from.data(), to.data(), &ne);
      linear = q == hipSuccess;
      for (size_t i = 0; linear && i < ne; ++i)
        for (size_t k = i + 1; k < ne; ++k)
          if (from[i] == from[k] || to[i] == to[k]) { linear = false; break; }       // a fork or a join
    }
    if (!linear) {
      (void)hipGraphDestroy(g);
      JCK_FAIL(JCK_E_ARG, "captured graph has parallel branches (" + std::to_string(nn) + " nodes, " + std::to_string(ne) +
                              " edges): not instantiated on this runtime - see DESIGN.md section 5.6");
    }
  }
  hipGraphExec_t ge = nullptr;
  hipError_t rc = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (rc != hipSuccess) JCK_FAIL(JCK_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(rc));
  *graph_exec = ge;
  return JCK_OK;
}
// abandon a capture after a failed phase (the stream leaves capture mode; nothing was executed)
extern "C" int jck_engine_capture_abort(jck_engine* e, void* stream) {
  if (e) e->capturing = false;
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture((hipStream_t)stream, &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  // side streams that joined the broken capture may be left invalidated: replace them
  if (e && e->overlap) {
    hipStream_t* ss[3] = {&e->sA, &e->sB, &e->sC};
    for (auto pp : ss) {
      if (*pp) (void)hipStreamDestroy(*pp);
      *pp = nullptr;
      HIPCHK(hipStreamCreateWithFlags(pp, hipStreamNonBlocking));
    }
    // ... and so may the events recorded inside it
    hipEvent_t* ev[8 + JCK_MAX_STAGES] = {&e->evWdone, &e->evWmid, &e->evHead, &e->evTail, &e->ev0, &e->evF, &e->evReal, &e->evGP};
    for (int i = 0; i < JCK_MAX_STAGES; ++i) ev[8 + i] = &e->evW[i];
    for (auto p : ev) {
      if (*p) (void)hipEventDestroy(*p);
      *p = nullptr;
      HIPCHK(hipEventCreateWithFlags(p, jck_event_flags()));
    }
    e->gp_inflight = false;
    e->join_pending = e->mid_recorded = false;
    (void)hipGetLastError();
  }
  return JCK_OK;
}
extern "C" int jck_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) JCK_FAIL(JCK_E_ARG, "null graph");
  HIPCHK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return JCK_OK;
}
extern "C" void jck_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

// Makes `stream` wait until the tail of D's gradient arena that PHASE_D_LOSS_A finalises is complete (its last writers: the
// BatchNorm backward of the top layer on the phase's stream, the top layer's weight gradient and conv5's on the weight-gradient
// stream).  For a caller that passed JCK_PHASE_LAZY_JOIN with PHASE_D_LOSS_A and starts the tail's all-reduce from `stream`.
extern "C" int jck_engine_order_after_tail(jck_engine* e, void* stream) {
  if (!e || !e->bound || !stream) JCK_FAIL(JCK_E_ARG, "engine not bound / null stream");
  if (!e->evTail) return JCK_OK;                    // no second stream: the tail is final in the phase's stream order
  if (e->tail_on_side) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e->evW[e->T.NS - 1], 0));
  HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e->evTail, 0));
  return JCK_OK;
}
// A PHASE_D_REAL_FWD already enqueued for the next step is abandoned (its weights are about to change: a re-broadcast of the
// replica guard, load_model): `stream` waits for it - it may still be reading D's packed operands on the weight-gradient stream -
// and the next D phase computes D(real) itself.
extern "C" int jck_engine_drop_prefetch(jck_engine* e, void* stream) {
  if (!e || !e->bound) JCK_FAIL(JCK_E_ARG, "engine not bound");
  if (e->real_fwd_step >= 0 && e->pre_on_side && e->evReal && stream) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e->evReal, 0));
  e->real_fwd_step = -1;
  e->pre_on_side = false;
  return JCK_OK;
}
extern "C" long long jck_engine_grad_tail(const jck_engine* e, int net) {
  if (!e || !e->bound || net != 1 || e->family != 0 || e->batched != 3) return -1;
  return (long long)find(e->LD, CWN[e->T.NS - 1])->offset;
}

// Call after a device synchronisation (e.g. when the step scalars are read): a grid barrier of a resident launch that timed out -
// its workgroups were not all resident, which happens when another process or stream fills the same GPU - leaves invalid
// results; the error is reported here, the barrier state re-armed.
extern "C" int jck_engine_check(jck_engine* e) {
  if (!e || !e->bound || !e->gsync) return JCK_OK;
  if (!jck_grid_sync_error(e->gsync)) return JCK_OK;
  HIPCHK(hipMemset(e->gsync, 0, jck_grid_sync_bytes()));
  JCK_FAIL(JCK_E_HIP, "a grid barrier of the resident BatchNorm backward timed out: the launch needs every CU of the device to itself "
                      "(another process or stream on this GPU?); the step's results are invalid - set JCK_BN_RES=0 to use the three-launch form");
}
extern "C" const float* jck_engine_scalars(const jck_engine* e) { return e ? e->scal_out : nullptr; }
extern "C" const float* jck_engine_scalars_at(const jck_engine* e, int step) { return e ? e->scal2 + 8 * (step & 1) : nullptr; }

extern "C" int jck_engine_sample(jck_engine* e, const float* z, const int64_t* labels, int n, float* out_nchw, void* stream) {
  if (!e || !e->bound) JCK_FAIL(JCK_E_ARG, "engine not bound");
  if (n < 1 || n > e->B) JCK_FAIL(JCK_E_ARG, "sample: n must be in [1, batch]");
  hipStream_t st = (hipStream_t)stream;
  JCK_TRY(g_forward(e, z, labels, n, st));
  return jck_nhwc4_to_nchw(e->prec, e->fake_raw, out_nchw, n, TT.HW, st);
}

extern "C" const void* jck_engine_tensor(const jck_engine* e, const char* name, long long* numel) {
  if (!e || !e->bound || !name) return nullptr;
  const long long img = (long long)e->B * TT.HW * 4;
  struct { const char* n; const void* p; long long c; } tab[] = {
      {"fake", e->fake, img}, {"fake_raw", e->fake_raw, img}, {"real_noisy", e->real_noisy, img}, {"xhat", e->xhat, img},
      {"d_gx", e->d_gx, img}, {"prob", e->prob, e->B}, {"ds", e->ds, e->B}, {"norms", e->norms, e->B}, {"acc", e->acc, (long long)8 * e->acc_ld},
      {"d_y1", e->d_y[0], (long long)e->B * (TT.S / 2) * (TT.S / 2) * 64}, {"d_a4", e->d_a[TT.NS - 1], (long long)e->B * TT.FEAT},
      {"g_y1", e->g_y[0], (long long)e->B * TT.FEAT}, {"g_a4", e->g_a[TT.NS - 1], (long long)e->B * (TT.S / 2) * (TT.S / 2) * 64}};
  for (auto& t : tab)
    if (!strcmp(t.n, name)) { if (numel) *numel = t.c; return t.p; }
  return nullptr;
}
