/* jckgan.h - C ABI of libjckgan_hip.so: the MI355X (gfx950) replacement for the DCGAN / CGAN
 * training hot path of hy-vision-learning/jck-generation.
 *
 * The reference has no FFI layer of its own: its hot path is torch.nn modules + autograd
 * (model/DCGAN.py, model/CGAN.py, train/dcgan_trainer.py, train/cgan_trainer.py).  Every entry
 * point below names the reference call site (file:line under the reference root) whose ATen work it
 * replaces.  The Python side (jck-generation_amd/hipgan) binds these symbols with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes, `void* stream` = hipStream_t (NULL = default)
 *   - every function returns 0 on success, a negative JCK_E_* code on error; jck_last_error()
 *     returns the message of the calling thread's last failure
 *   - no allocation, no ownership transfer: outputs and workspaces are caller-allocated
 *   - asynchronous with respect to the host; thread-compatible (one stream per caller thread)
 *   - `prec` selects storage / arithmetic:
 *       JCK_PREC_BF16  activations, gradients, GEMM operands bf16 in HBM; v_mfma_f32_16x16x32_bf16, fp32 accumulate (fast)
 *       JCK_PREC_F32   everything fp32 in HBM; v_mfma_f32_16x16x4_f32 = exact fp32 products + accumulation      (parity)
 *     "T" below means bf16 (2 bytes) or float according to `prec`
 *   - activations are NHWC; 3-channel images are stored with 4 channels (4th = 0)
 *   - a stride-2 stage is described by its BIG side [N,Hb,Wb,Cb] and SMALL side [N,Hb/2,Wb/2,Cs];
 *     Conv2d (D) maps big->small, ConvTranspose2d (G) maps small->big; both keep the weight as
 *     [Cs][Cb][4][4] fp32 (Conv2d.weight [Cout,Cin,4,4] / ConvTranspose2d.weight [Cin,Cout,4,4])
 */
#ifndef JCKGAN_H
#define JCKGAN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define JCK_PREC_BF16 0
#define JCK_PREC_F32 1

#define JCK_OK 0
#define JCK_E_ARG (-1)      /* bad argument / unsupported shape */
#define JCK_E_HIP (-2)      /* HIP runtime error (see jck_last_error) */
#define JCK_E_WS (-3)       /* workspace too small */

const char* jck_last_error(void);
int jck_version(void);
/* rows of a packed weight matrix for `c` output channels (tile padding): 16, 64 or a multiple of 128 */
int jck_pad_rows(int c);
/* channels per pixel as stored in HBM: 3 -> 4, otherwise c (must be a power of two) */
int jck_pad_chan(int c);

/* ---- weight packing: fp32 parameter -> GEMM operand of element type T ---------------------------------------------
 * down : [pad_rows(Cs)][16*pad_chan(Cb)]      operand of Conv2d forward / ConvTranspose2d dgrad
 * up   : [4][pad_rows(Cb)][4*Cs]              operand of ConvTranspose2d forward / Conv2d dgrad
 * g1   : [16*Co][CiPad]                       G.conv1, ConvTranspose2d(k4,s1,p0) on a 1x1 input (model/DCGAN.py:42)
 * head : float[16*C]                          D.conv5, Conv2d(512,1,k4,s1,p0) as a dot product (model/DCGAN.py:26) */
size_t jck_packed_bytes(int prec, long long elems);
int jck_pack_down(int prec, const float* w, int Cs, int Cb, void* wp, void* stream);
int jck_pack_up(int prec, const float* w, int Cs, int Cb, void* wp, void* stream);
int jck_pack_g1(int prec, const float* w, int Ci, int Co, int CiPad, void* wp, void* stream);
int jck_pack_head(const float* w, int C, float* wp, void* stream);

/* ---- convolution-shaped products (replace aten::convolution / convolution_backward) -----------------------
 * stats (optional): partial per-channel sums for the BatchNorm that follows, float[slots][2][C] written without
 * atomics by the GEMM epilogue; *stats_slots receives the slot count to hand to jck_bn_finalize.  The buffer must
 * hold jck_stats_floats(output pixels, C, nyrep) floats (nyrep = 16 for jck_g1_fwd, else 1). */
size_t jck_stats_floats(long long pixels, int C, int nyrep);
/* small = Conv2d_k4s2p1(big)            model/DCGAN.py:10-22 forward; dgrad of model/DCGAN.py:46-58 */
int jck_conv_down(int prec, const void* big, const void* w, void* small_out, float* stats, int* stats_slots,
                  int N, int Hb, int Wb, int Cb, int Cs, void* stream);
/* big = ConvTranspose2d_k4s2p1(small)   model/DCGAN.py:46-58 forward; dgrad of model/DCGAN.py:10-22.  epi_tanh=1 fuses model/DCGAN.py:66 */
int jck_conv_up(int prec, const void* small_in, const void* w, void* big_out, float* stats, int* stats_slots,
                int epi_tanh, int N, int Hs, int Ws, int Cs, int Cb, void* stream);
/* grad[Cs][Cb][4][4] (+)= sum small (x) gather(big)     weight gradient of either layer kind */
size_t jck_conv_wgrad_ws_bytes(int N, int Hb, int Wb, int Cb, int Cs);
int jck_conv_wgrad(int prec, const void* small_side, const void* big_side, float* ws, size_t ws_bytes, float* grad,
                   int accumulate, int N, int Hb, int Wb, int Cb, int Cs, void* stream);
/* out[B][16*Co] = z[B][CiPad] x W   (NHWC [B,4,4,Co]); stats over Co channels   model/DCGAN.py:42,62 */
int jck_g1_fwd(int prec, const void* z, const void* w, void* out, float* stats, int* stats_slots, int B, int CiPad,
               int Co, void* stream);
size_t jck_g1_wgrad_ws_bytes(int B, int CiPad, int Co);
int jck_g1_wgrad(int prec, const void* z, const void* dy, float* ws, size_t ws_bytes, float* grad, int accumulate, int B,
                 int Ci, int CiPad, int Co, void* stream);

/* ---- BatchNorm2d (training mode) + ReLU / LeakyReLU  (model/DCGAN.py:11-24,43-56; aten::native_batch_norm*) ----
 * aux: float[4*C] = scale(gamma*invstd) | shift | mean | invstd, produced by jck_bn_finalize */
int jck_bn_finalize(const float* stats, int slots, float count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                    float* aux, int C, void* stream);
int jck_bn_act_fwd(int prec, const void* y, const float* aux, float slope, void* a, long long rows, int C, void* stream);
/* sums: scratch of jck_bn_bwd_ws_floats(C) floats (first 2*C = final sum g_z, sum g_z*xhat; no zeroing needed);
 * g_y may alias g_a; dgamma/dbeta (optional) are accumulated into */
size_t jck_bn_bwd_ws_floats(int C);
int jck_bn_act_bwd(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y,
                   float* dgamma, float* dbeta, long long rows, int C, void* stream);
/* Grouped forms: `groups` independent BatchNorm batches stored back to back - [groups][rows][C] tensors, [groups][slots]
 * statistic slots, [groups][4C] aux, [groups][2C] (mean, unbiased var) records, [groups][jck_bn_bwd_ws_floats(C)] backward
 * workspace.  For the D passes that share weights and run as ONE conv launch (train/dcgan_trainer.py:162,173,118); every
 * group is normalised with its own batch statistics, exactly as in the separate passes.  Only groups < grad_groups add to
 * dgamma / dbeta. */
int jck_bn_finalize_grouped(const float* stats, int slots_per_group, float count, const float* gamma, const float* beta, float eps,
                            float* aux, float* stat_out, int C, int groups, void* stream);
/* jck_bn_finalize_grouped + jck_bn_act_fwd_grouped as ONE launch where the statistics rows are few (each workgroup owns a 64-channel
 * slice and sums that slice's rows itself; otherwise the two launches are issued): a = act(scale * y + shift) with the batch
 * statistics of each group, aux [groups][4C] and stat_out [groups][2C] (may be NULL) written as jck_bn_finalize_grouped writes
 * them; running_mean / running_var / nbt (may be NULL; groups == 1 only) updated as jck_bn_finalize updates them
 * (aten::native_batch_norm + the activation, model/DCGAN.py:30-33,62-65). */
int jck_bn_fwd(int prec, const void* y, const float* stats, int slots_per_group, float count, const float* gamma, const float* beta,
               float eps, float slope, void* a, float* aux, float* stat_out, float* running_mean, float* running_var, int64_t* nbt,
               float momentum, long long rows_per_group, int C, int groups, void* stream);
int jck_bn_act_fwd_grouped(int prec, const void* y, const float* aux, float slope, void* a, long long rows_per_group, int C,
                           int groups, void* stream);
int jck_bn_act_bwd_grouped(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y,
                           float* dgamma, float* dbeta, long long rows_per_group, int C, int groups, int grad_groups, void* stream);
/* Resident form of jck_bn_act_bwd_grouped (round 4; csrc/bnres.hpp): ONE launch; one workgroup per CU keeps its share of a
 * group's (g_a, y) in registers across a grid barrier, so every tensor byte is read once (HBM passes 5 -> 3, launches 3 -> 1).
 * Same arguments and results (summation order differs: fp32 rounding), plus sync_ws = jck_grid_sync_bytes() bytes of device
 * memory zeroed ONCE by the caller and used by no two launches at the same time (the launch occupies every CU: one stream).
 * Falls back to jck_bn_act_bwd_grouped when the layer does not fit the chip's register file, for fp32, for C < 64, with
 * sync_ws == NULL or JCK_BN_RES=0.  jck_grid_sync_error: 1 if a barrier timed out (results invalid; re-zero sync_ws). */
size_t jck_grid_sync_bytes(void);
int jck_grid_sync_error(const void* sync_ws);
int jck_bn_act_bwd_res(int prec, const void* g_a, const void* y, const float* aux, float slope, float* sums, void* g_y,
                       float* dgamma, float* dbeta, long long rows_per_group, int C, int groups, int grad_groups, void* sync_ws,
                       void* stream);
/* jck_conv_down / jck_conv_up with the forward statistics laid out per BatchNorm group of `group_images` images
 * (N % group_images == 0): *stats_slots rows, the first *stats_slots / (N / group_images) of them belong to group 0, and so
 * on - what jck_bn_finalize_grouped reads.  Large launches write one row per (workgroup, group) instead of one per (tile, wave). */
int jck_conv_down_grouped(int prec, const void* big, const void* w, void* small_out, float* stats, int* stats_slots, int N, int Hb,
                          int Wb, int Cb, int Cs, int group_images, void* stream);
int jck_conv_up_grouped(int prec, const void* small_in, const void* w, void* big_out, float* stats, int* stats_slots, int N, int Hs,
                        int Ws, int Cs, int Cb, int group_images, void* stream);
/* ---- images, noise, heads, loss ----------------------------------------------------------------------------- */
/* out NHWC4 T = keep*img + mix*noise (NCHW fp32 inputs; noise may be NULL)   train/dcgan_trainer.py:157-160 */
int jck_img_prep(int prec, const float* img_nchw, const float* noise_nchw, float keep, float mix, void* out, int N, int HW,
                 void* stream);
/* Device-resident input pipeline: gathers B images by index from a uint8 dataset [Ntot][3][Hs][Ws] kept in HBM (idx NULL:
 * the first B) and applies the reference's transform chain on the fly - Resize(2x) exactly as PIL's bilinear upscale
 * (horizontal then vertical pass, each rounded to uint8), ToTensor, Normalize(0.5,0.5)
 * (preprocess/dcgan_data_preprocessor.py:38-43) - then the instance-noise mix keep*x + mix*noise
 * (train/dcgan_trainer.py:160).  out_nhwc4 [B][2Hs][2Ws][4] (element type of prec) and/or out_nchw fp32 [B][3][2Hs][2Ws]
 * (the transformed image without noise); either may be NULL. */
int jck_img_prep_u8(int prec, const unsigned char* data, const int64_t* idx, const float* noise, float keep, float mix,
                    void* out_nhwc4, float* out_nchw, int B, int Hs, int Ws, void* stream);
/* Evaluation branch (train/dcgan_trainer.py:202-206, train/cgan_trainer.py:227-231): out = (resize(pre_scale*in + pre_shift,
 * [OH,OW]) - mean[c]) / std[c], bilinear with align_corners = False exactly as aten::upsample_bilinear2d (what
 * torchvision.transforms.functional.resize does to a tensor; an upscale, so antialiasing is moot).  NCHW fp32 in / out;
 * mean / std: device float[C]. */
int jck_resize_norm(const float* in, float* out, int N, int C, int H, int W, int OH, int OW, float pre_scale, float pre_shift,
                    const float* mean, const float* stdv, void* stream);
int jck_nhwc4_to_nchw(int prec, const void* in, float* out_nchw, int N, int HW, void* stream);
/* out = keep*x + mix*noise for an NHWC4 x                                      train/dcgan_trainer.py:171 */
int jck_axpy_noise(int prec, const void* x, const float* noise_nchw, float keep, float mix, void* out, int N, int HW,
                   void* stream);
/* x_hat = alpha*a + (1-alpha)*b                                                 train/dcgan_trainer.py:111-112 */
int jck_interp(int prec, const void* a, const void* b, const float* alpha, void* out, int N, int HW, void* stream);
/* jck_axpy_noise (noise_nchw given) or jck_axpy_noise_rng (rng given; exactly one of the two) followed by
 * jck_interp(real, out, alpha) -> xhat, as one launch with the same results bit for bit (train/dcgan_trainer.py:171 then :111-113). */
int jck_mix_interp(int prec, const void* x, const float* noise_nchw, const unsigned* rng, int tensor_id, float keep, float mix, void* out,
                   const void* real, const float* alpha, void* xhat, int N, int HW, void* stream);
/* scal[slot*scal_ld + n] = (||g[n]||_2 - 1)^2 (plain store per image; the caller sums the row in a fixed order - no float
 * atomics, so the logged penalty is bitwise reproducible); norms[n] optional     train/dcgan_trainer.py:125-126 */
int jck_gp_norm(int prec, const void* g, int N, int HW, float* scal, int slot, int scal_ld, float* norms, void* stream);
/* g_out = scale * g * (1 - y^2)                                                 tanh backward, model/DCGAN.py:66 */
int jck_tanh_bwd(int prec, const void* g, const void* y, float scale, void* out, long long numel, void* stream);
/* D head: logit = <a4[n], wp>, p = sigmoid (model/DCGAN.py:34), BCELoss with the -100 clamp (train/dcgan_trainer.py:64,163);
 * mode 0: ds = dLoss/dlogit for mean BCE against `target`; scal[slot_loss*scal_ld + n] = loss_n; mode 1: ds = p(1-p) (GP
 * pass).  scal[slot_p*scal_ld + n] = p_n.  slot < 0 disables.  `scal` is a per-image table [slots][scal_ld >= B]: plain
 * stores, summed in a fixed order by the step tail (no float atomics anywhere on the path). */
int jck_head_fwd(int prec, const void* a4, const float* wp, const float* bias /* device scalar or NULL */, int B, int K,
                 float target, int mode, float* prob, float* ds, float* scal, int slot_loss, int slot_p, int scal_ld, void* stream);
/* G (<= 4) batches of B rows stacked in a4 / prob / ds (the real | fake | penalty groups of a batched D pass), each with its own
 * target, mode and scalar slots, in one launch; the scalar table is indexed by the row inside its group */
int jck_head_fwd_grouped(int prec, const void* a4, const float* wp, const float* bias, int B, int K, int G, const float* targets,
                         const int* modes, float* prob, float* ds, float* scal, const int* slot_loss, const int* slot_p,
                         int scal_ld, void* stream);
/* floats of workspace for the weight-gradient partial rows of jck_head_bwd / jck_head_bwd_conv (K = 16*C there) */
size_t jck_head_bwd_ws_floats(int K);
/* g_a4[n][k] = ds[n]*wp[k];  dwp[k] (+)= sum_n ds[n]*a4[n][k] through `ws` (partial rows summed in order: deterministic) */
int jck_head_bwd(int prec, const float* ds, const float* wp, const void* a4, int B, int K, void* g_a4, float* dwp,
                 int accumulate, float* ws, void* stream);
/* grad[1][C][4][4] (+)= dwp (packed (h,w,c) order) */
int jck_head_unpack_grad(const float* dwp, int C, float* grad, int accumulate, void* stream);
/* D.conv5 backward in one launch (DCGAN engine path): g_a4[n][k] = ds[n]*wp[k] (skipped when g_a4 is NULL) and
 * grad[c][t] += sum_n ds[n]*a4[n][t*C+c] into the PyTorch-layout gradient of conv5.weight [1][C][4][4] (skipped when grad
 * is NULL): 16 partial rows in `ws` (jck_head_bwd_ws_floats(16*C) floats), summed in order by a second small launch - no
 * float atomics.  Replaces aten::convolution_backward behind model/DCGAN.py:26 in train/dcgan_trainer.py:164,175,187. */
int jck_head_bwd_conv(int prec, const float* ds, const float* wp, const void* a4, int B, int C, void* g_a4, float* grad,
                      float* ws, void* stream);
/* the same over B rows plus, in the same launch, the input gradient alone of the B_more rows stored behind them (ds, a4 and g_a4
 * hold B + B_more rows; the weight gradient sums the first B only): the loss groups and the penalty group of one batched D pass. */
int jck_head_bwd_conv2(int prec, const float* ds, const float* wp, const void* a4, int B, int B_more, int C, void* g_a4, float* grad,
                       float* ws, void* stream);

/* ---- CGAN pieces (model/CGAN.py:79-162, train/cgan_trainer.py:173-213) ---------------------------------------------
 * Linear layers run on the gather-GEMM kernels as plain row-major products; our activation order is NHWC, so the
 * first permC*permHW weight columns (the flattened conv features, NCHW order in the reference, model/CGAN.py:119-120)
 * are permuted when packing and un-permuted when the gradient is written back. */
int jck_pack_linear(int prec, const float* w, int N, int K, int rows, int cols, int transpose, int permC, int permHW, void* wp,
                    void* stream);
/* out[B][NStore] = x[B][Kpad] * wp^T (+ bias); ksplit > 1: fp32 partial slabs [ksplit][B][NStore] for jck_linear_finish */
int jck_linear_fwd(int prec, const void* x, const void* wp, const float* bias, void* out, int B, int Kpad, int N, int NStore,
                   int ksplit, void* stream);
/* h = sum of slabs + bias; hd = h * mask * scale (nn.Dropout, model/CGAN.py:105); h or hd may be NULL */
int jck_linear_finish(int prec, const float* slab, int Z, const float* bias, const float* mask, float scale, void* h, void* hd,
                      int B, int N, void* stream);
size_t jck_linear_wgrad_ws_bytes(int B, int Kpad, int N);
int jck_linear_wgrad(int prec, const void* gy, int ldgy, const void* x, int Kpad, float* ws, size_t ws_bytes, float* gradp,
                     int accumulate, int B, int N, void* stream);
int jck_unperm_linear_grad(const float* gp, int N, int K, int ldp, int permC, int permHW, float* grad, int accumulate, void* stream);
/* label path: e = LeakyReLU(Linear(100,200)(onehot.float())) written into columns [col0, col0+NO) of cbuf (model/CGAN.py:111) */
int jck_label_embed_fwd(int prec, const int64_t* labels, const float* W, const float* b, float slope, int B, int NI, int NO,
                        void* cbuf, int ld, int col0, float* pre, void* stream);
int jck_label_embed_bwd(int prec, const void* gc, int ld, int col0, const float* pre, const int64_t* labels, float slope, int B,
                        int NI, int NO, float* dW, float* db, void* stream);
/* `_tiled`: the B rows are label_period-row batches stacked on top of each other that share one [label_period][NI] label
 * tensor - the real | fake | penalty groups of a CGAN step (train/cgan_trainer.py:181-203); label_period 0 = one label row per row */
int jck_label_embed_fwd_tiled(int prec, const int64_t* labels, const float* W, const float* b, float slope, int B, int NI, int NO,
                              void* cbuf, int ld, int col0, float* pre, int label_period, void* stream);
int jck_label_embed_bwd_tiled(int prec, const void* gc, int ld, int col0, const float* pre, const int64_t* labels, float slope,
                              int B, int NI, int NO, float* dW, float* db, int label_period, void* stream);
/* torch.cat([flatten(a4), e], 1) (model/CGAN.py:117-120) and its backward split */
int jck_concat_rows(int prec, const void* a4, int K0, void* cbuf, int ld, int B, void* stream);
int jck_split_rows(int prec, const void* gc, int ld, int K0, void* ga4, int B, void* stream);
int jck_dropout(int prec, const void* x, const float* mask, float scale, void* y, long long n, void* stream);
int jck_colsum(int prec, const void* g, int B, int N, int ld, float* db, void* stream);
int jck_sum_vec(const float* x, int n, float* out, void* stream);
/* G input [z | one-hot] (model/CGAN.py:154-155) */
int jck_cgan_z(int prec, const float* z, const int64_t* labels, int B, int NZ, int NL, int CiPad, void* out, void* stream);
/* back-propagated gradient penalty (train/cgan_trainer.py:200-203): u = dL/dg, second-order BatchNorm and head terms;
 * the closed form is derived and checked against autograd in tests/test_gp_double_backward_math.py */
int jck_gp_grad(int prec, const void* g, const float* norms, float coef, int N, int HW, void* u, void* stream);
/* ws: float[B + jck_head_bwd_ws_floats(K)] */
int jck_gp_head2(int prec, const void* ughd, const float* w2, const float* prob, int B, int K, float* rs, float* dw2, float* ws,
                 void* stream);
size_t jck_bn2_ws_floats(int C);
int jck_bn2_vchain(int prec, const void* v, const void* y, const void* gy, const float* aux, const float* s1, const float* gamma,
                   float slope, float* ws, void* u, void* xdir, float* dgamma, long long rows, int C, void* stream);
int jck_bn2_reverse(int prec, const void* ua, const void* y, const void* xdir, const float* aux, const float* gamma,
                    const float* vsums, float slope, float* ws, void* uy, float* dgamma, float* dbeta, long long rows, int C,
                    void* stream);

/* ---- optimiser (torch.optim.Adam as built at train/dcgan_trainer.py:61-62) over a flat fp32 arena ---------- */
int jck_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
             int step, float grad_scale, void* stream);

/* ---- whole-step engine (train/dcgan_trainer.py:155-189 as one native schedule) --------------------------- */
typedef struct jck_engine jck_engine;
/* family 0 = DCGAN, 1 = CGAN.  Layout queries let the host build flat parameter arenas with the reference's state-dict order. */
int jck_engine_create(jck_engine** out, int family, int prec, int batch);
/* image_size 64 = the reference's nets (model/DCGAN.py:10-27,42-59); 128 = one more stride-2 stage at the deep end (D 3-64-128-
 * 256-512-1024-1, G 100-1024-...-64-3; DCGAN only) for BASELINE.json configs[4] - no reference behaviour exists for it. */
int jck_engine_create_sized(jck_engine** out, int family, int prec, int batch, int image_size);
int jck_engine_image_size(const jck_engine*);
/* layout queries of a created engine (its own image size); same meaning as the family-keyed ones below */
int jck_engine_num_tensors_of(const jck_engine*, int net);
int jck_engine_tensor_info_of(const jck_engine*, int net, int idx, char* name, int name_cap, int* kind, long long* offset,
                              long long* numel, int* shape4);
long long jck_engine_arena_numel_of(const jck_engine*, int net, int which);
void jck_engine_destroy(jck_engine*);
int jck_engine_num_tensors(int family, int net /*0=G,1=D*/);
/* kind: 0 = parameter, 1 = running_mean, 2 = running_var ; name buffer gets the state-dict key */
int jck_engine_tensor_info(int family, int net, int idx, char* name, int name_cap, int* kind, long long* offset,
                           long long* numel, int* shape4);
size_t jck_engine_workspace_bytes(const jck_engine*);
/* arenas: params/grads/m/v are flat fp32 of jck_engine_arena_numel(net, 0); bn = running stats arena (kind 1,2);
 * nbt = int64[4] per net */
long long jck_engine_arena_numel(int family, int net, int which /*0 params, 1 bn buffers*/);
int jck_engine_bind(jck_engine*, void* workspace, size_t ws_bytes, float* g_params, float* g_grads, float* g_m, float* g_v,
                    float* g_bn, int64_t* g_nbt, float* d_params, float* d_grads, float* d_m, float* d_v, float* d_bn,
                    int64_t* d_nbt);
/* re-derive the bf16 GEMM operand planes from the fp32 parameters (after init / load_state_dict) */
int jck_engine_repack(jck_engine*, int net, void* stream);
/* phases of one step; between them the host may all-reduce the grads arenas (data parallel).
 *   PHASE_D_LOSS : zero D grads; D(real'), D(fake): forward+backward; G forward   (dcgan_trainer.py:155-176)
 *   PHASE_D_GP   : gradient-penalty pass (value only in DCGAN)                     (:178-179)
 *   PHASE_D_STEP : Adam on D, repack D                                             (:180)
 *   PHASE_G_LOSS : zero G grads; D(fake) with the new D; backward into G           (:182-188)
 *   PHASE_G_STEP : Adam on G, repack G, finalise the scalars                       (:189) */
#define JCK_PHASE_D_LOSS 0
#define JCK_PHASE_D_GP 1
#define JCK_PHASE_D_STEP 2
#define JCK_PHASE_G_LOSS 3
#define JCK_PHASE_G_STEP 4
/* PHASE_D_LOSS = PHASE_D_REAL (:155-165, needs no G) followed by PHASE_D_FAKE (:168-176 + start of the penalty pass).
 * Issued separately, D_REAL of step k+1 may run on another stream beside the G phase of step k: it uses its own
 * activation set, and scalar accumulators / BatchNorm records are double-buffered on the parity of `step`. */
#define JCK_PHASE_D_REAL 5
#define JCK_PHASE_D_FAKE 6
/* PHASE_D_REAL_FWD (DCGAN, batched schedule): the forward half of D(real) of step k+1 only - input transform, instance noise,
 * conv stack with its BatchNorm statistics (:160-162) - issued with the inputs of step k+1 between PHASE_G_LOSS and
 * PHASE_G_STEP of step k, i.e. while G's gradient all-reduce of step k is in flight (data parallel): it needs D's weights
 * (final since PHASE_D_STEP of step k) and nothing of G.  PHASE_D_LOSS of step k+1 then skips that part and runs
 * [fake | penalty] as one 2B forward behind it; everything else (heads, losses, the 3B backward) is unchanged, so results
 * are bitwise those of the plain order.  Returns JCK_E_ARG when the engine's schedule has no such split (CGAN, per-pass). */
#define JCK_PHASE_D_REAL_FWD 7
/* PHASE_D_LOSS = PHASE_D_LOSS_A + PHASE_D_LOSS_B (DCGAN, batched schedule; data parallel).  After _A the tail of D's gradient
 * arena [jck_engine_grad_tail() .. end) - conv4.weight, norm4.*, conv5.weight, 76 % of its bytes - is final in stream order:
 * the caller starts its all-reduce, issues _B (the remaining ~0.5 ms of the backward pass), then all-reduces the head of the
 * arena.  Same kernels in the same order on every tensor: bitwise the results of PHASE_D_LOSS. */
#define JCK_PHASE_D_LOSS_A 8
#define JCK_PHASE_D_LOSS_B 9
/* The gradient penalty alone, for a caller that keeps the reference's loop on the HIP modules (compute_gradient_penalty,
 * train/dcgan_trainer.py:110-127, train/cgan_trainer.py:114-131): real_nchw = real_data and noise_real = fake_data, both
 * [B,3,S,S] fp32 taken as they are, alpha [B] (CGAN: labels, drop_mask[2]).  Afterwards jck_engine_tensor("norms") holds the
 * per-image gradient norms - the penalty is mean((norm - 1)^2) - and, CGAN, D's gradient arena holds d(penalty)/d(theta_D)
 * (cleared first; lambda = 1): the double backward the reference obtains with create_graph=True, in closed form
 * (hipgan/functional.py: gradient_penalty). */
#define JCK_PHASE_GP_ONLY 10
/* OR-ed into PHASE_D_LOSS / PHASE_D_GP (CGAN): the caller reads nothing of D's gradient arena before PHASE_D_STEP (no
 * all-reduce in between), so the phase need not wait for the weight-gradient stream before it returns.  PHASE_D_STEP then
 * runs Adam over everything but the bottom conv weight - the last product of that stream - while it finishes, and that one
 * tensor behind it.  Any other phase that follows joins first.  Same kernels on the same values: bitwise the plain order. */
#define JCK_PHASE_LAZY_JOIN 0x100
/* OR-ed into any phase: the BatchNorm backward launches of THIS call take the three-launch form instead of the resident
 * (grid-barrier) one.  A resident launch needs every CU of the device: while a collective is in flight (data parallel, N > 1:
 * RCCL's kernel holds CUs until every peer has arrived) its missing workgroups could not be placed and the placed ones would
 * spin at the barrier until its time bound - hipgan/engine.py sets the flag on the phases it issues between the start of an
 * all-reduce and the wait for it (train/dcgan_trainer.py:180,189 are the exchange points).  Same sums in another order: held to
 * the oracle like the resident form. */
#define JCK_PHASE_NO_RESIDENT 0x200
typedef struct jck_step_inputs {
  const float* real_nchw; /* [B,3,64,64] fp32 */
  const float* noise_real; /* [B,3,64,64] N(0,1); NULL (with noise_fake NULL): drawn inside the kernels (jck_engine_set_noise_seed) */
  const float* z;          /* [B,100] N(0,1); NULL: the engine's own draw for this step (jck_engine_set_step, jck_step_rng) */
  const float* noise_fake; /* [B,3,64,64] N(0,1) */
  const float* alpha;      /* [B] U[0,1); NULL: the engine's own draw */
  float lr;
  float grad_scale;        /* 1/world_size when grads were SUM-all-reduced, else 1 */
  int step;                /* 1-based optimiser step (Adam bias correction) */
  /* family 1 (CGAN) only: */
  const int64_t* labels;   /* [B,100] one-hot int64 (preprocess/cgan_data_preprocessor.py:11-16) */
  const float* drop_mask[4]; /* [B,256] 0/1 keep masks of nn.Dropout(0.25) for the 4 D passes (real, fake, GP, G phase); all four
                              * NULL: the engine's own draws.  Masks 0..2 back to back in memory let the head run once over 3B rows */
  /* device-resident dataset (optional; used instead of real_nchw when real_u8 != NULL): uint8 [Ntot,3,32,32] + the batch's
   * indices int64 [B]; the step applies the input transform itself (jck_img_prep_u8) */
  const unsigned char* real_u8;
  const int64_t* real_idx;
} jck_step_inputs;
int jck_engine_phase(jck_engine*, int phase, const jck_step_inputs* in, void* stream);
/* device pointer to float[8]: loss_d, loss_g, D(x), D(G(z))_1, D(G(z))_2, gp, loss_real, loss_fake (valid after PHASE_G_STEP) */
/* first element of the gradient-arena tail that PHASE_D_LOSS_A finalises (net 1 = D); -1 when the schedule has no such split */
long long jck_engine_grad_tail(const jck_engine*, int net);
/* PHASE_D_LOSS_A | JCK_PHASE_LAZY_JOIN (DCGAN, data parallel): the phase returns without making its stream wait for the weight-
 * gradient stream (the stall cost more than the early all-reduce hid); the caller starts the tail's all-reduce from ANOTHER stream
 * that it first passes here - it then waits for the tail's last writers on both engine streams. */
int jck_engine_order_after_tail(jck_engine*, void* stream);
/* abandons a PHASE_D_REAL_FWD enqueued ahead for the next step (D's weights are about to be overwritten from outside the step:
 * the replica guard's re-broadcast, load_model): `stream` waits for that forward, the next D phase recomputes it */
int jck_engine_drop_prefetch(jck_engine*, void* stream);
/* after a device synchronisation: JCK_E_HIP if a grid barrier of a resident launch (jck_bn_act_bwd_res) timed out since the last
 * call - that launch went on with incomplete sums, so the step's results are invalid; the engine's optimiser phases read the same
 * word on the device and leave parameters and Adam moments untouched while it is set.  The call re-arms the barrier state. */
int jck_engine_check(jck_engine*);
const float* jck_engine_scalars(const jck_engine*);
const float* jck_engine_scalars_at(const jck_engine*, int step);   /* buffer of the given (1-based) step's parity */
/* G forward only (train/dcgan_trainer.py:199-200, train-mode BN: running stats move); out NCHW fp32 [n,3,64,64] */
int jck_engine_sample(jck_engine*, const float* z, const int64_t* labels /* family 1 */, int n, float* out_nchw, void* stream);
/* debug / parity access to internal NHWC tensors: name in {"fake","real_noisy",...}; returns device ptr or NULL */
const void* jck_engine_tensor(const jck_engine*, const char* name, long long* numel);

/* ---- evaluation branch: the metric network (reference metrics.py:46-51,80-94: torchvision inception_v3 with a
 * Linear(2048,100) head, eval mode) as a chain of NHWC fp32 kernels; jck-generation_amd/inception.py holds the topology and
 * the local-weights loader.  conv2d: out[n,oy,ox, out_coff + co] = act(scale[co] * sum_{kh,kw,ci} x[n, oy*SH-PH+kh, ox*SW-PW+kw, ci]
 * * w_kc[(kh*KW + kw)*Cin + ci][co] + shift[co]) - eval-mode BatchNorm folded into scale / shift (NULL: 1 / 0), exact-fp32
 * MFMA, the output written into a channel slice of a [.., out_cstride] tensor (Inception concatenations need no copy).
 * pool2d mode 0: max (aten::max_pool2d, no padding value enters), mode 1: average with count_include_pad = True
 * (F.avg_pool2d default).  mean_cov: column means and the unbiased covariance of x[N][D] in fp64 (np.mean / np.cov of
 * metrics.py:120-126 on the device). */
int jck_conv2d_nhwc_f32(const float* x, const float* w_kc, const float* scale, const float* shift, float* out, int N, int H, int W,
                        int Cin, int KH, int KW, int SH, int SW, int PH, int PW, int Cout, int out_cstride, int out_coff, int relu,
                        void* stream);
int jck_pool2d_nhwc_f32(const float* x, float* out, int N, int H, int W, int C, int k, int stride, int pad, int mode, int out_cstride,
                        int out_coff, void* stream);
int jck_global_avgpool_nhwc_f32(const float* x, float* out, int N, int HW, int C, void* stream);
int jck_nchw_to_nhwc_f32(const float* x, float* out, int N, int C, int H, int W, void* stream);
int jck_mean_cov_f64(const float* x, double* mean, double* cov, int N, int D, void* stream);

/* Per-step optimiser scalars into device memory (so that a captured graph of the step has no per-step kernel argument), and
 * hipGraph capture of a sequence of jck_engine_phase calls: begin -> phases on `stream` (not the default stream) -> end
 * returns an executable graph; launch replays it.  The jck_step_inputs pointers are baked: keep the buffers in place,
 * refresh their contents, keep one graph per step parity, call jck_engine_set_step before every launch. */
int jck_engine_set_step(jck_engine*, int step, float lr, void* stream);
/* Instance noise drawn INSIDE the image kernels (steps whose jck_step_inputs.noise_real / noise_fake are NULL): Philox4x32-10
 * keyed by `seed`, counter = (pixel, tensor, optimiser step), Box-Muller normals - no 25 MB noise tensor per step.  The *_rng
 * entry points are the per-op forms (rng: device uint32[4] = {seed lo, seed hi, step, 0}; tensor_id separates real / fake). */
int jck_engine_set_noise_seed(jck_engine*, unsigned long long seed);
/* The step's small random inputs as jck_engine_set_step draws them when jck_step_inputs carries none (z, alpha, CGAN's
 * Dropout keep masks NULL): z [nz] ~ N(0,1) (train/dcgan_trainer.py:168), alpha [nalpha] ~ U[0,1) (:111), masks [nmask] in {0,1}
 * with P(1) = keep_p (model/CGAN.py:105); Philox4x32-10, counter = (index/4, tensor id, step), key = seed.  hp: 8 floats scratch. */
int jck_step_rng(float* hp, int step, unsigned long long seed, float* z, long long nz, float* alpha, long long nalpha,
                 float* masks, long long nmask, float keep_p, void* stream);
int jck_img_prep_rng(int prec, const float* img_nchw, const unsigned* rng, int tensor_id, float keep, float mix, void* out, int N, int HW,
                     void* stream);
int jck_img_prep_u8_rng(int prec, const unsigned char* data, const int64_t* idx, const unsigned* rng, int tensor_id, float keep,
                        float mix, void* out_nhwc4, int B, int Hs, int Ws, void* stream);
int jck_axpy_noise_rng(int prec, const void* x, const unsigned* rng, int tensor_id, float keep, float mix, void* out, int N, int HW,
                       void* stream);
int jck_engine_capture_begin(jck_engine*, void* stream);
int jck_engine_capture_end(jck_engine*, void* stream, void** graph_exec);
int jck_engine_capture_abort(jck_engine*, void* stream);
int jck_graph_launch(void* graph_exec, void* stream);
void jck_graph_destroy(void* graph_exec);

/* Kernel-selection knob (same names as the JCK_<KEY> environment presets, lower case: "igemm_256", "wgrad_gt", ...): lets
 * one process A/B two variants on one device and lets a test force a variant at a small shape.  Unknown key -> JCK_E_ARG. */
int jck_tune(const char* key, int value);
/* per-launch HIP-event timing of the MFMA kernels (bench.py roofline leg).  enable(1) ... run ... collect():
 * per (kernel variant, HIP stream the launches ran on): launches, total milliseconds, total algorithmic FLOPs, total algorithmic
 * bytes (streaming kernels), the stream.  A variant launched on two streams comes back as two rows.  Returns the number of rows. */
int jck_prof_enable(int on);
int jck_prof_collect(int cap, const char** name_out, int* count_out, double* ms_out, double* flops_out, double* bytes_out,
                     void** stream_out);

/* debug probe: lane l of one wave returns the 8 elements wgrad's transposed LDS read hands it from a
 * [32][ld] 16-bit tile: out[l*8+j] must equal in[(8*(l>>4)+j)*ld + (l&15)] */
int jck_debug_tr_read(const void* in, int ld, void* out, void* stream);
/* development aid: the resident BatchNorm launches write [256][8] s_memrealtime stamps (uint64) of their workgroup leaders to buf
 * (device memory; NULL switches it off) */
int jck_debug_bnres_stamps(void* buf);
/* development probe: per-wave s_memtime totals {wait+barrier, DMA issue, LDS reads+MFMA, whole kernel} of the last
 * weight-gradient launch made with JCK_WGRAD_STAMP=1 (n = number of 64-bit values to copy, 4 per wave, 8 waves per workgroup
 * slot, first 1024 workgroups); synchronises the device */
int jck_debug_wgrad_stamps(unsigned long long* out, int n);

/* ---- RCCL gradient all-reduce {init, enqueue, wait} ------------------------------------------------------------------------
 * The reference's multi-GPU form is DistributedDataParallel around G and D: what optimizer_d.step() / optimizer_g.step() consume
 * (train/dcgan_trainer.py:180,189, train/cgan_trainer.py:204,212) are gradients averaged over the ranks.  A network's gradients
 * are one flat fp32 arena here, so the exchange is one SUM all-reduce per arena (or slice); the 1/world factor goes into jck_adam's
 * grad_scale.  The collective runs on the communicator's OWN stream: enqueue orders it behind everything `producer_stream` holds
 * at the time of the call, wait makes `consumer_stream` wait for it on the device; the host never blocks.  Up to 8 tickets in
 * flight.  librccl is resolved at the first call (a copy already in the process - PyTorch's - is preferred; JCK_RCCL_LIB names
 * another), so a single-GPU process never maps it.
 *   rank 0: jck_comm_unique_id(id); every rank receives the 128 bytes over any host channel; every rank: jck_comm_create (a
 *   collective call, with its device current). */
typedef struct jck_comm jck_comm;
#define JCK_COMM_ID_BYTES 128
int jck_comm_unique_id(unsigned char* id128);
int jck_comm_create(jck_comm** out, const unsigned char* id128, int world, int rank);
int jck_comm_world(const jck_comm*);
int jck_comm_allreduce_enqueue(jck_comm*, float* buf, size_t count, void* producer_stream, int* ticket);
int jck_comm_wait(jck_comm*, int ticket, void* consumer_stream);
int jck_comm_destroy(jck_comm*);

#ifdef __cplusplus
}
#endif
#endif
