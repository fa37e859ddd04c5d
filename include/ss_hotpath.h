/* ss_hotpath.h -- C ABI of the MI355X (gfx950) Silent-Speech per-clip hot path.
 *
 * The reference (davdwan21/Silent-Speech) has no FFI: its path sits behind a Python nn.Module
 * (train_model_official.py:253-310), NumPy helpers (record_landmarks_official.py:52-118,
 * live_infer_official.py:141-187) and torch's optimiser/loss objects
 * (train_model_official.py:403-405, 433-439).  This header is the boundary a maintainer would
 * bind with ctypes: plain device pointers and sizes, no torch types.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless a comment says otherwise; buffers are borrowed for
 *    the duration of the call (asynchronously: until the work enqueued on `stream` has finished);
 *  - no entry point allocates, frees or synchronises; workspaces are passed in, so every call is
 *    capturable into a hipGraph;
 *  - `stream` is a hipStream_t (0 = the default stream); calls are re-entrant per stream;
 *  - return value: 0 (SS_OK) or a negative ss_status; the Python shim raises RuntimeError on != 0;
 *  - all floating-point buffers are fp32 row-major unless stated; gate order everywhere is
 *    torch's r | z | n.
 */
#ifndef SS_HOTPATH_H
#define SS_HOTPATH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ss_stream_t; /* hipStream_t */

enum ss_status {
  SS_OK = 0,
  SS_ERR_ARG = -1,         /* null pointer / non-positive size / inconsistent flags */
  SS_ERR_LAUNCH = -2,      /* hipGetLastError() after the launch was not hipSuccess */
  SS_ERR_UNSUPPORTED = -3  /* shape outside what the kernels are built for (e.g. hidden size) */
};

/* ABI version, bumped on any signature change. */
int ss_abi_version(void);
const char* ss_status_string(int status);

/* ---- a1: per-frame landmark feature fuse -------------------------------------------------
 * replaces extract_feature + mouth_width_px (record_landmarks_official.py:52-100;
 * live_infer_official.py:141-169).
 * lm      (B,T,K,2) MediaPipe-normalised x,y of the K selected landmarks
 * reset   (B,T) uint8 or NULL: non-zero = the velocity state prev_xy is None at that frame
 *         (t == 0 always resets)
 * a_*     positions of landmarks 61, 291, 13, 14 inside the K-list
 * variant 0 = recorder arithmetic (mouth width in float64), 1 = live (float32 points)
 * X       (B,T, ldx>=2K+4): [x0,y0,...,vel,open_px,width_px,aspect]
 * center  (B,T,2) float32 pixel centre; fourth (B,T) float64: width+1e-6 (recorder) or width (live) */
int ss_feature_fuse(const float* lm, const uint8_t* reset, int B, int T, int K, int w, int h, int a_left,
                    int a_right, int a_up, int a_lo, int variant, float* X, int ldx, float* center,
                    double* fourth, ss_stream_t stream);

/* The live chain's front for n DISTINCT streams, one frame each (live_infer_official.py:264-296): the distance gate
 * band_lo <= mouth_w <= band_hi (MOUTH_W_MIN_PX / MAX_PX, :23-24, :272) decides whether a frame is KEPT; a kept frame gets
 * extract_feature with the stream's velocity state, a dropped frame of a recording stream clears that state (:295-296), so the
 * first kept frame after re-entry has vel = 0.  State, resident in HBM, zero-initialised by the caller: prev_lm (S,K,2) = raw
 * landmarks of the stream's previous kept frame, has_prev (S) u8.  recording (n) u8 or NULL (= all recording): 0 leaves the
 * stream untouched (``if recording`` false).  Outputs per frame i: X row (zeros when dropped), center, fourth (as
 * ss_feature_fuse), kept[i].  stream_ids must be distinct and in [0, n_streams) -- the caller checks. */
int ss_feature_fuse_stream(const float* lm, const int32_t* stream_ids, const uint8_t* recording, int n, int n_streams, int K,
                           int w, int h, int a_left, int a_right, int a_up, int a_lo, int variant, double band_lo,
                           double band_hi, float* prev_lm, uint8_t* has_prev, float* X, int ldx, float* center,
                           double* fourth, uint8_t* kept, ss_stream_t stream);

/* ---- a4 + a5 for ANY ROI size (csrc/roi_cnn_generic.hip): the building blocks of the layer-by-layer form the host sequences for
 * frame sizes outside the fused kernels' set (ss_roi_cnn_stash_size returns SS_ERR_UNSUPPORTED for them): every 3x3 convolution is
 * ss_im2col3x3 + ss_gemm_f32_batched against nn.Conv2d's weight as it stands ((Cout, C, 3, 3) = [n][k], k = c*9 + ky*3 + kx).
 *   ss_roi_norm    R (N, HW) u8 -> xn (N, HW) f32 = (u/255 - mu)/sd (train_model_official.py:286-291; standardize 0: u/255);
 *                  stats (N, 2) = mu, sd or NULL
 *   ss_im2col3x3   src (N, C, H, W) planar -> col (N H W, ld_col >= 9 C), zero padding at the borders and in columns [9 C, ld_col)
 *   ss_relu_pool2  y (N H W, C) pixel-major pre-activation -> a (N, C, H/2, W/2) = maxpool(relu(y)), idx = winner of the 2x2 window
 *                  (0..3 row-major, the first on ties)
 *   ss_relu_mean   y (N P, C) -> feat (N, C) = mean over the P pixels of relu(y); mask (N P, C) = y > 0 (NULL: not written)
 *   ss_mask_scale  dy (N P, C) = mask ? dfeat[n][c] / P : 0
 *   ss_col2im3x3   dcol (N H W, ld_col) -> d (N, C, H, W): the transposed convolution's gather
 *   ss_pool2_bwd   da, a, idx (N, C, H2, W2) -> dy (N 2H2 2W2, C) pixel-major: the winner gets da where a > 0, zeros elsewhere */
int ss_roi_norm(const uint8_t* R, int N, int HW, int standardize, float* xn, float* stats, ss_stream_t stream);
int ss_im2col3x3(const float* src, int N, int C, int H, int W, float* col, int ld_col, ss_stream_t stream);
int ss_relu_pool2(const float* y, int N, int C, int H, int W, float* a, uint8_t* idx, ss_stream_t stream);
int ss_relu_mean(const float* y, int N, int P, int C, float* feat, uint8_t* mask, ss_stream_t stream);
int ss_mask_scale(const uint8_t* mask, const float* dfeat, int N, int P, int C, float* dy, ss_stream_t stream);
int ss_col2im3x3(const float* dcol, int ld_col, int N, int C, int H, int W, float* d, ss_stream_t stream);
int ss_pool2_bwd(const float* da, const float* a, const uint8_t* idx, int N, int C, int H2, int W2, float* dy, ss_stream_t stream);

/* ---- a2: ROI crop rectangle ---------------------------------------------------------------
 * replaces the index arithmetic of crop_roi (record_landmarks_official.py:106-114, variant 0) and
 * crop_roi_gray (live_infer_official.py:172-181, variant 1).  Bit-exact integers.
 * box (n,5) int32: x1, x2, y1, y2, valid */
int ss_roi_crop_idx(const float* center, const double* scale, int n, int w, int h, int variant, int32_t* box,
                    ss_stream_t stream);

/* ---- a2: crop -> BGR2GRAY -> resize behind the crop box --------------------------
 * replaces cv2.cvtColor + cv2.resize in crop_roi (record_landmarks_official.py:116-118, INTER_LINEAR: interp 0) and
 * crop_roi_gray (live_infer_official.py:184-186, INTER_AREA: interp 1).  frames_bgr (N,h,w,3) u8, box (N,5) from
 * ss_roi_crop_idx, out (N,roi_h,roi_w) u8 (zeros where the box is invalid).  OpenCV's 8-bit algorithms restated;
 * parity with OpenCV itself is unpinned: +-1 grey level is the contract. */
int ss_crop_gray_resize(const uint8_t* frames_bgr, int N, int h, int w, const int32_t* box, int roi_h, int roi_w,
                        int interp, uint8_t* out, ss_stream_t stream);

/* ---- a4+a5(+a6): ROI normalise + TinyROICNN, fused ------------------------------------------
 * replaces train_model_official.py:286-291 and TinyROICNN.forward (:212-229); with
 * standardize = 0 the live variant (live_infer_official.py:126-127).
 * R    (N,H,W) uint8 raw 0..255;  H, W multiples of 4 (two 2x2 pools)
 * w1 (8,1,3,3) b1 (8) w2 (16,8,3,3) b2 (16) w3 (24,16,3,3) b3 (24) wfc (E,24) bfc (E), E <= 64
 * out  row n written at out + n*ld_out, E floats (pass Z + x_dim with ld_out = x_dim+E to fuse
 *      the torch.cat of train_model_official.py:297) */
int ss_roi_cnn_fwd(const uint8_t* R, int N, int H, int W, int standardize, const float* w1, const float* b1,
                   const float* w2, const float* b2, const float* w3, const float* b3, const float* wfc,
                   const float* bfc, int E, float* out, int ld_out, ss_stream_t stream);

/* Workgroups the two persistent ROI-CNN kernels launch (default 256 = one per CU; 0 restores the default).
 * A caller that runs a second stream beside them (micro-batch pipelining) leaves CUs free this way. */
int ss_roi_cnn_set_max_workgroups(int n);

/* per-frame sizes of the six stash arrays for an (H, W) the kernels are built for, sizes[6] (host memory) =
 * { floats of the pooled-1 map, floats of the pooled-2 map, bytes of the pool-1 argmax map (padded planes), bytes of the
 *   pool-2 argmax map, bytes of the conv3 sign mask, floats of an st_feat row } */
#define SS_ROI_CNN_STASH_SIZES 6
int ss_roi_cnn_stash_size(int H, int W, int* sizes);

/* Same forward, additionally writing what the backward needs (the six st_* pointers are either
 * all NULL -- then this is ss_roi_cnn_fwd -- or all valid):
 * st_a1 (N, a1_floats) f32: the pooled conv1 map in the kernels' zero-haloed LDS layout (8 planes of
 *   (H/2+2) rows x (W/2+2) floats, plane stride padded; sizes from ss_roi_cnn_stash_size) so that the backward
 *   streams it back with linear 16-byte copies; st_i1 (N, i1_bytes) u8: its 2x2 argmax (0..3, row-major window),
 *   8 planes of Geom::I1S = H/2*W/2 bytes;
 * st_a2 (N, a2_floats) f32 and st_i2 (N,H/4,W/4,16) u8 (pixel-major): the same for conv2;
 * st_m3 (N,H/4*W/4,32) u8, pixel-major, channels 24..31 zero: conv3 output > 0; st_feat (N,52): globally averaged conv3 features, then the
 * per-channel counts of positive conv3 outputs, then the mean and the standard deviation of the frame (the backward kernel reuses them).
 * stash_sizes (host memory, 6 ints; may be NULL when the st_* are): the per-frame sizes the caller allocated the six arrays
 *   with (what ss_roi_cnn_stash_size returned).  The forward and the backward kernel each compare ALL of them with the
 *   layout THEY were compiled with and return SS_ERR_ARG on a mismatch instead of writing / reading out of bounds (round 1
 *   lost a GPU process to two objects built against different versions of that layout: DESIGN.md section 9). */
int ss_roi_cnn_fwd_stash(const uint8_t* R, int N, int H, int W, int standardize, const float* w1, const float* b1,
                         const float* w2, const float* b2, const float* w3, const float* b3, const float* wfc,
                         const float* bfc, int E, float* out, int ld_out, float* st_a1, uint8_t* st_i1,
                         float* st_a2, uint8_t* st_i2, uint8_t* st_m3, float* st_feat, const int* stash_sizes,
                         ss_stream_t stream);

/* autograd of the above w.r.t. the eight parameter tensors (the uint8 input has no gradient),
 * i.e. what loss.backward() (train_model_official.py:437) leaves in roi_cnn.*.grad.
 * d_out row n read at d_out + n*ld_dout (E floats).  Gradients are ACCUMULATED (+=) into g_*.
 * Needs the stash of ss_roi_cnn_fwd_stash for the same R and weights. */
int ss_roi_cnn_bwd(const uint8_t* R, int N, int H, int W, int standardize, const float* w1, const float* b1,
                   const float* w2, const float* b2, const float* w3, const float* b3, const float* wfc,
                   const float* bfc, int E, const float* st_a1, const uint8_t* st_i1, const float* st_a2,
                   const uint8_t* st_i2, const uint8_t* st_m3, const float* st_feat, const int* stash_sizes,
                   const float* d_out, int ld_dout, float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* g_w3, float* g_b3,
                   float* g_wfc, float* g_bfc, ss_stream_t stream);

/* ---- padded batches: only the frames that belong to a clip ---------------------------------
 * The reference pads every clip to T frames and runs TinyROICNN over all B*T of them (train_model_official.py:286-297) although
 * pack_padded_sequence (:300) drops rows t >= lengths[b] before the recurrence: their embeddings never reach the logits and
 * their gradient is exactly zero.  ss_roi_active_frames lists the rows that count -- frames[0] = how many, frames[1 ...] = the
 * row numbers b*T + t with t < min(max(lengths[b], 0), T), ascending; frames needs 1 + B*T ints of device memory -- and, when
 * emb is given, clears emb[row*ld_emb + 0 .. emb_cols) of every other row (pass Z + x_dim: the input-projection GEMM still
 * reads those rows).  ss_roi_cnn_fwd_frames / ss_roi_cnn_bwd_frames are ss_roi_cnn_fwd_stash / ss_roi_cnn_bwd walking only the
 * listed frames (frames == NULL: all N); `out` rows and stash slots stay indexed by the frame number, so the two launches of a
 * step must be given the same list.  Logits and gradients are those of the full walk. */
int ss_roi_active_frames(const int32_t* lengths, int B, int T, int* frames, float* emb, int ld_emb, int emb_cols,
                         ss_stream_t stream);
int ss_roi_cnn_fwd_frames(const uint8_t* R, int N, int H, int W, int standardize, const float* w1, const float* b1,
                          const float* w2, const float* b2, const float* w3, const float* b3, const float* wfc,
                          const float* bfc, int E, float* out, int ld_out, float* st_a1, uint8_t* st_i1,
                          float* st_a2, uint8_t* st_i2, uint8_t* st_m3, float* st_feat, const int* stash_sizes,
                          const int* frames, ss_stream_t stream);
int ss_roi_cnn_bwd_frames(const uint8_t* R, int N, int H, int W, int standardize, const float* w1, const float* b1,
                          const float* w2, const float* b2, const float* w3, const float* b3, const float* wfc,
                          const float* bfc, int E, const float* st_a1, const uint8_t* st_i1, const float* st_a2,
                          const uint8_t* st_i2, const uint8_t* st_m3, const float* st_feat, const int* stash_sizes,
                          const float* d_out, int ld_dout, float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* g_w3,
                          float* g_b3, float* g_wfc, float* g_bfc, const int* frames, ss_stream_t stream);

/* ---- dense contraction used by a7/a9 and their gradients -----------------------------------
 * C[M,N] (+)= opA[M,K] * opB[K,N] (+ bias[N]) (then ReLU), exact-f32 MFMA.
 * a_kcontig = 1: A is stored [M][K] (lda = row stride); 0: stored [K][M].
 * b_kcontig = 1: B is stored [N][K] (torch Linear weight layout); 0: stored [K][N].
 * *_group/_gstride/_off remap the storage ROW index r of that operand to
 *   (r / group) * gstride + (r % group) + off   (identity: group = INT32_MAX, off = 0);
 *   used to pair dG[b][t] with h[b][t-1] without materialising a shifted copy.
 * a_colsum (may be NULL; needs a_kcontig = 0): a_colsum[m] += sum_k A[k][m] -- the bias gradient of a
 *   Linear layer rides on its weight-gradient GEMM instead of a separate reduction pass.
 * flags bit0: accumulate into C; bit1: ReLU epilogue; bit2: accumulate with float atomics even when K is not
 * split (several streams add into one C).  splits > 1 slices K over blockIdx.z and adds with float atomics
 * (atomics require bit0, C pre-initialised, no ReLU).
 * flags bit4 (ss_gemm_f32_batched only): the `batch` problems are SUMMED into one C (stride_c ignored) -- the K loop of a
 * workgroup runs through all (A, B) pairs, so e.g. d layer_in = dG_fwd W_fwd + dG_rev W_rev needs neither atomics nor a cleared C.
 * flags bit3 (alone; no bias): C is a split-K workspace of ss_gemm_splitk_ws_floats() floats (16-byte aligned) instead of
 * the result: every workgroup leaves its raw accumulators there with plain stores and ss_gemm_splitk_reduce adds the
 * slices into the real C afterwards -- the weight-gradient GEMMs (M x N tiny, K = B*T) spend as long in float atomics
 * as in their K loop otherwise. */
int ss_gemm_f32(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, int a_group,
                int a_gstride, int a_off, const float* B, int ldb, int b_group, int b_gstride, int b_off,
                float* C, int ldc, const float* bias, float* a_colsum, int flags, int splits, ss_stream_t stream);

/* `batch` independent problems of identical shape in one launch (the two GRU directions): problem b uses
 * A + b*stride_a, B + b*stride_b, C + b*stride_c, bias + b*stride_bias, a_colsum + b*stride_colsum (element strides). */
int ss_gemm_f32_batched(int a_kcontig, int b_kcontig, int M, int N, int K, const float* A, int lda, int a_group,
                        int a_gstride, int a_off, const float* B, int ldb, int b_group, int b_gstride, int b_off,
                        float* C, int ldc, const float* bias, float* a_colsum, int flags, int splits, int batch,
                        long stride_a, long stride_b, long stride_c, long stride_bias, long stride_colsum,
                        ss_stream_t stream);

/* Second half of a split-K GEMM run with flags bit3: C[b][m][n] += sum over the K slices left in ws.  M, N, K, splits and
 * batch are those of the GEMM call; stride_c = element stride between the problems' C. */
int ss_gemm_splitk_ws_floats(int M, int N, int K, int splits, int batch, long* floats);
int ss_gemm_splitk_reduce(const float* ws, int M, int N, int K, int splits, int batch, float* C, int ldc, long stride_c,
                          ss_stream_t stream);

/* What a training step does before its first real kernel, in one launch: grads[0:n_grads] = 0 (the flat gradient bucket of
 * optimizer.zero_grad, train_model_official.py:433), scalars[0:n_scalars] = 0 and correct[0] = 0 (loss / sum of squares /
 * hit counter; correct may be NULL), lengths32[b] = lengths64[b] (NULL: skipped), Z[r][0:cols] = X[r][0:cols] for `rows`
 * rows (the landmark half of torch.cat((X, roi_emb)), train_model_official.py:297; X NULL: skipped); frames (may be NULL; needs
 * lengths64, X and rows = B*T): what ss_roi_active_frames does, from the int64 lengths, clearing Z[row][cols .. cols + emb_cols). */
int ss_train_prologue(float* grads, long n_grads, float* scalars, int n_scalars, int32_t* correct,
                      const int64_t* lengths64, int32_t* lengths32, int B, const float* X, int ld_x, float* Z, int ld_z,
                      int rows, int cols, int* frames, int emb_cols, ss_stream_t stream);

/* Several split-K problems C_j (ldc_j, stride_c_j between the batch members) += op(A_j) op(B_j) in ONE GEMM launch plus ONE
 * reduce launch (the three weight-gradient GEMMs of a GRU layer; every launch boundary on a stream costs the tail of one
 * kernel and the ramp of the next).  Fields as the arguments of ss_gemm_f32_batched; n <= 4; all problems of one operand
 * layout.  ws: ss_gemm_splitk_group_ws_floats() floats, 16-byte aligned, contents irrelevant. */
typedef struct ss_gemm_problem {
  int a_kcontig, b_kcontig, M, N, K;
  const float* A;
  int lda, a_group, a_gstride, a_off;
  const float* B;
  int ldb, b_group, b_gstride, b_off;
  float* C;
  int ldc, splits, batch;
  long stride_a, stride_b, stride_c;
} ss_gemm_problem;
int ss_gemm_splitk_group_ws_floats(const ss_gemm_problem* problems, int n, long* floats);
/* ws_floats = the floats the caller allocated at ws: the entry points recompute the layout and return SS_ERR_ARG when it does not fit
 * (a caller that sized ws for another group would otherwise be written past its end, silently). */
int ss_gemm_f32_splitk_group(const ss_gemm_problem* problems, int n, float* ws, long ws_floats, int flags, ss_stream_t stream);
/* flags bit 0: the launch has the chip to itself -> k-major groups run as 192 x 192 output tiles, one 512-thread workgroup per CU,
 * K slices chosen by the library so that the group fills the chip in one round (`splits` ignored): dG and the layer input are
 * fetched 2 - 3 x instead of 5 - 6 x and the launch takes 12 - 18 % less time alone on the chip.  Beside a latency-bound kernel on
 * another stream (the lower layer's BPTT) the 128 x 64 form with three workgroups per CU is the better neighbour: leave the bit 0. */
/* The same for bf16 operands (config 5; csrc/gemm_bf16.hip): A and B point at bf16 data (k-major, a_kcontig = b_kcontig = 0; lda /
 * ldb / strides in bf16 elements, multiples of 8; K a multiple of 64; n <= 8), C[b] += A[b]^T B[b] in f32.  A group with at least
 * half a chip of 256 x 128 output tiles runs one workgroup per tile over all of K (neighbouring tiles share operand panels in L2);
 * a smaller one has its K tiles dealt evenly over the CUs ("stream-K"), every workgroup leaves raw accumulators in `ws`
 * (ss_gemm_bf16_splitk_group_ws_floats() floats, 16-byte aligned, contents irrelevant) and a second launch folds them into C.
 * `splits` is ignored. */
int ss_gemm_bf16_splitk_group_ws_floats(const ss_gemm_problem* problems, int n, long* floats);
int ss_gemm_bf16_splitk_group(const ss_gemm_problem* problems, int n, float* ws, long ws_floats, ss_stream_t stream);

/* a[0..na) = 0 and b[0..nb) = 0 in one launch (either may be empty; 16-byte aligned): the destinations the d layer_in GEMMs
 * sum into with float atomics are cleared by this, off the critical path, instead of two library fills */
int ss_zero_f32x2(float* a, long na, float* b, long nb, ss_stream_t stream);

/* column sums: out[n] += sum_r A[r*lda + n]  (bias gradients) */
int ss_colsum_f32(const float* A, int rows, int cols, int lda, float* out, ss_stream_t stream);

/* ---- a3: batch assembly on the device ----------------------------------------------
 * replaces NPZWordDataset.__getitem__ + collate_fn (train_model_official.py:122-204) for clips that are
 * resident in HBM as one ragged frame store: additive feature noise (:143-145), interior frame drop (:146-152,
 * expressed through the map), zero padding / trimming to max_t (:93-118), stacking (:174-204).
 * frame_map (rows = B*max_t) int32: row of the store that lands in each destination row, -1 = zero padding.
 * f32: dst (rows, D) = gathered rows + noise on the rows whose noise_map entry is >= 0 (NULL map: none):
 *      noise != NULL -> noise[noise_map[r]] (host-drawn, bit-exact with the reference's np.random.normal);
 *      noise == NULL and noise_std > 0 -> noise_std * N(0,1) from the Philox stream (seed, element index).
 * u8 : dst (rows, frame_bytes), frame_bytes a multiple of 16 (ROI frames). */
int ss_batch_gather_f32(const float* src, int D, const int32_t* frame_map, long rows, const float* noise,
                        const int32_t* noise_map, float noise_std, uint64_t seed, float* dst, ss_stream_t stream);
int ss_batch_gather_u8(const uint8_t* src, int frame_bytes, const int32_t* frame_map, long rows, uint8_t* dst,
                       ss_stream_t stream);

/* ---- sliding-window serving of many streams ---------------------------------------
 * per stream a ring of the last max_t frames (features (S,max_t,D) f32, optional ROI (S,max_t,frame_bytes) u8), a head
 * and a count: the deque(maxlen=max_t) of inactive/live_feed.py:155.  A push appends one frame for each of n DISTINCT
 * streams (feats (n,D), rois (n,frame_bytes)) and bumps frames_seen; the window map lists, oldest first, the rows of the
 * rings of n selected streams (-1 = zero padding; live_feed.py:203-207) for ss_batch_gather_f32/u8, lengths = frames held.
 * ss_mouth_openness: the openness signal itself, float64 like the reference's Python floats.  lm (n,K,2) f32 normalised
 *   (x,y) landmarks; mode 0: |y[i_bot] - y[i_top]| / (dist2d(lm[i_eye_l], lm[i_eye_r]) + 1e-6)
 *   (important_landmarks.py:64-67, 131-133; within one ulp of its ``** 0.5``); mode 1: max(y) - min(y) over the K
 *   landmarks (inactive/live_test_5.py:92-94; the i_* arguments are ignored); mode 2: |lm[i_top] - lm[i_bot]| / (|lm[i_eye_r] -
 *   lm[i_eye_l]| + 1e-6) in float32, every operation rounded (inactive/live_feed.py:69-78 with i_top/i_bot = 13/14 and the
 *   mouth corners 61/291 in the two "eye" slots).  All three are pinned by the reference's own outputs (tests/golden/serving.npz).
 * ss_mouth_gate: EMA (alpha) of the openness and its open/close hysteresis, carried in float64
 *   (important_landmarks.py:136-144: ``mouth_ema`` is a Python float).
 * ss_clip_gate: openness-gated clip segmentation of n DISTINCT streams (inactive/live_test_5.py:146-152, 233-272):
 *   state (S,4) int32 = {speaking, above_ct, below_ct, clip_len}, zero-initialised.  Per pushed frame:
 *   openv > open_thresh bumps above_ct and clears below_ct, else the reverse; idle streams start a clip (clip_len = 0)
 *   after start_n consecutive frames above -- the starting frame itself is not kept; speaking streams append the frame
 *   (feats row -> clip_x (S,max_clip,D), rois -> clip_r (S,max_clip,frame_bytes)) and stop after end_n consecutive
 *   frames below or at max_clip frames.  append_row (n): row written, -1 if none; emit_len (n): length of the clip that
 *   just ended if >= min_clip (the reference classifies clips of >= 6 frames), else 0.  face_present (n) u8 may be
 *   NULL; 0 resets the stream (the "NO FACE" branch, :293-301). */
int ss_ring_push(float* ring_x, uint8_t* ring_r, int n_streams, int max_t, int D, int frame_bytes,
                 const int32_t* stream_ids, int n, const float* feats, const uint8_t* rois, int32_t* head, int32_t* count,
                 int32_t* frames_seen, ss_stream_t stream);
/* a camera frame in which no face was found, for n DISTINCT streams: frames_seen runs on (live_feed.py:173 counts every camera
 * frame, and the prediction rule :201 tests that count), the ring keeps what it holds (:179-185 ``continue``) */
int ss_ring_tick(const int32_t* stream_ids, int n, int32_t* frames_seen, ss_stream_t stream);
int ss_ring_window_map(const int32_t* stream_ids, int n, int max_t, const int32_t* head, const int32_t* count,
                       int32_t* frame_map, int64_t* lengths, ss_stream_t stream);
int ss_mouth_openness(const float* lm, int n, int K, int mode, int i_top, int i_bot, int i_eye_l, int i_eye_r,
                      double* openness, ss_stream_t stream);
int ss_mouth_gate(const int32_t* stream_ids, int n, const double* openness, double alpha, double open_thr,
                  double close_thr, double* ema, uint8_t* state_open, ss_stream_t stream);
int ss_clip_gate(const int32_t* stream_ids, int n, const double* openv, const uint8_t* face_present, double open_thresh,
                 int start_n, int end_n, int max_clip, int min_clip, int32_t* state, int D, int frame_bytes,
                 const float* feats, const uint8_t* rois, float* clip_x, uint8_t* clip_r, int32_t* append_row,
                 int32_t* emit_len, ss_stream_t stream);

/* ---- BASELINE config 5 (100 words, 96x96 ROI, CNN 16/32/64/96, BiGRU H = 512, bf16 MFMA) ----------------------
 * The reference defines no such model ("build-defined"); it is the same module
 * (train_model_official.py:209-229, 253-310) with wider layers, run on v_mfma_f32_16x16x32_bf16: operands rounded to
 * bf16 on their way into LDS / registers, f32 accumulation, f32 master weights, activations between the GRU kernels f32.
 *
 * ss_gemm_bf16_batched: C[M,N] (+)= opA * opB (+ bias), the conventions of ss_gemm_f32_batched (f32 operands in HBM,
 *   a_kcontig / b_kcontig, storage-row remap, batch strides in elements); flags bit0 accumulate, bit2 float atomics;
 *   splits > 1 slices K and needs bit0 (atomics into a C the caller initialised).  Contiguous dimensions % 4 == 0.
 *   flags bit3: A and B already ARE bf16 in HBM (uint16 bit patterns; lda / ldb / strides in bf16 elements, multiples of 8,
 *   leading dimensions zero-padded to a multiple of 8 where the extent is not one: ss_cvt_bf16_rows): no conversion pass,
 *   half the operand bytes, 64-deep k tiles. */
int ss_gemm_bf16_batched(int a_kcontig, int b_kcontig, int M, int N, int K, const void* A, int lda, int a_group,
                         int a_gstride, int a_off, const void* B, int ldb, int b_group, int b_gstride, int b_off,
                         float* C, int ldc, const float* bias, int flags, int splits, int batch, long stride_a,
                         long stride_b, long stride_c, long stride_bias, ss_stream_t stream);

/* GRU recurrence for wide hidden states (H % 128 == 0), both directions, same masking semantics and buffer layouts as
 * ss_gru_fwd / ss_gru_bwd (gi (2,N,3H), out (N,2H), save / d_g (2,N,4,H)).
 * ss_gru_bf16_prep: W_hh of both directions -> whh_bf16 (2,3H,H) and its transpose whh_t_bf16 (2,H,3H), once per
 *   optimiser step; with wih_bf16 (may be NULL) also W_ih (3H,K) of both directions -> (2,3H,Kp) bf16, Kp = K rounded up
 *   to 8, zero-padded: the operand of the layer's input-projection / d layer_in GEMMs (ss_gemm_bf16_batched, bit 3).  ws: ss_gru_bf16_ws_bytes(B, H) bytes (state / gate-gradient hand-over between step launches).
 * Two forms behind one entry point.  With sync_ws (ss_gru_bf16_sync_bytes(B, T, H) bytes, > 0 for H = 128 ... 512;
 *   ZEROED ONCE by the caller, kept consistent by the kernels afterwards) a layer is ONE persistent launch per clip chunk:
 *   a (16-clip slice, direction) is spread over H/64 workgroups that keep their rows of W_hh in registers for all T steps
 *   and exchange the state (forward) / partial d h_prev (backward) through tagged 8-byte granules (csrc/gru_bf16_pers.h).
 *   sync_ws = NULL, or a shape outside that range: one launch per time step.  sync_bytes = the bytes the caller allocated at
 *   sync_ws: the layout is a function of (B, T, H) -- and not monotone in B -- so the entry points recompute it and return
 *   SS_ERR_ARG when an area sized for another batch is too small for this one.
 *   sync_ws words: [0] launch generation, [1] arrivals, [2] bounded waits that gave up (a lost partner: the workgroups
 *   concerned emit NaN from then on; the owner reads this word where it synchronises), [3] same-XCD fast-path count,
 *   [5] fault injection for tests (1 + index of a workgroup that plays dead; 0 = off).
 * By-products (each may be NULL): out_bf16 (N,2H) = bf16 copy of out; out_drop_bf16 (N,2H) = bf16 of dropout(out) with
 *   ss_dropout's Philox stream at (seed, offset) over the (N,2H) index space -- nn.GRU's inter-layer dropout, ready as
 *   the next layer's MFMA operand (drop_p = 0: a plain copy); d_g_bf16 (2,N,4,H) = bf16 copy of d_g (with it, the
 *   persistent form also accepts d_g = NULL: 126 MB per launch less to write at config 5); g_b* (all four or
 *   none): d b_ih += column sums of (d r, d z, d n), d b_hh += column sums of (d r, d z, d hn) (what ss_gru_bias_grad
 *   computes from d_g).  ss_gru_bf16_bwd: drop_p / seed / offset = the same mask re-drawn on d_out (see ss_gru_bwd).
 * ss_cvt_bf16_rows: y[r][c] = bf16(x[r][c] * mask), c < cols; y[r][cols .. ld_y) = 0 (GEMM operands are read in whole
 *   8-element chunks); cols, ld_x, ld_y % 4 == 0; the mask indexes the SOURCE element r * ld_x + c. */
int ss_gru_bf16_prep(const float* w_hh_f, const float* w_hh_r, int H, uint16_t* whh_bf16, uint16_t* whh_t_bf16,
                     const float* w_ih_f, const float* w_ih_r, int K, uint16_t* wih_bf16, ss_stream_t stream);
int ss_gru_bf16_ws_bytes(int B, int H, long* bytes);
int ss_gru_bf16_sync_bytes(int B, int T, int H, long* bytes);
int ss_gru_bf16_fwd(const float* gi, const uint16_t* whh_bf16, const float* b_hh_f, const float* b_hh_r,
                    const int32_t* lengths, int B, int T, int H, float* out, float* save, uint16_t* out_bf16,
                    uint16_t* out_drop_bf16, float drop_p, uint64_t seed, uint64_t offset, void* ws, void* sync_ws,
                    long sync_bytes, ss_stream_t stream);
int ss_gru_bf16_bwd(const float* d_out, const float* out, const float* save, const uint16_t* whh_t_bf16,
                    const int32_t* lengths, int B, int T, int H, float* d_g, uint16_t* d_g_bf16, float drop_p,
                    uint64_t seed, uint64_t offset, float* g_bih_f, float* g_bhh_f, float* g_bih_r, float* g_bhh_r,
                    void* ws, void* sync_ws, long sync_bytes, ss_stream_t stream);
int ss_cvt_bf16_rows(const float* x, int ld_x, uint16_t* y, int ld_y, long rows, int cols, float drop_p, uint64_t seed,
                     uint64_t offset, ss_stream_t stream);

/* bf16 ROI-CNN of config 5: 96x96 uint8 frame -> normalise -> [conv3x3 + ReLU + maxpool2] x 3 (1->16->32->64) ->
 * conv3x3 64->96 + ReLU -> global average -> Linear(96 -> E): TinyROICNN (train_model_official.py:209-229) with a fourth
 * block.  One persistent kernel per layer; between the layers the pooled maps are NHWC bf16 (N,H,W,C) in HBM with one
 * argmax byte per pooled element (0..3 = position in the 2x2 window, 4 = maximum not positive: no gradient).
 *   ss_c5_conv1_fwd      R (N,96,96) u8 -> a1 (N,48,48,16) bf16, i1 (N,48,48,16) u8, st (N,2) f32 mean/std (may be NULL)
 *   ss_c5_conv_fwd       layer 2: a1 -> a2 (N,24,24,32), i2;  layer 3: a2 -> a3 (N,12,12,64), i3
 *   ss_c5_conv_last_fwd  a3 -> z rows (E floats at z + n*ld_z); training also mask (N,144,96) u8 and feat (N,96) f32
 * Backward (gradients ACCUMULATE into g_*; d a maps are UNMASKED, the argmax byte of the layer below applies its ReLU):
 *   ss_c5_conv_last_wgrad  d z (N,E at ld_dz) -> g_w4, g_b4, g_wfc, g_bfc;   ss_c5_conv_last_dgrad -> da3 (N,12,12,64)
 *   ss_c5_conv_wgrad / ss_c5_conv_dgrad  layer 3: (a2, da3, i3) -> g_w3, g_b3 / da2;  layer 2: (a1, da2, i2) -> g_w2, g_b2 / da1
 *   ss_c5_conv1_wgrad      (R, st, da1, i1 | w1, b1) -> g_w1, g_b1 */
int ss_c5_conv1_fwd(const uint8_t* R, int N, int standardize, const float* w1, const float* b1, uint16_t* a1, uint8_t* i1,
                    float* st, ss_stream_t stream);
int ss_c5_conv_fwd(int layer, const uint16_t* in, int N, const float* w, const float* b, uint16_t* out, uint8_t* idx,
                   ss_stream_t stream);
int ss_c5_conv_last_fwd(const uint16_t* in, int N, const float* w, const float* b, const float* wfc, const float* bfc, int E,
                        float* z, int ld_z, uint8_t* mask, float* feat, ss_stream_t stream);
int ss_c5_conv_wgrad(int layer, const uint16_t* a_in, const uint16_t* da_out, const uint8_t* idx, int N, float* g_w, float* g_b,
                     ss_stream_t stream);
int ss_c5_conv_dgrad(int layer, const uint16_t* da_out, const uint8_t* idx, int N, const float* w, uint16_t* da_in,
                     ss_stream_t stream);
int ss_c5_conv_last_wgrad(const uint16_t* a_in, const float* dz, int ld_dz, int E, const float* wfc, const uint8_t* mask,
                          const float* feat, int N, float* g_w, float* g_b, float* g_wfc, float* g_bfc, ss_stream_t stream);
int ss_c5_conv_last_dgrad(const float* dz, int ld_dz, int E, const float* wfc, const uint8_t* mask, int N, const float* w,
                          uint16_t* da_in, ss_stream_t stream);
int ss_c5_conv1_wgrad(const uint8_t* R, int N, int standardize, const float* st, const uint16_t* da1, const uint8_t* i1,
                      const float* w1, const float* b1, float* g_w1, float* g_b1, ss_stream_t stream);
/* ss_c5_conv_dgrad(2) and ss_c5_conv1_wgrad (recompute form) in ONE kernel: the gradient w.r.t. the pooled conv1 map (73.7 KB per
 * frame, the largest gradient of the net) is born band by band in LDS and consumed there: 1.13 GB per step at config 5 that no
 * longer crosses HBM.  da1 may be NULL (the product path); non-NULL: the map is also stored (N,48,48,16), for tests. */
int ss_c5_conv2_dgrad_conv1_wgrad(const uint16_t* da2, const uint8_t* i2, int N, const float* w2, const uint8_t* R, const float* st,
                                  int standardize, const float* w1, const float* b1, uint16_t* da1, float* g_w1, float* g_b1,
                                  ss_stream_t stream);
/* The fused forms the engine uses (the pooled conv1 map -- 74 KB per frame + 37 KB of argmax bytes, the largest tensor of the
 * net -- never reaches HBM):
 *   ss_c5_conv12_fwd      R -> a2 (N,24,24,32), i2, st: conv1 lands in conv2's LDS image
 *   ss_c5_conv2_wgrad_rc  layer 2's weight gradient with a1 recomputed per band from R and st (conv1: w1, b1)
 *   ss_c5_conv1_wgrad     with i1 == NULL recomputes conv1's pool winners from R (needs w1, b1) */
int ss_c5_conv12_fwd(const uint8_t* R, int N, int standardize, const float* w1, const float* b1, const float* w2, const float* b2,
                     uint16_t* a2, uint8_t* i2, float* st, ss_stream_t stream);
/* Layer 4 with its Linear and the Linear's gradients as the caller's GEMMs over all frames (three small f32 GEMMs instead of a
 * per-frame walk over W_fc inside the persistent kernels):
 *   ss_c5_conv_last_fwd_feat  a3 -> feat (N,96) f32 (+ mask (N,144,96) u8 when training); z = feat . W_fc^T + b_fc is the caller's
 *   ss_c5_conv_last_wgrad_df / _dgrad_df  take dfeat (N,96) f32 = d z . W_fc (unscaled: the kernels divide by 144);
 *                             g_wfc += d z^T . feat and g_bfc += column sums of d z are the caller's */
int ss_c5_conv_last_fwd_feat(const uint16_t* in, int N, const float* w, const float* b, uint8_t* mask, float* feat, ss_stream_t stream);
int ss_c5_conv_last_wgrad_df(const uint16_t* a_in, const float* dfeat, const uint8_t* mask, int N, float* g_w, float* g_b,
                             float* part, long part_floats, ss_stream_t stream);
/* Weight gradients with the workgroups' partial sums through a scratch buffer instead of float atomics onto g_w:
 * part = (min(N, CUs)) x (COUT * CIN * 9) floats, 16-byte aligned, contents irrelevant; NULL or too small: the atomics form.
 * (256 workgroups adding 55 k floats each onto the same 55 k addresses was half of the last layer's launch.) */
int ss_c5_conv_wgrad_ws(int layer, const uint16_t* a_in, const uint16_t* da_out, const uint8_t* idx, int N, float* g_w, float* g_b,
                        float* part, long part_floats, ss_stream_t stream);
int ss_c5_conv2_wgrad_rc_ws(const uint8_t* R, const float* st, int standardize, const float* w1, const float* b1,
                            const uint16_t* da_out, const uint8_t* idx, int N, float* g_w, float* g_b, float* part, long part_floats,
                            ss_stream_t stream);
int ss_c5_conv_last_dgrad_df(const float* dfeat, const uint8_t* mask, int N, const float* w, uint16_t* da_in, ss_stream_t stream);
/* The same two with conv1's pool winners i1 (N,48,48,16) u8 left in HBM by the forward kernel and read back by the fused backward
 * kernel instead of being recomputed from the frame (37 KB per frame each way; i1 = NULL: the forms above). */
int ss_c5_conv12_fwd_i1(const uint8_t* R, int N, int standardize, const float* w1, const float* b1, const float* w2, const float* b2,
                        uint16_t* a2, uint8_t* i2, float* st, uint8_t* i1, ss_stream_t stream);
int ss_c5_conv2_dgrad_conv1_wgrad_i1(const uint16_t* da2, const uint8_t* i2, int N, const float* w2, const uint8_t* R, const float* st,
                                     int standardize, const float* w1, const float* b1, uint16_t* da1, float* g_w1, float* g_b1,
                                     const uint8_t* i1, ss_stream_t stream);
int ss_c5_conv2_wgrad_rc(const uint8_t* R, const float* st, int standardize, const float* w1, const float* b1,
                         const uint16_t* da_out, const uint8_t* idx, int N, float* g_w, float* g_b, ss_stream_t stream);

/* ---- a7: one bidirectional GRU layer, recurrence only ---------------------------------------
 * replaces pack_padded_sequence -> nn.GRU -> pad_packed_sequence (train_model_official.py:301-305).
 * gi      (2, B*T, 3H): W_ih x + b_ih per direction (forward, reverse), from ss_gemm_f32
 * lengths (B) int32, 1 <= len <= T
 * out     (B,T,2H): forward states in [0,H), reverse in [H,2H); zeros for t >= len
 * save    (2, B*T, 4, H) or NULL: r, z, n, W_hn h + b_hn for the backward pass
 * sync_ws NULL, or ss_gru_sync_bytes() bytes of device memory the caller zeroed ONCE and then leaves alone: with it,
 *         a batch small enough to leave most CUs idle runs each (16-clip slice, direction) on several CUs that
 *         exchange the state through `out` every step (same results, lower latency).  Launches that may run
 *         concurrently (different streams) need different sync_ws; launches on one stream can share one.
 * H in {64, 192} */
int ss_gru_fwd(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f,
               const float* b_hh_r, const int32_t* lengths, int B, int T, int H, float* out, float* save,
               void* sync_ws, ss_stream_t stream);
/* The same with nn.GRU's inter-layer dropout as a by-product: out_drop (B,T,2H) = out x the keep-scales of ss_dropout's Philox
 * stream at (drop_seed, drop_offset) -- what the next layer's input projection multiplies (train_model_official.py:261-267,
 * dropout=0.2 between the layers).  out_drop != NULL needs the multi-CU form (sync_ws given, ss_gru_sync_bytes(B,T,H) != 0);
 * otherwise SS_ERR_UNSUPPORTED and the caller runs ss_dropout on `out`. */
int ss_gru_fwd_drop(const float* gi, const float* w_hh_f, const float* w_hh_r, const float* b_hh_f, const float* b_hh_r,
                    const int32_t* lengths, int B, int T, int H, float* out, float* save, float* out_drop, float drop_p,
                    uint64_t drop_seed, uint64_t drop_offset, void* sync_ws, ss_stream_t stream);

/* bytes of sync_ws the multi-CU recurrence needs for this shape; 0 = the shape always takes the one-CU-per-slice
 * kernels (pass NULL). */
int ss_gru_sync_bytes(int B, int T, int H, long* bytes);

/* BPTT.  d_out (B,T,2H) gradient of the layer output; d_g (2, B*T, 4, H) receives
 * d(gi_r), d(gi_z), d(gi_n) and d(W_hn h + b_hn) = d(gi_n) * r; rows with t >= len are zero.
 * drop_p > 0: d_out is the gradient w.r.t. the DROPPED-OUT output (nn.GRU's inter-layer dropout); the mask is the
 * ss_dropout stream (drop_seed, drop_offset) over the (B*T, 2H) tensor and is applied while d_out is read.
 * g_bih_f, g_bhh_f, g_bih_r, g_bhh_r (each 3H floats; all NULL = not wanted): the bias gradients of the layer are summed up
 * while the recurrence runs and ADDED here (what ss_gru_bias_grad computes from d_g in a separate pass). */
int ss_gru_bwd(const float* d_out, const float* out, const float* save, const float* w_hh_f,
               const float* w_hh_r, const int32_t* lengths, int B, int T, int H, float* d_g,
               float drop_p, uint64_t drop_seed, uint64_t drop_offset, float* g_bih_f, float* g_bhh_f, float* g_bih_r,
               float* g_bhh_r, void* sync_ws, ss_stream_t stream);

/* bias gradients of one GRU layer from d_g: g_bih_* (3H) += colsum(d_g[dir][:, 0:3H]),
 * g_bhh_* (3H) += colsum(d_g[dir][:, 0:2H] | d_g[dir][:, 3H:4H]).  N = B*T rows per direction. */
int ss_gru_bias_grad(const float* d_g, int N, int H, float* g_bih_f, float* g_bhh_f, float* g_bih_r, float* g_bhh_r,
                     ss_stream_t stream);

/* ---- a8: AttnPool (train_model_official.py:231-248) ------------------------------------------
 * h (B,T,D), score weight (D) + bias (1); masked (-1e9) softmax over t, weighted sum.
 * attn (B,T) softmax weights (saved for backward), pooled (B,D). */
int ss_attn_pool_fwd(const float* h, const int32_t* lengths, const float* w_score, const float* b_score, int B,
                     int T, int D, float* attn, float* pooled, ss_stream_t stream);
/* d_h (B,T,D) is written (not accumulated); g_w (D), g_b (1) are accumulated. */
int ss_attn_pool_bwd(const float* h, const int32_t* lengths, const float* w_score, const float* attn,
                     const float* d_pooled, int B, int T, int D, float* d_h, float* g_w, float* g_b,
                     ss_stream_t stream);

/* ---- a9: head = LayerNorm -> Linear -> ReLU -> Dropout -> Linear (train_model_official.py:271-277)
 * The two Linear layers go through ss_gemm_f32; these are the LayerNorm halves.
 * xhat (B,D) normalised rows and rstd (B) are saved for the backward. */
int ss_layernorm_fwd(const float* x, const float* gamma, const float* beta, int B, int D, float eps, float* y,
                     float* xhat, float* rstd, ss_stream_t stream);
int ss_layernorm_bwd(const float* d_y, const float* xhat, const float* rstd, const float* gamma, int B, int D,
                     float* d_x, float* g_gamma, float* g_beta, ss_stream_t stream);
/* y = x * mask(philox(seed, offset)) / (1-p), or the same mask applied to a gradient; p = 0 is a no-op.
 * relu_of (may be NULL): additionally zero where relu_of <= 0 (ReLU backward fused). */
int ss_dropout(const float* x, float* y, long n, float p, uint64_t seed, uint64_t offset, const float* relu_of,
               ss_stream_t stream);

/* ---- a8+a9(+CE) fused: the whole classifier tail of one clip per workgroup ----------------------
 * AttnPool -> LayerNorm -> Linear(D,MID) -> ReLU -> Dropout(drop_p) -> Linear(MID,C), and, when y != NULL,
 * CrossEntropyLoss(label_smoothing)/denom with d(loss)/d(logits) (train_model_official.py:231-248, 271-277, 405).
 * Stash outputs (attn (B,T), xhat (B,D), rstd (B), ln (B,D) = LayerNorm output, mid (B,MID) = post-ReLU,
 * mid_d (B,MID) = after dropout) may each be NULL for inference.  Dropout draws the ss_dropout stream (seed, offset)
 * at element index b*MID + o. */
int ss_tail_fwd(const float* h, const int32_t* lengths, const float* w_score, const float* b_score, const float* gamma,
                const float* beta, const float* w1, const float* b1, const float* w4, const float* b4, const int64_t* y,
                int B, int T, int D, int MID, int C, float ln_eps, float drop_p, uint64_t seed, uint64_t offset,
                float label_smoothing, float denom, float* attn, float* xhat, float* rstd, float* ln, float* mid,
                float* mid_d, float* logits, float* d_logits, float* loss_sum, int32_t* correct, ss_stream_t stream);
/* Backward of the tail from d_logits (B,C) down to d_h (B,T,D) (written) and d_mid (B,MID) (written: the input of the
 * first Linear's weight-gradient GEMM); g_gamma, g_beta (D), g_wscore (D), g_bscore (1) are accumulated -- unless
 * col_part (B,3,D) is given: then every clip STORES its terms of g_gamma | g_beta | g_wscore there and the caller adds the
 * column sums (ss_colsum_f32, lda = 3 D) to the three gradients; 256 clips' float atomics on the same D addresses serialise.  The two
 * Linear weight/bias gradients are ss_gemm_f32 calls on (d_logits, mid_d) and (d_mid, ln). */
int ss_tail_bwd(const float* h, const int32_t* lengths, const float* w_score, const float* gamma, const float* w1,
                const float* w4, const float* attn, const float* xhat, const float* rstd, const float* mid,
                const float* d_logits, int B, int T, int D, int MID, int C, float drop_p, uint64_t seed,
                uint64_t offset, float* d_mid, float* d_h, float* g_gamma, float* g_beta, float* g_wscore,
                float* g_bscore, float* col_part, ss_stream_t stream);

/* ---- a10: loss, clip, Adam -------------------------------------------------------------------
 * CrossEntropyLoss(label_smoothing) mean-reduced over `denom` clips (the GLOBAL batch under data
 * parallelism), forward and d(loss)/d(logits) in one pass (train_model_official.py:405, 434-437).
 * loss_sum (1) is accumulated: sum over this call's clips of the per-clip loss / denom.
 * correct (1) int32 accumulated: argmax == y count (train_model_official.py:442). */
int ss_ce_ls_fwd_bwd(const float* logits, const int64_t* y, int B, int C, float label_smoothing, float denom,
                     float* d_logits, float* loss_sum, int32_t* correct, ss_stream_t stream);

/* a12: softmax over the classes + the k most probable of every clip, largest first
 * (topk_from_logits, live_infer_official.py:223-226: softmax -> argsort descending -> first k).
 * probs (B,k) f32, idx (B,k) int32 (class ids; -1 / 0 when k > C). */
int ss_softmax_topk(const float* logits, int B, int C, int k, float* probs, int32_t* idx, ss_stream_t stream);

/* sumsq (1) += sum x^2 over the flat gradient bucket (clip_grad_norm_'s global L2 norm). */
int ss_sumsq_f32(const float* x, long n, float* sumsq, ss_stream_t stream);

/* clip_grad_norm_(max_norm) folded into Adam (torch defaults, no weight decay, no amsgrad)
 * (train_model_official.py:403, 438-439): g' = g * grad_scale * min(1, max_norm / (sqrt(sumsq)*grad_scale + 1e-6)).
 * grad_scale = 1/world after a summing all-reduce.  step >= 1.  p, m, v updated in place. */
int ss_adam_clip(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float grad_scale,
                 float max_norm, float lr, float beta1, float beta2, float eps, int step, ss_stream_t stream);

/* strided row copy dst[r*ld_dst + c] = src[r*ld_src + c], c < cols (places X into Z, train_model_official.py:297) */
int ss_copy_rows_f32(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, ss_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SS_HOTPATH_H */
