/* nerfhip.h -- C ABI of libnerfhip.so: the MI355X (gfx950) NeRF render + training hot path.
 *
 * This is the drop-in boundary underneath the Python API of krrish94/nerf-pytorch.  The reference has no
 * FFI of its own for this path (it is pure PyTorch); every entry point below names the reference
 * function it replaces (file:line relative to the reference tree).  The binding a maintainer adds on the
 * reference side is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer marked "dev" is a device pointer (hipMalloc'ed / a torch CUDA
 *     tensor's data_ptr()), fp32 row-major contiguous unless a stride argument says otherwise;
 *   - returns 0 on success, a negative NERFHIP_ERR_* code otherwise; nerfhip_last_error() then returns a
 *     thread-local message.  Nothing throws across the ABI;
 *   - the library never allocates or frees caller-visible memory, never creates streams, never
 *     synchronises the device: work is enqueued on the `stream` argument (a hipStream_t passed as void*;
 *     NULL = the legacy default stream) and the caller owns all buffers for the duration of that work;
 *   - `plan` handles are host-only objects (weight packing tables, kernel schedules).
 *
 * Random draws: wherever the reference calls torch.rand / torch.randn the caller may pass the draws in
 * (parity mode: identical results for identical draws) or pass NULL and a (seed, ray_offset) pair, in
 * which case a counter-based Philox4x32-10 generator is evaluated in-kernel (production mode).  Stream
 * ids: 0 = stratified t_rand, 1 = coarse sigma noise, 2 = inverse-CDF u, 3 = fine sigma noise.
 * nerfhip_rng_fill() writes exactly the numbers the kernels would draw.
 */
#ifndef NERFHIP_H_
#define NERFHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nerfhip_stream_t;

#define NERFHIP_OK 0
#define NERFHIP_ERR_ARG (-1)
#define NERFHIP_ERR_UNSUPPORTED (-2)
#define NERFHIP_ERR_LAUNCH (-3)
#define NERFHIP_ERR_WORKSPACE (-4)

#define NERFHIP_RNG_UNIFORM 0
#define NERFHIP_RNG_NORMAL 1

/* ---- library --------------------------------------------------------------------------------------------------- */
int nerfhip_version(void);
const char* nerfhip_last_error(void);
/* 0 for the product library.  (The CPU wave-emulator build used by the test-suite returns 1.) */
int nerfhip_is_emulated(void);

/* Writes n draws of the in-kernel generator: element e of `stream_id` under `seed`, e = first .. first+n-1. */
int nerfhip_rng_fill(int kind, uint64_t seed, uint32_t stream_id, uint64_t first, int64_t n, float* out_dev,
                     nerfhip_stream_t stream);

/* Per-kernel timing: while enabled, every kernel launch is bracketed by HIP events recorded on the launch stream.
 * nerfhip_profile_report waits for them and writes "kernel_name launches total_ms\n" lines into buf, then clears. */
int nerfhip_profile_enable(int on);
int nerfhip_profile_report(char* buf, int64_t cap);
/* pre-creates the event pairs of `launches` launches (reused after every report), so that none is created while timing */
int nerfhip_profile_reserve(int64_t launches);
/* Shader clock under load: while profiling is enabled the three MLP kernels stamp every workgroup with the shader-clock
 * counter (s_memtime) and the constant 100 MHz counter (s_memrealtime).  out[3k + 0/1/2] = shader cycles / 100 MHz ticks /
 * workgroups summed over the workgroups of kernel k (0 = k_mlp_fwd16, 1 = k_mlp_dgrad16, 2 = k_wgrad):
 * clock = 0.1 GHz * out[3k] / out[3k + 1].  Waits for the device; clears the counters.  out: host uint64[9]. */
int nerfhip_profile_clocks(uint64_t* out);

/* ---- K1: rays -------------------------------------------------------------------------------------------------- */
/* get_ray_bundle (nerf/nerf_helpers.py:67-110) incl. meshgrid_xy (:28-40).  c2w: dev, rows >= 3, row stride
 * c2w_ld floats (columns 0..2 rotation, column 3 translation).  pixels: dev int64 linear ids (row*width+col) of
 * the n rays wanted, or NULL for the whole image (then n must be height*width; output is (H,W,3) row-major).
 * ray_origins / ray_directions: dev [n,3].  Directions are NOT normalised (reference behaviour). */
int nerfhip_ray_bundle(int height, int width, float focal, const float* c2w, int c2w_ld, const int64_t* pixels,
                       int64_t n, float* ray_origins, float* ray_directions, nerfhip_stream_t stream);

/* ndc_rays (nerf/nerf_helpers.py:170-197).  cw = -1/(W/(2*focal)), ch = -1/(H/(2*focal)), two_near = 2*near,
 * neg_two_near = -2*near are evaluated by the caller exactly as the reference's Python scalar arithmetic does. */
int nerfhip_ndc_rays(float near, float cw, float ch, float two_near, float neg_two_near, const float* rays_o,
                     const float* rays_d, int64_t n, float* out_o, float* out_d, nerfhip_stream_t stream);
/* Its vector-Jacobian product (what autograd computes through nerf/nerf_helpers.py:170-197 when the rays require grad --
 * pose optimisation on LLFF scenes): g_rays_o/d [n,3] = J^T (g_out_o, g_out_d) at (rays_o, rays_d). */
int nerfhip_ndc_rays_bwd(float near, float cw, float ch, float two_near, float neg_two_near, const float* rays_o,
                         const float* rays_d, const float* g_out_o, const float* g_out_d, int64_t n, float* g_rays_o,
                         float* g_rays_d, nerfhip_stream_t stream);

/* viewdirs + ray packing of run_one_iter_of_nerf (nerf/train_utils.py:143-168):
 * rays[n, 8|11] = [o(3) d(3) near far (d/||d||)(3)].  viewdir_src (dev [n,3]) is the PRE-ndc direction the
 * reference normalises; pass NULL for use_viewdirs = false (row width 8). */
int nerfhip_pack_rays(const float* rays_o, const float* rays_d, const float* viewdir_src, float near, float far,
                      int64_t n, float* rays_out, nerfhip_stream_t stream);

/* ---- K3: positional encoding ----------------------------------------------------------------------------------- */
/* positional_encoding (nerf/nerf_helpers.py:113-157).  x: dev [m,d]; freqs: dev [num_freqs] frequency bands as the
 * reference builds them; out: dev [m, d*(include_input + 2*num_freqs)] laid out
 * [x | sin(f0 x) | cos(f0 x) | sin(f1 x) | ...]. */
int nerfhip_positional_encoding(const float* x, int64_t m, int d, const float* freqs, int num_freqs, int include_input,
                                float* out, nerfhip_stream_t stream);

/* ---- K2: stratified depth samples ------------------------------------------------------------------------------ */
/* predict_and_render_radiance lines nerf/train_utils.py:38-65.  rays: dev rows of ray_stride floats with near/far in
 * columns 6/7; t_vals: dev [nc] = torch.linspace(0,1,nc); t_rand: dev [n,nc] uniform draws or NULL (in-kernel
 * stream 0).  z_out: dev [n,nc]. */
int nerfhip_stratified_z(const float* rays, int ray_stride, int64_t n, const float* t_vals, int nc, int lindisp,
                         int perturb, const float* t_rand, uint64_t seed, uint64_t ray_offset, float* z_out,
                         nerfhip_stream_t stream);

/* ---- K5: sigma/alpha compositing ------------------------------------------------------------------------------- */
/* cumprod_exclusive (nerf/nerf_helpers.py:43-64) over the last dimension of x[rows, cols]. */
int nerfhip_cumprod_exclusive(const float* x, int64_t rows, int cols, float* out, nerfhip_stream_t stream);

/* Its backward (autograd of nerf/nerf_helpers.py:43-64, as tiny_nerf.py:100-101 needs it): y = the forward result,
 * g_y = cotangent of y, g_x (out) = cotangent of x; all dev [rows, cols].  Division-free: exact for rows with zeros. */
int nerfhip_cumprod_exclusive_bwd(const float* x, const float* y, const float* g_y, int64_t rows, int cols, float* g_x,
                                  nerfhip_stream_t stream);

/* volume_render_radiance_field (nerf/volume_rendering_utils.py:6-53).  raw: dev [n,s,4]; z: dev [n,s];
 * rd: dev rows of rd_stride floats whose first 3 entries are the ray direction; noise: dev [n,s] N(0,1) draws or
 * NULL (in-kernel stream `rng_stream` when noise_std > 0).  Outputs (any may be NULL): rgb [n,3], disp [n],
 * acc [n], weights [n,s], depth [n]. */
int nerfhip_volume_render_fwd(const float* raw, const float* z, const float* rd, int rd_stride, int64_t n, int s,
                              float noise_std, const float* noise, uint64_t seed, uint32_t rng_stream,
                              uint64_t ray_offset, int white_background, float* rgb, float* disp, float* acc,
                              float* weights, float* depth, nerfhip_stream_t stream);

/* Closed-form backward of the above (what autograd computes for nerf/volume_rendering_utils.py:26-50).
 * g_rgb [n,3], g_depth [n], g_acc [n], g_weights [n,s] are the output cotangents (each may be NULL = 0; the
 * disparity cotangent is folded into g_depth/g_acc by the caller).  g_raw: dev [n,s,4]. */
int nerfhip_volume_render_bwd(const float* raw, const float* z, const float* rd, int rd_stride, int64_t n, int s,
                              float noise_std, const float* noise, uint64_t seed, uint32_t rng_stream,
                              uint64_t ray_offset, int white_background, const float* g_rgb, const float* g_depth,
                              const float* g_acc, const float* g_weights, float* g_raw, nerfhip_stream_t stream);

/* ---- K6: inverse-CDF importance sampling ----------------------------------------------------------------------- */
/* sample_pdf_2 (nerf/nerf_helpers.py:260-302) including the torchsearchsorted call (:288, side="right").
 * bins: dev [n,nbins]; weights: dev [n,nbins-1]; u: dev [n,nf] uniform draws, or NULL with det=1 (then u_det: dev
 * [nf] = torch.linspace(0,1,nf)) or NULL with det=0 (in-kernel stream 2).  samples: dev [n,nf];
 * inds (optional): dev int64 [n,nf] searchsorted result; cdf (optional): dev [n,nbins]. */
int nerfhip_sample_pdf(const float* bins, const float* weights, int64_t n, int nbins, const float* u, int det,
                       const float* u_det, int nf, uint64_t seed, uint64_t ray_offset, float* samples, int64_t* inds,
                       float* cdf, nerfhip_stream_t stream);

/* The hierarchical step of predict_and_render_radiance (nerf/train_utils.py:96-105): z_vals_mid, sample_pdf_2 on
 * weights[...,1:-1], detach, sort(cat(z_coarse, z_samples)).  z_coarse, weights: dev [n,nc]; z_samples (optional):
 * dev [n,nf]; z_fine: dev [n,nc+nf] ascending. */
int nerfhip_hierarchical_z(const float* z_coarse, const float* weights, int64_t n, int nc, const float* u, int det,
                           const float* u_det, int nf, uint64_t seed, uint64_t ray_offset, float* z_samples,
                           float* z_fine, nerfhip_stream_t stream);

/* ---- K4/K8: the MLP (models.FlexibleNeRFModel, nerf/models.py:185-256) ----------------------------------------- */
typedef struct nerfhip_model_cfg {
    int num_layers;         /* models.py:188 */
    int hidden_size;        /* models.py:189; 2..512 (kernel widths 64 / 128 / 256 / 512; other sizes ride zero-padded on the next width) */
    int skip_connect_every; /* models.py:190; cat(h, xyz) before layers_xyz[i] iff i % skip == 0 and i > 0 */
    int num_encoding_fn_xyz; /* 0..16 (every config .yml of the reference: <= 10) */
    int num_encoding_fn_dir; /* 0..10 (every config .yml of the reference: <= 4) */
    int include_input_xyz;
    int include_input_dir;
    int log_sampling_xyz;
    int log_sampling_dir;
    int use_viewdirs;
} nerfhip_model_cfg;

typedef struct nerfhip_plan* nerfhip_plan_t;

/* Host-only.  Returns NULL (and sets the error string) for an unsupported geometry. */
nerfhip_plan_t nerfhip_plan_create(const nerfhip_model_cfg* cfg);
/* Arithmetic of the plan's GEMMs.  FP32 (= nerfhip_plan_create): exact fp32 products on v_mfma_f32_16x16x4_f32 / 32x32x2 -- the
 * reference's own arithmetic, the path the headline benchmark runs; never replaced implicitly.
 *
 * F16X3 family (round 4): every GEMM as THREE fp16 MFMAs on operands split into two IEEE fp16 pieces, hi = f16(v), lo = f16(v - hi),
 * both round-to-nearest: x.w ~ xh.wh + xh.wl + xl.wh with fp32 accumulation.  A piece carries 11 significant bits and the rounding
 * error of hi is at most half an ulp, so hi + lo reproduces v to 2^-24 relative -- fp32's own rounding -- wherever the low piece is
 * a normal or subnormal fp16 number (the gfx950 matrix pipe multiplies fp16 subnormals exactly; measured: scripts/probe/
 * f16_mfma_probe.hip); the dropped xl.wl term is 2^-24 relative as well.  So a product block carries ~3 x 2^-24: fp32-grade
 * arithmetic, and the GPU parity suite holds these plans to the SAME bounds as the fp32 kernels (tests/tolerances.py).  What fp16's
 * 5-bit exponent costs is handled inside the library, all in exact powers of two: the packed weight pieces (and
 * biases) carry 2^8 -- so |w| must stay below 255.9; larger weights are saturated by nerfhip_pack_weights_plan, not turned into Inf --;
 * every SAMPLE carries the exponent of its current activations / d(pre-activation), chosen by the epilogue that produced them so that
 * its largest value lands in [2^13, 2^14) -- no activation range is out of reach, no cotangent too small --; the stash and every
 * output are plain fp32 values; the weight-gradient kernel splits a region at the power of two its producers recorded for it (one
 * word per region behind the stash / scratch, no host synchronisation) and divides it out exactly in its reduction.  Forward and
 * data gradient run with two waves per SIMD on v_mfma_f32_16x16x32_f16 (csrc/mlp_f16w.hip), the weight gradient on
 * v_mfma_f32_32x32x16_f16 (csrc/wgrad_f16.hip).
 *   F16X3            INFERENCE-ONLY plan: its own packed image (nerfhip_plan_packed_floats / nerfhip_plan_pack_index /
 *                    nerfhip_pack_weights_plan), serves nerfhip_mlp_fwd without a stash and the render entry points with
 *                    training = 0; every training / backward entry point refuses it.
 *   F16X3_FWD        training-capable: the forward passes on fp16 pieces (writes the same fp32 stash and ReLU masks), the backward
 *                    kernels the exact fp32 ones; the packed image holds the fp32 image and the fp16-piece image.
 *   F16X3_FWD_DGRAD  ... and the data-gradient chain on fp16 pieces; the weight-gradient GEMMs stay fp32.
 *   F16X3_TRAIN      ... and the large weight-gradient GEMMs (hidden x hidden blocks: ~94 % of those FLOPs) on the fp16 MFMAs, with
 *                    the thin blocks that share a region with one of them -- a skip layer's xyz columns, fc_alpha, the direction
 *                    columns -- riding along in the same launch; layer1's block and fc_rgb | fc_out stay on the fp32 kernel.
 * Supported for kernel widths 64, 128 and 256 (hidden_size <= 256; 64-wide nets: new in round 5 -- config/fern.yml's declared 4 x 64 --,
 * their weight-gradient GEMMs all stay on the fp32 kernel: _TRAIN is _FWD_DGRAD there), num_encoding_fn_xyz <= 10,
 * num_encoding_fn_dir <= 4.  Opt-in, labelled.
 * Values 1 .. 4 named round 3's bf16-piece plans (~2^-16 per product: they did not hold the parity bounds); removed in round 5,
 * nerfhip_plan_create_ex refuses them with a message, the numbers stay reserved. */
#define NERFHIP_PRECISION_FP32 0
#define NERFHIP_PRECISION_F16X3 5
#define NERFHIP_PRECISION_F16X3_FWD 6
#define NERFHIP_PRECISION_F16X3_FWD_DGRAD 7
#define NERFHIP_PRECISION_F16X3_TRAIN 8
nerfhip_plan_t nerfhip_plan_create_ex(const nerfhip_model_cfg* cfg, int precision);
int nerfhip_plan_precision(nerfhip_plan_t plan);
void nerfhip_plan_destroy(nerfhip_plan_t plan);
/* Number of fp32 parameters of the model = length of the flat parameter/gradient vector, laid out as the
 * concatenation of the reference state_dict tensors in registration order (layer1.weight, layer1.bias,
 * layers_xyz.{i}.weight/.bias, layers_dir.0.weight/.bias, fc_alpha.*, fc_rgb.*, fc_feat.* | fc_out.*). */
int64_t nerfhip_plan_num_params(nerfhip_plan_t plan);
int nerfhip_plan_dim_xyz(nerfhip_plan_t plan);
int nerfhip_plan_dim_dir(nerfhip_plan_t plan);
/* Number of tensors and, for tensor i, its name, offset into the flat vector and 2-D shape (cols = 0 for a bias). */
int nerfhip_plan_num_tensors(nerfhip_plan_t plan);
int nerfhip_plan_tensor_info(nerfhip_plan_t plan, int i, const char** name, int64_t* offset, int* rows, int* cols);
/* Human-readable kernel schedule of the plan (host-only): kernel width, and one line per weight-gradient job -- tiles, wave
 * grid, per-wave patch, split-K cost, and the thin weight block riding on it as side tiles, if any. */
int nerfhip_plan_describe(nerfhip_plan_t plan, char* buf, int64_t cap);
/* MFMA-packed weight image: number of floats, and the gather table (host int32[packed_floats]: source index into
 * the flat parameter vector, or -1 for zero padding). */
int64_t nerfhip_plan_packed_floats(nerfhip_plan_t plan);
int nerfhip_plan_pack_index(nerfhip_plan_t plan, int32_t* host_table);
/* packed[i] = table[i] >= 0 ? params[table[i]] : 0   (run once per optimiser step). */
int nerfhip_pack_weights(const float* params, const int32_t* table, int64_t n, float* packed, nerfhip_stream_t stream);
/* The same for any plan: fp32 plans gather as above; F16X3* plans write, per table entry, the bias word (times 2^8) or the two fp16
 * pieces hi = f16(2^8 w), lo = f16(2^8 w - hi) of the weight into the high / low blocks of the image.  table: dev
 * int32[packed_floats] (nerfhip_plan_pack_index), packed: dev, packed_floats 32-bit words. */
int nerfhip_pack_weights_plan(nerfhip_plan_t plan, const float* params, const int32_t* table, float* packed,
                              nerfhip_stream_t stream);
/* Bytes of activation stash a training forward over m sample points needs (0-filled is not required). */
int64_t nerfhip_plan_stash_bytes(nerfhip_plan_t plan, int64_t m);
/* Bytes of scratch nerfhip_mlp_bwd needs for m sample points. */
int64_t nerfhip_plan_bwd_scratch_bytes(nerfhip_plan_t plan, int64_t m);
/* Compacted backward (round 6; off by default).  sigma_a = relu(raw[..., 3] + noise) (nerf/volume_rendering_utils.py:38): wherever
 * that ReLU is off, and behind the sample at which a ray's transmittance reaches 0, a sample's weight is exactly 0, so its
 * d(loss)/d(raw) is exactly zero in all four channels -- and with it every d(pre-activation) row of that sample in every layer, a zero
 * term of every weight-gradient sum the reference's autograd (train_nerf.py:259) computes densely.  With the option on,
 * nerfhip_mlp_bwd (and the render backward entry points, which end in it) first lists the samples whose g_out row is not all zero
 * (ascending sample index; two small launches, no atomics, no host synchronisation), the data-gradient kernels walk that list and
 * the weight-gradient kernels sum over it: the same sums in a fixed order with their zero terms dropped -- equal to the dense
 * gradient up to the rounding of a different split of the sample range over the workgroups.  The forward and the stash are
 * unchanged.  A launch whose regions exceed 4 GiB (m >= 2^22 sample points) runs dense regardless.
 *   on = 0  dense (default);
 *   on = 1  compacted: the training forward writes the whole stash as always, the weight-gradient kernels gather the listed samples'
 *           rows out of it;
 *   on = 2  compacted AND recomputed, inside the fused render entry points (nerfhip_render_fwd / _bwd and their _parts forms): the
 *           training forward of this plan writes NO stash (it is the inference instantiation of the same kernel: bit-identical
 *           outputs), and the backward, once it has the list, re-runs the forward for the listed samples only, which leaves their
 *           activation rows and ReLU masks in list order -- the weight-gradient kernels then read contiguous blocks.  Pays where most
 *           rows are dropped and the stash-writing forward is much slower than the plain one (the fp16-piece plans: DESIGN.md).
 *           nerfhip_mlp_fwd / nerfhip_mlp_bwd, whose caller owns the stash between the two calls, treat 2 as 1.
 *   on = 3  fused (plans with an LDS-resident image only: fp32, hidden_size <= 64, view directions, at most 4 layers none of which is a
 *           skip layer, num_encoding_fn_xyz <= 10, num_encoding_fn_dir <= 4 -- config/fern.yml, config/llff.yml; other plans:
 *           NERFHIP_ERR_ARG), inside the fused render entry points: the training forward writes no stash, and ONE persistent kernel per
 *           net keeps all its weights in LDS, recomputes the forward of 128 sample points at a time, runs the data-gradient chain and
 *           sums the weight gradients in registers -- no stash, no d(pre-activation) images, no separate weight-gradient kernel; a
 *           fixed-order reduction of one partial per workgroup follows (no atomics: bit-reproducible).  Every sample is differentiated.
 *   on = 4  fused over the list: the same kernel walks the samples whose d(raw) row is not all zero (the list of mode 1).
 *   on = 5  fused over a register-image stash (same plans as 3): the training forward -- the persistent LDS-resident kernel -- also
 *           stores the registers the backward's chain works on (both encodings, every layer's activations: 64 L + 192 floats per
 *           sample point, whole-KiB stores, inside the plan's stash region of the render workspace), and the fused kernel reads them
 *           back (each array re-loaded for the next 64 samples right behind its last use) instead of recomputing the forward.  Same
 *           arithmetic on the same values in the same order as 3: the gradient is bit-identical; a third fewer multiplies in the
 *           backward for 2 x 1.8 KB of HBM traffic per sample point.  Every sample is differentiated.
 *           A render backward that must leave the d(pre-activation) images (nerfhip_render_bwd_rays with g_rays) runs 3 / 4 / 5 as 2;
 *           nerfhip_mlp_fwd / nerfhip_mlp_bwd treat them as 1. */
int nerfhip_plan_set_bwd_compaction(nerfhip_plan_t plan, int on);
int nerfhip_plan_bwd_compaction(nerfhip_plan_t plan);
/* Byte offset, inside a backward scratch for m sample points, of int32[2] = {samples the last compacted backward kept, samples of
 * that launch} (written on the launch stream; meaningful after a compacted nerfhip_mlp_bwd only).  Inside a render workspace the
 * scratch of a net is the region "bwd_scratch_coarse" / "bwd_scratch_fine" (nerfhip_render_workspace_region), m = n_rays * samples. */
int64_t nerfhip_plan_bwd_stats_offset(nerfhip_plan_t plan, int64_t m);
/* Frequency bands (host float[16] each) exactly as the reference builds them are supplied by the caller. */
int nerfhip_plan_set_freqs(nerfhip_plan_t plan, const float* freqs_xyz, const float* freqs_dir);

/* FlexibleNeRFModel.forward (nerf/models.py:233-256) on already-encoded rows x: dev [m, dim_xyz+dim_dir] ->
 * out: dev [m,4] = cat(rgb_raw, sigma_raw).  stash: NULL (inference) or dev buffer of
 * nerfhip_plan_stash_bytes(plan, m) for a later nerfhip_mlp_bwd. */
int nerfhip_mlp_fwd(nerfhip_plan_t plan, const float* packed, const float* x, int64_t m, float* out, void* stash,
                    nerfhip_stream_t stream);
/* Backward w.r.t. all parameters (what autograd computes for models.py:233-256): g_out: dev [m,4]; g_params: dev
 * flat gradient vector (overwritten); scratch: dev, nerfhip_plan_bwd_scratch_bytes. */
int nerfhip_mlp_bwd(nerfhip_plan_t plan, const float* packed, const float* g_out, int64_t m, const void* stash,
                    void* scratch, int64_t scratch_bytes, float* g_params, nerfhip_stream_t stream);
/* Gradient w.r.t. the encoded input x of the same forward (autograd gives it for models.py:233-256 when x requires
 * grad; the render path never does): call after nerfhip_mlp_bwd with the same m and scratch (it reads the
 * d(pre-activation) images left there).  params: dev flat parameter vector (reference layout, not the packed image);
 * g_x: dev [m, dim_xyz+dim_dir] (overwritten). */
int nerfhip_mlp_bwd_input(nerfhip_plan_t plan, const float* params, int64_t m, const void* scratch, float* g_x,
                          nerfhip_stream_t stream);

/* ---- fused render (predict_and_render_radiance, nerf/train_utils.py:28-127, + run_network :8-25) --------------- */
typedef struct nerfhip_render_cfg {
    int num_coarse;
    int num_fine; /* 0 = coarse only */
    int perturb;
    int lindisp;
    int white_background;
    float noise_std; /* radiance_field_noise_std */
    int ray_stride;  /* 8 or 11 floats per ray row */
} nerfhip_render_cfg;

/* Caller-supplied random draws (each may be NULL -> in-kernel generator). */
typedef struct nerfhip_render_rand {
    const float* t_rand;       /* [n, num_coarse]            stream 0 */
    const float* noise_coarse; /* [n, num_coarse]            stream 1 */
    const float* u;            /* [n, num_fine]              stream 2 */
    const float* noise_fine;   /* [n, num_coarse + num_fine] stream 3 */
} nerfhip_render_rand;

/* Outputs; any pointer may be NULL.  *_fine are ignored when num_fine == 0. */
typedef struct nerfhip_render_out {
    float* rgb_coarse;   /* [n,3] */
    float* disp_coarse;  /* [n]   */
    float* acc_coarse;   /* [n]   */
    float* depth_coarse; /* [n]   (not returned by the reference) */
    float* rgb_fine;
    float* disp_fine;
    float* acc_fine;
    float* depth_fine;
} nerfhip_render_out;

/* training: 0 = inference (no activation stash); 1 = training with one set of backward buffers per net (required when
 * the two nets' backward chains run on different streams, nerfhip_render_bwd_parts); 2 = training with ONE set of
 * backward buffers shared by the two nets (smaller: by the coarse net's d(pre-activation) scratch; valid for
 * nerfhip_render_bwd and for parts calls that pass NERFHIP_PART_SHARED_BWD). */
int64_t nerfhip_render_workspace_bytes(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine,
                                       const nerfhip_render_cfg* cfg, int64_t n_rays, int training);

/* Where a forward leaves its per-sample intermediates inside the workspace (byte offset and size), for inspection:
 * name = "z_coarse" [n,nc] (nerf/train_utils.py:58-65), "raw_coarse" [n,nc,4] (run_network's output, :70-77),
 * "weights_coarse" [n,nc] (:86), "z_fine" [n,nc+nf] (:103-105), "raw_fine" [n,nc+nf,4] (:108-115).  The reference
 * function returns none of them; the parity tests read the sample depths to count inverse-CDF index flips.
 * Training layouts also name "bwd_scratch_coarse" / "bwd_scratch_fine": the scratch each net's backward runs in. */
int nerfhip_render_workspace_region(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine, const nerfhip_render_cfg* cfg,
                                    int64_t n_rays, int training, const char* name, int64_t* offset, int64_t* bytes);

/* rays: dev [n, ray_stride]; t_vals: dev [num_coarse] linspace(0,1); u_det: dev [num_fine] linspace(0,1) (used
 * when perturb == 0); workspace: dev, nerfhip_render_workspace_bytes (keeps everything render_bwd needs when
 * training != 0). */
int nerfhip_render_fwd(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine, const nerfhip_render_cfg* cfg,
                       const float* rays, int64_t n_rays, const float* packed_coarse, const float* packed_fine,
                       const float* t_vals, const float* u_det, const nerfhip_render_rand* rnd, uint64_t seed,
                       uint64_t ray_offset, const nerfhip_render_out* out, void* workspace, int64_t workspace_bytes,
                       int training, nerfhip_stream_t stream);

/* Backward of the fused render w.r.t. both nets' parameters.  g_rgb_coarse / g_rgb_fine: dev [n,3] cotangents of
 * the two colour maps (the only outputs the reference's loss touches, train_nerf.py:244-258).  g_params_*: dev
 * flat gradient vectors (overwritten).  The workspace must be the one a training nerfhip_render_fwd filled (sized
 * with training = 1 or 2), with the same rays / packed weights / random arguments. */
int nerfhip_render_bwd(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine, const nerfhip_render_cfg* cfg,
                       const float* rays, int64_t n_rays, const float* packed_coarse, const float* packed_fine,
                       const nerfhip_render_rand* rnd, uint64_t seed, uint64_t ray_offset, const float* g_rgb_coarse,
                       const float* g_rgb_fine, void* workspace, int64_t workspace_bytes, float* g_params_coarse,
                       float* g_params_fine, nerfhip_stream_t stream);

/* The same two calls cut at the coarse / fine seam (`parts`: NERFHIP_PART_COARSE, NERFHIP_PART_FINE or both).  The
 * reference runs coarse forward -> fine forward -> one backward sequentially (nerf/train_utils.py:68-117,
 * train_nerf.py:244-259); the coarse net's backward only needs the coarse colour map, so a caller may enqueue it on a
 * second stream while the fine forward still runs, and launch the fine net's gradient all-reduce while the coarse
 * backward is in flight (TrainEngine does both).  The fine part reads what the coarse part left in the workspace
 * (depths, weights); each net's backward uses its own scratch region of the workspace.  Cotangents of the depth and
 * accumulation maps are optional (NULL = 0): losses that only touch the colour maps pass g_rgb_* alone. */
#define NERFHIP_PART_COARSE 1
#define NERFHIP_PART_FINE 2
/* nerfhip_render_bwd_parts only: the caller runs the two nets' backward chains one after the other on one stream, so
 * they may share one set of backward buffers (workspace sized with training = 2 or 1). */
#define NERFHIP_PART_SHARED_BWD 4
typedef struct nerfhip_render_cotangents {
    const float* g_rgb_coarse;   /* [n,3] */
    const float* g_acc_coarse;   /* [n]   */
    const float* g_depth_coarse; /* [n]   */
    const float* g_rgb_fine;
    const float* g_acc_fine;
    const float* g_depth_fine;
} nerfhip_render_cotangents;
int nerfhip_render_fwd_parts(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine, const nerfhip_render_cfg* cfg,
                             const float* rays, int64_t n_rays, const float* packed_coarse, const float* packed_fine,
                             const float* t_vals, const float* u_det, const nerfhip_render_rand* rnd, uint64_t seed,
                             uint64_t ray_offset, const nerfhip_render_out* out, void* workspace,
                             int64_t workspace_bytes, int training, int parts, nerfhip_stream_t stream);
int nerfhip_render_bwd_parts(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine, const nerfhip_render_cfg* cfg,
                             const float* rays, int64_t n_rays, const float* packed_coarse, const float* packed_fine,
                             const nerfhip_render_rand* rnd, uint64_t seed, uint64_t ray_offset,
                             const nerfhip_render_cotangents* g, void* workspace, int64_t workspace_bytes,
                             float* g_params_coarse, float* g_params_fine, int parts, nerfhip_stream_t stream);

/* nerfhip_render_bwd_parts that ALSO returns d(loss)/d(rays): under autograd the reference differentiates
 * pts = ro + rd * z (nerf/train_utils.py:67,107) and dists * ||rd|| (nerf/volume_rendering_utils.py:24) w.r.t. the ray
 * batch (pose optimisation).  params_*: dev flat parameter vectors (reference layout -- the packed images do not hold
 * the encoding columns in a usable order); tmp: dev scratch of nerfhip_render_bwd_rays_tmp_bytes; g_rays: dev
 * [n, ray_stride], overwritten: columns 0..2 d/d(origin), 3..5 d/d(direction), 8..10 d/d(viewdirs), the rest 0 (the
 * depths are constants of the ray: near / far carry no gradient, as for every loss the reference's scripts build).
 * g_rays == NULL (then params_* / tmp may be NULL) is exactly nerfhip_render_bwd_parts. */
int64_t nerfhip_render_bwd_rays_tmp_bytes(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine, const nerfhip_render_cfg* cfg,
                                          int64_t n_rays);
int nerfhip_render_bwd_rays(nerfhip_plan_t plan_coarse, nerfhip_plan_t plan_fine, const nerfhip_render_cfg* cfg,
                            const float* rays, int64_t n_rays, const float* packed_coarse, const float* packed_fine,
                            const nerfhip_render_rand* rnd, uint64_t seed, uint64_t ray_offset,
                            const nerfhip_render_cotangents* g, void* workspace, int64_t workspace_bytes,
                            float* g_params_coarse, float* g_params_fine, int parts, const float* params_coarse,
                            const float* params_fine, void* tmp, int64_t tmp_bytes, float* g_rays, nerfhip_stream_t stream);

/* ---- loss + optimiser (train_nerf.py:244-270) ------------------------------------------------------------------ */
/* mse_loss(rgb_coarse, target) + mse_loss(rgb_fine, target) and its cotangents; loss_out: dev float[3] =
 * {coarse_mse, fine_mse, sum}.  target rows have target_stride floats (RGB or RGBA; only [:3] is used). */
int nerfhip_mse_loss_fwd_bwd(const float* rgb_coarse, const float* rgb_fine, const float* target, int target_stride,
                             int64_t n, float grad_scale, float* g_rgb_coarse, float* g_rgb_fine, float* loss_out,
                             nerfhip_stream_t stream);
/* torch.optim.Adam single-tensor step (no amsgrad, no weight decay) on a flat vector; step counts from 1. */
int nerfhip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                      float beta1, float beta2, float eps, int64_t step, float grad_scale, nerfhip_stream_t stream);

/* ---- next to the path: training-ray selection and the 8-bit output stage -------- */
/* Distinct sample positions: out[i] = P(first + i), P = a keyed pseudo-random permutation of [0, population)
 * (6-round Feistel network, cycle-walking; round keys = Philox4x32-10(seed, step)).  Replaces
 * np.random.choice(population, n, replace=False) of train_nerf.py:185-189 and :219-221: distinct, uniformly
 * spread, reproducible, O(1) per index; ranks of a data-parallel job take disjoint [first, first+n) ranges of the
 * same permutation.  population <= 2^32, first + n <= population. */
int nerfhip_select_indices(uint64_t seed, uint64_t step, int64_t population, int64_t first, int64_t n, int64_t* out,
                           nerfhip_stream_t stream);

typedef struct nerfhip_select_cfg {
    int32_t height, width; /* select index k addresses pixel (row k % height, col k / height): the reference goes
                              through coords = stack(meshgrid_xy(arange(H), arange(W)), -1).reshape(-1, 2)
                              (train_nerf.py:214-225) */
    float focal;
    float near, far;       /* options.dataset.near / far (train_utils.py:164-165) */
    int32_t use_viewdirs;  /* rows of 11 floats (else 8) */
    int32_t ndc;           /* options.dataset.no_ndc is False: ndc_rays(H, W, focal, 1.0, ...) (train_utils.py:156-160) */
    float ndc_near, ndc_cw, ndc_ch, ndc_two_near, ndc_neg_two_near; /* as for nerfhip_ndc_rays */
    int32_t channels;      /* floats per target pixel (3 or 4); the loss reads [:3] */
    uint64_t seed, step;   /* permutation key, used when select_inds == NULL */
    int64_t first;         /* first permutation position of this rank */
} nerfhip_select_cfg;

/* Image branch of the training loop (train_nerf.py:210-227) fused with run_one_iter_of_nerf's ray packing
 * (train_utils.py:143-168): for n selected pixels generate ONLY their rays from the pose (c2w: dev, row-major,
 * row stride c2w_ld >= 4), write packed rows rays[n, 8|11] and gather target[n, channels] from image[H, W, channels]
 * (image/target may be NULL).  select_inds: dev int64 [n] flat select indices as the reference draws them, or NULL
 * to draw them here (nerfhip_select_indices with population H*W); inds_out (optional) receives the indices used. */
int nerfhip_select_rays(const nerfhip_select_cfg* cfg, const float* c2w, int c2w_ld, const float* image,
                        const int64_t* select_inds, int64_t n, float* rays, float* target, int64_t* inds_out,
                        nerfhip_stream_t stream);
/* Cached branch (train_nerf.py:175-194): rows of a stored ray bundle (ray_origins / ray_directions: dev
 * [population, 3]) and of targets[population, channels]. */
int nerfhip_select_cached_rays(const nerfhip_select_cfg* cfg, const float* ray_origins, const float* ray_directions,
                               const float* targets, int64_t population, const int64_t* select_inds, int64_t n,
                               float* rays, float* target, int64_t* inds_out, nerfhip_stream_t stream);

/* cast_to_image (eval_nerf.py:23-29): ToPILImage of a float image = mul(255) then byte conversion (truncation).
 * rgb: dev [pixels, in_channels >= 3] (first three used); out: dev uint8 [pixels, 3] (H, W, 3 byte order). */
int nerfhip_cast_to_image(const float* rgb, int in_channels, int64_t pixels, uint8_t* out, nerfhip_stream_t stream);
/* cast_to_disparity_image (eval_nerf.py:32-35): min-max normalise, clamp(0,1)*255, truncate.  NaN pixels make
 * min()/max() NaN in the reference, which zeroes the whole image -- reproduced.  scratch3: dev float[3]. */
int nerfhip_cast_to_disparity_image(const float* disparity, int64_t pixels, float* scratch3, uint8_t* out,
                                    nerfhip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFHIP_H_ */
