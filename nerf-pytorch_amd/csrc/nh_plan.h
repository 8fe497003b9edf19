// nh_plan.h -- host-side description of one FlexibleNeRFModel (nerf/models.py:185-256) mapped onto the MFMA kernels:
// flat parameter layout, packed-weight image, activation stash layout and the weight-gradient job list.
//
// Register/tile vocabulary (see DESIGN.md "MLP data layout"); the forward / data-gradient kernels (mlp16.hip) run on
// v_mfma_f32_16x16x4_f32:
//   * a wavefront owns 16 sample points; lane l = (j = l & 15: sample, g = l >> 4: k-group); an 8-wave workgroup is
//     128 samples = four 32-sample stash tiles (wave w: tile w >> 1, samples 16*(w & 1) ..);
//   * an activation of F features lives in F/4 registers per lane: register r of lane (j,g) holds feature
//       feat16(r,g) = 16*(r>>2) + 4*g + (r&3)
//     = the C/D layout of the instruction (tile r>>2, register r&3), and k-step r of the next layer consumes exactly
//     register r as its B operand (B[k=g][j]): activations never leave the register file;
//   * encodings use "slots": xyz 16 registers = 64 slots, dir 8 registers = 32 slots; the slot -> reference-column map
//     is baked into the packed weights; slot rows of the stash are numbered g*KR + r;
//   * a layer image is [bias: 512 floats][k-step][quad of 4 output tiles][64 lanes][4 floats]: one ds_read_b128 yields
//     the A operands of 4 tiles; the kernel streams it in chunks of a few k-steps covering ALL output tiles;
//   * stash / gradient regions are sample-major images [tile][32 samples][rows]: a lane's 4 consecutive registers
//     are 4 consecutive rows (one 16-byte store), and the weight-gradient GEMM (wgrad.hip) reads the operands of up to
//     four 32-row tiles of one sample with one LDS instruction.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/nerfhip.h"

constexpr int NH_MAX_LAYERS = 32;  // num_layers limit (per-layer offset tables travel by value in the kernel arguments)
constexpr int NH_MAX_JOBS = 64;  // weight-gradient jobs of one model (== the kernel-argument table of wgrad.hip)

// Encoding registers per lane group (slots = 4 x that) of the forward kernel: 16 / 8 cover the reference's defaults and
// every config/*.yml (num_encoding_fn_xyz <= 10, num_encoding_fn_dir <= 4); the extended instantiation (32 / 16) covers
// num_encoding_fn_xyz <= 16 and num_encoding_fn_dir <= 10.  nerfhip_plan::krx / krd say which one a plan uses.
constexpr int NH16_KRX = 16, NH16_KRX_EXT = 32;
constexpr int NH16_KRD = 8, NH16_KRD_EXT = 16;
static inline int nh_feat16(int r, int g) { return 16 * (r >> 2) + 4 * g + (r & 3); }
static inline int nh16_tq(int tiles) { return (tiles + 3) / 4; }
// bias block in front of every layer image (one float per output row, 16 per tile): 512 floats cover the 17 tiles of a
// 256-wide head, the 33 tiles of a 512-wide one need 1024
constexpr int nh16_bias_floats(int W) { return W >= 512 ? 1024 : 512; }
static inline int64_t nh16_image_floats(int kr, int tiles, int W) { return nh16_bias_floats(W) + (int64_t)kr * nh16_tq(tiles) * 256; }
// 32-bit words of ReLU bits per lane and layer: one bit per activation register (W / 4 of them), at least two words
constexpr int nh16_mask_words(int W) { return W >= 512 ? 4 : 2; }

// ---- fp16-piece images (mlp_f16w.hip, pack_f16.hip; plans created with NERFHIP_PRECISION_F16X3*) -------------------------------
// A layer image = 512 fp32 bias words (one per output row) + per (k-block kb, 16-row output tile t): a 1-KiB block of fp16 HIGH
// pieces and a 1-KiB block of fp16 LOW pieces, lane l of a block holding the 8 elements
// W[16 t + (l & 15)][in(kb, l >> 4, e)], e = 0..7, a k-block being 32 inputs deep.  Hidden inputs: in = nhw_unit(kb, g, e) -- the
// unit accumulator register e & 3 of output tile 2 kb + (e >> 2) holds for lane group g, so a layer's converted accumulators ARE
// the next layer's B operands.  Encoding inputs: slot 32 kb + 8 g + e (sin / cos of pair slot >> 1 = 3 f + axis in slots 2p, 2p+1;
// the raw coordinates in slots NS-4 .. NS-2).  Every weight element gives two fp16 pieces = one 32-bit word of image, so an image
// has 512 + nk * nt * 512 words -- which is also the length of its gather table (nerfhip_plan_pack_index: one source index per bias
// word, then one per weight element in (kb, t, lane, e) order).  Blocks are streamed in linear (kb, t) order, nhw_chunk_blocks of
// them per LDS chunk buffer.
static inline int64_t nhb_image_words(int nk, int nt) { return 512 + (int64_t)nk * nt * 512; }
static inline int nhw_unit(int kb, int g, int e) { return 32 * kb + 16 * (e >> 2) + 4 * g + (e & 3); }
constexpr int NHW_XBLOCKS = 2, NHW_DBLOCKS = 1;  // k-blocks of the xyz (64 slots) / direction (32 slots) encodings
constexpr int NHW_XSLOTS = 32 * NHW_XBLOCKS, NHW_DSLOTS = 32 * NHW_DBLOCKS;
constexpr int nhw_chunk_blocks(int W) { return W >= 256 ? 32 : 16; }  // 2-KiB (hi + lo) blocks per chunk buffer
constexpr int nhw_first_bytes(int nblocks, int W) { return 2048 + (nblocks < nhw_chunk_blocks(W) ? nblocks : nhw_chunk_blocks(W)) * 2048; }

// fp16-piece plans (include/nerfhip.h NERFHIP_PRECISION_F16X3*): `level` says which kernels run on the fp16 MFMAs -- 0 none (fp32),
// 1 the inference forward only (inference-only plan), 2 + the training forward, 3 + the data-gradient chain, 4 + the large
// weight-gradient blocks.  (Precisions 1 .. 4 were the bf16-piece plans of round 3: removed in round 5, the numbers stay reserved.)
static inline int nh_prec_level(int precision) { return precision >= NERFHIP_PRECISION_F16X3 ? precision - 4 : 0; }
static inline bool nh_prec_f16(int precision) { return precision >= NERFHIP_PRECISION_F16X3; }
// the packed weights (and biases) carry this exact power of two -- a weight's low piece is then a normal fp16 number down to
// |w| = 2^-10; what is multiplied with them carries per-SAMPLE exponents the kernels keep themselves (mlp_f16w.hip header)
constexpr float NHB_F16_WSCALE = 256.0f;

struct NhTensor {
    std::string name;
    int64_t off;
    int rows, cols;  // cols == 0 for a bias
};

// Offsets (in floats) of every layer inside the packed image, handed to the kernels by value.
struct NhPackedOffsets {
    int64_t f_layer1;
    int64_t f_xyz[NH_MAX_LAYERS];
    int64_t f_head;  // fc_feat + fc_alpha rows (viewdirs) or fc_out (no viewdirs)
    int64_t f_dir;
    int64_t f_rgb;
    int64_t b_rgb;   // transposed images for the data-gradient chain
    int64_t b_dir;
    int64_t b_head;
    int64_t b_xyz[NH_MAX_LAYERS];
};

// Words behind a training stash / a backward scratch in which the fp16 kernels record the largest magnitude stored per region
// (mlp_f16w.hip `rmax`; indices: H[k] / P[k] -> k, FEAT / PFEAT -> L, DIRH / PDIR -> L + 1): zeroed before the producing launch
constexpr int NH_RMAX_WORDS = 64;
// ... and for the regions that ride as guests on the large weight-gradient blocks: the xyz / direction slot regions of the stash, the
// d(raw output) region of the scratch
constexpr int nh_rmax_x(int L) { return L + 2; }
constexpr int nh_rmax_d(int L) { return L + 3; }
constexpr int nh_rmax_pout(int L) { return L + 2; }

// Activation stash regions: [tiles][32 samples][rows] each, `row_prefix` rows precede it inside a tile group.
// 512-wide nets: a 512-row activation is stored as TWO consecutive 256-row regions (rows 0..255 at row_prefix, rows
// 256..511 at row_prefix + 256; NhRegion::rows == 256 names the first half): the weight-gradient kernel reads whole
// regions as contiguous blocks and its workgroups hold at most 8 x 8 accumulator tiles.
struct NhRegion {
    int rows;
    int64_t row_prefix;  // offset (floats) = 32 * n_tiles * row_prefix
};
struct NhStashLayout {
    NhRegion X, D, H[NH_MAX_LAYERS], FEAT, DIRH;
    int64_t total_rows;
    // After the row regions: ReLU bit masks, [16-sample wave tile][n_masks][64 lanes][nh16_mask_words(W)]; mask index
    // k-1 = H_k (k >= 1), L-1 = FEAT, L = DIRH.
    int n_masks;
};
struct NhGradLayout {  // d(pre-activation) scratch written by the data-gradient kernel
    NhRegion P[NH_MAX_LAYERS], PFEAT, PDIR, POUT;
    int64_t total_rows;
};

// One weight-gradient job: dW[out_rows, in_rows] = sum_samples A[out_row][sample] * B[in_row][sample].
struct NhJob {
    int a_region_rows;    // rows of the A region (tile stride = rows*32 floats)
    int64_t a_row_prefix; // A region offset = 32*n_tiles*a_row_prefix (in the grad scratch)
    int a_tiles;          // 32-row tiles of A covered by this job (starting at row 0 of the region)
    int b_region_rows;
    int64_t b_row_prefix; // B region offset in the activation stash
    int b_row0;           // first row of B covered
    int b_tiles;
    int wo, wi, po, pi;   // wave grid (wo*wi <= 8) and per-wave patch in tiles (po*pi <= 8)
    // unpack: out row r is real iff r_lo <= r < r_hi and maps to parameter row (r - r_lo)
    int r_lo, r_hi;
    int64_t w_off;        // flat offset of the weight tensor
    int w_ld;             // its column count
    int col_kind;         // 0: col = col_base + in_row (valid iff < col_base + col_count); 1: xyz slot map; 2: dir slot map
    int col_base, col_count;
    int64_t bias_off;     // flat offset of the bias tensor, or -1 (only one job per layer carries the bias)
    int cost;             // relative time one workgroup spends per sample tile (split-K allocation)
    // A second, thin weight block riding on this job (wgrad.hip "side tiles"): kind 1 = ONE extra A tile (a 32-row region of
    // the gradient scratch) against this job's B tiles; kind 2 = side_tiles (1 | 2) extra B tiles (a 32- / 64-row region of
    // the stash) against this job's A tiles.  s_*: how the reduce kernel unpacks it (as r_lo .. bias_off above).
    int side_kind, side_rows, side_tiles;
    int64_t side_row_prefix;
    int s_r_lo, s_r_hi, s_w_ld, s_col_kind, s_col_base, s_col_count;
    int64_t s_w_off, s_bias_off;
};

// A weight block whose gradient the fp16-piece weight-gradient kernel computes (NERFHIP_PRECISION_F16X3_TRAIN; wgrad_f16.hip):
// dW[r][c] = sum_samples A[r][s] B[c][s] for r < r_hi, c < col_count, written to w_off + r * w_ld + c; bias = row sums of A.
struct NhJobB {
    int a_idx, b_idx;                    // the regions' slots among the recorded maxima (NH_RMAX_WORDS)
    int a_rows, b_rows;                  // rows of the A region (d(pre-activation) image) and of the B region (activation stash)
    int64_t a_row_prefix, b_row_prefix;  // region offsets = 32 * n_tiles * prefix floats
    int r_hi, col_count, w_ld;
    int64_t w_off, bias_off;
    // A thin block riding on this one (wgrad_f16.hip "SA / SB"), taken out of `jobs`: side_kind 1 = a 32-row region of the gradient
    // scratch as ONE more A tile against this block's B region (side rows s_r_lo .. s_r_hi - 1 are the guest's parameter rows);
    // 2 = a 32- / 64-row slot region of the stash as more B tiles against this block's A region (columns through the slot map
    // s_col_kind: 1 xyz, 2 dir).  side_idx: the region's slot among the recorded maxima.
    int side_kind, side_rows, side_idx;
    int64_t side_row_prefix;
    int s_r_lo, s_r_hi, s_w_ld, s_col_kind, s_col_base, s_col_count;
    int64_t s_w_off, s_bias_off;
};

struct nerfhip_plan {
    nerfhip_model_cfg cfg;
    int W, H, L, skip, Dx, Dd, view;  // W: kernel width (64 | 128 | 256 | 512) >= H: the model's hidden_size (units H..W-1 are zero padding)
    std::vector<NhTensor> tensors;
    int64_t nparams;
    int t_layer1_w, t_layer1_b, t_xyz_w[NH_MAX_LAYERS], t_xyz_b[NH_MAX_LAYERS];
    int t_dir_w, t_dir_b, t_alpha_w, t_alpha_b, t_rgb_w, t_rgb_b, t_feat_w, t_feat_b, t_out_w, t_out_b;
    int precision;                    // NERFHIP_PRECISION_*: nh_prec_level() / nh_prec_f16() above
    NhPackedOffsets pob;              // f16x3 plans: word offsets of the fp16-piece layer images inside the packed buffer
    int64_t packed32_floats;          // words of the fp32 image in front of them (0 for _F16X3, the whole buffer for _FP32)
    int xyz_slot_b[NHW_XSLOTS];       // f16x3 plans: encoding slot -> reference column, or -1
    int dir_slot_b[NHW_DSLOTS];
    int krx, krd;                     // encoding registers per lane group: NH16_KRX / NH16_KRD or the _EXT pair
    int xyz_col16[4][NH16_KRX_EXT];  // slot (r,g) -> reference column of the xyz encoding, or -1  (rows of krx entries)
    int dir_col16[4][NH16_KRD_EXT];
    int xyz_slot_col[4 * NH16_KRX_EXT];  // stash slot row g*krx + r -> reference column, used by the weight-gradient scatter
    int dir_slot_col[4 * NH16_KRD_EXT];
    float freqs_xyz[16], freqs_dir[16];
    bool freqs_set;
    NhPackedOffsets po;
    int64_t packed_floats;
    NhStashLayout stash;
    NhGradLayout grad;
    std::vector<NhJob> jobs;
    std::vector<NhJobB> bjobs;        // (F16X3_TRAIN: the large blocks, taken out of `jobs`)
    int wgrad_waves;         // waves per workgroup of the weight-gradient kernel: 8 (256- and 512-wide nets) or 4 (narrower)
    // nerfhip_plan_set_bwd_compaction: 0 dense; 1 the backward drops the samples whose d(raw output) row is zero (list + gathered stash
    // rows); 2 the same, and inside the fused render the training forward writes NO stash -- the backward re-runs the forward for the
    // listed samples only and leaves a compacted stash (no gather in the weight-gradient kernels); 3 / 4 (plans with a resident image,
    // nh_r64.h, inside the fused render): the stash-free forward and ONE fused backward kernel over every sample (3) / over the list (4)
    int bwd_compact = 0;
    // float offset, inside the packed buffer, of the LDS-resident image of the fused backward for 64-wide nets (nh_r64.h), or -1
    int64_t r64_off = -1;
    bool is_skip(int i) const { return i % skip == 0 && i > 0; }
};
// floats of a training stash of `tiles` 32-sample tiles: the row regions, then the ReLU masks (the region maxima follow)
static inline int64_t nh_stash_floats(const nerfhip_plan* p, int64_t tiles) {
    return tiles * (p->stash.total_rows * 32 + (int64_t)p->stash.n_masks * 128 * nh16_mask_words(p->W));
}
