// nh_r64.h -- the LDS-resident weight image of the fused backward for 64-wide nets (mlp64r.hip), shared by the kernel and by the
// host code that lays it out behind a plan's packed images (plan.cpp) and fills its gather table.
//
// Which plans have one (nh_r64_eligible): fp32, kernel width 64, view directions, 1..4 layers, no skip layer, the default encoding
// registers -- config/fern.yml / config/llff.yml nets (4 x 64, skip_connect_every 3: no layer of theirs is a skip layer).
//
// The image is what one workgroup keeps in LDS for its whole life: every weight matrix ROW-MAJOR [out row][k column], k column = the
// feature index of the input (hidden inputs: the unit; encodings: nh_feat16(r, g) of slot register r of lane group g), rows of 64
// floats (no padding) whose sixteen 16-byte chunks are XOR-swizzled with the row: chunk q of row r sits at chunk q ^ (r & 15)
// (r64_pos).  ONE copy then serves both orientations without bank conflicts (MI355X_MICROARCH.md, LDS: ds_read_b128 is served in
// four groups of 16 lanes -- {0-3, 12-15, 20-27}, ... -- over 64 banks, ds_read_b32 in two groups of 32 lanes over 32 banks):
//   forward   A[i = out row][k]: lane (g, i) reads W[16 t + i][16 R + 4 g .. + 3] -- one ds_read_b128 = four k-steps -- at chunk
//             (4 R + g) ^ i: a group's 8 lanes of one g and 8 of the next cover all 16 chunk positions (the first build, rows padded
//             to 68 floats, measured SQ_LDS_BANK_CONFLICT = 22 % of the LDS cycles: that rule assumed contiguous quarter waves);
//   transposed A'[i = in unit][k = out row]: lane (g, i) reads W[nh_feat16(r, g)][16 t + i] -- one ds_read_b32 per k-step -- at
//             dword 16 (t ^ g) + 4 ((i >> 2) ^ c) + (i & 3) of its row: 32 lanes, 32 banks.
// Segments: RES (layers_xyz, fc_feat + fc_alpha, layers_dir, fc_rgb, every bias), then L1 (layer1's weights: only the forward reads
// them); one copy of the whole image at kernel start.  Behind it in LDS: the operand-exchange area (R64_TILES tiles).
#pragma once
#include "nh_plan.h"

constexpr int R64_MAX_LAYERS = 4;
#ifdef NH64_PAD_LAYOUT  // (A/B builds only: the first build's padded rows, measured 22 % bank-conflict cycles)
constexpr int R64_S = 68, R64_SD = 100, R64_SR = 36;
constexpr int r64_pos(int, int kc) { return kc; }
#else
constexpr int R64_S = 64;    // row stride of a matrix with 64 k columns
constexpr int R64_SD = 128;  // layers_dir: 64 hidden columns, then a 64-float half row that holds the 32 direction slots
constexpr int R64_SR = 36;   // fc_rgb: 32 hidden (+ 4), not swizzled (read once per round)
// float offset of k column kc (< 64) inside (the 64-float half of) row `row`
constexpr int r64_pos(int row, int kc) { return 4 * ((kc >> 2) ^ (row & 15)) + (kc & 3); }
#endif
constexpr int R64_WAVES = 8;                    // waves per workgroup: R64_TILES chain waves + R64_TILES weight-gradient waves
constexpr int R64_TILES = 4;                    // 16-sample tiles per round (one per chain wave)
constexpr int R64_TILE_BLOCKS = 9;              // 16-feature x 16-sample blocks one tile may hold in the exchange area
constexpr int R64_TILE_F = R64_TILE_BLOCKS * 256;
constexpr int R64_EX_F = R64_TILES * R64_TILE_F;  // floats of the exchange area (36 KiB)

struct R64Layout {
    int xyz[R64_MAX_LAYERS];  // layers_xyz[i]: 64 rows x R64_S
    int head;                 // fc_feat rows 0..63, fc_alpha row 64: 65 rows x R64_S
    int dir;                  // layers_dir[0]: 32 rows x R64_SD
    int rgb;                  // fc_rgb: 4 rows x R64_SR (row 3 zero)
    int b_l1, b_xyz[R64_MAX_LAYERS], b_feat, b_alpha, b_dir, b_rgb;
    int res_floats;           // RES segment, a multiple of 256 floats (1-KiB copy pieces)
    int l1;                   // layer1: 64 rows x R64_S, at res_floats in the image (R64_L1_F floats: whole 1-KiB pieces)
    int image_floats;
};
constexpr int r64_up(int v, int m) { return (v + m - 1) / m * m; }
constexpr R64Layout r64_layout(int L) {
    R64Layout y = {};
    int off = 0;
    for (int i = 0; i < L - 1; ++i) {
        y.xyz[i] = off;
        off += 64 * R64_S;
    }
    y.head = off;
    off += 65 * R64_S;
    off = r64_up(off, 4);
    y.dir = off;
    off += 32 * R64_SD;
    y.rgb = off;
    off += 4 * R64_SR;
    y.b_l1 = off;
    off += 64;
    for (int i = 0; i < L - 1; ++i) {
        y.b_xyz[i] = off;
        off += 64;
    }
    y.b_feat = off;
    off += 64;
    y.b_alpha = off;
    off += 16;
    y.b_dir = off;
    off += 32;
    y.b_rgb = off;
    off += 16;
    y.res_floats = r64_up(off, 256);
    y.l1 = y.res_floats;
    y.image_floats = y.res_floats + r64_up(64 * R64_S, 256);  // (64 * 64 floats = 16 copy pieces)
    return y;
}
// ... and the hand-over area: per tile the 16 registers of the xyz encoding and the 8 of the direction encoding,
// [register quad][lane][4] (the weight-gradient waves encode the next round while the chain waves are in their forward: mlp64r.hip)
constexpr int R64_PASS_TILE_F = 24 * 64;
constexpr int R64_PASS_F = R64_TILES * R64_PASS_TILE_F;
constexpr int r64_lds_floats(int L) { return r64_layout(L).image_floats + R64_EX_F + R64_PASS_F; }
static_assert((r64_lds_floats(R64_MAX_LAYERS) + 32) * 4 + 32 <= 160 * 1024, "the image + the exchange area must fit the 160 KB of LDS");

// The register-image stash of the fused backward's stashed variant (mode 5, mlp64r.hip k_fwd64r<L, true> -> k_bwd64r<L, true>): per
// 16-sample tile the registers the chain waves would otherwise recompute, exactly as a wave holds them -- [quad][lane][4], quad = four
// consecutive registers of an activation: X (4 quads), D (2), H_0 .. H_{L-1} (4 each), FEAT (4), DIRH (2).  Every store / load
// instruction moves one whole KiB; 64 L + 192 floats per sample point (4 x 64: 1792 B -- what the general stash takes: the region
// of the render workspace is the same one).
constexpr int R64_SQ_X = 0, R64_SQ_D = 4, R64_SQ_H = 6;
constexpr int r64_sq_feat(int L) { return R64_SQ_H + 4 * L; }
constexpr int r64_sq_dirh(int L) { return R64_SQ_H + 4 * L + 4; }
constexpr int r64_stash_quads(int L) { return 4 * L + 12; }
constexpr int r64_stash_tile_floats(int L) { return r64_stash_quads(L) * 256; }

// Accumulators of one weight-gradient wave (mlp64r.hip "units"): NU accumulator tiles (4 registers each) + NB row-sum registers (biases)
constexpr int r64_units(int L) { return 9 + 4 * L; }
constexpr int r64_bias_regs(int L) { return 3 + L; }
constexpr int r64_regs(int L) { return 4 * r64_units(L) + r64_bias_regs(L); }
// floats of one workgroup's partial: [weight-gradient wave][register][lane]
constexpr int r64_partial_floats(int L) { return R64_TILES * r64_regs(L) * 64; }

static inline bool nh_r64_eligible(const nerfhip_plan* p) {
    if (p->precision != NERFHIP_PRECISION_FP32 || p->W != 64 || !p->view || p->L < 1 || p->L > R64_MAX_LAYERS || p->krx != NH16_KRX ||
        p->krd != NH16_KRD)
        return false;
    for (int i = 0; i < p->L - 1; ++i)
        if (p->is_skip(i)) return false;
    return true;
}
// ... and its stashed variant fits the stash region the render workspace reserves for the plan (nh_stash_floats per 32-sample tile)
static inline bool nh_r64_stash_fits(const nerfhip_plan* p) {
    return nh_r64_eligible(p) && nh_stash_floats(p, 1) >= 2 * (int64_t)r64_stash_tile_floats(p->L);
}
