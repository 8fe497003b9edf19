// plan.cpp -- host-only: builds the plan of nh_plan.h for a FlexibleNeRFModel geometry (nerf/models.py:185-231).
#include <stdlib.h>
#include <string.h>

#include <functional>

#include "nh_host.h"
#include "nh_mlp.h"
#include "nh_plan.h"
#include "nh_r64.h"

namespace {

int add_tensor(nerfhip_plan* p, const std::string& name, int rows, int cols) {
    NhTensor t;
    t.name = name;
    t.off = p->nparams;
    t.rows = rows;
    t.cols = cols;
    p->nparams += (int64_t)rows * (cols ? cols : 1);
    p->tensors.push_back(t);
    return (int)p->tensors.size() - 1;
}

// slot (r,g) -> reference column.  Group g < 3 owns the (frequency, axis) pairs [g*C, (g+1)*C), C = KR/2; group 3 owns
// [3C, 3C + (KR-3)/2) and carries the raw coordinates in its last three registers.  Pair p = 3*f + axis sits in
// registers 2q (sin), 2q+1 (cos), q = p - g*C.
bool build_slot_map16(int L, int include_input, int kr, int* col_flat, int row_stride) {
    auto col = [&](int g) { return col_flat + g * row_stride; };
    const int P = 3 * L, C = kr / 2, C3 = (kr - 3) / 2, base = include_input ? 3 : 0;
    if (P > 3 * C + C3) return false;
    for (int g = 0; g < 4; ++g)
        for (int r = 0; r < row_stride; ++r) col(g)[r] = -1;
    for (int g = 0; g < 4; ++g)
        for (int q = 0; q < (g < 3 ? C : C3); ++q) {
            const int pr = g * C + q;
            if (pr >= P) continue;
            const int f = pr / 3, a = pr % 3;
            col(g)[2 * q] = base + 6 * f + a;
            col(g)[2 * q + 1] = base + 6 * f + 3 + a;
        }
    if (include_input)
        for (int a = 0; a < 3; ++a) col(3)[kr - 3 + a] = a;
    return true;
}

struct GemmSpec16 {
    int kr = 0, tiles = 0;                    // k-steps (B registers), 16-row output tiles
    std::function<int64_t(int, int, int)> w;  // (out_row, r, g) -> flat param index or -1
    std::function<int64_t(int)> b;            // out_row -> flat param index or -1
};

void fill_spec16(const GemmSpec16& s, int64_t off, int32_t* table, int W) {
    const int tq = nh16_tq(s.tiles), bfl = nh16_bias_floats(W);
    for (int i = 0; i < bfl; ++i) table[off + i] = (i < 16 * s.tiles && s.b) ? (int32_t)s.b(i) : -1;
    int32_t* img = table + off + bfl;
    for (int r = 0; r < s.kr; ++r)
        for (int q = 0; q < tq; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int t = 4 * q + e;
                    const int64_t src = t < s.tiles ? s.w(16 * t + (lane & 15), r, lane >> 4) : -1;
                    img[(((int64_t)r * tq + q) * 64 + lane) * 4 + e] = (int32_t)src;
                }
}

struct Specs16 {
    GemmSpec16 f_layer1, f_xyz[NH_MAX_LAYERS], f_head, f_dir, f_rgb, b_rgb, b_dir, b_head, b_xyz[NH_MAX_LAYERS];
};

// Kernel width W (64, 128, 256 or 512) >= the model's hidden_size H: units H..W-1 (H/2..W/2-1 of the direction layer) are
// padding -- every weight and bias of a padded unit is the constant 0 (index -1), so it stays exactly 0 through the
// forward chain, its ReLU bit is 0, and nothing flows through it in the backward chain.
void build_specs16(const nerfhip_plan* p, Specs16& S) {
    const int W = p->W, H = p->H, H2 = H / 2, KH = W / 4, Dx = p->Dx, Dd = p->Dd, L = p->L, TW = W / 16;
    auto T = [p](int idx) { return p->tensors[idx]; };
    {
        GemmSpec16& s = S.f_layer1;
        s.kr = p->krx;
        s.tiles = TW;
        NhTensor w = T(p->t_layer1_w), b = T(p->t_layer1_b);
        s.w = [=](int o, int r, int g) -> int64_t {
            int c = p->xyz_col16[g][r];
            return (o < H && c >= 0) ? w.off + (int64_t)o * Dx + c : -1;
        };
        s.b = [=](int o) -> int64_t { return o < H ? b.off + o : -1; };
    }
    for (int i = 0; i < L - 1; ++i) {
        GemmSpec16& s = S.f_xyz[i];
        const bool sk = p->is_skip(i);
        s.kr = KH + (sk ? p->krx : 0);
        s.tiles = TW;
        NhTensor w = T(p->t_xyz_w[i]), b = T(p->t_xyz_b[i]);
        const int ld = H + (sk ? Dx : 0);
        s.w = [=](int o, int r, int g) -> int64_t {
            if (o >= H) return -1;
            if (r < KH) return nh_feat16(r, g) < H ? w.off + (int64_t)o * ld + nh_feat16(r, g) : -1;
            int c = p->xyz_col16[g][r - KH];
            return c >= 0 ? w.off + (int64_t)o * ld + H + c : -1;
        };
        s.b = [=](int o) -> int64_t { return o < H ? b.off + o : -1; };
        GemmSpec16& bt = S.b_xyz[i];  // dh_in[f] = sum_u W[u][f] dpre[u]   (hidden columns only)
        bt.kr = KH;
        bt.tiles = TW;
        bt.w = [=](int f, int r, int g) -> int64_t {
            return (nh_feat16(r, g) < H && f < H) ? w.off + (int64_t)nh_feat16(r, g) * ld + f : -1;
        };
    }
    if (p->view) {
        NhTensor fw = T(p->t_feat_w), fb = T(p->t_feat_b), aw = T(p->t_alpha_w), ab = T(p->t_alpha_b);
        NhTensor dw = T(p->t_dir_w), db = T(p->t_dir_b), rw = T(p->t_rgb_w), rb = T(p->t_rgb_b);
        {
            GemmSpec16& s = S.f_head;  // rows 0..W-1 = fc_feat, row W = fc_alpha
            s.kr = KH;
            s.tiles = TW + 1;
            s.w = [=](int o, int r, int g) -> int64_t {
                const int f = nh_feat16(r, g);
                if (f >= H) return -1;
                if (o < H) return fw.off + (int64_t)o * H + f;
                if (o == W) return aw.off + f;
                return -1;
            };
            s.b = [=](int o) -> int64_t { return o < H ? fb.off + o : (o == W ? ab.off : -1); };
        }
        {
            GemmSpec16& s = S.f_dir;
            s.kr = KH + p->krd;
            s.tiles = TW / 2;
            const int ld = H + Dd;
            s.w = [=](int o, int r, int g) -> int64_t {
                if (o >= H2) return -1;
                if (r < KH) return nh_feat16(r, g) < H ? dw.off + (int64_t)o * ld + nh_feat16(r, g) : -1;
                int c = p->dir_col16[g][r - KH];
                return c >= 0 ? dw.off + (int64_t)o * ld + H + c : -1;
            };
            s.b = [=](int o) -> int64_t { return o < H2 ? db.off + o : -1; };
        }
        {
            GemmSpec16& s = S.f_rgb;
            s.kr = KH / 2;
            s.tiles = 1;
            s.w = [=](int o, int r, int g) -> int64_t {
                return (o < 3 && nh_feat16(r, g) < H2) ? rw.off + (int64_t)o * H2 + nh_feat16(r, g) : -1;
            };
            s.b = [=](int o) -> int64_t { return o < 3 ? rb.off + o : -1; };
        }
        {
            GemmSpec16& s = S.b_rgb;  // d(dir hidden)[f] = sum_{rho<3} Wrgb[rho][f] d_rgb[rho]; ONE k-step: group g carries rho = g
            s.kr = 1;
            s.tiles = TW / 2;
            s.w = [=](int f, int r, int g) -> int64_t { return (r == 0 && g < 3 && f < H2) ? rw.off + (int64_t)g * H2 + f : -1; };
        }
        {
            GemmSpec16& s = S.b_dir;  // d(feat)[f] = sum_u Wdir[u][f] dpre_dir[u]
            s.kr = KH / 2;
            s.tiles = TW;
            const int ld = H + Dd;
            s.w = [=](int f, int r, int g) -> int64_t {
                return (nh_feat16(r, g) < H2 && f < H) ? dw.off + (int64_t)nh_feat16(r, g) * ld + f : -1;
            };
        }
        {
            GemmSpec16& s = S.b_head;  // dh[f] = sum_u Wfeat[u][f] dpre_feat[u] + Walpha[0][f] d_alpha (k-step KH, group 0)
            s.kr = KH + 1;
            s.tiles = TW;
            s.w = [=](int f, int r, int g) -> int64_t {
                if (f >= H) return -1;
                if (r < KH) return nh_feat16(r, g) < H ? fw.off + (int64_t)nh_feat16(r, g) * H + f : -1;
                return g == 0 ? aw.off + f : -1;
            };
        }
    } else {
        NhTensor ow = T(p->t_out_w), ob = T(p->t_out_b);
        GemmSpec16& s = S.f_head;  // fc_out
        s.kr = KH;
        s.tiles = 1;
        s.w = [=](int o, int r, int g) -> int64_t {
            return (o < 4 && nh_feat16(r, g) < H) ? ow.off + (int64_t)o * H + nh_feat16(r, g) : -1;
        };
        s.b = [=](int o) -> int64_t { return o < 4 ? ob.off + o : -1; };
        GemmSpec16& bt = S.b_head;  // ONE k-step: group g carries d(out row g)
        bt.kr = 1;
        bt.tiles = TW;
        bt.w = [=](int f, int r, int g) -> int64_t { return (r == 0 && f < H) ? ow.off + (int64_t)g * H + f : -1; };
    }
}

template <class SpecsT, class Fn>
void for_each_spec(const nerfhip_plan* p, SpecsT& S, NhPackedOffsets& o, Fn fn) {
    fn(S.f_layer1, &o.f_layer1);
    for (int i = 0; i < p->L - 1; ++i) fn(S.f_xyz[i], &o.f_xyz[i]);
    fn(S.f_head, &o.f_head);
    if (p->view) {
        fn(S.f_dir, &o.f_dir);
        fn(S.f_rgb, &o.f_rgb);
        fn(S.b_rgb, &o.b_rgb);
        fn(S.b_dir, &o.b_dir);
    }
    fn(S.b_head, &o.b_head);
    for (int i = 0; i < p->L - 1; ++i) fn(S.b_xyz[i], &o.b_xyz[i]);
}

void layout_packed(nerfhip_plan* p) {
    int64_t off = 0;
    memset(&p->po, 0, sizeof(p->po));
    Specs16 S;
    build_specs16(p, S);
    for_each_spec(p, S, p->po, [&](const GemmSpec16& s, int64_t* dst) {
        *dst = off;
        off += nh16_image_floats(s.kr, s.tiles, p->W);
    });
    p->packed_floats = off;
}

// ---- fp16-piece images (nh_plan.h "fp16-piece images", mlp_f16w.hip / pack_f16.hip) -------------------------------------
struct GemmSpecB {
    int nk = 0, nt = 0;                            // k-blocks of 32 inputs, 16-row output tiles
    std::function<int64_t(int, int, int, int)> w;  // (out_row, kb, h, e) -> flat param index or -1
    std::function<int64_t(int)> b;                 // out_row -> flat param index or -1
};
struct SpecsB {
    GemmSpecB f_layer1, f_xyz[NH_MAX_LAYERS], f_head, f_dir, f_rgb;
    GemmSpecB b_rgb, b_dir, b_head, b_xyz[NH_MAX_LAYERS];  // transposed images of the data-gradient chain (no bias)
};

// encoding slot -> reference column (nerf/nerf_helpers.py:130-157: [x], then per frequency sin(3), cos(3))
bool build_slot_map_b(int L, int include_input, int nslots, int* col) {
    if (6 * L > nslots - 4) return false;
    for (int s = 0; s < nslots; ++s) col[s] = -1;
    const int base = include_input ? 3 : 0;
    for (int pr = 0; pr < 3 * L; ++pr) {
        col[2 * pr] = base + 6 * (pr / 3) + pr % 3;
        col[2 * pr + 1] = base + 6 * (pr / 3) + 3 + pr % 3;
    }
    if (include_input)
        for (int a = 0; a < 3; ++a) col[nslots - 4 + a] = a;
    return true;
}

void build_specs_b(const nerfhip_plan* p, SpecsB& S) {
    // (the geometry of mlp_f16w.hip: `h` is the lane group l >> 4 and a k-block is 32 inputs deep -- nh_plan.h)
    const int W = p->W, H = p->H, H2 = H / 2, KBH = W / 32, TH = W / 16, Dx = p->Dx, Dd = p->Dd, L = p->L;
    const int XBLOCKS = NHW_XBLOCKS, DBLOCKS = NHW_DBLOCKS;
    auto T = [p](int idx) { return p->tensors[idx]; };
    auto xcol = [p](int kb, int h, int e) { return p->xyz_slot_b[32 * kb + 8 * h + e]; };
    auto dcol = [p](int kb, int h, int e) { return p->dir_slot_b[32 * kb + 8 * h + e]; };
    auto nhb_unit = [](int kb, int h, int e) { return nhw_unit(kb, h, e); };
    {
        GemmSpecB& s = S.f_layer1;
        s.nk = XBLOCKS;
        s.nt = TH;
        NhTensor w = T(p->t_layer1_w), b = T(p->t_layer1_b);
        s.w = [=](int o, int kb, int h, int e) -> int64_t {
            const int c = xcol(kb, h, e);
            return (o < H && c >= 0) ? w.off + (int64_t)o * Dx + c : -1;
        };
        s.b = [=](int o) -> int64_t { return o < H ? b.off + o : -1; };
    }
    for (int i = 0; i < L - 1; ++i) {
        GemmSpecB& s = S.f_xyz[i];
        const bool sk = p->is_skip(i);
        s.nk = KBH + (sk ? XBLOCKS : 0);
        s.nt = TH;
        NhTensor w = T(p->t_xyz_w[i]), b = T(p->t_xyz_b[i]);
        const int ld = H + (sk ? Dx : 0);
        s.w = [=](int o, int kb, int h, int e) -> int64_t {
            if (o >= H) return -1;
            if (kb < KBH) return nhb_unit(kb, h, e) < H ? w.off + (int64_t)o * ld + nhb_unit(kb, h, e) : -1;
            const int c = xcol(kb - KBH, h, e);
            return c >= 0 ? w.off + (int64_t)o * ld + H + c : -1;
        };
        s.b = [=](int o) -> int64_t { return o < H ? b.off + o : -1; };
    }
    if (p->view) {
        NhTensor fw = T(p->t_feat_w), fb = T(p->t_feat_b), aw = T(p->t_alpha_w), ab = T(p->t_alpha_b);
        NhTensor dw = T(p->t_dir_w), db = T(p->t_dir_b), rw = T(p->t_rgb_w), rb = T(p->t_rgb_b);
        {
            GemmSpecB& s = S.f_head;  // rows 0..W-1 = fc_feat, row W = fc_alpha (models.py:248-249)
            s.nk = KBH;
            s.nt = TH + 1;
            s.w = [=](int o, int kb, int h, int e) -> int64_t {
                const int f = nhb_unit(kb, h, e);
                if (f >= H) return -1;
                if (o < H) return fw.off + (int64_t)o * H + f;
                if (o == W) return aw.off + f;
                return -1;
            };
            s.b = [=](int o) -> int64_t { return o < H ? fb.off + o : (o == W ? ab.off : -1); };
        }
        {
            GemmSpecB& s = S.f_dir;
            s.nk = KBH + DBLOCKS;
            s.nt = TH / 2;
            const int ld = H + Dd;
            s.w = [=](int o, int kb, int h, int e) -> int64_t {
                if (o >= H2) return -1;
                if (kb < KBH) return nhb_unit(kb, h, e) < H ? dw.off + (int64_t)o * ld + nhb_unit(kb, h, e) : -1;
                const int c = dcol(kb - KBH, h, e);
                return c >= 0 ? dw.off + (int64_t)o * ld + H + c : -1;
            };
            s.b = [=](int o) -> int64_t { return o < H2 ? db.off + o : -1; };
        }
        {
            GemmSpecB& s = S.f_rgb;
            s.nk = KBH / 2;
            s.nt = 1;
            s.w = [=](int o, int kb, int h, int e) -> int64_t {
                return (o < 3 && nhb_unit(kb, h, e) < H2) ? rw.off + (int64_t)o * H2 + nhb_unit(kb, h, e) : -1;
            };
            s.b = [=](int o) -> int64_t { return o < 3 ? rb.off + o : -1; };
        }
    } else {
        NhTensor ow = T(p->t_out_w), ob = T(p->t_out_b);
        GemmSpecB& s = S.f_head;  // fc_out
        s.nk = KBH;
        s.nt = 1;
        s.w = [=](int o, int kb, int h, int e) -> int64_t {
            return (o < 4 && nhb_unit(kb, h, e) < H) ? ow.off + (int64_t)o * H + nhb_unit(kb, h, e) : -1;
        };
        s.b = [=](int o) -> int64_t { return o < 4 ? ob.off + o : -1; };
    }
    // ---- the data-gradient chain (NERFHIP_PRECISION_F16X3_FWD_DGRAD and up): d(in)[f] = sum_u W[u][f] dpre[u]; k-block element
    // (kb, h, e) is unit u = nhb_unit(kb, h, e) of the layer's OUTPUT, output row f a unit of its input
    for (int i = 0; i < L - 1; ++i) {
        GemmSpecB& s = S.b_xyz[i];
        NhTensor w = T(p->t_xyz_w[i]);
        const int ld = H + (p->is_skip(i) ? Dx : 0);
        s.nk = KBH;
        s.nt = TH;
        s.w = [=](int f, int kb, int h, int e) -> int64_t {
            const int u = nhb_unit(kb, h, e);
            return (u < H && f < H) ? w.off + (int64_t)u * ld + f : -1;
        };
    }
    if (p->view) {
        NhTensor fw = T(p->t_feat_w), aw = T(p->t_alpha_w), dw = T(p->t_dir_w), rw = T(p->t_rgb_w);
        {
            GemmSpecB& s = S.b_rgb;  // ONE k-block: elements 0..2 of lane half 0 carry d(rgb raw)
            s.nk = 1;
            s.nt = TH / 2;
            s.w = [=](int f, int kb, int h, int e) -> int64_t { return (kb == 0 && h == 0 && e < 3 && f < H2) ? rw.off + (int64_t)e * H2 + f : -1; };
        }
        {
            GemmSpecB& s = S.b_dir;  // d(feat)[f] = sum_u Wdir[u][f] dpre_dir[u]
            s.nk = KBH / 2;
            s.nt = TH;
            const int ld = H + Dd;
            s.w = [=](int f, int kb, int h, int e) -> int64_t {
                const int u = nhb_unit(kb, h, e);
                return (u < H2 && f < H) ? dw.off + (int64_t)u * ld + f : -1;
            };
        }
        {
            // dh[f] = sum_u Wfeat[u][f] dpre_feat[u] + Walpha[0][f] d(sigma raw): the rank-one term rides in the image's BIAS row
            // (row f: Walpha[0][f]); the kernel starts its accumulators at bias * d(sigma raw) -- one fp32 multiply per unit
            // instead of a k-block whose single useful column would have to share the hidden inputs' fp16 exponent
            GemmSpecB& s = S.b_head;
            s.nk = KBH;
            s.nt = TH;
            s.w = [=](int f, int kb, int h, int e) -> int64_t {
                if (f >= H) return -1;
                return nhb_unit(kb, h, e) < H ? fw.off + (int64_t)nhb_unit(kb, h, e) * H + f : -1;
            };
            s.b = [=](int f) -> int64_t { return f < H ? aw.off + f : -1; };
        }
    } else {
        NhTensor ow = T(p->t_out_w);
        GemmSpecB& s = S.b_head;  // ONE k-block: elements 0..3 of lane half 0 carry d(out raw)
        s.nk = 1;
        s.nt = TH;
        s.w = [=](int f, int kb, int h, int e) -> int64_t { return (kb == 0 && h == 0 && e < 4 && f < H) ? ow.off + (int64_t)e * H + f : -1; };
    }
}

template <class Fn>
void for_each_spec_b(const nerfhip_plan* p, SpecsB& S, NhPackedOffsets& o, Fn fn) {
    fn(S.f_layer1, &o.f_layer1);
    for (int i = 0; i < p->L - 1; ++i) fn(S.f_xyz[i], &o.f_xyz[i]);
    fn(S.f_head, &o.f_head);
    if (p->view) {
        fn(S.f_dir, &o.f_dir);
        fn(S.f_rgb, &o.f_rgb);
    }
    if (nh_prec_level(p->precision) >= 3) {
        if (p->view) {
            fn(S.b_rgb, &o.b_rgb);
            fn(S.b_dir, &o.b_dir);
        }
        fn(S.b_head, &o.b_head);
        for (int i = 0; i < p->L - 1; ++i) fn(S.b_xyz[i], &o.b_xyz[i]);
    }
}

void fill_spec_b(const GemmSpecB& s, int64_t off, int32_t* table) {
    const int R = 16;  // rows of an output tile; a lane is (row l & (R - 1), group l / R)
    for (int i = 0; i < 512; ++i) table[off + i] = (i < R * s.nt && s.b) ? (int32_t)s.b(i) : -1;
    int32_t* img = table + off + 512;
    for (int kb = 0; kb < s.nk; ++kb)
        for (int t = 0; t < s.nt; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e)
                    img[(((int64_t)kb * s.nt + t) * 64 + lane) * 8 + e] = (int32_t)s.w(R * t + (lane & (R - 1)), kb, lane / R, e);
}

// the fp16-piece layer images, behind the first `base` words of the packed buffer
void layout_packed_b(nerfhip_plan* p, int64_t base) {
    int64_t off = base;
    memset(&p->pob, 0, sizeof(p->pob));
    SpecsB S;
    build_specs_b(p, S);
    for_each_spec_b(p, S, p->pob, [&](const GemmSpecB& s, int64_t* dst) {
        *dst = off;
        off += nhb_image_words(s.nk, s.nt);
    });
    p->packed32_floats = base;
    p->packed_floats = off;
}

// the resident image of the fused 64-wide backward (nh_r64.h): row-major matrices, k column = nh_feat16(r, g) of the input register,
// 16-byte chunks swizzled with the row (r64_pos)
void fill_r64(const nerfhip_plan* p, int32_t* table) {
    const R64Layout Y = r64_layout(p->L);
    int32_t* img = table + p->r64_off;
    for (int i = 0; i < Y.image_floats; ++i) img[i] = -1;
    const int H = p->H, H2 = H / 2, Dx = p->Dx, Dd = p->Dd;
    auto T = [p](int idx) { return p->tensors[idx]; };
    // slot register (r, g) of k column kc
    auto slot_r = [](int kc) { return 4 * (kc >> 4) + (kc & 3); };
    auto slot_g = [](int kc) { return (kc >> 2) & 3; };
    for (int o = 0; o < H; ++o) {
        for (int kc = 0; kc < 64; ++kc) {
            const int c = p->xyz_col16[slot_g(kc)][slot_r(kc)];
            if (c >= 0) img[Y.l1 + o * R64_S + r64_pos(o, kc)] = (int32_t)(T(p->t_layer1_w).off + (int64_t)o * Dx + c);
        }
        img[Y.b_l1 + o] = (int32_t)(T(p->t_layer1_b).off + o);
        for (int i = 0; i < p->L - 1; ++i) {
            for (int k = 0; k < H; ++k) img[Y.xyz[i] + o * R64_S + r64_pos(o, k)] = (int32_t)(T(p->t_xyz_w[i]).off + (int64_t)o * H + k);
            img[Y.b_xyz[i] + o] = (int32_t)(T(p->t_xyz_b[i]).off + o);
        }
        for (int k = 0; k < H; ++k) img[Y.head + o * R64_S + r64_pos(o, k)] = (int32_t)(T(p->t_feat_w).off + (int64_t)o * H + k);
        img[Y.b_feat + o] = (int32_t)(T(p->t_feat_b).off + o);
    }
    for (int k = 0; k < H; ++k) img[Y.head + 64 * R64_S + r64_pos(64, k)] = (int32_t)(T(p->t_alpha_w).off + k);
    img[Y.b_alpha] = (int32_t)T(p->t_alpha_b).off;
    for (int o = 0; o < H2; ++o) {
        for (int k = 0; k < H; ++k) img[Y.dir + o * R64_SD + r64_pos(o, k)] = (int32_t)(T(p->t_dir_w).off + (int64_t)o * (H + Dd) + k);
        for (int kc = 0; kc < 32; ++kc) {
            const int c = p->dir_col16[slot_g(kc)][slot_r(kc)];
            if (c >= 0) img[Y.dir + o * R64_SD + 64 + r64_pos(o, kc)] = (int32_t)(T(p->t_dir_w).off + (int64_t)o * (H + Dd) + H + c);
        }
        img[Y.b_dir + o] = (int32_t)(T(p->t_dir_b).off + o);
    }
    for (int o = 0; o < 3; ++o) {
        for (int k = 0; k < H2; ++k) img[Y.rgb + o * R64_SR + k] = (int32_t)(T(p->t_rgb_w).off + (int64_t)o * H2 + k);
        img[Y.b_rgb + o] = (int32_t)(T(p->t_rgb_b).off + o);
    }
}

NhRegion add_region(int64_t* total, int rows) {
    NhRegion r;
    r.rows = rows > 256 ? 256 : rows;  // a 512-row activation: two consecutive 256-row regions, named by the first
    r.row_prefix = *total;
    *total += rows;
    return r;
}

void add_job1(nerfhip_plan* p, const NhRegion& A, int a_tiles, const NhRegion& B, int b_row0, int b_tiles,
              int r_lo, int r_hi, int w_tensor, int col_kind, int col_base, int col_count, int bias_tensor) {
    NhJob j;
    memset(&j, 0, sizeof(j));
    j.a_region_rows = A.rows;
    j.a_row_prefix = A.row_prefix;
    j.a_tiles = a_tiles;
    j.b_region_rows = B.rows;
    j.b_row_prefix = B.row_prefix;
    j.b_row0 = b_row0;
    j.b_tiles = b_tiles;
    // wave grid: the 8 waves of a workgroup (two per SIMD) take equal patches of at most 8 accumulator tiles each
    int best_wo = 1, best_wi = 1, best_cost = 1 << 30;
    const int max_waves = p->wgrad_waves;
    for (int wo = 1; wo <= max_waves && wo <= a_tiles; wo *= 2)
        for (int wi = 1; wo * wi <= max_waves && wi <= b_tiles; wi *= 2) {
            int po = (a_tiles + wo - 1) / wo, pi = (b_tiles + wi - 1) / wi;
            // a wave's P tiles are P interleaved row sets of its block (wgrad.hip): the grid must tile the job exactly
            if (po > 4 || pi > 4 || po * pi > 8 || po == 3 || pi == 3 || po * wo != a_tiles || pi * wi != b_tiles) continue;
            // SIMD time per sample tile ~ (waves per SIMD) * patch
            const int per_simd = (wo * wi + 3) / 4;
            int cost = (per_simd * po * pi) * 64 + (max_waves - wo * wi) * 4 + (po + pi);  // ties: more waves (two per SIMD overlap), then fewer operand reads
            if (cost < best_cost) {
                best_cost = cost;
                best_wo = wo;
                best_wi = wi;
            }
        }
    j.wo = best_wo;
    j.wi = best_wi;
    j.po = (a_tiles + j.wo - 1) / j.wo;
    j.pi = (b_tiles + j.wi - 1) / j.wi;
    // Relative time one workgroup spends per sample tile (split-K allocation), fitted to per-workgroup timestamps on
    // MI355X for the 8x256, 4x128 and 8x128 nets (scripts/wgrad_timeline.py, profiles/r02_wgrad_timeline.txt): per tile
    // t = 0.30 us * (MFMAs per k-step of the busiest SIMD) + 0.20 us * (operand dwords its waves read per k-step) + 0.45 us
    // (stage hand-over), within 4 % for every job shape that occurs.
    {
        const int per_simd = (j.wo * j.wi + 3) / 4;
        j.cost = 30 * per_simd * j.po * j.pi + 20 * per_simd * (j.po + j.pi) + 45;
        // residuals of that fit in the final timeline (profiles/r02_wgrad_timeline.txt): one-tile patches run 4-7 % longer
        // than modelled, the 2 x 1 patches of the encoding-column jobs 3 % shorter, the 2 x 2 patches 2 % longer
        if (j.po * j.pi == 1) j.cost += 7;
#ifndef NH_COST_21_ADJ  // (A/B builds only)
#define NH_COST_21_ADJ (-9)
#endif
        if (j.po == 2 && j.pi == 1) j.cost += NH_COST_21_ADJ;
        if (j.po == 2 && j.pi == 2 && per_simd == 2) j.cost += 8;
    }
    j.r_lo = r_lo;
    j.r_hi = r_hi;
    j.w_off = p->tensors[w_tensor].off;
    j.w_ld = p->tensors[w_tensor].cols;
    j.col_kind = col_kind;
    j.col_base = col_base;
    j.col_count = col_count;
    j.bias_off = bias_tensor >= 0 ? p->tensors[bias_tensor].off : -1;
    p->jobs.push_back(j);
}

// One weight block = one job -- or, for 512-row activations (512-wide nets: two consecutive 256-row regions each), one
// job per pair of halves.  `a_rows` / `b_rows`: rows of the whole activation (A.rows / B.rows name its first region).  Row
// and column bounds of the job are the block's, shifted into the half's local indices; the bias gradient (row sums of A)
// rides with the first B half only.
void add_job(nerfhip_plan* p, const NhRegion& A, int a_tiles, const NhRegion& B, int b_row0, int b_tiles, int r_lo,
             int r_hi, int w_tensor, int col_kind, int col_base, int col_count, int bias_tensor) {
    for (int a0 = 0; a0 < a_tiles; a0 += 8)
        for (int b0 = 0; b0 < b_tiles; b0 += 8) {
            NhRegion Ah = A, Bh = B;
            Ah.row_prefix += 32 * a0;  // (a0, b0 > 0 only for the second 256-row region of a 512-row activation)
            Bh.row_prefix += 32 * b0;
            const int ar = 32 * a0, bc = 32 * b0;
            add_job1(p, Ah, a_tiles - a0 < 8 ? a_tiles - a0 : 8, Bh, b_row0, b_tiles - b0 < 8 ? b_tiles - b0 : 8, r_lo - ar,
                     r_hi - ar, w_tensor, col_kind, col_base + (col_kind == 0 ? bc : 0), col_kind == 0 ? col_count - bc : col_count,
                     b0 == 0 ? bias_tensor : -1);
        }
}

// Thin weight blocks become side tiles of the job that streams the same region anyway (wgrad.hip): a block with ONE A
// tile (fc_alpha: POUT x H_{L-1}) rides on the job with the same B region (fc_feat) when every B tile of a column's patch
// finds a wave (pi <= wo); a block with one or two B tiles (the direction columns: PDIR x D; a skip layer's encoding
// columns: P_{i+1} x X) rides on the job with the same A region when its po * tiles pairs per wave row fit two per wave,
// and when the stage still holds one sample tile of all three regions.  The guest's job disappears from the list.
void attach_sides(nerfhip_plan* p, int stage_floats) {
    if (p->W > 256) return;  // (512-wide nets: the hosts are half-region jobs; their stages are full)
    std::vector<NhJob>& J = p->jobs;
    for (size_t gi = 0; gi < J.size(); ++gi) {
        const NhJob g = J[gi];
        if (g.side_kind) continue;
        int host = -1, kind = 0;
        for (size_t hi = 0; hi < J.size() && host < 0; ++hi) {
            const NhJob& h = J[hi];
            if (hi == gi || h.side_kind || h.a_tiles * h.b_tiles < 8) continue;
            if (32 * (h.a_region_rows + h.b_region_rows + 32 * (g.a_tiles == 1 ? 1 : g.b_tiles)) > stage_floats) continue;
            if (g.a_tiles == 1 && g.a_region_rows == 32 && g.r_hi - g.r_lo == 1 && h.b_row_prefix == g.b_row_prefix && h.b_tiles == g.b_tiles &&
                h.a_region_rows == h.b_region_rows && h.a_region_rows == (p->wgrad_waves == 8 ? 256 : 128) &&
                ((p->wgrad_waves == 8 && h.po == 4 && h.pi == 2) || (p->wgrad_waves == 4 && h.po == 2 && h.pi == 2))) {
                host = (int)hi, kind = 1;
            } else if (g.b_tiles <= 2 && g.b_region_rows == 32 * g.b_tiles && g.a_tiles > 1 && h.a_row_prefix == g.a_row_prefix &&
                       h.a_tiles == g.a_tiles) {
#ifdef NH_WGRAD_SIDE2  // (A/B builds only; see wgrad.hip)
                const bool two = g.b_tiles == 2 && p->wgrad_waves == 8 && h.po == 4 && h.pi == 2 && h.wi == 4;
#else
                const bool two = false;
#endif
                const bool one = g.b_tiles == 1 && h.po <= h.wi &&
                                 ((p->wgrad_waves == 8 && h.po == 2 && h.pi == 2) ||
                                  (p->wgrad_waves == 4 && ((h.po == 1 && h.pi == 2) || (h.po == 2 && h.pi == 1))));
                if (two || one) host = (int)hi, kind = 2;
            }
        }
        if (host < 0) continue;
        NhJob& h = J[host];
        h.side_kind = kind;
        h.side_rows = kind == 1 ? g.a_region_rows : g.b_region_rows;
        h.side_tiles = kind == 1 ? 1 : g.b_tiles;
        h.side_row_prefix = kind == 1 ? g.a_row_prefix : g.b_row_prefix;
        h.s_r_lo = g.r_lo;
        h.s_r_hi = g.r_hi;
        h.s_w_off = g.w_off;
        h.s_w_ld = g.w_ld;
        h.s_col_kind = g.col_kind;
        h.s_col_base = g.col_base;
        h.s_col_count = g.col_count;
        h.s_bias_off = g.bias_off;
        // its MFMAs now run inside the host's k-steps: one or two more per wave, one more operand
        const int per_simd = (h.wo * h.wi + 3) / 4;
        // measured on MI355X by A/B of the split-K allocation (profiles/r03_variant_ab.txt): what a B-side tile adds to its
        // host's time per sample tile, in the cost model's units -- 4-wave mode 140 (k_wgrad<128> 0.589 -> 0.611 of peak; with
        // 50 the side hosts finished last and the kernel was SLOWER than without sides), 8-wave mode 80 (0.840 -> 0.846).
        // The A-side row is a few VALU instructions per k-step of one wave per column.
#ifndef NH_SIDE_COST_A  // (A/B builds only)
#define NH_SIDE_COST_A 10
#define NH_SIDE_COST_B (p->wgrad_waves == 4 ? 140 : 80)
#endif
        h.cost += per_simd * (kind == 1 ? NH_SIDE_COST_A : NH_SIDE_COST_B) * (kind == 2 && g.b_tiles == 2 ? 2 : 1);
        J.erase(J.begin() + gi);
        --gi;
    }
}

void build_layouts_and_jobs(nerfhip_plan* p) {
    const int W = p->W, L = p->L;
    NhStashLayout& S = p->stash;
    memset(&S, 0, sizeof(S));
    S.total_rows = 0;
    S.X = add_region(&S.total_rows, 4 * p->krx);  // slot rows g*KR + r
    if (p->view) S.D = add_region(&S.total_rows, 4 * p->krd);
    for (int k = 0; k < L; ++k) S.H[k] = add_region(&S.total_rows, W);
    if (p->view) {
        S.FEAT = add_region(&S.total_rows, W);
        S.DIRH = add_region(&S.total_rows, W / 2);
    }
    S.n_masks = p->view ? L + 1 : (L > 1 ? L - 1 : 1);
    NhGradLayout& G = p->grad;
    memset(&G, 0, sizeof(G));
    G.total_rows = 0;
    for (int k = 0; k < L; ++k) G.P[k] = add_region(&G.total_rows, W);
    if (p->view) {
        G.PFEAT = add_region(&G.total_rows, W);
        G.PDIR = add_region(&G.total_rows, W / 2);
    }
    G.POUT = add_region(&G.total_rows, 32);

    p->jobs.clear();
#ifdef NH_WGRAD_WIDE_ONLY  // (A/B builds only: scripts/build_variant.sh)
    p->wgrad_waves = 8;
#else
    p->wgrad_waves = W >= 256 ? 8 : 4;
#endif
    // (tiles cover the kernel width W; only the rows / columns of the real H hidden units are unpacked)
    const int TW = W / 32, H = p->H, H2 = H / 2;
    // F16X3_TRAIN: the hidden x hidden blocks go to the fp16-piece weight-gradient kernel instead (wgrad_f16.hip)
    p->bjobs.clear();
    // (128- and 256-wide nets; 128: the four full blocks only -- the half-height block of layers_dir stays a thin job.  Round 3's
    // kernel lost on the 128 x 128 blocks (4x128 step 4.23 -> 5.04 ms); round 4's, which waits for HBM and nothing else, wins:
    // 4.35 -> 4.04 ms, profiles/r04_wgrad_128_ab.txt.  Other widths: _TRAIN is _FWD_DGRAD)
    const bool big_b = nh_prec_level(p->precision) == 4 && (W == 128 || W == 256);
    // ... and a thin block whose operands such a block streams anyway rides on it as a guest (wgrad_f16.hip SA / SB) instead of
    // reading them a second time as an fp32 job of its own: a skip layer's xyz columns, fc_alpha's row, layers_dir's direction columns.
    // (the guest regions need recorded maxima, which the kernels of mlp_f16w.hip write)
#ifdef NHW_NO_GUESTS  // (A/B builds only)
    const bool guests = false;
#else
    const bool guests = big_b;
#endif
    auto add_big = [&](const NhRegion& A, int a_rows, const NhRegion& B, int r_hi, int w_tensor, int bias_tensor, int a_idx, int b_idx) {
        NhJobB j;
        memset(&j, 0, sizeof(j));
        j.s_bias_off = -1;
        j.a_idx = a_idx;
        j.b_idx = b_idx;
        j.a_rows = a_rows;
        j.b_rows = W;
        j.a_row_prefix = A.row_prefix;
        j.b_row_prefix = B.row_prefix;
        j.r_hi = r_hi;
        j.col_count = H;
        j.w_ld = p->tensors[w_tensor].cols;
        j.w_off = p->tensors[w_tensor].off;
        j.bias_off = p->tensors[bias_tensor].off;
        p->bjobs.push_back(j);
    };
    // layer1: dP_0 x X
    add_job(p, G.P[0], TW, S.X, 0, p->krx / 8, 0, H, p->t_layer1_w, 1, 0, p->Dx, p->t_layer1_b);
    for (int i = 0; i < L - 1; ++i) {
        if (big_b)
            add_big(G.P[i + 1], W, S.H[i], H, p->t_xyz_w[i], p->t_xyz_b[i], i + 1, i);
        else
            add_job(p, G.P[i + 1], TW, S.H[i], 0, TW, 0, H, p->t_xyz_w[i], 0, 0, H, p->t_xyz_b[i]);
        if (p->is_skip(i)) {
            if (guests && S.X.rows == 64) {  // the xyz columns: P_{i+1} x X next to P_{i+1} x H_i
                NhJobB& j = p->bjobs.back();
                j.side_kind = 2, j.side_rows = S.X.rows, j.side_idx = nh_rmax_x(L), j.side_row_prefix = S.X.row_prefix;
                j.s_w_off = p->tensors[p->t_xyz_w[i]].off, j.s_w_ld = p->tensors[p->t_xyz_w[i]].cols;
                j.s_col_kind = 1, j.s_col_base = H, j.s_col_count = p->Dx;
            } else {
                add_job(p, G.P[i + 1], TW, S.X, 0, p->krx / 8, 0, H, p->t_xyz_w[i], 1, H, p->Dx, -1);
            }
        }
    }
    if (p->view) {
        if (big_b)
            add_big(G.PFEAT, W, S.H[L - 1], H, p->t_feat_w, p->t_feat_b, L, L - 1);
        else
            add_job(p, G.PFEAT, TW, S.H[L - 1], 0, TW, 0, H, p->t_feat_w, 0, 0, H, p->t_feat_b);
        if (guests) {  // fc_alpha: row 3 of POUT x H_{L-1} next to PFEAT x H_{L-1}
            NhJobB& j = p->bjobs.back();
            j.side_kind = 1, j.side_rows = 32, j.side_idx = nh_rmax_pout(L), j.side_row_prefix = G.POUT.row_prefix;
            j.s_r_lo = 3, j.s_r_hi = 4;
            j.s_w_off = p->tensors[p->t_alpha_w].off, j.s_w_ld = p->tensors[p->t_alpha_w].cols;
            j.s_bias_off = p->tensors[p->t_alpha_b].off;
        } else {
            add_job(p, G.POUT, 1, S.H[L - 1], 0, TW, 3, 4, p->t_alpha_w, 0, 0, H, p->t_alpha_b);
        }
        if (big_b && W >= 256)
            add_big(G.PDIR, W / 2, S.FEAT, H2, p->t_dir_w, p->t_dir_b, L + 1, L);
        else
            add_job(p, G.PDIR, TW / 2, S.FEAT, 0, TW, 0, H2, p->t_dir_w, 0, 0, H, p->t_dir_b);
        if (guests && W >= 256 && S.D.rows == 32 && p->Dd > 0) {  // the direction columns: PDIR x D next to PDIR x FEAT
            NhJobB& j = p->bjobs.back();
            j.side_kind = 2, j.side_rows = S.D.rows, j.side_idx = nh_rmax_d(L), j.side_row_prefix = S.D.row_prefix;
            j.s_w_off = p->tensors[p->t_dir_w].off, j.s_w_ld = p->tensors[p->t_dir_w].cols;
            j.s_col_kind = 2, j.s_col_base = H, j.s_col_count = p->Dd;
        } else {
            add_job(p, G.PDIR, TW / 2, S.D, 0, p->krd / 8, 0, H2, p->t_dir_w, 2, H, p->Dd, -1);
        }
        add_job(p, G.POUT, 1, S.DIRH, 0, TW / 2, 0, 3, p->t_rgb_w, 0, 0, H2, p->t_rgb_b);
    } else {
        add_job(p, G.POUT, 1, S.H[L - 1], 0, TW, 0, 4, p->t_out_w, 0, 0, H, p->t_out_b);
    }
#ifndef NH_WGRAD_NO_SIDES  // (A/B builds only)
    attach_sides(p, p->wgrad_waves == 4 ? 9216 : 18432);  // (the stage sizes of wgrad.hip's two modes)
#endif
}

}  // namespace

static nerfhip_plan_t plan_create_impl(const nerfhip_model_cfg* cfg, int precision);
extern "C" nerfhip_plan_t nerfhip_plan_create(const nerfhip_model_cfg* cfg) { return plan_create_impl(cfg, NERFHIP_PRECISION_FP32); }
extern "C" nerfhip_plan_t nerfhip_plan_create_ex(const nerfhip_model_cfg* cfg, int precision) {
    if (precision < NERFHIP_PRECISION_FP32 || precision > NERFHIP_PRECISION_F16X3_TRAIN) {
        nh_set_error("plan_create_ex: unknown precision %d", precision);
        return nullptr;
    }
    if (precision > NERFHIP_PRECISION_FP32 && precision < NERFHIP_PRECISION_F16X3) {
        nh_set_error("plan_create_ex: precision %d named a bf16-piece plan (round 3's experiment); those kernels were removed in round 5 -- "
                     "the fp16-piece plans NERFHIP_PRECISION_F16X3* are as fast and hold the fp32 parity bounds", precision);
        return nullptr;
    }
    return plan_create_impl(cfg, precision);
}
extern "C" int nerfhip_plan_precision(nerfhip_plan_t plan) { return plan ? plan->precision : -1; }

static nerfhip_plan_t plan_create_impl(const nerfhip_model_cfg* cfg, int precision) {
    if (!cfg) {
        nh_set_error("plan_create: cfg is NULL");
        return nullptr;
    }
    if (cfg->hidden_size < 2 || cfg->hidden_size > 512) {
        nh_set_error("plan_create: hidden_size must be in [2,512] (got %d)", cfg->hidden_size);
        return nullptr;
    }
    if (cfg->num_layers < 1 || cfg->num_layers > NH_MAX_LAYERS) {
        nh_set_error("plan_create: num_layers must be in [1,%d] (got %d)", NH_MAX_LAYERS, cfg->num_layers);
        return nullptr;
    }
    if (cfg->skip_connect_every < 1) {
        nh_set_error("plan_create: skip_connect_every must be >= 1");
        return nullptr;
    }
    if (cfg->num_encoding_fn_xyz < 0 || cfg->num_encoding_fn_xyz > 16 || cfg->num_encoding_fn_dir < 0 ||
        cfg->num_encoding_fn_dir > 10) {
        nh_set_error("plan_create: num_encoding_fn_xyz must be <= 16 and num_encoding_fn_dir <= 10 (got %d, %d)",
                     cfg->num_encoding_fn_xyz, cfg->num_encoding_fn_dir);
        return nullptr;
    }
    nerfhip_plan* p = new nerfhip_plan();
    p->cfg = *cfg;
    p->precision = precision;
    p->H = cfg->hidden_size;
    // the kernels exist for four widths; a model rides zero-padded on the next one (build_specs16)
    p->W = p->H <= 64 ? 64 : (p->H <= 128 ? 128 : (p->H <= 256 ? 256 : 512));
    p->L = cfg->num_layers;
    p->skip = cfg->skip_connect_every;
    p->view = cfg->use_viewdirs ? 1 : 0;
    p->Dx = (cfg->include_input_xyz ? 3 : 0) + 6 * cfg->num_encoding_fn_xyz;
    p->Dd = p->view ? (cfg->include_input_dir ? 3 : 0) + 6 * cfg->num_encoding_fn_dir : 0;
    if (p->Dx == 0) {
        nh_set_error("plan_create: empty xyz encoding");
        delete p;
        return nullptr;
    }
    p->nparams = 0;
    p->freqs_set = false;
    const int W = p->H, L = p->L;  // (tensor shapes: the model's own hidden_size)
    // registration order of nerf/models.py:205-229
    p->t_layer1_w = add_tensor(p, "layer1.weight", W, p->Dx);
    p->t_layer1_b = add_tensor(p, "layer1.bias", W, 0);
    for (int i = 0; i < L - 1; ++i) {
        std::string n = "layers_xyz." + std::to_string(i);
        p->t_xyz_w[i] = add_tensor(p, n + ".weight", W, W + (p->is_skip(i) ? p->Dx : 0));
        p->t_xyz_b[i] = add_tensor(p, n + ".bias", W, 0);
    }
    p->t_dir_w = p->t_dir_b = p->t_alpha_w = p->t_alpha_b = p->t_rgb_w = p->t_rgb_b = p->t_feat_w = p->t_feat_b = -1;
    p->t_out_w = p->t_out_b = -1;
    if (p->view) {
        p->t_dir_w = add_tensor(p, "layers_dir.0.weight", W / 2, W + p->Dd);
        p->t_dir_b = add_tensor(p, "layers_dir.0.bias", W / 2, 0);
        p->t_alpha_w = add_tensor(p, "fc_alpha.weight", 1, W);
        p->t_alpha_b = add_tensor(p, "fc_alpha.bias", 1, 0);
        p->t_rgb_w = add_tensor(p, "fc_rgb.weight", 3, W / 2);
        p->t_rgb_b = add_tensor(p, "fc_rgb.bias", 3, 0);
        p->t_feat_w = add_tensor(p, "fc_feat.weight", W, W);
        p->t_feat_b = add_tensor(p, "fc_feat.bias", W, 0);
    } else {
        p->t_out_w = add_tensor(p, "fc_out.weight", 4, W);
        p->t_out_b = add_tensor(p, "fc_out.bias", 4, 0);
    }
    {
        // the default slot registers hold the reference's own settings; anything longer takes the extended instantiation
        // of the forward kernel (both encodings together: one extra template instance, not two)
        const bool ext = cfg->num_encoding_fn_xyz > 10 || (p->view && cfg->num_encoding_fn_dir > 4);
        p->krx = ext ? NH16_KRX_EXT : NH16_KRX;
        p->krd = ext ? NH16_KRD_EXT : NH16_KRD;
        const bool okx = build_slot_map16(cfg->num_encoding_fn_xyz, cfg->include_input_xyz ? 1 : 0, p->krx, &p->xyz_col16[0][0],
                                          NH16_KRX_EXT);
        const bool okd = build_slot_map16(p->view ? cfg->num_encoding_fn_dir : 0, (p->view && cfg->include_input_dir) ? 1 : 0,
                                          p->krd, &p->dir_col16[0][0], NH16_KRD_EXT);
        if (!(okx && okd)) {
            nh_set_error("plan_create: encoding does not fit the slot registers");
            delete p;
            return nullptr;
        }
        for (int row = 0; row < 4 * NH16_KRX_EXT; ++row)
            p->xyz_slot_col[row] = row < 4 * p->krx ? p->xyz_col16[row / p->krx][row % p->krx] : -1;
        for (int row = 0; row < 4 * NH16_KRD_EXT; ++row)
            p->dir_slot_col[row] = row < 4 * p->krd ? p->dir_col16[row / p->krd][row % p->krd] : -1;
    }
    for (int k = 0; k < 16; ++k) {
        p->freqs_xyz[k] = 0.f;
        p->freqs_dir[k] = 0.f;
    }
    if (precision != NERFHIP_PRECISION_FP32) {
        // the fp16-piece kernels exist for the 128- and 256-wide nets with the reference's own encoding sizes; everything else
        // is the fp32 path's business
        const bool okx = build_slot_map_b(cfg->num_encoding_fn_xyz, cfg->include_input_xyz ? 1 : 0, NHW_XSLOTS, p->xyz_slot_b);
        const bool okd = build_slot_map_b(p->view ? cfg->num_encoding_fn_dir : 0, (p->view && cfg->include_input_dir) ? 1 : 0,
                                          NHW_DSLOTS, p->dir_slot_b);
        if (!(okx && okd) || (p->W != 64 && p->W != 128 && p->W != 256)) {
            nh_set_error("plan_create_ex: f16x3 plans need hidden_size <= 256, num_encoding_fn_xyz <= 10 and "
                         "num_encoding_fn_dir <= 4 (got %d, %d, %d)", cfg->hidden_size, cfg->num_encoding_fn_xyz, cfg->num_encoding_fn_dir);
            delete p;
            return nullptr;
        }
        memset(&p->po, 0, sizeof(p->po));
        p->packed_floats = 0;
        if (nh_prec_level(precision) != 1) {
            layout_packed(p);  // the fp32 image: its transposed layers feed the data-gradient kernel
            // the training forward stores the encodings in ITS slot order: that is what the weight-gradient scatter must undo
            for (int row = 0; row < 4 * NH16_KRX_EXT; ++row) p->xyz_slot_col[row] = row < NHW_XSLOTS ? p->xyz_slot_b[row] : -1;
            for (int row = 0; row < 4 * NH16_KRD_EXT; ++row) p->dir_slot_col[row] = row < NHW_DSLOTS ? p->dir_slot_b[row] : -1;
        }
        layout_packed_b(p, p->packed_floats);
    } else {
        layout_packed(p);
        if (nh_r64_eligible(p)) {  // the LDS-resident image of the fused backward (mlp64r.hip), behind the layer images
            p->r64_off = p->packed_floats;
            p->packed_floats += r64_layout(p->L).image_floats;
        }
        p->packed32_floats = p->packed_floats;
        memset(&p->pob, 0, sizeof(p->pob));
    }
    build_layouts_and_jobs(p);
    // (the weight-gradient reduce kernel's table holds one record per job and one per attached side block)
    int reduce_records = 0;
    for (const NhJob& j : p->jobs) reduce_records += j.side_kind ? 2 : 1;
    if ((int)p->jobs.size() > NH_MAX_JOBS || reduce_records > NH_MAX_JOBS) {
        nh_set_error("plan_create: too many gradient jobs (%d jobs, %d reduce records; limit %d)", (int)p->jobs.size(), reduce_records, NH_MAX_JOBS);
        delete p;
        return nullptr;
    }
    // (the fp16-piece weight-gradient kernel's launch tables: refused here, with wgrad_f16.hip's message, not at the first backward)
    if (nh_wgrad_x3_partial_floats(p, 4) < 0) {
        delete p;
        return nullptr;
    }
    return p;
}

extern "C" void nerfhip_plan_destroy(nerfhip_plan_t plan) { delete plan; }
extern "C" int64_t nerfhip_plan_num_params(nerfhip_plan_t plan) { return plan ? plan->nparams : -1; }
extern "C" int nerfhip_plan_dim_xyz(nerfhip_plan_t plan) { return plan ? plan->Dx : -1; }
extern "C" int nerfhip_plan_dim_dir(nerfhip_plan_t plan) { return plan ? plan->Dd : -1; }
extern "C" int nerfhip_plan_num_tensors(nerfhip_plan_t plan) { return plan ? (int)plan->tensors.size() : -1; }
extern "C" int nerfhip_plan_tensor_info(nerfhip_plan_t plan, int i, const char** name, int64_t* offset, int* rows,
                                        int* cols) {
    NH_REQUIRE(plan && i >= 0 && i < (int)plan->tensors.size(), "plan_tensor_info: bad index");
    const NhTensor& t = plan->tensors[i];
    if (name) *name = t.name.c_str();
    if (offset) *offset = t.off;
    if (rows) *rows = t.rows;
    if (cols) *cols = t.cols;
    return NERFHIP_OK;
}
extern "C" int64_t nerfhip_plan_packed_floats(nerfhip_plan_t plan) { return plan ? plan->packed_floats : -1; }

extern "C" int nerfhip_plan_describe(nerfhip_plan_t plan, char* buf, int64_t cap) {
    NH_REQUIRE(plan && buf && cap > 0, "plan_describe: bad arguments");
    int64_t used = 0;
    auto put = [&](const char* fmt, auto... v) {
        if (used >= cap - 1) return;
        const int w = snprintf(buf + used, (size_t)(cap - used), fmt, v...);
        if (w > 0) used += w < cap - used ? w : cap - used - 1;
    };
    put("kernel_width %d hidden_size %d layers %d params %lld packed_floats %lld wgrad_waves %d jobs %d precision %d two_wave_images %d\n",
        plan->W, plan->H, plan->L, (long long)plan->nparams, (long long)plan->packed_floats, plan->wgrad_waves, (int)plan->jobs.size(),
        plan->precision, plan->precision != NERFHIP_PRECISION_FP32 ? 1 : 0);
    for (size_t q = 0; q < plan->jobs.size(); ++q) {
        const NhJob& j = plan->jobs[q];
        put("job %d tiles %dx%d waves %dx%d patch %dx%d cost %d side %d side_tiles %d\n", (int)q, j.a_tiles, j.b_tiles, j.wo, j.wi, j.po,
            j.pi, j.cost, j.side_kind, j.side_kind ? j.side_tiles : 0);
    }
    return NERFHIP_OK;
}

extern "C" int nerfhip_plan_pack_index(nerfhip_plan_t plan, int32_t* host_table) {
    NH_REQUIRE(plan && host_table, "plan_pack_index: bad arguments");
    if (plan->precision != NERFHIP_PRECISION_FP32) {
        SpecsB S;
        build_specs_b(plan, S);
        NhPackedOffsets o = plan->pob;
        for_each_spec_b(plan, S, o, [&](const GemmSpecB& s, int64_t* dst) { fill_spec_b(s, *dst, host_table); });
        if (nh_prec_level(plan->precision) == 1) return NERFHIP_OK;
    }
    Specs16 S16;
    build_specs16(plan, S16);
    NhPackedOffsets o = plan->po;
    for_each_spec(plan, S16, o, [&](const GemmSpec16& s, int64_t* dst) { fill_spec16(s, *dst, host_table, plan->W); });
    if (plan->r64_off >= 0) fill_r64(plan, host_table);
    return NERFHIP_OK;
}

extern "C" int nerfhip_plan_set_freqs(nerfhip_plan_t plan, const float* freqs_xyz, const float* freqs_dir) {
    NH_REQUIRE(plan && freqs_xyz, "plan_set_freqs: bad arguments");
    for (int k = 0; k < 16; ++k) {
        plan->freqs_xyz[k] = k < plan->cfg.num_encoding_fn_xyz ? freqs_xyz[k] : 0.f;
        plan->freqs_dir[k] = (freqs_dir && plan->view && k < plan->cfg.num_encoding_fn_dir) ? freqs_dir[k] : 0.f;
    }
    plan->freqs_set = true;
    return NERFHIP_OK;
}

extern "C" int64_t nerfhip_plan_stash_bytes(nerfhip_plan_t plan, int64_t m) {
    if (!plan || m < 0) return -1;
    int64_t tiles = nh_ceil_div(m, 128) * 4;
    // per 32-sample tile: the row regions, then the ReLU masks of its two 16-sample wave tiles; behind them the region maxima
    return (nh_stash_floats(plan, tiles) + NH_RMAX_WORDS) * (int64_t)sizeof(float);
}
