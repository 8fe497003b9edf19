// wgrad.hip -- weight gradients of FlexibleNeRFModel (what autograd computes for the Linear layers of
// nerf/models.py:233-256): dW[out, in] = sum over samples of dP[out][sample] * act[in][sample], bias gradient = row sums
// of dP, as a split-K fp32-MFMA kernel (v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fmaf chain) over the
// sample-major images [tile][32 samples][rows] the forward (activation stash) and data-gradient (dP scratch) kernels
// of mlp16.hip write; a second kernel reduces the split-K partials in a fixed order (bit-reproducible: no atomics) and
// scatters them into the reference parameter layout.
//
// Structure (measured on MI355X: scripts/loop_mock.hip, profiles/r02_loop_mock.txt):
//  * a workgroup is 8 waves = TWO per SIMD, each owning a patch of PO x PI accumulator tiles (<= 8 tiles = 128 registers);
//  * the operands of a run of sample tiles are two CONTIGUOUS blocks of HBM ([32 G samples][a_rows] of the gradient
//    scratch, [32 G samples][b_rows] of the stash: every job covers whole regions), copied once per workgroup into LDS
//    by LDS-DMA (1 KiB per instruction, no VGPRs), double buffered, one barrier per stage;
//  * WIDE operand reads: a lane fetches the operands of ALL its tiles of one k-step with ONE LDS instruction.  Tile x of
//    a wave's block of 32*P rows is defined as rows {P*i + x : i < 32}, so lane (i, k) reads the P consecutive floats
//    lds[sample 2e+k][32*P*block + P*i ..] -- ds_read_b128 for P = 4, ds_read_b64 for P = 2 -- instead of P separate
//    ds_read_b32 (the mock: 96.4 % -> 99.0 % of the matrix pipe for the 4 x 2 patch).  Which 32 rows form a tile only
//    matters to the reduce kernel, which undoes the interleave when it scatters;
//  * SIDE tiles (round 3): a weight block with one or two tiles on one side -- fc_alpha (1 x TW tiles), the direction
//    columns of layers_dir (TW/2 x 1), the encoding columns of a skip layer (TW x 2) -- used to be a job of its own that
//    paid the full per-stage hand-over for one or two MFMAs per k-step (the cost model, plan.cpp: 22 % of this kernel's
//    time for 4 % of its work on the 4x128 nets).  Such a block now rides on the job that streams the same A (or B)
//    region anyway: the side region's rows are a third block of the stage.  Extra B tiles (the direction columns) become
//    one more accumulator tile per wave, fed by an A operand the wave already holds -- picked with v_cndmask on its grid
//    position -- and the side operand.  An extra A tile with ONE useful row (fc_alpha: row 3 of the 32-row d(raw output)
//    region; an MFMA tile would be 31/32 waste) is not multiplied on the matrix pipe at all: the row's value of the
//    lane's sample is a broadcast LDS read, and the wave of row 0 of each column adds value * B operand to PI scalars per
//    k-step in the shadow of its MFMAs; the sums land where the reduce kernel expects that row of an accumulator tile;
//  * bias sums (VALU adds next to the MFMAs; the mock: -3.6 % when two of the eight waves carry all of them, because
//    every stage ends in a barrier) are spread over the waves that share an A block: wave (ow, iw) sums tile x = iw only.
#include <stdlib.h>

#include "nh_mlp.h"

namespace {

// Kernel arguments are limited to 4 KB, and NH_MAX_JOBS = 64 records must fit: the two kernels get a table each, with
// only the fields they read (plain ints: packing them into shorts cost k_wgrad SGPR spills).
struct JobDev {  // k_wgrad
    int a_rows, a_prefix, a_tiles;
    int b_rows, b_prefix, b_tiles;
    int wo, wi, po, pi;
    int wg_start;
    int g;  // 32-sample tiles per LDS stage
    int side;         // kind (0 none, 1 one extra A tile, 2 extra B tiles) | tiles << 8 | rows of the side region << 16
    int side_prefix;  // its row prefix (A-side: in the gradient scratch, B-side: in the stash)
};
struct JobRed {  // k_wgrad_reduce
    int a_tiles, b_tiles, po, pi;
    int r_lo, r_hi;
    int w_off, w_ld;
    int col_kind, col_base, col_count;
    int bias_off;
    int wg_start;  // (a side block follows its host job in this table with the SAME wg_start: it shares those workgroups)
    int ks_log;    // log2 of the K-slices per element in k_wgrad_reduce: 0, 2 or 4 (fewer accumulator tiles -> more slices)
                   // | (first bias tile inside a workgroup's bias partials) << 8
    int tile0;     // first accumulator tile of this block inside a workgroup's partial (0, or behind the host job's tiles)
};
constexpr int NH_JOBS_DEV = NH_MAX_JOBS;
static_assert(sizeof(JobDev) * NH_JOBS_DEV + 80 <= 4096 && sizeof(JobRed) * NH_JOBS_DEV + 248 <= 4096,
              "the job tables must fit the 4 KB kernel-argument limit");
// Workgroup shapes.  256-wide nets: 8 waves per workgroup (two per SIMD), two LDS stages of 16384 floats (+ slack for
// the operand prefetch that runs one k-step past the end of a stage): one workgroup per CU.  128-wide nets (jobs of at
// most 4 x 4 tiles): 4 waves per workgroup with 2 x 2 patches -- 1.0 instead of 1.5 operand dwords per MFMA -- and
// stages of 8192 floats, so that TWO workgroups share a CU: the two waves of a SIMD then belong to different
// workgroups and do not meet at the same barriers.
// CX: the compacted backward (compact.hip) -- the A operands are the first ceil(active / 32) tiles of the compacted d(pre-activation)
// images, the B operands the activation rows of the listed samples, gathered row by row.  Its own instantiation of the kernel: the
// dense kernel's registers (256 VGPRs, no scratch) are not touched by the gather's address arithmetic.
template <int NWV_, int STAGE_, bool CX_ = false>
struct WMode {
    static constexpr int NWV = NWV_, STAGE = STAGE_, LDS_DATA = 2 * STAGE_ * 4 + 4096, LDS_BYTES = LDS_DATA + NH_CLK_LDS_BYTES;
    static constexpr bool CX = CX_;
};
// (stage = one 32-sample tile of a 256 + 256 (128 + 128) row job plus a 64-row side region)
using WModeWide = WMode<8, 18432>;
using WModeNarrow = WMode<4, 9216>;
using WModeWideCx = WMode<8, 18432, true>;
using WModeNarrowCx = WMode<4, 9216, true>;
// split-K partial of one workgroup: [accumulator tiles of the largest job][16 regs][64 lanes], then 512 bias partials
// (then 128 floats of per-wave timeline records in the instrumented build); WgradArgs::part_bias / part_stride
#ifdef NH_WGRAD_TIMELINE
constexpr int NH_PART_EXTRA = 512 + 128;
#else
constexpr int NH_PART_EXTRA = 512;
#endif

struct WgradArgs {
    const float* stash;
    const float* grad;
    float* partial;
    int64_t nt;
    int njobs, total_wgs;
    int part_bias, part_stride;  // floats: offset of the bias partials inside a workgroup's partial, size of a partial
    unsigned long long* clk;     // shader-clock probe counters, or NULL (nh_prof_clock_slot)
    const int* cidx;             // compacted backward (MD::CX launches): the sample list and its statistics
    const int* cstats;
    JobDev jobs[NH_JOBS_DEV];
};
struct ReduceArgs {
    const float* partial;
    float* g_params;
    const unsigned* gscale;  // fp16 data-gradient chains: device word with the bits of max|g_out| (the d(pre-activation) images
                             // carry nh_gscale_of of it, the reduction multiplies by nh_gscale_inv -- exact); else NULL
    float w_unscale;         // a constant factor on the weight gradients (1: the stash rows are plain values)
    int njobs, total_wgs;
    int part_bias, part_stride;
    JobRed jobs[NH_JOBS_DEV];
    signed char xslot[4 * NH16_KRX_EXT];  // stash slot row -> reference column of the encoding (< 100), or -1
    signed char dslot[4 * NH16_KRD_EXT];
};

// the P operands (one per tile) of one k-step: P consecutive floats of one sample
template <int P>
struct WOp {
    float v[P];
};
template <int P>
NH_DEVICE void wop_load(WOp<P>& o, const float* p) {
    if constexpr (P == 4) {
        const float4 t = *(const float4*)p;
        o.v[0] = t.x, o.v[1] = t.y, o.v[2] = t.z, o.v[3] = t.w;
    } else if constexpr (P == 2) {
        const float2 t = *(const float2*)p;
        o.v[0] = t.x, o.v[1] = t.y;
    } else {
        o.v[0] = p[0];
    }
}
// SK: 0 no side block; 1: ONE extra A tile (its operand S.v[0]) against the job's B tiles; 2: SB extra B tiles (operands
// S.v[0..SB-1], interleaved rows like every block) against the job's A tiles.  NSP: side accumulator tiles per wave.
template <int SK_, int SB_, int NSP_>
struct WSide {
    static constexpr int SK = SK_, SB = SB_, NSP = NSP_, SW = SK_ == 2 ? SB_ : 1, NACC = SK_ == 2 ? NSP_ : 1;
    static constexpr int SR = SK_ ? 32 * SW : 0;  // rows of the side region
};
using NoSide = WSide<0, 1, 1>;
template <int PO, int PI, class SD>
struct WStep {
    WOp<PO> A;
    WOp<PI> B;
    WOp<SD::SW> S;
};
template <int PO, int PI, class SD>
NH_DEVICE void wstep_load(WStep<PO, PI, SD>& o, const float* pa, const float* pb, const float* ps) {
    wop_load<PO>(o.A, pa);
    wop_load<PI>(o.B, pb);
    if constexpr (SD::SK != 0) wop_load<SD::SW>(o.S, ps);
}
// wave-uniform pick of one of P register values (v_cndmask chain: one or three VALU instructions)
template <int P>
NH_DEVICE float wop_pick(const WOp<P>& o, int idx) {
    float r = o.v[0];
#pragma unroll
    for (int q = 1; q < P; ++q) r = idx == q ? o.v[q] : r;
    return r;
}
// which (A tile x, side tile y) / B tile a wave's j-th side accumulator belongs to (see WSideMap)
struct WSideSel {
    int x[2], y[2];
    bool on[2], bias;
};
// BX: which A tiles this wave sums for the bias gradient: -1 none, 0..3 that tile only, 4 all of them
template <int PO, int PI, int BX, class SD>
NH_DEVICE void wstep_mfma(const WStep<PO, PI, SD>& o, f32x16 (&acc)[PO][PI], float (&bsum)[PO], f32x16 (&sacc)[SD::NACC], float& sbsum,
                          float (&srow)[PI], const WSideSel& ss) {
#pragma unroll
    for (int x = 0; x < PO; ++x) {
        if (BX == 4 || BX == x) bsum[x] += o.A.v[x];
#pragma unroll
        for (int y = 0; y < PI; ++y) acc[x][y] = nh_mfma32(o.A.v[x], o.B.v[y], acc[x][y]);
    }
    if constexpr (SD::SK == 1) {  // (the side region's one useful row: S.v[0] = its value for this lane's sample) x (the B operands)
        // (unconditional in every wave: a wave-uniform `if` here is if-converted into a select per value, which costs more
        // than the PI + 1 redundant VALU instructions; only the waves of grid row 0 store their sums)
        sbsum += o.S.v[0];
#pragma unroll
        for (int y = 0; y < PI; ++y) srow[y] = fmaf(o.S.v[0], o.B.v[y], srow[y]);
    } else if constexpr (SD::SK == 2) {  // (A tile ss.x[j] of this wave's patch) x (side B tile ss.y[j])
#pragma unroll
        for (int j = 0; j < SD::NSP; ++j)
            if (ss.on[j]) sacc[j] = nh_mfma32(wop_pick<PO>(o.A, ss.x[j]), wop_pick<SD::SW>(o.S, ss.y[j]), sacc[j]);
    }
}

// One stage copy, cut into 1-KiB pieces (one DMA instruction each): block A (ntile * a_fl floats) -> stage[0 ..), block
// B (ntile * b_fl floats) -> stage[g * a_fl ..).  Wave w issues pieces w, w+8, ...; `issue(n)` emits the next n of
// them, so that the copy of stage n+1 is spread over the MFMA groups of stage n.  LDS destinations are byte addresses.
struct WStageDma {
    NhDmaSrc sa, sb, sc;
    unsigned dst;  // LDS byte address of the stage
    int pa, pab, ptot, boff, coff, q, lane16;
    // gathered blocks (compacted backward): sample slot0 + k of the stage is row cidx[slot0 + k] of the WHOLE region the descriptor
    // then spans; b_sh / s_sh = log2 of a row's bytes; s_gather: the side block lives in the stash (B-side) and is gathered too
    const int* cidx;
    int slot0, b_sh, s_sh;
    bool s_gather;
    // the row of this wave's NEXT gathered B piece, fetched one piece ahead: its scalar load is long complete at the piece's turn
    // (fetched at the piece itself, the wave would wait for the scalar cache once per piece, ~10 % of a stage)
    unsigned nxt_row;
    int nxt_pq;
    NH_MEMBER void prefetch_b(int pq) {
        nxt_pq = -1;
        if (b_sh >= 10 && pq >= 0 && pq < pab - pa) {
            nxt_pq = pq;
            nxt_row = (unsigned)nh_uload_i32(cidx, slot0 + (int)(((unsigned)pq << 10) >> b_sh));
        }
    }
    // blocks A, B and (s_fl > 0) the side block: ntile * {a,b,s}_fl floats each, landing at stage + 0, g*a_fl, g*(a_fl+b_fl)
    NH_MEMBER void init(const float* ga, const float* gb, const float* gs, int a_fl, int b_fl, int s_fl, int ntile, int g,
                        unsigned stage_addr, int wave, int lane) {
        sa = nh_dma_src(ga, (unsigned)(ntile * a_fl * 4));
        sb = nh_dma_src(gb, (unsigned)(ntile * b_fl * 4));
        dst = stage_addr;
        pa = ntile * a_fl / 256;
        pab = pa + ntile * b_fl / 256;
        ptot = pab;
        boff = (g * a_fl - pa * 256) * 4;  // piece q >= pa lands at stage + (g*a_fl + (q - pa)*256) floats
        coff = 0;
        if (s_fl > 0) {
            sc = nh_dma_src(gs, (unsigned)(ntile * s_fl * 4));
            ptot = pab + ntile * s_fl / 256;
            coff = (g * (a_fl + b_fl) - pab * 256) * 4;
        }
        q = wave;
        lane16 = lane * 16;
    }
    // the same for a compacted stage: block A is contiguous (the compacted image); B0 (and S0 for a B-side block) name the whole
    // region, `slot` the stage's first sample slot
    template <int NWV>
    NH_MEMBER void init_cx(const float* ga, const float* B0, const float* gs, unsigned b_bytes, unsigned s_bytes, int a_fl, int b_fl,
                           int s_fl, int ntile, int g, unsigned stage_addr, int wave, int lane, int slot) {
        init(ga, B0, gs, a_fl, b_fl, s_fl, ntile, g, stage_addr, wave, lane);
        sb = nh_dma_src(B0, b_bytes);
        if (s_fl > 0 && s_gather) sc = nh_dma_src(gs, s_bytes);
        slot0 = slot;
        prefetch_b((wave < pa ? wave + (pa - wave + NWV - 1) / NWV * NWV : wave) - pa);  // this wave's first B piece
    }
    // 1-KiB piece `pq` of a gathered block whose rows are (1 << sh) bytes; row_known: `row0` is the piece's row (prefetch_b)
    NH_MEMBER void gather_piece(const NhDmaSrc& src, int pq, int sh, unsigned lds_dst, bool row_known = false, unsigned row0 = 0u) const {
        if (sh >= 10) {  // a quarter / half / whole row of ONE sample: the source offset is wave-uniform
            const unsigned byte0 = (unsigned)pq << 10;
            const unsigned row = row_known ? row0 : (unsigned)nh_uload_i32(cidx, slot0 + (int)(byte0 >> sh));
            nh_dma16a(src, lane16, (int)((row << sh) + (byte0 & ((1u << sh) - 1u))), lds_dst);
        } else {         // 2 / 4 / 8 whole rows: each lane picks its sample's
            const int n = 1 << (10 - sh), s0 = slot0 + pq * n, e = lane16 >> sh;
            unsigned row = (unsigned)nh_uload_i32(cidx, s0);
            for (int k = 1; k < n; ++k) {
                const unsigned rk = (unsigned)nh_uload_i32(cidx, s0 + k);
                row = e == k ? rk : row;  // (a rolled loop over scalar loads: one v_cndmask per row)
            }
            nh_dma16a(src, (int)((row << sh) + ((unsigned)lane16 & ((1u << sh) - 1u))), 0, lds_dst);
        }
    }
    template <int NWV, bool SIDE, bool CX>
    NH_MEMBER void issue(int n) {
        for (int c = 0; c < n && q < ptot; ++c, q += NWV) {
            // (the instrumented build, and the compacted kernel: with more scalar state live -- the stamp bookkeeping; the list
            // pointer and row shifts -- ROCm 7.2's clang hands VGPRs to the "s" descriptor operand of the copy instruction and the
            // build fails; the descriptors are re-uniformised per piece.  This costs the instrumented kernel ~10 % -- more on small
            // jobs: its timeline shows the order of events and the wait / barrier shares, not the product kernel's absolute times.)
#if defined(NH_WGRAD_TIMELINE) && !defined(NERFHIP_EMU)
            constexpr bool reuniform = true;
#elif !defined(NERFHIP_EMU)
            constexpr bool reuniform = CX;
#else
            constexpr bool reuniform = false;
#endif
            if constexpr (reuniform) {
#ifndef NERFHIP_EMU
                NhDmaSrc ta = sa, tb = sb, tc = sc;
                for (int e = 0; e < 4; ++e) {
                    ta.r[e] = __builtin_amdgcn_readfirstlane(ta.r[e]);
                    tb.r[e] = __builtin_amdgcn_readfirstlane(tb.r[e]);
                    if (SIDE) tc.r[e] = __builtin_amdgcn_readfirstlane(tc.r[e]);
                }
                piece<NWV, SIDE, CX>(ta, tb, tc);
#endif
            } else {
                piece<NWV, SIDE, CX>(sa, sb, sc);
            }
        }
    }
    template <int NWV, bool SIDE, bool CX>
    NH_MEMBER void piece(const NhDmaSrc& ta, const NhDmaSrc& tb, const NhDmaSrc& tc) {
        if (q < pa)
            nh_dma16a(ta, lane16, q * 1024, dst + q * 1024);
        else if (!SIDE || q < pab) {
            if (CX && cidx) {
                gather_piece(tb, q - pa, b_sh, dst + boff + q * 1024, nxt_pq == q - pa, nxt_row);
                prefetch_b(q - pa + NWV);
            } else
                nh_dma16a(tb, lane16, (q - pa) * 1024, dst + boff + q * 1024);
        } else {
            if (CX && cidx && s_gather)
                gather_piece(tc, q - pab, s_sh, dst + coff + q * 1024);
            else
                nh_dma16a(tc, lane16, (q - pab) * 1024, dst + coff + q * 1024);
        }
    }
};

// AR / BR: rows of the A / B region when known at compile time (0: read from the job) -- with constant strides the
// operand addresses of a whole stage are immediates of ONE base register.
template <class MD, int PO, int PI, int AR, int BR, int BX, class SD>
NH_DEVICE void wgrad_body(const WgradArgs& a, const JobDev& jb, int64_t t0, int64_t t1, int ow, int iw, int wave, int lane,
                          int64_t wg, bool active, float* lds) {
    const int i = lane & 31, k = lane >> 5;
    f32x16 acc[PO][PI];
    float bsum[PO];
    // side block (SD::SK != 0).  A-side: see wstep_mfma.  B-side: the po * SB pairs (A tile x, side tile y) of a wave row
    // are dealt to its wi waves, e = iw + j * wi -> x = e / SB, y = e % SB.
    constexpr bool SIDE = SD::SK != 0;
    f32x16 sacc[SD::NACC];
    float sbsum = 0.0f, srow[PI];
#pragma unroll
    for (int y = 0; y < PI; ++y) srow[y] = 0.0f;
    WSideSel ss;
    ss.bias = false;
#pragma unroll
    for (int j = 0; j < 2; ++j) ss.x[j] = ss.y[j] = 0, ss.on[j] = false;
    constexpr int srows = SD::SR;
    if constexpr (SD::SK == 1) {
        ss.on[0] = active && ow == 0;  // one wave per column of the grid carries the row for the column's PI tiles
        ss.bias = active && ow == 0 && iw == 0;
    } else if constexpr (SD::SK == 2) {
#pragma unroll
        for (int j = 0; j < SD::NSP; ++j) {
            const int e = iw + j * jb.wi;
            ss.on[j] = active && e < PO * SD::SB && ow * PO + e / SD::SB < jb.a_tiles;
            ss.x[j] = e / SD::SB;
            ss.y[j] = e % SD::SB;
        }
    }
#pragma unroll
    for (int j = 0; j < SD::NACC; ++j)
#pragma unroll
        for (int c = 0; c < 16; ++c) sacc[j][c] = 0.0f;
#pragma unroll
    for (int x = 0; x < PO; ++x) {
        bsum[x] = 0.0f;
#pragma unroll
        for (int y = 0; y < PI; ++y)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[x][y][c] = 0.0f;
    }
    const int ar = AR ? AR : jb.a_rows, br = BR ? BR : jb.b_rows;
    constexpr int NH_WG_STAGE_FLOATS = MD::STAGE;
    // fixed row counts: tiles per stage
    constexpr int G_FIXED = NH_WG_STAGE_FLOATS / (32 * ((AR && BR) ? AR + BR + SD::SR : 1));
    static_assert(G_FIXED >= 1, "a fixed-shape body must fit the mode's stage buffer");
    const int g = (AR && BR) ? G_FIXED : jb.g;
    const int a_fl = 32 * ar, b_fl = 32 * br, s_fl = 32 * srows;
    const float* A0 = a.grad + (size_t)32 * (size_t)a.nt * (size_t)jb.a_prefix;
    const float* B0 = a.stash + (size_t)32 * (size_t)a.nt * (size_t)jb.b_prefix;
    // (an A-side region lives in the gradient scratch, a B-side region in the activation stash)
    const float* S0 = SIDE ? (SD::SK == 1 ? a.grad : a.stash) + (size_t)32 * (size_t)a.nt * (size_t)jb.side_prefix : nullptr;
    const int nstage = (int)((t1 - t0 + g - 1) / g);
    const float* ga = A0 + (size_t)t0 * a_fl;  // stage n+1's blocks (running pointers: one 64-bit add per stage)
    const float* gb = B0 + (size_t)t0 * b_fl;
    const float* gs = SIDE ? S0 + (size_t)t0 * s_fl : nullptr;
    int left = (int)(t1 - t0);                 // tiles not yet requested
    const unsigned lds_addr = nh_lds_addr(lds);
    constexpr bool CX = MD::CX;
    // compacted backward: tile t of the job is slots 32 t .. 32 t + 31 of the sample list -- the A block (and an A-side block) are
    // tile t of the compacted images, the B block (and a B-side block) the listed samples' rows of the whole stash region
    constexpr bool SGATHER = CX && SD::SK == 2;
    int slot = (int)t0 * 32;
    const unsigned b_bytes = (unsigned)((size_t)a.nt * 32 * (size_t)br * 4), s_bytes = (unsigned)((size_t)a.nt * 32 * (size_t)(srows ? srows : 1) * 4);
    WStageDma dma;
    dma.ptot = 0;
    if constexpr (CX) {
        dma.cidx = a.cidx;
        dma.b_sh = 31 - __builtin_clz((unsigned)(br * 4));
        dma.s_sh = 31 - __builtin_clz((unsigned)((srows ? srows : 1) * 4));
        dma.s_gather = SGATHER;
    }
    if (nstage > 0) {
        const int nt0 = left < g ? left : g;
        if (CX && a.cidx) {  // (a.cidx == NULL in a CX launch: the stash is in list order itself -- contiguous blocks, the list's count)
            dma.template init_cx<MD::NWV>(ga, B0, SGATHER ? S0 : gs, b_bytes, s_bytes, a_fl, b_fl, s_fl, nt0, g, lds_addr, wave, lane, slot);
            slot += 32 * nt0;
        } else {
            dma.init(ga, gb, gs, a_fl, b_fl, s_fl, nt0, g, lds_addr, wave, lane);
        }
        dma.template issue<MD::NWV, SIDE, CX>(1 << 20);
        ga += (size_t)nt0 * a_fl, gb += (size_t)nt0 * b_fl, left -= nt0;
        if (SIDE) gs += (size_t)nt0 * s_fl;
    }
    int ntile = (int)(t1 - t0) < g ? (int)(t1 - t0) : g;  // tiles of the stage being multiplied
#ifdef NH_WGRAD_TIMELINE
    unsigned long long tl_wait = 0, tl_bar = 0, tl_loop = 0, tl_t;
#define NH_TL(acc)                                   \
    do {                                             \
        const unsigned long long _n = nh_core_clock(); \
        acc += _n - tl_t;                            \
        tl_t = _n;                                   \
    } while (0)
    tl_t = nh_core_clock();
#else
#define NH_TL(acc)
#endif
    for (int n = 0; n < nstage; ++n) {
        const float* buf = lds + (n & 1) * NH_WG_STAGE_FLOATS;
        // lane (i, k): sample 2e + k of k-step e, rows PO*i .. of the wave's A block / PI*i .. of its B block
        const float* pa = buf + k * ar + 32 * PO * ow + PO * i;
        const float* pb = buf + g * a_fl + k * br + 32 * PI * iw + PI * i;
        // side operand(s) of lane (i, k): its rows SW * i .. of the side B tiles, or (A-side) THE row of the side region
        const float* ps = SIDE ? buf + g * (a_fl + b_fl) + k * srows + (SD::SK == 1 ? (jb.side >> 24) : SD::SW * i) : buf;
        NH_TL(tl_loop);
        nh_wait_vmem();
        NH_TL(tl_wait);
        nh_block_sync();  // stage n has landed for every wave; everybody is done reading the other buffer
        NH_TL(tl_bar);
        WStep<PO, PI, SD> c0, c1;
        if (active) wstep_load(c0, pa, pb, ps);
        nh_sched_fence();  // first operand reads leave before the scalar set-up of the next copy
        const int ntn = left < g ? left : g;
        dma.ptot = 0;
        if (ntn > 0) {
            if (CX && a.cidx) {
                dma.template init_cx<MD::NWV>(ga, B0, SGATHER ? S0 : gs, b_bytes, s_bytes, a_fl, b_fl, s_fl, ntn, g,
                            lds_addr + (unsigned)(((n + 1) & 1) * NH_WG_STAGE_FLOATS * 4), wave, lane, slot);
                slot += 32 * ntn;
            } else {
                dma.init(ga, gb, gs, a_fl, b_fl, s_fl, ntn, g, lds_addr + (unsigned)(((n + 1) & 1) * NH_WG_STAGE_FLOATS * 4), wave, lane);
            }
        }
        ga += (size_t)ntn * a_fl, gb += (size_t)ntn * b_fl, left -= ntn;
        if (SIDE) gs += (size_t)ntn * s_fl;
        if (active) {
            if (AR && BR && G_FIXED == 1) {
                // one tile per stage, constant strides: 16 k-steps fully unrolled, every operand address an immediate
#pragma unroll
                for (int s = 0; s < 16; s += 2) {
                    wstep_load(c1, pa + (s + 1) * 2 * AR, pb + (s + 1) * 2 * BR, ps + (s + 1) * 2 * srows);
                    dma.template issue<MD::NWV, SIDE, CX>(1);
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BX, SD>(c0, acc, bsum, sacc, sbsum, srow, ss);
                    // (last: one k-step past the stage, unused)
                    wstep_load(c0, pa + (s + 2) * 2 * AR, pb + (s + 2) * 2 * BR, ps + (s + 2) * 2 * srows);
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BX, SD>(c1, acc, bsum, sacc, sbsum, srow, ss);
                }
            } else {
                const int steps = 16 * ntile;  // k-steps of two samples each
                for (int s = 0; s < steps; s += 2) {
                    pa += 2 * ar, pb += 2 * br, ps += 2 * srows;
                    wstep_load(c1, pa, pb, ps);
                    dma.template issue<MD::NWV, SIDE, CX>(1);  // the next stage streams in underneath the MFMAs (at most 9 pieces per wave and stage)
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BX, SD>(c0, acc, bsum, sacc, sbsum, srow, ss);
                    pa += 2 * ar, pb += 2 * br, ps += 2 * srows;
                    wstep_load(c0, pa, pb, ps);  // the last one reads one k-step past the stage (slack / other block): unused
                    nh_sched_fence();
                    wstep_mfma<PO, PI, BX, SD>(c1, acc, bsum, sacc, sbsum, srow, ss);
                }
            }
        }
        dma.template issue<MD::NWV, SIDE, CX>(1 << 20);  // idle waves, and whatever a short stage left over
        ntile = ntn;
    }
#ifdef NH_WGRAD_TIMELINE
    NH_TL(tl_loop);
    if (lane == 0 && active) {
        unsigned long long* dbg = (unsigned long long*)(a.partial + (size_t)wg * a.part_stride + a.part_bias + 512) + wave * 8;
        dbg[6] = (tl_wait << 32) | (tl_bar & 0xffffffffull);
        dbg[7] = tl_loop;
    }
#endif
#undef NH_TL
    if (!active) return;
    // split-K partial of this workgroup: accumulator tile (a_t, b_t) of the job at [(a_t * b_tiles + b_t)][16 regs][64 lanes]
    float* part = a.partial + (size_t)wg * a.part_stride;
#pragma unroll
    for (int x = 0; x < PO; ++x) {
        const int a_t = ow * PO + x;
        if (a_t >= jb.a_tiles) continue;
#pragma unroll
        for (int y = 0; y < PI; ++y) {
            const int b_t = iw * PI + y;
            if (b_t >= jb.b_tiles) continue;
            float* dst = part + (size_t)(a_t * jb.b_tiles + b_t) * 1024 + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c * 64] = acc[x][y][c];
        }
        if (BX == 4 || BX == x) {  // bias gradient = row sums of A over this workgroup's samples
            const float tot = bsum[x] + nh_shfl_xor(bsum[x], 32);
            if (k == 0) part[a.part_bias + a_t * 32 + i] = tot;
        }
    }
    if constexpr (SD::SK == 2) {
        // side accumulator tiles behind the job's own: pair (a_t, y) at a_tiles * b_tiles + a_t * SB + y
        const int base = jb.a_tiles * jb.b_tiles;
#pragma unroll
        for (int j = 0; j < SD::NACC; ++j) {
            if (!ss.on[j]) continue;
            float* dst = part + (size_t)(base + (ow * PO + ss.x[j]) * SD::SB + ss.y[j]) * 1024 + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c * 64] = sacc[j][c];
        }
    } else if constexpr (SD::SK == 1) {
        // The side row as the reduce kernel reads it: row r of the side A tile against B tile b_t is accumulator tile
        // a_tiles * b_tiles + b_t, register c / lane half h with (c & 3) + 8 (c >> 2) + 4 h == r, lane & 31 = the B row index
        // i.  (Nothing else of those tiles is ever read: the reduce kernel only unpacks rows r_lo .. r_hi - 1 = r.)
        const int r = jb.side >> 24, c = (r & 3) + 4 * (r >> 3), h = (r >> 2) & 1;
        if (ss.on[0]) {
#pragma unroll
            for (int y = 0; y < PI; ++y) {
                const float tot = srow[y] + nh_shfl_xor(srow[y], 32);
                const int b_t = iw * PI + y;
                if (k == 0 && b_t < jb.b_tiles) part[(size_t)(jb.a_tiles * jb.b_tiles + b_t) * 1024 + c * 64 + 32 * h + i] = tot;
            }
        }
        if (ss.bias) {  // the side region's row sums (its bias gradient) behind the job's bias partials (every entry: the sum)
            const float tot = sbsum + nh_shfl_xor(sbsum, 32);
            if (k == 0) part[a.part_bias + jb.a_tiles * 32 + i] = tot;
        }
    }
}

template <class MD, int PO, int PI, int AR, int BR, class SD>
NH_DEVICE void wgrad_bias_dispatch(const WgradArgs& a, const JobDev& jb, int64_t t0, int64_t t1, int ow, int iw, int wave,
                                   int lane, int64_t wg, bool active, float* lds) {
    // the bias sums of A tile x are taken by the wave of column iw == x % wi: one tile per wave when wi >= PO
    // (computed even when the job carries no bias tensor: the reduce kernel ignores them)
    if (jb.wi >= PO && PO > 1) {
        switch (iw) {
            case 0: wgrad_body<MD, PO, PI, AR, BR, 0, SD>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
            case 1: wgrad_body<MD, PO, PI, AR, BR, (PO > 1 ? 1 : -1), SD>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
            case 2: wgrad_body<MD, PO, PI, AR, BR, (PO > 2 ? 2 : -1), SD>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
            case 3: wgrad_body<MD, PO, PI, AR, BR, (PO > 3 ? 3 : -1), SD>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
            default: wgrad_body<MD, PO, PI, AR, BR, -1, SD>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        }
    } else if (iw == 0) {
        wgrad_body<MD, PO, PI, AR, BR, 4, SD>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    } else {
        wgrad_body<MD, PO, PI, AR, BR, -1, SD>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    }
}

// side blocks exist for the patches they occur with (plan.cpp attach_sides): A-side (fc_alpha) on the fc_feat job and
// B-side with two tiles (a skip layer's encoding columns) on a hidden job -- 4 x 2 patches (8 waves) / 2 x 2 (4 waves) --,
// B-side with one tile (the direction columns) on the layers_dir job -- 2 x 2 / 1 x 2 patches
template <class MD, int PO, int PI, int AR, int BR>
NH_DEVICE void wgrad_side_dispatch(const WgradArgs& a, const JobDev& jb, int64_t t0, int64_t t1, int ow, int iw, int wave, int lane,
                                   int64_t wg, bool active, float* lds) {
    const int kind = jb.side & 255, st = (jb.side >> 8) & 255;
    constexpr bool host = (MD::NWV == 8 && PO == 4 && PI == 2) || (MD::NWV == 4 && PO == 2 && PI == 2);
    constexpr bool dirj = (MD::NWV == 8 && PO == 2 && PI == 2) || (MD::NWV == 4 && ((PO == 1 && PI == 2) || (PO == 2 && PI == 1)));
    if constexpr (host && AR != 0) {  // (fixed-shape bodies only: plan.cpp attaches a side row to 256 x 256 / 128 x 128 jobs)
        if (kind == 1) return wgrad_bias_dispatch<MD, PO, PI, AR, BR, WSide<1, 1, 1>>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    }
    if constexpr (host) {
#ifdef NH_WGRAD_SIDE2  // (A/B builds only: two more accumulator tiles per wave push the 4 x 2 patch body over 256 VGPRs)
        if constexpr (MD::NWV == 8) {  // (a 64-row side region next to 128 + 128 rows does not fit the 4-wave mode's stage)
            if (kind == 2 && st == 2)
                return wgrad_bias_dispatch<MD, PO, PI, AR, BR, WSide<2, 2, 2>>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
        }
#endif
    }
    if constexpr (dirj && AR == 0) {
        if (kind == 2 && st == 1) return wgrad_bias_dispatch<MD, PO, PI, AR, BR, WSide<2, 1, 1>>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    }
    wgrad_bias_dispatch<MD, PO, PI, AR, BR, NoSide>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
}

template <class MD, int PO, int PI>
NH_DEVICE void wgrad_dispatch(const WgradArgs& a, const JobDev& jb, int64_t t0, int64_t t1, int ow, int iw, int wave, int lane,
                              int64_t wg, bool active, float* lds) {
    // fixed-shape bodies only where they are launched (and fit the mode's stage buffer)
    if constexpr (MD::NWV == 8 && PO == 4 && PI == 2) {
        if (jb.a_rows == 256 && jb.b_rows == 256)  // the 256x256 jobs: 91 % of the 8x256 FLOPs
            return wgrad_side_dispatch<MD, PO, PI, 256, 256>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    }
    if constexpr (MD::NWV == 4 && PO == 2 && PI == 2) {
        if (jb.a_rows == 128 && jb.b_rows == 128)  // the 128x128 jobs of 128-wide nets
            return wgrad_side_dispatch<MD, PO, PI, 128, 128>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
    }
    wgrad_side_dispatch<MD, PO, PI, 0, 0>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds);
}

template <class MD>
NH_KERNEL void NH_LB(64 * MD::NWV, 2) k_wgrad(WgradArgs a) {
    NH_DYN_LDS(smem);
    float* lds = (float*)smem;
    nh_clk_begin(a.clk, (unsigned long long*)(smem + MD::LDS_DATA));
#ifdef NH_WGRAD_TIMELINE
    const unsigned long long t_begin = nh_wall_clock();
    const unsigned long long c_begin = nh_core_clock();
#endif
    const int64_t wg = blockIdx.x;
    int ji = 0;
    for (int q = 1; q < a.njobs; ++q)
        if ((int)wg >= a.jobs[q].wg_start) ji = q;
    const JobDev jb = a.jobs[ji];
    const int nks = (ji + 1 < a.njobs ? a.jobs[ji + 1].wg_start : a.total_wgs) - jb.wg_start;
    const int ks = (int)wg - jb.wg_start;
    // (compacted: the job's tiles are those of the sample list, however many the launch's cotangents left)
    const int64_t ntl = MD::CX ? (int64_t)((nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) + 31) >> 5) : a.nt;
    const int64_t t0 = ntl * ks / nks, t1 = ntl * (ks + 1) / nks;
    const int lane = nh_lane(), wave = nh_wave_in_block();
    const bool active = wave < jb.wo * jb.wi;  // idle waves still copy and synchronise
    // wave -> patch (ow, iw); the column index is rotated by the row so that the waves of one column (which carry the
    // same bias tile index) sit on different SIMDs (wave w runs on SIMD w % 4)
    const int ow = wave / jb.wi, iw = (wave % jb.wi + ow) % jb.wi;
    const int sel = jb.po * 8 + jb.pi;
    switch (sel) {
        case 4 * 8 + 2: wgrad_dispatch<MD, 4, 2>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 2 * 8 + 4: wgrad_dispatch<MD, 2, 4>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 4 * 8 + 1: wgrad_dispatch<MD, 4, 1>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 1 * 8 + 4: wgrad_dispatch<MD, 1, 4>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 2 * 8 + 2: wgrad_dispatch<MD, 2, 2>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 2 * 8 + 1: wgrad_dispatch<MD, 2, 1>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        case 1 * 8 + 2: wgrad_dispatch<MD, 1, 2>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
        default: wgrad_dispatch<MD, 1, 1>(a, jb, t0, t1, ow, iw, wave, lane, wg, active, lds); break;
    }
    nh_clk_end((const unsigned long long*)(smem + MD::LDS_DATA));
#ifdef NH_WGRAD_TIMELINE
    if (lane == 0 && active) {  // timeline record (8 x u64 per wave): wall begin/end, job, K-slice, core-clock begin/end
        unsigned long long* dbg = (unsigned long long*)(a.partial + (size_t)wg * a.part_stride + a.part_bias + 512) + wave * 8;
        dbg[0] = t_begin;
        dbg[1] = nh_wall_clock();
        dbg[2] = (unsigned long long)ji;
        dbg[3] = (unsigned long long)ks;
        dbg[4] = c_begin;
        dbg[5] = nh_core_clock();
    }
#endif
}

// fixed-order split-K reduction + scatter into the reference parameter layout.  Accumulator tile (a_t, b_t), register c,
// lane l holds dW[out_row][in_row] with (MFMA row m = (c&3) + 8(c>>2) + 4(l>>5), column j = l&31, and the row
// interleave of the wide operand reads)  out_row = 32*po*(a_t/po) + po*m + a_t%po,  in_row = 32*pi*(b_t/pi) + pi*j + b_t%pi.
NH_KERNEL void k_wgrad_reduce(ReduceArgs a) {
    NH_SHARED float part[256];
    const int ji = (int)(blockIdx.x >> 8);
    const JobRed jb = a.jobs[ji];
    const float unscale = a.gscale ? nh_gscale_inv(*a.gscale) : 1.0f;
    // A job owns 256 blocks of 256 threads.  Jobs with few accumulator tiles would leave most of them idle and a handful of
    // threads with ~150 dependent partial loads each (the 64- and 128-wide nets: this kernel was 3-6 % of their step), so a
    // block covers 256 >> ksl elements with (1 << ksl) K-slices per element: slice s sums partials s, s + KS, ... and the
    // slices are combined through LDS in slice order -- a fixed order: bit-reproducible, no atomics.
    const int ksl = jb.ks_log & 255, bias_t0 = jb.ks_log >> 8, epb = 256 >> ksl, nsl = 1 << ksl;
    const int e_local = (int)threadIdx.x & (epb - 1), slice = (int)threadIdx.x >> (8 - ksl);
    const int local = (int)(blockIdx.x & 255u) * epb + e_local;
    const int lane = local & 63, c = (local >> 6) & 15, tile = local >> 10;  // accumulator tile (a_t, b_t) = a_t * b_tiles + b_t
    const int a_t = tile / jb.b_tiles, b_t = tile % jb.b_tiles;
    const bool in_job = a_t < jb.a_tiles;
    int nxt = ji + 1;  // (a side block shares its host's workgroups: skip entries with the same wg_start)
    while (nxt < a.njobs && a.jobs[nxt].wg_start == jb.wg_start) ++nxt;
    const int nks = (nxt < a.njobs ? a.jobs[nxt].wg_start : a.total_wgs) - jb.wg_start;
    const int m = (c & 3) + 8 * (c >> 2) + 4 * (lane >> 5);
    const int out_row = 32 * jb.po * (a_t / jb.po) + jb.po * m + a_t % jb.po;
    const int in_row = 32 * jb.pi * (b_t / jb.pi) + jb.pi * (lane & 31) + b_t % jb.pi;
    int col = -1;
    if (in_job && out_row >= jb.r_lo && out_row < jb.r_hi) {
        if (jb.col_kind == 0) {
            if (in_row < jb.col_count) col = jb.col_base + in_row;
        } else {
            const int cc = jb.col_kind == 1 ? (int)a.xslot[in_row] : (int)a.dslot[in_row];  // stash slot row -> column
            if (cc >= 0) col = jb.col_base + cc;
        }
    }
    float total = 0.0f;
    if (col >= 0) {
        // eight interleaved running sums (a fixed order) keep eight loads in flight per lane
        float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* p = a.partial + (size_t)jb.wg_start * a.part_stride + ((size_t)(jb.tile0 + tile) * 16 + c) * 64 + lane;
        const size_t step = (size_t)nsl * a.part_stride;
        int q = slice;
        for (; q + 7 * nsl < nks; q += 8 * nsl) {
#pragma unroll
            for (int u = 0; u < 8; ++u) sum[u] += p[(size_t)q * a.part_stride + u * step];
        }
        for (; q < nks; q += nsl) sum[0] += p[(size_t)q * a.part_stride];
        total = ((sum[0] + sum[1]) + (sum[2] + sum[3])) + ((sum[4] + sum[5]) + (sum[6] + sum[7]));
    }
    if (ksl > 0) {
        part[threadIdx.x] = total;
        nh_block_sync();
        if (slice == 0) {
            total = 0.0f;
            for (int sidx = 0; sidx < nsl; ++sidx) total += part[sidx * epb + e_local];
        }
    }
    if (col >= 0 && slice == 0) a.g_params[(size_t)jb.w_off + (size_t)(out_row - jb.r_lo) * jb.w_ld + col] = total * (unscale * a.w_unscale);
    if (slice == 0 && in_job && jb.bias_off >= 0 && b_t == 0 && c == 0 && lane < 32) {
        const int brow = 32 * jb.po * (a_t / jb.po) + jb.po * lane + a_t % jb.po;
        if (brow >= jb.r_lo && brow < jb.r_hi) {
            // (the same eight interleaved sums: a one-deep chain of nks dependent loads used to set this kernel's duration)
            float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* p = a.partial + (size_t)jb.wg_start * a.part_stride + a.part_bias + (bias_t0 + a_t) * 32 + lane;
            int q = 0;
            for (; q + 8 <= nks; q += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) sum[u] += p[(size_t)(q + u) * a.part_stride];
            }
            for (; q < nks; ++q) sum[0] += p[(size_t)q * a.part_stride];
            a.g_params[(size_t)jb.bias_off + (brow - jb.r_lo)] =
                (((sum[0] + sum[1]) + (sum[2] + sum[3])) + ((sum[4] + sum[5]) + (sum[6] + sum[7]))) * unscale;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
// Split-K allocation: job j gets ks_j workgroups with ks_j proportional to its per-tile cost (every workgroup then
// runs for about the same time), and sum ks_j == NH_WGRAD_TARGET_WGS exactly (largest-remainder rounding) -- the grid
// is a whole number of rounds over the chip, with no straggler round: THREE rounds of 256 eight-wave workgroups (one per
// CU) for 256-wide nets, TWO rounds of 512 four-wave workgroups (two per CU) for 128-wide ones.  Measured on MI355X:
// 512 / 768 / 1024 / 1280 / 2048 workgroups -> k_wgrad 0.845 / 0.855 / 0.852 / 0.846 / 0.841 of peak for 8x256 nets
// (more workgroups: more partials to write and reduce; fewer: a coarser tail), 0.551 / 0.564 / 0.584 for 4x128.
// Round 3, with the side tiles (7 jobs for 4x128): 768 / 1024 / 1280 / 1536 -> 0.600 / 0.607 / 0.618 / 0.607 for 4x128
// (1280 taken), 512 / 640 / 768 / 1024 -> 0.846 / 0.710 / 0.851 / 0.849 for 8x256 (768 stays; profiles/r03_variant_ab.txt).
void wgrad_schedule(const nerfhip_plan* p, int64_t nt, WgradArgs& w, ReduceArgs* r) {
#ifndef NH_WGS_NARROW  // (A/B builds only)
#define NH_WGS_NARROW 1280
#define NH_WGS_WIDE 768
#endif
    const int NH_WGRAD_TARGET_WGS = p->wgrad_waves == 4 ? NH_WGS_NARROW : NH_WGS_WIDE;
    const int stage_floats = p->wgrad_waves == 4 ? WModeNarrow::STAGE : WModeWide::STAGE;
    w.njobs = (int)p->jobs.size();  // <= NH_MAX_JOBS == NH_JOBS_DEV: nerfhip_plan_create refuses larger job lists
    int64_t cost[NH_JOBS_DEV];
    for (int q = 0; q < w.njobs; ++q) cost[q] = p->jobs[q].cost;
    int64_t total_cost = 0;
    for (int q = 0; q < w.njobs; ++q) total_cost += cost[q];
    int64_t ks[NH_JOBS_DEV], rem[NH_JOBS_DEV];
    int64_t used = 0;
    for (int q = 0; q < w.njobs; ++q) {
        const int64_t num = (int64_t)NH_WGRAD_TARGET_WGS * cost[q];
        ks[q] = num / total_cost;
        rem[q] = num % total_cost;
        if (ks[q] < 1) {
            ks[q] = 1;
            rem[q] = 0;
        }
        used += ks[q];
    }
    while (used < NH_WGRAD_TARGET_WGS) {  // hand out the remaining workgroups by largest remainder
        int best = 0;
        for (int q = 1; q < w.njobs; ++q)
            if (rem[q] > rem[best]) best = q;
        ks[best] += 1;
        rem[best] = -1;
        used += 1;
    }
    while (used > NH_WGRAD_TARGET_WGS) {  // (only if many jobs were lifted to 1) take from the largest
        int best = 0;
        for (int q = 1; q < w.njobs; ++q)
            if (ks[q] > ks[best]) best = q;
        if (ks[best] <= 1) break;
        ks[best] -= 1;
        used -= 1;
    }
    int start = 0, nred = 0;
    for (int q = 0; q < w.njobs; ++q) {
        const NhJob& j = p->jobs[q];
        if (ks[q] > nt) ks[q] = nt;
        JobDev& d = w.jobs[q];
        d.a_rows = j.a_region_rows;
        d.a_prefix = (int)j.a_row_prefix;
        d.a_tiles = j.a_tiles;
        d.b_rows = j.b_region_rows;
        d.b_prefix = (int)j.b_row_prefix;
        d.b_tiles = j.b_tiles;
        d.wo = j.wo;
        d.wi = j.wi;
        d.po = j.po;
        d.pi = j.pi;
        d.wg_start = start;
        d.side = j.side_kind | (j.side_tiles << 8) | (j.side_rows << 16) | ((j.side_kind == 1 ? j.s_r_lo : 0) << 24);
        d.side_prefix = (int)j.side_row_prefix;
        d.g = stage_floats / (32 * (j.a_region_rows + j.b_region_rows + (j.side_kind ? j.side_rows : 0)));
        if (d.g < 1) d.g = 1;
        if (r) {
            auto ks_log = [](int tiles) { return tiles <= 4 ? 4 : (tiles <= 16 ? 2 : 0); };  // 256 blocks x (256 >> ks_log) elements cover tiles * 1024
            if (nred + (j.side_kind ? 2 : 1) > NH_JOBS_DEV) {  // (refused by nerfhip_plan_create; never write past the table)
                nred += j.side_kind ? 2 : 1;
                start += (int)ks[q];
                continue;
            }
            JobRed& e = r->jobs[nred++];
            e.a_tiles = j.a_tiles;
            e.b_tiles = j.b_tiles;
            e.po = j.po;
            e.pi = j.pi;
            e.r_lo = j.r_lo;
            e.r_hi = j.r_hi;
            e.w_off = (int)j.w_off;
            e.w_ld = j.w_ld;
            e.col_kind = j.col_kind;
            e.col_base = j.col_base;
            e.col_count = j.col_count;
            e.bias_off = (int)j.bias_off;
            e.wg_start = start;
            e.ks_log = ks_log(j.a_tiles * j.b_tiles);
            e.tile0 = 0;
            if (j.side_kind) {  // the side block: a second weight block unpacked from the same workgroups' partials
                JobRed& f = r->jobs[nred++];
                f.a_tiles = j.side_kind == 1 ? 1 : j.a_tiles;
                f.b_tiles = j.side_kind == 1 ? j.b_tiles : j.side_tiles;
                f.po = j.side_kind == 1 ? 1 : j.po;
                f.pi = j.side_kind == 1 ? j.pi : j.side_tiles;
                f.r_lo = j.s_r_lo;
                f.r_hi = j.s_r_hi;
                f.w_off = (int)j.s_w_off;
                f.w_ld = j.s_w_ld;
                f.col_kind = j.s_col_kind;
                f.col_base = j.s_col_base;
                f.col_count = j.s_col_count;
                f.bias_off = (int)j.s_bias_off;
                f.wg_start = start;
                f.ks_log = ks_log(f.a_tiles * f.b_tiles) | ((j.side_kind == 1 ? j.a_tiles : 0) << 8);
                f.tile0 = j.a_tiles * j.b_tiles;
            }
        }
        start += (int)ks[q];
    }
    w.total_wgs = start;
    int tiles = 1;
    for (int q = 0; q < w.njobs; ++q) {
        const NhJob& j = p->jobs[q];
        const int tq = j.a_tiles * j.b_tiles + (j.side_kind == 1 ? j.b_tiles : (j.side_kind == 2 ? j.a_tiles * j.side_tiles : 0));
        if (tq > tiles) tiles = tq;
    }
    w.part_bias = tiles * 1024;
    w.part_stride = w.part_bias + NH_PART_EXTRA;
    if (r) {
        r->njobs = nred;
        r->total_wgs = w.total_wgs;
        r->part_bias = w.part_bias;
        r->part_stride = w.part_stride;
        for (int q = 0; q < 4 * NH16_KRX_EXT; ++q) r->xslot[q] = (signed char)p->xyz_slot_col[q];
        for (int q = 0; q < 4 * NH16_KRD_EXT; ++q) r->dslot[q] = (signed char)p->dir_slot_col[q];
    }
}

template <class MD>
int launch_wgrad(const WgradArgs& w, const char* name, nerfhip_stream_t stream) {
#ifndef NERFHIP_EMU
    hipError_t e = hipFuncSetAttribute((const void*)k_wgrad<MD>, hipFuncAttributeMaxDynamicSharedMemorySize, MD::LDS_BYTES);
    if (e != hipSuccess) {
        nh_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", MD::LDS_BYTES, hipGetErrorString(e));
        return NERFHIP_ERR_LAUNCH;
    }
#endif
    NH_LAUNCH_NAMED(name, (k_wgrad<MD>), w.total_wgs, 64 * MD::NWV, MD::LDS_BYTES, stream, w);
    return nh_launch_status("wgrad");
}

}  // namespace

int64_t nh_wgrad_partial_floats(nerfhip_plan* p, int64_t nt) {
    WgradArgs w;
    wgrad_schedule(p, nt > 0 ? nt : 1, w, nullptr);
    return (int64_t)w.total_wgs * w.part_stride;
}

int nh_wgrad(nerfhip_plan* p, int64_t nt, const float* stash, const float* grad, float* partial, float* g_params,
             const unsigned* gscale, const NhCompact* cx, nerfhip_stream_t stream) {
    WgradArgs w;
    ReduceArgs red;
    memset(&w, 0, sizeof(w));
    memset(&red, 0, sizeof(red));
    wgrad_schedule(p, nt, w, &red);
    w.stash = stash;
    w.grad = grad;
    w.partial = partial;
    red.partial = partial;
    red.g_params = g_params;
    red.gscale = gscale;
    red.w_unscale = 1.0f;  // (the stash rows are plain fp32 values in every precision)
    // (one reduce record per job AND per side block: nerfhip_plan_create counts both against NH_MAX_JOBS)
    NH_REQUIRE(red.njobs <= NH_JOBS_DEV, "wgrad: %d reduce records exceed the table of %d", red.njobs, NH_JOBS_DEV);
    w.nt = nt;
    w.clk = nh_prof_clock_slot(NH_CLK_WGRAD);
    w.cidx = (cx && !cx->stash_in_list_order) ? cx->idx : nullptr;
    w.cstats = cx ? cx->stats : nullptr;
    for (int q = 0; q < w.njobs; ++q) {
        const NhJob& j = p->jobs[q];
        NH_REQUIRE(j.b_row0 == 0 && 32 * j.a_tiles == j.a_region_rows && 32 * j.b_tiles == j.b_region_rows &&
                       32 * (j.a_region_rows + j.b_region_rows + (j.side_kind ? j.side_rows : 0)) <=
                           (p->wgrad_waves == 4 ? WModeNarrow::STAGE : WModeWide::STAGE) &&
                       j.a_tiles * j.b_tiles <= 64 && j.wo * j.wi <= p->wgrad_waves &&
                       j.wo * j.po == j.a_tiles && j.wi * j.pi == j.b_tiles,
                   "wgrad: job %d does not tile its regions exactly (%d x %d tiles, %d x %d waves, %d x %d patches)", q,
                   j.a_tiles, j.b_tiles, j.wo, j.wi, j.po, j.pi);
    }
    int rc = NERFHIP_OK;
    // (profile name: a level-4 plan leaves this kernel the thin blocks only -- a different amount of work under the same symbol)
    const char* const name = p->bjobs.empty() ? "k_wgrad<MD>" : "k_wgrad<MD>[thin blocks]";
    if (cx) {
        // (a gathered row's byte offset is a 32-bit buffer offset: nh_mlp_backward compacts only launches whose regions fit)
        const char* const name_cx = p->bjobs.empty() ? "k_wgrad<MD, compacted>" : "k_wgrad<MD, compacted>[thin blocks]";
        rc = p->wgrad_waves == 4 ? launch_wgrad<WModeNarrowCx>(w, name_cx, stream) : launch_wgrad<WModeWideCx>(w, name_cx, stream);
    } else if (p->wgrad_waves == 4)
        rc = launch_wgrad<WModeNarrow>(w, name, stream);
    else
        rc = launch_wgrad<WModeWide>(w, name, stream);
    if (rc) return rc;
    NH_LAUNCH(k_wgrad_reduce, red.njobs * 256, 256, 0, stream, red);
    return nh_launch_status("wgrad_reduce");
}
