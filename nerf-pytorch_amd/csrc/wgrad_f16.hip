// wgrad_f16.hip -- the large weight-gradient GEMMs of NERFHIP_PRECISION_F16X3_TRAIN plans on the fp16 MFMAs (the fp32
// original is wgrad.hip; the reference has no such function: it is autograd of nerf/models.py:233-258, train_nerf.py:259).
//
//     dW[r][c] = sum over samples s of A[r][s] * B[c][s],    A = a d(pre-activation) image, B = an activation image of the stash,
//
// both fp32, sample-major ([32-sample tile][sample][rows]).  v_mfma_f32_32x32x16_f16 wants a lane's 8 consecutive k (= samples)
// of ONE row in 16 contiguous bytes, and each fp32 value split into two IEEE fp16 pieces (hi = f16(v), lo = f16(v - hi);
// A.B ~ Ah.Bh + Ah.Bl + Al.Bh, fp32 accumulation: the arithmetic of mlp_f16w.hip).  So a workgroup (8 waves) walks
// over its range of samples in steps of 16 (one k-block), two phases per step:
//   convert : the step's A and B blocks (contiguous in HBM, brought to LDS by LDS-DMA into one of two fp32 stages: the copies of
//             the next two steps are in flight while this one is converted and multiplied) are read row-wise -- thread t owns row t mod rows: 8 samples of it per item, strided dword reads, conflict-free across
//             the wave --, scaled, split, and written back k-minor as MFMA operand blocks ([32-row tile][k-block][lane] x 16 B); the same
//             thread keeps the running row sum of A (the bias gradient);
//   multiply: wave (wo, wi) of the 4 x 2 grid (8 waves, two per SIMD) owns an eighth of the block (2 x 4 accumulator tiles = 128
//             AGPRs for a 256 x 256 block): 4 + 8 operand reads and 24 MFMAs per step.
// Split-K over the workgroups of a block; the partials (accumulator tiles as [tile][16 registers][64 lanes], then the 256 threads'
// bias sums) are summed in a fixed order by k_wgrad_reduce_f16x3 -- bit-reproducible, no atomics -- and scattered into the
// reference parameter layout.
//
// fp16's range: the values of a block's A region (d(pre-activation): 1e-3 ... 1e-12, and whatever the transposed layers amplify) and of
// its B region are multiplied, BEFORE they are split, by the power of two that brings the region's bound -- the launch that wrote it
// recorded 256 + log2 of it, WgBArgs::amax / bmax -- to 2^14; the reduction divides the two out again (exact).
#include <vector>

#include "nh_device.h"
#include "nh_diag.h"
#include "nh_mlp.h"

namespace {

constexpr int NHW_MAX_JOBS = 40;
#ifndef NHW_WAVES  // (A/B builds only) waves per workgroup: 4 (one per SIMD, 4 x 4 accumulator tiles each) or 8 (two per SIMD, 2 x 4:
                   // measured 10 % faster -- the two waves of a SIMD hide each other's LDS latencies inside a phase)
#define NHW_WAVES 8
#endif
constexpr int NHW_THREADS = 64 * NHW_WAVES;
#ifndef NHW_STAGES  // (A/B builds only) fp32 stages in LDS = 16-sample steps whose copies are in flight; 2 measured 4 % faster than 3
#define NHW_STAGES 2
#endif

struct WgBJob {
    int64_t a_off, b_off;  // float offsets of the two regions inside the grad scratch / the stash
    int wg0, nwg;          // this block's workgroups [wg0, wg0 + nwg)
    int r_hi, col_count, w_ld;
    int64_t w_off, bias_off;
    int a_idx, b_idx;      // slots of the two regions among the recorded maxima
    int64_t s_w_off;       // the thin block riding on this one (launches with SA or SB rows): its weight tensor ...
    int s_w_ld;            // ... and that tensor's column count
};
struct WgBArgs {
    const float* stash;
    const float* grad;
    float* partial;
    float* g_params;
    int64_t nt;
    int njobs, part_stride;  // floats per workgroup partial: AR * BR accumulators + 256 bias sums
    const unsigned* amax;    // fp16: region maxima recorded by the data-gradient launch that wrote `grad` (bit patterns), or NULL
    const unsigned* bmax;    // fp16: ... by the forward launch that wrote `stash`, or NULL
    // The thin blocks riding on this launch's blocks (SA / SB launches; one region for all of them): the region's float offset (SA:
    // inside the grad scratch, SB: inside the stash), its slot among the recorded maxima, and how the reduction unpacks a side block
    // -- SA: side rows s_r_lo .. s_r_hi - 1 are parameter rows 0 .. of WgBJob::s_w_off (columns = the host block's), their row sums
    // the bias at s_bias_off; SB: side row (encoding slot) k is column s_col_base + scol[k] of the host block's rows
    int64_t s_off, s_bias_off;
    int s_idx, s_r_lo, s_r_hi, s_col_base, s_col_count;
    signed char scol[64];    // SB launches: encoding slot -> reference column of the side block, or -1
    // compacted backward (CX launches; compact.hip): the sample list and its statistics.  The A regions (and an SA guest's) are then the
    // compacted d(pre-activation) images -- sample slot s is row s --, the B regions (and an SB guest's) are gathered: slot s is row
    // cidx[s] of the stash region
    const int* cidx;
    const int* cstats;
    WgBJob jobs[NHW_MAX_JOBS];
};

// a region's recorded word (256 + log2 of the bound on its magnitudes; 0: nothing recorded) -> the shift that brings the bound to 2^14
// (0: nothing recorded, 1: only all-zero samples -- nothing to scale)
NH_DEVICE int region_shift(unsigned word) { return word <= 1u ? 0 : 14 - ((int)word - 256); }
// a sum of products of values split at 2^sa and 2^sb, brought back: two exact multiplies (one factor 2^(-sa - sb) would leave the range
// of a normal float for tiny cotangents: sa reaches 110 + 14)
NH_DEVICE float unscale2(float v, int sa, int sb) { return (v * nh_pow2i(-sa)) * nh_pow2i(-sb); }

// SA / SB: rows of a thin block's region that rides on the launch's blocks -- SA (32): ONE more A tile against the block's B tiles
// (fc_alpha's row of POUT against H_{L-1}, next to fc_feat); SB (32 | 64): one or two more B tiles against the block's A tiles (the
// direction slots next to layers_dir's hidden columns, the xyz slots next to a skip layer's hidden columns).  The host block streams
// its regions anyway: the guest's own job would read them a second time (wgrad.hip does the same for the fp32 blocks).
template <int AR, int BR, int SA = 0, int SB = 0>
struct WShape {
    static_assert(SA == 0 || SB == 0, "one guest per block");
    static_assert(NHW_WAVES == 8, "the side tiles are dealt to a 4 x 2 wave grid");
    static constexpr int PO = AR / (16 * NHW_WAVES), PI = BR / 64, TA = AR / 32, TB = BR / 32;  // wave grid (NHW_WAVES / 2) x 2
    static constexpr int SR = SA + SB;                            // rows of the side region
    static constexpr int STAGE_A = AR * 64, STAGE_B = BR * 64, STAGE_S = SR * 64;  // bytes of one 16-sample step of a region, fp32
    static constexpr int STAGE = STAGE_A + STAGE_B + STAGE_S;     // (NHW_STAGES of them: the copies of the next steps land while step u is converted)
    static constexpr int OPER_A = AR * 32, OPER_B = BR * 32, OPER_S = SR * 32;  // bytes of its high (or low) operand blocks
    static constexpr int LDS_BYTES = NHW_STAGES * STAGE + 2 * (OPER_A + OPER_B + OPER_S);
    // side accumulator tiles: SB -- (A tile, side tile), TA x SB / 32; SA -- (the side tile, B tile), TB
    static constexpr int SIDE_TILES = SB ? TA * (SB / 32) : (SA ? TB : 0);
    static constexpr int PART = AR * BR + SIDE_TILES * 1024 + NHW_THREADS + (SA ? 64 : 0);
};

// rows -> operand blocks of one 16-sample k-block: item id = row + rows * q handles samples 8 q .. 8 q + 7 of `row` (q = the lane
// half that supplies them).  Two steps, each side of a scheduling fence: ALL of a thread's values are read from the stage first
// (left to itself the compiler waits for every pair of dwords right after asking for it: 16 exposed LDS latencies per step),
// then split and written back.  rows_load returns nothing; rows_store returns the sum of the values (the bias gradient's share).
template <int ROWS>
struct RowItems {
    static constexpr int N = (ROWS * 2 + NHW_THREADS - 1) / NHW_THREADS;  // items per thread
    float v[N][8];
};
template <int ROWS>
NH_DEVICE void rows_load(const float* stage, int tid, RowItems<ROWS>& r) {
#pragma unroll
    for (int it = 0; it < RowItems<ROWS>::N; ++it) {
        const int id = tid + NHW_THREADS * it;
        const bool on = ROWS * 2 % NHW_THREADS == 0 || id < ROWS * 2;
        const int row = id % ROWS, q = on ? id / ROWS : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[it][e] = on ? stage[(8 * q + e) * ROWS + row] : 0.0f;
    }
}
// (scale, fp16: the region's power of two -- applied to what is split, not to the bias sum)
template <int ROWS>
NH_DEVICE float rows_store(const RowItems<ROWS>& r, char* hi_blocks, char* lo_blocks, int tid, float scale) {
    float sum = 0.0f;
#pragma unroll
    for (int it = 0; it < RowItems<ROWS>::N; ++it) {
        const int id = tid + NHW_THREADS * it;
        if (ROWS * 2 % NHW_THREADS != 0 && id >= ROWS * 2) break;
        const int row = id % ROWS, q = id / ROWS;
        nh_f16x8 h8, l8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = r.v[it][e];
            sum += v;
#ifdef NHW_EXP_NO_CONVERT  // (NH_DIAG builds only, wrong results: what the split costs this kernel -- nothing, it waits for HBM)
            h8[e] = nh_to_f16(v);
            l8[e] = h8[e];
            continue;
#endif
            v *= scale;
            const nh_f16 hi = nh_to_f16(v);
            h8[e] = hi;
            l8[e] = nh_to_f16(v - nh_from_f16(hi));
        }
        // operand block of the 32-row tile row >> 5: lane (row & 31) + 32 q, 16 bytes per lane
        const int off = (((row >> 5) * 2 + q) * 32 + (row & 31)) * 16;
        *(nh_f16x8*)(hi_blocks + off) = h8;
        *(nh_f16x8*)(lo_blocks + off) = l8;
    }
    return sum;
}

// 1-KiB piece p of one 16-sample step of a gathered region (ROWS rows per sample): a piece holds N = 1024 / (4 ROWS) whole rows (or a
// part of one); rows[e]: the list entries of the samples it covers (gather_rows), each lane copies 16 bytes of its sample's row
template <int ROWS>
struct GatherN {
    static constexpr int RB = ROWS * 4, N = RB >= 1024 ? 1 : 1024 / RB;
};
template <int ROWS>
NH_DEVICE void gather_rows(unsigned* rows, const int* cidx, int s0, int p) {
    constexpr int RB = GatherN<ROWS>::RB, N = GatherN<ROWS>::N;
    const int first = RB >= 1024 ? s0 + (p * 1024) / RB : s0 + p * N;
#pragma unroll
    for (int e = 0; e < N; ++e) rows[e] = (unsigned)nh_uload_i32(cidx, first + e);
}
template <int ROWS>
NH_DEVICE void gather_piece_f16(const NhDmaSrc& src, const unsigned* rows, int p, int lane, unsigned lds_dst) {
    constexpr int RB = GatherN<ROWS>::RB, N = GatherN<ROWS>::N;
    if constexpr (RB >= 1024) {
        nh_dma16a(src, lane * 16, (int)(rows[0] * (unsigned)RB + ((unsigned)p * 1024u) % RB), lds_dst);
    } else {
        const int e = (lane * 16) / RB;
        unsigned row = rows[0];
#pragma unroll
        for (int k = 1; k < N; ++k) {
            row = e == k ? rows[k] : row;
#ifndef NERFHIP_EMU
            // (one v_cndmask per row: left alone, the compiler turns the chain into a dynamically indexed read of a private array,
            // i.e. 4 N bytes of scratch memory per lane and a scratch load per piece)
            asm volatile("" : "+v"(row));
#endif
        }
        nh_dma16a(src, (int)(row * (unsigned)RB + (unsigned)(lane * 16) % RB), 0, lds_dst);
    }
}

// CX: the compacted backward -- its own instantiations, the dense kernels' registers are not touched
template <int AR, int BR, int SA = 0, int SB = 0, bool CX = false>
NH_KERNEL void NH_LB(NHW_THREADS, NHW_WAVES / 4) k_wgrad_f16x3(WgBArgs a) {
    using S = WShape<AR, BR, SA, SB>;
    constexpr int PO = S::PO, PI = S::PI, SR = S::SR;
    NH_DYN_LDS(lds);
    char* const ah_blk = lds + NHW_STAGES * S::STAGE;
    char* const al_blk = ah_blk + S::OPER_A;
    char* const bh_blk = al_blk + S::OPER_A;
    char* const bl_blk = bh_blk + S::OPER_B;
    char* const sh_blk = bl_blk + S::OPER_B;
    char* const sl_blk = sh_blk + S::OPER_S;
    const int lane = nh_lane(), wave = nh_wave_in_block(), tid = (int)threadIdx.x;
    // this workgroup's block and its range of 16-sample steps (two per sample tile)
    int jq = 0;
    while (jq + 1 < a.njobs && (int)blockIdx.x >= a.jobs[jq + 1].wg0) ++jq;
    const WgBJob& jb = a.jobs[jq];
    const int64_t k = (int64_t)blockIdx.x - jb.wg0;
    // (compacted: the block's tiles are those of the sample list, however many the launch's cotangents left)
    const int64_t ntl = CX ? (int64_t)((nh_uload_i32(a.cstats, NH_CSTAT_ACTIVE) + 31) >> 5) : a.nt;
    const int64_t u0 = 2 * (ntl * k / jb.nwg), u1 = 2 * (ntl * (k + 1) / jb.nwg);
    const float* const a_reg = a.grad + jb.a_off;
    const float* const b_reg = a.stash + jb.b_off;
    const float* const s_reg = SR ? (SA ? a.grad : a.stash) + a.s_off : nullptr;
    const float a_scale = a.amax ? nh_pow2i(region_shift(a.amax[jb.a_idx])) : 1.0f;
    const float b_scale = a.bmax ? nh_pow2i(region_shift(a.bmax[jb.b_idx])) : 1.0f;
    const float s_scale = (SR && a.amax && a.bmax) ? nh_pow2i(region_shift((SA ? a.amax : a.bmax)[a.s_idx])) : 1.0f;
    const unsigned lds0 = nh_lds_addr((const float*)lds);
    constexpr int PIECES = (S::STAGE_A + S::STAGE_B) / (1024 * NHW_WAVES);  // copy instructions per wave and step ...
    constexpr int SPIECES = S::STAGE_S / 1024;                               // ... and one more for the first SPIECES waves
    static_assert(S::STAGE_A % (1024 * NHW_WAVES) == 0 && S::STAGE_B % (1024 * NHW_WAVES) == 0, "every wave issues the same number of pieces");
    static_assert(SPIECES <= NHW_WAVES, "at most one side piece per wave");
    // (CX) the list entries of the samples this wave's gathered pieces of a step cover -- scalar registers, fetched by preload(u) well
    // before issue(u) needs them: at the top of the iteration that ends in it (fetched at the pieces themselves, the wave waited for
    // the scalar cache two or three times per step: + 16 % on this HBM-bound kernel)
    constexpr int NPB = S::STAGE_B / 1024 / NHW_WAVES;  // gathered B pieces per wave and step
    static_assert(!CX || NPB * NHW_WAVES * 1024 == S::STAGE_B, "every wave gathers the same number of B pieces");
    unsigned rb[NPB * GatherN<BR>::N > 0 ? NPB * GatherN<BR>::N : 1], rs[GatherN<(SB ? SB : 32)>::N];
    auto preload = [&](int64_t u) {
        if (CX && a.cidx) {  // (a.cidx == NULL in a CX launch: the stash is in list order itself -- contiguous steps, the list's count)
#pragma unroll
            for (int j = 0; j < NPB; ++j) gather_rows<BR>(rb + j * GatherN<BR>::N, a.cidx, (int)u * 16, wave + j * NHW_WAVES);
            // (every wave: a conditional fill would park the array in scratch memory)
            if (SB) gather_rows<(SB ? SB : 32)>(rs, a.cidx, (int)u * 16, wave < SPIECES ? wave : 0);
        }
    };
    auto issue = [&](int64_t u) {  // one step of the regions -> stage u % NHW_STAGES: 1-KiB pieces dealt to the waves
        const NhDmaSrc da = nh_dma_src(a_reg + (size_t)u * 16 * AR, (unsigned)S::STAGE_A);
        const unsigned st = lds0 + (unsigned)((int)(u % NHW_STAGES) * S::STAGE);
        for (int p = wave; p < S::STAGE_A / 1024; p += NHW_WAVES) nh_dma16a(da, lane * 16, p * 1024, st + (unsigned)(p * 1024));
        if (CX && a.cidx) {  // the B rows of the step's 16 listed samples, out of the whole region (their list entries: preload(u))
            const NhDmaSrc db = nh_dma_src(b_reg, (unsigned)((size_t)a.nt * 32 * BR * 4));
#pragma unroll
            for (int j = 0; j < NPB; ++j)
                gather_piece_f16<BR>(db, rb + j * GatherN<BR>::N, wave + j * NHW_WAVES, lane, st + (unsigned)(S::STAGE_A + (wave + j * NHW_WAVES) * 1024));
        } else {
            const NhDmaSrc db = nh_dma_src(b_reg + (size_t)u * 16 * BR, (unsigned)S::STAGE_B);
            for (int p = wave; p < S::STAGE_B / 1024; p += NHW_WAVES) nh_dma16a(db, lane * 16, p * 1024, st + (unsigned)(S::STAGE_A + p * 1024));
        }
        if (SR) {
            if (CX && SB && a.cidx) {  // (an SB guest lives in the stash: gathered; an SA guest in the compacted gradient scratch: contiguous)
                const NhDmaSrc ds = nh_dma_src(s_reg, (unsigned)((size_t)a.nt * 32 * (SR ? SR : 1) * 4));
                if (wave < SPIECES) gather_piece_f16<(SB ? SB : 32)>(ds, rs, wave, lane, st + (unsigned)(S::STAGE_A + S::STAGE_B + wave * 1024));
            } else {
                const NhDmaSrc ds = nh_dma_src(s_reg + (size_t)u * 16 * SR, (unsigned)S::STAGE_S);
                if (wave < SPIECES) nh_dma16a(ds, lane * 16, wave * 1024, st + (unsigned)(S::STAGE_A + S::STAGE_B + wave * 1024));
            }
        }
    };
    f32x16 acc[PO][PI];
#pragma unroll
    for (int x = 0; x < PO; ++x)
#pragma unroll
        for (int y = 0; y < PI; ++y)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[x][y][c] = 0.0f;
    // side tiles of this wave (wave grid (wo, wi) = (wave >> 1, wave & 1)): SB -- its PO A tiles against side tile wi (one side tile:
    // the waves with wi == 0 only); SA -- the side tile against B tile tb0 + wo % PI (BR = 128: the waves with wo < PI only)
    constexpr int NSACC = SB ? PO : (SA ? 1 : 1);
    f32x16 sacc[NSACC];
#pragma unroll
    for (int x = 0; x < NSACC; ++x)
#pragma unroll
        for (int c = 0; c < 16; ++c) sacc[x][c] = 0.0f;
    const int wo = wave >> 1, wi = wave & 1;
    const bool side_on = SB ? (SB == 64 || wi == 0) : (SA ? wo < PI : false);
    const int ys = wo % PI;  // (SA) this wave's B tile among its PI
    float bias = 0.0f, sbias = 0.0f;
    const int ta0 = wo * PO, tb0 = wi * PI;
#pragma unroll
    for (int d = 0; d < NHW_STAGES; ++d)
        if (u0 + d < u1) {
            preload(u0 + d);
            issue(u0 + d);
        }
    for (int64_t u = u0; u < u1; ++u) {
        if (u + NHW_STAGES < u1) preload(u + NHW_STAGES);  // (consumed by the issue() at the end of this iteration)
        // the copy of step u is this wave's OLDEST outstanding one: wait for it alone while the later steps' copies stay in flight
#ifndef NHW_EXP_NO_WAIT  // (NH_DIAG builds only, wrong results)
        const int64_t behind = u1 - 1 - u < NHW_STAGES - 1 ? u1 - 1 - u : NHW_STAGES - 1;  // steps in flight behind step u
        if (behind >= 2) {
            if (wave < SPIECES) nh_wait_vmem_keep<2 * (PIECES + 1)>();
            else nh_wait_vmem_keep<2 * PIECES>();
        } else if (behind == 1) {
            if (wave < SPIECES) nh_wait_vmem_keep<PIECES + 1>();
            else nh_wait_vmem_keep<PIECES>();
        } else {
            nh_wait_vmem();
        }
#endif
        nh_block_sync();  // step u has landed for everyone; every wave is done multiplying the previous step's operand blocks
        const char* const stage = lds + (int)(u % NHW_STAGES) * S::STAGE;
        RowItems<AR> ra;
        RowItems<BR> rb;
        RowItems<(SR ? SR : 32)> rs;
        rows_load<AR>((const float*)stage, tid, ra);
        rows_load<BR>((const float*)(stage + S::STAGE_A), tid, rb);
        if (SR) rows_load<(SR ? SR : 32)>((const float*)(stage + S::STAGE_A + S::STAGE_B), tid, rs);
        nh_sched_fence();
        bias += rows_store<AR>(ra, ah_blk, al_blk, tid, a_scale);
        (void)rows_store<BR>(rb, bh_blk, bl_blk, tid, b_scale);
        if (SR) sbias += rows_store<(SR ? SR : 32)>(rs, sh_blk, sl_blk, tid, s_scale);
        nh_block_sync();  // operand blocks complete; this step's stage is free
        if (u + NHW_STAGES < u1) issue(u + NHW_STAGES);  // (into the stage just converted)
        nh_f16x8 ah[PO], al[PO], bh[PI], bl[PI], sh, sl;
#pragma unroll
        for (int x = 0; x < PO; ++x) {
            ah[x] = *(const nh_f16x8*)(ah_blk + (ta0 + x) * 1024 + lane * 16);
            al[x] = *(const nh_f16x8*)(al_blk + (ta0 + x) * 1024 + lane * 16);
        }
#pragma unroll
        for (int y = 0; y < PI; ++y) {
            bh[y] = *(const nh_f16x8*)(bh_blk + (tb0 + y) * 1024 + lane * 16);
            bl[y] = *(const nh_f16x8*)(bl_blk + (tb0 + y) * 1024 + lane * 16);
        }
        if (SR) {  // (SB == 64: side tile wi; else the one side tile)
            const int st = SB == 64 ? wi : 0;
            sh = *(const nh_f16x8*)(sh_blk + st * 1024 + lane * 16);
            sl = *(const nh_f16x8*)(sl_blk + st * 1024 + lane * 16);
        }
        nh_sched_fence();
#ifdef NHW_EXP_ONE_MFMA  // (NH_DIAG builds only, wrong results: one of the three sweeps -- what the multiplies cost this kernel)
#pragma unroll
        for (int x = 0; x < PO; ++x)
#pragma unroll
            for (int y = 0; y < PI; ++y) acc[x][y] = nh_mfma_f16(ah[x], bh[y], acc[x][y]);
        continue;
#endif
        // (three sweeps over the accumulator tiles, the small terms first: no MFMA reads the accumulator its predecessor wrote)
#pragma unroll
        for (int x = 0; x < PO; ++x)
#pragma unroll
            for (int y = 0; y < PI; ++y) acc[x][y] = nh_mfma_f16(al[x], bh[y], acc[x][y]);
#pragma unroll
        for (int x = 0; x < PO; ++x)
#pragma unroll
            for (int y = 0; y < PI; ++y) acc[x][y] = nh_mfma_f16(ah[x], bl[y], acc[x][y]);
#pragma unroll
        for (int x = 0; x < PO; ++x)
#pragma unroll
            for (int y = 0; y < PI; ++y) acc[x][y] = nh_mfma_f16(ah[x], bh[y], acc[x][y]);
        if (SB) {
            if (side_on) {
#pragma unroll
                for (int x = 0; x < PO; ++x) sacc[x] = nh_mfma_f16(al[x], sh, sacc[x]);
#pragma unroll
                for (int x = 0; x < PO; ++x) sacc[x] = nh_mfma_f16(ah[x], sl, sacc[x]);
#pragma unroll
                for (int x = 0; x < PO; ++x) sacc[x] = nh_mfma_f16(ah[x], sh, sacc[x]);
            }
        } else if (SA) {
            if (side_on) {
#pragma unroll
                for (int y = 0; y < PI; ++y)
                    if (y == ys) {  // (wave-uniform)
                        sacc[0] = nh_mfma_f16(sl, bh[y], sacc[0]);
                        sacc[0] = nh_mfma_f16(sh, bl[y], sacc[0]);
                        sacc[0] = nh_mfma_f16(sh, bh[y], sacc[0]);
                    }
            }
        }
    }
    // the partial: accumulator tile (ta, tb) as [16 registers][64 lanes], the side tiles likewise, then the threads' bias sums
    float* const part = a.partial + (size_t)blockIdx.x * (size_t)a.part_stride;
#pragma unroll
    for (int x = 0; x < PO; ++x)
#pragma unroll
        for (int y = 0; y < PI; ++y)
#pragma unroll
            for (int c = 0; c < 16; ++c) part[((ta0 + x) * S::TB + (tb0 + y)) * 1024 + c * 64 + lane] = acc[x][y][c];
    if (SB) {  // side tile (ta, st) at index ta * (SB / 32) + st
        if (side_on) {
            const int st = SB == 64 ? wi : 0;
#pragma unroll
            for (int x = 0; x < PO; ++x)
#pragma unroll
                for (int c = 0; c < 16; ++c) part[AR * BR + ((ta0 + x) * (SB / 32) + st) * 1024 + c * 64 + lane] = sacc[x][c];
        }
    } else if (SA) {  // side tile (0, tb) at index tb
        if (side_on) {
#pragma unroll
            for (int c = 0; c < 16; ++c) part[AR * BR + (tb0 + ys) * 1024 + c * 64 + lane] = sacc[0][c];
        }
    }
    part[AR * BR + S::SIDE_TILES * 1024 + tid] = bias;  // (thread t summed row t mod AR)
    if (SA && tid < 64) part[AR * BR + S::SIDE_TILES * 1024 + NHW_THREADS + tid] = sbias;  // (thread t < 2 SA summed side row t mod SA)
}

// sums the split-K partials of a block in workgroup order and scatters them into the reference parameter layout; element
// (tile (ta, tb), register c, lane l) is row 32 ta + (c & 3) + 8 (c >> 2) + 4 (l >> 5), column 32 tb + (l & 31)
template <int AR, int BR, int SA = 0, int SB = 0>
NH_KERNEL void k_wgrad_reduce_f16x3(WgBArgs a) {
    using S = WShape<AR, BR, SA, SB>;
    constexpr int TB = BR / 32, E = AR * BR, SE = S::SIDE_TILES * 1024, NALL = E + AR + SE + SA, GX = (NALL + 255) / 256;  // GX workgroups per block
    const int jq = (int)blockIdx.x / GX;
    const WgBJob& jb = a.jobs[jq];
    const int e = ((int)blockIdx.x % GX) * 256 + (int)threadIdx.x;
    const bool scaled = a.amax && a.bmax;
    // eight interleaved running sums (a fixed order) keep eight loads in flight per lane: one chain of ~100 dependent loads made
    // this kernel 0.7 ms per step
    auto sum_partials = [&](int64_t off) {
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float* const src = a.partial + (size_t)jb.wg0 * (size_t)a.part_stride + off;
        int k = 0;
        for (; k + 8 <= jb.nwg; k += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) s8[q] += src[(size_t)(k + q) * (size_t)a.part_stride];
        }
        for (int q = 0; k < jb.nwg; ++k, ++q) s8[q] += src[(size_t)k * (size_t)a.part_stride];
        return ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    };
    if (e < E) {
        const int tile = e >> 10, c = (e >> 6) & 15, l = e & 63;
        const int row = 32 * (tile / TB) + (c & 3) + 8 * (c >> 2) + 4 * (l >> 5), col = 32 * (tile % TB) + (l & 31);
        if (row < jb.r_hi && col < jb.col_count) {
            // (the powers of two the kernel split this block's regions at: exact to divide out)
            const float v = sum_partials(e);
            a.g_params[jb.w_off + (int64_t)row * jb.w_ld + col] = scaled ? unscale2(v, region_shift(a.amax[jb.a_idx]), region_shift(a.bmax[jb.b_idx])) : v;
        }
    } else if (e < E + AR) {  // bias: thread t of every workgroup summed row t mod AR
        const int row = e - E;
        if (row < jb.r_hi && jb.bias_off >= 0) {
            // (the same eight interleaved sums: as ONE chain this was up to 1024 dependent loads -- the kernel's long pole)
            float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* const src = a.partial + (size_t)jb.wg0 * (size_t)a.part_stride + E + SE + row;
            int k = 0;
            for (; k + 8 <= jb.nwg; k += 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int t = 0; t < NHW_THREADS; t += AR) s8[q] += src[(size_t)(k + q) * (size_t)a.part_stride + t];
            }
            for (int q = 0; k < jb.nwg; ++k, ++q)
#pragma unroll
                for (int t = 0; t < NHW_THREADS; t += AR) s8[q] += src[(size_t)k * (size_t)a.part_stride + t];
            a.g_params[jb.bias_off + row] = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));  // (sums of plain values)
        }
    } else if (e < E + AR + SE) {  // the side block
        const int se = e - E - AR, tile = se >> 10, c = (se >> 6) & 15, l = se & 63;
        const int r32 = (c & 3) + 8 * (c >> 2) + 4 * (l >> 5);
        if (SB) {  // tile (ta, st): row of the host block, encoding slot 32 st + (l & 31)
            const int row = 32 * (tile / (SB / 32)) + r32, slot = 32 * (tile % (SB / 32)) + (l & 31);
            const int col = (int)a.scol[slot];
            if (row < jb.r_hi && col >= 0 && col < a.s_col_count) {
                const float v = sum_partials(E + se);
                a.g_params[jb.s_w_off + (int64_t)row * jb.s_w_ld + a.s_col_base + col] =
                    scaled ? unscale2(v, region_shift(a.amax[jb.a_idx]), region_shift(a.bmax[a.s_idx])) : v;
            }
        } else {  // tile (0, tb): side row r32, column of the host block
            const int col = 32 * tile + (l & 31);
            if (r32 >= a.s_r_lo && r32 < a.s_r_hi && col < jb.col_count) {
                const float v = sum_partials(E + se);
                a.g_params[jb.s_w_off + (int64_t)(r32 - a.s_r_lo) * jb.s_w_ld + col] =
                    scaled ? unscale2(v, region_shift(a.amax[a.s_idx]), region_shift(a.bmax[jb.b_idx])) : v;
            }
        }
    } else if (SA && e < NALL) {  // the side block's bias: threads t and t + SA of every workgroup summed side row t
        const int r = e - E - AR - SE;
        if (r >= a.s_r_lo && r < a.s_r_hi && a.s_bias_off >= 0)
            a.g_params[a.s_bias_off + (r - a.s_r_lo)] = sum_partials(E + SE + NHW_THREADS + r) + sum_partials(E + SE + NHW_THREADS + SA + r);
    }
}

template <class K>
int w_lds_limit(K kern, int bytes) {
#ifndef NERFHIP_EMU
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        nh_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", bytes, hipGetErrorString(e));
        return NERFHIP_ERR_LAUNCH;
    }
#else
    (void)kern;
    (void)bytes;
#endif
    return NERFHIP_OK;
}

// workgroups per launch, dealt evenly to its blocks: whole rounds of one workgroup per CU.  Blocks of one shape -- rows of the A
// region, rows of a guest's side region -- share a launch (the kernel is a template over them): the full-height blocks without a
// guest get NHW_WGS workgroups between them, every other block NHW_WGS_HALF of its own (a launch with fewer workgroups than CUs
// leaves the rest of the chip idle for as long as one of them runs: the half-height block once got 45).
#ifndef NHW_WGS  // (A/B builds only)
#define NHW_WGS 768
#endif
#ifndef NHW_WGS_HALF
#define NHW_WGS_HALF 256
#endif

struct LaunchB {
    int ar, sa, sb;  // rows of the A region; rows of an A-side / a B-side guest region
    WgBArgs w;
};

// fills the job tables of the launches (one per block shape that occurs); returns the partial floats they need between them
int64_t schedule(nerfhip_plan* p, int64_t nt, std::vector<LaunchB>* out) {
    const int W = p->W;
    std::vector<LaunchB> L;
    std::vector<char> side_set;  // (per launch: a guest region has been noted)
    for (const NhJobB& j : p->bjobs) {
        const int sa = j.side_kind == 1 ? j.side_rows : 0, sb = j.side_kind == 2 ? j.side_rows : 0;
        size_t q = 0;
        while (q < L.size() && !(L[q].ar == j.a_rows && L[q].sa == sa && L[q].sb == sb)) ++q;
        if (q == L.size()) {
            LaunchB l;
            memset(&l, 0, sizeof(l));
            l.ar = j.a_rows, l.sa = sa, l.sb = sb;
            l.w.s_bias_off = -1;
            for (int k = 0; k < 64; ++k) l.w.scol[k] = -1;
            L.push_back(l);
            side_set.push_back(0);
        }
        WgBArgs& w = L[q].w;
        if (w.njobs >= NHW_MAX_JOBS) {
            nh_set_error("wgrad_f16: more than %d blocks of one shape in a launch", NHW_MAX_JOBS);
            return -1;
        }
        WgBJob& d = w.jobs[w.njobs++];
        d.a_off = 32 * nt * j.a_row_prefix;
        d.b_off = 32 * nt * j.b_row_prefix;
        d.r_hi = j.r_hi;
        d.col_count = j.col_count;
        d.w_ld = j.w_ld;
        d.w_off = j.w_off;
        d.bias_off = j.bias_off;
        d.a_idx = j.a_idx;
        d.b_idx = j.b_idx;
        d.s_w_off = j.s_w_off;
        d.s_w_ld = j.s_w_ld;
        if (j.side_kind) {
            // ONE side region per launch (WgBArgs carries one): plan.cpp attaches the guests of a shape to the same region -- the xyz
            // slots to every skip layer, the direction slots to layers_dir, POUT to fc_feat.  A second region of the same shape would
            // silently take the first one's place, so it is refused here
            const int64_t s_off = 32 * nt * j.side_row_prefix;
            const bool first = !side_set[q];
            if (!first && (w.s_off != s_off || w.s_idx != j.side_idx || w.s_r_lo != j.s_r_lo || w.s_r_hi != j.s_r_hi ||
                           w.s_col_base != j.s_col_base || w.s_col_count != j.s_col_count ||
                           (j.s_bias_off >= 0 && w.s_bias_off >= 0 && w.s_bias_off != j.s_bias_off))) {
                nh_set_error("wgrad_f16: two guest regions of one shape in a launch (rows %d, side rows %d / %d)", j.a_rows, sa, sb);
                return -1;
            }
            side_set[q] = true;
            w.s_off = s_off;
            w.s_idx = j.side_idx;
            w.s_r_lo = j.s_r_lo;
            w.s_r_hi = j.s_r_hi;
            w.s_col_base = j.s_col_base;
            w.s_col_count = j.s_col_count;
            if (j.s_bias_off >= 0) w.s_bias_off = j.s_bias_off;
            const int* map = j.s_col_kind == 1 ? p->xyz_slot_col : p->dir_slot_col;
            if (j.side_kind == 2)
                for (int k = 0; k < j.side_rows && k < 64; ++k) w.scol[k] = (signed char)map[k];
        }
    }
    int64_t floats = 0;
    for (LaunchB& l : L) {
        WgBArgs& w = l.w;
        const bool plain_full = l.ar == W && !l.sa && !l.sb;
        int64_t nwg = (plain_full ? NHW_WGS / w.njobs : NHW_WGS_HALF);
        if (nwg < 1) nwg = 1;
        if (nwg > nt) nwg = nt > 0 ? nt : 1;
        int wg = 0;
        for (int q = 0; q < w.njobs; ++q) {
            w.jobs[q].wg0 = wg;
            w.jobs[q].nwg = (int)nwg;
            wg += (int)nwg;
        }
        const int side_tiles = l.sb ? (l.ar / 32) * (l.sb / 32) : (l.sa ? W / 32 : 0);
        w.part_stride = l.ar * W + side_tiles * 1024 + NHW_THREADS + (l.sa ? 64 : 0);
        w.nt = nt;
        floats += (int64_t)wg * w.part_stride;
    }
    if (out) *out = L;
    return floats;
}

template <int AR, int BR, int SA, int SB>
int launch(WgBArgs& w, nerfhip_stream_t stream) {
    if (w.njobs == 0) return NERFHIP_OK;
    using S = WShape<AR, BR, SA, SB>;
    static_assert(S::PART == AR * BR + S::SIDE_TILES * 1024 + NHW_THREADS + (SA ? 64 : 0), "schedule() sizes the partials");
    const int wgs = w.jobs[w.njobs - 1].wg0 + w.jobs[w.njobs - 1].nwg;
    int rc = NERFHIP_OK;
    if (w.cstats) {
        rc = w_lds_limit(k_wgrad_f16x3<AR, BR, SA, SB, true>, S::LDS_BYTES);
        if (rc) return rc;
        NH_LAUNCH_NAMED(AR == BR ? (SA || SB ? "k_wgrad_f16x3<full+side, compacted>" : "k_wgrad_f16x3<full, compacted>")
                                 : (SA || SB ? "k_wgrad_f16x3<half+side, compacted>" : "k_wgrad_f16x3<half, compacted>"),
                        (k_wgrad_f16x3<AR, BR, SA, SB, true>), wgs, NHW_THREADS, (S::LDS_BYTES), stream, w);
    } else {
        rc = w_lds_limit(k_wgrad_f16x3<AR, BR, SA, SB>, S::LDS_BYTES);
        if (rc) return rc;
        NH_LAUNCH_NAMED(AR == BR ? (SA || SB ? "k_wgrad_f16x3<full+side>" : "k_wgrad_f16x3<full>")
                                 : (SA || SB ? "k_wgrad_f16x3<half+side>" : "k_wgrad_f16x3<half>"),
                        (k_wgrad_f16x3<AR, BR, SA, SB>), wgs, NHW_THREADS, (S::LDS_BYTES), stream, w);
    }
    rc = nh_launch_status("wgrad_f16x3");
    if (rc) return rc;
    constexpr int NALL = AR * BR + AR + S::SIDE_TILES * 1024 + SA;
    NH_LAUNCH_NAMED("k_wgrad_f16_reduce", (k_wgrad_reduce_f16x3<AR, BR, SA, SB>), ((NALL + 255) / 256) * w.njobs, 256, 0, stream, w);
    return nh_launch_status("wgrad_f16_reduce");
}

}  // namespace

int64_t nh_wgrad_x3_partial_floats(nerfhip_plan* p, int64_t nt) {
    if (p->bjobs.empty()) return 0;
    return schedule(p, nt > 0 ? nt : 1, nullptr);  // (-1: a plan the schedule refuses; nh_wgrad_f16 reports it)
}

int nh_wgrad_f16(nerfhip_plan* p, int64_t nt, const float* stash, const float* grad, float* partial, float* g_params,
                 const unsigned* amax, const unsigned* bmax, const NhCompact* cx, nerfhip_stream_t stream) {
    if (p->bjobs.empty()) return NERFHIP_OK;
    NH_REQUIRE((int)p->bjobs.size() <= NHW_MAX_JOBS, "wgrad_f16: too many blocks");
    std::vector<LaunchB> L;
    if (schedule(p, nt, &L) < 0) return NERFHIP_ERR_UNSUPPORTED;
    int rc = NERFHIP_OK;
    int64_t off = 0;  // (a launch's partials follow its predecessor's)
    for (LaunchB& l : L) {
        WgBArgs& w = l.w;
        w.stash = stash;
        w.grad = grad;
        w.g_params = g_params;
        w.amax = amax;
        w.bmax = bmax;
        w.cidx = (cx && !cx->stash_in_list_order) ? cx->idx : nullptr;
        w.cstats = cx ? cx->stats : nullptr;
        w.partial = partial + off;
        off += (int64_t)(w.jobs[w.njobs - 1].wg0 + w.jobs[w.njobs - 1].nwg) * w.part_stride;
        const int W = p->W;
        if (W == 256 && l.ar == 256 && !l.sa && !l.sb) rc = launch<256, 256, 0, 0>(w, stream);
        else if (W == 256 && l.ar == 256 && l.sb == 64) rc = launch<256, 256, 0, 64>(w, stream);
        else if (W == 256 && l.ar == 256 && l.sa == 32) rc = launch<256, 256, 32, 0>(w, stream);
        else if (W == 256 && l.ar == 128 && !l.sa && !l.sb) rc = launch<128, 256, 0, 0>(w, stream);
        else if (W == 256 && l.ar == 128 && l.sb == 32) rc = launch<128, 256, 0, 32>(w, stream);
        else if (W == 128 && l.ar == 128 && !l.sa && !l.sb) rc = launch<128, 128, 0, 0>(w, stream);
        else if (W == 128 && l.ar == 128 && l.sb == 64) rc = launch<128, 128, 0, 64>(w, stream);
        else if (W == 128 && l.ar == 128 && l.sa == 32) rc = launch<128, 128, 32, 0>(w, stream);
        else {
            nh_set_error("wgrad_f16: no kernel for kernel width %d, %d-row blocks with side rows %d / %d", W, l.ar, l.sa, l.sb);
            return NERFHIP_ERR_UNSUPPORTED;
        }
        if (rc) return rc;
    }
    return rc;
}
