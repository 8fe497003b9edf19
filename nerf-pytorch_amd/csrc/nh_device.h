// nh_device.h -- thin device-side vocabulary used by every kernel in this library.
//
// The product build is HIP for gfx950 only (hipcc --offload-arch=gfx950).  The same kernel
// sources can also be compiled as plain C++ against tests/emu/nh_emu.h (-DNERFHIP_EMU): a
// fibre-based wavefront emulator that exists ONLY so that the CPU test-suite can execute the
// kernel index algebra without a GPU.  The emulator is test infrastructure; the product library
// (libnerfhip.so) never contains it and the Python package never loads it.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <string.h>

#ifdef NERFHIP_EMU
#include "nh_emu.h"
#else
#include <hip/hip_runtime.h>

#define NH_KERNEL __global__
#define NH_LB(threads, waves_per_simd) __launch_bounds__(threads, waves_per_simd)
#define NH_DEVICE __device__ __forceinline__
#define NH_MEMBER __device__ __forceinline__
#define NH_SHARED __shared__
#define NH_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

NH_DEVICE int nh_lane() { return (int)(threadIdx.x & 63u); }
// wave index inside the workgroup, as a provably wave-uniform (SGPR) value
NH_DEVICE int nh_wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
NH_DEVICE void nh_block_sync() { __syncthreads(); }

NH_DEVICE float nh_shfl(float v, int src) { return __shfl(v, src, 64); }
NH_DEVICE int nh_shfl_i(int v, int src) { return __shfl(v, src, 64); }
NH_DEVICE float nh_shfl_up(float v, int d) { return __shfl_up(v, d, 64); }
NH_DEVICE float nh_shfl_down(float v, int d) { return __shfl_down(v, d, 64); }
NH_DEVICE float nh_shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
NH_DEVICE int nh_shfl_xor_i(int v, int m) { return __shfl_xor(v, m, 64); }
NH_DEVICE double nh_shfl_d(double v, int src) { return __shfl(v, src, 64); }
NH_DEVICE double nh_shfl_up_d(double v, int d) { return __shfl_up(v, d, 64); }
NH_DEVICE double nh_shfl_down_d(double v, int d) { return __shfl_down(v, d, 64); }
NH_DEVICE double nh_shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }

// D = A(32x2) * B(2x32) + C, exact fp32 (k-ordered fmaf chain).  Lane l supplies A[l&31][l>>5] and
// B[l>>5][l&31]; D register c of lane l is D[(c&3) + 8*(c>>2) + 4*(l>>5)][l&31].
NH_DEVICE f32x16 nh_mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// D = A(16x4) * B(4x16) + C, exact fp32.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; D register c of lane l is
// D[4*(l>>4) + c][l&15].  32 cycles per instruction, 40 cycles dependent latency.
NH_DEVICE f32x4 nh_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// IEEE fp16 pieces of the f16x3 plans (mlp_f16w.hip, wgrad_f16.hip).  nh_to_f16: round-to-nearest-even incl. subnormal results
// (v_cvt_f16_f32; the kernels run with fp16 denormals on, HIP's default); nh_mfma_f16: D = A(32x16) * B(16x32) + C on
// v_mfma_f32_32x32x16_f16, fp32 accumulation; lane l supplies A[l&31][8*(l>>5) + e] and B[8*(l>>5) + e][l&31], e = 0..7; D registers
// as for nh_mfma32 -- the matrix pipe takes fp16 subnormal inputs at face value (measured on MI355X: scripts/probe/f16_mfma_probe.hip).
typedef _Float16 nh_f16;
typedef _Float16 nh_f16x8 __attribute__((ext_vector_type(8)));
NH_DEVICE nh_f16 nh_to_f16(float v) { return (nh_f16)v; }
NH_DEVICE float nh_from_f16(nh_f16 h) { return (float)h; }
NH_DEVICE f32x16 nh_mfma_f16(nh_f16x8 a, nh_f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_f16 (the two-waves-per-SIMD kernels of mlp_f16w.hip): D = A(16x32) * B(32x16) + C; lane l supplies
// A[l&15][8*(l>>4) + e] and B[8*(l>>4) + e][l&15], e = 0..7; D register c of lane l is D[4*(l>>4) + c][l&15] -- the C/D layout of
// nh_mfma16.  16 cycles per instruction and SIMD.
NH_DEVICE f32x4 nh_mfma_f16_16(nh_f16x8 a, nh_f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// eight pieces times a power of two (v_pk_mul_f16: exact unless a piece leaves fp16's range at the bottom)
NH_DEVICE nh_f16x8 nh_f16x8_scale(nh_f16x8 v, float pow2) { return v * (nh_f16)pow2; }

// ds_read_b64_tr_b16 (gfx950): the 16 lanes of a quarter wave each name 8 bytes (four 16-bit elements) in LDS; read as a
// 4 x 16 matrix in lane order (lane k holds row k >> 2, columns 4 (k & 3) .. + 3), lane i gets COLUMN i: element j of its result
// is element (i & 3) of lane 4 j + (i >> 2)'s bytes.  Turns a sample-major image into k-minor MFMA operands for free.
NH_DEVICE unsigned long long nh_lds_tr16(const char* lds_ptr) {
    typedef short nh_s4 __attribute__((__vector_size__(4 * sizeof(short))));
    const nh_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) nh_s4*)lds_ptr);
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    return u;
}
// sum of the two fp16 values packed in `pair` (+ c), in fp32: v_dot2c_f32_f16 against (1, 1)
NH_DEVICE float nh_pair_sum_f16(unsigned pair, float c) {
    typedef _Float16 nh_h2 __attribute__((ext_vector_type(2)));
    nh_h2 a, one = {(_Float16)1.0f, (_Float16)1.0f};
    __builtin_memcpy(&a, &pair, 4);
    return __builtin_amdgcn_fdot2(a, one, c, false);
}
// maximum of v over the wave, as a wave-uniform value: six v_max_u32 with DPP operands (row_shr 1 2 4 8, row_bcast 15 31) and a
// v_readlane -- no LDS traffic
NH_DEVICE unsigned nh_wave_max_u32(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false), v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false), v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false), v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false), v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false), v = v > t ? v : t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false), v = v > t ? v : t;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
NH_DEVICE float nh_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }  // v_med3_f32: clamp in one instruction
NH_DEVICE void nh_atomic_add(float* p, float v) { atomicAdd(p, v); }
NH_DEVICE void nh_atomic_max_u32(unsigned* p, unsigned v) { atomicMax(p, v); }
// Asynchronous global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): lane l's 16 bytes at `g` land at
// lds_wave_base + 16*l (the LDS destination is wave-uniform base + lane*16).  Completion: nh_wait_vmem() + barrier.
NH_DEVICE void nh_glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same copy through a buffer descriptor: source = desc.base + soff (wave-uniform, SGPR) + voff (per lane), all in
// bytes.  One piece costs two scalar adds + buffer_load_dwordx4 ... offen lds -- no VALU address arithmetic -- and
// bytes beyond `bytes` are bounds-checked away by the descriptor instead of faulting.
// The copy is issued through inline assembly ON PURPOSE: when the compiler sees an LDS-DMA it cannot prove that later
// ds_reads of the same LDS array do not alias its destination, and inserts s_waitcnt vmcnt(0) in front of the first
// ds_read that follows -- which exposes the whole copy latency in every chunk of a double-buffered pipeline (seen in
// the ISA of the 16x16x4 kernels).  The kernels order DMA completion themselves: nh_wait_vmem() + barrier before any
// wave reads the buffer.  (Untracked VMEM operations only make the compiler's own vmcnt waits more conservative.)
typedef int nh_i32x4 __attribute__((ext_vector_type(4)));
struct NhDmaSrc {
    nh_i32x4 r;  // raw buffer descriptor: base, stride 0, num_records = bytes, flags
};
NH_DEVICE NhDmaSrc nh_dma_src(const float* base, unsigned bytes) {
    NhDmaSrc s;
    const unsigned long long b = (unsigned long long)base;
    s.r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    s.r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xFFFFu));
    s.r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    s.r[3] = 0x00020000;
    return s;
}
NH_DEVICE void nh_dma16(const NhDmaSrc& s, int voff, int soff, float* lds_wave_base) {
    const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base;
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(__builtin_amdgcn_readfirstlane((int)m0v)), "v"(voff), "s"(s.r), "s"(__builtin_amdgcn_readfirstlane(soff))
        : "memory");  // M0 is a reserved register: the compiler never keeps a live value in it across statements
}
// the same copy with the LDS destination given as a byte address (nh_lds_addr, computed once per kernel): no
// generic -> LDS pointer cast (and its null check) per piece
NH_DEVICE unsigned nh_lds_addr(const float* lds_ptr) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_ptr;
}
NH_DEVICE void nh_dma16a(const NhDmaSrc& s, int voff, int soff, unsigned lds_wave_addr) {
    asm volatile(
        "s_mov_b32 m0, %0\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds"
        :
        : "s"(__builtin_amdgcn_readfirstlane((int)lds_wave_addr)), "v"(voff), "s"(s.r), "s"(__builtin_amdgcn_readfirstlane(soff))
        : "memory");
}
// p[i] for a wave-uniform address in memory that no launch on the device writes while this kernel runs (a sample list built by the
// launch before): read through the constant address space, i.e. s_load_dword -- a scalar register, counted by lgkmcnt, NOT by vmcnt:
// the kernels that call it keep LDS-DMA copies in flight whose completion they count themselves (nh_wait_vmem_keep), and a vector
// load the compiler tracks would make it wait for all of them at the load's first use.
NH_DEVICE int nh_uload_i32(const int* p, int i) {
    typedef const __attribute__((address_space(4))) int* nh_cptr;
    return ((nh_cptr)(unsigned long long)p)[i];
}
NH_DEVICE void nh_wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// ... until at most N of this wave's vector-memory operations are outstanding (the N newest: they complete in order)
template <int N>
NH_DEVICE void nh_wait_vmem_keep() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// nothing is scheduled across this point (pins "issue the prefetch BEFORE the MFMAs")
NH_DEVICE void nh_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
NH_DEVICE void nh_sincos(float x, float* s, float* c) { sincosf(x, s, c); }
NH_DEVICE unsigned long long nh_wall_clock() { return wall_clock64(); }  // constant 100 MHz
NH_DEVICE unsigned long long nh_core_clock() { return (unsigned long long)clock64(); }  // shader-clock cycles (s_memtime)
// Shader-clock probe of the MLP kernels (active only while nerfhip_profile_enable is on: `slot` is NULL otherwise).
// Thread 0 of a workgroup parks the counter address and its entry stamps (s_memtime: shader-clock cycles;
// s_memrealtime: the constant 100 MHz counter) in 24 bytes of LDS -- nothing stays live in registers across the kernel,
// not even the kernel-argument pointer -- and at exit adds the two deltas and a workgroup count to slot[0..2].
// 100 MHz * slot[0] / slot[1] is the clock the kernel's workgroups really ran at.
NH_DEVICE void nh_clk_begin(unsigned long long* slot, unsigned long long* lds3) {
    if (threadIdx.x == 0) {
        lds3[2] = (unsigned long long)slot;
        if (slot) {
            lds3[0] = nh_core_clock();
            lds3[1] = nh_wall_clock();
        }
    }
}
NH_DEVICE void nh_clk_end(const unsigned long long* lds3) {
    unsigned t = threadIdx.x;
    asm volatile("" : "+v"(t));  // (opaque: otherwise the exec mask of nh_clk_begin's test is kept in SGPRs across the kernel)
    if (t == 0) {
        unsigned long long* slot = (unsigned long long*)lds3[2];
        if (slot) {
            atomicAdd(slot, nh_core_clock() - lds3[0]);
            atomicAdd(slot + 1, nh_wall_clock() - lds3[1]);
            atomicAdd(slot + 2, 1ull);
        }
    }
}
#endif  // NERFHIP_EMU
#ifdef NERFHIP_EMU
static inline void nh_clk_begin(unsigned long long*, unsigned long long*) {}
static inline void nh_clk_end(const unsigned long long*) {}
#endif
constexpr int NH_CLK_LDS_BYTES = 32;
enum { NH_CLK_FWD = 0, NH_CLK_DGRAD = 1, NH_CLK_WGRAD = 2, NH_CLK_KERNELS = 3 };

// 16-byte store of a stash / d(pre-activation) row piece.  (A/B builds: -DNH_STASH_NT issues it non-temporal -- measured on
// MI355X and rejected: a row's 128-byte lines are completed by several instructions and rely on L2 to merge them; nt costs
// 0.4 % for the fp32 kernels and 44 % for the split-bf16 training kernels, profiles/r03_variant_ab.txt section 8.)
NH_DEVICE void nh_store4(float* dst, float x, float y, float z, float w) {
#if defined(NH_STASH_NT) && !defined(NERFHIP_EMU)
    typedef float nh_f4 __attribute__((ext_vector_type(4)));
    nh_f4 v = {x, y, z, w};
    __builtin_nontemporal_store(v, (nh_f4*)dst);
#else
    float4 v;
    v.x = x, v.y = y, v.z = z, v.w = w;
    *(float4*)dst = v;
#endif
}

// ---- helpers shared by both builds -------------------------------------------------------------------------------

// ReLU / mask helpers shaped to cost one or two VALU instructions each (every VALU instruction of a one-wave-per-SIMD
// kernel is taken from the matrix pipe, profiles/r01_mfma_issue_cost.txt):
//   nh_relu: v_max_i32 on the bit pattern (negative floats and -0 have the sign bit set -> +0; no NaN canonicalisation);
//   nh_pos_bit: 1 iff v > 0 for v >= 0 (v_min_u32 of the bit pattern with 1);
//   nh_gate: v if bit k of `word` is set, else +0 (v_bfe_i32 -> 0 / -1, v_and).
NH_DEVICE float nh_relu(float v) {
    int i;
    memcpy(&i, &v, 4);
    i = i > 0 ? i : 0;
    memcpy(&v, &i, 4);
    return v;
}
NH_DEVICE unsigned nh_pos_bit(float v) {
    unsigned u;
    memcpy(&u, &v, 4);
    return u < 1u ? u : 1u;
}
NH_DEVICE float nh_gate(float v, unsigned word, int k) {
    int m = ((int)(word << (31 - k))) >> 31;
    asm("" : "+v"(m));  // (keeps the two-instruction form: otherwise the AND is rewritten into and + compare + select)
    int i;
    memcpy(&i, &v, 4);
    i &= m;
    memcpy(&v, &i, 4);
    return v;
}

// Power-of-two scales of the fp16 data-gradient chain (exact to apply and to undo).  fp16's normal range is [2^-14, 2^16) and a
// value's low piece sits 2^-12 below it, so a value keeps both pieces exact down to 2^-2 and loses ABSOLUTE precision (2^-25)
// below: what is multiplied should sit well above 1, with head room for what the transposed layers amplify.
//   nh_pow2_to(bits, t): the power of two S with |x| * S in [2^t, 2^(t+1)) for x = the float with bit pattern `bits` (degenerate
//       x -- 0, subnormal, Inf / NaN -- gives S = 1); returned as its biased exponent (1 .. 253: S and 1 / S both normal).
//   per SAMPLE (k_mlp_dgrad_f16x3): the chain of a sample runs on d(raw output) * nh_pow2_to(max |d(raw output)| of the sample, 6)
//       -- every sample keeps fp32-like RELATIVE precision however small its cotangent --;
//   per LAUNCH (what the d(pre-activation) images carry, and the weight-gradient reduction divides out): max over the launch
//       (k_absmax_bits) to 2^8: nh_gscale_of / nh_gscale_inv.
NH_DEVICE int nh_pow2_to(unsigned bits, int t) {
    const int e = (int)((bits >> 23) & 255u);
    if (e == 0 || e == 255) return 127;
    const int se = 254 + t - e;
    return se < 1 ? 1 : (se > 253 ? 253 : se);
}
// 2^k as a float, k clamped to the normal range
NH_DEVICE float nh_pow2i(int k) {
    const unsigned u = (unsigned)((k < -126 ? -126 : (k > 127 ? 127 : k)) + 127) << 23;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// the shift k with |x| * 2^k in [2^t, 2^(t+1)) for x = the float with bit pattern `bits`; 0 for x = 0 / subnormal / Inf / NaN
NH_DEVICE int nh_shift_to(unsigned bits, int t) {
    const int e = (int)((bits >> 23) & 255u);
    return (e == 0 || e == 255) ? 0 : t + 127 - e;
}
NH_DEVICE float nh_pow2_float(int biased_exp) {
    const unsigned u = (unsigned)biased_exp << 23;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
constexpr int NH_GSCALE_LAUNCH_LOG2 = 8, NH_GSCALE_SAMPLE_LOG2 = 6;
NH_DEVICE float nh_gscale_of(unsigned maxbits) { return nh_pow2_float(nh_pow2_to(maxbits, NH_GSCALE_LAUNCH_LOG2)); }
NH_DEVICE float nh_gscale_inv(unsigned maxbits) { return nh_pow2_float(254 - nh_pow2_to(maxbits, NH_GSCALE_LAUNCH_LOG2)); }

// Row (feature) index held by accumulator register c (0..15) of MFMA tile t for lane-half h.
NH_DEVICE int nh_feat_of(int t, int c, int h) { return 32 * t + (c & 3) + 8 * (c >> 2) + 4 * h; }

// wave-wide sums (all 64 lanes end with the total)
NH_DEVICE float nh_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += nh_shfl_xor(v, m);
    return v;
}
NH_DEVICE double nh_wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += nh_shfl_xor_d(v, m);
    return v;
}

// Philox4x32-10 counter based RNG (Salmon et al.); production-mode random draws (see rng.hip).
struct nh_u4 {
    uint32_t x, y, z, w;
};
NH_DEVICE void nh_mulhilo(uint32_t a, uint32_t b, uint32_t* hi, uint32_t* lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}
NH_DEVICE nh_u4 nh_philox(uint64_t seed, uint64_t ctr_lo, uint32_t stream) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    nh_u4 c = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), stream, 0x9E3779B9u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        nh_mulhilo(0xD2511F53u, c.x, &hi0, &lo0);
        nh_mulhilo(0xCD9E8D57u, c.z, &hi1, &lo1);
        nh_u4 n = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// uniform in [0,1): top 24 bits, exactly like torch's float conversion
NH_DEVICE float nh_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// element e of stream `stream`: uniform
NH_DEVICE float nh_rand_uniform(uint64_t seed, uint32_t stream, uint64_t e) {
    nh_u4 r = nh_philox(seed, e, stream);
    return nh_u01(r.x);
}
// element e of stream `stream`: standard normal (Box-Muller on two independent words)
NH_DEVICE float nh_rand_normal(uint64_t seed, uint32_t stream, uint64_t e) {
    nh_u4 r = nh_philox(seed, e, stream);
    float u1 = 1.0f - nh_u01(r.x);  // (0,1]
    float u2 = nh_u01(r.y);
    float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    nh_sincos(6.283185307179586f * u2, &s, &c);
    return rad * c;
}
