// nh_emu.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal wavefront emulator: lets the CPU test-suite execute the library's HIP kernel sources
// (compiled as plain C++ with -DNERFHIP_EMU) so that index algebra, weight packing, LDS addressing
// and barrier placement can be checked in a container that has no GPU.  Every thread of a
// workgroup is a fibre (hand-rolled x86-64 context switch); wave-level operations (shuffles, MFMA)
// and __syncthreads() are rendezvous points.  Blocks run one after another.  Nothing here is part
// of the product: libnerfhip.so is built by hipcc for gfx950 and never sees this header.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <math.h>
#include <functional>

namespace emu {
struct Dim3 {
    unsigned x, y, z;
    Dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct Fiber {
    void* sp;
    char* stack;
    Dim3 tid;
    int lin, lane, wave;
    int xphase;
    bool done;
};
struct WaveState {
    int live, arrived;
    unsigned gen;
    uint64_t xa[2][64];
    uint64_t xb[2][64];
};
extern Fiber* cur;
extern Dim3 g_blockIdx, g_blockDim, g_gridDim;
extern char* g_dyn_smem;
WaveState& cur_wave();
void wave_barrier();
void block_barrier();
void launch(Dim3 grid, Dim3 block, size_t smem, const std::function<void()>& body);
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

#define NH_KERNEL
#define NH_LB(threads, waves_per_simd)
#define NH_DEVICE static inline
#define NH_MEMBER inline
#define NH_SHARED static
#define NH_DYN_LDS(name) char* name = emu::g_dyn_smem

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct float4 {
    float x, y, z, w;
};
struct float2 {
    float x, y;
};
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

NH_DEVICE int nh_lane() { return emu::cur->lane; }
NH_DEVICE int nh_wave_in_block() { return emu::cur->wave; }
NH_DEVICE void nh_block_sync() { emu::block_barrier(); }

template <class T>
static inline T nh_emu_xchg(T v, int src) {
    static_assert(sizeof(T) <= 8, "xchg");
    emu::WaveState& w = emu::cur_wave();
    int ph = emu::cur->xphase;
    emu::cur->xphase ^= 1;
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    w.xa[ph][emu::cur->lane] = raw;
    emu::wave_barrier();
    T out = v;
    if (src >= 0 && src < 64) memcpy(&out, &w.xa[ph][src], sizeof(T));
    return out;
}
NH_DEVICE float nh_shfl(float v, int src) { return nh_emu_xchg(v, src & 63); }
NH_DEVICE int nh_shfl_i(int v, int src) { return nh_emu_xchg(v, src & 63); }
NH_DEVICE float nh_shfl_up(float v, int d) { return nh_emu_xchg(v, emu::cur->lane - d); }
NH_DEVICE float nh_shfl_down(float v, int d) { return nh_emu_xchg(v, emu::cur->lane + d); }
NH_DEVICE float nh_shfl_xor(float v, int m) { return nh_emu_xchg(v, emu::cur->lane ^ m); }
NH_DEVICE int nh_shfl_xor_i(int v, int m) { return nh_emu_xchg(v, emu::cur->lane ^ m); }
NH_DEVICE double nh_shfl_d(double v, int src) { return nh_emu_xchg(v, src & 63); }
NH_DEVICE double nh_shfl_up_d(double v, int d) { return nh_emu_xchg(v, emu::cur->lane - d); }
NH_DEVICE double nh_shfl_down_d(double v, int d) { return nh_emu_xchg(v, emu::cur->lane + d); }
NH_DEVICE double nh_shfl_xor_d(double v, int m) { return nh_emu_xchg(v, emu::cur->lane ^ m); }

// v_mfma_f32_32x32x2_f32 semantics as documented for gfx950: lane l supplies A[l&31][l>>5] and
// B[l>>5][l&31]; result register c of lane l is D[(c&3)+8*(c>>2)+4*(l>>5)][l&31]; the sum over k
// is a k-ordered fmaf chain.
NH_DEVICE f32x16 nh_mfma32(float a, float b, f32x16 c) {
    emu::WaveState& w = emu::cur_wave();
    int ph = emu::cur->xphase;
    emu::cur->xphase ^= 1;
    int lane = emu::cur->lane;
    uint64_t ra = 0, rb = 0;
    memcpy(&ra, &a, 4);
    memcpy(&rb, &b, 4);
    w.xa[ph][lane] = ra;
    w.xb[ph][lane] = rb;
    emu::wave_barrier();
    int j = lane & 31, h = lane >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &w.xa[ph][i + 32 * k], 4);
            memcpy(&bv, &w.xb[ph][j + 32 * k], 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_16x16x4_f32: lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; result register c of lane l is
// D[4*(l>>4)+c][l&15]; the sum over k is a k-ordered fmaf chain.
NH_DEVICE f32x4 nh_mfma16(float a, float b, f32x4 c) {
    emu::WaveState& w = emu::cur_wave();
    int ph = emu::cur->xphase;
    emu::cur->xphase ^= 1;
    int lane = emu::cur->lane;
    uint64_t ra = 0, rb = 0;
    memcpy(&ra, &a, 4);
    memcpy(&rb, &b, 4);
    w.xa[ph][lane] = ra;
    w.xb[ph][lane] = rb;
    emu::wave_barrier();
    int j = lane & 15, g = lane >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, &w.xa[ph][i + 16 * k], 4);
            memcpy(&bv, &w.xb[ph][j + 16 * k], 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// IEEE fp16 pieces (mlp_f16w.hip, wgrad_f16.hip): software round-to-nearest-even conversion incl. subnormals;
// v_mfma_f32_32x32x16_f16 semantics -- lane l supplies A[l&31][8*(l>>5)+e] and B[8*(l>>5)+e][l&31]; D registers as for nh_mfma32;
// exact products, fp32 fmaf chain (the hardware's internal summation order is not part of what the tests pin down)
struct nh_f16 {
    uint16_t bits;
};
struct nh_f16x8 {
    nh_f16 v[8];
    nh_f16& operator[](int i) { return v[i]; }
    const nh_f16& operator[](int i) const { return v[i]; }
};
NH_DEVICE float nh_from_f16(nh_f16 h) {
    const uint32_t sign = (uint32_t)(h.bits & 0x8000u) << 16, ex = (h.bits >> 10) & 31u, man = h.bits & 1023u;
    float f;
    if (ex == 31u) {
        const uint32_t u = sign | 0x7f800000u | (man << 13);
        memcpy(&f, &u, 4);
        return f;
    }
    if (ex == 0u) {
        f = ldexpf((float)man, -24);
        return sign ? -f : f;
    }
    const uint32_t u = sign | ((ex + 112u) << 23) | (man << 13);
    memcpy(&f, &u, 4);
    return f;
}
NH_DEVICE nh_f16 nh_to_f16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return nh_f16{(uint16_t)(sign | 0x7e00u)};
    if (a >= 0x47800000u) return nh_f16{(uint16_t)(sign | 0x7c00u)};  // >= 65536: inf (65520..65536 handled by the rounding below)
    float af;
    memcpy(&af, &a, 4);
    if (a < 0x38800000u) {  // below 2^-14: a multiple of 2^-24, round-to-nearest-even (nearbyintf under the default mode)
        const float q = nearbyintf(ldexpf(af, 24));
        return nh_f16{(uint16_t)(sign | (uint16_t)q)};   // (q == 1024 is the smallest normal: the bit pattern carries over)
    }
    uint32_t r = a + 0xfffu + ((a >> 13) & 1u);  // round the 13 dropped bits to nearest even
    const uint32_t ex = (r >> 23) - 112u, man = (r >> 13) & 1023u;
    if (ex >= 31u) return nh_f16{(uint16_t)(sign | 0x7c00u)};
    return nh_f16{(uint16_t)(sign | (ex << 10) | man)};
}
NH_DEVICE nh_f16x8 nh_f16x8_scale(nh_f16x8 v, float pow2) {
    nh_f16x8 o;
    for (int e = 0; e < 8; ++e) o[e] = nh_to_f16(nh_from_f16(v[e]) * nh_from_f16(nh_to_f16(pow2)));
    return o;
}
NH_DEVICE f32x16 nh_mfma_f16(nh_f16x8 a, nh_f16x8 b, f32x16 c) {
    emu::WaveState& w = emu::cur_wave();
    const int lane = emu::cur->lane, j = lane & 31, h = lane >> 5;
    f32x16 d = c;
    for (int part = 0; part < 2; ++part) {
        int ph = emu::cur->xphase;
        emu::cur->xphase ^= 1;
        memcpy(&w.xa[ph][lane], &a.v[4 * part], 8);
        memcpy(&w.xb[ph][lane], &b.v[4 * part], 8);
        emu::wave_barrier();
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
            float acc = d[r];
            for (int hh = 0; hh < 2; ++hh)
                for (int e = 0; e < 4; ++e) {
                    nh_f16 av[4], bv[4];
                    memcpy(av, &w.xa[ph][i + 32 * hh], 8);
                    memcpy(bv, &w.xb[ph][j + 32 * hh], 8);
                    acc = fmaf(nh_from_f16(av[e]), nh_from_f16(bv[e]), acc);
                }
            d[r] = acc;
        }
    }
    return d;
}

// v_mfma_f32_16x16x32_f16 (mlp_f16w.hip): lane l supplies A[l&15][8*(l>>4)+e] and B[8*(l>>4)+e][l&15]; D register c of lane l is
// D[4*(l>>4)+c][l&15]; exact products, fp32 fmaf chain in k order
NH_DEVICE f32x4 nh_mfma_f16_16(nh_f16x8 a, nh_f16x8 b, f32x4 c) {
    emu::WaveState& w = emu::cur_wave();
    const int lane = emu::cur->lane, j = lane & 15, g = lane >> 4;
    f32x4 d = c;
    for (int part = 0; part < 2; ++part) {  // elements 4*part .. 4*part+3 of every lane: 8 bytes per exchange
        int ph = emu::cur->xphase;
        emu::cur->xphase ^= 1;
        memcpy(&w.xa[ph][lane], &a.v[4 * part], 8);
        memcpy(&w.xb[ph][lane], &b.v[4 * part], 8);
        emu::wave_barrier();
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r;
            float acc = d[r];
            for (int gg = 0; gg < 4; ++gg)
                for (int e = 0; e < 4; ++e) {
                    nh_f16 av[4], bv[4];
                    memcpy(av, &w.xa[ph][i + 16 * gg], 8);
                    memcpy(bv, &w.xb[ph][j + 16 * gg], 8);
                    acc = fmaf(nh_from_f16(av[e]), nh_from_f16(bv[e]), acc);
                }
            d[r] = acc;
        }
    }
    return d;
}

// ds_read_b64_tr_b16 semantics (nh_device.h): element j of lane i's result = element (i & 3) of lane 4 j + (i >> 2) of its quarter wave
NH_DEVICE unsigned long long nh_lds_tr16(const char* lds_ptr) {
    emu::WaveState& w = emu::cur_wave();
    const int lane = emu::cur->lane, base = lane & ~15, i = lane & 15;
    int ph = emu::cur->xphase;
    emu::cur->xphase ^= 1;
    unsigned long long mine;
    memcpy(&mine, lds_ptr, 8);
    w.xa[ph][lane] = mine;
    emu::wave_barrier();
    unsigned long long out = 0;
    for (int j = 0; j < 4; ++j) out |= ((w.xa[ph][base + 4 * j + (i >> 2)] >> (16 * (i & 3))) & 0xffffull) << (16 * j);
    return out;
}
NH_DEVICE float nh_pair_sum_f16(unsigned pair, float c) {
    return (c + nh_from_f16(nh_f16{(uint16_t)(pair & 0xffffu)})) + nh_from_f16(nh_f16{(uint16_t)(pair >> 16)});
}
NH_DEVICE unsigned nh_wave_max_u32(unsigned v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned o = (unsigned)nh_shfl_xor_i((int)v, d);
        v = o > v ? o : v;
    }
    return v;
}
NH_DEVICE float nh_med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
NH_DEVICE void nh_atomic_add(float* p, float v) { *p += v; }
NH_DEVICE void nh_atomic_max_u32(unsigned* p, unsigned v) { if (v > *p) *p = v; }
NH_DEVICE void nh_glds16(const float* g, float* lds_wave_base) { memcpy(lds_wave_base + 4 * emu::cur->lane, g, 16); }
struct NhDmaSrc {
    const char* base;
    unsigned bytes;
};
NH_DEVICE NhDmaSrc nh_dma_src(const float* base, unsigned bytes) { return NhDmaSrc{(const char*)base, bytes}; }
NH_DEVICE void nh_dma16(const NhDmaSrc& s, int voff, int soff, float* lds_wave_base) {
    const unsigned off = (unsigned)(voff + soff);
    if (off + 16 <= s.bytes)  // out-of-range lanes: the descriptor's bounds check (nothing is read)
        memcpy(lds_wave_base + 4 * emu::cur->lane, s.base + off, 16);
}
NH_DEVICE unsigned nh_lds_addr(const float* lds_ptr) { return (unsigned)((const char*)lds_ptr - emu::g_dyn_smem); }
NH_DEVICE void nh_dma16a(const NhDmaSrc& s, int voff, int soff, unsigned lds_wave_addr) {
    nh_dma16(s, voff, soff, (float*)(emu::g_dyn_smem + lds_wave_addr));
}
NH_DEVICE int nh_uload_i32(const int* p, int i) { return p[i]; }
NH_DEVICE void nh_wait_vmem() {}
template <int N>
NH_DEVICE void nh_wait_vmem_keep() {}
NH_DEVICE void nh_sched_fence() {}
NH_DEVICE unsigned long long nh_wall_clock() { return 0ull; }
NH_DEVICE unsigned long long nh_core_clock() { return 0ull; }
NH_DEVICE void nh_sincos(float x, float* s, float* c) {
    *s = sinf(x);
    *c = cosf(x);
}
