// nh_emu.cpp -- TEST INFRASTRUCTURE ONLY (see nh_emu.h).  Fibre scheduler for the wavefront emulator.
#include "nh_emu.h"

#include <stdio.h>
#include <stdlib.h>

#include <vector>

extern "C" void nh_emu_switch(void** save_sp, void* load_sp);
asm(".text\n"
    ".globl nh_emu_switch\n"
    ".type nh_emu_switch,@function\n"
    "nh_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp,(%rdi)\n"
    "  movq %rsi,%rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size nh_emu_switch,.-nh_emu_switch\n");

namespace emu {
Fiber* cur = nullptr;
Dim3 g_blockIdx, g_blockDim, g_gridDim;
char* g_dyn_smem = nullptr;

namespace {
constexpr size_t kStack = 256 * 1024;
std::vector<Fiber> g_fibers;
std::vector<WaveState> g_waves;
std::vector<char*> g_stack_pool;
void* g_sched_sp = nullptr;
const std::function<void()>* g_body = nullptr;
int g_blk_live = 0, g_blk_arrived = 0;
unsigned g_blk_gen = 0;
unsigned long g_progress = 0;

void to_scheduler() { nh_emu_switch(&cur->sp, g_sched_sp); }

void fiber_entry() {
    (*g_body)();
    Fiber* f = cur;
    f->done = true;
    ++g_progress;
    // an exited thread counts as "arrived" for everybody still waiting
    WaveState& w = g_waves[f->wave];
    --w.live;
    if (w.live > 0 && w.arrived == w.live) {
        w.arrived = 0;
        ++w.gen;
    }
    --g_blk_live;
    if (g_blk_live > 0 && g_blk_arrived == g_blk_live) {
        g_blk_arrived = 0;
        ++g_blk_gen;
    }
    to_scheduler();
    fprintf(stderr, "nh_emu: resumed a finished fibre\n");
    abort();
}
}  // namespace

WaveState& cur_wave() { return g_waves[cur->wave]; }

void wave_barrier() {
    WaveState& w = g_waves[cur->wave];
    unsigned gen = w.gen;
    if (++w.arrived == w.live) {
        w.arrived = 0;
        ++w.gen;
        ++g_progress;
        return;
    }
    while (w.gen == gen) to_scheduler();
}

void block_barrier() {
    unsigned gen = g_blk_gen;
    if (++g_blk_arrived == g_blk_live) {
        g_blk_arrived = 0;
        ++g_blk_gen;
        ++g_progress;
        return;
    }
    while (g_blk_gen == gen) to_scheduler();
}

void launch(Dim3 grid, Dim3 block, size_t smem, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    g_gridDim = grid;
    g_blockDim = block;
    g_body = &body;
    while ((int)g_stack_pool.size() < nthreads) {
        void* p = nullptr;
        if (posix_memalign(&p, 64, kStack) != 0) abort();
        g_stack_pool.push_back((char*)p);
    }
    std::vector<char> dyn(smem + 64);
    g_dyn_smem = (char*)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
    const char* shuffle_env = getenv("NH_EMU_REVERSE");
    const bool reverse = shuffle_env && shuffle_env[0] == '1';

    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                g_blockIdx = Dim3(bx, by, bz);
                g_fibers.assign(nthreads, Fiber());
                g_waves.assign(nwaves, WaveState());
                for (int wv = 0; wv < nwaves; ++wv) {
                    int lo = wv * 64, hi = lo + 64 > nthreads ? nthreads : lo + 64;
                    g_waves[wv].live = hi - lo;
                    g_waves[wv].arrived = 0;
                    g_waves[wv].gen = 0;
                }
                g_blk_live = nthreads;
                g_blk_arrived = 0;
                g_blk_gen = 0;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = g_fibers[t];
                    f.stack = g_stack_pool[t];
                    f.lin = t;
                    f.lane = t & 63;
                    f.wave = t >> 6;
                    f.tid = Dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.xphase = 0;
                    f.done = false;
                    uint64_t* top = (uint64_t*)(((uintptr_t)(f.stack + kStack)) & ~(uintptr_t)15);
                    *--top = 0;                         // alignment slot (entry never returns)
                    *--top = (uint64_t)&fiber_entry;    // 'ret' target of the first switch
                    for (int r = 0; r < 6; ++r) *--top = 0;
                    f.sp = top;
                }
                int remaining = nthreads;
                unsigned long last_progress = g_progress;
                long idle_switches = 0;
                while (remaining > 0) {
                    remaining = 0;
                    for (int k = 0; k < nthreads; ++k) {
                        int t = reverse ? nthreads - 1 - k : k;
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        cur = &f;
                        nh_emu_switch(&g_sched_sp, f.sp);
                        if (!f.done) ++remaining;
                    }
                    if (g_progress == last_progress) {
                        if (++idle_switches > 4) {
                            fprintf(stderr, "nh_emu: deadlock in block (%u,%u,%u): %d threads stuck at a barrier\n", bx, by,
                                    bz, remaining);
                            abort();
                        }
                    } else {
                        idle_switches = 0;
                        last_progress = g_progress;
                    }
                }
            }
    cur = nullptr;
    g_body = nullptr;
}
}  // namespace emu
