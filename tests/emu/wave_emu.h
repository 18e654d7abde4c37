// wave_emu.h -- TEST-ONLY SIMT emulator.  NOT part of the product and never linked into libhadi.
//
// Lets tests/emu/emu_driver.cpp compile the *same* kernel source (csrc/hadi_kernels.h) with g++ and
// run it on the host -- one thread per wavefront, its 64 lanes as fibers in lock step (see "Execution
// model" below), a counting barrier for __syncthreads -- so that layout / indexing / table
// bugs surface here, where there is no GPU, instead of on a GPU box.  It checks kernel logic only;
// performance, parity claims and every shipped code path use the real gfx950 build.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <sys/mman.h>

#include <condition_variable>
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#ifndef __restrict__
#define __restrict__ __restrict
#endif

struct emu_dim3 {
    unsigned x, y, z;
};
struct double2 {
    double x, y;
};
struct float2 {
    float x, y;
};
struct float4 {
    float x, y, z, w;
};

namespace emu {
// Execution model (round 4): one OS thread per WAVEFRONT, its lanes are fibers on that thread, resumed round-robin by the
// wavefront's scheduler (run_wave below).  A wave-level rendezvous (the lock step behind __shfl, hadi_wave_rendezvous) is a
// user-space context switch per lane -- rounds 1 - 3 ran every lane as an OS thread and met at pthread barriers: 64 futex
// waits per rendezvous, up to 1024 threads per block, nine tenths of the suite's time in the kernel.  __syncthreads() meets
// the other wavefronts of the block at a counting barrier that, like the hardware's, stops counting a wavefront that has
// ended.  Wavefronts of a block still run concurrently (flag polling between them works as before).
extern "C" void emu_ctx_switch(void **save_sp, void *load_sp);
struct Lane {
    void *sp = nullptr;
    char *stack = nullptr;
    int state = 0;  // 0 runnable, 1 at the wave rendezvous, 2 at the block barrier, 3 finished
    unsigned tid = 0;
};
struct BlockState;
struct WaveState {
    Lane lane[64];
    int nlanes = 0, cur = -1;
    void *sched_sp = nullptr;
    const std::function<void()> *body = nullptr;
    double slot[64];
    double pub[8][64];  // register images published once (hadi_pb_load_table) so that static readlanes need no rendezvous
};
struct BlockState {
    std::mutex mu;
    std::condition_variable cv;
    int live = 0, arrived = 0;
    unsigned long generation = 0;
    std::vector<WaveState *> waves;
    unsigned char *dyn_smem = nullptr;
    // called by a wavefront's scheduler once all its live lanes wait at __syncthreads()
    void arrive() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long g = generation;
        if (++arrived >= live) { arrived = 0; generation++; cv.notify_all(); return; }
        cv.wait(lk, [&] { return generation != g; });
    }
    // a wavefront has ended: the waiting ones no longer wait for it
    void leave() {
        std::unique_lock<std::mutex> lk(mu);
        live--;
        if (live > 0 && arrived >= live) { arrived = 0; generation++; cv.notify_all(); }
    }
};
extern thread_local emu_dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockState *t_block;
extern thread_local WaveState *t_wave;
extern thread_local int t_lane;

inline void lane_yield(int state) {
    WaveState *w = t_wave;
    Lane &l = w->lane[w->cur];
    l.state = state;
    emu_ctx_switch(&l.sp, w->sched_sp);
}
inline void wave_barrier() { lane_yield(1); }
inline void block_barrier() { lane_yield(2); }

inline double shfl_idx(double v, int src_lane) {
    WaveState *w = t_wave;
    w->slot[t_lane] = v;
    wave_barrier();
    const double out = w->slot[src_lane & 63];
    wave_barrier();
    return out;
}
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

inline void __syncthreads() { emu::block_barrier(); }
// HIP semantics: a source lane outside [0,63] returns the caller's own value.
inline double __shfl_up(double v, int delta) {
    const int src = emu::t_lane - delta;
    return emu::shfl_idx(v, src < 0 ? emu::t_lane : src);
}
inline double __shfl_down(double v, int delta) {
    const int src = emu::t_lane + delta;
    return emu::shfl_idx(v, src > 63 ? emu::t_lane : src);
}
inline double __shfl(double v, int lane) { return emu::shfl_idx(v, lane); }

namespace emu {
constexpr size_t LANE_STACK = 512 * 1024;
inline void lane_entry() {
    WaveState *w = t_wave;
    (*w->body)();
    lane_yield(3);
    std::abort();  // (a finished lane is never resumed)
}
// Scheduler of one wavefront (runs on the wavefront's OS thread).
inline void run_wave(BlockState *bs, WaveState *w, unsigned block, unsigned b, unsigned grid) {
    t_block = bs;
    t_wave = w;
    t_blockIdx = {b, 0, 0};
    t_blockDim = {block, 1, 1};
    t_gridDim = {grid, 1, 1};
    char *stacks = static_cast<char *>(mmap(nullptr, LANE_STACK * w->nlanes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (stacks == MAP_FAILED) { std::perror("emu: mmap"); std::abort(); }
    for (int k = 0; k < w->nlanes; k++) {
        Lane &l = w->lane[k];
        l.stack = stacks + LANE_STACK * k;
        // initial frame for emu_ctx_switch: six callee-saved registers, then the entry point as its return address; the stack
        // pointer is 8 mod 16 when lane_entry starts, as after a call
        void **top = reinterpret_cast<void **>(l.stack + LANE_STACK - 64);
        top[-1] = nullptr;
        top[-2] = reinterpret_cast<void *>(&lane_entry);
        for (int r = 3; r <= 8; r++) top[-r] = nullptr;
        l.sp = top - 8;
        l.state = 0;
    }
    for (;;) {
        int done = 0, at_wave = 0, at_block = 0;
        for (int k = 0; k < w->nlanes; k++) {
            Lane &l = w->lane[k];
            if (l.state == 0) {
                w->cur = k;
                t_lane = (int)(l.tid & 63);
                t_threadIdx = {l.tid, 0, 0};
                emu_ctx_switch(&w->sched_sp, l.sp);
            }
            done += l.state == 3;
            at_wave += l.state == 1;
            at_block += l.state == 2;
        }
        if (done == w->nlanes) break;
        if (at_wave && at_block) {
            std::fprintf(stderr, "emu: wavefront %u of block %u: %d lanes wait at a wave rendezvous, %d at __syncthreads()\n", w->lane[0].tid / 64, b, at_wave, at_block);
            std::abort();
        }
        if (at_block) bs->arrive();
        for (int k = 0; k < w->nlanes; k++)
            if (w->lane[k].state != 3) w->lane[k].state = 0;
    }
    bs->leave();
    munmap(stacks, LANE_STACK * w->nlanes);
}
// Runs body() for every block (sequentially) with `block` threads each.  Every lane of a wavefront must reach the same
// shuffles / barriers, exactly as on hardware with a full EXEC mask; a lane that returns early simply drops out (its wave
// mates must not depend on it afterwards, which holds for the wave- or block-uniform early exits the kernels use).
template <class F>
void launch(unsigned grid, unsigned block, F body, size_t dyn_smem_bytes = 0, bool concurrent = false) {
    // `concurrent`: all blocks of the grid at once, each with its own LDS -- for kernels whose blocks wait for each other
    // (the team barrier of the instance-resident launch); otherwise block after block, as rounds 1 - 3 did
    const std::function<void()> fn = body;
    const unsigned nw = (block + 63) / 64;
    const size_t smem_doubles = (dyn_smem_bytes + 7) / 8 + 2;
    const unsigned group = concurrent ? grid : 1;
    for (unsigned b0 = 0; b0 < grid; b0 += group) {
        // LDS is NOT zero on a GPU: poison it so that reads of never-written LDS surface here
        std::vector<double> smem_store(smem_doubles * group, std::nan(""));
        std::vector<BlockState> bss(group);
        std::vector<std::thread> th;
        for (unsigned g = 0; g < group; g++) {
            BlockState &bs = bss[g];
            bs.dyn_smem = reinterpret_cast<unsigned char *>(smem_store.data() + smem_doubles * g);
            bs.live = (int)nw;
            for (unsigned w = 0; w < nw; w++) {
                WaveState *ws = new WaveState;
                ws->nlanes = (int)std::min(64u, block - w * 64);
                for (int k = 0; k < ws->nlanes; k++) ws->lane[k].tid = w * 64 + k;
                ws->body = &fn;
                std::memset(ws->slot, 0, sizeof(ws->slot));
                bs.waves.push_back(ws);
            }
        }
        if (group == 1 && nw == 1) {
            run_wave(&bss[0], bss[0].waves[0], block, b0, grid);
        } else {
            th.reserve((size_t)group * nw);
            for (unsigned g = 0; g < group; g++)
                for (unsigned w = 0; w < nw; w++) th.emplace_back([&, g, w]() { run_wave(&bss[g], bss[g].waves[w], block, b0 + g, grid); });
            for (auto &x : th) x.join();
        }
        for (auto &bs : bss)
            for (auto *ws : bs.waves) delete ws;
    }
}
}  // namespace emu

// x86-64 System V: save the callee-saved registers and the stack pointer of the running context, load the other one's
asm(R"(
    .text
    .globl emu_ctx_switch
    .type emu_ctx_switch,@function
emu_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_ctx_switch,.-emu_ctx_switch
)");
