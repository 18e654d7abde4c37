// wave_emu.h -- TEST-ONLY SIMT emulator.  NOT part of the product and never linked into libhadi.
//
// Lets tests/emu/emu_driver.cpp compile the *same* kernel source (csrc/hadi_kernels.h) with g++ and
// run it on host threads -- one std::thread per "lane", 64-lane wavefronts, pthread barriers for
// __syncthreads and for the lock-step exchange behind __shfl_* -- so that layout / indexing / table
// bugs surface here, where there is no GPU, instead of on a GPU box.  It checks kernel logic only;
// performance, parity claims and every shipped code path use the real gfx950 build.
#pragma once
#include <pthread.h>
#include <sched.h>

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
struct WaveState {
    pthread_barrier_t bar;
    double slot[64];
    double pub[8][64];  // register images published once (hadi_pb_load_table) so that static readlanes need no rendezvous
};
struct BlockState {
    pthread_barrier_t bar;
    std::vector<WaveState *> waves;
    unsigned char *dyn_smem = nullptr;
};
extern thread_local emu_dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockState *t_block;
extern thread_local WaveState *t_wave;
extern thread_local int t_lane;

inline double shfl_idx(double v, int src_lane) {
    WaveState *w = t_wave;
    w->slot[t_lane] = v;
    pthread_barrier_wait(&w->bar);
    const double out = w->slot[src_lane & 63];
    pthread_barrier_wait(&w->bar);
    return out;
}
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

inline void __syncthreads() { pthread_barrier_wait(&emu::t_block->bar); }
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
// Runs kernel(args...) for every block (sequentially) with `block` threads each.  Every thread of a
// wave must reach the same shuffles / barriers, exactly as on hardware with a full EXEC mask; a
// thread that returns early simply drops out (its wave mates must not shuffle afterwards, which
// holds for the wave- or block-uniform early exits the kernels use).
template <class F>
void launch(unsigned grid, unsigned block, F body, size_t dyn_smem_bytes = 0) {
    // LDS is NOT zero on a GPU: poison it so that reads of never-written LDS surface here
    std::vector<double> smem_store((dyn_smem_bytes + 7) / 8 + 2, std::nan(""));
    for (unsigned b = 0; b < grid; b++) {
        BlockState bs;
        bs.dyn_smem = reinterpret_cast<unsigned char *>(smem_store.data());
        pthread_barrier_init(&bs.bar, nullptr, block);
        const unsigned nw = (block + 63) / 64;
        for (unsigned w = 0; w < nw; w++) {
            WaveState *ws = new WaveState;
            const unsigned cnt = std::min(64u, block - w * 64);
            pthread_barrier_init(&ws->bar, nullptr, cnt);
            std::memset(ws->slot, 0, sizeof(ws->slot));
            bs.waves.push_back(ws);
        }
        std::vector<std::thread> th;
        th.reserve(block);
        for (unsigned t = 0; t < block; t++) {
            th.emplace_back([&, t, b]() {
                t_threadIdx = {t, 0, 0};
                t_blockIdx = {b, 0, 0};
                t_blockDim = {block, 1, 1};
                t_gridDim = {grid, 1, 1};
                t_block = &bs;
                t_wave = bs.waves[t / 64];
                t_lane = (int)(t % 64);
                body();
            });
        }
        for (auto &x : th) x.join();
        for (auto *ws : bs.waves) {
            pthread_barrier_destroy(&ws->bar);
            delete ws;
        }
        pthread_barrier_destroy(&bs.bar);
    }
}
}  // namespace emu
