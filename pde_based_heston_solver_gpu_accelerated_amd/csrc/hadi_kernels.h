// hadi_kernels.h -- the gfx950 kernels of the Douglas ADI sweep.
//
// One Douglas step (device_solver.hpp:226-265) = two streaming passes over the state:
//
//   pass A (row pass, hadi_pass_a<B>): one 64-lane wavefront marches over R consecutive v-rows of
//     one instance.  Lane l keeps the B s-nodes i = 1+B*l .. B*l+B of five v-rows in registers
//     (rolling window), so the 9-point A0 stencil, the A1 3-point and the A2 5-point products
//     (hes_a0_kernels.hpp:59-94, hes_a1_kernels.hpp:111-135, hes_a2_shuffled_kernels.hpp:180-239)
//     need no LDS; it forms Y0 (device_solver.hpp:236-250), solves the A1 tridiagonal system of
//     the row (hes_a1_kernels.hpp:139-161) with a wavefront-parallel partition method
//     (in-lane Thomas on B-1 interior unknowns + parallel cyclic reduction over the 64 interface
//     unknowns via cross-lane shuffles) and writes the A2 right-hand side (device_solver.hpp:254-260).
//     Reads U once (+4 halo rows per tile), writes Y once: 16 B per point.
//
//   pass B (column pass, hadi_pass_b): lane <-> one s-column, so every v-row is read fully
//     coalesced and the reference's shuffle/unshuffle transposes (hes_a2_shuffled_kernels.hpp:12-44)
//     do not exist.  The pentadiagonal system (hes_a2_shuffled_kernels.hpp:243-299) has the same
//     matrix for every column and time step, so its factorisation is precomputed per instance
//     (hadi_core.h); the v-rows are cut into P chunks, one wavefront each, whose local solves run
//     in registers and are coupled by a SPIKE reduced system exchanged through LDS.  Applies the
//     American projection (device_solver.hpp:358-372).  Reads Y once, writes U once: 16 B per point.
//
// No MFMA: ~75 flop per 32 B.
#pragma once
#include "hadi_core.h"

struct HadiSweepArgs {
    // state, internal layout [inst][row][rowp]
    double *U;         // solution
    double *Y;         // A2 right-hand side between the passes
    double *LAM;       // lambda_bar (American) or nullptr
    double *R1, *C2;   // Craig-Sneyd only: predictor quantities reused by the corrector (see hadi_row_step)
    const double *U0;  // payoff (American) or nullptr
    const int *pay_mis;  // American: per instance, != 0 if the payoff differs between v-rows (0 = it depends on s only)
    // tables
    const double *scoef, *b2row, *rowc, *pb, *rinv;
    const HadiInstPar *ipar;
    HadiLayout L;
    int n_inst;
    int R, ntiles;   // pass A: rows per wave, tiles per instance
    int RS, sblocks; // strip row pass: v-rows per wavefront strip, 8-strip blocks per instance
    int ctiles;      // pass B: 64-column tiles per instance
    int btpw, bgroups;  // pass B: column tiles per block, blocks per instance
    int tile_il;        // pass B: 1 = the blocks of an instance take the full column tiles INTERLEAVED (block g: g, g + G, g + 2 G ...)
                        // instead of btpw consecutive ones each (hadi_pb_tiles)
    int american;
    int pos_m1;      // storage position of i = m1 (lambda_bar is forced to 0 there)
    int *err;        // the handle's sticky error word (host-pinned, device-visible): kernels OR a HADI_DEVERR_* code into it,
                     // the host reads it after the sweep and fails the call (hadi.h: HADI_ERR_INTERNAL)
    int debug;       // test hooks, 0 in production (hadi_set_tuning "debug_fault"): HADI_DEBUG_* bits
};

// Device-side error codes (bits of *HadiSweepArgs.err)
#define HADI_DEVERR_RENDEZVOUS 1  // a pair rendezvous of the two-wavefront rows ran out of polls: the partner's token never came
// Test hooks (bits of HadiSweepArgs.debug)
#define HADI_DEBUG_WITHHOLD_TOKEN 1  // the high half of every two-wavefront row withholds its token on v-row 1
#define HADI_DEBUG_TEAM_NO_ROWS 16   // hadi_team_kernel, timing diagnostics (results are wrong): skip the row phase's work
#define HADI_DEBUG_TEAM_NO_COLS 32   // ... skip the column phase's work
#define HADI_DEBUG_TEAM_NO_BARRIER 64  // ... skip the team barriers
#define HADI_DEBUG_COL_NO_SOLVE 256     // hadi_pass_b1 / hadi_pass_b2, timing diagnostics (results are wrong): tiles are loaded and stored
                                       // but not solved -- what the memory system gives the pass's access pattern alone
#define HADI_DEBUG_COL_NO_REDUCED 512   // column pass, timing diagnostics (results are wrong): the interface exchange and its barrier run, the
                                       // reduced system t = R^-1 z does not (its 4 x 4P broadcast-operand FMAs per lane)
#define HADI_DEBUG_TEAM_DESERT 128     // hadi_team_kernel: block 1 of every team leaves before the first barrier (the others must
                                       // time out, report HADI_DEVERR_TEAM, and the host must solve the batch on the streaming path)


// Bounded poll of the pair rendezvous: ~0.2 s on the GPU (a resident partner answers within microseconds; under the
// host-thread emulator every poll is a sched_yield of one of 512 threads).  With the test hook set the bound is short,
// so that the forced failure costs microseconds.
#define HADI_RENDEZVOUS_POLLS(debug) (((debug) & HADI_DEBUG_WITHHOLD_TOKEN) ? (1 << 12) : (1 << 22))
// Guard exhausted: record it where the host will see it.  The row is solved with whatever the exchange buffer holds -- the
// kernel must drain, a hang would cost the GPU -- and the host turns the recorded code into HADI_ERR_INTERNAL: a stale-value
// solve is never returned as HADI_OK.
HADI_DEV HADI_FORCEINLINE void hadi_report(int *err, int code) {
#if defined(HADI_EMU)
    __atomic_fetch_or(err, code, __ATOMIC_RELAXED);
#else
    __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// Blocks b and b+8 share an XCD (and its L2).  Map the dispatch index so that consecutive logical
// ids -- neighbouring row tiles of one instance, which share halo rows -- land on the same XCD.
HADI_DEV HADI_FORCEINLINE int hadi_xcd_remap(int bid, int nblk_padded) {
    const int per = nblk_padded >> 3;
    return (bid & 7) * per + (bid >> 3);
}

template <int B>
HADI_DEV HADI_FORCEINLINE void hadi_load_row(const double *__restrict__ row, int lane, bool valid, double (&u)[B]) {
    if (!valid) {
#pragma unroll
        for (int r = 0; r < B; r++) u[r] = 0.0;
        return;
    }
    if constexpr (B == 1) {
        u[0] = row[lane];
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            const double2 t = *reinterpret_cast<const double2 *>(row + q * 128 + 2 * lane);
            u[2 * q] = t.x;
            u[2 * q + 1] = t.y;
        }
    }
}

template <int B>
HADI_DEV HADI_FORCEINLINE void hadi_store_row(double *__restrict__ row, int lane, const double (&u)[B]) {
    if constexpr (B == 1) {
        row[lane] = u[0];
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            double2 t;
            t.x = u[2 * q];
            t.y = u[2 * q + 1];
            *reinterpret_cast<double2 *>(row + q * 128 + 2 * lane) = t;
        }
    }
}

// Value of `v` held by lane `src` (0..63) of this wavefront.
HADI_DEV HADI_FORCEINLINE double hadi_lane_get(double v, int src) {
#if defined(HADI_EMU)
    return __shfl(v, src);
#else
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
#endif
}

// Value of `v` held by lane - 1 / lane + 1: a DPP wave shift (two v_mov_b32 on the VALU) instead of a ds_bpermute round
// trip through the LDS pipe, which the eight wavefronts of a CU share.  Lane 0 (resp. 63) gets 0 (bound_ctrl:0 -- which
// also spares the move that would initialise the destination).
HADI_DEV HADI_FORCEINLINE double hadi_lane_prev(double v) {
#if defined(HADI_EMU)
    const double t = __shfl(v, (emu::t_lane - 1) & 63);
    return emu::t_lane == 0 ? 0.0 : t;
#else
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
#endif
}
HADI_DEV HADI_FORCEINLINE double hadi_lane_next(double v) {
#if defined(HADI_EMU)
    const double t = __shfl(v, (emu::t_lane + 1) & 63);
    return emu::t_lane == 63 ? 0.0 : t;
#else
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
#endif
}

#if defined(HADI_STAMPS) && !defined(HADI_EMU)
// Diagnostic build only (tools/stamps.py): per-phase cycle sums of the row pass, never in the product.
// Each wavefront accumulates its own sums and adds them to the global array once, at kernel end.
__device__ unsigned long long g_hadi_stamps[32];
#define HADI_STAMP_ACC unsigned long long *stamp_acc_;
#define HADI_STAMP_DECL(accptr) unsigned long long *sacc_ = (accptr); unsigned long long stamp_prev_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev_) :: "memory");
// HADI_STAMPS=1: only the per-iteration stamps (8, 9, 10; small perturbation); 2: also the row phases
#define HADI_STAMP(k) do { if ((k) >= 8 || HADI_STAMPS >= 2) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    sacc_[k] += t_ - stamp_prev_; stamp_prev_ = t_; } } while (0)
// HADI_STAMPS=3: column-pass phases (16..23) instead
#define HADI_STAMPB(k) do { if (HADI_STAMPS == 3) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    sacc_[k] += t_ - stamp_prev_; stamp_prev_ = t_; } } while (0)
// HADI_STAMPS=4: strip row pass phases (24..29)
#define HADI_STAMPC(k) do { if (HADI_STAMPS == 4) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    sacc_[k] += t_ - stamp_prev_; stamp_prev_ = t_; } } while (0)
#define HADI_STAMPB_WAIT(n) do { if (HADI_STAMPS == 3) { if ((n) >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); \
    else if ((n) >= 33) asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } } while (0)
#else
#define HADI_STAMP_ACC
#define HADI_STAMP_DECL(accptr)
#define HADI_STAMP(k)
#define HADI_STAMPB(k)
#define HADI_STAMPC(k)
#define HADI_STAMPB_WAIT(n)
#endif

// ---- buffer addressing: wave-uniform base + scalar row offset + one per-lane 32-bit offset ----------
// (raw buffer loads/stores take the row offset in an SGPR, so walking down a column costs no VALU
// address arithmetic and no address VGPR pairs: cdna_hip_programming.md T8)
#if defined(HADI_EMU)
struct HadiBuf { const void *p; };
HADI_DEV HADI_FORCEINLINE HadiBuf hadi_make_buf(const void *base, size_t) { return HadiBuf{base}; }
HADI_DEV HADI_FORCEINLINE double hadi_buf_load(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    return *reinterpret_cast<const double *>(static_cast<const char *>(b.p) + voff_bytes + soff_bytes);
}
HADI_DEV HADI_FORCEINLINE void hadi_buf_store(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes, double v) {
    if (voff_bytes >= 0x80000000u) return;  // HADI_BUF_DROP: the hardware range check discards the lane's store
    *reinterpret_cast<double *>(const_cast<char *>(static_cast<const char *>(b.p)) + voff_bytes + soff_bytes) = v;
}
template <class T>
HADI_DEV HADI_FORCEINLINE double hadi_buf_load_t(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    return (double)*reinterpret_cast<const T *>(static_cast<const char *>(b.p) + voff_bytes + soff_bytes);
}
HADI_DEV HADI_FORCEINLINE double hadi_buf_load_sc1(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes) { return hadi_buf_load(b, voff_bytes, soff_bytes); }
HADI_DEV HADI_FORCEINLINE void hadi_buf_load2_sc1(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes, double &x, double &y) {
    x = hadi_buf_load(b, voff_bytes, soff_bytes); y = hadi_buf_load(b, voff_bytes + 8, soff_bytes);
}
template <class T>
HADI_DEV HADI_FORCEINLINE void hadi_buf_store_t(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes, double v) {
    if (voff_bytes >= 0x80000000u) return;
    *reinterpret_cast<T *>(const_cast<char *>(static_cast<const char *>(b.p)) + voff_bytes + soff_bytes) = (T)v;
}
#else
typedef unsigned hadi_u32x2 __attribute__((ext_vector_type(2)));
// Cache policy of the column pass: non-temporal (aux bit 1 = nt) loads and stores.  Every load of a sweep is a last
// use (the array is overwritten by the next pass) -- streaming loads do not displace the freshly written array from
// the 256 MB memory-side cache, which the next pass (walking the instances the other way round) then hits.
// Measured on MI355X, 256 instances of 512x256: step 0.303 -> 0.284 ms; tools/mallbench.hip shows the effect on a
// plain ping-pong copy (512 MB working set: 5.4 -> 7.3 TB/s).
#ifndef HADI_AUX_NT
#define HADI_AUX_NT 2
#endif
struct HadiBuf { __amdgpu_buffer_rsrc_t r; };
HADI_DEV HADI_FORCEINLINE HadiBuf hadi_make_buf(const void *base, size_t bytes) {
    return HadiBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000)};
}
HADI_DEV HADI_FORCEINLINE double hadi_buf_load(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    const hadi_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff_bytes, soff_bytes, HADI_AUX_NT);
    return __hiloint2double((int)v.y, (int)v.x);
}
// Agent-coherent loads (cache policy sc1 on top of nt: gfx940+ cpol bit 4): served by the L2, never by this CU's vector L1 --
// what the instance-resident kernel reads the rows other CUs of its team wrote with.
#define HADI_AUX_NT_SC1 (HADI_AUX_NT | 16)
HADI_DEV HADI_FORCEINLINE double hadi_buf_load_sc1(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    const hadi_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff_bytes, soff_bytes, HADI_AUX_NT_SC1);
    return __hiloint2double((int)v.y, (int)v.x);
}
typedef unsigned hadi_u32x4 __attribute__((ext_vector_type(4)));
HADI_DEV HADI_FORCEINLINE void hadi_buf_load2_sc1(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes, double &x, double &y) {
    const hadi_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff_bytes, soff_bytes, HADI_AUX_NT_SC1);
    x = __hiloint2double((int)v.y, (int)v.x);
    y = __hiloint2double((int)v.w, (int)v.z);
}
HADI_DEV HADI_FORCEINLINE void hadi_buf_store(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes, double v) {
    hadi_u32x2 d;
    d.x = (unsigned)__double2loint(v);
    d.y = (unsigned)__double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(d, b.r, voff_bytes, soff_bytes, HADI_AUX_NT);
}
#endif

#if !defined(HADI_EMU)
// element type T of the state: double, or float for the fp32-state sweep (widened on load, rounded on store)
template <class T>
HADI_DEV HADI_FORCEINLINE double hadi_buf_load_t(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes) {
    if constexpr (sizeof(T) == 8) {
        return hadi_buf_load(b, voff_bytes, soff_bytes);
    } else {
        return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff_bytes, soff_bytes, HADI_AUX_NT));
    }
}
template <class T>
HADI_DEV HADI_FORCEINLINE void hadi_buf_store_t(HadiBuf b, unsigned voff_bytes, unsigned soff_bytes, double v) {
    if constexpr (sizeof(T) == 8) {
        hadi_buf_store(b, voff_bytes, soff_bytes, v);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), b.r, voff_bytes, soff_bytes, HADI_AUX_NT);
    }
}
#endif

// A raw buffer store whose VGPR offset is >= num_records is dropped by the hardware (the SGPR offset takes no part
// in the range check): lanes that must not store get this offset instead of an exec-mask branch per row.
#define HADI_BUF_DROP 0x80000000u
// Cache policy of the row pass's LDS-DMA loads: non-temporal, like the column pass's loads (see HADI_AUX_NT).
// Measured with the strip kernel at 256 instances of 512x256: 1.22e11 -> 1.29e11 point-steps/s.
#ifndef HADI_DMA_POLICY
#define HADI_DMA_POLICY " nt"
#endif

// Value of `v` held by lane `src` of this wavefront, `src` wave-uniform: two v_readlane_b32, result in SGPRs.
HADI_DEV HADI_FORCEINLINE double hadi_read_lane(double v, int src) {
#if defined(HADI_EMU)
    return __shfl(v, src);
#else
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
#endif
}
// The HADI_RCL scalars of one row-table entry through the scalar data cache into SGPRs (the entry is wave-uniform):
// three s_load_dwordx8.  Inline asm because hipcc would issue per-lane vector loads here (it cannot prove that the
// kernel's own stores leave the table alone), and those would also drain the LDS-DMA prefetch.
#if !defined(HADI_EMU)
typedef int hadi_i32x8 __attribute__((ext_vector_type(8)));
struct HadiSRow { hadi_i32x8 q0, q1, q2; };
#else
struct HadiSRow { const double *p; };
#endif
// issue only: the three SGPR octets are NOT valid until hadi_sload_wait()
HADI_DEV HADI_FORCEINLINE void hadi_sload_issue(const double *__restrict__ entry, HadiSRow &r) {
#if defined(HADI_EMU)
    r.p = entry;
#else
    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x20\n\ts_load_dwordx8 %2, %3, 0x40"
                 : "=&s"(r.q0), "=&s"(r.q1), "=&s"(r.q2)
                 : "s"(entry)
                 : "memory");
#endif
}
HADI_DEV HADI_FORCEINLINE void hadi_sload_wait(HadiSRow &r, double (&rt)[HADI_RCL]) {
#if defined(HADI_EMU)
    for (int k = 0; k < HADI_RCL; k++) rt[k] = r.p[k];
#else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.q0), "+s"(r.q1), "+s"(r.q2) : : "memory");
    rt[0] = __hiloint2double(r.q0[1], r.q0[0]); rt[1] = __hiloint2double(r.q0[3], r.q0[2]);
    rt[2] = __hiloint2double(r.q0[5], r.q0[4]); rt[3] = __hiloint2double(r.q0[7], r.q0[6]);
    rt[4] = __hiloint2double(r.q1[1], r.q1[0]); rt[5] = __hiloint2double(r.q1[3], r.q1[2]);
    rt[6] = __hiloint2double(r.q1[5], r.q1[4]); rt[7] = __hiloint2double(r.q1[7], r.q1[6]);
    rt[8] = __hiloint2double(r.q2[1], r.q2[0]); rt[9] = __hiloint2double(r.q2[3], r.q2[2]);
    rt[10] = __hiloint2double(r.q2[5], r.q2[4]); rt[11] = __hiloint2double(r.q2[7], r.q2[6]);
#endif
}
// A wave-uniform double moved to SGPRs (two v_readfirstlane): loop-invariant scalars then cost no VGPRs.
HADI_DEV HADI_FORCEINLINE double hadi_uniform_d(double x) {
#if defined(HADI_EMU)
    return x;
#else
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
#endif
}
// Issue priority of this wavefront (s_setprio 0..3).  The row step raises it as the row progresses -- 1 for the forward
// sweep, 3 for the cyclic reduction -- so that of the two wavefronts sharing a SIMD the one deep in its dependent
// chains (short instructions waiting on ds_bpermute round trips) is served the moment it can issue, while the other
// one's long independent streams fill the gaps.  Measured: strip kernel (512x256 x256) 0.138 -> 0.127 ms/launch,
// shared-ring kernel (256x128 x1024) 0.187 -> 0.179 ms.
HADI_DEV HADI_FORCEINLINE void hadi_set_prio(int p) {
#if !defined(HADI_EMU)
    if (p == 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
#else
    (void)p;
#endif
}
// The lanes of a wavefront run in lock step on the GPU; the host-thread emulator needs a rendezvous wherever one
// lane reads LDS another lane of the same wavefront wrote.
HADI_DEV HADI_FORCEINLINE void hadi_wave_rendezvous() {
#if defined(HADI_EMU)
    pthread_barrier_wait(&emu::t_wave->bar);
#endif
}

// ---- LDS row ring helpers ---------------------------------------------------------------------------
// Asynchronous copy of one state row (rowp doubles, HBM layout == LDS layout) into the ring by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs).  Issued through inline asm on purpose:
// with the builtin hipcc sees an LDS write and puts s_waitcnt vmcnt(0) in front of the very next ds_read
// (it cannot know the ring slots differ), which serialises the prefetch with the row it should overlap.
// The asm form is invisible to that bookkeeping, so completion is OUR job: hadi_wait_vmcnt() (+ the barrier, where
// other wavefronts read the row)
// before anyone reads the rows (cdna_hip_programming.md 5.7).  Rows outside the allocation are zeros.
// T = double, or float for the fp32-state sweep (state stored as fp32, all arithmetic fp64): the row is copied as raw
// bytes either way, rowp * sizeof(T) is a multiple of 32.
template <class T>
HADI_DEV HADI_FORCEINLINE void hadi_row_to_lds(const T *__restrict__ grow, T *lrow, int rowp, int lane, bool exists) {
    constexpr int EPV = 16 / (int)sizeof(T);  // elements per 16-byte vector
    const int nvec = rowp / EPV;
    if (exists) {
        for (int v0 = 0; v0 < nvec; v0 += 64) {
            if (v0 + lane < nvec) {
#if defined(HADI_EMU)
                for (int e = 0; e < EPV; e++) lrow[EPV * (v0 + lane) + e] = grow[EPV * (v0 + lane) + e];
#else
                const T *gsrc = grow + EPV * (v0 + lane);
                // wave-uniform LDS byte address of this 1 KiB piece; the hardware adds lane*16
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) char *)(lrow + EPV * v0));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(gsrc), "s"(lds_dst)
                             : "memory");
#endif
            }
        }
    } else {
        for (int v = lane; v < nvec; v += 64)
            for (int e = 0; e < EPV; e++) lrow[EPV * v + e] = (T)0;
    }
}
// The same copy for a pitch known at compile time (rowp = 64 B G + 8): NFULL unmasked 1 KiB pieces and one
// partial piece, fully unrolled -- no loop counters, compares or exec-mask juggling in the row loop.
template <int B, class T, int G = 1>
HADI_DEV HADI_FORCEINLINE void hadi_row_to_lds_fixed(const T *__restrict__ grow, T *lrow, int lane, bool exists) {
    constexpr int EPV = 16 / (int)sizeof(T), ROWP = 64 * B * G + HADI_ROW_PAD(B, (int)sizeof(T)), NVEC = ROWP / EPV, NFULL = NVEC / 64, REM = NVEC - 64 * NFULL;
    static_assert(ROWP % EPV == 0, "row pitch must be a whole number of 16-byte vectors");
    if (exists) {
#if defined(HADI_EMU)
        for (int v = lane; v < NVEC; v += 64)
            for (int e = 0; e < EPV; e++) lrow[EPV * v + e] = grow[EPV * v + e];
#else
        const T *gsrc = grow + EPV * lane;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)lrow);
#pragma unroll
        for (int q = 0; q < NFULL; q++) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gsrc + EPV * 64 * q), "s"(lds0 + 1024u * q)
                         : "memory");
        }
        if constexpr (REM > 0) {
            if (lane < REM) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(gsrc + EPV * 64 * NFULL), "s"(lds0 + 1024u * NFULL)
                             : "memory");
            }
        }
#endif
    } else {
        for (int v = lane; v < NVEC; v += 64)
            for (int e = 0; e < EPV; e++) lrow[EPV * v + e] = (T)0;
    }
}
// G = 2 strips: the copy of ONE HALF of a row -- the pieces of 1 KiB that hold this wavefront's nodes (pairs, or quads
// with an fp32 state: piece q of half h starts at element q*PW*2 + PW*h, PW = 128 resp. 256) and, for the low half, the
// 128-byte pad piece with the i = 0 slot.  Returns the number of vector-memory instructions issued (wave-uniform).
template <int B, class T>
HADI_DEV HADI_FORCEINLINE int hadi_half_row_to_lds(const T *__restrict__ grow, T *lrow, int half, int lane, bool exists) {
    constexpr int EPV = 16 / (int)sizeof(T), PW = 64 * EPV, NP = B / EPV, PAD = HADI_ROW_PAD(B, (int)sizeof(T));
    constexpr int PADV = PAD / EPV;  // 16-byte vectors of the pad piece
    static_assert(B % EPV == 0 && PAD % EPV == 0 && PADV <= 64, "row layout");
    if (exists) {
#if defined(HADI_EMU)
        for (int q = 0; q < NP; q++)
            for (int e = 0; e < EPV; e++) lrow[q * PW * 2 + PW * half + EPV * lane + e] = grow[q * PW * 2 + PW * half + EPV * lane + e];
        if (half == 0 && lane < PADV)
            for (int e = 0; e < EPV; e++) lrow[64 * B * 2 + EPV * lane + e] = grow[64 * B * 2 + EPV * lane + e];
#else
        const T *gsrc = grow + PW * half + EPV * lane;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)(lrow + PW * half));
#pragma unroll
        for (int q = 0; q < NP; q++) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(gsrc + q * PW * 2), "s"(lds0 + 2048u * q)
                         : "memory");
        }
        if (half == 0) {  // wave-uniform
            if (lane < PADV) {
                unsigned keep;
                const unsigned ldsp = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)(lrow + 64 * B * 2));
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(grow + 64 * B * 2 + EPV * lane), "s"(ldsp)
                             : "memory");
            }
        }
#endif
        return NP + (half == 0 ? 1 : 0);
    }
    for (int q = 0; q < NP; q++)
        for (int e = 0; e < EPV; e++) lrow[q * PW * 2 + PW * half + EPV * lane + e] = (T)0;
    if (half == 0 && lane < PADV)
        for (int e = 0; e < EPV; e++) lrow[64 * B * 2 + EPV * lane + e] = (T)0;
    return 0;
}
// number of vector-memory instructions hadi_row_to_lds issues for an existing row
template <class T>
HADI_DEV HADI_FORCEINLINE int hadi_row_dma_count(int rowp) {
    return (rowp / (16 / (int)sizeof(T)) + 63) / 64;
}

template <int B>
HADI_DEV HADI_FORCEINLINE void hadi_lds_row(const double *lrow, int lane, double (&u)[B]) {
    if constexpr (B == 1) {
        u[0] = lrow[lane];
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            const double2 t = *reinterpret_cast<const double2 *>(lrow + q * 128 + 2 * lane);
            u[2 * q] = t.x;
            u[2 * q + 1] = t.y;
        }
    }
}

// pass A.  Block = W*G wavefronts working on W consecutive v-rows of one instance at a time, G wavefronts
// per row.  The rows j-2 .. j+W+1 they need (9-point A0, 5-point A2) sit in an LDS ring of 2W+4 rows that is
// refilled by LDS-DMA one iteration ahead, so HBM latency hides behind the line solves.  Lane l of wave g owns
// the s-nodes i = 1 + 64*B*g + B*l .. of its row.  The s-direction coefficient arrays live in LDS too.
// With G = 2 the row's tridiagonal system is split at the wave boundary: each half is solved against one
// unknown boundary value (a second right-hand side carried through the cyclic reduction) and the two
// boundary values follow from a 2x2 system exchanged through LDS -- small per-lane state (B = 4 at
// m1 = 512) is what lets four wavefronts share a SIMD.
template <class T>
struct HadiRowCtxT {
    const double *coef;  // LDS: Bm, Bp, Dm, Dp, each 64*B*G doubles in row layout
    double *xch;         // LDS: [W][8] boundary exchange between the two waves of a row (G = 2): 4 values + 2 tokens
    T *Yi;               // instance base of Y (T = float: fp32-state sweep)
    const double *Li;    // instance base of lambda_bar (American)
    const double *rowc;  // LDS copy of the row table of this block's tile: entry (j - j0)
    int j0;              // first v-row of the tile
    const double *b2r;   // instance b2 row (global)
    double *R1i, *C2i;   // instance bases of the Craig-Sneyd carry-over arrays (MODE 1 writes, MODE 2 reads)
    int lane, half, wrow, posL, posR, rowp;
    double dt, thdt, qd, half_rd, e_nm1, e_n;
    double hr0, inv0;    // i = 0 row of A1: reaction term (0 for the call) and 1 / (1 + theta dt hr0)
    // American without the lambda_bar array (AMER == 2, see hadi_row_step): LDS copy of the payoff row (it depends on s
    // only), 1/dt, and which (lane, node) of this wavefront is i = m1 (lambda_bar is forced to 0 there), -1 if none
    const double *payrow;
    double inv_dt;
    int m1_lane, m1_r;
    int *err;            // HadiSweepArgs.err / .debug (used by the G = 2 rendezvous only)
    int debug;
    HADI_STAMP_ACC
};
typedef HadiRowCtxT<double> HadiRowCtx;

// Loads this lane's B values of a row-layout array (LDS or global): pair q at q*128*G + 128*half + 2*lane.
// T = double (16-byte pairs) or float (fp32 state: 8-byte pairs, widened on load / rounded on store).
template <class T> struct HadiPair;
template <> struct HadiPair<double> { typedef double2 type; };
template <> struct HadiPair<float> { typedef float2 type; };
template <int B, int G, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_get_block(const T *row, int half, int lane, double (&u)[B]) {
    if constexpr (sizeof(T) == 4 && B >= 4) {  // fp32 state: quads (hadi_pos_f32), 16-byte accesses
#pragma unroll
        for (int q = 0; q < B / 4; q++) {
            const float4 t = *reinterpret_cast<const float4 *>(row + q * 256 * G + 256 * half + 4 * lane);
            u[4 * q] = (double)t.x; u[4 * q + 1] = (double)t.y; u[4 * q + 2] = (double)t.z; u[4 * q + 3] = (double)t.w;
        }
    } else if constexpr (B == 1) {
        u[0] = (double)row[64 * half + lane];
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            const typename HadiPair<T>::type t =
                *reinterpret_cast<const typename HadiPair<T>::type *>(row + q * 128 * G + 128 * half + 2 * lane);
            u[2 * q] = (double)t.x;
            u[2 * q + 1] = (double)t.y;
        }
    }
}
template <int B, int G, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_put_block(T *row, int half, int lane, const double (&u)[B]) {
    if constexpr (sizeof(T) == 4 && B >= 4) {
#pragma unroll
        for (int q = 0; q < B / 4; q++) {
            float4 t;
            t.x = (float)u[4 * q]; t.y = (float)u[4 * q + 1]; t.z = (float)u[4 * q + 2]; t.w = (float)u[4 * q + 3];
            *reinterpret_cast<float4 *>(row + q * 256 * G + 256 * half + 4 * lane) = t;
        }
    } else if constexpr (B == 1) {
        row[64 * half + lane] = (T)u[0];
    } else {
#pragma unroll
        for (int q = 0; q < B / 2; q++) {
            typename HadiPair<T>::type t;
            t.x = (T)u[2 * q];
            t.y = (T)u[2 * q + 1];
            *reinterpret_cast<typename HadiPair<T>::type *>(row + q * 128 * G + 128 * half + 2 * lane) = t;
        }
    }
}

// Vector stores hadi_put_block issues per lane for one row block: the counted vmcnt waits add this as the LOWER bound of
// the operations a row step puts behind a DMA batch (too high a count would let the wait pass with a DMA piece still in
// flight).  fp32 state at 4 or 8 nodes per lane stores QUADS (B/4 instructions), everything else pairs.
template <int B, class T>
HADI_DEV constexpr int hadi_put_block_stores() {
    return B == 1 ? 1 : (sizeof(T) == 4 && B >= 4) ? B / 4 : B / 2;
}

// One v-row: explicit stage, Y0, A1 line solve, A2 right-hand side.  LAST = this is the v-row that
// carries b2 (hes_boundary_kernels.hpp:62-66); AMER adds lambda_bar (device_solver.hpp:325-331).
// `active` is wave-uniform; with G = 2 every wave of the block must call this (it contains a barrier).
// MODE 0: Douglas step.  MODE 1 / 2: predictor / corrector of Craig-Sneyd (solver.hpp:781-907).  With
//   Y1rhs = Y0 + theta dt (b1 e_n - (A1U + b1 e_{n-1})),  C2 = theta dt (b2 e_n - (A2U + b2 e_{n-1}))
// the corrector's A1 right-hand side is Y0~ + theta dt (...) = Y1rhs + dt/2 (A0 Y2 - A0 U): MODE 1 is a
// Douglas row step that also stores R1 = Y1rhs - dt/2 A0U and C2; MODE 2 takes its rows from Y2, forms
// R1 + dt/2 A0 Y2, runs the same A1 solve and adds C2 -- it never needs U, A1U or A2U again.
// AMER == 2: American in the P representation.  After the projection  U = max(P, U0),  lambda_bar = max(0, (U0 - P)/dt)
// with  P = U_bar - dt lambda_bar_old  (device_solver.hpp:358-372 rewritten), so ONE array -- P, stored where U is --
// carries both, provided the payoff U0 depends on s only (then it is a per-lane constant here).  The row pass rebuilds U on
// the five stencil rows and lambda_bar on row j from P; no lambda_bar array is read or written by the sweep.
template <int B, int G, int AMER, bool LAST, int MODE = 0, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_row_step(const HadiRowCtxT<T> &c, bool active, int j, const T *rm2, const T *rm1,
                                             const T *r0, const T *rp1, const T *rp2) {
    const int lane = c.lane, rowp = c.rowp, half = c.half;
    constexpr int c0slot = 64 * B * G;
    constexpr int NB = B - 1;
    const double dt = c.dt, thdt = c.thdt, qd = c.qd, half_rd = c.half_rd, e_nm1 = c.e_nm1, e_n = c.e_n;
    const bool first_half = (half == 0), last_half = (half == G - 1);
    // state that survives the exchange barrier
    double ys[B], ps[B], gs[B], A2U[B], b2v[B], r1v[B], c2v[B];
    double Ysol = 0.0, Ssol = 0.0, yout_c0 = 0.0;

    HADI_STAMP_DECL(c.stamp_acc_)
    if (active) {
        // (LDS, not global: an ordinary global load here would make hipcc drain the in-flight LDS-DMA
        // prefetch with vmcnt(0) at the start of every row)
        const double *rc = c.rowc + (size_t)(j - c.j0) * HADI_RCL;
        const double v = rc[RC_V];
        const double wm = rc[RC_WM], wz = rc[RC_WZ], wp = rc[RC_WP];
        const double a2l2 = rc[RC_L2], a2l1 = rc[RC_L1], a2m = rc[RC_M], a2u1 = rc[RC_U1], a2u2 = rc[RC_U2];
        const double b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        const bool b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;  // (two entries on one v-row: m2 > m1 only)
        const int b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
        // which (wave, lane, slot) holds the b1 node of this v-row
        const int b1e = b1col - 1;
        const int b1half = (b1col >= 1) ? b1e / (64 * B) : -1;
        const int b1el = b1e - b1half * 64 * B;
        const int b1lane = (b1col >= 1 && b1half == half) ? b1el / B : -1;
        const int b1r = b1el - (b1el / B) * B;

        // ---- column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act) ------------
        double c0 = (double)r0[c0slot];
        double c0m2 = (double)rm2[c0slot], c0m1 = (double)rm1[c0slot], c0p1 = (double)rp1[c0slot], c0p2 = (double)rp2[c0slot];
        double lamc0 = 0.0;
        if constexpr (AMER == 1) lamc0 = c.Li[(size_t)j * rowp + c0slot];
        if constexpr (AMER == 2) {
            const double pay0 = c.payrow[c0slot];
            lamc0 = fmax(0.0, (pay0 - c0) * c.inv_dt);
            c0 = fmax(c0, pay0); c0m2 = fmax(c0m2, pay0); c0m1 = fmax(c0m1, pay0); c0p1 = fmax(c0p1, pay0); c0p2 = fmax(c0p2, pay0);
        }
        const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c0 + a2u1 * c0p1 + a2u2 * c0p2;
        const double b1c0 = b1_at0 ? b1val : 0.0;
        const double b2c0 = LAST ? c.b2r[c0slot] : 0.0;
        const double a1c0 = -c.hr0 * c0;  // A1 row 0: empty for the call (hr0 = 0), the reaction term for the put
        double y0c0 = c0 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
        y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
        double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
        if constexpr (MODE == 1) {  // A0 is zero on i = 0: R1 = Y1rhs there
            if (lane == 0 && first_half) {
                c.R1i[(size_t)j * rowp + c0slot] = y0c0;
                c.C2i[(size_t)j * rowp + c0slot] = c2c0;
            }
        }
        if constexpr (MODE == 2) {
            y0c0 = c.R1i[(size_t)j * rowp + c0slot];
            c2c0 = c.C2i[(size_t)j * rowp + c0slot];
        }
        const double x0 = y0c0 * c.inv0;  // A1 row 0 is decoupled: the identity for the call (hes_a1_kernels.hpp:56-61)
        yout_c0 = x0 + c2c0;

        HADI_STAMP(0);  // row scalars + column 0
        // ---- explicit operators.  A0 = (s-derivative) o (v-derivative): first the v-combination
        // t = wm u(j-1) + wz u(j) + wp u(j+1) on the block and its two s-neighbours, then the B-weights.
        double u0[B], tt[B];
        double lam[B], pay[B];
        if constexpr (AMER == 2) hadi_get_block<B, G>(c.payrow, half, lane, pay);
        {
            double um[B], up[B], u2[B];
            hadi_get_block<B, G, T>(r0, half, lane, u0);
            hadi_get_block<B, G, T>(rm1, half, lane, um);
            hadi_get_block<B, G, T>(rp1, half, lane, up);
            if constexpr (AMER == 2) {
#pragma unroll
                for (int r = 0; r < B; r++) {
                    lam[r] = fmax(0.0, (pay[r] - u0[r]) * c.inv_dt);  // from the raw P of row j
                    if (lane == c.m1_lane && r == c.m1_r) lam[r] = 0.0;
                    u0[r] = fmax(u0[r], pay[r]);
                    um[r] = fmax(um[r], pay[r]);
                    up[r] = fmax(up[r], pay[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < B; r++) {
                tt[r] = wm * um[r] + wz * u0[r] + wp * up[r];
                A2U[r] = a2l1 * um[r] + a2m * u0[r] + a2u1 * up[r];
            }
            hadi_get_block<B, G, T>(rm2, half, lane, u2);
            if constexpr (AMER == 2) {
#pragma unroll
                for (int r = 0; r < B; r++) u2[r] = fmax(u2[r], pay[r]);
            }
#pragma unroll
            for (int r = 0; r < B; r++) A2U[r] = fma(a2l2, u2[r], A2U[r]);
            hadi_get_block<B, G, T>(rp2, half, lane, u2);
            if constexpr (AMER == 2) {
#pragma unroll
                for (int r = 0; r < B; r++) u2[r] = fmax(u2[r], pay[r]);
            }
#pragma unroll
            for (int r = 0; r < B; r++) A2U[r] = fma(a2u2, u2[r], A2U[r]);
        }
        const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
        const double b1l = (lane == b1lane) ? b1val * cb1 : 0.0;  // this row's b1 entry, in the lane that owns its node
        double u0L = (double)r0[c.posL], u0R = (double)r0[c.posR];
        double m1L = (double)rm1[c.posL], m1R = (double)rm1[c.posR], p1L = (double)rp1[c.posL], p1R = (double)rp1[c.posR];
        if constexpr (AMER == 2) {
            const double payL = c.payrow[c.posL], payR = c.payrow[c.posR];
            u0L = fmax(u0L, payL); m1L = fmax(m1L, payL); p1L = fmax(p1L, payL);
            u0R = fmax(u0R, payR); m1R = fmax(m1R, payR); p1R = fmax(p1R, payR);
        }
        const double tL = wm * m1L + wz * u0L + wp * p1L;
        const double tR = wm * m1R + wz * u0R + wp * p1R;

        HADI_STAMP(1);  // LDS rows -> tt, A2U
        if constexpr (AMER == 1) hadi_get_block<B, G>(c.Li + (size_t)j * rowp, half, lane, lam);
        if constexpr (LAST) hadi_get_block<B, G>(c.b2r, half, lane, b2v);
        if constexpr (MODE == 2) {
            hadi_get_block<B, G>(c.R1i + (size_t)j * rowp, half, lane, r1v);
            hadi_get_block<B, G>(c.C2i + (size_t)j * rowp, half, lane, c2v);
        }

        // ---- Y0 (device_solver.hpp:236-250) fused with the forward sweep of the in-lane Thomas ----------
        //   x[r] = ys[r] - XL*ps[r] - X*gs[r],  XL = interface unknown of lane-1, X = own x[B-1]
        // The central FD weights follow from sum_k beta_s = sum_k delta_s = 0: B0 = -(Bm+Bp), D0 = -(Dm+Dp).
        hadi_set_prio(1);  // see hadi_set_prio
        double Bm[B], Bp[B], Dm[B], Dp[B];
        hadi_get_block<B, G>(c.coef + 0 * 64 * B * G, half, lane, Bm);
        hadi_get_block<B, G>(c.coef + 1 * 64 * B * G, half, lane, Bp);
        hadi_get_block<B, G>(c.coef + 2 * 64 * B * G, half, lane, Dm);
        hadi_get_block<B, G>(c.coef + 3 * 64 * B * G, half, lane, Dp);
        double iu[B], cp[B];
        double il_last = 0.0, im_last = 1.0, d_last = 0.0;
        double il_moved = 0.0;  // the i = 1 row's coupling to x_0 once it has been moved to the right-hand side
#pragma unroll
        for (int r = 0; r < B; r++) {
            const double uL = (r == 0) ? u0L : u0[r == 0 ? 0 : r - 1];
            const double uR = (r == B - 1) ? u0R : u0[r == B - 1 ? r : r + 1];
            const double tl = (r == 0) ? tL : tt[r == 0 ? 0 : r - 1];
            const double tr = (r == B - 1) ? tR : tt[r == B - 1 ? r : r + 1];
            const double lo = fma(v, Dm[r], qd * Bm[r]);
            const double up = fma(v, Dp[r], qd * Bp[r]);
            const double mn = -((lo + up) + half_rd);  // = -(v (Dm + Dp) + q (Bm + Bp) + r_d / 2)
            const double A1U = lo * uL + mn * u0[r] + up * uR;
            const double A0U = Bm[r] * tl - (Bm[r] + Bp[r]) * tt[r] + Bp[r] * tr;
            // Y0 = U + dt (A0U + A1U + A2U + b e_{n-1} [+ lambda]) + theta dt (b1 e_n - (A1U + b1 e_{n-1})); the b1
            // entry of this v-row (a single node) contributes b1 * cb1, cb1 = dt e_{n-1} + theta dt (e_n - e_{n-1})
            double S = A0U + A1U + A2U[r];
            if constexpr (LAST) S += b2v[r] * e_nm1;
            if constexpr (AMER != 0) S += lam[r];
            double y = fma(dt, S, u0[r]);
            y = fma(-thdt, A1U, y);
            y = fma(b1l, (r == b1r) ? 1.0 : 0.0, y);  // wave-uniform selector: one FMA with a scalar operand
            if constexpr (MODE == 1) r1v[r] = fma(-0.5 * dt, A0U, y);
            if constexpr (MODE == 2) y = fma(0.5 * dt, A0U, r1v[r]);  // A0U is A0 applied to Y2 here
            double il = -thdt * lo;
            const double im = 1.0 - thdt * mn;
            iu[r] = -thdt * up;
            if (r == 0 && lane == 0 && first_half) {  // x_0 is known: move it to the right-hand side
                y -= il * x0;
                il_moved = il;
                il = 0.0;
            }
            if (r < NB) {
                // normalised rows (x[r] + cp[r] x[r+1] = ys[r] - ps[r] XL): the back substitution is then one FMA per vector
                if (r == 0) {
                    const double inv = hadi_rcp(im);
                    cp[0] = iu[0] * inv;
                    ys[0] = y * inv;
                    ps[0] = il * inv;
                } else {
                    const double inv = hadi_rcp(fma(-il, cp[r - 1], im));
                    cp[r] = iu[r] * inv;
                    ys[r] = fma(-il, ys[r - 1], y) * inv;
                    ps[r] = -(il * ps[r - 1]) * inv;
                }
            } else {
                il_last = il;
                im_last = im;
                d_last = y;
            }
        }
        (void)il_moved;
        HADI_STAMP(2);  // coefficients + Y0 + forward Thomas
        // reduced (interface) row of this lane:  ra*X(l-1) + rb*X(l) + rcc*X(l+1) = rf [- rs * boundary value]
        double ra, rb, rcc, rf, rs = 0.0;
        const bool edge_hi = (G > 1) && !last_half && lane == 63;  // next node belongs to the other wave
        const bool edge_lo = (G > 1) && !first_half && lane == 0;  // previous node belongs to the other wave
        if constexpr (NB > 0) {
            gs[NB - 1] = cp[NB - 1];
#pragma unroll
            for (int r = NB - 2; r >= 0; r--) {
                ys[r] = fma(-cp[r], ys[r + 1], ys[r]);
                ps[r] = fma(-cp[r], ps[r + 1], ps[r]);
                gs[r] = -cp[r] * gs[r + 1];
            }
            double p0n = hadi_lane_next(ps[0]), g0n = hadi_lane_next(gs[0]), y0n = hadi_lane_next(ys[0]);
            if (edge_hi) { p0n = 0.0; g0n = 0.0; y0n = 0.0; }
            ra = -il_last * ps[NB - 1];
            rb = im_last - il_last * gs[NB - 1] - iu[B - 1] * p0n;
            rcc = -iu[B - 1] * g0n;
            rf = d_last - il_last * ys[NB - 1] - iu[B - 1] * y0n;
        } else {
            ra = il_last;
            rb = im_last;
            rcc = iu[0];
            rf = d_last;
        }
        if constexpr (G > 1) {
            if (edge_hi) { rs = iu[B - 1]; rcc = 0.0; }  // couples to t = first node of the other half
            if (edge_lo) { rs = ra; ra = 0.0; }          // couples to the last node of the other half
        }
        HADI_STAMP(3);  // backward Thomas + reduced row
        // ---- parallel cyclic reduction over the 64 interface unknowns (normalised rows) -------------
        // Lanes without a partner at distance s have ra == 0 (left) / rcc == 0 (right) by induction, so the
        // (wrapped) values they fetch are multiplied by zero: no lane masks are needed.
        if constexpr (NB == 0) {
            // One node per lane (m1 <= 64: the reference's calibration grids): the cyclic reduction IS the whole line solve,
            // so two nodes a tiny interval apart (S_0 inserted 7e-6 beside a node: off-diagonals of 1e7 against a row sum of
            // ~1) are two of its unknowns.  In the plain update the new diagonal 1 - a cL - c aR is a difference of numbers
            // that agree to 7 digits; the two nodes come out with independent errors of cond * eps each, and the NEXT
            // step multiplies their difference by the 1e7 coupling again (found by the extended-precision adjudicator,
            // oracle/heston_oracle_xp.c: fuzz seed 5 case 279, field error 1.7e-7 against 4.5e-10 for the Thomas sweep
            // of the reference).  Carrying every row's EXCESS d = 1 + a + c (diagonal dominance; known analytically,
            // 1 + theta dt r_d / 2 before normalisation) removes the cancellation: with cL = dL - 1 - aL, aR = dR - 1 - cR
            //   new excess   e  = d - a dL - c dR          (for an M-matrix row: a sum of non-negative terms)
            //   new diagonal bn = e + a aL + c cR           (likewise)
            // Same number of cross-lane fetches as the plain update (dL, dR replace cL, aR), three more VALU operations
            // per level.  The algebra holds for any signs; only the no-cancellation property needs a, c <= 0.
            hadi_set_prio(3);
            double rd = (1.0 + thdt * half_rd) - il_moved;  // row sum il + im + iu of I - theta dt A1 (il of the first node moved out)
            rb = rd - ra - rcc;                              // diagonal from the off-diagonals and the excess
            const double rinv0 = hadi_rcp(rb);
            ra *= rinv0;
            rcc *= rinv0;
            rf *= rinv0;
            rd *= rinv0;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int up_lane = (lane - s) & 63, dn_lane = (lane + s) & 63;
                double aL, dL, fL, cR, dR, fR;
                if (s == 1) {
                    aL = hadi_lane_prev(ra); dL = hadi_lane_prev(rd); fL = hadi_lane_prev(rf);
                    cR = hadi_lane_next(rcc); dR = hadi_lane_next(rd); fR = hadi_lane_next(rf);
                } else if (s == 32) {  // lane - 32 and lane + 32 are the same lane (mod 64)
                    aL = hadi_lane_get(ra, up_lane); cR = hadi_lane_get(rcc, up_lane);
                    dL = dR = hadi_lane_get(rd, up_lane); fL = fR = hadi_lane_get(rf, up_lane);
                } else {
                    aL = hadi_lane_get(ra, up_lane); dL = hadi_lane_get(rd, up_lane); fL = hadi_lane_get(rf, up_lane);
                    cR = hadi_lane_get(rcc, dn_lane); dR = hadi_lane_get(rd, dn_lane); fR = hadi_lane_get(rf, dn_lane);
                }
                const double e = fma(-rcc, dR, fma(-ra, dL, rd));
                const double bn = fma(rcc, cR, fma(ra, aL, e));
                const double rn = hadi_rcp(bn);
                rf = fma(-rcc, fR, fma(-ra, fL, rf)) * rn;
                if (s < 32) {  // the last level only needs the right-hand side
                    const double an = -(ra * aL) * rn;
                    const double cn = -(rcc * cR) * rn;
                    ra = an;
                    rcc = cn;
                    rd = e * rn;
                }
            }
        } else {
            hadi_set_prio(3);
            const double rinv0 = hadi_rcp(rb);
            ra *= rinv0;
            rcc *= rinv0;
            rf *= rinv0;
            if constexpr (G > 1) rs *= rinv0;
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const int up_lane = (lane - s) & 63, dn_lane = (lane + s) & 63;
                double aL, cL, fL, aR, cR, fR;
                if (s == 1) {  // (constant after unrolling) the first level's neighbours are one lane away
                    aL = hadi_lane_prev(ra); cL = hadi_lane_prev(rcc); fL = hadi_lane_prev(rf);
                    aR = hadi_lane_next(ra); cR = hadi_lane_next(rcc); fR = hadi_lane_next(rf);
                } else if (s == 32) {  // lane - 32 and lane + 32 are the same lane (mod 64): one fetch serves both sides
                    aL = aR = hadi_lane_get(ra, up_lane); cL = cR = hadi_lane_get(rcc, up_lane); fL = fR = hadi_lane_get(rf, up_lane);
                } else {
                    aL = hadi_lane_get(ra, up_lane); cL = hadi_lane_get(rcc, up_lane); fL = hadi_lane_get(rf, up_lane);
                    aR = hadi_lane_get(ra, dn_lane); cR = hadi_lane_get(rcc, dn_lane); fR = hadi_lane_get(rf, dn_lane);
                }
                const double bn = fma(-rcc, aR, fma(-ra, cL, 1.0));
                const double rn = hadi_rcp(bn);
                rf = fma(-rcc, fR, fma(-ra, fL, rf)) * rn;
                if constexpr (G > 1) {
                    const double sL = (s == 1) ? hadi_lane_prev(rs) : hadi_lane_get(rs, up_lane);
                    const double sR = (s == 1) ? hadi_lane_next(rs) : (s == 32) ? sL : hadi_lane_get(rs, dn_lane);
                    rs = fma(-rcc, sR, fma(-ra, sL, rs)) * rn;
                }
                if (s < 32) {  // the last level only needs the right-hand sides
                    const double an = -(ra * aL) * rn;
                    const double cn = -(rcc * cR) * rn;
                    ra = an;
                    rcc = cn;
                }
            }
        }
        hadi_set_prio(0);
        HADI_STAMP(4);  // PCR
        Ysol = rf;
        Ssol = rs;
        if constexpr (G > 1) {
            // X(l) = Ysol - bv * Ssol with bv the other half's adjacent node.  Publish what the 2x2 needs:
            //   low half, lane 63:  x_hi = A - t*Bc              (A = Ysol, Bc = Ssol; x_hi = its own X)
            //   high half, lane 0:  t = C - x_hi*D   (t = its first node = ys0 - XL ps0 - X gs0, XL = x_hi)
            if (edge_hi) {
                c.xch[8 * c.wrow + 0] = Ysol;
                c.xch[8 * c.wrow + 1] = Ssol;
            }
            if (edge_lo) {
                if constexpr (NB > 0) {
                    c.xch[8 * c.wrow + 2] = ys[0] - Ysol * gs[0];
                    c.xch[8 * c.wrow + 3] = ps[0] - Ssol * gs[0];
                } else {
                    c.xch[8 * c.wrow + 2] = Ysol;
                    c.xch[8 * c.wrow + 3] = Ssol;
                }
            }
        }
    }
    if constexpr (G > 1) {
#if defined(HADI_EMU) || defined(HADI_BLOCK_EXCHANGE)
        __syncthreads();
#else
        // Rendezvous of the TWO wavefronts of this v-row only (the other rows of the block run on): each publishes a token
        // behind its two values (same lane, so the LDS unit sees data before flag) and polls the partner's.  Both are
        // resident wavefronts of one block and `active` is the same for both, so the partner always arrives; the loop-top
        // barrier of the next iteration separates this exchange from the next use of the slots.  The poll is bounded so
        // that a logic error can never hang the GPU; running out of polls is reported through the handle's error word
        // (hadi_report) and fails the call.
        if (active) {
            int *flags = reinterpret_cast<int *>(c.xch + 8 * c.wrow + 4);
            const int token = j + 1;
            const bool publisher = (!last_half && lane == 63) || (!first_half && lane == 0);  // the lanes that wrote the values
            const bool withhold = (c.debug & HADI_DEBUG_WITHHOLD_TOKEN) && half == 1 && j == 1;  // (test hook)
            if (publisher && !withhold) __hip_atomic_store(flags + half, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            int guard = 0;
            const int polls = HADI_RENDEZVOUS_POLLS(c.debug);
            while (__hip_atomic_load(flags + (1 - half), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != token && ++guard < polls)
                __builtin_amdgcn_s_sleep(1);
            if (guard >= polls && lane == 0) hadi_report(c.err, HADI_DEVERR_RENDEZVOUS);
        }
#endif
    }
    if (active) {
        double X = Ysol, XL;
        if constexpr (G > 1) {
            const double A = c.xch[8 * c.wrow + 0], Bc = c.xch[8 * c.wrow + 1];
            const double Cc = c.xch[8 * c.wrow + 2], Dd = c.xch[8 * c.wrow + 3];
            const double xhi = (A - Bc * Cc) / (1.0 - Bc * Dd);  // last node of the low half
            const double tlo = Cc - Dd * xhi;                    // first node of the high half
            X = Ysol - (first_half ? tlo : xhi) * Ssol;
            XL = hadi_lane_prev(X);
            if (lane == 0) XL = first_half ? 0.0 : xhi;
        } else {
            XL = hadi_lane_prev(X);
        }
        // ---- Y1 -> right-hand side of the A2 solve (device_solver.hpp:254-260) and store ----------
        double yo[B];
#pragma unroll
        for (int r = 0; r < B; r++) {
            double x;
            if (r < NB) x = ys[r] - XL * ps[r] - X * gs[r];
            else x = X;
            double corr;
            if constexpr (MODE == 2) corr = c2v[r];
            else if constexpr (LAST) corr = thdt * (b2v[r] * e_n - (A2U[r] + b2v[r] * e_nm1));
            else corr = -thdt * A2U[r];
            yo[r] = x + corr;
            if constexpr (MODE == 1) c2v[r] = corr;
        }
        if constexpr (MODE == 1) {
            hadi_put_block<B, G>(c.R1i + (size_t)j * rowp, half, lane, r1v);
            hadi_put_block<B, G>(c.C2i + (size_t)j * rowp, half, lane, c2v);
        }
        hadi_put_block<B, G, T>(c.Yi + (size_t)j * rowp, half, lane, yo);
        if (lane == 0 && first_half) c.Yi[(size_t)j * rowp + c0slot] = (T)yout_c0;
        HADI_STAMP(5);  // final correction + store
    }
}

// Counted wait: at most `n` of this wavefront's youngest vector-memory operations may still be in flight.
HADI_DEV HADI_FORCEINLINE void hadi_wait_vmcnt(int n) {
#if defined(HADI_STRICT_VMCNT) && !defined(HADI_EMU)
    // Checking build (libhadi_strict.so, tests only): every counted wait becomes a full drain.  The counted waits rest on
    // hand-kept instruction counts (DMA pieces per row, stores per row); if a compiler change ever broke that bookkeeping
    // the product build would read stale ring rows while this build stays right -- tests/test_gpu_parity.py compares the
    // two bit for bit on every strip / ring shape.
    (void)n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif !defined(HADI_EMU)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;  // n >= 25: stricter than asked, still safe
    }
#else
    (void)n;
#endif
}

// NG = row tiles handled by one block (each by its own group of W*G wavefronts with its own LDS ring; the
// s-coefficient arrays are shared), PD = prefetch depth in iterations: the ring holds (PD+1)*W + 4 rows.
// Every shape runs NG = 1 (hadi_plan.h: two groups behind one barrier measured slower); PD = 1 at 8 nodes per lane
// (two 4-wave blocks per CU), 2 below.  Large batches at 8 nodes per lane use hadi_pass_a_strip instead.
// T = float: fp32-state sweep (a.U / a.Y then point to float arrays of the same element layout; European Douglas only).
template <int B, int G, int W, int NG, int PD, int AMER, int MODE = 0, class T = double>
// Minimum blocks per CU of the shared-ring kernel at 4 nodes per lane: 3 (170 VGPRs) -- at 4 (128 VGPRs) the American
// variants spill into scratch inside the row loop (measured, 256x128 x512 American puts: 0.0966 -> 0.0942 ms per launch)
#ifndef HADI_RING_OCC_B4
#define HADI_RING_OCC_B4 3
#endif
__global__ void __launch_bounds__(64 * W * G * NG, (B >= 8 ? 2 : B == 4 ? HADI_RING_OCC_B4 : 4)) hadi_pass_a(HadiSweepArgs a, int n) {
    static_assert(sizeof(T) == 8 || (!AMER && MODE == 0), "the fp32-state sweep covers the European Douglas step only");
    HADI_DYN_SMEM(double, smem);
    constexpr int RING = (PD + 1) * W + 4;
    constexpr int NT = 64 * W * G * NG;
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int grp = wave / (W * G), wv = wave - grp * (W * G);
    const int wrow = wv / G, half = wv - wrow * G;
    const int tblocks = (a.ntiles + NG - 1) / NG;  // blocks per instance
    const int total = a.n_inst * tblocks;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= total) return;
    const int inst = logical / tblocks, tb = logical - inst * tblocks;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int nrows = a.L.nrows, npad = a.L.nrows_pad, rowp = a.L.rowp;
    const int tile = tb * NG + grp;
    const int j0 = tile * a.R;  // may be >= nrows for the last block's spare group: that group only joins barriers
    const int j1 = (j0 + a.R < nrows) ? j0 + a.R : nrows;

    HadiRowCtxT<T> c;
    c.lane = lane;
    c.half = half;
    c.wrow = wrow;
    c.rowp = rowp;
    c.dt = ip.dt; c.thdt = ip.thdt; c.qd = ip.q; c.half_rd = ip.half_rd;
    c.hr0 = ip.hr0; c.inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    c.e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
    c.e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
    const T *__restrict__ Ub = reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride;
    c.Yi = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride;
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.b2r = a.b2row + (size_t)inst * rowp;
    c.R1i = MODE ? a.R1 + (size_t)inst * a.L.inst_stride : nullptr;
    c.C2i = MODE ? a.C2 + (size_t)inst * a.L.inst_stride : nullptr;
    c.j0 = j0;
    c.err = a.err; c.debug = a.debug;
    constexpr int c0slot = 64 * B * G;
    // storage positions of the s-neighbours of this lane's block (node before its first, node after its
    // last).  Before i = 1 comes the i = 0 slot; after the row's last node comes a pad slot (always 0).
    {
        const int ifirst = 1 + 64 * B * half + B * lane;
        if constexpr (sizeof(T) == 4) {
            c.posL = hadi_pos_f32(B, G, ifirst - 1);
            c.posR = (ifirst + B <= 64 * B * G) ? hadi_pos_f32(B, G, ifirst + B) : c0slot + 1;
        } else {
            c.posL = hadi_pos(B, G, ifirst - 1);
            c.posR = (ifirst + B <= 64 * B * G) ? hadi_pos(B, G, ifirst + B) : c0slot + 1;
        }
    }

    // LDS: [NG rings of RING rows of T] [4 coefficient arrays of 64*B*G] [NG*W*8 exchange] [NG compact row tables]
    T *ring = reinterpret_cast<T *>(smem) + (size_t)grp * RING * rowp;
    double *coef = reinterpret_cast<double *>(reinterpret_cast<T *>(smem) + (size_t)NG * RING * rowp);
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B * G;
        for (int e = threadIdx.x; e < 4 * 64 * B * G; e += NT) coef[e] = sc[e];
    }
    c.coef = coef;
    c.xch = coef + 4 * 64 * B * G + grp * 8 * W;  // per v-row: 4 exchange values + the two rendezvous tokens
    if (threadIdx.x < 8 * W * NG) coef[4 * 64 * B * G + threadIdx.x] = 0.0;  // (tokens start at 0; the first loop barrier publishes this)
    {
        double *rtab = coef + 4 * 64 * B * G + NG * 8 * W + (size_t)grp * a.R * HADI_RCL;
        const double *__restrict__ rg = a.rowc + ((size_t)inst * nrows + j0) * HADI_RC;
        const int tl = threadIdx.x - grp * 64 * W * G;
        for (int e = tl; e < (j1 - j0) * HADI_RCL; e += 64 * W * G) rtab[e] = rg[(e / HADI_RCL) * HADI_RC + e % HADI_RCL];
        c.rowc = rtab;
        c.payrow = nullptr; c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
        if constexpr (AMER == 2) {  // payoff row (v-row 0 of the packed payoff; it depends on s only) after the tables
            double *prow = coef + 4 * 64 * B * G + NG * 8 * W + (size_t)NG * a.R * HADI_RCL;
            const double *__restrict__ pg = a.U0 + (size_t)inst * a.L.inst_stride;
            for (int e = threadIdx.x; e < rowp; e += NT) prow[e] = pg[e];
            c.payrow = prow;
            c.inv_dt = 1.0 / ip.dt;
            const int e1 = a.L.m1 - 1;  // node i = m1 is element m1-1 of the row's 64*B*G interior nodes
            if (e1 / (64 * B) == half) {
                c.m1_lane = (e1 - half * 64 * B) / B;
                c.m1_r = (e1 - half * 64 * B) % B;
            }
        }
    }

    const int iters = (j1 > j0) ? (j1 - j0 + W - 1) / W : 0;  // this group's iterations
    const int iters_all = (a.R + W - 1) / W;                   // every group of the block runs this many barriers
    auto slot = [&](int jj) { return ring + (size_t)((jj + 4 * RING) % RING) * rowp; };
    // fetch returns the number of vector-memory instructions it issued
    auto fetch = [&](int jj) -> int {
        const bool exists = jj >= 0 && jj < npad;
        if constexpr ((64 * B * G + HADI_ROW_PAD(B, (int)sizeof(T))) % (16 / (int)sizeof(T)) == 0)
            hadi_row_to_lds_fixed<B, T, G>(Ub + (ptrdiff_t)jj * rowp, slot(jj), lane, exists);
        else
            hadi_row_to_lds(Ub + (size_t)jj * rowp, slot(jj), rowp, lane, exists);
        return exists ? hadi_row_dma_count<T>(rowp) : 0;
    };
    // prologue: rows of iterations 0 .. PD-1
    if (iters > 0)
        for (int rr = wv; rr < PD * W + 4; rr += W * G) fetch(j0 - 2 + rr);

    // Vector-memory operations retire in issue order.  ya[k] = (lower bound of the) number of operations this
    // wavefront issued after the DMA batch that iteration it+k needs, so hadi_wait_vmcnt(ya[0]) retires that
    // batch and leaves younger batches and result stores in flight.
    int ya[PD];
#pragma unroll
    for (int k = 0; k < PD; k++) ya[k] = 0;
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c.stamp_acc_ = stamp_store_;
#endif
    HADI_STAMP_DECL(stamp_store_)
    HADI_STAMP(8);  // prologue
    for (int it = 0; it < iters_all; it++) {
        const int J = j0 + it * W;
        hadi_wait_vmcnt(ya[0]);
        __syncthreads();  // this iteration's rows have landed; everyone is done with the rows replaced below
        HADI_STAMP(9);  // barrier wait (incl. DMA drain)
        int z = 0;
        if (it + PD < iters && wv < W) z = fetch(J + PD * W + 2 + wv);
#pragma unroll
        for (int k = 0; k + 1 < PD; k++) ya[k] = ya[k + 1] + z;
        ya[PD - 1] = 0;
        if constexpr (PD == 1) ya[0] = 0;
        const int j = J + wrow;
        const bool active = it < iters && j < j1;
        if constexpr (G == 1) {
            if (!active) continue;
        }
        if (j == nrows - 1)
            hadi_row_step<B, G, AMER, true, MODE, T>(c, active, j, slot(j - 2), slot(j - 1), slot(j), slot(j + 1), slot(j + 2));
        else
            hadi_row_step<B, G, AMER, false, MODE, T>(c, active, j, slot(j - 2), slot(j - 1), slot(j), slot(j + 1), slot(j + 2));
        if (active) {  // B/2 (one for B = 1) vector stores of the block; the i = 0 store is not counted (lower bound)
#pragma unroll
            for (int k = 0; k < PD; k++) ya[k] += hadi_put_block_stores<B, T>();
        }
        HADI_STAMP(10);  // whole row step (+ fetch issue)
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (lane == 0)
        for (int k = 0; k < 12; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// ------------------------------------------------------------------------------------------------
// Strip row pass (8 nodes per lane, one wavefront per v-row: 256 < m1 <= 512).  Same arithmetic as hadi_row_step,
// different data movement: every wavefront owns a strip of RS consecutive v-rows and walks down it ALONE -- no
// barrier in the loop.  The rows j-2, j-1, j of the stencil stay in registers from the previous steps, rows j+1 and
// j+2 sit in the wavefront's private 4-slot LDS ring, rows j+3 and j+4 are in flight (LDS-DMA issued by this
// wavefront, retired by its own counted vmcnt wait).  The s-neighbours of a lane's block come from the adjacent
// lanes (ds_bpermute), the i = 0 column from the row's extra slot, the row's table entry through the scalar cache
// into SGPRs.  Against the shared-ring kernel:
// no block-wide barrier (its wait was ~40 % of a wavefront's time there), twice the rows in flight per CU, a quarter
// of the LDS reads; the price is 4 halo rows per strip read again (mostly L2 hits).
// American P representation on strips: from this many nodes per lane on, the raw P of row j is read again from its ring slot
// (kept one step longer) instead of being carried in registers.  At 8 nodes per lane that ends the spilling (256 VGPRs + 12
// spilled -> 228; 512x256 x256 American: row pass 0.140 -> 0.121 ms per launch); at 4 the kernel fits either way and the
// shorter prefetch costs more than the registers gain (256x128 x512 American puts: 0.0841 -> 0.0866).
#ifndef HADI_STRIP_CREG_MAX_B
#define HADI_STRIP_CREG_MAX_B 4  // strips of at most this many nodes per lane keep the s-coefficients in registers
#endif
#ifndef HADI_AMP_KEEP_MIN_B
#define HADI_AMP_KEEP_MIN_B 8
#endif
template <class T>
struct HadiStripCtxT {
    const double *coef;  // LDS: Bm, Bp, Dm, Dp, each 64*B doubles in row layout
    T *Yi;               // instance base of Y
    const double *Li;    // instance base of lambda_bar (American)
    const double *b2r;   // instance b2 row (global)
    int lane, rowp;
    double dt, thdt, e_nm1, e_n;
    double c1, kap;           // 1 + theta dt r_d / 2, (1 - theta) / theta
    double hr0, inv0;         // i = 0 row of A1: reaction term (0 for the call) and 1 / (1 + theta dt hr0)
    double inv_dt;            // P representation: 1 / dt and the (lane, slot) of the s_max node
    int m1_lane, m1_r;
    int half;                 // G = 2: which half of the row this wavefront owns (0: nodes 1..64B, 1: the rest)
    double *xch;              // G = 2: LDS exchange of the wavefront pair, [2 row parities][4 values + 2 tokens + 2 spare]
    int *err;                 // HadiSweepArgs.err / .debug (G = 2 rendezvous)
    int debug;
    HADI_STAMP_ACC
};

// Pair rendezvous flags in LDS (G = 2 strips and the shared ring): release store / acquire load at workgroup scope.
HADI_DEV HADI_FORCEINLINE void hadi_flag_store(int *f, int v) {
#if defined(HADI_EMU)
    __atomic_store_n(f, v, __ATOMIC_RELEASE);
#else
    __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
HADI_DEV HADI_FORCEINLINE int hadi_flag_load(int *f) {
#if defined(HADI_EMU)
    return __atomic_load_n(f, __ATOMIC_ACQUIRE);
#else
    return __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// AMER: 0 European, 1 American with the explicit (U, lambda_bar) pair (lambda_bar loaded here), 2 American in the P
// representation: the caller rebuilt U = max(P, U_0) on the five rows and hands over the raw P of row j (p_raw) and
// lambda_bar of the i = 0 column; lambda_bar = max(0, (U_0 - P)/dt) = (U - P)/dt is formed here, right before the sweep
// that consumes it (formed by the caller it stayed live across the explicit operators and the kernel spilled).
// G = 2 (512 < m1 <= 1024, European): the row is shared by a PAIR of wavefronts, each owning one half.  eb / e0 / ea are
// the values, on the rows behind / at / ahead of j, of the one node next to this half that belongs to the partner; the
// tridiagonal system is split at the boundary exactly as in hadi_row_step (second right-hand side through the cyclic
// reduction, 2x2 system exchanged through LDS), with a rendezvous of the two wavefronts only.
template <int B, int AMER, bool LAST, class T = double, int G = 1, int CREG = 0>
HADI_DEV HADI_FORCEINLINE void hadi_strip_step(const HadiStripCtxT<T> &c, int j, const double (&rt)[HADI_RCL],
                                               const double (&um2)[B], const double (&um1)[B], const double (&u0)[B],
                                               const double (&up1)[B], const double (&up2)[B], double c0m2, double c0m1,
                                               double c00, double c0p1, double c0p2, const double (&p_raw)[B],
                                               double lamc0_in, const T *next_row, double (&u_next)[B],
                                               double eb = 0.0, double e0 = 0.0, double ea = 0.0, const T *raw_row = nullptr,
                                               const double *pay_row = nullptr, const double *cf = nullptr) {
    static_assert(G == 1 || (G == 2 && (AMER == 0 || sizeof(T) == 8)), "paired strips: American sweeps with the fp64 state only");
    const int lane = c.lane, rowp = c.rowp;
    const int half = (G > 1) ? c.half : 0;
    // P representation: where the raw P of row j comes from -- its ring slot again (8 nodes per lane, one wavefront per row:
    // the kernel keeps that slot one step longer) or the caller's registers (the 3-slot ring of the paired strips has no
    // slot to spare)
    constexpr bool RAW_FROM_RING = (B >= HADI_AMP_KEEP_MIN_B && G == 1);
    // P representation on paired strips: u0 ARRIVES RAW (P of row j) and U = max(P, U_0) is formed where it is used, from
    // the payoff row in LDS (pay_row) -- once for the explicit operators, once more pair by pair inside the forward sweep,
    // where lambda_bar = (U - P)/dt falls out of the same two values.  Carrying U and P (or lambda_bar) side by side is
    // 16 registers more than the 256 this kernel has (hipcc spilled 8 of them into the row loop).
    constexpr bool RAW_U0 = (AMER == 2 && !RAW_FROM_RING && G == 2);
    const bool first_half = (half == 0), last_half = (half == G - 1);
    constexpr int c0slot = 64 * B * G;
    constexpr int NB = B - 1;
    HADI_STAMP_DECL(c.stamp_acc_)
    const double dt = c.dt, thdt = c.thdt, c1 = c.c1, kap = c.kap, e_nm1 = c.e_nm1, e_n = c.e_n;
    // rt = the entries RC_L2 .. RC_WPS of the row's table entry (HADI_SRC0): wm, wz, wp are the SCALED A0 v-weights,
    // -w / (theta dt (r_d - r_f)), to go with the scaled s-coefficient arrays (below)
    const double vth = rt[RC_VTH - HADI_SRC0];
    const double wm = rt[RC_WMS - HADI_SRC0], wz = rt[RC_WZS - HADI_SRC0], wp = rt[RC_WPS - HADI_SRC0];
    const double a2l2 = rt[RC_L2 - HADI_SRC0], a2l1 = rt[RC_L1 - HADI_SRC0], a2m = rt[RC_M - HADI_SRC0], a2u1 = rt[RC_U1 - HADI_SRC0],
                 a2u2 = rt[RC_U2 - HADI_SRC0];
    const double b1val = rt[RC_B1VAL - HADI_SRC0];
    const int b1raw = (int)rt[RC_B1COL - HADI_SRC0];
    const bool b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;  // (two entries on one v-row: m2 > m1 only)
    const int b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    const int b1e = b1col - 1;
    const int b1half = (G > 1 && b1col >= 1) ? b1e / (64 * B) : 0;  // which wavefront of the pair holds the b1 node
    const int b1el = b1e - b1half * 64 * B;
    const int b1lane = (b1col >= 1 && b1half == half) ? b1el / B : -1;
    const int b1r = b1el - (b1el / B) * B;

    // ---- column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act) ----------------
    const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
    const double b1c0 = b1_at0 ? b1val : 0.0;
    const double b2c0 = LAST ? c.b2r[c0slot] : 0.0;
    const double lamc0 = (AMER == 1) ? c.Li[(size_t)j * rowp + c0slot] : (AMER == 2) ? lamc0_in : 0.0;
    const double a1c0 = -c.hr0 * c00;  // A1 row 0: empty for the call (hr0 = 0), the reaction term for the put
    double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
    y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
    const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
    const double x0 = y0c0 * c.inv0;  // A1 row 0 is decoupled: the identity for the call (hes_a1_kernels.hpp:56-61)
    const double yout_c0 = x0 + c2c0;

    // ---- explicit operators (same evaluation order as hadi_row_step) -------------------------------------
    double tt[B], A2U[B];
    double u0_first = u0[0], u0_last = u0[B - 1];
    if constexpr (RAW_U0) {
        // (the payoff pair by pair, behind a compiler barrier: as one 16-register block it stayed live from the caller's
        // max operations to the end of this loop and the kernel spilled 16 registers into the row loop)
        double payx[B];
#pragma unroll
        for (int r = 0; r < B; r++) {
            if ((r & 1) == 0) {
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
                const double2 pp = *reinterpret_cast<const double2 *>(pay_row + (r >> 1) * 128 * G + 128 * half + 2 * lane);
                payx[r] = pp.x; payx[r + 1 < B ? r + 1 : r] = pp.y;
            }
            const double um = fmax(u0[r], payx[r]);
            tt[r] = wm * um1[r] + wz * um + wp * up1[r];
            A2U[r] = a2l1 * um1[r] + a2m * um + a2u1 * up1[r];
            if (r == 0) u0_first = um;
            if (r == B - 1) u0_last = um;
        }
    } else {
#pragma unroll
        for (int r = 0; r < B; r++) {
            tt[r] = wm * um1[r] + wz * u0[r] + wp * up1[r];
            A2U[r] = a2l1 * um1[r] + a2m * u0[r] + a2u1 * up1[r];
        }
    }
    // The second neighbours j-2, j+2 enter A2 only on the upwind rows (v_j > 1: hes_a2_shuffled_kernels.hpp:131-140) and on
    // row 0 (the gamma stencil): three rows in four have both weights zero -- a wave-uniform branch (the weights sit in
    // SGPRs) around the 2 B FMAs.  fma(0, u, A) = A: bit-identical.
    if (a2l2 != 0.0 || a2u2 != 0.0) {
#pragma unroll
        for (int r = 0; r < B; r++) A2U[r] = fma(a2l2, um2[r], A2U[r]);
#pragma unroll
        for (int r = 0; r < B; r++) A2U[r] = fma(a2u2, up2[r], A2U[r]);
    }
    // s-neighbours of the block: last node of lane-1, first node of lane+1; lane 0 borders i = 0, lane 63 the pad (0)
    double u0L = hadi_lane_prev(u0_last), tL = hadi_lane_prev(tt[B - 1]);
    double u0R = hadi_lane_next(u0_first), tR = hadi_lane_next(tt[0]);
    if (lane == 0 && first_half) {
        u0L = c00;
        tL = wm * c0m1 + wz * c00 + wp * c0p1;
    }
    if constexpr (G > 1) {  // the node across the pair boundary
        const double te = wm * eb + wz * e0 + wp * ea;
        if (lane == 0 && !first_half) { u0L = e0; tL = te; }
        if (lane == 63 && !last_half) { u0R = e0; tR = te; }
    }
    // (lane 63 of the last half: hadi_lane_next delivered the pad's zeros already)
    const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
    const double b1l = (lane == b1lane) ? b1val * cb1 : 0.0;  // this row's b1 entry, in the lane that owns its node

    double lam[B], b2v[B];
    if constexpr (AMER == 1) hadi_get_block<B, G>(c.Li + (size_t)j * rowp, half, lane, lam);
    if constexpr (AMER == 2) {
        // 8 nodes per lane: the raw P of row j is read again from its ring slot, which stays intact through this step (the
        // kernel keeps one slot behind the prefetch for it) -- carried in 16 more registers the kernel spilled
        double praw8[B];
        if constexpr (RAW_FROM_RING) hadi_get_block<B, 1, T>(raw_row, 0, lane, praw8);
        else {
#pragma unroll
            for (int r = 0; r < B; r++) praw8[r] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < B; r++) {
            if constexpr (RAW_U0) lam[r] = 0.0;  // (formed inside the sweep)
            else {
                lam[r] = (u0[r] - (RAW_FROM_RING ? praw8[r] : p_raw[r])) * c.inv_dt;  // exactly max(0, (U_0 - P) / dt): U = max(P, U_0)
                if (lane == c.m1_lane && r == c.m1_r) lam[r] = 0.0;  // s_max keeps lambda_bar = 0, as in hadi_row_step
            }
        }
    }
    if constexpr (LAST) hadi_get_block<B, G>(c.b2r, half, lane, b2v);

    // ---- Y0 (device_solver.hpp:236-250) fused with the forward sweep of the in-lane Thomas ----------
    // The s-coefficients are read pair by pair inside the sweep (a compiler barrier keeps hipcc from hoisting all 16
    // LDS reads to the top: 32 live doubles there are what pushed this kernel into scratch).
    hadi_set_prio(1);
    double Bm[B], Bp[B], Dm[B], Dp[B];
    double ys[B], ps[B], gs[B], iu[B], cp[B];
    double il_last = 0.0, im_last = 1.0, d_last = 0.0;
    double um[B];  // RAW_U0: U = max(P, U_0) of this row, two nodes ahead of the sweep
    if constexpr (RAW_U0) {
        const double2 pp = *reinterpret_cast<const double2 *>(pay_row + 128 * half + 2 * lane);
        um[0] = fmax(u0[0], pp.x);
        um[1] = fmax(u0[1], pp.y);
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
        if ((r & 1) == 0) {
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
            const int q = r >> 1;
            const int co = q * 128 * G + 128 * half + 2 * lane;
            if constexpr (RAW_U0) {
                if (r + 2 < B) {
                    const double2 pp = *reinterpret_cast<const double2 *>(pay_row + co + 128 * G);
                    um[r + 2 < B ? r + 2 : 0] = fmax(u0[r + 2 < B ? r + 2 : 0], pp.x);
                    um[r + 3 < B ? r + 3 : 0] = fmax(u0[r + 3 < B ? r + 3 : 0], pp.y);
                }
            }
            if constexpr (CREG >= 2) {  // (8 nodes per lane: the two diffusion arrays, or only the second, from registers)
                const double2 t0 = *reinterpret_cast<const double2 *>(c.coef + 0 * 64 * B * G + co);
                const double2 t1 = *reinterpret_cast<const double2 *>(c.coef + 1 * 64 * B * G + co);
                Bm[r] = t0.x; Bm[r + 1] = t0.y;
                Bp[r] = t1.x; Bp[r + 1] = t1.y;
                if constexpr (CREG == 2) { Dm[r] = cf[2 * B + r]; Dm[r + 1] = cf[2 * B + r + 1]; }
                else { const double2 t2 = *reinterpret_cast<const double2 *>(c.coef + 2 * 64 * B * G + co); Dm[r] = t2.x; Dm[r + 1] = t2.y; }
                Dp[r] = cf[3 * B + r]; Dp[r + 1] = cf[3 * B + r + 1];
            } else if constexpr (CREG) {
                // 2 and 4 nodes per lane: the lane's s-coefficients stay in registers for the whole strip (cf: [4][B], the
                // kernel has the registers to spare) -- the LDS pipe, shared by all wavefronts of the CU, is what these row
                // widths run out of first (hadi_pass_a_strip)
                Bm[r] = cf[0 * B + r]; Bm[r + 1] = cf[0 * B + r + 1];
                Bp[r] = cf[1 * B + r]; Bp[r + 1] = cf[1 * B + r + 1];
                Dm[r] = cf[2 * B + r]; Dm[r + 1] = cf[2 * B + r + 1];
                Dp[r] = cf[3 * B + r]; Dp[r + 1] = cf[3 * B + r + 1];
            } else {
                const double2 t0 = *reinterpret_cast<const double2 *>(c.coef + 0 * 64 * B * G + co);
                const double2 t1 = *reinterpret_cast<const double2 *>(c.coef + 1 * 64 * B * G + co);
                const double2 t2 = *reinterpret_cast<const double2 *>(c.coef + 2 * 64 * B * G + co);
                const double2 t3 = *reinterpret_cast<const double2 *>(c.coef + 3 * 64 * B * G + co);
                Bm[r] = t0.x; Bm[r + 1] = t0.y;
                Bp[r] = t1.x; Bp[r + 1] = t1.y;
                Dm[r] = t2.x; Dm[r + 1] = t2.y;
                Dp[r] = t3.x; Dp[r + 1] = t3.y;
            }
        }
        const double u0r = RAW_U0 ? um[r] : u0[r];
        const double uL = (r == 0) ? u0L : (RAW_U0 ? um[r == 0 ? 0 : r - 1] : u0[r == 0 ? 0 : r - 1]);
        const double uR = (r == B - 1) ? u0R : (RAW_U0 ? um[r == B - 1 ? r : r + 1] : u0[r == B - 1 ? r : r + 1]);
        const double tl = (r == 0) ? tL : tt[r == 0 ? 0 : r - 1];
        const double tr = (r == B - 1) ? tR : tt[r == B - 1 ? r : r + 1];
        if constexpr (RAW_U0) {
            lam[r] = (u0r - u0[r]) * c.inv_dt;  // exactly max(0, (U_0 - P) / dt)
            if (lane == c.m1_lane && r == c.m1_r) lam[r] = 0.0;
        }
        // I - theta dt A1 directly (theta dt v comes with the row's table entry): il, im, iu; theta dt A1 U from the
        // same three; Y0 - theta dt A1 U = U + dt (A0 U + A2 U + ...) + (1 - theta)/theta (theta dt A1 U).
        // Bm, Bp hold E = -theta dt (r_d - r_f) s beta_s (scaled while the block copied the arrays to LDS): the convection
        // part of il / iu costs no multiplication, and A0 U = s beta_s (x) [w u] = E (x) [(-w / (theta dt (r_d - r_f))) u]
        // comes out of the same arrays with the scaled v-weights of the row table -- 2 operations per node fewer.
        double il = fma(-vth, Dm[r], Bm[r]);
        iu[r] = fma(-vth, Dp[r], Bp[r]);
        const double sm = il + iu[r];
        const double im = c1 - sm;  // 1 + theta dt (lo + up + r_d / 2)
        // theta dt A1 U = -il uL - iu uR + (1 - im) u0   (c1 - c2 = 1)
        const double T1 = fma(-iu[r], uR, fma(-il, uL, fma(-im, u0r, u0r)));
        const double A0U = Bm[r] * tl - (Bm[r] + Bp[r]) * tt[r] + Bp[r] * tr;
        double S = A0U + A2U[r];
        if constexpr (LAST) S += b2v[r] * e_nm1;
        if constexpr (AMER) S += lam[r];
        double y = fma(dt, S, u0r);
        y = fma(kap, T1, y);
        y = fma(b1l, (r == b1r) ? 1.0 : 0.0, y);  // wave-uniform selector: one FMA with a scalar operand (a scalar branch
                                                  // around a single add measured slower: 0.1108 vs 0.1099 ms per launch)
        if (r == 0 && lane == 0 && first_half) {  // x_0 is known: move it to the right-hand side
            y -= il * x0;
            il = 0.0;
        }
        if (r < NB) {
            // normalised rows (x[r] + cp[r] x[r+1] = ys[r] - ps[r] XL): the back substitution is then one FMA per vector
            if (r == 0) {
                const double inv = hadi_rcp(im);
                cp[0] = iu[0] * inv;
                ys[0] = y * inv;
                ps[0] = il * inv;
            } else {
                const double inv = hadi_rcp(fma(-il, cp[r - 1], im));
                cp[r] = iu[r] * inv;
                ys[r] = fma(-il, ys[r - 1], y) * inv;
                ps[r] = -(il * ps[r - 1]) * inv;
            }
        } else {
            il_last = il;
            im_last = im;
            d_last = y;
        }
    }
    HADI_STAMPC(26);  // explicit operators + Y0 + forward Thomas
    // reduced (interface) row of this lane
    double ra, rb, rcc, rf, rs = 0.0;
    const bool edge_hi = (G > 1) && !last_half && lane == 63;  // next node belongs to the partner wavefront
    const bool edge_lo = (G > 1) && !first_half && lane == 0;  // previous node belongs to the partner wavefront
    {
        gs[NB - 1] = cp[NB - 1];
#pragma unroll
        for (int r = NB - 2; r >= 0; r--) {
            ys[r] = fma(-cp[r], ys[r + 1], ys[r]);
            ps[r] = fma(-cp[r], ps[r + 1], ps[r]);
            gs[r] = -cp[r] * gs[r + 1];
        }
        double p0n = hadi_lane_next(ps[0]), g0n = hadi_lane_next(gs[0]), y0n = hadi_lane_next(ys[0]);
        if constexpr (G > 1) {
            if (edge_hi) { p0n = 0.0; g0n = 0.0; y0n = 0.0; }
        }
        ra = -il_last * ps[NB - 1];
        rb = im_last - il_last * gs[NB - 1] - iu[B - 1] * p0n;
        rcc = -iu[B - 1] * g0n;
        rf = d_last - il_last * ys[NB - 1] - iu[B - 1] * y0n;
        if constexpr (G > 1) {
            if (edge_hi) { rs = iu[B - 1]; rcc = 0.0; }  // couples to t = first node of the partner's half
            if (edge_lo) { rs = ra; ra = 0.0; }          // couples to the last node of the partner's half
        }
    }
    HADI_STAMPC(27);  // backward Thomas + reduced row
    // ---- parallel cyclic reduction over the 64 interface unknowns (normalised rows, see hadi_row_step) ----
    {
        hadi_set_prio(3);
        const double rinv0 = hadi_rcp(rb);
        ra *= rinv0;
        rcc *= rinv0;
        rf *= rinv0;
        if constexpr (G > 1) rs *= rinv0;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int up_lane = (lane - s) & 63, dn_lane = (lane + s) & 63;
            double aL, cL, fL, aR, cR, fR;
            if (s == 1) {  // (constant after unrolling) the first level's neighbours are one lane away
                aL = hadi_lane_prev(ra); cL = hadi_lane_prev(rcc); fL = hadi_lane_prev(rf);
                aR = hadi_lane_next(ra); cR = hadi_lane_next(rcc); fR = hadi_lane_next(rf);
            } else if (s == 32) {  // lane - 32 and lane + 32 are the same lane (mod 64): one fetch serves both sides
                aL = aR = hadi_lane_get(ra, up_lane); cL = cR = hadi_lane_get(rcc, up_lane); fL = fR = hadi_lane_get(rf, up_lane);
            } else {
                aL = hadi_lane_get(ra, up_lane); cL = hadi_lane_get(rcc, up_lane); fL = hadi_lane_get(rf, up_lane);
                aR = hadi_lane_get(ra, dn_lane); cR = hadi_lane_get(rcc, dn_lane); fR = hadi_lane_get(rf, dn_lane);
            }
            const double bn = fma(-rcc, aR, fma(-ra, cL, 1.0));
            const double rn = hadi_rcp(bn);
            rf = fma(-rcc, fR, fma(-ra, fL, rf)) * rn;
            if constexpr (G > 1) {  // the second right-hand side (coupling to the partner's boundary node)
                const double sL = (s == 1) ? hadi_lane_prev(rs) : hadi_lane_get(rs, up_lane);
                const double sR = (s == 1) ? hadi_lane_next(rs) : (s == 32) ? sL : hadi_lane_get(rs, dn_lane);
                rs = fma(-rcc, sR, fma(-ra, sL, rs)) * rn;
            }
            if (s < 32) {
                const double an = -(ra * aL) * rn;
                const double cn = -(rcc * cR) * rn;
                ra = an;
                rcc = cn;
            }
        }
    }
    HADI_STAMPC(28);  // PCR
    hadi_set_prio(0);
    // the next row (this step's "row ahead") again from its ring slot, intact until the next step: issued here so that the
    // read flies during the final combination and the stores instead of being waited for at the end of the step
    hadi_get_block<B, G, T>(next_row, half, lane, u_next);
    double X = rf, XL;
    if constexpr (G > 1) {
        // X(l) = rf - bv rs with bv the partner's boundary node.  Publish what the 2x2 system needs (hadi_row_step):
        //   low half, lane 63:  x_hi = A - t Bc   (A = rf, Bc = rs; x_hi = its own X)
        //   high half, lane 0:  t = C - x_hi D    (t = its first node = ys0 - XL ps0 - X gs0, XL = x_hi)
        // into the buffer of this row's parity, then the token behind the values (same lane: the LDS unit sees data before
        // flag).  The partner walks the same strip in the same direction, so it always arrives; it can be at most one row
        // away, hence two buffers are enough.  The poll is bounded (a logic error must not hang the GPU); running out of
        // polls is reported through the handle's error word (hadi_report) and fails the call.
        double *xb = c.xch + 8 * (j & 1);
        int *flags = reinterpret_cast<int *>(xb + 4);
        const int token = j + 1;
        const bool withhold = (c.debug & HADI_DEBUG_WITHHOLD_TOKEN) && half == 1 && j == 1;  // (test hook)
        if (edge_hi) {
            xb[0] = rf;
            xb[1] = rs;
            hadi_flag_store(flags + 0, token);
        }
        if (edge_lo) {
            xb[2] = ys[0] - rf * gs[0];
            xb[3] = ps[0] - rs * gs[0];
            if (!withhold) hadi_flag_store(flags + 1, token);
        }
        hadi_wave_rendezvous();  // (emulator: this wavefront's own publisher lane has written)
        int guard = 0;
        const int polls = HADI_RENDEZVOUS_POLLS(c.debug);
        while (hadi_flag_load(flags + (1 - half)) != token && ++guard < polls) {
#if defined(HADI_EMU)
            sched_yield();
#else
            __builtin_amdgcn_s_sleep(1);
#endif
        }
        if (guard >= polls && lane == 0) hadi_report(c.err, HADI_DEVERR_RENDEZVOUS);
        const double A = xb[0], Bc = xb[1], Cc = xb[2], Dd = xb[3];
        const double xhi = (A - Bc * Cc) / (1.0 - Bc * Dd);  // last node of the low half
        const double tlo = Cc - Dd * xhi;                    // first node of the high half
        X = rf - (first_half ? tlo : xhi) * rs;
        XL = hadi_lane_prev(X);
        if (lane == 0 && !first_half) XL = xhi;
    } else {
        XL = hadi_lane_prev(X);
    }
    // ---- Y1 -> right-hand side of the A2 solve (device_solver.hpp:254-260) and store ----------
    double yo[B];
#pragma unroll
    for (int r = 0; r < B; r++) {
        double x;
        if (r < NB) x = ys[r] - XL * ps[r] - X * gs[r];
        else x = X;
        double corr;
        if constexpr (LAST) corr = thdt * (b2v[r] * e_n - (A2U[r] + b2v[r] * e_nm1));
        else corr = -thdt * A2U[r];
        yo[r] = x + corr;
    }
    // Plain global stores on purpose.  Raw BUFFER stores here (SGPR row offset, one 32-bit lane offset: two VGPRs and the
    // 64-bit address arithmetic saved, 0.5 % faster) were tried in round 2 and are WRONG for this kernel: the counted vmcnt
    // waits rely on vector-memory operations retiring in issue order, which holds among GLOBAL operations (the LDS-DMA loads
    // and these stores) but not between MUBUF and GLOBAL ones -- with buffer stores the counter reached its target while a
    // DMA piece was still in flight and the next step read a stale ring row (caught by the libhadi_strict.so comparison
    // and the oracle tests at 2 and 8 nodes per lane).
    hadi_put_block<B, G, T>(c.Yi + (size_t)j * rowp, half, lane, yo);
    if (lane == 0 && first_half) c.Yi[(size_t)j * rowp + c0slot] = (T)yout_c0;
    HADI_STAMPC(29);  // final correction + store issue
}

// LDS: [HADI_STRIP_WAVES wavefronts][4 ring slots][rowp] + the 4 s-coefficient arrays.  Grid = n_inst * sblocks blocks.
// T = float: fp32-state sweep (European only), as in hadi_pass_a.
// G = 2 (512 < m1 <= 1024, European): the 8 wavefronts form 4 PAIRS, each pair walks one strip, wavefront h of the pair owns
// half h of every row (its own pieces of the pair's ring slot, fetched by its own LDS-DMA and retired by its own counted
// wait -- no wavefront ever reads ring data its partner fetched, except the one boundary node, see below).  LDS:
// [4 pairs][NS slots][rowp] + 4 coefficient arrays of 1024 + the pairs' exchange buffers; with an fp64 state only NS = 3
// slots fit the 160 KB (rows j+1, j+2 landed, j+3 in flight), with an fp32 state 4 as above.
template <int B, int AMER, class T = double, int G = 1>
#ifndef HADI_STRIP_OCC_B4
#define HADI_STRIP_OCC_B4 2
#endif
// (2 nodes per lane, American P representation: at 4 waves per SIMD -- 128 VGPRs -- the kernel spills two registers, and a
// scratch reload inside the row loop drains the DMA prefetch: 3 there)
__global__ void __launch_bounds__(64 * HADI_STRIP_WAVES(B), (B >= 8 ? 2 : B == 4 ? HADI_STRIP_OCC_B4 : AMER == 2 ? 3 : 4)) hadi_pass_a_strip(HadiSweepArgs a, int n) {
    static_assert(sizeof(T) == 8 || AMER == 0, "the fp32-state sweep is European only");
    static_assert(G == 1 || (G == 2 && B == 8), "paired strips: 8 nodes per lane");
    HADI_DYN_SMEM(double, smem);
    constexpr int NS = HADI_STRIP_NS(B, G, (int)sizeof(T)), NWV = HADI_STRIP_WAVES(B), NPAIR = NWV / G, c0slot = 64 * B * G;
    // American P representation at 8 nodes per lane: one slot stays BEHIND the prefetch -- row j itself, whose raw P the step
    // reads again for lambda_bar -- so the row D = NS - 1 ahead is fetched, not the row NS ahead
    constexpr int KEEP = (AMER == 2 && B >= HADI_AMP_KEEP_MIN_B && G == 1) ? 1 : 0, D = NS - KEEP;
    constexpr int NA = D - 2;  // DMA batches in flight behind the one that is waited for
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int pair = wave / G, half = wave - pair * G;  // (G = 1: pair = wave, half = 0)
    const int total = a.n_inst * a.sblocks;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= total) return;
    const int inst = logical / a.sblocks, sb = logical - inst * a.sblocks;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int nrows = a.L.nrows, npad = a.L.nrows_pad, rowp = a.L.rowp;
    double *coef = reinterpret_cast<double *>(reinterpret_cast<T *>(smem) + (size_t)NPAIR * NS * rowp);
    // P representation: the payoff row (it depends on s only: v-row 0 of the packed payoff) behind the coefficient arrays;
    // re-read from LDS every row rather than held in 2 B registers per lane (that version spilled)
    const double *payl = coef + 4 * 64 * B * G;
    const int j0 = (sb * NPAIR + pair) * a.RS;
    const bool has_strip = j0 < nrows;  // (wave-uniform; a wavefront without a strip only helps with the shared copies below)
    const int j1 = (j0 + a.RS < nrows) ? j0 + a.RS : nrows;

    HadiStripCtxT<T> c;
    c.lane = lane;
    c.rowp = rowp;
    c.coef = coef;
    c.half = half;
    double *const xch0 = coef + 4 * 64 * B * G + (AMER == 2 ? rowp : 0);  // the pairs' exchange buffers (behind the payoff row)
    c.xch = xch0 + pair * 16;
    c.err = a.err; c.debug = a.debug;
    c.dt = hadi_uniform_d(ip.dt); c.thdt = hadi_uniform_d(ip.thdt);
    c.c1 = hadi_uniform_d(1.0 + ip.thdt * ip.half_rd);
    c.kap = hadi_uniform_d((ip.dt - ip.thdt) / ip.thdt);  // the host keeps theta = 0 off this kernel
    c.hr0 = hadi_uniform_d(ip.hr0); c.inv0 = hadi_uniform_d(1.0 / (1.0 + ip.thdt * ip.hr0));
    c.e_nm1 = hadi_uniform_d(exp(ip.bc_rate * ip.dt * (n - 1)));  // device_solver.hpp:238
    c.e_n = hadi_uniform_d(exp(ip.bc_rate * ip.dt * n));          // device_solver.hpp:246
    const T *__restrict__ Ub = reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride;
    c.Yi = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride;
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.b2r = a.b2row + (size_t)inst * rowp;
    // P representation: 1/dt, and which node is s_max (lambda_bar stays 0 there, as in hadi_row_step)
    c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
    if constexpr (AMER == 2) {
        c.inv_dt = hadi_uniform_d(1.0 / ip.dt);
        const int e1 = a.L.m1 - 1;  // node i = m1 is element m1 - 1 of the row's 64 B G interior nodes
        if (e1 / (64 * B) == half) {
            c.m1_lane = (e1 - half * 64 * B) / B;
            c.m1_r = (e1 - half * 64 * B) % B;
        }
    }

    T *ring = reinterpret_cast<T *>(smem) + (size_t)pair * NS * rowp;
    auto slot = [&](int jj) { return ring + (size_t)((NS & (NS - 1)) == 0 ? (jj & (NS - 1)) : (jj + 12) % NS) * rowp; };  // (jj >= -4)
    // returns the number of vector-memory instructions issued (rows outside the allocation are zero-filled)
    auto fetch = [&](int jj) -> int {
        const bool exists = jj >= 0 && jj < npad;
        if constexpr (G > 1) {
            return hadi_half_row_to_lds<B, T>(Ub + (ptrdiff_t)jj * rowp, slot(jj), half, lane, exists);
        } else {
            hadi_row_to_lds_fixed<B, T>(Ub + (ptrdiff_t)jj * rowp, slot(jj), lane, exists);
            return exists ? hadi_row_dma_count<T>(rowp) : 0;
        }
    };
    // Direction of the walk: even strips go up (j0 -> j1-1), odd strips come down (j1-1 -> j0).  Neighbouring strips
    // then touch their shared halo rows at the same time -- both start there or both end there -- so the second reader
    // finds them in L2 instead of fetching them again ~100 us later (HBM reads of this pass 9.8 -> ~9 B per node).
    // Below, "behind" = rows already passed (registers), "ahead" = rows still to come (LDS ring / in flight); for a
    // descending strip the row-table scalars of the +1/+2 and -1/-2 neighbours simply swap roles.
    const int dir = (((sb * NPAIR + pair) & 1) == 0) ? 1 : -1;
    const int cnt = j1 - j0;
    const int js = dir > 0 ? j0 : j1 - 1;
    auto row_ok = [&](int jj) { return jj >= 0 && jj < npad; };
    // ---- prologue: the next rows ahead to the ring, the two rows behind and the first row to registers ----
    // aft[k] = vector-memory instructions issued after the DMA of the row 2 + k ahead: aft[0] belongs to the row that is
    // waited for next, the row NS - 1 ahead is the youngest DMA (nothing behind it yet)
    int aft[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) aft[k] = 0;
    // Order of the prologue: this wavefront's row fetches (LDS-DMA) and register loads are ISSUED first, then the block
    // copies the shared s-coefficient arrays (global -> LDS) and meets at the only block-wide barrier -- the two memory
    // round trips overlap instead of following each other (a launch of short strips is mostly prologue: 64 instances of
    // 512x256, 9-row strips: 0.0380 -> see DESIGN.md section 5).
    if (has_strip) {
        if constexpr (KEEP) fetch(js);  // (the first row too: the step reads its raw P from the ring)
        fetch(js + dir);
        fetch(js + 2 * dir);
#pragma unroll
        for (int q = 3; q < D; q++) {
            const int zq = fetch(js + q * dir);
#pragma unroll
            for (int k = 0; k < NA; k++)
                if (k + 2 < q) aft[k] += zq;
        }
    }
    // rows behind by 2, behind by 1 (carried in the state's own type: with an fp32 state they are exact floats and cost
    // half the registers), current row (double: used throughout the step)
    T um2[B], um1[B];
    double u0[B];
    // The i = 0 column of the five stencil rows is wave-uniform: ONE register pair carries it, spread over the lanes
    // (lane k = row j - 2 + k in walking order), read with v_readlane where needed and shifted by a DPP move per step.
    // G = 2: the low half owns the i = 0 column; `evec` carries, the same way, the partner's node next to this half (the
    // high half's first node for the low half and vice versa) on the rows behind / at / ahead of j (lanes 1, 2, 3).
    double c0vec, evec = 0.0;
    int epos = 0;
    if constexpr (G > 1) {
        const int inode = (half == 0) ? 64 * B + 1 : 64 * B;
        epos = (sizeof(T) == 4) ? hadi_pos_f32(B, G, inode) : hadi_pos(B, G, inode);
    }
    double t2[B], t1[B];
#pragma unroll
    for (int r = 0; r < B; r++) t2[r] = t1[r] = u0[r] = 0.0;
    c0vec = 0.0;
    if (has_strip) {
        if (row_ok(js - 2 * dir)) hadi_get_block<B, G, T>(Ub + (ptrdiff_t)(js - 2 * dir) * rowp, half, lane, t2);
        if (row_ok(js - dir)) hadi_get_block<B, G, T>(Ub + (ptrdiff_t)(js - dir) * rowp, half, lane, t1);
        hadi_get_block<B, G, T>(Ub + (size_t)js * rowp, half, lane, u0);
        const int rr = js + (lane - 2) * dir;
        c0vec = (half == 0 && lane < 4 && row_ok(rr)) ? (double)Ub[(ptrdiff_t)rr * rowp + c0slot] : 0.0;
        if constexpr (G > 1) evec = (lane < 4 && row_ok(rr)) ? (double)Ub[(ptrdiff_t)rr * rowp + epos] : 0.0;
    }
    {   // s-coefficient arrays to LDS; the two beta arrays scaled by -theta dt (r_d - r_f) on the way (hadi_strip_step)
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B * G;
        const double mq = -(ip.thdt * ip.q);
        for (int e = threadIdx.x; e < 4 * 64 * B * G; e += 64 * NWV) coef[e] = (e < 2 * 64 * B * G) ? mq * sc[e] : sc[e];
    }
    if constexpr (AMER == 2) {
        const double *__restrict__ pg = a.U0 + (size_t)inst * a.L.inst_stride;
        double *pw = coef + 4 * 64 * B * G;
        for (int e = threadIdx.x; e < rowp; e += 64 * NWV) pw[e] = pg[e];
    }
    if constexpr (G > 1) {  // the pairs' exchange buffers (values + rendezvous tokens, all zero: no row has token 0)
        if (threadIdx.x < NPAIR * 16) xch0[threadIdx.x] = 0.0;
    }
    __syncthreads();  // the only block-wide barrier: the coefficient arrays are shared
    if (!has_strip) return;
    // 8 nodes per lane, European fp64 (the headline kernel): two of the four arrays fit the registers left over (224 -> 250
    // VGPRs, no spill): 8 of the 16 coefficient reads per row step less on the LDS pipe, +0.7 % on 512x256 x256 (three
    // interleaved runs of each build on one box, gpurun_out/r03aa); 3: only the last array (no gain measured)
#ifndef HADI_STRIP_CREG8
#define HADI_STRIP_CREG8 2
#endif
    constexpr int CREG = (B <= HADI_STRIP_CREG_MAX_B && G == 1) ? 1 : (B == 8 && G == 1 && AMER == 0 && sizeof(T) == 8) ? HADI_STRIP_CREG8 : 0;
    double cf[4 * B];
    if constexpr (CREG) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double t[B];
            hadi_get_block<B, 1>(coef + q * 64 * B, 0, lane, t);
#pragma unroll
            for (int r = 0; r < B; r++) cf[q * B + r] = t[r];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4 * B; e++) cf[e] = 0.0;
    }
    if constexpr (AMER == 2) {  // U = max(P, U_0) on the rows behind (the current row keeps its raw P for lambda_bar)
        double pay[B];
        hadi_get_block<B, G>(payl, half, lane, pay);
#pragma unroll
        for (int r = 0; r < B; r++) {
            t2[r] = fmax(t2[r], pay[r]);
            t1[r] = fmax(t1[r], pay[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
        um2[r] = (T)t2[r];
        um1[r] = (T)t1[r];
    }
#if !defined(HADI_EMU)
    // Consume the prologue's register loads HERE: otherwise hipcc parks their s_waitcnt vmcnt(0) at the loop header,
    // where it would retire the DMA prefetch and the result stores in every iteration.
#pragma unroll
    for (int r = 0; r < B; r++) asm volatile("" : "+v"(um2[r]), "+v"(um1[r]), "+v"(u0[r]));  // (T and double operands)
    asm volatile("" : "+v"(c0vec));
    if constexpr (G > 1) asm volatile("" : "+v"(evec));
#endif

#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    HADI_STAMP_DECL(c.stamp_acc_)
    for (int t = 0; t < cnt; t++) {
        const int j = js + dir * t;
        HADI_STAMPC(30);  // carry + loop
        HadiSRow srow;
        hadi_sload_issue(a.rowc + ((size_t)inst * nrows + j) * HADI_RC + HADI_SRC0, srow);  // flies during the DMA wait
        hadi_wave_rendezvous();
        // the row D ahead goes to the slot of row j (of row j - 1 when one slot is kept behind): that row is in registers,
        // and this wavefront's last read of the slot (in the previous step) has been retired there.  Issued BEFORE the
        // wait below, so that the prefetch does not queue behind it.
        int z = 0;
        if (t + D <= cnt + 1) z = fetch(j + D * dir);
        hadi_wait_vmcnt(aft[0] + z);  // the row two ahead has landed (the row one ahead landed a step earlier)
        HADI_STAMPC(24);  // wait for the DMA
#pragma unroll
        for (int k = 0; k + 1 < NA; k++) aft[k] = aft[k + 1] + z;
        aft[NA - 1] = 0;
        hadi_wave_rendezvous();
        double up1[B], up2[B];
        hadi_get_block<B, G, T>(slot(j + dir), half, lane, up1);
        hadi_get_block<B, G, T>(slot(j + 2 * dir), half, lane, up2);
        if (half == 0) {  // (wave-uniform; always true for G = 1)
            const double c0new = (double)slot(j + 2 * dir)[c0slot];  // (every lane reads the same word)
            c0vec = (lane == 4) ? c0new : c0vec;
        }
        double rt[HADI_RCL];
        hadi_sload_wait(srow, rt);  // one lgkmcnt(0) for the table entry and the LDS reads above
        if (dir < 0) {  // descending: "behind" rows are j+1, j+2 -- swap the neighbour weights instead of the arrays
            double w;
            w = rt[RC_WMS - HADI_SRC0]; rt[RC_WMS - HADI_SRC0] = rt[RC_WPS - HADI_SRC0]; rt[RC_WPS - HADI_SRC0] = w;
            w = rt[RC_L2 - HADI_SRC0]; rt[RC_L2 - HADI_SRC0] = rt[RC_U2 - HADI_SRC0]; rt[RC_U2 - HADI_SRC0] = w;
            w = rt[RC_L1 - HADI_SRC0]; rt[RC_L1 - HADI_SRC0] = rt[RC_U1 - HADI_SRC0]; rt[RC_U1 - HADI_SRC0] = w;
        }
        HADI_STAMPC(25);  // LDS reads + table entry + DMA issue
        double praw[B], lamc0 = 0.0;
        const double c0m2 = hadi_read_lane(c0vec, 0), c0m1 = hadi_read_lane(c0vec, 1), c00 = hadi_read_lane(c0vec, 2);
        const double c0p1 = hadi_read_lane(c0vec, 3), c0p2 = hadi_read_lane(c0vec, 4);
        double e0m2 = c0m2, e0m1 = c0m1, e00 = c00, e0p1 = c0p1, e0p2 = c0p2;  // (the carried i = 0 values stay raw)
#pragma unroll
        for (int r = 0; r < B; r++) praw[r] = 0.0;
        if constexpr (AMER == 2) {
            double pay[B];
            hadi_get_block<B, G>(payl, half, lane, pay);
            const double pay_c0 = payl[c0slot];
#pragma unroll
            for (int r = 0; r < B; r++) {
                if constexpr (G == 2) {
                    // paired strips: u0 stays the raw P (hadi_strip_step, RAW_U0); the row behind was carried raw as well
                    um1[r] = (T)fmax((double)um1[r], pay[r]);
                } else {
                    if constexpr (!KEEP) praw[r] = u0[r];  // the raw P of row j: lambda_bar comes from it inside the step
                    u0[r] = fmax(u0[r], pay[r]);
                }
                up1[r] = fmax(up1[r], pay[r]);
                up2[r] = fmax(up2[r], pay[r]);
            }
            lamc0 = fmax(0.0, (pay_c0 - c00) * c.inv_dt);
            e0m2 = fmax(c0m2, pay_c0); e0m1 = fmax(c0m1, pay_c0); e00 = fmax(c00, pay_c0);
            e0p1 = fmax(c0p1, pay_c0); e0p2 = fmax(c0p2, pay_c0);
        }
        double dm2[B], dm1[B];
#pragma unroll
        for (int r = 0; r < B; r++) {
            dm2[r] = (double)um2[r];
            dm1[r] = (double)um1[r];
        }
        double un[B];
        double xb_ = 0.0, x0_ = 0.0, xa_ = 0.0;  // the partner's boundary node on the rows behind / at / ahead (G = 2)
        if constexpr (G > 1) {
            xb_ = hadi_read_lane(evec, 1); x0_ = hadi_read_lane(evec, 2); xa_ = hadi_read_lane(evec, 3);
            if constexpr (AMER == 2) {  // (the carried values stay raw P: U = max(P, U_0) on the partner's node too)
                const double pay_e = payl[epos];
                xb_ = fmax(xb_, pay_e); x0_ = fmax(x0_, pay_e); xa_ = fmax(xa_, pay_e);
            }
        }
        if (j == nrows - 1) hadi_strip_step<B, AMER, true, T, G, CREG>(c, j, rt, dm2, dm1, u0, up1, up2, e0m2, e0m1, e00, e0p1, e0p2, praw, lamc0, slot(j + dir), un, xb_, x0_, xa_, slot(j), payl, cf);
        else hadi_strip_step<B, AMER, false, T, G, CREG>(c, j, rt, dm2, dm1, u0, up1, up2, e0m2, e0m1, e00, e0p1, e0p2, praw, lamc0, slot(j + dir), un, xb_, x0_, xa_, slot(j), payl, cf);
#pragma unroll
        for (int k = 0; k < NA; k++) aft[k] += hadi_put_block_stores<B, T>();  // the row's vector stores (the i = 0 store is not counted: lower bound)
        double enew = 0.0;
        if constexpr (G > 1) {
            // The partner's boundary node of the row TWO ahead, from the partner's half of the ring slot.  Safe here and only
            // here: the partner retired its DMA of that row before it published this step's token (which the exchange inside
            // the step has just seen), and it refills that slot two steps on -- after the next exchange, which needs this
            // wavefront's next token.
            enew = (double)slot(j + 2 * dir)[epos];  // (every lane reads the same word)
        }
#pragma unroll
        for (int r = 0; r < B; r++) {
            um2[r] = um1[r];
            um1[r] = (T)u0[r];
            u0[r] = un[r];
        }
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the re-read is retired before the next step reuses that slot
#endif
        c0vec = hadi_lane_next(c0vec);  // lane k takes lane k + 1: one row on
        if constexpr (G > 1) {
            evec = hadi_lane_next(evec);
            evec = (lane == 3) ? enew : evec;
        }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev_) :: "memory");  // the step stamped itself
#endif
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 4 && lane == 0)
        for (int k = 24; k < 31; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// ------------------------------------------------------------------------------------------------
// Pair strips (128 < m1 <= 256): TWO strips per wavefront, 32 lanes x 8 nodes each.
// At 4 nodes per lane the strip kernel spends as many instructions on a 256-node row as the 8-node kernel on a 512-node one
// in everything that is per LANE rather than per node -- the six levels of cyclic reduction, the row scalars, the ring
// bookkeeping: 380 VALU + 133 SALU + 70 LDS instructions per row and wavefront (PMC, profiles/r02_c3_pmc_summary.json)
// against 513 + ~150 + ~90 for twice the nodes, and the SIMDs are issue-bound (two wavefronts, 108 % of the issue cycles
// active).  Here a wavefront runs the 8-node arithmetic on TWO independent strips at once: lanes 0..31 walk strip A, lanes
// 32..63 strip B of the same instance, lane h of a half owning the nodes 8h+1 .. 8h+8 of its strip's current row.  Same
// storage layout as the 4-node kernels (row of 272 doubles: the column pass does not change), so a lane's nodes are two
// 32-byte chunks of a row: {0,1,4,5} and {2,3,6,7}.
//   cross-lane traffic   the distance-1 exchanges are the same wave shifts; what crosses the half boundary (lane 31 <-> 32)
//                        is multiplied by a zero coefficient on either side (node 256 or a pad has no upper neighbour: Bp =
//                        Dp = 0; the first node's coupling to i = 0 is moved to the right-hand side), so no fix-up is needed;
//                        the cyclic reduction has five levels, its permutes stay inside the half
//   row scalars          differ between the halves: both halves' table entries come through the scalar cache as before (two
//                        sets of SGPRs, issued at the loop top) and are moved to per-lane registers under the halves' exec
//                        masks -- 40 moves per step; vector loads of the entries (tried first) kept 48 more registers live
//                        across the step and the kernel spilled
//   ring                 a slot holds the two rows of a step interleaved in 512-byte pieces [piece][half] (one LDS-DMA
//                        instruction moves a piece of BOTH rows: lanes 0..31 from row A, 32..63 from row B), the two 128-byte
//                        tails adjacent: 4352 B per slot, 6 DMA instructions per step
//   out-of-range rows    the strips of a wavefront are equally long or the second is shorter / empty; a finished or empty
//                        half keeps computing on clamped rows and stores nothing.  Rows j-2 .. j+2 outside the grid are
//                        clamped too: they only ever meet zero weights (as in hadi_small_seq_kernel)
// Counted waits as in hadi_pass_a_strip: the row two ahead has landed, younger DMA batches and the result stores stay in flight.
#define HADI_PAIR_SLOT 544   // doubles per ring slot
#define HADI_PAIR_WAVES 4    // wavefronts (= 8 strips) per block

HADI_DEV HADI_FORCEINLINE void hadi_pair_get(const double *p, int ch1, double (&u)[8]) {
    const double2 a = *reinterpret_cast<const double2 *>(p), b = *reinterpret_cast<const double2 *>(p + 2);
    const double2 c = *reinterpret_cast<const double2 *>(p + ch1), d = *reinterpret_cast<const double2 *>(p + ch1 + 2);
    u[0] = a.x; u[1] = a.y; u[4] = b.x; u[5] = b.y; u[2] = c.x; u[3] = c.y; u[6] = d.x; u[7] = d.y;
}
HADI_DEV HADI_FORCEINLINE void hadi_pair_put(double *p, const double (&u)[8]) {  // global row: the chunks are 128 doubles apart
    double2 a, b, c, d;
    a.x = u[0]; a.y = u[1]; b.x = u[4]; b.y = u[5]; c.x = u[2]; c.y = u[3]; d.x = u[6]; d.y = u[7];
    *reinterpret_cast<double2 *>(p) = a; *reinterpret_cast<double2 *>(p + 2) = b;
    *reinterpret_cast<double2 *>(p + 128) = c; *reinterpret_cast<double2 *>(p + 130) = d;
}
// LDS-DMA of the two rows `grow` (per lane: the row of this lane's half) into ring slot `slot`.  6 vector-memory instructions.
HADI_DEV HADI_FORCEINLINE void hadi_pair_fetch(const double *__restrict__ grow, double *slot, int lane) {
    const int H = lane >> 5, h = lane & 31;
#if defined(HADI_EMU)
    for (int pc = 0; pc < 4; pc++)
        for (int e = 0; e < 2; e++) slot[pc * 128 + H * 64 + 2 * h + e] = grow[64 * pc + 2 * h + e];
    if (h < 8)
        for (int e = 0; e < 2; e++) slot[512 + 16 * H + 2 * h + e] = grow[256 + 2 * h + e];
#else
    const double *gsrc = grow + 2 * h;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)slot);
#pragma unroll
    for (int pc = 0; pc < 4; pc++) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc + 64 * pc), "s"(lds0 + 1024u * pc) : "memory");
    }
    // the 128-byte tails (slot 256 = i = 0 and the pads): half A to bytes 4096.., half B right behind it (the hardware adds
    // 16 x lane to M0: 512 for lane 32, hence the base 4096 + 128 - 512)
    if (lane < 8) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc + 256), "s"(lds0 + 4096u) : "memory");
    }
    if (lane >= 32 && lane < 40) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc + 256), "s"(lds0 + 3712u) : "memory");
    }
#endif
}
#define HADI_PAIR_DMA 6

// One step of both strips of the wavefront.  rv: this lane's row scalars (entries RC_L2 .. RC_WPS of its half's row); the
// stencil rows as in hadi_strip_step (B = 8); c0*: the i = 0 column of the five rows of this lane's half; yrow: this half's
// output row (global); act: this half still has rows (stores are skipped otherwise).
template <int AMER, bool LAST>
HADI_DEV HADI_FORCEINLINE void hadi_pair_step(const HadiStripCtxT<double> &c, int h, bool act, bool is_last, const double (&rv)[HADI_RCL],
                                              const double (&um2)[8], const double (&um1)[8], const double (&u0)[8],
                                              const double (&up1)[8], const double (&up2)[8], double c0m2, double c0m1, double c00,
                                              double c0p1, double c0p2, double lamc0_in, const double *raw_chunk,
                                              const double *next_chunk, double (&u_next)[8], double *yrow, const double *lrow) {
    constexpr int B = 8, NB = 7, c0slot = 256;
    const int lane = c.lane;
    const bool first = (h == 0);
    const double dt = c.dt, thdt = c.thdt, c1 = c.c1, kap = c.kap, e_nm1 = c.e_nm1, e_n = c.e_n;
    const double vth = rv[RC_VTH - HADI_SRC0];
    const double wm = rv[RC_WMS - HADI_SRC0], wz = rv[RC_WZS - HADI_SRC0], wp = rv[RC_WPS - HADI_SRC0];
    const double a2l2 = rv[RC_L2 - HADI_SRC0], a2l1 = rv[RC_L1 - HADI_SRC0], a2m = rv[RC_M - HADI_SRC0], a2u1 = rv[RC_U1 - HADI_SRC0],
                 a2u2 = rv[RC_U2 - HADI_SRC0];
    const double b1val = rv[RC_B1VAL - HADI_SRC0];
    const int b1raw = (int)rv[RC_B1COL - HADI_SRC0];
    const bool b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;
    const int b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    const int b1e = b1col - 1;
    const int b1k = (b1col >= 1 && (b1e >> 3) == h) ? (b1e & 7) : -1;  // the node of this lane that carries the row's b1 entry

    // ---- column i = 0 ----------------------------------------------------------------------------------
    const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
    const double b1c0 = b1_at0 ? b1val : 0.0;
    const double b2c0 = (LAST && is_last) ? c.b2r[c0slot] : 0.0;
    const double lamc0 = (AMER == 1) ? lrow[c0slot] : (AMER == 2) ? lamc0_in : 0.0;
    const double a1c0 = -c.hr0 * c00;
    double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
    y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
    const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
    const double x0 = y0c0 * c.inv0;
    const double yout_c0 = x0 + c2c0;

    // ---- explicit operators ---------------------------------------------------------------------------------
    double tt[B], A2U[B];
#pragma unroll
    for (int r = 0; r < B; r++) {
        tt[r] = wm * um1[r] + wz * u0[r] + wp * up1[r];
        A2U[r] = a2l1 * um1[r] + a2m * u0[r] + a2u1 * up1[r];
    }
#pragma unroll
    for (int r = 0; r < B; r++) A2U[r] = fma(a2l2, um2[r], A2U[r]);
#pragma unroll
    for (int r = 0; r < B; r++) A2U[r] = fma(a2u2, up2[r], A2U[r]);
    double u0L = hadi_lane_prev(u0[B - 1]), tL = hadi_lane_prev(tt[B - 1]);
    const double u0R = hadi_lane_next(u0[0]), tR = hadi_lane_next(tt[0]);  // (lane 31: lane 32's values, times Bp = Dp = 0)
    if (first) {
        u0L = c00;
        tL = wm * c0m1 + wz * c00 + wp * c0p1;
    }
    const double b1add = b1val * (dt * e_nm1 + thdt * (e_n - e_nm1));

    double lam[B], b2v[B];
    if constexpr (AMER == 1) hadi_pair_get(lrow + 4 * h, 128, lam);
    if constexpr (AMER == 2) {
        double praw[B];
        hadi_pair_get(raw_chunk, 256, praw);  // the raw P of row j, still intact in its ring slot
#pragma unroll
        for (int r = 0; r < B; r++) {
            lam[r] = (u0[r] - praw[r]) * c.inv_dt;  // exactly max(0, (U_0 - P) / dt): U = max(P, U_0)
            if (h == c.m1_lane && r == c.m1_r) lam[r] = 0.0;
        }
    }
    if constexpr (LAST) {
#pragma unroll
        for (int r = 0; r < B; r++) b2v[r] = 0.0;
        if (is_last) hadi_pair_get(c.b2r + 4 * h, 128, b2v);
    }

    hadi_set_prio(1);
    double Bm[B], Bp[B], Dm[B], Dp[B];
    double ys[B], ps[B], gs[B], iu[B], cp[B];
    double il_last = 0.0, im_last = 1.0, d_last = 0.0;
#pragma unroll
    for (int r = 0; r < B; r++) {
        if ((r & 3) == 0 || (r & 3) == 2) {
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
            // nodes {0,1,4,5} sit in the lane's first chunk, {2,3,6,7} in the second (128 doubles on)
            const int co = 4 * h + ((r & 2) ? 128 : 0) + ((r & 4) ? 2 : 0);
            const double2 t0 = *reinterpret_cast<const double2 *>(c.coef + 0 * 256 + co);
            const double2 t1 = *reinterpret_cast<const double2 *>(c.coef + 1 * 256 + co);
            const double2 t2 = *reinterpret_cast<const double2 *>(c.coef + 2 * 256 + co);
            const double2 t3 = *reinterpret_cast<const double2 *>(c.coef + 3 * 256 + co);
            Bm[r] = t0.x; Bm[r + 1] = t0.y;
            Bp[r] = t1.x; Bp[r + 1] = t1.y;
            Dm[r] = t2.x; Dm[r + 1] = t2.y;
            Dp[r] = t3.x; Dp[r + 1] = t3.y;
        }
        const double uL = (r == 0) ? u0L : u0[r == 0 ? 0 : r - 1];
        const double uR = (r == B - 1) ? u0R : u0[r == B - 1 ? r : r + 1];
        const double tl = (r == 0) ? tL : tt[r == 0 ? 0 : r - 1];
        const double tr = (r == B - 1) ? tR : tt[r == B - 1 ? r : r + 1];
        double il = fma(-vth, Dm[r], Bm[r]);  // (Bm, Bp hold -theta dt (r_d - r_f) s beta_s: hadi_strip_step)
        iu[r] = fma(-vth, Dp[r], Bp[r]);
        const double sm = il + iu[r];
        const double im = c1 - sm;
        const double T1 = fma(-iu[r], uR, fma(-il, uL, fma(-im, u0[r], u0[r])));
        const double A0U = Bm[r] * tl - (Bm[r] + Bp[r]) * tt[r] + Bp[r] * tr;
        double S = A0U + A2U[r];
        if constexpr (LAST) S += b2v[r] * e_nm1;
        if constexpr (AMER) S += lam[r];
        double y = fma(dt, S, u0[r]);
        y = fma(kap, T1, y);
        y = fma(b1add, (b1k == r) ? 1.0 : 0.0, y);
        if (r == 0 && first) {  // x_0 is known: move it to the right-hand side
            y -= il * x0;
            il = 0.0;
        }
        if (r < NB) {
            if (r == 0) {
                const double inv = hadi_rcp(im);
                cp[0] = iu[0] * inv;
                ys[0] = y * inv;
                ps[0] = il * inv;
            } else {
                const double inv = hadi_rcp(fma(-il, cp[r - 1], im));
                cp[r] = iu[r] * inv;
                ys[r] = fma(-il, ys[r - 1], y) * inv;
                ps[r] = -(il * ps[r - 1]) * inv;
            }
        } else {
            il_last = il;
            im_last = im;
            d_last = y;
        }
    }
    double ra, rb, rcc, rf;
    {
        gs[NB - 1] = cp[NB - 1];
#pragma unroll
        for (int r = NB - 2; r >= 0; r--) {
            ys[r] = fma(-cp[r], ys[r + 1], ys[r]);
            ps[r] = fma(-cp[r], ps[r + 1], ps[r]);
            gs[r] = -cp[r] * gs[r + 1];
        }
        const double p0n = hadi_lane_next(ps[0]), g0n = hadi_lane_next(gs[0]), y0n = hadi_lane_next(ys[0]);
        ra = -il_last * ps[NB - 1];
        rb = im_last - il_last * gs[NB - 1] - iu[B - 1] * p0n;
        rcc = -iu[B - 1] * g0n;
        rf = d_last - il_last * ys[NB - 1] - iu[B - 1] * y0n;
    }
    {   // parallel cyclic reduction over the 32 interface unknowns of each half (normalised rows, see hadi_row_step)
        hadi_set_prio(3);
        const double rinv0 = hadi_rcp(rb);
        ra *= rinv0;
        rcc *= rinv0;
        rf *= rinv0;
#pragma unroll
        for (int s = 1; s < 32; s <<= 1) {
            const int up_lane = (lane & 32) | ((lane - s) & 31), dn_lane = (lane & 32) | ((lane + s) & 31);
            double aL, cL, fL, aR, cR, fR;
            if (s == 1) {
                aL = hadi_lane_prev(ra); cL = hadi_lane_prev(rcc); fL = hadi_lane_prev(rf);
                aR = hadi_lane_next(ra); cR = hadi_lane_next(rcc); fR = hadi_lane_next(rf);
            } else if (s == 16) {  // h - 16 and h + 16 are the same lane (mod 32)
                aL = aR = hadi_lane_get(ra, up_lane); cL = cR = hadi_lane_get(rcc, up_lane); fL = fR = hadi_lane_get(rf, up_lane);
            } else {
                aL = hadi_lane_get(ra, up_lane); cL = hadi_lane_get(rcc, up_lane); fL = hadi_lane_get(rf, up_lane);
                aR = hadi_lane_get(ra, dn_lane); cR = hadi_lane_get(rcc, dn_lane); fR = hadi_lane_get(rf, dn_lane);
            }
            const double bn = fma(-rcc, aR, fma(-ra, cL, 1.0));
            const double rn = hadi_rcp(bn);
            rf = fma(-rcc, fR, fma(-ra, fL, rf)) * rn;
            if (s < 16) {
                const double an = -(ra * aL) * rn;
                const double cn = -(rcc * cR) * rn;
                ra = an;
                rcc = cn;
            }
        }
    }
    hadi_set_prio(0);
    hadi_pair_get(next_chunk, 256, u_next);  // the row ahead again from its ring slot (flies during the stores)
    const double X = rf, XL = hadi_lane_prev(X);  // (lane 32: lane 31's X, times ps = 0)
    double yo[B];
#pragma unroll
    for (int r = 0; r < B; r++) {
        double x;
        if (r < NB) x = ys[r] - XL * ps[r] - X * gs[r];
        else x = X;
        double corr;
        if constexpr (LAST) corr = thdt * (b2v[r] * e_n - (A2U[r] + b2v[r] * e_nm1));
        else corr = -thdt * A2U[r];
        yo[r] = x + corr;
    }
    // (plain global stores: the counted waits need them in the same in-order queue as the LDS-DMA loads, hadi_strip_step)
    if (act) {
        hadi_pair_put(yrow + 4 * h, yo);
        if (first) yrow[c0slot] = yout_c0;
    }
}
#define HADI_PAIR_STORES 4  // vector stores per step counted by the waits (the i = 0 store is not: lower bound)

// LDS: [4 wavefronts][NS slots][544] | 4 coefficient arrays of 256 | payoff row of 272 (AMER == 2) | i = 0 history [4][2][4].
template <int AMER>
__global__ void __launch_bounds__(64 * HADI_PAIR_WAVES, 2) hadi_pass_a_pairs(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    // 4-slot ring.  European / explicit pair: the rows 1 .. 4 ahead in the ring (1, 2 landed, 3, 4 in flight); P representation:
    // the slot of row j itself is kept for the step's re-read of the raw P, so the rows 1 .. 3 ahead.
    constexpr int NS = 4, D = (AMER == 2) ? 3 : 4, NWV = HADI_PAIR_WAVES, c0slot = 256, ROWP = 272;
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int H = lane >> 5, h = lane & 31;
    const int total = a.n_inst * a.sblocks;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= total) return;
    const int inst = logical / a.sblocks, sb = logical - inst * a.sblocks;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int nrows = a.L.nrows;
    double *ring = smem + (size_t)wave * NS * HADI_PAIR_SLOT;
    double *coef = smem + (size_t)NWV * NS * HADI_PAIR_SLOT;
    double *payl = coef + 4 * 256;
    double *hist = payl + (AMER == 2 ? ROWP : 0) + (size_t)(wave * 2 + H) * 4;  // this half's last four i = 0 values
    // the two strips of this wavefront: 2 (sb NWV + wave) and the next one; the second may be shorter or empty
    const int sA = 2 * (sb * NWV + wave);
    const int j0A = sA * a.RS, j0B = j0A + a.RS;
    const int cntA = HADI_UNIFORM(j0A < nrows ? ((j0A + a.RS < nrows ? j0A + a.RS : nrows) - j0A) : 0);
    const int cntB = HADI_UNIFORM(j0B < nrows ? ((j0B + a.RS < nrows ? j0B + a.RS : nrows) - j0B) : 0);
    const int j0 = H ? j0B : j0A, cnt = H ? cntB : cntA;
    const int dir = ((sb * NWV + wave) & 1) ? -1 : 1;  // (both strips of a wavefront walk the same way)
    const int js = dir > 0 ? j0 : j0 + cnt - 1;

    HadiStripCtxT<double> c;
    c.lane = lane; c.rowp = ROWP; c.coef = coef; c.half = 0; c.xch = nullptr; c.err = a.err; c.debug = a.debug;
    c.dt = hadi_uniform_d(ip.dt); c.thdt = hadi_uniform_d(ip.thdt);
    c.c1 = hadi_uniform_d(1.0 + ip.thdt * ip.half_rd);
    c.kap = hadi_uniform_d((ip.dt - ip.thdt) / ip.thdt);
    c.hr0 = hadi_uniform_d(ip.hr0); c.inv0 = hadi_uniform_d(1.0 / (1.0 + ip.thdt * ip.hr0));
    c.e_nm1 = hadi_uniform_d(exp(ip.bc_rate * ip.dt * (n - 1)));  // device_solver.hpp:238
    c.e_n = hadi_uniform_d(exp(ip.bc_rate * ip.dt * n));          // device_solver.hpp:246
    const double *__restrict__ Ub = a.U + (size_t)inst * a.L.inst_stride;
    double *__restrict__ Yb = a.Y + (size_t)inst * a.L.inst_stride;
    const double *__restrict__ Lb = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.Yi = Yb; c.Li = Lb;
    c.b2r = a.b2row + (size_t)inst * ROWP;
    c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
    if constexpr (AMER == 2) {
        c.inv_dt = hadi_uniform_d(1.0 / ip.dt);
        c.m1_lane = (a.L.m1 - 1) >> 3;
        c.m1_r = (a.L.m1 - 1) & 7;
    }
    // rows of this lane's half, clamped to the grid (out-of-range rows only meet zero weights; a finished half stores nothing)
    auto grow = [&](int jj) { return Ub + (size_t)(jj < 0 ? 0 : (jj >= nrows ? nrows - 1 : jj)) * ROWP; };
    auto slot = [&](int q) { return ring + (size_t)(((q % NS) + NS) % NS) * HADI_PAIR_SLOT; };  // q = step index of the row (any sign)
    const int chunk_off = (h >> 4) * 128 + H * 64 + (h & 15) * 4;  // this lane's first chunk inside a slot (doubles)
    const int c0_off = 512 + 16 * H;
    const double *__restrict__ rtab = a.rowc + (size_t)inst * nrows * HADI_RC + HADI_SRC0;
    auto clampj = [&](int jj) { return jj < 0 ? 0 : (jj >= nrows ? nrows - 1 : jj); };
    const int jsA = dir > 0 ? j0A : j0A + cntA - 1, jsB = dir > 0 ? j0B : j0B + cntB - 1;  // (wave-uniform)

    // ---- prologue (memory round trips first, then the shared copies and the block's only barrier: hadi_pass_a_strip) ----
    // step index t <-> row js + dir t; the ring slot of a row is its step index mod NS
    double um2[8], um1[8], u0[8];
    // aft[k] = vector-memory instructions issued after the DMA of the row 2 + k ahead (hadi_pass_a_strip)
    constexpr int NA = D - 2;
    int aft[NA];
#pragma unroll
    for (int k = 0; k < NA; k++) aft[k] = 0;
    if (cntA > 0) {
        if constexpr (AMER == 2) hadi_pair_fetch(grow(js), slot(0), lane);
        hadi_pair_fetch(grow(js + dir), slot(1), lane);
        hadi_pair_fetch(grow(js + 2 * dir), slot(2), lane);
#pragma unroll
        for (int q = 3; q < D; q++) {
            hadi_pair_fetch(grow(js + q * dir), slot(q), lane);
#pragma unroll
            for (int k = 0; k < NA; k++)
                if (k + 2 < q) aft[k] += HADI_PAIR_DMA;
        }
        hadi_pair_get(grow(js - 2 * dir) + 4 * h, 128, um2);
        hadi_pair_get(grow(js - dir) + 4 * h, 128, um1);
        hadi_pair_get(grow(js) + 4 * h, 128, u0);
        if (h == 0) {  // the i = 0 values of the rows js - 2 .. js + 1 (steps -2 .. 1)
#pragma unroll
            for (int q = -2; q <= 1; q++) hist[(q + 4) & 3] = grow(js + q * dir)[c0slot];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 8; r++) um2[r] = um1[r] = u0[r] = 0.0;
    }
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 256;
        const double mq = -(ip.thdt * ip.q);
        for (int e = threadIdx.x; e < 4 * 256; e += 64 * NWV) coef[e] = (e < 2 * 256) ? mq * sc[e] : sc[e];
    }
    if constexpr (AMER == 2) {
        const double *__restrict__ pg = a.U0 + (size_t)inst * a.L.inst_stride;
        for (int e = threadIdx.x; e < ROWP; e += 64 * NWV) payl[e] = pg[e];
    }
    __syncthreads();
    if (cntA == 0) return;
    if constexpr (AMER == 2) {  // U = max(P, U_0) on the rows behind
        double pay[8];
        hadi_pair_get(payl + 4 * h, 128, pay);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            um2[r] = fmax(um2[r], pay[r]);
            um1[r] = fmax(um1[r], pay[r]);
        }
    }
#if !defined(HADI_EMU)
#pragma unroll
    for (int r = 0; r < 8; r++) asm volatile("" : "+v"(um2[r]), "+v"(um1[r]), "+v"(u0[r]));
#endif
    hadi_wave_rendezvous();

    for (int t = 0; t < cntA; t++) {
        const int j = js + dir * t;      // this half's row (meaningless once t >= cnt: clamped)
        const bool act = t < cnt;
        HadiSRow srA, srB;  // both halves' row-table entries through the scalar cache; they fly during the DMA wait
        hadi_sload_issue(rtab + (size_t)clampj(jsA + dir * t) * HADI_RC, srA);
        hadi_sload_issue(rtab + (size_t)clampj(jsB + dir * t) * HADI_RC, srB);
        hadi_wave_rendezvous();
        int z = 0;
        if (t + D <= cntA + 1) {  // into the slot of the row that has just left the ring
            hadi_pair_fetch(grow(j + D * dir), slot(t + D), lane);
            z = HADI_PAIR_DMA;
        }
        hadi_wait_vmcnt((NA > 0 ? aft[0] : 0) + z);  // the row two ahead has landed
#pragma unroll
        for (int k = 0; k + 1 < NA; k++) aft[k] = aft[k + 1] + z;
        if (NA > 0) aft[NA - 1] = 0;
        hadi_wave_rendezvous();
        double up1[8], up2[8];
        hadi_pair_get(slot(t + 1) + chunk_off, 256, up1);
        hadi_pair_get(slot(t + 2) + chunk_off, 256, up2);
        const double c0p2r = slot(t + 2)[c0_off];
        const double c0m2r = hist[(t + 2) & 3], c0m1r = hist[(t + 3) & 3], c00r = hist[t & 3], c0p1r = hist[(t + 1) & 3];
        double rvs[HADI_RCL];
        {
            double rtA[HADI_RCL], rtB[HADI_RCL];
            hadi_sload_wait(srA, rtA);
            hadi_sload_wait(srB, rtB);
            if (dir < 0) {  // descending: the rows behind are j+1, j+2 -- swap the neighbour weights (scalar registers)
                double w;
                w = rtA[RC_WMS - HADI_SRC0]; rtA[RC_WMS - HADI_SRC0] = rtA[RC_WPS - HADI_SRC0]; rtA[RC_WPS - HADI_SRC0] = w;
                w = rtA[RC_L2 - HADI_SRC0]; rtA[RC_L2 - HADI_SRC0] = rtA[RC_U2 - HADI_SRC0]; rtA[RC_U2 - HADI_SRC0] = w;
                w = rtA[RC_L1 - HADI_SRC0]; rtA[RC_L1 - HADI_SRC0] = rtA[RC_U1 - HADI_SRC0]; rtA[RC_U1 - HADI_SRC0] = w;
                w = rtB[RC_WMS - HADI_SRC0]; rtB[RC_WMS - HADI_SRC0] = rtB[RC_WPS - HADI_SRC0]; rtB[RC_WPS - HADI_SRC0] = w;
                w = rtB[RC_L2 - HADI_SRC0]; rtB[RC_L2 - HADI_SRC0] = rtB[RC_U2 - HADI_SRC0]; rtB[RC_U2 - HADI_SRC0] = w;
                w = rtB[RC_L1 - HADI_SRC0]; rtB[RC_L1 - HADI_SRC0] = rtB[RC_U1 - HADI_SRC0]; rtB[RC_U1 - HADI_SRC0] = w;
            }
            // to per-lane registers under the halves' exec masks (RC_LAST is not needed: is_last below)
#pragma unroll
            for (int k = 0; k < HADI_RCL; k++) rvs[k] = 0.0;
            if (H == 0) {
#pragma unroll
                for (int k = 0; k < HADI_RCL; k++)
                    if (k != RC_LAST - HADI_SRC0) rvs[k] = rtA[k];
            } else {
#pragma unroll
                for (int k = 0; k < HADI_RCL; k++)
                    if (k != RC_LAST - HADI_SRC0) rvs[k] = rtB[k];
            }
        }
        double e0m2 = c0m2r, e0m1 = c0m1r, e00 = c00r, e0p1 = c0p1r, e0p2 = c0p2r, lamc0 = 0.0;
        if constexpr (AMER == 2) {
            double pay[8];
            hadi_pair_get(payl + 4 * h, 128, pay);
            const double pay_c0 = payl[c0slot];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                u0[r] = fmax(u0[r], pay[r]);
                up1[r] = fmax(up1[r], pay[r]);
                up2[r] = fmax(up2[r], pay[r]);
            }
            lamc0 = fmax(0.0, (pay_c0 - c00r) * c.inv_dt);
            e0m2 = fmax(c0m2r, pay_c0); e0m1 = fmax(c0m1r, pay_c0); e00 = fmax(c00r, pay_c0);
            e0p1 = fmax(c0p1r, pay_c0); e0p2 = fmax(c0p2r, pay_c0);
        }
        const bool is_last = act && (j == nrows - 1);
        double un[8];
        double *yrow = Yb + (size_t)(act ? j : 0) * ROWP;
        const double *lrow = (AMER == 1) ? Lb + (size_t)(j < 0 ? 0 : (j >= nrows ? nrows - 1 : j)) * ROWP : nullptr;
        // ONE copy of the step, the b2 terms under the per-half predicate `is_last` (+16 registers, ~24 instructions per row).
        // Two copies selected by "does any half sit on the last row" -- the first version -- were laid out by hipcc as "if (x) A;
        // if (!x) B" with everything B's explicit stage reads (the five max'ed rows, both halves' row scalars) kept alive THROUGH
        // A: 244 live registers in A against 146 in B, 44 - 60 of them spilled into the row loop, 2.4x slower than the kernel
        // it was to replace (tools/experiments/README.md).
        hadi_pair_step<AMER, true>(c, h, act, is_last, rvs, um2, um1, u0, up1, up2, e0m2, e0m1, e00, e0p1, e0p2, lamc0,
                                   slot(t) + chunk_off, slot(t + 1) + chunk_off, un, yrow, lrow);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            um2[r] = um1[r];
            um1[r] = u0[r];
            u0[r] = un[r];
        }
#pragma unroll
        for (int k = 0; k < NA; k++) aft[k] += HADI_PAIR_STORES;
        if (h == 0) hist[(t + 2) & 3] = c0p2r;  // (raw: step t + 1 reads it as c0p1, ... step t + 4 has overwritten it)
#if !defined(HADI_EMU)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring re-reads are retired before the next step reuses the slot
#endif
        hadi_wave_rendezvous();
    }
}

// ------------------------------------------------------------------------------------------------
// pass B.  Block = P wavefronts (P*64 threads); wavefront p owns v-rows [p*HADI_LC, (p+1)*HADI_LC) of its
// instance (rows past nrows are identity padding, so there are no tail branches) and
// keeps one 64-column tile of them in registers.  A block walks over `btpw` consecutive column tiles
// with two register buffers: the loads of tile t+1 are in flight while tile t is solved and stored, so
// the memory pipe stays busy through the dependent forward/backward chains.
struct HadiPassBCtx {
    const double *Yi;   // instance base of Y
    double *Ui;         // instance base of U
    HadiBuf Yb, Ub;     // the same two as buffer resources (uniform)
    HadiBuf Lb;         // lambda_bar as a buffer resource (American, explicit pair)
    double *Li;         // instance base of lambda_bar (American)
    const double *P0i;  // instance base of the payoff (American)
    int pay1d;          // the payoff does not depend on v: one load per column instead of one per node
    double inv_dt;      // 1/dt (P representation)
    double tab[5];      // this chunk's table (33 rows x 9 scalars) spread over the lanes: lane l holds entries l + 64 q
    const double *tabl; // hadi_pb_solve<true>: the chunk's table rows in LDS, [HADI_LC][HADI_PBW] (instance-resident kernel)
    const double *Ri;   // this wavefront's four rows of the reduced inverse in LDS, [4P][4]: for column m the
                        // coefficients of (left-neighbour last two, right-neighbour first two)
    double *zsh;        // LDS exchange, 2 buffers of P*4*64 (ONE buffer with the matrix-core reduced system, below)
    const double *RT;   // hadi_pb_solve<.., MF>: the selected rows of the reduced inverse in LDS, TRANSPOSED: RT[k][r], r = 4 w + q the
                        // q-th coefficient row of wavefront w (0, 1: its left neighbour's last two unknowns, 2, 3: its right
                        // neighbour's first two), pitch MP = 16 ceil(P / 4), zero beyond r = 4 P
    double *Tsh;        // ... and its result T = R Z in LDS, [MP][64]
    int lane, wave, P, ja, rowp, american, pos_m1;
    int nrows;          // real v-rows (m2 + 1); rows nrows .. P*HADI_LC-1 are identity padding
    double dt;
    int debug;          // HadiSweepArgs.debug (timing diagnostics only)
    HADI_STAMP_ACC
};

// Scalar m of chunk row k of the column-pass table.  The table is identical for all 64 columns; reading it from LDS costs
// a full-width LDS return per row and phase (a broadcast ds_read_b128 still moves 1 KiB to the VGPRs): ~280 reads per
// tile and wavefront, which kept the LDS pipe of the CU busy for a third of the tile time.  Spread over the lanes of 5
// register pairs and fetched with v_readlane (static lane index, result in SGPRs, used as an FMA operand) it costs
// only VALU slots, of which this kernel has plenty.
#if defined(HADI_EMU)  // same register / lane mapping, read from the image the lanes published after loading the table
#define HADI_PB_T(c, k, m) (emu::t_wave->pub[((k) * 9 + (m)) >> 6][((k) * 9 + (m)) & 63])
#else
#define HADI_PB_T(c, k, m) hadi_read_lane((c).tab[((k) * 9 + (m)) >> 6], ((k) * 9 + (m)) & 63)
#endif

HADI_DEV HADI_FORCEINLINE void hadi_pb_load_table(HadiPassBCtx &c, const double *__restrict__ pbg) {
    static_assert(HADI_LC * 9 <= 5 * 64, "table does not fit 5 registers per lane");
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const int e = c.lane + 64 * q;
        const int k = e / 9, m = e - k * 9;
        // (image positions 0, 1 hold the SCALED forward multipliers PB_LQ, PB_L2Q: the kernels never use the raw ones)
        c.tab[q] = (k < HADI_LC) ? pbg[(size_t)k * HADI_PBW + (m == PB_L ? PB_LQ : m == PB_L2 ? PB_L2Q : m)] : 0.0;
#if defined(HADI_EMU)
        emu::t_wave->pub[q][c.lane] = c.tab[q];
#endif
    }
    hadi_wave_rendezvous();
}

// Storage row of chunk row k.  Identity padding rows (>= nrows; only the last chunk has any) are all mapped to the FIRST
// padding row: they hold zeros in Y (the row pass never writes them) and receive zeros in U, so one cached row serves
// every padded load and store and the padding costs no HBM traffic (scalar min + multiply per row, no VALU).
HADI_DEV HADI_FORCEINLINE unsigned hadi_pb_row(const HadiPassBCtx &c, int k) {
    const int r = c.ja + k;
    return (unsigned)(r < c.nrows ? r : c.nrows);
}

template <class T = double, bool SC1 = false>
HADI_DEV HADI_FORCEINLINE void hadi_pb_load(const HadiPassBCtx &c, int ctile, double (&y)[HADI_LC]) {
    constexpr unsigned ES = (unsigned)sizeof(T);
    const int col = ctile * 64 + c.lane;
    const int colc = col < c.rowp ? col : c.rowp - 1;  // lanes past the pitch read a valid address, never store
    const unsigned voff = (unsigned)colc * ES;
    const unsigned rstride = (unsigned)c.rowp * ES;
#pragma unroll
    for (int k = 0; k < HADI_LC; k++) {
        if constexpr (SC1) y[k] = hadi_buf_load_sc1(c.Yb, voff, hadi_pb_row(c, k) * rstride);  // (agent-coherent: the team kernel)
        else y[k] = hadi_buf_load_t<T>(c.Yb, voff, hadi_pb_row(c, k) * rstride);
    }
}

// D(16 x 16) += A(16 x 4) B(4 x 16) on the matrix core, fp64 (v_mfma_f64_16x16x4_f64).  Operand mapping, verified on gfx950 by
// tools/mfma_probe.hip: lane 16 k + i holds A[i][k], lane 16 k + j holds B[k][j]; register r of lane 16 q + j holds D[4 r + q][j].
HADI_DEV HADI_FORCEINLINE void hadi_mfma_f64_16x16x4(double a, double b, double (&acc)[4]) {
#if defined(HADI_EMU)
    const int q = emu::t_lane >> 4, j = emu::t_lane & 15;
    double av[4][4], bv[4];
    for (int k = 0; k < 4; k++) {
        bv[k] = __shfl(b, 16 * k + j);
        for (int r = 0; r < 4; r++) av[r][k] = __shfl(a, 16 * k + 4 * r + q);
    }
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++) acc[r] = fma(av[r][k], bv[k], acc[r]);
#else
    typedef double hadi_d4 __attribute__((ext_vector_type(4)));
    hadi_d4 c = {acc[0], acc[1], acc[2], acc[3]};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; acc[3] = c[3];
#endif
}
// Stages RT for instance `inst` (whole block; the caller's barrier follows).  R^-1 is the dense inverse of the 4 P x 4 P SPIKE
// reduced system (hadi_setup_instance); wavefront w needs the rows of its left neighbour's last two and its right neighbour's
// first two unknowns (spikes are zero where there is no neighbour, so any row will do there).
HADI_DEV HADI_FORCEINLINE void hadi_pb_stage_rt(const double *__restrict__ Rg, int P, double *__restrict__ RT, int nthreads) {
    const int n4 = 4 * P, MP = hadi_pb_mp(P);
    for (int e = threadIdx.x; e < n4 * MP; e += nthreads) {
        const int k = e / MP, r = e - k * MP, w = r >> 2, q = r & 3;
        double v = 0.0;
        if (w < P) {
            const int rl0 = (w > 0) ? 4 * (w - 1) + 2 : 0, rr0 = (w < P - 1) ? 4 * (w + 1) : 0;
            v = Rg[(size_t)((q < 2) ? rl0 + q : rr0 + (q - 2)) * n4 + k];
        }
        RT[e] = v;
    }
}

// The column pass's LDS (whole block; the caller's barrier follows).  Returns the first free double behind it.
//   MF:      [Z: 4 P x 64] [RT: 4 P x MP] [T: MP x 64]
//   else:    [Z: 1 or 2 buffers of 4 P x 64] [each wavefront's four rows of the reduced inverse: P x 4 x 4 P]
template <bool MF>
HADI_DEV HADI_FORCEINLINE double *hadi_pb_setup_lds(HadiPassBCtx &c, double *smem, const double *__restrict__ Rg, int zbuffers) {
    const int P = c.P, n4 = 4 * P;
    c.zsh = smem;
    c.RT = nullptr; c.Tsh = nullptr; c.Ri = nullptr;
    if constexpr (MF) {
        double *RT = smem + (size_t)n4 * 64;
        hadi_pb_stage_rt(Rg, P, RT, 64 * P);
        c.RT = RT;
        c.Tsh = RT + (size_t)n4 * hadi_pb_mp(P);
        return c.Tsh + (size_t)hadi_pb_mp(P) * 64;
    } else {
        double *tsh = smem + (size_t)P * zbuffers * 4 * 64;
        double *__restrict__ rw = tsh + (size_t)c.wave * 4 * n4;
        const int rl0 = (c.wave > 0) ? 4 * (c.wave - 1) + 2 : 0, rr0 = (c.wave < P - 1) ? 4 * (c.wave + 1) : 0;
        for (int e = c.lane; e < 4 * n4; e += 64) {
            const int m = e >> 2, q = e & 3;
            rw[e] = Rg[(size_t)((q < 2) ? rl0 + q : rr0 + (q - 2)) * n4 + m];
        }
        c.Ri = rw;
        return tsh + (size_t)16 * P * P;
    }
}

// Chunk-local solve + interface exchange + spike correction of one 64-column tile held in y (no memory traffic).
// LDSTAB: the table scalars come from an LDS copy (broadcast reads) instead of v_readlane on the register image.  The
// streaming kernels stream tiles through eight wavefronts and are short of LDS bandwidth, not of VALU slots: readlanes there.
// The instance-resident kernel solves ONE tile per step and waits for it: there the 594 readlanes per tile (two VALU slots
// per scalar) are a third of the phase's instruction chain, and 165 broadcast reads replace them.
// ONEBUF: ONE exchange buffer instead of two alternating ones -- a second block-wide barrier in front of the exchange
// write keeps a fast wavefront from overwriting values a slow one still reads (hadi_pass_b2: the LDS of the second buffer
// holds prefetched rows there).
#ifndef HADI_PB_MF
#define HADI_PB_MF 1  // the streaming column kernels run the reduced system on the matrix core (0: the broadcast-operand FMA loop, for A/B builds)
#endif
// MF: the reduced system t = R^-1 z on the MATRIX CORE.  Every wavefront needs four rows of T = R Z (its own four coefficient
// rows times the 4 P x 64 exchange values of the tile's 64 columns): together the block computes a dense (4 P) x (4 P) x 64
// product per tile.  As 4 x 4 P broadcast-operand FMAs per lane that product was two thirds of the whole solve -- every FMA
// pulled a wave-uniform coefficient through the LDS return path (two broadcast ds_read_b128 per exchange row and wavefront:
// 2.5 MB per 16-chunk tile, ~8 us of LDS pipe per tile against ~1.7 us of arithmetic; gpurun_out/r04g/colpass_ab.txt: the
// 1024x512 column pass 0.136 -> 0.112 ms per launch with the loop removed).  Here wavefront w computes the 16 x 16 block
// (w / 4, w % 4) of T with P v_mfma_f64_16x16x4_f64 (operands: one double per lane each, conflict-free ds_read_b64), writes it
// to LDS, and after a second barrier every wavefront picks its four rows.  ONE exchange buffer: the two barriers of a tile
// order every reuse.  The stencil sweep itself stays on the vector units; this is the one dense contraction of the scheme.
template <bool LDSTAB = false, bool ONEBUF = false, bool MF = false>
HADI_DEV HADI_FORCEINLINE void hadi_pb_solve(const HadiPassBCtx &c, int parity, double (&y)[HADI_LC], int younger = 0) {
#undef HADI_PB_T
#if defined(HADI_EMU)
#define HADI_PB_T(c, k, m) (LDSTAB ? (c).tabl[(k) * HADI_PBW + (m)] : emu::t_wave->pub[((k) * 9 + (m)) >> 6][((k) * 9 + (m)) & 63])
#else
#define HADI_PB_T(c, k, m) (LDSTAB ? (c).tabl[(k) * HADI_PBW + (m)] : hadi_read_lane((c).tab[((k) * 9 + (m)) >> 6], ((k) * 9 + (m)) & 63))
#endif
    HADI_STAMP_DECL(c.stamp_acc_)
    HADI_STAMPB_WAIT(younger);
    HADI_STAMPB(16);  // this tile's loads have landed
    // forward elimination with the chunk-local factorisation
    {
        double ym1 = 0.0, ym2 = 0.0;
#pragma unroll
        for (int k = 0; k < HADI_LC; k++) {
            // (scaled multipliers: one operation on the dependent chain; LDSTAB reads table slots, the register image keeps the
            // scaled pair in positions PB_L, PB_L2)
            const double lq = LDSTAB ? c.tabl[k * HADI_PBW + PB_LQ] : HADI_PB_T(c, k, PB_L);
            const double l2q = LDSTAB ? c.tabl[k * HADI_PBW + PB_L2Q] : HADI_PB_T(c, k, PB_L2);
            const double yk = fma(-lq, ym1, fma(-l2q, ym2, y[k] * HADI_PB_T(c, k, PB_Q)));
            y[k] = yk;
            ym2 = ym1;
            ym1 = yk;
        }
    }
    HADI_STAMPB(17);  // forward
    // back substitution
    {
        double xp1 = 0.0, xp2 = 0.0;
#pragma unroll
        for (int k = HADI_LC - 1; k >= 0; k--) {
            const double xk = fma(-HADI_PB_T(c, k, PB_C), xp1, fma(-HADI_PB_T(c, k, PB_C2), xp2, y[k]));
            y[k] = xk;
            xp2 = xp1;
            xp1 = xk;
        }
    }
    HADI_STAMPB(18);  // backward
    const int P = c.P;
    if (P > 1) {
        // interface exchange.  Two LDS buffers alternate by tile parity, so one barrier per tile is
        // enough: a wave can only overwrite buffer b two tiles later, after every wave has passed the
        // barrier of the tile in between, i.e. has finished reading b.
        double *__restrict__ z = c.zsh + (size_t)((ONEBUF || MF) ? 0 : parity) * P * 4 * 64;
        if constexpr (ONEBUF && !MF) __syncthreads();  // every wavefront has read the previous tile's exchange values
        z[(c.wave * 4 + 0) * 64 + c.lane] = y[0];
        z[(c.wave * 4 + 1) * 64 + c.lane] = y[1];
        z[(c.wave * 4 + 2) * 64 + c.lane] = y[HADI_LC - 2];
        z[(c.wave * 4 + 3) * 64 + c.lane] = y[HADI_LC - 1];
        __syncthreads();
        HADI_STAMPB(19);  // exchange + barrier
        // t = Rinv z : this chunk needs the previous chunk's last two and the next chunk's first two
        const int n4 = 4 * P;
        double tl0 = 0.0, tl1 = 0.0, tr0 = 0.0, tr1 = 0.0;
        if constexpr (MF) {
            const int MP = hadi_pb_mp(P), nblk = MP >> 2;  // (MP / 16 row blocks x 4 column blocks)
            const int kq = c.lane >> 4, ij = c.lane & 15;
            for (int blk = c.wave; blk < ((c.debug & HADI_DEBUG_COL_NO_REDUCED) ? 0 : nblk); blk += P) {
                const int bi = blk >> 2, bj = blk & 3;
                const double *__restrict__ ap = c.RT + kq * MP + 16 * bi + ij;
                const double *__restrict__ bp = z + kq * 64 + 16 * bj + ij;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int sK = 0; sK < P; sK++) hadi_mfma_f64_16x16x4(ap[(size_t)sK * 4 * MP], bp[(size_t)sK * 4 * 64], acc);
#pragma unroll
                for (int r = 0; r < 4; r++) c.Tsh[(16 * bi + 4 * r + kq) * 64 + 16 * bj + ij] = acc[r];
            }
            __syncthreads();
            tl0 = c.Tsh[(4 * c.wave + 0) * 64 + c.lane];
            tl1 = c.Tsh[(4 * c.wave + 1) * 64 + c.lane];
            tr0 = c.Tsh[(4 * c.wave + 2) * 64 + c.lane];
            tr1 = c.Tsh[(4 * c.wave + 3) * 64 + c.lane];
        } else {
            const double *__restrict__ Ri = c.Ri;
#pragma unroll 8
            for (int m = 0; m < ((c.debug & HADI_DEBUG_COL_NO_REDUCED) ? 0 : n4); m++) {
                const double zz = z[m * 64 + c.lane];
                tl0 = fma(Ri[4 * m + 0], zz, tl0);
                tl1 = fma(Ri[4 * m + 1], zz, tl1);
                tr0 = fma(Ri[4 * m + 2], zz, tr0);
                tr1 = fma(Ri[4 * m + 3], zz, tr1);
            }
        }
        HADI_STAMPB(20);  // reduced system
        // spikes are zero where there is no neighbour (first chunk: V = 0, last chunk: W = 0)
#pragma unroll
        for (int k = 0; k < HADI_LC; k++) {
            y[k] = y[k] - HADI_PB_T(c, k, PB_V0) * tl0 - HADI_PB_T(c, k, PB_V1) * tl1 - HADI_PB_T(c, k, PB_W0) * tr0 -
                   HADI_PB_T(c, k, PB_W1) * tr1;
        }
    }
    HADI_STAMPB(21);  // spike correction
}

// Stores the solved tile `ctile` (with the Ikonen-Toivanen projection for American).  RELOAD: every row's register
// is refilled with the same row of tile `ctile + 1` right behind its store, so one register buffer serves both
// tiles and the loads of the next tile are in flight as soon as the stores have been issued.
template <int AMER, bool RELOAD, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_pb_store(const HadiPassBCtx &c, int ctile, double (&y)[HADI_LC], int next_tile = -1) {
    static_assert(sizeof(T) == 8 || AMER == 0, "the fp32-state sweep is European only");
    constexpr unsigned ES = (unsigned)sizeof(T);
    HADI_STAMP_DECL(c.stamp_acc_)
    const int coln = (next_tile < 0 ? ctile + 1 : next_tile) * 64 + c.lane;
    const unsigned voffn = (unsigned)(coln < c.rowp ? coln : c.rowp - 1) * ES;
    const int col = ctile * 64 + c.lane;
    const bool valid = col < c.rowp;
    const int colc = valid ? col : c.rowp - 1;
    const size_t base = (size_t)c.ja * c.rowp + colc;
    if constexpr (AMER == 0) {
        const unsigned voff = (unsigned)colc * ES;
        const unsigned rstride = (unsigned)c.rowp * ES;
        if constexpr (RELOAD) {
            const unsigned voffs = valid ? voff : HADI_BUF_DROP;
            // all stores first, then all loads: issued pairwise (store k, load k) the 1024x512 column pass ran 0.142 ms per
            // launch against 0.133 -- the memory pipe turns round between writes and reads 33 times per wavefront and tile
#pragma unroll
            for (int k = 0; k < HADI_LC; k++) hadi_buf_store_t<T>(c.Ub, voffs, hadi_pb_row(c, k) * rstride, y[k]);
#pragma unroll
            for (int k = 0; k < HADI_LC; k++) y[k] = hadi_buf_load_t<T>(c.Yb, voffn, hadi_pb_row(c, k) * rstride);
        } else if (valid) {
#pragma unroll
            for (int k = 0; k < HADI_LC; k++) hadi_buf_store_t<T>(c.Ub, voff, hadi_pb_row(c, k) * rstride, y[k]);
        }
    } else {
        const unsigned row0 = (unsigned)c.ja * (unsigned)c.rowp * 8u, rstride = (unsigned)c.rowp * 8u;
        // Ikonen-Toivanen projection, device_solver.hpp:358-372.  Plain pointer accesses on purpose: with raw buffer
        // operations hipcc cannot tell that the lambda_bar load of row k+1 does not alias the store of row k, keeps them in
        // program order and exposes one memory latency per row (measured 0.204 vs 0.179 ms/launch at 256x128 x512).
        double *__restrict__ dst = c.Ui + base;
        double *__restrict__ Lb = c.Li + base;
        const double *__restrict__ P0 = c.P0i + base;
        const double dt = c.dt;
        const bool is_smax = (col == c.pos_m1);
        // A call / put payoff depends on s only (every driver of the reference builds U_0 that way,
        // heston_calibration.cpp:183-192): then one load per column replaces 33 (8 of the 40 B per node of this pass).
        // Identity padding rows then see the payoff instead of 0: their results are never read.
        const bool pay1d = (AMER == 2) || c.pay1d != 0;
        const double pay_col = pay1d ? c.P0i[colc] : 0.0;
#pragma unroll
        for (int k = 0; k < HADI_LC; k++) {
            const size_t off = (size_t)k * c.rowp;
            const double U_bar = y[k];
            if constexpr (AMER == 2) {
                // P representation (see hadi_row_step): lambda_bar_old = max(0, (U0 - P_old)/dt), P_new = U_bar - dt lambda_bar_old.
                // One load and one store per node, both on the array that otherwise holds U; no lambda_bar array.
                double lamo = fmax(0.0, (pay_col - dst[off]) * c.inv_dt);
                if (is_smax) lamo = 0.0;
                if (valid) dst[off] = U_bar - dt * lamo;
            } else {
                const double lamv = Lb[off];
                double pay = pay_col;
                if (!pay1d) pay = P0[off];
                const double un = fmax(U_bar - dt * lamv, pay);
                double ln = fmax(0.0, lamv + (pay - U_bar) / dt);
                if (is_smax) ln = 0.0;
                if (valid) {
                    dst[off] = un;
                    Lb[off] = ln;
                }
            }
            if constexpr (RELOAD) y[k] = hadi_buf_load(c.Yb, voffn, row0 + (unsigned)k * rstride);
        }
    }
    HADI_STAMPB(22);  // projection + store issue
}

// American, double-buffered kernel: what the projection of tile `ctile` needs besides the solved values -- the old P
// (P representation) or lambda_bar (explicit pair) -- is fetched into a third register set BEFORE the tile is solved,
// so it arrives during the solve.  Loaded inside the store loop (after the solve's barriers, which no load may cross)
// every tile paid one full memory latency with nothing else to do: 0.28 ms per launch against 0.12 for the European
// column pass on 512x256 x256.
template <int AMER>
HADI_DEV HADI_FORCEINLINE void hadi_pb_load_old(const HadiPassBCtx &c, int ctile, double (&po)[HADI_LC]) {
    const int col = ctile * 64 + c.lane;
    const int colc = col < c.rowp ? col : c.rowp - 1;
    const unsigned voff = (unsigned)colc * 8u;
    const unsigned rstride = (unsigned)c.rowp * 8u;
#pragma unroll
    for (int k = 0; k < HADI_LC; k++) po[k] = hadi_buf_load(AMER == 2 ? c.Ub : c.Lb, voff, hadi_pb_row(c, k) * rstride);
}
// Ikonen-Toivanen projection (device_solver.hpp:358-372) of the solved tile with the prefetched old values; payoff that
// depends on s only (one value per column).  Raw buffer stores; lanes past the pitch are dropped by the range check.
template <int AMER>
HADI_DEV HADI_FORCEINLINE void hadi_pb_store_am(const HadiPassBCtx &c, int ctile, const double (&y)[HADI_LC],
                                                const double (&po)[HADI_LC]) {
    HADI_STAMP_DECL(c.stamp_acc_)
    const int col = ctile * 64 + c.lane;
    const bool valid = col < c.rowp;
    const int colc = valid ? col : c.rowp - 1;
    const unsigned voffs = valid ? (unsigned)colc * 8u : HADI_BUF_DROP;
    const unsigned row0 = (unsigned)c.ja * (unsigned)c.rowp * 8u, rstride = (unsigned)c.rowp * 8u;
    const double dt = c.dt;
    const bool is_smax = (col == c.pos_m1);
    const double pay = c.P0i[colc];
    // (the projection writes the payoff, not zeros, to padding rows: they keep their own storage rows here)
#pragma unroll
    for (int k = 0; k < HADI_LC; k++) {
        const double U_bar = y[k];
        if constexpr (AMER == 2) {
            // lambda_bar_old = max(0, (U0 - P_old)/dt), P_new = U_bar - dt lambda_bar_old (see hadi_row_step)
            double lamo = fmax(0.0, (pay - po[k]) * c.inv_dt);
            if (is_smax) lamo = 0.0;
            hadi_buf_store(c.Ub, voffs, row0 + (unsigned)k * rstride, U_bar - dt * lamo);
        } else {
            const double lamv = po[k];
            const double un = fmax(U_bar - dt * lamv, pay);
            double ln = fmax(0.0, lamv + (pay - U_bar) / dt);
            if (is_smax) ln = 0.0;
            hadi_buf_store(c.Ub, voffs, row0 + (unsigned)k * rstride, un);
            hadi_buf_store(c.Lb, voffs, row0 + (unsigned)k * rstride, ln);
        }
    }
    HADI_STAMPB(22);  // projection + store issue
}

template <int AMER, class T = double>
HADI_DEV HADI_FORCEINLINE void hadi_pb_solve_store(const HadiPassBCtx &c, int ctile, int parity, double (&y)[HADI_LC],
                                                   int younger = 0) {
    if (!(c.debug & HADI_DEBUG_COL_NO_SOLVE)) hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, parity, y, younger);
    hadi_pb_store<AMER, false, T>(c, ctile, y);
}

// Column tiles of block `grp` of an instance.  The pitch is 64*B*G + pad, so the last tile is always a SHORT one (8..32
// columns: little traffic, but a full solve).  The full tiles are dealt out btpw per block and the short tile rides with the
// last block, which is the one that may hold fewer full tiles (1024x512: 16 full tiles on 4 blocks = 4,4,4,4+short instead of
// 5,5,5,2 -- the launch takes 4.3 tile times instead of 5).  With btpw = 1 and one block more than full tiles the short tile
// gets a block of its own (the one-tile-per-block geometry of the small-chunk grids).
// Which full tiles: consecutive ones (block g: g btpw .. g btpw + btpw - 1), or INTERLEAVED (tile_il: block g: g, g + G, g + 2 G,
// ... with G blocks holding full tiles).  The blocks of an instance sit on one XCD in consecutive dispatch slots and walk their
// tiles at the same pace, so interleaved they read -- and write -- G ADJACENT 512-byte segments of every v-row at about the
// same time (one 2 KB piece of a DRAM page per row instead of four pieces 2 KB apart).
struct HadiTileSet {
    int first, stride, nfull_mine, cnt, short_tile;  // tile(i) = i < nfull_mine ? first + i stride : short_tile
};
HADI_DEV HADI_FORCEINLINE HadiTileSet hadi_pb_tiles(const HadiSweepArgs &a, int grp) {
    const int nfull = a.L.rowp >> 6;
    HadiTileSet ts;
    ts.short_tile = nfull;
    int t0 = grp * a.btpw;
    if (t0 > nfull) t0 = nfull;
    const int t1 = (grp == a.bgroups - 1) ? a.ctiles : (t0 + a.btpw < nfull ? t0 + a.btpw : nfull);
    const int has_short = (t1 > nfull) ? 1 : 0;
    if (a.tile_il && nfull > 0) {
        const int gf = (nfull + a.btpw - 1) / a.btpw;  // blocks that hold full tiles
        ts.first = grp; ts.stride = gf;
        ts.nfull_mine = grp < gf ? (nfull - grp + gf - 1) / gf : 0;
    } else {
        ts.first = t0; ts.stride = 1;
        ts.nfull_mine = (t1 < nfull ? t1 : nfull) - t0;
        if (ts.nfull_mine < 0) ts.nfull_mine = 0;
    }
    ts.cnt = ts.nfull_mine + has_short;
    return ts;
}
HADI_DEV HADI_FORCEINLINE int hadi_pb_tile(const HadiTileSet &ts, int i) { return i < ts.nfull_mine ? ts.first + i * ts.stride : ts.short_tile; }

// Dynamic LDS: P * (2*4*64 + 16*P) doubles (two interface-exchange buffers, each wavefront's four rows of the reduced
// inverse); the chunk tables live in registers (HADI_PB_T).
// MAXP only sets the launch bound (register budget): 8 -> 512 threads, 16 -> 1024 threads.
template <int MAXP, int AMER, class T = double>
__global__ void __launch_bounds__(64 * MAXP) hadi_pass_b(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    HadiPassBCtx c;
    c.lane = threadIdx.x & 63;
    c.wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    c.P = a.L.P;
    // blocks walk the instances in DESCENDING order: the row pass writes Y ascending, so the column pass starts on the
    // part of Y that is still in the memory-side cache (and leaves the low instances of U there for the next row pass).
    // XCD-aware: the blocks of one instance (they read the same chunk tables and reduced-inverse rows) get consecutive
    // logical ids on ONE XCD, so the second and third reader find the tables in that XCD's L2.
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= a.n_inst * a.bgroups) return;  // grid padded to a multiple of 8 (whole block: uniform)
    const int binst = logical / a.bgroups, grp = logical - binst * a.bgroups;
    const int inst = a.n_inst - 1 - binst;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;  // whole block: uniform
    const int nrows = a.L.nrows_pad;
    c.nrows = a.L.nrows;
    c.rowp = a.L.rowp;
    c.ja = c.wave * HADI_LC;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;  // (pointer-based accesses: American, T = double only)
    c.Ui = a.U + (size_t)inst * a.L.inst_stride;
    c.Yb = hadi_make_buf(reinterpret_cast<const T *>(a.Y) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Ub = hadi_make_buf(reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.Lb = hadi_make_buf(c.Li, (AMER == 1) ? (size_t)a.L.inst_stride * sizeof(double) : 0);
    c.P0i = AMER ? a.U0 + (size_t)inst * a.L.inst_stride : nullptr;
    c.pay1d = AMER ? (a.pay_mis[inst] == 0) : 0;
    c.inv_dt = 1.0 / ip.dt;
    c.american = a.american; c.debug = a.debug;
    c.pos_m1 = a.pos_m1;
    c.dt = ip.dt;
    c.tabl = nullptr;
    const HadiTileSet ts = hadi_pb_tiles(a, grp);
    const int cnt = ts.cnt;
    auto tile = [&](int i) { return hadi_pb_tile(ts, i); };

#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    double ya[HADI_LC], yb[HADI_LC];
    hadi_pb_load<T>(c, tile(0), ya);
    const bool am_fast = (AMER == 2) || (AMER == 1 && c.pay1d != 0);  // block-uniform
    // the chunk's table (identical for every column) is spread over the lanes' registers once per block
    hadi_pb_load_table(c, a.pb + ((size_t)inst * nrows + c.ja) * HADI_PBW);
    hadi_pb_setup_lds<HADI_PB_MF != 0>(c, smem, a.rinv + (size_t)inst * 16 * c.P * c.P, 2);
    __syncthreads();
    if constexpr (AMER != 0) {
        if (am_fast) {
            double po[HADI_LC];
            for (int i = 0; i < cnt; i += 2) {
                hadi_pb_load_old<AMER>(c, tile(i), po);  // first: it is needed before the next tile's values
                if (i + 1 < cnt) hadi_pb_load<T>(c, tile(i + 1), yb);
                hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, 0, ya, 0);
                hadi_pb_store_am<AMER>(c, tile(i), ya, po);
                if (i + 1 < cnt) {
                    hadi_pb_load_old<AMER>(c, tile(i + 1), po);
                    if (i + 2 < cnt) hadi_pb_load<T>(c, tile(i + 2), ya);
                    hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, 1, yb, 0);
                    hadi_pb_store_am<AMER>(c, tile(i + 1), yb, po);
                }
            }
            return;
        }
    }
    // European: THREE register buffers (3 x 33 rows = 198 VGPRs of 256) -- the loads of two tiles are in flight while one
    // is solved and stored.  With two buffers a tile's loads had only the short solve of its predecessor (~4 k cycles)
    // to arrive in, less than the memory latency under load; measured 512x256 x256: 0.117 -> 0.113 ms per launch at the
    // same 3 tiles per block (a block then has all its loads in flight from the start).
    if constexpr (AMER == 0) {
        double yc[HADI_LC];
        if (1 < cnt) hadi_pb_load<T>(c, tile(1), yb);
        for (int i = 0; i < cnt; i += 3) {
            if (i + 2 < cnt) hadi_pb_load<T>(c, tile(i + 2), yc);
            hadi_pb_solve_store<AMER, T>(c, tile(i), i & 1, ya, 0);
            if (i + 1 < cnt) {
                if (i + 3 < cnt) hadi_pb_load<T>(c, tile(i + 3), ya);
                hadi_pb_solve_store<AMER, T>(c, tile(i + 1), (i + 1) & 1, yb, 0);
            }
            if (i + 2 < cnt) {
                if (i + 4 < cnt) hadi_pb_load<T>(c, tile(i + 4), yb);
                hadi_pb_solve_store<AMER, T>(c, tile(i + 2), (i + 2) & 1, yc, 0);
            }
        }
        return;
    }
    for (int i = 0; i < cnt; i += 2) {  // American with a payoff that depends on v: two buffers, loads inside the store loop
        // `younger` = vector-memory operations issued after the loads of the tile being solved (diagnostic build only)
        if (i + 1 < cnt) hadi_pb_load<T>(c, tile(i + 1), yb);
        hadi_pb_solve_store<AMER, T>(c, tile(i), 0, ya, (i + 1 < cnt ? HADI_LC : 0) + (i > 0 ? HADI_LC : 0));
        if (i + 2 < cnt) hadi_pb_load<T>(c, tile(i + 2), ya);
        if (i + 1 < cnt) hadi_pb_solve_store<AMER, T>(c, tile(i + 1), 1, yb, (i + 2 < cnt ? HADI_LC : 0) + HADI_LC);
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 3 && c.lane == 0)
        for (int k = 16; k < 24; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// Single-buffer variant for more than 8 chunks (m2 > 263): a 1024-thread block has 128 VGPRs per lane, too few for two
// 33-row register buffers (the double-buffered code spills 650 B per lane there).  The next tile is loaded into the
// registers of the current one row by row, right behind the stores.  Measured on MI355X: 1024x512 grid 0.250 ms per
// launch against 0.382; at 512x256 (P = 8) the double-buffered kernel above wins, 0.144 against 0.206.
template <int MAXP, int AMER, class T = double>
__global__ void __launch_bounds__(64 * MAXP, 4) hadi_pass_b1(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    HadiPassBCtx c;
    c.lane = threadIdx.x & 63;
    c.wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    c.P = a.L.P;
    // blocks walk the instances in DESCENDING order: the row pass writes Y ascending, so the column pass starts on the
    // part of Y that is still in the memory-side cache (and leaves the low instances of U there for the next row pass).
    // XCD-aware: the blocks of one instance (they read the same chunk tables and reduced-inverse rows) get consecutive
    // logical ids on ONE XCD, so the second and third reader find the tables in that XCD's L2.
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);
    if (logical >= a.n_inst * a.bgroups) return;  // grid padded to a multiple of 8 (whole block: uniform)
    const int binst = logical / a.bgroups, grp = logical - binst * a.bgroups;
    const int inst = a.n_inst - 1 - binst;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;  // whole block: uniform
    const int nrows = a.L.nrows_pad;
    c.nrows = a.L.nrows;
    c.rowp = a.L.rowp;
    c.ja = c.wave * HADI_LC;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;  // (pointer-based accesses: American, T = double only)
    c.Ui = a.U + (size_t)inst * a.L.inst_stride;
    c.Yb = hadi_make_buf(reinterpret_cast<const T *>(a.Y) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Ub = hadi_make_buf(reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Li = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride : nullptr;
    c.Lb = hadi_make_buf(c.Li, (AMER == 1) ? (size_t)a.L.inst_stride * sizeof(double) : 0);
    c.P0i = AMER ? a.U0 + (size_t)inst * a.L.inst_stride : nullptr;
    c.pay1d = AMER ? (a.pay_mis[inst] == 0) : 0;
    c.inv_dt = 1.0 / ip.dt;
    c.american = a.american; c.debug = a.debug;
    c.pos_m1 = a.pos_m1;
    c.dt = ip.dt;
    c.tabl = nullptr;
    const HadiTileSet ts = hadi_pb_tiles(a, grp);
    const int cnt = ts.cnt;
    auto tile = [&](int i) { return hadi_pb_tile(ts, i); };

#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    double y[HADI_LC];
    hadi_pb_load<T>(c, tile(0), y);
    // the chunk's table (identical for every column) is spread over the lanes' registers once per block
    hadi_pb_load_table(c, a.pb + ((size_t)inst * nrows + c.ja) * HADI_PBW);
    hadi_pb_setup_lds<HADI_PB_MF != 0>(c, smem, a.rinv + (size_t)inst * 16 * c.P * c.P, 2);
    __syncthreads();
    // (fp32 state: holding the NEXT tile in 33 float registers so that its loads fly during the solve was tried -- 66 + 33 +
    // 10 table registers leave too few of the 128 for the reduced-system loop, the kernel spills 18 registers and the
    // scratch reloads drain the prefetch: 0.088 -> 0.099 ms per launch at 1024x512 x64.)
    for (int i = 0; i < cnt; i++) {
        if (!(a.debug & HADI_DEBUG_COL_NO_SOLVE)) hadi_pb_solve<false, false, HADI_PB_MF != 0>(c, i & 1, y, 0);
        if (i + 1 < cnt) hadi_pb_store<AMER, true, T>(c, tile(i), y, tile(i + 1));
        else hadi_pb_store<AMER, false, T>(c, tile(i), y);
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 3 && c.lane == 0)
        for (int k = 16; k < 24; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// ------------------------------------------------------------------------------------------------
// hadi_pass_b2: the single-buffer column pass with part of the NEXT tile prefetched into LDS (9 .. 16 chunks, European).
// hadi_pass_b1 cannot hold a second tile -- 16 wavefronts per CU leave 128 VGPRs per lane, and a 16-chunk tile is 263 KB
// against the 512 KB of a CU's whole register file -- so it stores a tile, loads the next one behind the stores and waits:
// every tile pays the turn-round of the memory pipe and a full load latency with nothing in flight during the solve
// (PMC: VALU busy 0.08, waiting 0.37 of the wave cycles).  LDS-DMA needs no registers: here every wavefront fetches the first
// NPF rows of its chunk of tile t+1 into a private LDS area BEFORE it solves tile t, so they fly during the solve and the
// stores; only the other 33 - NPF rows are loaded into the registers behind the stores.  The LDS comes from the second
// exchange buffer (one buffer + a second barrier per tile: hadi_pb_solve<.., ONEBUF>): 16 chunks: 32 KB exchange + 32 KB
// reduced-inverse rows + 16 x NPF x 64 elements.  One dwordx4 DMA instruction moves 64 x 16 B = RPI rows of this tile (a row
// is 64 columns = 512 B as doubles, 256 B as floats): RPI = 2 resp. 4 rows, lanes grouped by row.
// Completion: the wavefront's explicit s_waitcnt vmcnt(0) behind the register loads (they are younger than the DMA and the
// stores; everything has to be there before the solve anyway) -- no counted waits, nothing depends on the retirement order
// of different kinds of vector-memory operations.
template <class T, int NPF>
HADI_DEV HADI_FORCEINLINE void hadi_pb_dma(const HadiPassBCtx &c, const T *__restrict__ Yt, int ctile, T *pf) {
    constexpr int ES = (int)sizeof(T), EPV = 16 / ES, LPR = 64 / EPV, RPI = 64 / LPR;
    static_assert(NPF % RPI == 0 && NPF <= HADI_LC, "whole DMA instructions");
    const int sub = c.lane / LPR, l = c.lane - sub * LPR;
    int col = ctile * 64 + EPV * l;
    if (col + EPV > c.rowp) col = c.rowp - EPV;  // lanes past the pitch (short tile) fetch a valid address; their columns are never stored
#if !defined(HADI_EMU)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)pf);
#endif
#pragma unroll
    for (int q = 0; q < NPF / RPI; q++) {
        const unsigned row = hadi_pb_row(c, q * RPI + sub);  // (per lane: the lanes of one instruction cover RPI rows)
#if defined(HADI_EMU)
        for (int e = 0; e < EPV; e++) pf[(q * RPI + sub) * 64 + EPV * l + e] = Yt[(size_t)row * c.rowp + col + e];
#else
        const unsigned voff = (row * (unsigned)c.rowp + (unsigned)col) * (unsigned)ES;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" HADI_DMA_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(Yt), "s"(lds0 + 1024u * q)
                     : "memory");
#endif
    }
}
// rows [K0, HADI_LC) of tile `ctile` into the registers (hadi_pb_load for a part of the chunk)
template <class T, int K0>
HADI_DEV HADI_FORCEINLINE void hadi_pb_load_from(const HadiPassBCtx &c, int ctile, double (&y)[HADI_LC]) {
    constexpr unsigned ES = (unsigned)sizeof(T);
    const int col = ctile * 64 + c.lane;
    const int colc = col < c.rowp ? col : c.rowp - 1;
    const unsigned voff = (unsigned)colc * ES;
    const unsigned rstride = (unsigned)c.rowp * ES;
#pragma unroll
    for (int k = K0; k < HADI_LC; k++) y[k] = hadi_buf_load_t<T>(c.Yb, voff, hadi_pb_row(c, k) * rstride);
}

template <int MAXP, class T, int NPF>
__global__ void __launch_bounds__(64 * MAXP, 4) hadi_pass_b2(HadiSweepArgs a, int n) {
    HADI_DYN_SMEM(double, smem);
    HadiPassBCtx c;
    c.lane = threadIdx.x & 63;
    c.wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    c.P = a.L.P;
    const int logical = hadi_xcd_remap(blockIdx.x, gridDim.x);  // (block order and XCD placement as in hadi_pass_b)
    if (logical >= a.n_inst * a.bgroups) return;
    const int binst = logical / a.bgroups, grp = logical - binst * a.bgroups;
    const int inst = a.n_inst - 1 - binst;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int nrows = a.L.nrows_pad;
    c.nrows = a.L.nrows;
    c.rowp = a.L.rowp;
    c.ja = c.wave * HADI_LC;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;
    c.Ui = a.U + (size_t)inst * a.L.inst_stride;
    const T *__restrict__ Yt = reinterpret_cast<const T *>(a.Y) + (size_t)inst * a.L.inst_stride;
    c.Yb = hadi_make_buf(Yt, (size_t)a.L.inst_stride * sizeof(T));
    c.Ub = hadi_make_buf(reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride, (size_t)a.L.inst_stride * sizeof(T));
    c.Li = nullptr; c.Lb = hadi_make_buf(nullptr, 0); c.P0i = nullptr; c.pay1d = 0;
    c.inv_dt = 1.0 / ip.dt;
    c.american = 0; c.debug = a.debug;
    c.pos_m1 = a.pos_m1;
    c.dt = ip.dt;
    c.tabl = nullptr;
    const HadiTileSet ts = hadi_pb_tiles(a, grp);
    const int cnt = ts.cnt;
    auto tile = [&](int i) { return hadi_pb_tile(ts, i); };
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[32] = {0};
    c.stamp_acc_ = stamp_store_;
#endif
    double y[HADI_LC];
    hadi_pb_load<T>(c, tile(0), y);
    hadi_pb_load_table(c, a.pb + ((size_t)inst * nrows + c.ja) * HADI_PBW);
    // ONE exchange buffer either way; this wavefront's prefetch area behind the reduced system's LDS
    T *const pf = reinterpret_cast<T *>(hadi_pb_setup_lds<HADI_PB_MF != 0>(c, smem, a.rinv + (size_t)inst * 16 * c.P * c.P, 1)) + (size_t)c.wave * NPF * 64;
    __syncthreads();
    for (int i = 0; i < cnt; i++) {
        const bool more = i + 1 < cnt;  // (block-uniform)
        const int t = tile(i), tn = tile(more ? i + 1 : i);
        if (more) hadi_pb_dma<T, NPF>(c, Yt, tn, pf);  // flies during the solve and the stores of tile t
        if (!(a.debug & HADI_DEBUG_COL_NO_SOLVE)) hadi_pb_solve<false, true, HADI_PB_MF != 0>(c, 0, y, 0);
        if (more) {
            hadi_pb_store<0, false, T>(c, t, y);
            hadi_pb_load_from<T, NPF>(c, tn, y);
#if !defined(HADI_EMU)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA (older than everything above) has landed
#endif
            hadi_wave_rendezvous();
#pragma unroll
            for (int k = 0; k < NPF; k++) y[k] = (double)pf[k * 64 + c.lane];
#if !defined(HADI_EMU)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and has been read: the next DMA may overwrite the area
#endif
            hadi_wave_rendezvous();
        } else {
            hadi_pb_store<0, false, T>(c, t, y);
        }
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (HADI_STAMPS == 3 && c.lane == 0)
        for (int k = 16; k < 24; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
}

// ------------------------------------------------------------------------------------------------
// Instance-resident execution: the WHOLE time loop of up to 8 large instances in ONE launch (European Douglas sweeps, fp64,
// one wavefront per v-row: 128 < m1 <= 512, m2 <= 263) -- what the reference's team kernel does for every instance
// (device_solver.hpp:83-88, 226-265: all N steps inside one kernel).  The batched path above needs 2 N dependent launches;
// for ONE 512x256 instance each of them is a few microseconds of work on a sliver of the chip behind a ~1.5 us kernel
// boundary, 17.6 ms per 1000 steps.  Here a TEAM of `nb` blocks, all on the same XCD, keeps the instance in that XCD's L2:
//   row phase     the team's 8 nb wavefronts take the v-rows round-robin: five rows of U straight to registers (L1-bypassing
//                 loads), hadi_strip_step, Y stored;
//   team barrier  every wavefront drains its stores (they are in the XCD's L2 then), one lane per block adds to a
//                 monotonic counter in L2 and polls it;
//   column phase  block t of the team takes column tile t: hadi_pb_load / hadi_pb_solve / hadi_pb_store as in hadi_pass_b;
//   team barrier.
// Which XCD a block runs on is READ from the hardware (HW_REG_XCC_ID), not inferred from blockIdx: blocks that read the
// same id share an L2, so the stores one of them has retired are what the L1-bypassing loads of the others return -- no L2
// write-back, no invalidate, which is what makes the barrier cost ~1 us instead of the 4-5 us of a chip-wide one.  Team k
// = the blocks on XCD k, instance k.  Every wait is bounded; a team that does not fill up (the dispatcher owes nobody a
// round-robin placement), a barrier that runs out of polls, or a block that finds itself on another XCD after a barrier
// (wave save / restore by the driver) records HADI_DEVERR_TEAM in the handle's error word, and the host solves the batch
// again on the streaming path.
struct HadiTeamArgs {
    int *form;   // [8] arrival counters, one per XCD (zeroed before the launch)
    int *bar;    // [8] monotonic barrier counters, one per XCD (zeroed before the launch), each on a cache line of its own
    int nb;      // blocks per team
    int N;       // time steps
    unsigned long long *stamps;  // diagnostic build only (HADI_TEAM_STAMPS): [16]
};
#define HADI_DEVERR_TEAM 2  // instance-resident launch: a team did not form, a team barrier timed out, or a block moved
#define HADI_TEAM_POLLS (1 << 18)

// Diagnostic build only (-DHADI_TEAM_STAMPS, tools/team_stamps.py): shader-clock stamps of the phases of time step
// HADI_TEAM_STAMPS, written by wavefront 0 of the team's blocks 0 (a column-phase block) and nb - 1 (a row-phase-only block).
#if defined(HADI_TEAM_STAMPS) && !defined(HADI_EMU)
#define HADI_TSTAMP(k, drain) do { if (n == HADI_TEAM_STAMPS && wave == 0 && (rank == 0 || rank == nb - 1)) { \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    if (lane == 0) ta.stamps[(rank == 0 ? 0 : 8) + (k)] = t_; } } while (0)
#else
#define HADI_TSTAMP(k, drain)
#endif

HADI_DEV HADI_FORCEINLINE int hadi_xcc_id() {
#if defined(HADI_EMU)
    return (int)(blockIdx.x & 7);
#else
    return (int)(__builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 7);
#endif
}
// Cross-CU visibility inside a team (HADI_TEAM_COHERENCE): how a wavefront is guaranteed to see what ANOTHER CU of its XCD wrote
// before the team barrier.  The writer side is the same either way -- every wavefront drains its stores (the vector L1 is
// write-through: an acknowledged store is in the XCD's L2) before its block arrives at the barrier.
//   1  reader: agent-coherent loads (cache policy sc1, on top of nt): served by the L2, never by this CU's vector L1 -- the
//      gfx942 / gfx950 memory model's "load atomic monotonic, agent scope", applied to the only loads that cross CUs (the five
//      stencil rows of the row phase, the tile of the column phase).  No invalidate.
//   2  reader: `buffer_inv sc1` behind the barrier's poll (the model's agent-scope acquire), plain nt loads.  Measured 1.7 us
//      per time step slower than (1) on 512x256 (16.2 -> 17.9 ms per 1000 steps, profiles/r04_team_ab.txt): the invalidate also
//      throws out the row tables and the lines the next phase's first loads would have hit.
//   0  nt loads only (round 3): nt is a streaming HINT, not a coherence guarantee.  Kept for A/B timing only.
#ifndef HADI_TEAM_COHERENCE
#define HADI_TEAM_COHERENCE 1
#endif
// this lane's B values of a row-layout global row of the instance behind `ub` (byte offset `row_bytes`, wave-uniform)
template <int B>
HADI_DEV HADI_FORCEINLINE void hadi_get_block_l2(HadiBuf ub, const double *row, unsigned row_bytes, int lane, double (&u)[B]) {
#if !defined(HADI_EMU)
    typedef double hadi_d2 __attribute__((ext_vector_type(2)));
#endif
#pragma unroll
    for (int q = 0; q < B / 2; q++) {
#if defined(HADI_EMU)
        (void)ub; (void)row_bytes;
        u[2 * q] = row[q * 128 + 2 * lane]; u[2 * q + 1] = row[q * 128 + 2 * lane + 1];
#elif HADI_TEAM_COHERENCE == 1
        (void)row;
        hadi_buf_load2_sc1(ub, (unsigned)(q * 128 + 2 * lane) * 8u, row_bytes, u[2 * q], u[2 * q + 1]);
#else
        (void)ub; (void)row_bytes;
        const hadi_d2 t = __builtin_nontemporal_load(reinterpret_cast<const hadi_d2 *>(row + q * 128 + 2 * lane));
        u[2 * q] = t.x; u[2 * q + 1] = t.y;
#endif
    }
}
HADI_DEV HADI_FORCEINLINE double hadi_get_l2(HadiBuf ub, const double *p, unsigned off_bytes) {
#if defined(HADI_EMU)
    (void)ub; (void)off_bytes;
    return *p;
#elif HADI_TEAM_COHERENCE == 1
    (void)p;
    return hadi_buf_load_sc1(ub, 0u, off_bytes);
#else
    (void)ub; (void)off_bytes;
    return __builtin_nontemporal_load(p);
#endif
}
// Barrier of the team's blocks.  `dead` (LDS) is set when the wait ran out of polls or the block has moved; returns false then.
HADI_DEV HADI_FORCEINLINE bool hadi_team_barrier(int *ctr, int target, int xcc_team, int *dead, int *err) {
#if !defined(HADI_EMU)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores have been acknowledged by the L2
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
#if defined(HADI_EMU)
        __atomic_fetch_add(ctr, 1, __ATOMIC_SEQ_CST);
        int guard = 0;
        while (__atomic_load_n(ctr, __ATOMIC_SEQ_CST) < target && ++guard < HADI_TEAM_POLLS) sched_yield();
#else
        // Release side: every wavefront of the block drained its stores above (the vector L1 is write-through: an
        // acknowledged store IS in this XCD's L2), so the counter update itself can be relaxed.  An agent-scope RELEASE
        // would add `buffer_wbl2 sc1` -- a write-back of the L2's dirty lines, i.e. of the instance the team keeps there on
        // purpose -- for readers that share this very L2 (checked below through HW_REG_XCC_ID).
        __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int guard = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++guard < HADI_TEAM_POLLS)
            __builtin_amdgcn_s_sleep(2);
        // Acquire side: see HADI_TEAM_COHERENCE above -- agent-coherent (sc1) loads of everything that crosses CUs, or, in the
        // alternative build, an invalidate of this CU's vector L1 right here.
#if HADI_TEAM_COHERENCE == 2
        asm volatile("buffer_inv sc1" ::: "memory");
#endif
#endif
        if (guard >= HADI_TEAM_POLLS || hadi_xcc_id() != xcc_team) {
            hadi_report(err, HADI_DEVERR_TEAM);
            *dead = 1;
        }
    }
    __syncthreads();
    return *dead == 0;
}

template <int B>
__global__ void __launch_bounds__(512, 2) hadi_team_kernel(HadiSweepArgs a, HadiTeamArgs ta) {
    HADI_DYN_SMEM(double, smem);
    constexpr int c0slot = 64 * B;
    const int lane = threadIdx.x & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    const int xcc = hadi_xcc_id();
    if (xcc >= a.n_inst) return;  // (block-uniform: a workgroup lives on one XCD)
    // LDS: [4 coefficient arrays of 64 B] [the column pass's reduced system: exchange values Z, selected inverse rows RT, their
    //      product T (hadi_pb_mf_doubles)] [P chunk tables of the column pass] [flags]
    const int P = a.L.P, n4 = 4 * P;
    double *coef = smem;
    double *zsh = coef + 4 * 64 * B;
    double *rtsh = zsh + (size_t)n4 * 64;
    double *tprod = rtsh + (size_t)n4 * hadi_pb_mp(P);
    double *tabl = tprod + (size_t)hadi_pb_mp(P) * 64;  // the column-pass chunk tables, [P][HADI_LC][HADI_PBW]
    int *flags = reinterpret_cast<int *>(tabl + (size_t)P * HADI_LC * HADI_PBW);  // [0] rank of this block in its team, [1] dead
    if (threadIdx.x == 0) {
#if defined(HADI_EMU)
        flags[0] = __atomic_fetch_add(ta.form + xcc, 1, __ATOMIC_SEQ_CST);
#else
        flags[0] = __hip_atomic_fetch_add(ta.form + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        flags[1] = 0;
    }
    __syncthreads();
    const int rank = HADI_UNIFORM(flags[0]);
    const int nb = ta.nb;
    if (rank >= nb) return;  // more blocks of the grid landed on this XCD than the team takes
    if ((a.debug & HADI_DEBUG_TEAM_DESERT) && rank == 1) return;  // (test hook)
    const int inst = xcc;
    const HadiInstPar ip = a.ipar[inst];
    const int nrows = a.L.nrows, rowp = a.L.rowp;
    {   // s-coefficient arrays, the beta pair scaled as in hadi_pass_a_strip; this wavefront's rows of the reduced inverse
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B;
        const double mq = -(ip.thdt * ip.q);
        for (int e = threadIdx.x; e < 4 * 64 * B; e += 512) coef[e] = (e < 2 * 64 * B) ? mq * sc[e] : sc[e];
        hadi_pb_stage_rt(a.rinv + (size_t)inst * 16 * P * P, P, rtsh, 512);
        const double *__restrict__ pg = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
        for (int e = threadIdx.x; e < P * HADI_LC * HADI_PBW; e += 512) tabl[e] = pg[e];
    }
    __syncthreads();

    HadiStripCtxT<double> c;
    c.lane = lane; c.rowp = rowp; c.coef = coef; c.half = 0; c.xch = nullptr; c.err = a.err; c.debug = 0;
    c.dt = hadi_uniform_d(ip.dt); c.thdt = hadi_uniform_d(ip.thdt);
    c.c1 = hadi_uniform_d(1.0 + ip.thdt * ip.half_rd);
    c.kap = hadi_uniform_d((ip.dt - ip.thdt) / ip.thdt);
    c.hr0 = hadi_uniform_d(ip.hr0); c.inv0 = hadi_uniform_d(1.0 / (1.0 + ip.thdt * ip.hr0));
    double *const Ui = a.U + (size_t)inst * a.L.inst_stride;
    c.Yi = a.Y + (size_t)inst * a.L.inst_stride;
    c.Li = nullptr;
    c.b2r = a.b2row + (size_t)inst * rowp;
    c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;

    HadiPassBCtx cb;
    cb.lane = lane; cb.wave = wave; cb.P = P; cb.zsh = zsh; cb.Ri = nullptr; cb.RT = rtsh; cb.Tsh = tprod;
    cb.nrows = nrows; cb.rowp = rowp; cb.ja = wave * HADI_LC;
    cb.Yi = c.Yi; cb.Ui = Ui;
    cb.Yb = hadi_make_buf(c.Yi, (size_t)a.L.inst_stride * sizeof(double));
    cb.Ub = hadi_make_buf(Ui, (size_t)a.L.inst_stride * sizeof(double));
    cb.Li = nullptr; cb.Lb = hadi_make_buf(nullptr, 0); cb.P0i = nullptr; cb.pay1d = 0; cb.inv_dt = 0.0;
    cb.american = 0; cb.debug = 0; cb.pos_m1 = a.pos_m1; cb.dt = ip.dt;
    cb.tabl = tabl + (size_t)wave * HADI_LC * HADI_PBW;
    if (wave < P) hadi_pb_load_table(cb, a.pb + ((size_t)inst * a.L.nrows_pad + cb.ja) * HADI_PBW);

    const int wt = rank * 8 + wave, nwt = nb * 8;  // this wavefront's number in the team
    int *const bar = ta.bar + 32 * xcc;
    int arrivals = 0;
    const int N = ip.N < ta.N ? ip.N : ta.N;
    // boundary time factors e_n = exp(bc_rate dt n) (device_solver.hpp:238,246): one exp per step (e_{n-1} is last step's
    // e_n, the same bits), none for the call with r_f = 0 (exp(0) = 1 exactly)
    const bool unit_e = (ip.bc_rate == 0.0);
    double e_cur = 1.0;  // exp(bc_rate dt 0)
    for (int n = 1; n <= N; n++) {
        c.e_nm1 = hadi_uniform_d(e_cur);
        if (!unit_e) e_cur = exp(ip.bc_rate * ip.dt * n);
        c.e_n = hadi_uniform_d(e_cur);
        HADI_TSTAMP(0, false);
        // ---- row phase ---------------------------------------------------------------------------------------------
        for (int j = (a.debug & HADI_DEBUG_TEAM_NO_ROWS) ? nrows : wt; j < nrows; j += nwt) {
            HadiSRow srow;
            hadi_sload_issue(a.rowc + ((size_t)inst * nrows + j) * HADI_RC + HADI_SRC0, srow);
            double um2[B], um1[B], u0[B], up1[B], up2[B], un[B], praw[B];
#pragma unroll
            for (int r = 0; r < B; r++) um2[r] = um1[r] = up1[r] = up2[r] = praw[r] = 0.0;
            double c0m2 = 0.0, c0m1 = 0.0, c0p1 = 0.0, c0p2 = 0.0;
            const double *r0 = Ui + (size_t)j * rowp;
            const unsigned rb = (unsigned)rowp * 8u, o0 = (unsigned)j * rb, oc = (unsigned)c0slot * 8u;  // (byte offsets inside the instance)
            if (j >= 2) { hadi_get_block_l2<B>(cb.Ub, r0 - 2 * rowp, o0 - 2 * rb, lane, um2); c0m2 = hadi_get_l2(cb.Ub, r0 - 2 * rowp + c0slot, o0 - 2 * rb + oc); }
            if (j >= 1) { hadi_get_block_l2<B>(cb.Ub, r0 - rowp, o0 - rb, lane, um1); c0m1 = hadi_get_l2(cb.Ub, r0 - rowp + c0slot, o0 - rb + oc); }
            hadi_get_block_l2<B>(cb.Ub, r0, o0, lane, u0);
            const double c00 = hadi_get_l2(cb.Ub, r0 + c0slot, o0 + oc);
            if (j + 1 < nrows) { hadi_get_block_l2<B>(cb.Ub, r0 + rowp, o0 + rb, lane, up1); c0p1 = hadi_get_l2(cb.Ub, r0 + rowp + c0slot, o0 + rb + oc); }
            if (j + 2 < nrows) { hadi_get_block_l2<B>(cb.Ub, r0 + 2 * rowp, o0 + 2 * rb, lane, up2); c0p2 = hadi_get_l2(cb.Ub, r0 + 2 * rowp + c0slot, o0 + 2 * rb + oc); }
            double rt[HADI_RCL];
            hadi_sload_wait(srow, rt);
            hadi_wave_rendezvous();
            if (j == nrows - 1)
                hadi_strip_step<B, 0, true, double, 1>(c, j, rt, um2, um1, u0, up1, up2, hadi_uniform_d(c0m2), hadi_uniform_d(c0m1), hadi_uniform_d(c00),
                                                       hadi_uniform_d(c0p1), hadi_uniform_d(c0p2), praw, 0.0, coef, un);
            else
                hadi_strip_step<B, 0, false, double, 1>(c, j, rt, um2, um1, u0, up1, up2, hadi_uniform_d(c0m2), hadi_uniform_d(c0m1), hadi_uniform_d(c00),
                                                        hadi_uniform_d(c0p1), hadi_uniform_d(c0p2), praw, 0.0, coef, un);
        }
        HADI_TSTAMP(1, false);
        arrivals += nb;
        if (!(a.debug & HADI_DEBUG_TEAM_NO_BARRIER) && !hadi_team_barrier(bar, arrivals, xcc, flags + 1, a.err)) return;
        HADI_TSTAMP(2, false);
        // ---- column phase: tile t on block t of the team ---------------------------------------------------------------
        for (int t = (a.debug & HADI_DEBUG_TEAM_NO_COLS) ? a.ctiles : rank; t < a.ctiles; t += nb) {
            if (wave < P) {
                double y[HADI_LC];
                hadi_pb_load<double, HADI_TEAM_COHERENCE == 1>(cb, t, y);
                HADI_TSTAMP(3, true);   // (diagnostic build: waits for the loads)
                hadi_pb_solve<false, false, true>(cb, 0, y, 0);  // (reduced system on the matrix core: two block barriers inside)
                HADI_TSTAMP(4, false);
                hadi_pb_store<0, false, double>(cb, t, y);
                HADI_TSTAMP(5, false);
            } else if (P > 1) {
                __syncthreads();  // (the two barriers inside hadi_pb_solve: exchange values written, product written)
                __syncthreads();
            }
        }
        arrivals += nb;
        if (!(a.debug & HADI_DEBUG_TEAM_NO_BARRIER) && !hadi_team_barrier(bar, arrivals, xcc, flags + 1, a.err)) return;
        HADI_TSTAMP(6, false);
    }
}

// ------------------------------------------------------------------------------------------------
// Small grids (the reference's calibration / perf-harness sizes, e.g. 50x25: perfomance_test.cpp:46-57):
// the whole instance lives in LDS and ONE launch runs the entire time loop -- no HBM traffic and no kernel
// boundaries inside the loop.  Block = W wavefronts <-> one instance (W = 4: measured faster than 16 on MI355X,
// 0.37 vs 0.46 ms for 500 instances of 50x25x20 -- more blocks per CU beat more waves per instance).  Per step: the row pass is the same
// hadi_row_step as above (rows taken straight from the LDS-resident state, Y written to LDS), then the
// column pass walks each column sequentially in LDS (single chunk: m2+1 <= HADI_LC), with the American
// projection; discrete dividends are applied in place.
struct HadiSmallArgs {
    const int *div_flag;        // dividend index applied at the START of step n (n = 1..Nmax), or -1; nullptr = none.
                                // Instance k reads div_flag[k*flag_stride + n-1]: flag_stride 0 = one shared (N, dt)
    int flag_stride;
    const double *div_amounts;  // device copies of the schedule
    const double *div_pcts;
    const double *vec_s;        // [n_inst][m1+1] (dividend interpolation)
    int Nmax;
    const int *order;           // block b solves instance order[b] (longest time loops first), nullptr = identity
};

template <int B, int W, bool AMER>
__global__ void __launch_bounds__(64 * W) hadi_small_kernel(HadiSweepArgs a, HadiSmallArgs sm) {
    HADI_DYN_SMEM(double, smem);
    constexpr int G = 1, NT = 64 * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = HADI_UNIFORM((int)(threadIdx.x >> 6));
    if ((int)blockIdx.x >= a.n_inst) return;
    // multi-maturity batches: instances with many time steps are dispatched first, short ones fill the tail
    const int inst = sm.order ? sm.order[blockIdx.x] : (int)blockIdx.x;
    const HadiInstPar ip = a.ipar[inst];
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1;
    constexpr int c0slot = 64 * B;
    // LDS map: U with two zero rows above and below, Y, [lambda, payoff], coefficients, row table, column table
    const int rows_l = nrows + 4;
    double *Ul = smem + 2 * rowp;                 // row 0 of U (rows -2, -1 and nrows, nrows+1 are zero)
    double *Yl = smem + (size_t)rows_l * rowp;    // nrows rows
    double *LAMl = Yl + (size_t)nrows * rowp;
    double *U0l = LAMl + (AMER ? (size_t)nrows * rowp : 0);
    double *coef = U0l + (AMER ? (size_t)nrows * rowp : 0);
    double *rtab = coef + 4 * 64 * B;
    double *ptab = rtab + (size_t)nrows * HADI_RCL;

    double *__restrict__ Ug = a.U + (size_t)inst * a.L.inst_stride;
    // zero the halo rows of U and ALL of Y: the column pass also sweeps the pad slots of every row, which the row pass
    // never writes -- whatever LDS held there (possibly NaN) would reach U's pad slot, and lane 63 multiplies that slot
    // by a zero coefficient (NaN * 0 = NaN)
    for (int e = tid; e < (rows_l + nrows) * rowp; e += NT) smem[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < nrows * rowp; e += NT) Ul[e] = Ug[e];
    if constexpr (AMER) {
        const double *__restrict__ P0g = a.U0 + (size_t)inst * a.L.inst_stride;
        for (int e = tid; e < nrows * rowp; e += NT) {
            U0l[e] = P0g[e];
            LAMl[e] = 0.0;  // lambda_bar <- 0, device_solver.hpp:310-313
        }
    }
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B;
        for (int e = tid; e < 4 * 64 * B; e += NT) coef[e] = sc[e];
        const double *__restrict__ rg = a.rowc + (size_t)inst * nrows * HADI_RC;
        for (int e = tid; e < nrows * HADI_RCL; e += NT) rtab[e] = rg[(e / HADI_RCL) * HADI_RC + e % HADI_RCL];
        const double *__restrict__ pg = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
        for (int e = tid; e < nrows * HADI_PBW; e += NT) ptab[e] = pg[e];
    }
    HadiRowCtx c;
    c.lane = lane; c.half = 0; c.wrow = wave; c.rowp = rowp;
    c.dt = ip.dt; c.thdt = ip.thdt; c.qd = ip.q; c.half_rd = ip.half_rd;
    c.hr0 = ip.hr0; c.inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    c.Yi = Yl; c.Li = AMER ? LAMl : nullptr;
    c.rowc = rtab; c.j0 = 0;
    c.b2r = a.b2row + (size_t)inst * rowp;
    c.coef = coef; c.xch = nullptr; c.R1i = nullptr; c.C2i = nullptr; c.err = a.err; c.debug = 0;
    c.payrow = nullptr; c.inv_dt = 0.0; c.m1_lane = -1; c.m1_r = -1;
    {
        const int ifirst = 1 + B * lane;
        c.posL = hadi_pos(B, G, ifirst - 1);
        c.posR = (ifirst + B <= 64 * B) ? hadi_pos(B, G, ifirst + B) : c0slot + 1;
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    unsigned long long stamp_store_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c.stamp_acc_ = stamp_store_;
#endif
    __syncthreads();

    const int N = ip.N < sm.Nmax ? ip.N : sm.Nmax;
    HADI_STAMP_DECL(c.stamp_acc_)
    const double *__restrict__ vs = sm.vec_s ? sm.vec_s + (size_t)inst * (m1 + 1) : nullptr;
    for (int n = 1; n <= N; n++) {
        // ---- discrete dividend at the start of the step (device_solver.hpp:448-504) ---------------
        const int dv = sm.div_flag ? sm.div_flag[(size_t)inst * sm.flag_stride + n - 1] : -1;
        if (dv >= 0) {
            for (int e = tid; e < nrows * rowp; e += NT) Yl[e] = Ul[e];  // U_temp
            __syncthreads();
            const double amount = sm.div_amounts[dv], pct = sm.div_pcts[dv];
            for (int e = tid; e < nrows * (m1 + 1); e += NT) {
                const int j = e / (m1 + 1), i = e - j * (m1 + 1);
                const double *src = Yl + (size_t)j * rowp;
                const double new_s = vs[i] * (1.0 - pct) - amount;
                double out = ip.put ? src[c0slot] : 0.0;  // ex-dividend spot <= 0: a call is worth 0 (device_solver.hpp:499-503), a put its s = 0 value
                if (new_s > 0) {
                    int lo = 0, hi = m1 + 1;  // first k with s[k] > new_s (0 if none)
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (vs[mid] > new_s) hi = mid;
                        else lo = mid + 1;
                    }
                    const int idx = (lo <= m1) ? lo : 0;
                    if (idx > 0) {
                        const double s_low = vs[idx - 1], s_high = vs[idx];
                        const double weight = (new_s - s_low) / (s_high - s_low);
                        out = (1.0 - weight) * src[hadi_pos(B, G, idx - 1)] + weight * src[hadi_pos(B, G, idx)];
                    } else {
                        out = src[c0slot];
                    }
                }
                Ul[(size_t)j * rowp + hadi_pos(B, G, i)] = out;
            }
            __syncthreads();
        }
        // ---- row pass: 4 rows at a time, straight out of LDS -----------------------------------------
        c.e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));
        c.e_n = exp(ip.bc_rate * ip.dt * n);
        HADI_STAMP(8);
        for (int J = 0; J < nrows; J += W) {
            const int j = J + wave;
            if (j < nrows) {
                const double *r0 = Ul + (size_t)j * rowp;
                if (j == nrows - 1)
                    hadi_row_step<B, G, AMER, true>(c, true, j, r0 - 2 * rowp, r0 - rowp, r0, r0 + rowp, r0 + 2 * rowp);
                else
                    hadi_row_step<B, G, AMER, false>(c, true, j, r0 - 2 * rowp, r0 - rowp, r0, r0 + rowp, r0 + 2 * rowp);
            }
        }
        HADI_STAMP(10);  // row pass
        __syncthreads();
        HADI_STAMP(9);  // barrier
        // ---- column pass: one thread per storage column, sequential pentadiagonal sweeps in LDS --------
        for (int col = tid; col < rowp; col += NT) {
            // rounds of eight rows: the independent LDS reads first, then the dependent recurrence (as in hadi_small_seq_kernel)
            double ym1 = 0.0, ym2 = 0.0;
            int k = 0;
            for (; k + 8 <= nrows; k += 8) {
                double yv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) yv[q] = Yl[(size_t)(k + q) * rowp + col];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double *t = ptab + (size_t)(k + q) * HADI_PBW;
                    const double yk = (yv[q] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                    Yl[(size_t)(k + q) * rowp + col] = yk;
                    ym2 = ym1;
                    ym1 = yk;
                }
            }
            for (; k < nrows; k++) {
                const double *t = ptab + (size_t)k * HADI_PBW;
                const double yk = (Yl[(size_t)k * rowp + col] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                Yl[(size_t)k * rowp + col] = yk;
                ym2 = ym1;
                ym1 = yk;
            }
            double xp1 = 0.0, xp2 = 0.0;
            k = nrows - 1;
            if constexpr (!AMER) {
                for (; k >= 7; k -= 8) {
                    double yv[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) yv[q] = Yl[(size_t)(k - q) * rowp + col];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double *t = ptab + (size_t)(k - q) * HADI_PBW;
                        const double xk = yv[q] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                        xp2 = xp1;
                        xp1 = xk;
                        Ul[(size_t)(k - q) * rowp + col] = xk;
                    }
                }
            }
            for (; k >= 0; k--) {
                const double *t = ptab + (size_t)k * HADI_PBW;
                const double xk = Yl[(size_t)k * rowp + col] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                xp2 = xp1;
                xp1 = xk;
                if constexpr (AMER) {  // Ikonen-Toivanen projection, device_solver.hpp:358-372
                    const size_t off = (size_t)k * rowp + col;
                    const double lamv = LAMl[off], pay = U0l[off];
                    Ul[off] = fmax(xk - ip.dt * lamv, pay);
                    double ln = fmax(0.0, lamv + (pay - xk) / ip.dt);
                    if (col == a.pos_m1) ln = 0.0;
                    LAMl[off] = ln;
                } else {
                    Ul[(size_t)k * rowp + col] = xk;
                }
            }
        }
        __syncthreads();
    }
#if defined(HADI_STAMPS) && !defined(HADI_EMU)
    if (lane == 0)
        for (int k = 0; k < 12; k++) atomicAdd(&g_hadi_stamps[k], stamp_store_[k]);
#endif
    for (int e = tid; e < nrows * rowp; e += NT) Ug[e] = Ul[e];
    if constexpr (AMER) {
        double *__restrict__ Lg = a.LAM + (size_t)inst * a.L.inst_stride;
        for (int e = tid; e < nrows * rowp; e += NT) Lg[e] = LAMl[e];
    }
}

// ------------------------------------------------------------------------------------------------
// Small grids, European / dividend sweeps: ONE wavefront per instance, lines solved SEQUENTIALLY, one line per lane.
// The kernel above runs the big-grid row step on 51-node rows: a whole wavefront and six cyclic-reduction levels (54
// ds_bpermute, ~400 instructions) per row -- at one node per lane almost all of it is overhead, and 26 rows x 8
// wavefronts cost ~10 us per time step.  Here the roles are turned round, as in the reference's own team kernels
// (hes_a1_kernels.hpp:139-161, one thread per v-row; hes_a2_shuffled_kernels.hpp:243-299, one thread per s-column):
//   row pass     lane j <-> v-row j (nrows <= 33 lanes busy) walks i = 1 .. m1: explicit operators from a sliding window of
//                three columns (five new LDS values per node), Y0, forward Thomas with the pivot recomputed on the fly; the
//                back substitution walks i = m1 .. 1.  No cross-lane traffic at all.
//   column pass  lane i <-> s-column i: pentadiagonal forward / backward sweep with the precomputed factors.
// State in LDS in NATURAL order, pitch odd (conflict-free both ways): U (two zero halo rows above and below) and Y.  The
// forward sweep needs three values per node for the way back (the normalised right-hand side, the multiplier c', and the
// explicit A2 correction of the output) but only two arrays exist: the output is rewritten as
//   Y_i = x_i + corr_i = (ys_i + corr_i + c'_i corr_{i+1}) - c'_i Y_{i+1} = g_i - c'_i Y_{i+1},
// g_i goes to Y, and c'_i goes to column i-1 of U's own row -- every lane is at the same i (one wavefront), so that column
// has been consumed by all of them.  The column pass rebuilds U completely.  ~45 instructions per node against ~10 x that.
struct HadiSmallSeqLayout {
    int pitch;     // doubles per row in LDS: odd, >= m1 + 3 (columns m1 + 1, m1 + 2 stay zero: the s-neighbour of the last
                   // node and the column the one-ahead fetch touches behind it)
    int off_y, off_coef, off_b2, off_zero, off_dummy, off_ptab, total;  // offsets in doubles: U starts at 0 (nrows rows)
};
HADI_HD inline HadiSmallSeqLayout hadi_small_seq_layout(int m1, int nrows) {
    HadiSmallSeqLayout l;
    l.pitch = (m1 + 3) | 1;
    l.off_y = nrows * l.pitch;
    l.off_coef = l.off_y + nrows * l.pitch;
    l.off_coef = (l.off_coef + 1) & ~1;  // 16-byte aligned quads
    l.off_b2 = l.off_coef + 4 * (m1 + 2);
    l.off_zero = l.off_b2 + (m1 + 2);       // a row of zeros: the "b2 row" of every v-row but the last
    l.off_dummy = l.off_zero + (m1 + 2);    // where the idle lanes (>= nrows) put their results
    l.off_ptab = l.off_dummy + l.pitch;
    l.total = l.off_ptab + nrows * 5;
    return l;
}

template <int B>
__global__ void __launch_bounds__(64) hadi_small_seq_kernel(HadiSweepArgs a, HadiSmallArgs sm) {
    HADI_DYN_SMEM(double, smem);
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= a.n_inst) return;
    const int inst = sm.order ? sm.order[blockIdx.x] : (int)blockIdx.x;
    const HadiInstPar ip = a.ipar[inst];
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1;
    const HadiSmallSeqLayout Ls = hadi_small_seq_layout(m1, nrows);
    const int PL = Ls.pitch;
    double *Ul = smem;  // row 0 of U
    double *Yl = smem + Ls.off_y;
    double *coefl = smem + Ls.off_coef;  // [i][4]: Bm, Bp, Dm, Dp of node i
    double *b2l = smem + Ls.off_b2;
    double *ptab = smem + Ls.off_ptab;   // [k][5]: L, L2, Q, C, C2
    double *__restrict__ Ug = a.U + (size_t)inst * a.L.inst_stride;

    for (int e = lane; e < Ls.total; e += 64) smem[e] = 0.0;
    __syncthreads();
    for (int e = lane; e < nrows * (m1 + 1); e += 64) {
        const int j = e / (m1 + 1), i = e - j * (m1 + 1);
        Ul[j * PL + i] = Ug[(size_t)j * rowp + hadi_pos(B, 1, i)];
    }
    {
        const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * 64 * B;
        for (int e = lane; e < 4 * (m1 + 1); e += 64) {
            const int i = e >> 2, k = e & 3;
            coefl[e] = (i >= 1) ? sc[k * 64 * B + hadi_pos(B, 1, i)] : 0.0;
        }
        const double *__restrict__ b2g = a.b2row + (size_t)inst * rowp;
        for (int i = lane; i <= m1; i += 64) b2l[i] = b2g[hadi_pos(B, 1, i)];
        const double *__restrict__ pg = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
        for (int e = lane; e < nrows * 5; e += 64) ptab[e] = pg[(e / 5) * HADI_PBW + e % 5];
    }
    // this lane's v-row: its table entry stays in registers for the whole time loop
    const int j = lane;
    const bool act = j < nrows;
    const bool last = (j == nrows - 1);
    double v = 0.0, wm = 0.0, wz = 0.0, wp = 0.0, a2l2 = 0.0, a2l1 = 0.0, a2m = 0.0, a2u1 = 0.0, a2u2 = 0.0, b1val = 0.0;
    int b1col = -1;
    bool b1_at0 = false;
    if (act) {
        const double *__restrict__ rc = a.rowc + ((size_t)inst * nrows + j) * HADI_RC;
        v = rc[RC_V]; wm = rc[RC_WM]; wz = rc[RC_WZ]; wp = rc[RC_WP];
        a2l2 = rc[RC_L2]; a2l1 = rc[RC_L1]; a2m = rc[RC_M]; a2u1 = rc[RC_U1]; a2u2 = rc[RC_U2];
        b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;  // (two entries on one v-row: m2 > m1 only)
        b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    }
    const double dt = ip.dt, thdt = ip.thdt, qd = ip.q, half_rd = ip.half_rd;
    const double inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    const double *urow = Ul + (act ? j : 0) * PL;  // (idle lanes walk row 0 and store nothing)
    // The v-neighbours j-2 .. j+2, clamped to the grid instead of zero halo rows (1.7 KB that cost the sixth instance per CU):
    // a clamped row only ever meets a zero weight -- the first / last rows of A0 and A2 have no entries beyond the grid.
    // The v-neighbours j-2 .. j+2 of a column are the SAME column in the neighbouring LANES' rows: one LDS read of the own row
    // and four wave shifts (DPP) instead of five LDS reads per node -- at six wavefronts per CU the one LDS pipe, not the
    // SIMDs, is what this kernel fills (round 3).  Beyond the grid the shifts deliver 0 or an idle lane's (finite) value of
    // row 0; either only ever meets a zero weight -- the first / last rows of A0 and A2 have no entries beyond the grid.
    auto col5 = [&](const double own, double &m2v, double &m1v, double &p1v, double &p2v) {
        m1v = hadi_lane_prev(own); m2v = hadi_lane_prev(m1v);
        p1v = hadi_lane_next(own); p2v = hadi_lane_next(p1v);
    };
    double *yrow = act ? Yl + j * PL : smem + Ls.off_dummy;  // (idle lanes store into a dummy row)
    double *crow = act ? Ul + j * PL : smem + Ls.off_dummy;  // column i - 1 of this row receives c'_i
    const double *b2p = last ? b2l : smem + Ls.off_zero;      // b2 lives on the last v-row only
    __syncthreads();

    const int N = ip.N < sm.Nmax ? ip.N : sm.Nmax;
    const double *__restrict__ vs = sm.vec_s ? sm.vec_s + (size_t)inst * (m1 + 1) : nullptr;
    for (int n = 1; n <= N; n++) {
        // ---- discrete dividend at the start of the step (device_solver.hpp:448-504) ---------------
        const int dv = sm.div_flag ? sm.div_flag[(size_t)inst * sm.flag_stride + n - 1] : -1;
        if (dv >= 0) {
            for (int e = lane; e < nrows * PL; e += 64) Yl[e] = Ul[e];  // U_temp
            __syncthreads();
            const double amount = sm.div_amounts[dv], pct = sm.div_pcts[dv];
            for (int e = lane; e < nrows * (m1 + 1); e += 64) {
                const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
                const double *src = Yl + jj * PL;
                const double new_s = vs[i] * (1.0 - pct) - amount;
                double out = ip.put ? src[0] : 0.0;  // ex-dividend spot <= 0: a call is worth 0 (device_solver.hpp:499-503), a put its s = 0 value
                if (new_s > 0) {
                    int lo = 0, hi = m1 + 1;  // first k with s[k] > new_s (0 if none)
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (vs[mid] > new_s) hi = mid;
                        else lo = mid + 1;
                    }
                    const int idx = (lo <= m1) ? lo : 0;
                    if (idx > 0) {
                        const double s_low = vs[idx - 1], s_high = vs[idx];
                        const double weight = (new_s - s_low) / (s_high - s_low);
                        out = (1.0 - weight) * src[idx - 1] + weight * src[idx];
                    } else {
                        out = src[0];
                    }
                }
                Ul[jj * PL + i] = out;
            }
            __syncthreads();
        }
        const double e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
        const double e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
        const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
        const double b1l = b1val * cb1;
        // ---- row pass: lane <-> v-row, i = 1 .. m1 (same formulas as hadi_row_step) --------------------------------
        // Only the lanes that own a v-row run the sweeps: the LDS moves 16 or 8 bytes per ACTIVE lane, and with nrows of 64 lanes
        // busy that is less than half of what the idle lanes' dummy walk used to drag through the CU's one LDS pipe.  (The wave
        // shifts deliver 0 from a switched-off lane: the row beyond the last one only ever meets a zero weight.  The emulator's
        // lane threads all have to take part in its collective shuffles, so there every lane still runs.)
#if defined(HADI_EMU)
        const bool rowrun = true;
#else
        const bool rowrun = act;
#endif
        if (rowrun) {
        // column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act)
        const double c00 = urow[0];
        double c0m2, c0m1, c0p1, c0p2;
        col5(c00, c0m2, c0m1, c0p1, c0p2);
        // first interior column, raw: rows j-2 .. j+2
        double r_0 = urow[1], r_m2, r_m1, r_p1, r_p2;
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        double yout_c0, x0;
        {
            const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
            const double b1c0 = b1_at0 ? b1val : 0.0;
            const double b2c0 = b2p[0];
            const double a1c0 = -ip.hr0 * c00;  // A1 row 0: empty for the call (hr0 = 0), the reaction term for the put
            double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1);
            y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
            const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
            x0 = y0c0 * inv0;  // A1 row 0 is decoupled: the identity for the call (hes_a1_kernels.hpp:56-61)
            yout_c0 = x0 + c2c0;
        }
        hadi_wave_rendezvous();  // (emulator: everyone has read column 0 and 1 before c' overwrites column 0)
        double u_prev = c00, u_cur = r_0;
        double t_prev = wm * c0m1 + wz * c00 + wp * c0p1;
        double t_cur = wm * r_m1 + wz * r_0 + wp * r_p1;
        double a2u_cur = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
        double b2c = b2p[1];
        double corr_cur = thdt * (b2c * e_n - (a2u_cur + b2c * e_nm1));
        // raw values of column 2 (column m1 + 1 is the zero spare)
        r_0 = urow[2];
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        // x_0 is known and moves to the right-hand side of node 1: with ys_0 = x_0 and c'_0 = 0 the general step does exactly
        // that (pivot im - il 0, right-hand side y - il x_0)
        double cp_prev = 0.0, ys_prev = x0;
        // One node of the sweep.  On entry r_* hold the raw column i + 1; `cB`, `cD` are node i's coefficients, `b2n` the b2
        // entry of node i + 1.  The caller refills r_* with column i + 2 afterwards.
        auto node = [&](int i, const double2 cB, const double2 cD, const double b2n) {
            const double u_next = r_0;
            const double t_next = wm * r_m1 + wz * r_0 + wp * r_p1;
            const double a2u_next = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
            const double lo = fma(v, cD.x, qd * cB.x);
            const double up = fma(v, cD.y, qd * cB.y);
            const double mn = -((lo + up) + half_rd);
            const double A1U = lo * u_prev + mn * u_cur + up * u_next;
            const double A0U = cB.x * t_prev - (cB.x + cB.y) * t_cur + cB.y * t_next;
            // Y0 = U + dt (A0U + A1U + A2U + b e_{n-1}) + theta dt (b1 e_n - (A1U + b1 e_{n-1})), device_solver.hpp:236-250
            double S = A0U + A1U + a2u_cur;
            S += b2c * e_nm1;
            double y = fma(dt, S, u_cur);
            y = fma(-thdt, A1U, y);
            y += (i == b1col) ? b1l : 0.0;
            const double il = -thdt * lo;
            const double im = 1.0 - thdt * mn;
            const double iu = -thdt * up;
            const double inv = hadi_rcp(fma(-il, cp_prev, im));
            const double cp = iu * inv;
            const double ys = fma(-il, ys_prev, y) * inv;
            const double corr_next = thdt * (b2n * e_n - (a2u_next + b2n * e_nm1));
            yrow[i] = ys + corr_cur + cp * corr_next;  // g_i  (c'_{m1} = 0: the row ends there)
            crow[i - 1] = cp;
            u_prev = u_cur; u_cur = u_next;
            t_prev = t_cur; t_cur = t_next;
            a2u_cur = a2u_next; corr_cur = corr_next; b2c = b2n;
            cp_prev = cp; ys_prev = ys;
        };
        // Rounds of FOUR nodes: the own-row values of the columns i + 2 .. i + 5, the four nodes' coefficients and b2 entries
        // are all read first -- 16 independent LDS reads behind ONE wait -- then the four dependent steps run without touching
        // the LDS return path.  (Round 2 read five rows' values per node and waited for each node's coefficient and b2 reads:
        // ~900 LDS instructions per time step, and at the six wavefronts per CU that the 26 KB of an instance allow, the CU's
        // one LDS pipe was the busiest unit.)  Column indices reach i + 5 <= m1 + 2: the two zero spare columns of the pitch.
        int i = 1;
        for (; i + 3 <= m1; i += 4) {
            double Rm2[4], Rm1[4], R0[4], Rp1[4], Rp2[4], b2q[4];
            double2 cBq[4], cDq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                R0[q] = urow[i + 2 + q];
                cBq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q));
                cDq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q) + 2);
                b2q[q] = b2p[i + 1 + q];  // (entry m1 + 1 is zero)
            }
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");  // (the reads stay in front of the four steps' stores)
#endif
            hadi_wave_rendezvous();  // (emulator: every lane has read its columns before anybody's c' lands in them)
#pragma unroll
            for (int q = 0; q < 4; q++) col5(R0[q], Rm2[q], Rm1[q], Rp1[q], Rp2[q]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                node(i + q, cBq[q], cDq[q], b2q[q]);
                r_m2 = Rm2[q]; r_m1 = Rm1[q]; r_0 = R0[q]; r_p1 = Rp1[q]; r_p2 = Rp2[q];
                hadi_wave_rendezvous();  // (emulator: the lanes walk in lock step on the GPU)
            }
        }
        for (; i <= m1; i++) {  // the last m1 mod 4 nodes, one at a time
            const double n_0 = urow[i + 2];  // (column <= m1 + 2: a zero column)
            double n_m2, n_m1, n_p1, n_p2;
            col5(n_0, n_m2, n_m1, n_p1, n_p2);
            const double2 cB = *reinterpret_cast<const double2 *>(coefl + 4 * i);      // Bm, Bp
            const double2 cD = *reinterpret_cast<const double2 *>(coefl + 4 * i + 2);  // Dm, Dp
            const double b2n = b2p[i + 1];
            hadi_wave_rendezvous();
            node(i, cB, cD, b2n);
            r_m2 = n_m2; r_m1 = n_m1; r_0 = n_0; r_p1 = n_p1; r_p2 = n_p2;
            hadi_wave_rendezvous();  // (emulator: the lanes walk in lock step on the GPU)
        }
        // back substitution on the output itself: Y_i = g_i - c'_i Y_{i+1}
        {   // (idle lanes walk their dummy row)
            double Yn = yrow[m1];
            int i = m1 - 1;
            for (; i >= 8; i -= 8) {  // eight nodes per round: 16 independent LDS reads, then the dependent FMAs
                double g[8], cq[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { g[q] = yrow[i - q]; cq[q] = crow[i - q - 1]; }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    Yn = fma(-cq[q], Yn, g[q]);
                    yrow[i - q] = Yn;
                }
            }
            for (; i >= 1; i--) {
                Yn = fma(-crow[i - 1], Yn, yrow[i]);
                yrow[i] = Yn;
            }
            yrow[0] = yout_c0;
        }
        }  // (rowrun)
        __syncthreads();
        // ---- column pass: lane <-> s-column, sequential pentadiagonal sweeps (hes_a2_shuffled_kernels.hpp:243-299) ----
        // (measured and left out, 50x25 x3000: fetching a node's coefficients one iteration ahead 2.64 -> 2.72 ms; the column
        // held in 33 registers with all loads up front 2.64 -> 2.88 ms)
        for (int col = lane; col <= m1; col += 64) {
            // eight rows per round: the independent LDS reads first, then the dependent recurrence
            double ym1 = 0.0, ym2 = 0.0;
            int k = 0;
            for (; k + 8 <= nrows; k += 8) {
                double yv[8], tL[8], tL2[8], tQ[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {  // (the round's table entries with its column values: 32 reads, one wait)
                    const double *t = ptab + (k + q) * 5;
                    yv[q] = Yl[(k + q) * PL + col];
                    tL[q] = t[PB_L]; tL2[q] = t[PB_L2]; tQ[q] = t[PB_Q];
                }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double yk = (yv[q] - tL[q] * ym1 - tL2[q] * ym2) * tQ[q];
                    Yl[(k + q) * PL + col] = yk;
                    ym2 = ym1;
                    ym1 = yk;
                }
            }
            for (; k < nrows; k++) {
                const double *t = ptab + k * 5;
                const double yk = (Yl[k * PL + col] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                Yl[k * PL + col] = yk;
                ym2 = ym1;
                ym1 = yk;
            }
            double xp1 = 0.0, xp2 = 0.0;
            k = nrows - 1;
            for (; k >= 7; k -= 8) {
                double yv[8], tC[8], tC2[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double *t = ptab + (k - q) * 5;
                    yv[q] = Yl[(k - q) * PL + col];
                    tC[q] = t[PB_C]; tC2[q] = t[PB_C2];
                }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const double xk = yv[q] - tC[q] * xp1 - tC2[q] * xp2;
                    xp2 = xp1;
                    xp1 = xk;
                    Ul[(k - q) * PL + col] = xk;
                }
            }
            for (; k >= 0; k--) {
                const double *t = ptab + k * 5;
                const double xk = Yl[k * PL + col] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                xp2 = xp1;
                xp1 = xk;
                Ul[k * PL + col] = xk;
            }
        }
        __syncthreads();
    }
    for (int e = lane; e < nrows * (m1 + 1); e += 64) {
        const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
        Ug[(size_t)jj * rowp + hadi_pos(B, 1, i)] = Ul[jj * PL + i];
    }
}

// ------------------------------------------------------------------------------------------------
// hadi_small_seq_kernel with TWO instances per wavefront (round 3).  The row sweep above keeps nrows of the 64 lanes busy
// -- 26 for the reference's 50x25 grid -- and is nine tenths of the kernel's instructions: here lanes 0..31 walk the v-rows
// of one instance and lanes 32..63 those of a second one through the SAME instruction stream (per-lane LDS base pointers
// and instance scalars; the wave shifts that fetch the v-neighbours meet zero weights across the boundary between the two
// instances exactly as they do beyond a grid's own first and last row).  The column sweeps (lane <-> s-column) run once
// per instance.  LDS: two instances' arrays per wavefront (52 KB for 50x25: three wavefronts = six instances per CU, as
// before), but a wavefront now retires two instances' time steps in little more than the time of one.  Instances with
// different numbers of time steps (multi-maturity batches, dispatched longest first) simply stop at their own N.
// Needs nrows <= 32.  Same arithmetic, operation by operation, as hadi_small_seq_kernel: the results are bit-identical.
template <int B>
__global__ void __launch_bounds__(64) hadi_small_seq2_kernel(HadiSweepArgs a, HadiSmallArgs sm) {
    HADI_DYN_SMEM(double, smem);
    const int lane = threadIdx.x;
    const int half = lane >> 5, jl = lane & 31;
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1;
    const HadiSmallSeqLayout Ls = hadi_small_seq_layout(m1, nrows);
    const int PL = Ls.pitch;
    // the two instances of this wavefront (the second slot of the last block may be empty)
    const int slot0 = 2 * (int)blockIdx.x, slot1 = slot0 + 1;
    if (slot0 >= a.n_inst) return;
    const bool has1 = slot1 < a.n_inst;
    const int inst0 = sm.order ? sm.order[slot0] : slot0;
    const int inst1 = has1 ? (sm.order ? sm.order[slot1] : slot1) : inst0;
    const int inst = half ? inst1 : inst0;             // this lane's instance
    const HadiInstPar ip = a.ipar[inst];               // (per lane: two different structs in the wavefront)
    const HadiInstPar ip0 = a.ipar[inst0], ip1 = a.ipar[inst1];
    const int N0 = ip0.N < sm.Nmax ? ip0.N : sm.Nmax;
    const int N1 = has1 ? (ip1.N < sm.Nmax ? ip1.N : sm.Nmax) : 0;
    const int Nw = N0 > N1 ? N0 : N1;                  // (wave-uniform)
    const int Nl = half ? N1 : N0;                     // this lane's number of time steps
    double *const base0 = smem, *const base1 = smem + Ls.total;
    double *const bl = half ? base1 : base0;           // this lane's instance in LDS
    double *Ul = bl;
    double *Yl = bl + Ls.off_y;
    double *coefl = bl + Ls.off_coef;                  // [i][4]: Bm, Bp, Dm, Dp of node i
    double *b2l = bl + Ls.off_b2;

    for (int e = lane; e < 2 * Ls.total; e += 64) smem[e] = 0.0;
    __syncthreads();
    for (int h = 0; h < (has1 ? 2 : 1); h++) {         // (uniform loops: all 64 lanes copy one instance, then the other)
        const int ih = h ? inst1 : inst0;
        double *bh = h ? base1 : base0;
        const double *__restrict__ Ug = a.U + (size_t)ih * a.L.inst_stride;
        for (int e = lane; e < nrows * (m1 + 1); e += 64) {
            const int j = e / (m1 + 1), i = e - j * (m1 + 1);
            bh[j * PL + i] = Ug[(size_t)j * rowp + hadi_pos(B, 1, i)];
        }
        const double *__restrict__ sc = a.scoef + (size_t)ih * 4 * 64 * B;
        for (int e = lane; e < 4 * (m1 + 1); e += 64) {
            const int i = e >> 2, k = e & 3;
            bh[Ls.off_coef + e] = (i >= 1) ? sc[k * 64 * B + hadi_pos(B, 1, i)] : 0.0;
        }
        const double *__restrict__ b2g = a.b2row + (size_t)ih * rowp;
        for (int i = lane; i <= m1; i += 64) bh[Ls.off_b2 + i] = b2g[hadi_pos(B, 1, i)];
        const double *__restrict__ pg = a.pb + (size_t)ih * a.L.nrows_pad * HADI_PBW;
        for (int e = lane; e < nrows * 5; e += 64) bh[Ls.off_ptab + e] = pg[(e / 5) * HADI_PBW + e % 5];
    }
    // this lane's v-row of its instance: the table entry stays in registers for the whole time loop
    const int j = jl;
    const bool act = j < nrows && (half == 0 || has1);
    const bool last = (j == nrows - 1);
    double v = 0.0, wm = 0.0, wz = 0.0, wp = 0.0, a2l2 = 0.0, a2l1 = 0.0, a2m = 0.0, a2u1 = 0.0, a2u2 = 0.0, b1val = 0.0;
    int b1col = -1;
    bool b1_at0 = false;
    if (act) {
        const double *__restrict__ rc = a.rowc + ((size_t)inst * nrows + j) * HADI_RC;
        v = rc[RC_V]; wm = rc[RC_WM]; wz = rc[RC_WZ]; wp = rc[RC_WP];
        a2l2 = rc[RC_L2]; a2l1 = rc[RC_L1]; a2m = rc[RC_M]; a2u1 = rc[RC_U1]; a2u2 = rc[RC_U2];
        b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;
        b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    }
    const double dt = ip.dt, thdt = ip.thdt, qd = ip.q, half_rd = ip.half_rd;
    const double inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    const double *urow = Ul + (act ? j : 0) * PL;  // (idle lanes walk row 0 and store nothing)
    auto col5 = [&](const double own, double &m2v, double &m1v, double &p1v, double &p2v) {
        m1v = hadi_lane_prev(own); m2v = hadi_lane_prev(m1v);
        p1v = hadi_lane_next(own); p2v = hadi_lane_next(p1v);
    };
    double *const yrow_real = act ? Yl + j * PL : bl + Ls.off_dummy;
    double *const crow_real = act ? Ul + j * PL : bl + Ls.off_dummy;
    const double *b2p = last ? b2l : bl + Ls.off_zero;      // b2 lives on the last v-row only
    __syncthreads();

    for (int n = 1; n <= Nw; n++) {
        // ---- discrete dividends at the start of the step, instance by instance (device_solver.hpp:448-504) ----
        for (int h = 0; h < (has1 ? 2 : 1); h++) {
            const int ih = h ? inst1 : inst0, Nh = h ? N1 : N0;
            const int dv = (sm.div_flag && n <= Nh) ? sm.div_flag[(size_t)ih * sm.flag_stride + n - 1] : -1;
            if (dv >= 0) {  // (wave-uniform)
                double *Uh = h ? base1 : base0, *Yh = Uh + Ls.off_y;
                const double *__restrict__ vs = sm.vec_s + (size_t)ih * (m1 + 1);
                const int put_h = h ? ip1.put : ip0.put;
                for (int e = lane; e < nrows * PL; e += 64) Yh[e] = Uh[e];  // U_temp
                __syncthreads();
                const double amount = sm.div_amounts[dv], pct = sm.div_pcts[dv];
                for (int e = lane; e < nrows * (m1 + 1); e += 64) {
                    const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
                    const double *src = Yh + jj * PL;
                    const double new_s = vs[i] * (1.0 - pct) - amount;
                    double out = put_h ? src[0] : 0.0;
                    if (new_s > 0) {
                        int lo = 0, hi = m1 + 1;  // first k with s[k] > new_s (0 if none)
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (vs[mid] > new_s) hi = mid;
                            else lo = mid + 1;
                        }
                        const int idx = (lo <= m1) ? lo : 0;
                        if (idx > 0) {
                            const double s_low = vs[idx - 1], s_high = vs[idx];
                            const double weight = (new_s - s_low) / (s_high - s_low);
                            out = (1.0 - weight) * src[idx - 1] + weight * src[idx];
                        } else {
                            out = src[0];
                        }
                    }
                    Uh[jj * PL + i] = out;
                }
                __syncthreads();
            }
        }
        const double e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
        const double e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
        const double cb1 = dt * e_nm1 + thdt * (e_n - e_nm1);
        const double b1l = b1val * cb1;
        // a lane whose instance has finished its own N steps keeps walking (the wave shifts are collective) but stores into the
        // dummy row; on the GPU it is switched off altogether
        const bool live = act && n <= Nl;
        double *const yrow = live ? yrow_real : bl + Ls.off_dummy;
        double *const crow = live ? crow_real : bl + Ls.off_dummy;
#if defined(HADI_EMU)
        const bool rowrun = true;
#else
        const bool rowrun = live;
#endif
        if (rowrun) {
        // ---- row pass: lane <-> v-row of its instance (hadi_small_seq_kernel, operation by operation) ----
        const double c00 = urow[0];
        double c0m2, c0m1, c0p1, c0p2;
        col5(c00, c0m2, c0m1, c0p1, c0p2);
        double r_0 = urow[1], r_m2, r_m1, r_p1, r_p2;
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        double yout_c0, x0;
        {
            const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
            const double b1c0 = b1_at0 ? b1val : 0.0;
            const double b2c0 = b2p[0];
            const double a1c0 = -ip.hr0 * c00;
            double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1);
            y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
            const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
            x0 = y0c0 * inv0;
            yout_c0 = x0 + c2c0;
        }
        hadi_wave_rendezvous();
        double u_prev = c00, u_cur = r_0;
        double t_prev = wm * c0m1 + wz * c00 + wp * c0p1;
        double t_cur = wm * r_m1 + wz * r_0 + wp * r_p1;
        double a2u_cur = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
        double b2c = b2p[1];
        double corr_cur = thdt * (b2c * e_n - (a2u_cur + b2c * e_nm1));
        r_0 = urow[2];
        col5(r_0, r_m2, r_m1, r_p1, r_p2);
        double cp_prev = 0.0, ys_prev = x0;
        auto node = [&](int i, const double2 cB, const double2 cD, const double b2n) {
            const double u_next = r_0;
            const double t_next = wm * r_m1 + wz * r_0 + wp * r_p1;
            const double a2u_next = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
            const double lo = fma(v, cD.x, qd * cB.x);
            const double up = fma(v, cD.y, qd * cB.y);
            const double mn = -((lo + up) + half_rd);
            const double A1U = lo * u_prev + mn * u_cur + up * u_next;
            const double A0U = cB.x * t_prev - (cB.x + cB.y) * t_cur + cB.y * t_next;
            double S = A0U + A1U + a2u_cur;
            S += b2c * e_nm1;
            double y = fma(dt, S, u_cur);
            y = fma(-thdt, A1U, y);
            y += (i == b1col) ? b1l : 0.0;
            const double il = -thdt * lo;
            const double im = 1.0 - thdt * mn;
            const double iu = -thdt * up;
            const double inv = hadi_rcp(fma(-il, cp_prev, im));
            const double cp = iu * inv;
            const double ys = fma(-il, ys_prev, y) * inv;
            const double corr_next = thdt * (b2n * e_n - (a2u_next + b2n * e_nm1));
            yrow[i] = ys + corr_cur + cp * corr_next;
            crow[i - 1] = cp;
            u_prev = u_cur; u_cur = u_next;
            t_prev = t_cur; t_cur = t_next;
            a2u_cur = a2u_next; corr_cur = corr_next; b2c = b2n;
            cp_prev = cp; ys_prev = ys;
        };
        int i = 1;
        for (; i + 3 <= m1; i += 4) {
            double Rm2[4], Rm1[4], R0[4], Rp1[4], Rp2[4], b2q[4];
            double2 cBq[4], cDq[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                R0[q] = urow[i + 2 + q];
                cBq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q));
                cDq[q] = *reinterpret_cast<const double2 *>(coefl + 4 * (i + q) + 2);
                b2q[q] = b2p[i + 1 + q];
            }
#if !defined(HADI_EMU)
            asm volatile("" ::: "memory");
#endif
            hadi_wave_rendezvous();
#pragma unroll
            for (int q = 0; q < 4; q++) col5(R0[q], Rm2[q], Rm1[q], Rp1[q], Rp2[q]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                node(i + q, cBq[q], cDq[q], b2q[q]);
                r_m2 = Rm2[q]; r_m1 = Rm1[q]; r_0 = R0[q]; r_p1 = Rp1[q]; r_p2 = Rp2[q];
                hadi_wave_rendezvous();
            }
        }
        for (; i <= m1; i++) {
            const double n_0 = urow[i + 2];
            double n_m2, n_m1, n_p1, n_p2;
            col5(n_0, n_m2, n_m1, n_p1, n_p2);
            const double2 cB = *reinterpret_cast<const double2 *>(coefl + 4 * i);
            const double2 cD = *reinterpret_cast<const double2 *>(coefl + 4 * i + 2);
            const double b2n = b2p[i + 1];
            hadi_wave_rendezvous();
            node(i, cB, cD, b2n);
            r_m2 = n_m2; r_m1 = n_m1; r_0 = n_0; r_p1 = n_p1; r_p2 = n_p2;
            hadi_wave_rendezvous();
        }
        {   // back substitution on the output itself: Y_i = g_i - c'_i Y_{i+1}
            double Yn = yrow[m1];
            int ib = m1 - 1;
            for (; ib >= 8; ib -= 8) {
                double g[8], cq[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { g[q] = yrow[ib - q]; cq[q] = crow[ib - q - 1]; }
#if !defined(HADI_EMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    Yn = fma(-cq[q], Yn, g[q]);
                    yrow[ib - q] = Yn;
                }
            }
            for (; ib >= 1; ib--) {
                Yn = fma(-crow[ib - 1], Yn, yrow[ib]);
                yrow[ib] = Yn;
            }
            yrow[0] = yout_c0;
        }
        }  // (rowrun)
        __syncthreads();
        // ---- column pass, instance by instance: lane <-> s-column (hes_a2_shuffled_kernels.hpp:243-299) ----
        for (int h = 0; h < (has1 ? 2 : 1); h++) {
            if (n > (h ? N1 : N0)) continue;  // (wave-uniform)
            double *Uh = h ? base1 : base0, *Yh = Uh + Ls.off_y;
            const double *ptab = Uh + Ls.off_ptab;   // [k][5]: L, L2, Q, C, C2
            for (int col = lane; col <= m1; col += 64) {
                double ym1 = 0.0, ym2 = 0.0;
                int k = 0;
                for (; k + 8 <= nrows; k += 8) {
                    double yv[8], tL[8], tL2[8], tQ[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double *t = ptab + (k + q) * 5;
                        yv[q] = Yh[(k + q) * PL + col];
                        tL[q] = t[PB_L]; tL2[q] = t[PB_L2]; tQ[q] = t[PB_Q];
                    }
#if !defined(HADI_EMU)
                    asm volatile("" ::: "memory");
#endif
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double yk = (yv[q] - tL[q] * ym1 - tL2[q] * ym2) * tQ[q];
                        Yh[(k + q) * PL + col] = yk;
                        ym2 = ym1;
                        ym1 = yk;
                    }
                }
                for (; k < nrows; k++) {
                    const double *t = ptab + k * 5;
                    const double yk = (Yh[k * PL + col] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                    Yh[k * PL + col] = yk;
                    ym2 = ym1;
                    ym1 = yk;
                }
                double xp1 = 0.0, xp2 = 0.0;
                k = nrows - 1;
                for (; k >= 7; k -= 8) {
                    double yv[8], tC[8], tC2[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double *t = ptab + (k - q) * 5;
                        yv[q] = Yh[(k - q) * PL + col];
                        tC[q] = t[PB_C]; tC2[q] = t[PB_C2];
                    }
#if !defined(HADI_EMU)
                    asm volatile("" ::: "memory");
#endif
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const double xk = yv[q] - tC[q] * xp1 - tC2[q] * xp2;
                        xp2 = xp1;
                        xp1 = xk;
                        Uh[(k - q) * PL + col] = xk;
                    }
                }
                for (; k >= 0; k--) {
                    const double *t = ptab + k * 5;
                    const double xk = Yh[k * PL + col] - t[PB_C] * xp1 - t[PB_C2] * xp2;
                    xp2 = xp1;
                    xp1 = xk;
                    Uh[k * PL + col] = xk;
                }
            }
        }
        __syncthreads();
    }
    for (int h = 0; h < (has1 ? 2 : 1); h++) {
        const int ih = h ? inst1 : inst0;
        const double *bh = h ? base1 : base0;
        double *__restrict__ Ug = a.U + (size_t)ih * a.L.inst_stride;
        for (int e = lane; e < nrows * (m1 + 1); e += 64) {
            const int jj = e / (m1 + 1), i = e - jj * (m1 + 1);
            Ug[(size_t)jj * rowp + hadi_pos(B, 1, i)] = bh[jj * PL + i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Grids beyond the streaming kernels' shapes -- the reference bounds a grid by its total size only
// (src/perfomance_test.cpp:62).  Two sequential passes in the reference's own mapping (hes_a1_kernels.hpp:139-161: one
// thread per v-row; hes_a2_shuffled_kernels.hpp:243-299: one thread per s-column), correct for ANY shape, far from the
// roofline:
//   hadi_pass_a_seq  (m1 > 1024)   lane <-> v-row walks i = 1 .. m1 exactly as hadi_small_seq_kernel does in LDS, here on the
//                                  global arrays: explicit operators from a sliding window of three columns, Y0, forward
//                                  Thomas with the pivot recomputed on the fly; g_i goes to Y, the multiplier c'_i to a
//                                  scratch array (the handle's Craig-Sneyd buffer R1), the way back runs on Y in place.
//                                  Rows are kept in natural order (hadi_pick_shape: B = 1, slot of node i >= 1 is i - 1).
//   hadi_pass_b_seq  (m2 > 527)    lane <-> storage column (coalesced): forward sweep with the UNCHUNKED factorisation
//                                  (layout: one chunk of all rows, hadi_setup_instance) written over Y, backward sweep into U
//                                  with the Ikonen-Toivanen projection for American sweeps (explicit (U, lambda_bar) pair).
// European, dividend and American sweeps, call and put, fp64 state; the P representation, Craig-Sneyd and the fp32 state stay
// with the streaming kernels (the host keeps them off these shapes).
template <int AMER>
__global__ void __launch_bounds__(64) hadi_pass_a_seq(HadiSweepArgs a, int n) {
    typedef double T;
    static_assert(AMER == 0 || AMER == 1, "explicit (U, lambda_bar) pair only");
    const int lane = threadIdx.x;
    const int nrows = a.L.nrows, rowp = a.L.rowp, m1 = a.L.m1, nslot = 64 * a.L.B * a.L.G;
    const int jblocks = (nrows + 63) / 64;
    if ((int)blockIdx.x >= a.n_inst * jblocks) return;
    const int inst = blockIdx.x / jblocks, jb = blockIdx.x - inst * jblocks;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int j = jb * 64 + lane;
    const bool act = j < nrows, last = (j == nrows - 1);
    const int jr = act ? j : 0;  // (idle lanes walk row 0 and store nothing)
    const T *__restrict__ Ub = reinterpret_cast<const T *>(a.U) + (size_t)inst * a.L.inst_stride;
    T *__restrict__ Yr = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride + (size_t)jr * rowp;
    double *__restrict__ Wr = a.R1 + (size_t)inst * a.L.inst_stride + (size_t)jr * rowp;  // c'_i at slot i - 1
    const double *__restrict__ Lr = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride + (size_t)jr * rowp : nullptr;
    const double *__restrict__ sc = a.scoef + (size_t)inst * 4 * nslot;  // Bm, Bp, Dm, Dp of node i at [k * nslot + i - 1]
    const double *__restrict__ b2g = a.b2row + (size_t)inst * rowp;
    double v, wm, wz, wp, a2l2, a2l1, a2m, a2u1, a2u2, b1val;
    int b1col;
    bool b1_at0;
    {
        const double *__restrict__ rc = a.rowc + ((size_t)inst * nrows + jr) * HADI_RC;
        v = rc[RC_V]; wm = rc[RC_WM]; wz = rc[RC_WZ]; wp = rc[RC_WP];
        a2l2 = rc[RC_L2]; a2l1 = rc[RC_L1]; a2m = rc[RC_M]; a2u1 = rc[RC_U1]; a2u2 = rc[RC_U2];
        b1val = rc[RC_B1VAL];
        const int b1raw = (int)rc[RC_B1COL];
        b1_at0 = b1raw == 0 || b1raw >= HADI_B1_BOTH;
        b1col = b1raw >= HADI_B1_BOTH ? b1raw - HADI_B1_BOTH : b1raw;
    }
    const double dt = ip.dt, thdt = ip.thdt, qd = ip.q, half_rd = ip.half_rd;
    const double inv0 = 1.0 / (1.0 + ip.thdt * ip.hr0);
    // v-neighbours clamped to the grid: a clamped row only ever meets a zero weight (hadi_small_seq_kernel)
    const T *pm2 = Ub + (size_t)(jr >= 2 ? jr - 2 : 0) * rowp, *pm1 = Ub + (size_t)(jr >= 1 ? jr - 1 : 0) * rowp;
    const T *pr0 = Ub + (size_t)jr * rowp;
    const T *pp1 = Ub + (size_t)(jr + 1 < nrows ? jr + 1 : nrows - 1) * rowp, *pp2 = Ub + (size_t)(jr + 2 < nrows ? jr + 2 : nrows - 1) * rowp;
    // node i of a row: slot nslot for i = 0, slot i - 1 for 1 <= i <= m1, zero beyond (the s-neighbour of the last node)
    auto at = [&](const T *row, int i) -> double { return i == 0 ? (double)row[nslot] : (i <= m1 ? (double)row[i - 1] : 0.0); };
    auto b2at = [&](int i) -> double { return (last && i <= m1) ? (i == 0 ? b2g[nslot] : b2g[i - 1]) : 0.0; };
    const double e_nm1 = exp(ip.bc_rate * ip.dt * (n - 1));  // device_solver.hpp:238
    const double e_n = exp(ip.bc_rate * ip.dt * n);          // device_solver.hpp:246
    const double b1l = b1val * (dt * e_nm1 + thdt * (e_n - e_nm1));
    // column i = 0 (A0 and A1 rows are zero there; only A2 and the boundary act)
    const double c0m2 = at(pm2, 0), c0m1 = at(pm1, 0), c00 = at(pr0, 0), c0p1 = at(pp1, 0), c0p2 = at(pp2, 0);
    double r_m2 = at(pm2, 1), r_m1 = at(pm1, 1), r_0 = at(pr0, 1), r_p1 = at(pp1, 1), r_p2 = at(pp2, 1);
    double yout_c0, x0;
    {
        const double a2c0 = a2l2 * c0m2 + a2l1 * c0m1 + a2m * c00 + a2u1 * c0p1 + a2u2 * c0p2;
        const double b1c0 = b1_at0 ? b1val : 0.0;
        const double b2c0 = b2at(0);
        const double lamc0 = (AMER == 1) ? Lr[nslot] : 0.0;
        const double a1c0 = -ip.hr0 * c00;
        double y0c0 = c00 + dt * (a2c0 + a1c0 + (b1c0 + b2c0) * e_nm1 + lamc0);
        y0c0 = y0c0 + thdt * (b1c0 * e_n - (a1c0 + b1c0 * e_nm1));
        const double c2c0 = thdt * (b2c0 * e_n - (a2c0 + b2c0 * e_nm1));
        x0 = y0c0 * inv0;
        yout_c0 = x0 + c2c0;
    }
    double u_prev = c00, u_cur = r_0;
    double t_prev = wm * c0m1 + wz * c00 + wp * c0p1;
    double t_cur = wm * r_m1 + wz * r_0 + wp * r_p1;
    double a2u_cur = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
    double b2c = b2at(1);
    double corr_cur = thdt * (b2c * e_n - (a2u_cur + b2c * e_nm1));
    r_m2 = at(pm2, 2); r_m1 = at(pm1, 2); r_0 = at(pr0, 2); r_p1 = at(pp1, 2); r_p2 = at(pp2, 2);
    double cp_prev = 0.0, ys_prev = x0;  // x_0 is known: with ys_0 = x_0 and c'_0 = 0 the general step moves it to the right-hand side
    for (int i = 1; i <= m1; i++) {
        const double u_next = r_0;
        const double t_next = wm * r_m1 + wz * r_0 + wp * r_p1;
        const double a2u_next = fma(a2u2, r_p2, fma(a2l2, r_m2, a2l1 * r_m1 + a2m * r_0 + a2u1 * r_p1));
        const int inx = i + 2;
        r_m2 = at(pm2, inx); r_m1 = at(pm1, inx); r_0 = at(pr0, inx); r_p1 = at(pp1, inx); r_p2 = at(pp2, inx);
        const double Bm = sc[0 * nslot + i - 1], Bp = sc[1 * nslot + i - 1], Dm = sc[2 * nslot + i - 1], Dp = sc[3 * nslot + i - 1];
        const double lo = fma(v, Dm, qd * Bm);
        const double up = fma(v, Dp, qd * Bp);
        const double mn = -((lo + up) + half_rd);
        const double A1U = lo * u_prev + mn * u_cur + up * u_next;
        const double A0U = Bm * t_prev - (Bm + Bp) * t_cur + Bp * t_next;
        double S = A0U + A1U + a2u_cur;  // Y0 = U + dt (A0U + A1U + A2U + b e_{n-1} [+ lambda_bar]) + ..., device_solver.hpp:236-250
        S += b2c * e_nm1;
        if constexpr (AMER == 1) S += Lr[i - 1];
        double y = fma(dt, S, u_cur);
        y = fma(-thdt, A1U, y);
        y += (i == b1col) ? b1l : 0.0;
        const double il = -thdt * lo;
        const double im = 1.0 - thdt * mn;
        const double iu = -thdt * up;
        const double inv = hadi_rcp(fma(-il, cp_prev, im));
        const double cp = iu * inv;
        const double ys = fma(-il, ys_prev, y) * inv;
        const double b2n = b2at(i + 1);
        const double corr_next = thdt * (b2n * e_n - (a2u_next + b2n * e_nm1));
        if (act) {
            Yr[i - 1] = ys + corr_cur + cp * corr_next;  // g_i (c'_{m1} = 0: the row ends there)
            Wr[i - 1] = cp;
        }
        u_prev = u_cur; u_cur = u_next;
        t_prev = t_cur; t_cur = t_next;
        a2u_cur = a2u_next; corr_cur = corr_next; b2c = b2n;
        cp_prev = cp; ys_prev = ys;
    }
    if (act) {  // back substitution on the output itself: Y_i = g_i - c'_i Y_{i+1}
        double Yn = (double)Yr[m1 - 1];
        for (int i = m1 - 1; i >= 1; i--) {
            Yn = fma(-Wr[i - 1], Yn, (double)Yr[i - 1]);
            Yr[i - 1] = (T)Yn;
        }
        Yr[nslot] = (T)yout_c0;
    }
}

template <int AMER>
__global__ void __launch_bounds__(64) hadi_pass_b_seq(HadiSweepArgs a, int n) {
    typedef double T;
    static_assert(AMER == 0 || AMER == 1, "explicit (U, lambda_bar) pair only");
    const int lane = threadIdx.x;
    const int nrows = a.L.nrows, rowp = a.L.rowp;
    if ((int)blockIdx.x >= a.n_inst * a.ctiles) return;
    const int inst = blockIdx.x / a.ctiles, tile = blockIdx.x - inst * a.ctiles;
    const HadiInstPar ip = a.ipar[inst];
    if (n > ip.N) return;
    const int col = tile * 64 + lane;
    const bool valid = col < rowp;
    const int colc = valid ? col : rowp - 1;
    T *__restrict__ Yc = reinterpret_cast<T *>(a.Y) + (size_t)inst * a.L.inst_stride + colc;
    T *__restrict__ Uc = reinterpret_cast<T *>(a.U) + (size_t)inst * a.L.inst_stride + colc;
    const double *__restrict__ pb = a.pb + (size_t)inst * a.L.nrows_pad * HADI_PBW;
    // forward: y_k = (rhs_k - L y_{k-1} - L2 y_{k-2}) Q, written over the right-hand side
    double ym1 = 0.0, ym2 = 0.0;
    for (int k0 = 0; k0 < nrows; k0 += 8) {
        double rhs[8];
#pragma unroll
        for (int q = 0; q < 8; q++) rhs[q] = (k0 + q < nrows) ? (double)Yc[(size_t)(k0 + q) * rowp] : 0.0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (k0 + q < nrows) {
                const double *t = pb + (size_t)(k0 + q) * HADI_PBW;
                const double yk = (rhs[q] - t[PB_L] * ym1 - t[PB_L2] * ym2) * t[PB_Q];
                if (valid) Yc[(size_t)(k0 + q) * rowp] = yk;
                ym2 = ym1;
                ym1 = yk;
            }
        }
    }
    // backward: x_k = y_k - C x_{k+1} - C2 x_{k+2}  (+ Ikonen-Toivanen projection, device_solver.hpp:358-372)
    const double *__restrict__ P0 = (AMER == 1) ? a.U0 + (size_t)inst * a.L.inst_stride + colc : nullptr;
    double *__restrict__ Lc = (AMER == 1) ? a.LAM + (size_t)inst * a.L.inst_stride + colc : nullptr;
    const double dt = ip.dt;
    const bool is_smax = (col == a.pos_m1);
    double xp1 = 0.0, xp2 = 0.0;
    for (int k = nrows - 1; k >= 0; k--) {
        const double *t = pb + (size_t)k * HADI_PBW;
        const double yk = Yc[(size_t)k * rowp];
        const double xk = yk - t[PB_C] * xp1 - t[PB_C2] * xp2;
        xp2 = xp1;
        xp1 = xk;
        if constexpr (AMER == 1) {
            const double lamv = Lc[(size_t)k * rowp], pay = P0[(size_t)k * rowp];
            const double un = fmax(xk - dt * lamv, pay);
            double ln = fmax(0.0, lamv + (pay - xk) / dt);
            if (is_smax) ln = 0.0;
            if (valid) {
                Uc[(size_t)k * rowp] = (T)un;
                Lc[(size_t)k * rowp] = ln;
            }
        } else if (valid) {
            Uc[(size_t)k * rowp] = (T)xk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Setup: one block per instance builds all operator tables (replaces bounds.initialize and the three
// build_matrix calls at the top of every reference launcher, e.g. jacobian_computation.cpp:255-261).
struct HadiSetupArgs {
    HadiLayout L;
    int n_inst;
    const double *vec_s, *vec_v, *delta_s, *delta_v;  // [n][..] natural arrays (device)
    const double *par;  // [n][8]: rho, sigma, kappa, eta, dt, N (as double), strike (put), option type (0 call, 1 put)
    double r_d, r_f, theta;
    double *scoef, *b2row, *rowc, *a2i, *pb, *rinv, *rwork;
    HadiInstPar *ipar;
};

struct HadiBlockSync {
    HADI_DEV void operator()() const { __syncthreads(); }
};

__global__ void __launch_bounds__(256) hadi_setup_kernel(HadiSetupArgs s) {
    const int inst = blockIdx.x;
    if (inst >= s.n_inst) return;
    const HadiLayout &L = s.L;
    HadiSetupIn in;
    in.vec_s = s.vec_s + (size_t)inst * (L.m1 + 1);
    in.vec_v = s.vec_v + (size_t)inst * (L.m2 + 1);
    in.delta_s = s.delta_s + (size_t)inst * L.m1;
    in.delta_v = s.delta_v + (size_t)inst * L.m2;
    const double *par = s.par + (size_t)inst * 8;
    in.rho = par[0]; in.sigma = par[1]; in.kappa = par[2]; in.eta = par[3];
    in.dt = par[4]; in.N = (int)par[5];
    in.r_d = s.r_d; in.r_f = s.r_f; in.theta = s.theta;
    in.strike = par[6]; in.put = (par[7] != 0.0) ? 1 : 0;
    HadiTables t;
    const int n4 = 4 * L.P;
    t.scoef = s.scoef + (size_t)inst * 4 * 64 * L.B * L.G;
    t.b2row = s.b2row + (size_t)inst * L.rowp;
    t.rowc = s.rowc + (size_t)inst * L.nrows * HADI_RC;
    t.a2i = s.a2i + (size_t)inst * 5 * L.nrows_pad;
    t.pb = s.pb + (size_t)inst * L.nrows_pad * HADI_PBW;
    t.rinv = s.rinv + (size_t)inst * n4 * n4;
    t.rwork = s.rwork + (size_t)inst * n4 * 2 * n4;
    t.ipar = s.ipar + inst;
    hadi_setup_instance(L, in, t, (int)threadIdx.x, (int)blockDim.x, HadiBlockSync());
}

// ------------------------------------------------------------------------------------------------
// Layout conversion natural [inst][j][i] <-> internal [inst][j][pos(i)] (pads written as 0).
// Instance k of the internal array reads natural instance k % n_src (a Jacobian batch replicates U_0).
__global__ void __launch_bounds__(256) hadi_pack_kernel(HadiLayout L, int n_inst, int n_src,
                                                        const double *__restrict__ nat, double *__restrict__ internal) {
    const size_t total = (size_t)n_inst * L.nrows_pad * L.rowp;
    const size_t m = (size_t)(L.m1 + 1) * L.nrows;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int slot = (int)(e % L.rowp);
        const size_t rowid = e / L.rowp;
        const int j = (int)(rowid % L.nrows_pad);
        const size_t inst = rowid / L.nrows_pad;
        const int i = hadi_slot_to_i(L, slot);
        double v = 0.0;
        if (i >= 0 && i <= L.m1 && j < L.nrows) v = nat[(inst % (size_t)n_src) * m + (size_t)j * (L.m1 + 1) + i];
        internal[e] = v;
    }
}

__global__ void __launch_bounds__(256) hadi_unpack_kernel(HadiLayout L, int n_inst, const double *__restrict__ internal,
                                                          double *__restrict__ nat) {
    const size_t m = (size_t)(L.m1 + 1) * L.nrows;
    const size_t total = (size_t)n_inst * m;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % (L.m1 + 1));
        const size_t rowid = e / (L.m1 + 1);
        const int j = (int)(rowid % L.nrows);
        const size_t inst = rowid / L.nrows;
        nat[e] = internal[inst * L.inst_stride + (size_t)j * L.rowp + hadi_pos(L, i)];
    }
}

__global__ void __launch_bounds__(256) hadi_fill_kernel(double *__restrict__ p, size_t n, double v) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) p[e] = v;
}

// ------------------------------------------------------------------------------------------------
// Discrete dividend jump (device_solver.hpp:448-504) on the internal layout: U <- interp(UT) where
// UT is a copy of U taken before the jump.  One thread per (row, s-node); the reference's linear
// search "first k with s_k > new_s" is a binary search on the ascending s-grid.  Which dividend (if any) an
// instance pays at the start of step n comes from the host-built table div_flag (see HadiSmallArgs).
__global__ void __launch_bounds__(256) hadi_dividend_kernel(HadiLayout L, int n_inst, const HadiInstPar *__restrict__ ipar,
                                                            const double *__restrict__ vec_s,
                                                            const double *__restrict__ UT, double *__restrict__ U,
                                                            const int *__restrict__ div_flag, int flag_stride, int n,
                                                            const double *__restrict__ div_amounts,
                                                            const double *__restrict__ div_pcts) {
    const int m1 = L.m1;
    const size_t per = (size_t)L.nrows * (m1 + 1);
    const size_t total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(e % (m1 + 1));
        const size_t rowid = e / (m1 + 1);
        const int j = (int)(rowid % L.nrows);
        const size_t inst = rowid / L.nrows;
        const int dv = div_flag[inst * flag_stride + n - 1];
        if (dv < 0) continue;
        const double amount = div_amounts[dv], pct = div_pcts[dv];
        const double *__restrict__ s = vec_s + inst * (m1 + 1);
        const double *__restrict__ src = UT + inst * L.inst_stride + (size_t)j * L.rowp;
        const double old_s = s[i];
        const double new_s = old_s * (1.0 - pct) - amount;
        // ex-dividend spot <= 0: a call is worth 0 (device_solver.hpp:499-503), a put its s = 0 value
        double out = ipar[inst].put ? src[hadi_pos(L, 0)] : 0.0;
        if (new_s > 0) {
            // idx = first k in [0, m1] with s[k] > new_s, 0 if none
            int lo = 0, hi = m1 + 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s[mid] > new_s) hi = mid;
                else lo = mid + 1;
            }
            const int idx = (lo <= m1) ? lo : 0;
            if (idx > 0) {
                const double s_low = s[idx - 1], s_high = s[idx];
                const double weight = (new_s - s_low) / (s_high - s_low);
                const double val_low = src[hadi_pos(L, idx - 1)], val_high = src[hadi_pos(L, idx)];
                out = (1.0 - weight) * val_low + weight * val_high;
            } else {
                out = src[hadi_pos(L, 0)];
            }
        }
        U[inst * L.inst_stride + (size_t)j * L.rowp + hadi_pos(L, i)] = out;
    }
}

// ------------------------------------------------------------------------------------------------
// fp32-state sweep: the packed state is rounded to fp32 before the time loop and widened after it (same element layout).
__global__ void __launch_bounds__(256) hadi_narrow_kernel(HadiLayout L, const double *__restrict__ src, float *__restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / L.rowp;
        const int x = (int)(e - row * L.rowp), i = hadi_slot_to_i(L, x);  // fp64 slot -> node -> fp32 slot (pads map to themselves)
        dst[row * L.rowp + (i >= 0 ? hadi_pos_f32(L.B, L.G, i) : x)] = (float)src[e];
    }
}
__global__ void __launch_bounds__(256) hadi_widen_kernel(HadiLayout L, const float *__restrict__ src, double *__restrict__ dst, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / L.rowp;
        const int x = (int)(e - row * L.rowp), i = hadi_slot_to_i(L, x);
        dst[e] = (double)src[row * L.rowp + (i >= 0 ? hadi_pos_f32(L.B, L.G, i) : x)];
    }
}

// ------------------------------------------------------------------------------------------------
// American, P representation <-> explicit (U, lambda_bar), elementwise on the packed arrays (payoff = its v-row 0):
//   materialise:    U = max(P, U0),  lambda_bar = max(0, (U0 - P)/dt)  (0 at i = m1)
//   dematerialise:  P = lambda_bar > 0 ? U0 - dt lambda_bar : U        (after a projection lambda_bar > 0 implies U = U0)
// Used for the first step (the caller's initial U need not dominate the payoff), around dividend steps (the jump acts
// on U alone) and for the outputs.
__global__ void __launch_bounds__(256) hadi_am_materialise_kernel(HadiLayout L, int n_inst, const HadiInstPar *__restrict__ ipar,
                                                                  const double *__restrict__ P0, double *__restrict__ UP,
                                                                  double *__restrict__ LAM, int pos_m1) {
    const size_t per = (size_t)L.nrows * L.rowp, total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / per, r = e - inst * per;
        const int x = (int)(r % L.rowp);
        const double pay = P0[inst * L.inst_stride + x], p = UP[inst * L.inst_stride + r];
        double lam = fmax(0.0, (pay - p) / ipar[inst].dt);
        if (x == pos_m1) lam = 0.0;
        UP[inst * L.inst_stride + r] = fmax(p, pay);
        LAM[inst * L.inst_stride + r] = lam;
    }
}
__global__ void __launch_bounds__(256) hadi_am_dematerialise_kernel(HadiLayout L, int n_inst, const HadiInstPar *__restrict__ ipar,
                                                                    const double *__restrict__ P0, double *__restrict__ UP,
                                                                    const double *__restrict__ LAM) {
    const size_t per = (size_t)L.nrows * L.rowp, total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / per, r = e - inst * per;
        const int x = (int)(r % L.rowp);
        const double lam = LAM[inst * L.inst_stride + r];
        if (lam > 0.0) UP[inst * L.inst_stride + r] = P0[inst * L.inst_stride + x] - ipar[inst].dt * lam;
    }
}

// ------------------------------------------------------------------------------------------------
// American payoff shape: mis[inst] != 0 if any v-row of the packed payoff differs from its row 0 (mis is zeroed first).
__global__ void __launch_bounds__(256) hadi_payoff_shape_kernel(HadiLayout L, int n_inst, const double *__restrict__ P0, int *__restrict__ mis) {
    const size_t per = (size_t)L.nrows * L.rowp, total = (size_t)n_inst * per;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / per, r = e - inst * per;
        const size_t x = r % L.rowp;
        if (P0[inst * L.inst_stride + r] != P0[inst * L.inst_stride + x]) mis[inst] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// GridViews::rebuild_variance_views (grid_pod.hpp:25-73) for every instance of the batch, each for its own V_0: one
// block per instance.  v_j = d sinh(j asinh(V/d)/m2), j = 0..m2; V_0 is pushed, the m2+2 values are sorted and the
// largest is dropped (the reference bubble-sorts them on one thread, grid_pod.hpp:47-57; the raw nodes are ascending,
// so sorting = inserting V_0 behind the last node <= V_0).  Delta_v follows.  sinh/asinh are the device library's, as
// in the reference's in-kernel rebuild.
__global__ void __launch_bounds__(256) hadi_rebuild_variance_kernel(int m2, int n_inst, const double *__restrict__ v0_i,
                                                                    double V, double d, double *__restrict__ vec_v,
                                                                    double *__restrict__ delta_v) {
    HADI_DYN_SMEM(double, raw);  // 2 (m2 + 1) doubles
    const int inst = blockIdx.x;
    if (inst >= n_inst) return;
    const int n = m2 + 1;
    double *outv = raw + n;
    const double V_0 = v0_i[inst];
    const double Delta_eta = (1.0 / m2) * asinh(V / d);
    for (int j = threadIdx.x; j < n; j += blockDim.x) raw[j] = d * sinh(j * Delta_eta);
    __syncthreads();
    int lo = 0, hi = n;  // pos = number of raw nodes <= V_0 (first index with raw > V_0)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (raw[mid] > V_0) hi = mid;
        else lo = mid + 1;
    }
    const int pos = lo;  // pos == n: V_0 is the largest of the m2+2 values and is the one dropped
    for (int k = threadIdx.x; k < n; k += blockDim.x) outv[k] = (k < pos) ? raw[k] : (k == pos) ? V_0 : raw[k - 1];
    __syncthreads();
    double *vv = vec_v + (size_t)inst * n, *dv = delta_v + (size_t)inst * m2;
    for (int k = threadIdx.x; k < n; k += blockDim.x) vv[k] = outv[k];
    for (int k = threadIdx.x; k < m2; k += blockDim.x) dv[k] = outv[k + 1] - outv[k];
}

// ------------------------------------------------------------------------------------------------
// Levenberg-Marquardt normal equations of this rank's rows on the device (replaces KokkosBlas::gemm("T","N") /
// gemv("T") and the residual kernel, jacobian_computation.cpp:117,154, heston_calibration.cpp:271-275):
//   out[0..24] = J^T J (row-major), out[25..29] = J^T r, out[30] = sum r^2,  r = market - model.
// One block, fixed-shape tree reduction: the result does not depend on scheduling (n is a few thousand at most).
__global__ void __launch_bounds__(256) hadi_lm_partials_kernel(int n, const double *__restrict__ J,
                                                               const double *__restrict__ model,
                                                               const double *__restrict__ market, double *__restrict__ out) {
    HADI_DYN_SMEM(double, redm);  // 21 x 256 doubles
    double (*red)[256] = reinterpret_cast<double (*)[256]>(redm);
    double acc[21];
#pragma unroll
    for (int q = 0; q < 21; q++) acc[q] = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) {
        double jr[5];
#pragma unroll
        for (int a = 0; a < 5; a++) jr[a] = J[(size_t)k * 5 + a];
        const double r = market[k] - model[k];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 5; a++)
#pragma unroll
            for (int b = a; b < 5; b++) { acc[q] = fma(jr[a], jr[b], acc[q]); q++; }
#pragma unroll
        for (int a = 0; a < 5; a++) acc[15 + a] = fma(jr[a], r, acc[15 + a]);
        acc[20] = fma(r, r, acc[20]);
    }
#pragma unroll
    for (int q = 0; q < 21; q++) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int q = 0; q < 21; q++) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int q = 0;
        for (int a = 0; a < 5; a++)
            for (int b = a; b < 5; b++) {
                out[a * 5 + b] = red[q][0];
                out[b * 5 + a] = red[q][0];
                q++;
            }
        for (int a = 0; a < 5; a++) out[25 + a] = red[15 + a][0];
        out[30] = red[20][0];
    }
}

// ------------------------------------------------------------------------------------------------
// Price pick (jacobian_computation.cpp:275-288): first s-node with |s_i - S_0| < 1e-10, first
// v-node with |v_j - V_0| < 1e-10 (0 if none, grid_pod.hpp:76-87).  status[inst] = 1 if S_0 is off-grid.
__global__ void __launch_bounds__(64) hadi_pick_kernel(HadiLayout L, int n_inst, const double *__restrict__ vec_s,
                                                       const double *__restrict__ vec_v, const double *__restrict__ U,
                                                       double S_0, const double *__restrict__ V0_i, double V_0,
                                                       double *__restrict__ prices, int price_stride,
                                                       int *__restrict__ status) {
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= n_inst) return;
    const double *s = vec_s + (size_t)inst * (L.m1 + 1);
    const double *v = vec_v + (size_t)inst * (L.m2 + 1);
    const double v0 = V0_i ? V0_i[inst] : V_0;
    int is = -1, iv = 0;
    for (int i = 0; i <= L.m1; i++)
        if (fabs(s[i] - S_0) < 1e-10) { is = i; break; }
    for (int j = 0; j <= L.m2; j++)
        if (fabs(v[j] - v0) < 1e-10) { iv = j; break; }
    if (is < 0) {
        status[inst] = 1;
        prices[(size_t)inst * price_stride] = nan("");
        return;
    }
    status[inst] = 0;
    prices[(size_t)inst * price_stride] = U[(size_t)inst * L.inst_stride + (size_t)iv * L.rowp + hadi_pos(L, is)];
}

// J(k, param) = (perturbed price - base price) / eps from the 6 n0 prices of a flattened Jacobian sweep (groups: base, kappa,
// eta, sigma, rho, v0), jacobian_computation.cpp:329,360.
__global__ void __launch_bounds__(256) hadi_jacobian_rows_kernel(int n0, const double *__restrict__ prices, double eps,
                                                                 double *__restrict__ J, double *__restrict__ base) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n0) return;
    const double b = prices[k];
    base[k] = b;
    for (int g = 1; g <= 5; g++) J[(size_t)k * 5 + (g - 1)] = (prices[(size_t)g * n0 + k] - b) / eps;
}

// Diagnostics (hadi_debug_rcp): the reciprocal every line solve of the sweep uses, elementwise.
__global__ void __launch_bounds__(256) hadi_rcp_kernel(int n, const double *__restrict__ x, double *__restrict__ out) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) out[e] = hadi_rcp(x[e]);
}

// Replicate one of `nsrc` source rows (length len) into every instance's row: dst[inst] = src[sel[inst]]
// (sel == nullptr: src row 0).  Used to hand every instance the v-grid rebuilt for V_0 (or V_0+eps).
__global__ void __launch_bounds__(256) hadi_bcast_rows_kernel(int len, int n_inst, const double *__restrict__ src,
                                                              const int *__restrict__ sel, double *__restrict__ dst) {
    const size_t total = (size_t)n_inst * len;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t inst = e / len;
        const int k = (int)(e - inst * len);
        const int sr = sel ? sel[inst] : 0;
        dst[e] = src[(size_t)sr * len + k];
    }
}
