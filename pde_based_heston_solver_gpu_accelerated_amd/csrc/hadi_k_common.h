// hadi_k_common.h -- shared device helpers of the sweep kernels: the argument block, lane exchanges, buffer addressing, scalar row loads, LDS-DMA row copies.
// Part of libhadi's device code: include through hadi_kernels.h (which fixes the order).
#pragma once

struct HadiSweepArgs {
    // state, internal layout [inst][row][rowp]
    double *U;         // solution
    double *Y;         // A2 right-hand side between the passes
    double *LAM;       // lambda_bar (American) or nullptr
    double *R1, *C2;   // Craig-Sneyd only: predictor quantities reused by the corrector (see hadi_row_step)
    double *rs_tab;    // paired strips only: [inst][v-row][half][lane] -- the cyclic-reduction image of the pair's coupling column (hadi_strip_step, RSTAB)
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
#ifndef HADI_AUX_NT_ST   // (stores of the column pass: the same policy unless an experiment overrides it, tools/build_variant.py)
#define HADI_AUX_NT_ST HADI_AUX_NT
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
    __builtin_amdgcn_raw_buffer_store_b64(d, b.r, voff_bytes, soff_bytes, HADI_AUX_NT_ST);
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
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)v), b.r, voff_bytes, soff_bytes, HADI_AUX_NT_ST);
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
    emu::wave_barrier();
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

