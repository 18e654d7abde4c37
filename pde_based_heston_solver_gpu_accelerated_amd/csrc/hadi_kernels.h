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
//
// This header is the umbrella: the kernels live in the hadi_k_*.h files below, in dependency order.
#pragma once
#include "hadi_core.h"

#include "hadi_k_common.h"
#include "hadi_k_row_ring.h"
#include "hadi_k_row_strip.h"
#include "hadi_k_col.h"
#include "hadi_k_team.h"
#include "hadi_k_small.h"
#include "hadi_k_seq.h"
#include "hadi_k_aux.h"
