// ieskf_split.h — hand-off between the two kernels of the split IESKF path ("split" search mode).
//
//   K0  the persistent LDS-grid kernel (ieskf_lds_impl.h, SPLIT instantiation) runs the first
//       `split_iters` iterations with full correspondence searches — the iterations in which the
//       linearisation point still moves by decimetres — and then, at the state the next iteration
//       starts from, gathers for every query a CANDIDATE LIST: every target point within a stated
//       radius of the query's de-skewed position (the "anchor"), per class of rings.
//   K1  (ieskf_k1.hip) runs the remaining iterations without any grid: each iteration re-decides a
//       query's three points among its listed candidates with the reference's exact comparison
//       rules (SE:847-910, 973-1024), and accepts the decision only when it is CERTIFIED: every
//       point that is not in the list is provably farther than the chosen one (completeness radius
//       minus how far the query has drifted from its anchor, with slack for the f32 roundings).
//       An uncertified decision is replaced by an exhaustive search of that query's target cloud by
//       the whole workgroup (exact by construction) — rare: after `split_iters` iterations the
//       queries move by millimetres.
//
// Without the grid K1 needs ~7 KB of LDS, so four scans are resident per CU with every wave fully
// populated (one lane per query, queries packed densely), instead of two half-empty workgroups.
//
// Completeness claims of a list (all distances are the f32 squared distances the searches compute):
//   r_nn   every point of the cloud (all rings, all indices) within r_nn of the anchor is listed
//   r2/r3  every point of the class-2 / class-3 rings UNDER rho0 within r2 / r3 is listed, where rings
//          above rho0 only list indices below the forward end min(N_query, N_target) (the only ones the
//          reference's forward walk can reach, SE:859, 983) — or, when the class had no winner at the
//          anchor (flag NONE2 / NONE3), only the indices the walk around the anchor's nearest neighbour
//          j1 reaches: such a claim holds while the nearest neighbour is still j1.
#pragma once

#include "ieskf_binned.h"
#include "ieskf_device.h"

namespace lins {

constexpr int kSplitK = 40;  // candidate slots per query

enum { SPLIT_DONE = 0, SPLIT_CONTINUE = 1 };
enum { SPLITQ_NONE2 = 1, SPLITQ_NONE3 = 2 };

struct SplitScan {  // per scan, K0 -> K1
  double lin[19];   // linState_ the next iteration starts from
  double res_prev, res_last, upd_norm;
  int iter, status;
  int ring_start[2][kRingsBinned + 1];  // [0] surf targets, [1] corner targets (original index space)
  int dbg[4];
};

struct SplitQ {  // per query, 48 bytes
  float ax, ay, az, r_nn;  // anchor of the all-points claim: the query de-skewed with SplitScan::lin
  float bx, by, bz, r2;    // anchor of the class claims (= a until the list kernel re-establishes them)
  float r3;
  int meta;  // count | rho0 << 8 | flags << 16
  int j1;    // nearest neighbour at the anchor (original index), -1: none
  int jc;    // -1: class claims as the grid kernel stated them (rules above); >= 0: re-established by the list
             // kernel after an exhaustive walk: everything the walk around nearest neighbour jc reaches within
             // r2 / r3 of b is listed — they hold while the nearest neighbour is jc
};
static_assert(sizeof(SplitQ) == 48, "three 16-byte loads");

// candidate: (x, y, z, bits(j | ring << 16)), j = original index in its cloud
__device__ __forceinline__ float4 split_pack(float x, float y, float z, int j, int ring) {
  return make_float4(x, y, z, __int_as_float(j | (ring << 16)));
}

}  // namespace lins
