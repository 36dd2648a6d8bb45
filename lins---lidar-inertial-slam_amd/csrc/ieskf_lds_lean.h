// ieskf_lds_lean.h — the register-lean search core of the batch kernel (round 6).  Included by ieskf_lds_impl.h INSIDE its
// instantiation namespace, after the grid / window helpers (scan_cols, reach, reach_elev, ring_in_reach, coop_map ...).
//
// Why: the batch kernel (lds_mr::ieskf_lds_kernel<512, 1>) ran at 126 VGPRs = four waves per SIMD = two scans per CU; a
// third scan per CU needs <= 80 VGPRs (six waves per SIMD: 512 / 6 = 85, allocated in eights) and <= 53 248 B of LDS.
// Forcing the round-5 source to 80 registers spills 109 dwords per lane (0.76 ms against 0.56).  What this file changes
// against nn_lds / walk_lds / the per-lane carried state of ieskf_lds_impl.h — same searches, same results:
//
//   Top2      the running best of a search as TWO packed 64-bit keys + one float (5 registers; Best: 9).  The low word of a
//             key carries the tie key AND the identity of the point — nearest neighbour: (original index + 1) << 18 | grid
//             position << 4 | ring; walk: visit rank << 14 | grid position — so winner / runner-up need no side registers.
//             The tie key sits in the high bits: the order of two keys is the reference's (distance, index) / (distance,
//             visit rank) order, SE:851-910, 973-1024.
//   carry     what a query keeps from iteration to iteration (tracked candidates, certificates: 13 words) lives in GLOBAL
//             memory per query slot (KernelArgs::relay_lane, four 16-byte planes per scan) instead of 19 registers per lane
//             across the whole loop: the nearest-neighbour part is read with the query point at the top of the iteration,
//             the walk part behind the nearest-neighbour phase; a part is written back only when it changed (a search ran,
//             two tracked candidates changed places, the nearest neighbour moved).  The several-part updates of a large
//             batch hand over nothing per lane any more: the next part finds the records where this one left them.
//   owners    a lane whose selection is certified is FINISHED (prediction taken, record written) before the wave's searches
//             start, so that across a search only the de-skewed query, the selected positions and a few flags are live.
//
// Accesses to the carry records are device-coherent (sc1 buffer accesses): the parts of an update may run on different
// XCDs, and a wave re-reads in iteration k + 1 what it wrote in iteration k.
#pragma once

// ---- Top2: winner, runner-up and the closest of everybody else ------------------------------------------------------------
struct Top2 {
  unsigned long long k;   // winner (distance bits << 32 | low word); the sentinel (threshold, 0) until a candidate beats it
  unsigned long long k2;  // runner-up; sentinels: (+inf, 0xFFFFFFFF) = none yet, (threshold, 0) = the dethroned sentinel
  float omin;             // smallest squared distance of every OTHER candidate seen
};
constexpr unsigned long long kT2None = 0x7F800000FFFFFFFFull;
__device__ __forceinline__ Top2 t2_init(float thr) { return Top2{(unsigned long long)__float_as_uint(thr) << 32, kT2None, INFINITY}; }
__device__ __forceinline__ float t2_d(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }
// a key that names a point (low words of points are never 0 — see nn_low / walk_low — and never 0xFFFFFFFF)
__device__ __forceinline__ bool t2_real(unsigned long long k) { return (unsigned)k != 0u && (unsigned)k != 0xFFFFFFFFu; }
__device__ __forceinline__ unsigned nn_low(int j, int pos, int ring) { return ((unsigned)(j + 1) << 18) | ((unsigned)pos << 4) | (unsigned)ring; }
__device__ __forceinline__ int nn_pos(unsigned low) { return (int)((low >> 4) & 0x3FFFu); }
__device__ __forceinline__ int nn_ring(unsigned low) { return (int)(low & 15u); }
static_assert(kGridNpMax <= 0x3000 && kRingsBinned <= 16, "key layout: 14 bits of index + 1, 14 bits of position, 4 bits of ring");
// visit rank of the index walk in 15 bits: forward part (j1, ...) ascending first, then the backward part descending
__device__ __forceinline__ unsigned walk_low(int rank, int pos) {
  const unsigned r15 = rank < kBackRankL ? (unsigned)rank : (0x4000u | (unsigned)(rank - kBackRankL));
  return (r15 << 14) | (unsigned)pos;
}
__device__ __forceinline__ int walk_pos(unsigned low) { return (int)(low & 0x3FFFu); }

// insert a candidate into the top two; equal keys are the same point seen again.  (insert_key of ieskf_lds_impl.h: "pos2 >=
// 0" there is "the runner-up's low word is not 0" here — the dethroned sentinel never counts as a seen candidate; the
// (+inf, ...) sentinel does no harm in a minimum)
__device__ __forceinline__ void t2_insert(Top2& b, unsigned long long k) {
  if (k < b.k) {
    if ((unsigned)b.k2 != 0u) b.omin = fminf(b.omin, t2_d(b.k2));
    b.k2 = b.k, b.k = k;
  } else if (k > b.k) {
    if (k < b.k2) {
      if ((unsigned)b.k2 != 0u) b.omin = fminf(b.omin, t2_d(b.k2));
      b.k2 = k;
    } else if (k > b.k2) {
      b.omin = fminf(b.omin, t2_d(k));
    }
  }
}
// the scan loops' form (consider_scan of ieskf_lds_impl.h): nearly every point only lowers omin
template <class LowF>
__device__ __forceinline__ void t2_scan(Top2& b, bool ok, float d, LowF low) {
  const bool cand = ok && d <= t2_d(b.k2);
  b.omin = (ok && !cand) ? fminf(b.omin, d) : b.omin;
  if (cand) t2_insert(b, ((unsigned long long)__float_as_uint(d) << 32) | low());
}
__device__ __forceinline__ void t2_merge_from(Top2& b, int src_lane) {
  const unsigned lo = __shfl((unsigned)b.k, src_lane), hi = __shfl((unsigned)(b.k >> 32), src_lane);
  const unsigned lo2 = __shfl((unsigned)b.k2, src_lane), hi2 = __shfl((unsigned)(b.k2 >> 32), src_lane);
  b.omin = fminf(b.omin, __shfl(b.omin, src_lane));
  if (lo != 0u) t2_insert(b, ((unsigned long long)hi << 32) | lo);  // (a winner is the threshold sentinel or a point)
  if (lo2 != 0u && lo2 != 0xFFFFFFFFu) t2_insert(b, ((unsigned long long)hi2 << 32) | lo2);
}
// fold the partial results of a query's ln lanes (consecutive, a power of two, wave-uniform) into every one of them
__device__ __forceinline__ void t2_merge_lanes(Top2& b, int lane, int ln) {
#pragma unroll 1
  for (int m = 1; m < ln; m <<= 1) t2_merge_from(b, lane ^ m);
}
__device__ __forceinline__ float t2_cert_lb(const Top2& b, float margin) {  // cert_lb(): the sentinel winner's distance IS the threshold
  return fminf(bound_sqrtf(b.omin), bound_sqrtf(t2_d(b.k)) + margin);
}

#ifndef LINS_HARD_COLS
#define LINS_HARD_COLS 16
#endif
constexpr int kHardCols = LINS_HARD_COLS;  // columns of a cold search's window beyond which the search is deferred (nn_lean)
constexpr int kHardLanes = 32;             // lanes a deferred search may take (32 ring windows)
// ---- pass 1: exact nearest neighbour (nn_lds<0> on Top2).  ln lanes per query, role = this lane's number among them ---------
__device__ __forceinline__ Top2 nn_lean(const LdsStore& L, const LCloud& c, float sx, float sy, float sz, float rho, float qn3, float el_q, float gq,
                                        float thr, float margin, int rq, int ln, int role, int lane, int warm_pos, int warm_ring, bool reseed,
                                        bool may_defer, bool& deferred) {
  Top2 b = t2_init(thr);
  deferred = false;
  // Warm start: last search's nearest neighbour is still a candidate, and its distance to the re-de-skewed query bounds
  // the search from the start.  That bound is only as good as the query stood still: after a large step of the state (the
  // first iterations of an update whose prior is off by a metre) it is the step's length, and the windows it opens hold
  // hundreds of points — `reseed` (the owner's call: the query moved farther than prm.reseed_drift since its last search)
  // adds the cold search's seed scan, whose bound is the local point spacing.  Bounds only: the result is the exact arg-min
  // either way.
  const bool warm = warm_pos >= 0 && !reseed;
  if (warm_pos >= 0) {
    const unsigned low = nn_low(pt_idx(L, c, warm_pos), warm_pos, warm_ring);
    t2_insert(b, ((unsigned long long)__float_as_uint(pt_sqdist(L, c, warm_pos, sx, sy, sz)) << 32) | low);
  }
  rq = rq < 0 ? 0 : (rq >= kRingsBinned ? kRingsBinned - 1 : rq);
  const int a0 = az_col_of(gq, c.naz);
  int rcur = rq;
  auto f = [&](float x, float y, float z, int j, int p, bool ok) {
    t2_scan(b, ok, sqdist3(x, y, z, sx, sy, sz), [&] { return nn_low(j, p, rcur); });
  };
  const bool own = ring_nonempty(c, rq);
  {  // cold seed: columns a0-1..a0+1 of the query's own ring and of its two neighbours (one ring per lane when ln > 1)
    const int rs = ln == 1 ? rq : rq + (role == 0 ? 0 : (role == 1 ? 1 : -1));
    const bool seed = !warm && role < 3 && ring_nonempty(c, rs);
    rcur = rs;
    scan_cols(L, c, rs, seed ? a0 - 1 : 1, seed ? a0 + 1 : 0, f);
    if (ln == 1) {
#pragma unroll 1
      for (int dr = -1; dr <= 1; dr += 2) {
        const bool sd2 = !warm && ring_nonempty(c, rq + dr);
        rcur = rq + dr;
        scan_cols(L, c, rq + dr, sd2 ? a0 - 1 : 1, sd2 ? a0 + 1 : 0, f);
      }
    } else if (!warm) {  // (the lanes of a query agree on `warm`)
      t2_merge_lanes(b, lane, ln);
    }
  }
  const int cin = warm ? 0 : 2;  // first column either side that the seed has not covered
  const float sqrtB = bound_sqrtf(t2_d(b.k)) + margin;  // fixed bound for everything below, inflated by the certificate margin
  const ColWin cw = reach_cols(c, rho, sqrtB, gq);
  // A HARD search: the seed found nothing near (an empty neighbourhood, a query displaced by a prior that is a metre off) and
  // the windows of the bound it left hold hundreds to thousands of points — swept by this query's one or two lanes while
  // the other 62 of the wave, then the workgroup at its barrier, then the launch wait for it: round 6 found single waves
  // at 330-390 k ticks of nearest-neighbour phase against 30-60 k of the other seven in the updates that end a launch
  // (tools/wave_phases.py WP_SLOWEST=1).  Such a search is DEFERRED: the wave runs its hard searches afterwards, up to 32
  // lanes each (one ring window per lane).  The lanes of a query agree: b is the same in all of them after the seed's merge.
  if (may_defer && !warm && cw.hi - cw.lo > kHardCols) {
    deferred = true;
    return b;
  }
  const int own_r = a0 + cin, own_l = a0 - (warm ? 1 : 2);  // own ring: first column right / left of what the seed covered
  const float delta = reach_elev(qn3, sqrtB);
  // this lane's tasks as a bit mask: ln == 1: bit 0 / 1 = own ring right / left of the seed, bit 2 + r = ring r;
  // ln > 1: bit i <-> task t = role + ln i (task 0, 1 = own ring right / left, 2.. = rings rq+1, rq-1, rq+2, ...)
  constexpr int kTasks = 2 + 2 * (kRingsBinned - 1);
  unsigned todo = 0;
  const bool by_ring = ln == 1;  // (wave-uniform)
  if (by_ring) {
    todo = own ? ((cw.hi >= own_r ? 1u : 0u) | (cw.lo <= own_l ? 2u : 0u)) : 0u;
#pragma unroll
    for (int r = 0; r < kRingsBinned; ++r) todo |= (r != rq && ring_in_reach(c, r, el_q, delta)) ? (4u << r) : 0u;
  } else {
#pragma unroll 1
    for (int i = 0, t = role; t < kTasks; ++i, t += ln) {
      const int k = t - 2, off = (k >> 1) + 1;
      const int r = t < 2 ? rq : rq + ((k & 1) ? -off : off);
      bool go;
      if (t < 2)
        go = own && (t == 0 ? cw.hi >= own_r : cw.lo <= own_l);
      else
        go = r >= 0 && r < kRingsBinned && ring_in_reach(c, r < 0 ? 0 : (r >= kRingsBinned ? kRingsBinned - 1 : r), el_q, delta);
      todo |= go ? (1u << i) : 0u;
    }
  }
#pragma unroll 1
  while (todo) {
    const int i = __ffs(todo) - 1;
    todo &= todo - 1;
    const int t = by_ring ? (i < 2 ? i : 2) : role + ln * i;
    const int k = t - 2, off = (k >> 1) + 1;
    const int r = by_ring ? (i < 2 ? rq : i - 2) : (t < 2 ? rq : rq + ((k & 1) ? -off : off));
    rcur = r;
    scan_cols(L, c, r, t == 0 ? own_r : cw.lo, t == 1 ? own_l : cw.hi, f);
  }
  t2_merge_lanes(b, lane, ln);
  return b;
}

// ---- pass 2 (SE:859-910 surf, SE:983-1024 corner): walk_lds<0> on Top2 -------------------------------------------------------
__device__ __forceinline__ void walk_task_lean(const LdsStore& L, const LCloud& c, const WalkCtx& w, int r, bool seed_first, bool centre_done, float gq,
                                               float sx, float sy, float sz, float rho_q, float qn3, float el_q, float margin, Top2& cur) {
  bool go = walk_ring_has_candidates(c, w, r);
  if (go) go = ring_in_reach(c, r, el_q, reach_elev(qn3, bound_sqrtf(t2_d(cur.k)) + margin));
  auto f = [&](float x, float y, float z, int j, int p, bool ok) {
    int rank;
    const bool in_walk = walk_rank(w, j, rank);
    t2_scan(cur, ok && in_walk, sqdist3(x, y, z, sx, sy, sz), [&] { return walk_low(rank, p); });
  };
  const int a0 = az_col_of(gq, c.naz);
  const bool seed = go && seed_first;
  scan_cols(L, c, r, seed ? a0 - 1 : 1, seed ? a0 + 1 : 0, f);
  const bool done = seed_first || centre_done;
  int kk = done ? 1 : -1;  // columns a0-kk..a0+kk are covered (-1: nothing yet)
  const int half = c.naz / 2;
  // Widen progressively: the window needed for the current bound (reach_cols: exact to the column function's error), but at
  // most 4x the width already covered per round — when the seed window was empty the bound tightens as soon as the first
  // real candidate shows up, instead of one sweep over the whole search radius.
#pragma unroll 1
  for (int round = 0; round < 8 && go; ++round) {
    const ColWin cw = reach_cols(c, rho_q, bound_sqrtf(t2_d(cur.k)) + margin, gq);
    const int kr = cw.hi - a0, kl = a0 - cw.lo, K = kr > kl ? kr : kl;
    if (K <= kk) break;
    const int nk = kk < 1 ? K : (K < 4 * kk ? K : 4 * kk);
    scan_cols(L, c, r, a0 + kk + 1, a0 + nk < cw.hi ? a0 + nk : cw.hi, f);
    scan_cols(L, c, r, a0 - nk > cw.lo ? a0 - nk : cw.lo, a0 - (kk < 0 ? 1 : kk + 1), f);
    kk = nk;
    if (kk >= half) break;
  }
}
__device__ __forceinline__ void walk_lean(const LdsStore& L, const LCloud& c, bool is_surf, int nq, float thr, int j1, int rho, float sx, float sy,
                                          float sz, float rho_q, float qn3, float el_q, float gq, float margin, int ln, int role, int lane, int warm2,
                                          int warm3, bool check_class, bool reseed, Top2& c2, Top2& c3) {
  const WalkCtx w = make_walk_ctx(c, nq, j1, rho);
  c2 = t2_init(thr);
  c3 = t2_init(thr);
  auto warm_cand = [&](Top2& b, int pos, bool on_ring_rho) {
    int rank;
    if (pos < 0) return;
    const int j = pt_idx(L, c, pos);
    if (!walk_rank(w, j, rank)) return;
    if (check_class) {
      int r = 0;  // ring of index j: the last ring that starts at or before it
#pragma unroll
      for (int step = kRingsBinned / 2; step > 0; step >>= 1)
        if (c.ring_start[r + step] <= j) r += step;
      if ((r == rho) != on_ring_rho) return;
    }
    t2_insert(b, ((unsigned long long)__float_as_uint(pt_sqdist(L, c, pos, sx, sy, sz)) << 32) | walk_low(rank, pos));
  };
  warm_cand(c2, warm2, is_surf);  // second point: ring rho for planes, another ring for lines
  warm_cand(c3, warm3, false);    // third point (planes only): another ring
  // (reseed: the warm candidates stay in the lists, but the per-ring seed scans run as in a cold walk — see nn_lean)
  const bool w2 = (unsigned)c2.k != 0u && !reseed, w3 = (unsigned)c3.k != 0u && !reseed;
  if (is_surf && !w2) {  // class-2 seed on ring rho: all lanes of the query
    const int a0 = az_col_of(gq, c.naz);
    const bool go = walk_ring_has_candidates(c, w, rho);
    auto f = [&](float x, float y, float z, int j, int p, bool ok) {
      int rank;
      const bool in_walk = walk_rank(w, j, rank);
      t2_scan(c2, ok && in_walk, sqdist3(x, y, z, sx, sy, sz), [&] { return walk_low(rank, p); });
    };
    scan_cols(L, c, rho, go ? a0 - 1 : 1, go ? a0 + 1 : 0, f);
  }
  // tasks dealt round-robin to the query's lanes:
  //   surf    t0: rho (class 2, extensions)  t1: rho-1  t2: rho+1  t3: rho-2  t4: rho+2  (class 3)
  //   corner  t0: rho-1  t1: rho+1  t2: rho-2  t3: rho+2
  // Adjacent rings before the rings two away, for every number of lanes: a lane's running best carries from task to task,
  // and on the ground the next ring is metres closer than the one after it — with rho -2 dealt before rho +1 (rounds 1-5)
  // the first of a query's two lanes swept rho -2 and rho +2 under the 5 m threshold without ever seeing an adjacent ring.
#pragma unroll 1
  for (int t = role; t <= 4; t += ln) {
    int dr;
    if (is_surf)
      dr = t == 0 ? 0 : (t == 1 ? -1 : (t == 2 ? 1 : (t == 3 ? -2 : 2)));
    else
      dr = t == 0 ? -1 : (t == 1 ? 1 : (t == 2 ? -2 : (t == 3 ? 2 : 99)));
    const bool use2 = !is_surf || dr == 0;
    const bool have_bound = use2 ? w2 : w3;  // a warm bound replaces the per-ring seed scan
    const bool seed_first = !have_bound && (!is_surf || dr != 0);
    const bool centre_done = is_surf && dr == 0 && !w2;
    // (one running best in flight: the class-2 or the class-3 one, swapped in and out by value — wave-uniform for a plane
    // wave only when every lane has the same t, which the selects below do not need)
    Top2 cur = use2 ? c2 : c3;
    walk_task_lean(L, c, w, rho + dr, seed_first, centre_done, gq, sx, sy, sz, rho_q, qn3, el_q, margin, cur);
    if (use2)
      c2 = cur;
    else
      c3 = cur;
  }
  t2_merge_lanes(c2, lane, ln);
  if (is_surf) t2_merge_lanes(c3, lane, ln);  // (wave-uniform kind: line queries have no third point)
}

// ---- the carry records of a scan: four planes of kRelayLanes 16-byte words, by query slot --------------------------------------
//   plane 0  a1 | b1c << 16,  sel1 | ra1 << 16 | rb1 << 24,  lb1,  certA.x        nearest neighbour: tracked winner / runner-up
//   plane 1  certA.y, certA.z                                                     (grid positions, -1 = none), their rings, the
//   plane 2  a2 | b2c << 16,  a3 | b3c << 16,  lb2,  lb3                          last selection, the certificate's bound and
//   plane 3  certB.x, certB.y, certB.z                                            the query position it was established at;
//                                                                                 planes 2-3: the same for second / third point
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
__device__ __forceinline__ auto carry_rsrc(int* scan_base /*wave-uniform*/) {
  return __builtin_amdgcn_make_buffer_rsrc(scan_base, 0, kRelayRegionInts * 4, 0x00020000);
}
constexpr int kCarrySc1 = 16;  // (aux / cache-policy bit sc1: coherent at device scope)
__device__ __forceinline__ unsigned pack16(int a, int b) { return ((unsigned)a & 0xFFFFu) | ((unsigned)b << 16); }
__device__ __forceinline__ int lo16(unsigned w) { return (int)(short)(w & 0xFFFFu); }  // (-1 <-> 0xFFFF: positions are < 12288)
__device__ __forceinline__ int hi16(unsigned w) { return (int)w >> 16; }

// ---- the cooperative searches of one wave on Top2 (coop_nn / coop_walk of ieskf_lds_impl.h) ---------------------------------
// The cm.n searches a wave needs are served coop_lanes() lanes each.  Only the de-skewed query and the warm candidates travel
// from the owner to its serving lanes; the polar view of the query (two square roots, an arctangent, the column — it only
// feeds pruning windows) is made BY the serving lanes from the shipped point: four registers and four shuffles less than
// shipping it, and nobody holds it across a search.  Results come back as the low words of the two keys + the bound.
struct NnRes {
  unsigned low, low2;  // winner / runner-up (nn_low: position and ring inside; 0 = none); low2 = 0xFFFFFFFF: the search was deferred
  float lb;
};
struct WalkRes {
  unsigned p2, p3;  // second / third point: winner | runner-up << 16 (grid positions, 0xFFFF = none)
  float lb2, lb3;
};
__device__ __forceinline__ unsigned t2_pos_pair(const Top2& b) {  // (walk keys)
  const unsigned w = (unsigned)b.k != 0u ? (unsigned)walk_pos((unsigned)b.k) : 0xFFFFu;
  const unsigned r = t2_real(b.k2) ? (unsigned)walk_pos((unsigned)b.k2) : 0xFFFFu;
  return w | (r << 16);
}
__device__ __forceinline__ NnRes coop_nn_lean(const LdsStore& L, const LCloud& c, const CoopMap& cm, int coop_cap, bool need_nn, int lane, float sx,
                                              float sy, float sz, int rq /* ring | reseed << 8 */, int a1, int ra1, float thr, float margin, bool skip,
                                              bool may_defer) {
  const int ln = coop_lanes(cm.n, coop_cap);
  const int role = lane & (ln - 1), item = lane >> (31 - __clz(ln));
  bool valid = need_nn;
  if (ln > 1) {  // the inputs of search `item` travel from its owner to the ln lanes that serve it
    valid = item < cm.n;
    const int owner = __shfl(cm.owner_map, valid ? item : 0);
    sx = __shfl(sx, owner), sy = __shfl(sy, owner), sz = __shfl(sz, owner);
    rq = __shfl(rq, owner), a1 = __shfl(a1, owner), ra1 = __shfl(ra1, owner);
  }
  Top2 b = t2_init(thr);
  bool deferred = false;
  if (valid && !skip) {
    const float rho = sqrtf(sx * sx + sy * sy), qn3 = sqrtf(rho * rho + sz * sz);
    b = nn_lean(L, c, sx, sy, sz, rho, qn3, atan2f(sz, rho), az_col_f(sx, sy, c.naz), thr, margin, rq & 0xFF, ln, role, lane, a1, ra1, (rq & 0x100) != 0,
                may_defer, deferred);
  }
  NnRes r{(unsigned)b.k, deferred ? 0xFFFFFFFFu : (t2_real(b.k2) ? (unsigned)b.k2 : 0u), t2_cert_lb(b, margin)};
  if (ln > 1) {  // hand back: the owner of rank r reads the first lane of group r
    const int src = (cm.rank * ln) & 63;
    r.low = __shfl(r.low, src), r.low2 = __shfl(r.low2, src), r.lb = __shfl(r.lb, src);
  }
  return r;
}
__device__ __forceinline__ WalkRes coop_walk_lean(const LdsStore& L, const LCloud& c, bool is_surf, int nq, const CoopMap& cm, int coop_cap,
                                                  bool need_walk, int lane, float sx, float sy, float sz, int j1, int rho1, int w2, int w3,
                                                  int flags /* 1: the nearest neighbour changed, 2: reseed */, float thr, float margin, bool skip) {
  const int ln = coop_lanes(cm.n, coop_cap);
  const int role = lane & (ln - 1), item = lane >> (31 - __clz(ln));
  bool valid = need_walk;
  int chk = flags;
  if (ln > 1) {
    valid = item < cm.n;
    const int owner = __shfl(cm.owner_map, valid ? item : 0);
    sx = __shfl(sx, owner), sy = __shfl(sy, owner), sz = __shfl(sz, owner);
    j1 = __shfl(j1, owner), rho1 = __shfl(rho1, owner);
    w2 = __shfl(w2, owner), w3 = __shfl(w3, owner), chk = __shfl(chk, owner);
  }
  Top2 c2 = t2_init(thr), c3 = c2;
  if (valid && !skip) {
    const float rho = sqrtf(sx * sx + sy * sy), qn3 = sqrtf(rho * rho + sz * sz);
    walk_lean(L, c, is_surf, nq, thr, j1, rho1, sx, sy, sz, rho, qn3, atan2f(sz, rho), az_col_f(sx, sy, c.naz), margin, ln, role, lane, w2, w3,
              (chk & 1) != 0, (chk & 2) != 0, c2, c3);
  }
  WalkRes r{t2_pos_pair(c2), t2_pos_pair(c3), t2_cert_lb(c2, margin), t2_cert_lb(c3, margin)};
  if (ln > 1) {
    const int src = (cm.rank * ln) & 63;
    r.p2 = __shfl(r.p2, src), r.p3 = __shfl(r.p3, src), r.lb2 = __shfl(r.lb2, src), r.lb3 = __shfl(r.lb3, src);
  }
  return r;
}
