// tile_map.h -- index logic of conv_fwd_kernel that is plain C++ (no HIP types), so that it is testable on the host
// (tests/c/tile_map_test.cpp, tests/test_tile_map.py): the blockIdx -> logical tile map (a BIJECTION of [0, nwg) for every grid
// the host can launch) and the per-frame time-fold plan.
#pragma once
#if defined(__HIPCC__)
#define CVVAE_HD __host__ __device__ __forceinline__
#else
#define CVVAE_HD inline
#endif

namespace cvvae {

// Logical tile numbering: ((spatial tile * tiles_t + time tile) * phases + phase) * ntiles_n + ntile, i.e. N-tile (and upsample
// phase) fastest -- the `inner` = phases * ntiles_n workgroups of one pixel tile re-use its halo from L2 -- then TIME, then x, y,
// batch.  Hardware deals workgroups to the 8 XCDs round-robin (bid % 8), each XCD with its own L2:
//   * XCD-aware remap: XCD x owns a contiguous run of logical tiles, so neighbouring halo tiles share one L2;
//   * short tiles last (short_lo / short_hi leading / trailing time tiles of every spatial tile are SHORT: their frames take a
//     time fold, see conv_kernel.h): the logical space is split in PART 1 = (spatial tile, long time tile) and PART 2 =
//     (spatial tile, short time tile), each in the usual order; every XCD runs a contiguous share of part 1, then a contiguous
//     share of part 2 (longest-processing-time-first: the launch tail is bounded by a short tile).  The XCD's workgroup count
//     is fixed by the hardware, so its part-2 share is what remains after its part-1 share.  Requires part 2 to hold >= 8
//     tiles (the host checks), which keeps every XCD's part-1 share within its workgroup count.
//   * weight-stationary windows (win > 1; the folded upsample convs, whose `inner` = 4 phases x N-tiles weight sets of 1.5-3 MB each
//     do not fit an XCD's 4 MiB L2 together): inside every run of `win` consecutive pixel tiles the PIXEL TILE is the fastest
//     index and the weight set the next one -- the ~32 workgroups an XCD runs at a time cover `win` pixel tiles x 32/win weight
//     sets instead of 2 x 16, so a weight set is fetched once per `win` pixel tiles instead of once per pixel tile, while the
//     window's halos (win x ~0.35 MB) stay in L2.  split() turns a position inside a part into (pixel tile v, weight set w).
CVVAE_HD void tile_split(int idx, int inner, int nv, int win, int& v, int& w) {
  if (win <= 1) {
    w = idx % inner;
    v = idx / inner;
    return;
  }
  const int g = idx / (win * inner), rr = idx - g * (win * inner);
  const int first = g * win;
  const int wc = nv - first < win ? nv - first : win;  // (the last window of a part may be short)
  v = first + rr % wc;
  w = rr / wc;
}

CVVAE_HD int logical_tile_of_block(int nwg, int bid, int inner, int tiles_t, int short_lo, int short_hi, int win = 0) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int pre = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;  // logical tiles of the XCDs before mine
  const int j = bid >> 3;                                                 // my index inside my XCD's run
  const int ns = short_lo + short_hi;
  if (ns <= 0) {
    if (win <= 1) return pre + j;
    int v, w;
    tile_split(pre + j, inner, nwg / inner, win, v, w);
    return v * inner + w;
  }
  const int nsp = nwg / (inner * tiles_t);  // spatial tiles x batch
  const int tl = tiles_t - ns;
  const int size1 = nsp * tl * inner;
  const int q1 = size1 >> 3, r1 = size1 & 7;
  const int n1 = q1 + (xcd < r1 ? 1 : 0);
  const int pre1 = xcd < r1 ? xcd * (q1 + 1) : r1 * (q1 + 1) + (xcd - r1) * q1;
  const bool part1 = j < n1;
  const int idx = part1 ? pre1 + j : (pre - pre1) + (j - n1);
  const int per = part1 ? tl : ns;
  int w, v;
  tile_split(idx, inner, nsp * per, win, v, w);
  const int ts = v % per, sp = v / per;
  const int tt = part1 ? short_lo + ts : (ts < short_lo ? ts : tiles_t - short_hi + (ts - short_lo));
  return (sp * tiles_t + tt) * inner + w;
}

// Time-fold plan of ONE output frame `to` of a 3-tap time kernel (conv_kernel.h, "time folds"): input frames f0 + {0,1,2},
// f0 = to*ST - pad_front, mapped into [0, Tl) by the padding mode.  The frame's taps are walked as `ng` TIME GROUPS; group g
// multiplies LDS halo frame dt_g (relative to the frame's first tap) with weight slot slot_g, where slots 0,1,2 = W0,W1,W2 and
// 3,4,5 = W0+W1, W1+W2, W0+W1+W2 (cvvae_pack_weights_tfolds; `has_sum_slots`).  Group 2, when present, is always (W2, 2).
//   replicate padding: taps that read the same stored frame are merged (needs the summed slots);
//   zero padding: taps on a padding frame are dropped (exact; needs no extra slots).
// Host-tested: the plan reproduces  sum_dt W_dt * frame(f0 + dt)  for every (Tl, stride, pad, to)  (tests/c/tile_map_test.cpp).
struct TimeFoldPlan {
  int ng, slot0, slot1, dt0, dt1;
};
CVVAE_HD TimeFoldPlan time_fold_plan(int to, int ST, int pad_front, int Tl, bool replicate, bool has_sum_slots) {
  const int f0 = to * ST - pad_front;
  TimeFoldPlan q = {3, 0, 1, 0, 1};
  if (replicate) {
    if (has_sum_slots) {
      const int c0 = f0 < 0 ? 0 : (f0 >= Tl ? Tl - 1 : f0);
      const int c1 = f0 + 1 < 0 ? 0 : (f0 + 1 >= Tl ? Tl - 1 : f0 + 1);
      const int c2 = f0 + 2 < 0 ? 0 : (f0 + 2 >= Tl ? Tl - 1 : f0 + 2);
      const bool eq01 = c0 == c1, eq12 = c1 == c2;
      if (eq01 && eq12) q = {1, 5, 1, 1, 1};
      else if (eq01) q = {2, 3, 2, 1, 2};
      else if (eq12) q = {2, 0, 4, 0, 1};
    }
  } else {
    const bool z0 = f0 < 0 || f0 >= Tl, z2 = f0 + 2 < 0 || f0 + 2 >= Tl;
    if (z0 && z2) q = {1, 1, 1, 1, 1};
    else if (z0) q = {2, 1, 2, 1, 2};
    else if (z2) q = {2, 0, 1, 0, 1};
  }
  return q;
}

}  // namespace cvvae
