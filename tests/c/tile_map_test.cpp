// Host-side checks of cvvae_amd/csrc/tile_map.h: (1) conv_fwd_kernel's blockIdx -> logical tile map must be a bijection of
// [0, nwg) for every grid shape the host can launch, and with short tiles every XCD must run all its long tiles before its
// short ones; (2) the per-frame time-fold plan must reproduce the 3-tap sum under both padding modes.  Built with g++ by
// tests/test_tile_map.py (no GPU, no HIP).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tile_map.h"

static int check(int nsp, int tiles_t, int inner, int lo, int hi, int win = 0) {
  const int nwg = nsp * tiles_t * inner;
  std::vector<char> seen(nwg, 0);
  std::vector<int> last_long(8, -1), first_short(8, 1 << 30);
  for (int bid = 0; bid < nwg; ++bid) {
    const int l = cvvae::logical_tile_of_block(nwg, bid, inner, tiles_t, lo, hi, win);
    if (l < 0 || l >= nwg || seen[l]) {
      std::printf("not a bijection: nsp %d tiles_t %d inner %d lo %d hi %d: bid %d -> %d\n", nsp, tiles_t, inner, lo, hi, bid, l);
      return 1;
    }
    seen[l] = 1;
    const int tt = (l / inner) % tiles_t;
    const bool is_short = tt < lo || tt >= tiles_t - hi;
    const int xcd = bid & 7, j = bid >> 3;
    if (is_short) {
      if (j < first_short[xcd]) first_short[xcd] = j;
    } else if (j > last_long[xcd]) {
      last_long[xcd] = j;
    }
  }
  if (lo + hi > 0)
    for (int x = 0; x < 8; ++x)
      if (last_long[x] > first_short[x]) {
        std::printf("XCD %d runs a long tile after a short one: nsp %d tiles_t %d inner %d lo %d hi %d\n", x, nsp, tiles_t, inner, lo, hi);
        return 1;
      }
  return 0;
}

// The time-fold plan of an output frame must reproduce the 3-tap sum: for every stored frame s, the total weight the plan
// applies to s (slot 3 = W0+W1, 4 = W1+W2, 5 = W0+W1+W2) must equal the weight the taps apply to it under the padding mode.
static int check_time_folds() {
  long n = 0;
  for (int Tl = 1; Tl <= 9; ++Tl)
    for (int ST = 1; ST <= 2; ++ST)
      for (int pad = 0; pad <= 2; ++pad)
        for (int rep = 0; rep <= 1; ++rep)
          for (int sums = 0; sums <= 1; ++sums)
            for (int to = 0; to * ST - pad + 2 - (2 - pad) < Tl + 2 && to < Tl + 2; ++to) {
              const int f0 = to * ST - pad;
              if (!rep && (f0 + 1 < 0 || f0 + 1 >= Tl)) continue;  // zero padding never pads by more than one frame per side
              const cvvae::TimeFoldPlan q = cvvae::time_fold_plan(to, ST, pad, Tl, rep != 0, sums != 0);
              // expected[s][k]: coefficient of W_k on stored frame s
              int expect[16][3] = {}, got[16][3] = {};
              for (int dt = 0; dt < 3; ++dt) {
                int s = f0 + dt;
                if (rep) s = s < 0 ? 0 : (s >= Tl ? Tl - 1 : s);
                else if (s < 0 || s >= Tl) continue;
                expect[s][dt] += 1;
              }
              const int slots[3] = {q.slot0, q.slot1, 2}, dts[3] = {q.dt0, q.dt1, 2};
              for (int g = 0; g < q.ng; ++g) {
                int s = f0 + dts[g];  // the LDS halo frame holds the padded view: clamp / zero as the staging does
                if (rep) s = s < 0 ? 0 : (s >= Tl ? Tl - 1 : s);
                else if (s < 0 || s >= Tl) continue;
                const int sl = slots[g];
                if (sl >= 3 && !sums) {
                  std::printf("summed slot without summed weights\n");
                  return 1;
                }
                const bool w0 = sl == 0 || sl == 3 || sl == 5, w1 = sl == 1 || sl == 3 || sl == 4 || sl == 5,
                           w2 = sl == 2 || sl == 4 || sl == 5;
                got[s][0] += w0; got[s][1] += w1; got[s][2] += w2;
              }
              for (int s = 0; s < Tl; ++s)
                for (int k = 0; k < 3; ++k)
                  if (expect[s][k] != got[s][k]) {
                    std::printf("time-fold plan wrong: Tl %d ST %d pad %d rep %d sums %d to %d (ng %d slots %d %d dts %d %d): frame %d W%d "
                                "expect %d got %d\n", Tl, ST, pad, rep, sums, to, q.ng, q.slot0, q.slot1, q.dt0, q.dt1, s, k,
                                expect[s][k], got[s][k]);
                    return 1;
                  }
              ++n;
            }
  std::printf("time-fold plan ok: %ld frames\n", n);
  return 0;
}

int main() {
  if (check_time_folds()) return 1;
  long cases = 0;
  // plain map (no short tiles): any grid
  for (int nwg = 1; nwg <= 600; ++nwg) {
    if (check(nwg, 1, 1, 0, 0)) return 1;
    ++cases;
  }
  // short tiles last: the host enables it only when 0 < lo + hi < tiles_t and part 2 holds at least 8 tiles
  const int inners[] = {1, 2, 4, 8, 16};
  for (int tiles_t = 2; tiles_t <= 17; ++tiles_t)
    for (int lo = 0; lo <= 2; ++lo)
      for (int hi = 0; hi <= 2; ++hi) {
        if (lo + hi == 0 || lo + hi >= tiles_t) continue;
        for (int inner : inners)
          for (int nsp = 1; nsp <= 70; ++nsp) {
            if ((long)nsp * (lo + hi) * inner < 8) continue;
            if (check(nsp, tiles_t, inner, lo, hi)) return 1;
            ++cases;
          }
      }
  // weight-stationary windows (folded upsample convs): every window size against every grid, with and without short tiles;
  // and inside a full window the pixel tile must be the fastest index
  for (int win : {2, 4, 8, 16})
    for (int inner : {4, 8, 16})
      for (int tiles_t = 1; tiles_t <= 9; ++tiles_t)
        for (int nsp = 1; nsp <= 40; ++nsp) {
          if (check(nsp, tiles_t, inner, 0, 0, win)) return 1;
          ++cases;
          for (int lo = 0; lo <= 1; ++lo)
            for (int hi = 0; hi <= 1; ++hi) {
              if (lo + hi == 0 || lo + hi >= tiles_t || (long)nsp * (lo + hi) * inner < 8) continue;
              if (check(nsp, tiles_t, inner, lo, hi, win)) return 1;
              ++cases;
            }
        }
  {
    const int inner = 16, win = 8, nwg = 64 * inner;  // one XCD's run starts at a window boundary: positions 0..7 = 8 pixel tiles of set 0
    for (int k = 0; k < 8; ++k) {
      const int l = cvvae::logical_tile_of_block(nwg, k * 8, inner, 1, 0, 0, win);  // XCD 0, j = k
      if (l % inner != 0 || l / inner != k) {
        std::printf("window order wrong: j %d -> pixel tile %d set %d\n", k, l / inner, l % inner);
        return 1;
      }
    }
  }
  // the grids of BASELINE config 3 / 4 layers
  const int big[][5] = {{1024, 8, 1, 1, 0}, {256, 9, 1, 2, 0}, {256, 9, 8, 1, 1}, {64, 9, 2, 2, 0}, {16, 5, 2, 1, 1}, {1188, 8, 1, 1, 0}};
  for (auto& g : big) {
    if (check(g[0], g[1], g[2], g[3], g[4])) return 1;
    ++cases;
  }
  std::printf("tile map ok: %ld grids\n", cases);
  return 0;
}
