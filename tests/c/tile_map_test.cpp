// Host-side check of conv_fwd_kernel's blockIdx -> logical tile map (cv-vae_amd/csrc/tile_map.h): for every grid shape the
// host can launch it must be a bijection of [0, nwg); with short tiles, every XCD must run all its long tiles before its
// short ones.  Built with g++ by tests/test_tile_map.py (no GPU, no HIP).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tile_map.h"

static int check(int nsp, int tiles_t, int inner, int lo, int hi) {
  const int nwg = nsp * tiles_t * inner;
  std::vector<char> seen(nwg, 0);
  std::vector<int> last_long(8, -1), first_short(8, 1 << 30);
  for (int bid = 0; bid < nwg; ++bid) {
    const int l = cvvae::logical_tile_of_block(nwg, bid, inner, tiles_t, lo, hi);
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

int main() {
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
  // the grids of BASELINE config 3 / 4 layers
  const int big[][5] = {{1024, 8, 1, 1, 0}, {256, 9, 1, 2, 0}, {256, 9, 8, 1, 1}, {64, 9, 2, 2, 0}, {16, 5, 2, 1, 1}, {1188, 8, 1, 1, 0}};
  for (auto& g : big) {
    if (check(g[0], g[1], g[2], g[3], g[4])) return 1;
    ++cases;
  }
  std::printf("tile map ok: %ld grids\n", cases);
  return 0;
}
