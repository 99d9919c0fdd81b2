// Host-side check of the index arithmetic the on-chip smoother kernels rely on (csrc/smooth_onchip.h, csrc/smooth_mfma.h):
// compiled with `hipcc --cuda-host-only` and run on the CPU by tests/test_onchip_layout.py.  Prints "ok" or the first violation.
#include <cstdio>
#include <set>
#include <utility>
#include <vector>
#include "smooth_onchip.h"
#include "smooth_mfma.h"
using namespace odef;

template <int DPB>
static bool check_products() {
  using Pr = oc::Products<DPB>;
  // every unordered pair of tile columns (the diagonal included) has exactly one owner
  std::set<std::pair<int, int>> seen;
  for (int c = 0; c < DPB; ++c)
    for (int w = 0; w < Pr::owned(c); ++w) {
      const int cp = (c + w) % DPB;
      const auto key = std::make_pair(c < cp ? c : cp, c < cp ? cp : c);
      if (!seen.insert(key).second) { printf("DPB %d: pair (%d, %d) owned twice\n", DPB, key.first, key.second); return false; }
      if (w >= Pr::WMAX) { printf("DPB %d: more tiles than accumulators\n", DPB); return false; }
    }
  if ((int)seen.size() != DPB * (DPB + 1) / 2) { printf("DPB %d: %zu pairs owned, %d exist\n", DPB, seen.size(), DPB * (DPB + 1) / 2); return false; }
  // the swizzle is a permutation of the tile ...
  std::set<int> slots;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) slots.insert(Pr::sw(r, c));
  if (slots.size() != 256 || *slots.begin() != 0 || *slots.rbegin() != 255) { printf("swizzle is not a permutation of the tile\n"); return false; }
  // ... under which both fragment reads are bank-conflict free: a ds_read_b64 serves lanes {0..31} and {32..63} in turn, 64
  // banks of 4 bytes = 32 slots of 8 bytes
  for (int kk = 0; kk < 4; ++kk)
    for (int half = 0; half < 2; ++half)
      for (int transposed = 0; transposed < 2; ++transposed) {
        std::set<int> banks;
        for (int l = 32 * half; l < 32 * half + 32; ++l) {
          const int a = transposed ? Pr::sw(4 * kk + (l >> 4), l & 15) : Pr::sw(l & 15, 4 * kk + (l >> 4));
          banks.insert(a % 32);
        }
        if (banks.size() != 32) { printf("bank conflict: k-step %d, half %d, transposed %d: %zu distinct slots\n", kk, half, transposed, banks.size()); return false; }
      }
  // upper-tile index
  std::set<int> tiles;
  for (int j = 0; j < DPB; ++j)
    for (int jp = j; jp < DPB; ++jp) tiles.insert(Pr::tix(j, jp));
  if ((int)tiles.size() != Pr::NTU || *tiles.rbegin() != Pr::NTU - 1) { printf("DPB %d: tix is not a numbering of the upper tiles\n", DPB); return false; }
  // the LDS of one workgroup holds M, the row buffer(s) and three vectors
  if ((Pr::size + 3 * DPB * 16) * 8 > 160 * 1024) { printf("DPB %d: products do not fit the LDS\n", DPB); return false; }
  return true;
}

template <int d, int NB>
static bool check_tile_major() {
  using W = MfmaSmoothWs<d, NB>;
  std::vector<char> hit((size_t)W::MAT, 0);
  for (int r = 0; r < W::DP; ++r)
    for (int c = 0; c < W::DP; ++c) {
      const int a = W::tm(r, c);
      if (a < 0 || a >= W::MAT || hit[a]) { printf("d %d NB %d: tm is not a bijection at (%d, %d)\n", d, NB, r, c); return false; }
      hit[a] = 1;
      if (a != W::tile_at(r >> 4, c >> 4) + (r & 15) * 16 + (c & 15)) { printf("tile_at disagrees with tm\n"); return false; }
    }
  // the tiles of a tile column are contiguous, in row order
  for (int tc = 0; tc < W::DPB; ++tc)
    for (int tr = 0; tr + 1 < W::DPB; ++tr)
      if (W::tile_at(tr + 1, tc) != W::tile_at(tr, tc) + 256) { printf("tiles of a column are not contiguous\n"); return false; }
  return true;
}

int main() {
  bool ok = check_products<2>() && check_products<3>() && check_products<4>() && check_products<6>() && check_products<7>() && check_products<9>() &&
            check_products<10>() && check_products<11>();
  ok = ok && check_tile_major<28, 6>() && check_tile_major<28, 2>() && check_tile_major<16, 4>() && check_tile_major<12, 3>() && check_tile_major<4, 6>();
  if (ok) printf("ok\n");
  return ok ? 0 : 1;
}
