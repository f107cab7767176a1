// nsk_win_capi.cpp — host-only inspection hooks of the window format (no GPU needed): the CPU test-suite
// decodes the format against the CSR it was built from, and scripts/win_stats.py reads its byte counts.
#include <algorithm>
#include <cstring>

#include "../../include/nsk.h"
#include "nsk_tri.hpp"
#include "nsk_win.hpp"

using namespace nsk;

struct nsk_host_win_s {
  WinFormat W;
  std::vector<int> perm;  // perm[new] = old (identity for part 0 / natural ordering)
  int n_colors = 0;
};

extern "C" {

// part 0: all rows in the given order (SpMV); 1 / 2: strict lower / upper triangle of P A P^T with P the greedy
// multicolour permutation (ordering 1) or the identity (ordering 0), columns in the permuted numbering
void *nsk_host_win_create(int n, const int32_t *rp, const int32_t *col, int ordering, int part, int max_lines) {
  auto *H = new nsk_host_win_s();
  H->perm.resize((size_t)n);
  for (int i = 0; i < n; ++i) H->perm[(size_t)i] = i;
  if (max_lines <= 0) max_lines = kWinMaxLines;
  bool ok;
  if (part == 0) {
    ok = build_win_format(n, rp, col, nullptr, nullptr, nullptr, max_lines, 0, 0, H->W);
  } else {
    std::vector<int> color((size_t)n, 0), cuts;
    if (ordering == ORDER_MULTICOLOR) {
      std::vector<int> grp(rp, rp + n + 1), gcol(col, col + rp[n]);
      H->n_colors = greedy_color(n, grp, gcol, color);
      std::vector<int> cptr((size_t)H->n_colors + 1, 0);
      for (int i = 0; i < n; ++i) ++cptr[(size_t)color[(size_t)i] + 1];
      for (int q = 0; q < H->n_colors; ++q) cptr[(size_t)q + 1] += cptr[(size_t)q];
      for (int q = 1; q < H->n_colors; ++q) cuts.push_back(cptr[(size_t)q]);
      for (int i = 0; i < n; ++i) H->perm[(size_t)cptr[(size_t)color[(size_t)i]]++] = i;
    }
    std::vector<int> iperm((size_t)n), pcolor((size_t)n);
    for (int i = 0; i < n; ++i) { iperm[(size_t)H->perm[(size_t)i]] = i; pcolor[(size_t)i] = color[(size_t)H->perm[(size_t)i]]; }
    std::vector<int> hrp((size_t)n + 1, 0), hcol, hsrc;
    std::vector<std::pair<int, int>> ent;
    for (int i = 0; i < n; ++i) {
      const int r = H->perm[(size_t)i];
      ent.clear();
      for (int k = rp[r]; k < rp[r + 1]; ++k) {
        const int c = col[k] < n ? iperm[(size_t)col[k]] : -1;
        if (c < 0 || (part == 1 ? c >= i : c <= i)) continue;
        ent.emplace_back(c, k);
      }
      std::sort(ent.begin(), ent.end());
      for (auto &e : ent) { hcol.push_back(e.first); hsrc.push_back(e.second); }
      hrp[(size_t)i + 1] = (int)hcol.size();
    }
    ok = build_win_format(n, hrp.data(), hcol.data(), hsrc.data(), cuts.empty() ? nullptr : &cuts,
                          ordering == ORDER_MULTICOLOR ? pcolor.data() : nullptr, max_lines, part == 2 ? 1 : 0, 0, H->W);
  }
  if (!ok) { delete H; return nullptr; }
  return H;
}

// out[0..7] = n_runs, n_lines, n_slots, nnz, n_colors, bytes per pass, n_rows, n_roff
void nsk_host_win_sizes(void *h, double *out) {
  auto *H = (nsk_host_win_s *)h;
  out[0] = (double)H->W.runs.size(); out[1] = (double)H->W.lines.size(); out[2] = (double)H->W.n_slots;
  out[3] = (double)H->W.nnz; out[4] = H->n_colors; out[5] = H->W.bytes_per_apply(); out[6] = H->W.n_rows;
  out[7] = (double)H->W.roff.size();
}

void nsk_host_win_get(void *h, int32_t *runs8, int32_t *lines, uint16_t *roff, uint16_t *pos, int32_t *src, int32_t *perm) {
  auto *H = (nsk_host_win_s *)h;
  static_assert(sizeof(WinRun) == 32, "WinRun is 8 ints");
  if (runs8) std::memcpy(runs8, H->W.runs.data(), sizeof(WinRun) * H->W.runs.size());
  if (lines) std::memcpy(lines, H->W.lines.data(), sizeof(int) * H->W.lines.size());
  if (roff) std::memcpy(roff, H->W.roff.data(), sizeof(uint16_t) * H->W.roff.size());
  if (pos) std::memcpy(pos, H->W.pos.data(), sizeof(uint16_t) * H->W.pos.size());
  if (src) std::memcpy(src, H->W.src.data(), sizeof(int) * H->W.src.size());
  if (perm) std::memcpy(perm, H->perm.data(), sizeof(int) * H->perm.size());
}

void nsk_host_win_free(void *h) { delete (nsk_host_win_s *)h; }

}  // extern "C"
