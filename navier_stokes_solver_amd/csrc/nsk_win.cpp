// nsk_win.cpp — host-side builder of the window format (see nsk_win.hpp).  No HIP in this file.
#include "nsk_win.hpp"

#include <algorithm>

namespace nsk {

bool build_win_format(int n, const int *rp, const int *col, const int *srcpos, const std::vector<int> *cuts,
                      const int *level_of_row, int max_lines, int flags, WinFormat &out) {
  out = WinFormat{};
  out.n_rows = n;
  out.nnz = n > 0 ? (int64_t)rp[n] - rp[0] : 0;
  if (n <= 0) return true;
  int max_col = 0;
  for (int64_t k = rp[0]; k < rp[n]; ++k) max_col = std::max(max_col, col[k]);
  std::vector<int> stamp((size_t)max_col / kWinLine + 1, -1);

  // pass 1 (sequential, O(nnz)): greedy runs of consecutive rows and their sorted line lists
  size_t ci = 0;
  int r0 = 0;
  int64_t pairs = 0;
  std::vector<int> cur;
  while (r0 < n) {
    while (cuts && ci < cuts->size() && (*cuts)[ci] <= r0) ++ci;
    const int limit = (cuts && ci < cuts->size()) ? std::min(n, (*cuts)[ci]) : n;
    const int id = (int)out.runs.size();
    cur.clear();
    int r1 = r0;
    while (r1 < limit && r1 - r0 < kWinMaxRows && rp[r1 + 1] - rp[r0] <= kWinRunNnz) {
      const size_t mark = cur.size();
      for (int k = rp[r1]; k < rp[r1 + 1]; ++k) {
        const int ln = col[k] / kWinLine;
        if (stamp[ln] != id) { stamp[ln] = id; cur.push_back(ln); }
      }
      if ((int)cur.size() > max_lines) {
        for (size_t q = mark; q < cur.size(); ++q) stamp[cur[q]] = -1;
        cur.resize(mark);
        break;
      }
      ++r1;
    }
    if (r1 == r0) return false;  // one row alone exceeds the window or the run size
    std::sort(cur.begin(), cur.end());
    const int N = rp[r1] - rp[r0];
    WinRun R{};
    R.r0 = r0;
    R.nrows = r1 - r0;
    R.l0 = (int)out.lines.size();
    R.nl = (int)cur.size();
    R.q2 = (N + 2 * kWinThreads - 1) / (2 * kWinThreads);
    if (pairs + (int64_t)R.q2 * kWinThreads > (int64_t)0x3fffffff) return false;  // 32-bit pair index
    R.p0 = (int)pairs;
    R.roff0 = (int)out.roff.size();
    R.flags = flags | ((level_of_row ? level_of_row[r0] : 0) << 8);
    for (int r = r0; r <= r1; ++r) out.roff.push_back((uint16_t)(rp[r] - rp[r0]));
    out.lines.insert(out.lines.end(), cur.begin(), cur.end());
    out.runs.push_back(R);
    pairs += (int64_t)R.q2 * kWinThreads;
    r0 = r1;
  }
  out.n_slots = 2 * pairs;
  out.pos.assign((size_t)out.n_slots, 0);
  out.src.assign((size_t)out.n_slots, -1);

  // pass 2 (parallel over runs): window positions and source positions in the transposed slot order
#pragma omp parallel for schedule(dynamic, 64)
  for (long b = 0; b < (long)out.runs.size(); ++b) {
    const WinRun &R = out.runs[(size_t)b];
    const int *ln = out.lines.data() + R.l0;
    const int k0 = rp[R.r0], N = rp[R.r0 + R.nrows] - k0, per = 2 * R.q2;
    for (int e = 0; e < N; ++e) {
      const int t = e / per, i = e % per;
      const size_t slot = 2 * ((size_t)R.p0 + (size_t)(i / 2) * kWinThreads + (size_t)t) + (size_t)(i & 1);
      const int c = col[k0 + e], line = c / kWinLine;
      const int wl = (int)(std::lower_bound(ln, ln + R.nl, line) - ln);
      out.pos[slot] = (uint16_t)(wl * kWinLine + c % kWinLine);
      out.src[slot] = srcpos ? srcpos[k0 + e] : k0 + e;
    }
  }
  return true;
}

void win_append(WinFormat &a, const WinFormat &b) {
  const int64_t pair_shift = a.n_slots / 2;
  const int line_shift = (int)a.lines.size(), roff_shift = (int)a.roff.size();
  for (WinRun R : b.runs) {
    R.p0 += (int)pair_shift;
    R.l0 += line_shift;
    R.roff0 += roff_shift;
    a.runs.push_back(R);
  }
  a.lines.insert(a.lines.end(), b.lines.begin(), b.lines.end());
  a.roff.insert(a.roff.end(), b.roff.begin(), b.roff.end());
  a.pos.insert(a.pos.end(), b.pos.begin(), b.pos.end());
  a.src.insert(a.src.end(), b.src.begin(), b.src.end());
  a.n_slots += b.n_slots;
  a.nnz += b.nnz;
}

}  // namespace nsk
