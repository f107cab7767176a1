// nsk_win.cpp — host-side builder of the window format (see nsk_win.hpp).  No HIP in this file.
#include "nsk_win.hpp"

#include <algorithm>

namespace nsk {

bool build_win_format(int n, const int *rp, const int *col, const int *srcpos, const std::vector<int> *cuts,
                      const int *level_of_row, int max_lines, int flags, int balanced, WinFormat &out) {
  out = WinFormat{};
  out.n_rows = n;
  out.nnz = n > 0 ? (int64_t)rp[n] - rp[0] : 0;
  if (n <= 0) return true;
  int max_col = 0;
  for (int64_t k = rp[0]; k < rp[n]; ++k) max_col = std::max(max_col, col[k]);
  std::vector<int> stamp((size_t)max_col / kWinLine + 1, -1);

  // pass 1 (sequential, O(nnz) per attempt): runs of consecutive rows and their sorted line lists.  A chunk (the
  // rows up to the next cut) is split into k parts of about equal non-zero count, with the smallest k for which
  // every part respects the caps; without cuts the whole matrix is one chunk and is cut greedily.
  size_t ci = 0;
  int r0 = 0;
  int64_t pairs = 0;
  std::vector<int> cur, ends;
  int stamp_id = 0;
  auto fits = [&](int a, int b) {   // rows [a, b): within the row, non-zero and window caps?
    if (b - a > kWinMaxRows || rp[b] - rp[a] > kWinRunNnz) return false;
    ++stamp_id;
    int cnt = 0;
    for (int k = rp[a]; k < rp[b]; ++k) {
      const int ln = col[k] / kWinLine;
      if (stamp[ln] != stamp_id) { stamp[ln] = stamp_id; if (++cnt > max_lines) return false; }
    }
    return true;
  };
  auto emit = [&](int a, int b) {
    ++stamp_id;
    cur.clear();
    for (int k = rp[a]; k < rp[b]; ++k) {
      const int ln = col[k] / kWinLine;
      if (stamp[ln] != stamp_id) { stamp[ln] = stamp_id; cur.push_back(ln); }
    }
    std::sort(cur.begin(), cur.end());
    const int N = rp[b] - rp[a];
    WinRun R{};
    R.r0 = a;
    R.nrows = b - a;
    R.l0 = (int)out.lines.size();
    R.nl = (int)cur.size();
    R.q2 = (N + 2 * kWinThreads - 1) / (2 * kWinThreads);
    R.p0 = (int)pairs;
    R.roff0 = (int)out.roff.size();
    R.flags = (flags & 1) | (N << 1) | ((level_of_row ? level_of_row[a] : 0) << 16);
    for (int r = a; r <= b; ++r) out.roff.push_back((uint16_t)(rp[r] - rp[a]));
    out.lines.insert(out.lines.end(), cur.begin(), cur.end());
    out.runs.push_back(R);
    pairs += (int64_t)R.q2 * kWinThreads;
  };
  while (r0 < n) {
    while (cuts && ci < cuts->size() && (*cuts)[ci] <= r0) ++ci;
    const int limit = (cuts && ci < cuts->size()) ? std::min(n, (*cuts)[ci]) : n;
    if (!cuts || !balanced) {   // greedy: as many rows as the caps allow
      int r1 = r0 + 1;
      if (!fits(r0, r1)) return false;   // one row alone exceeds the window or the run size
      // (grow by doubling, then bisect: fits() is O(non-zeros of the candidate))
      int step = 8;
      while (r1 < limit && fits(r0, std::min(limit, r1 + step))) { r1 = std::min(limit, r1 + step); step *= 2; }
      while (step > 1) {
        step /= 2;
        if (r1 + step <= limit && fits(r0, r1 + step)) r1 += step;
      }
      emit(r0, r1);
      r0 = r1;
    } else {       // balanced: k parts of about equal weight (non-zeros + 8 per row)
      bool done = false;
      auto wt = [&](int r) { return (int64_t)rp[r] + 8 * (int64_t)r; };
      const int64_t total = wt(limit) - wt(r0);
      const int kmin = (int)std::max<int64_t>(1, std::max<int64_t>(((int64_t)rp[limit] - rp[r0] + kWinRunNnz - 1) / kWinRunNnz,
                                                                  (limit - r0 + kWinMaxRows - 1) / kWinMaxRows));
      for (int k = kmin; k <= limit - r0 && !done; ++k) {
        ends.clear();
        int a = r0;
        bool ok = true;
        for (int part = 1; part <= k && ok; ++part) {
          int b = limit;
          if (part < k) {
            const int64_t target = wt(r0) + total * part / k;
            int lo = a + 1, hi = limit - (k - part);   // first b with wt(b) >= target, leaving a row for every later part
            while (lo < hi) {
              const int mid = (lo + hi) / 2;
              if (wt(mid) < target) lo = mid + 1; else hi = mid;
            }
            b = lo;
          }
          ok = fits(a, b);
          ends.push_back(b);
          a = b;
        }
        if (ok) {
          a = r0;
          for (int b : ends) { emit(a, b); a = b; }
          done = true;
        }
      }
      if (!done) return false;   // one row alone exceeds the window or the run size
      r0 = limit;
    }
    if (pairs > (int64_t)0x3fffffff) return false;   // 32-bit pair index
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
