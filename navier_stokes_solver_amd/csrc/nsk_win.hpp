// nsk_win.hpp — "window" storage of a sparse block for the gfx950 kernels (host side, GPU-free).
//
// The CSR kernels of round 1 are bound by the address unit of the CU, not by HBM: every non-zero costs a
// 4-byte column load, an 8-byte value load and a 64-address gather of x (rocprofv3: TA busy 80-100 %).
// The window format removes the gathers and narrows the index stream:
//   * rows are cut into RUNS of consecutive whole rows holding at most kWinRunNnz non-zeros (one
//     256-thread workgroup per run);
//   * a run lists the distinct 128-byte LINES (16 doubles) of the gathered vector its rows touch; the
//     workgroup copies those lines into LDS with wide coalesced loads ("LDS-staged column tiles");
//   * a non-zero stores a 16-bit window position (slot * 16 + offset in the line) instead of a 32-bit
//     column: 10 B per non-zero instead of 12;
//   * the non-zeros of a run are dealt to the threads in consecutive chunks of 2 * q2 entries (q2 = pairs
//     per thread, uniform in the run: no per-row padding, only the tail of the run is padded), but STORED
//     transposed: pair j of thread t sits at pair index p0 + j * 256 + t, so that every wave loads 1 KB of
//     contiguous values (16 B per lane) per instruction.
// Padding slots hold value 0, position 0 and source -1.
#pragma once
#include <cstdint>
#include <vector>

namespace nsk {

constexpr int kWinThreads = 256;      // workgroup size of the window kernels
constexpr int kWinLine = 16;          // doubles per window line (128 B)
constexpr int kWinMaxLines = 160;     // lines per run the kernels reserve LDS for (20 KB)
constexpr int kWinMaxQ2 = 4;          // pairs per thread
constexpr int kWinRunNnz = 2 * kWinMaxQ2 * kWinThreads;  // 2048 non-zeros per run
constexpr int kWinMaxRows = 256;      // rows per run

struct WinRun {      // 8 ints per run, read by the kernels as two int4
  int r0, nrows;     // rows [r0, r0 + nrows) of the row order the format was built in
  int l0, nl;        // window lines: lines[l0 .. l0 + nl)
  int p0, q2;        // first pair and pairs per thread: entry e of the run lives in thread t = e / (2 q2) as its
                     // i = e % (2 q2)-th entry, in slot 2 * (p0 + (i / 2) * 256 + t) + i % 2
  int roff0;         // roff[roff0 + k], k = 0..nrows: entry offsets of the run's rows inside the run
  int flags;         // bit 0: upper half (triangular solves); bits 1-12: entries of the run; bits 16..: colour / level
};

struct WinFormat {
  int n_rows = 0;
  int64_t nnz = 0;                  // entries of the source pattern
  int64_t n_slots = 0;              // padded entry slots (= 2 * pairs)
  std::vector<WinRun> runs;
  std::vector<int> lines;           // line ids (index of the gathered vector / 16)
  std::vector<uint16_t> roff;       // per run nrows + 1 local row offsets
  std::vector<uint16_t> pos;        // per slot: window position (slot * 16 + offset), 0 for padding
  std::vector<int> src;             // per slot: position in the source value array, -1 for padding
  double bytes_per_apply() const {  // what one pass over the format reads: values + positions + lines + descriptors + row offsets
    return 10.0 * (double)n_slots + 4.0 * (double)lines.size() + 32.0 * (double)runs.size() + 2.0 * (double)roff.size();
  }
};

// Build the format for rows [0, n) given as CSR (rp, col; srcpos may be null = identity).
//   cuts          ascending row ids no run may cross (may be null)
//   balanced      1: the rows between two cuts (a CHUNK) become as few runs as the caps allow, of about equal size (a
//                 workgroup that owns a chunk solves its runs back to back, so the longest chunk sets the pace);
//                 0: greedy, every run as long as the caps allow
//   level_of_row  stored in the runs' flags (may be null)
// Returns false when a single row touches more than max_lines lines or holds more than kWinRunNnz entries.
bool build_win_format(int n, const int *rp, const int *col, const int *srcpos, const std::vector<int> *cuts,
                      const int *level_of_row, int max_lines, int flags, int balanced, WinFormat &out);

// Append `b` to `a`: used to put the lower and the upper half of a triangular factor into one run list.
void win_append(WinFormat &a, const WinFormat &b);

}  // namespace nsk
