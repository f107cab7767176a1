// nsk_tri.cpp — host symbolic analysis + device numeric/apply of ILU(0) and SGS.
#include "nsk_tri.hpp"

#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>

namespace nsk {

namespace {
constexpr int kSerialThreshold = 1024;  // levels smaller than this are fused into single-workgroup runs
// ... of the solves.  The numeric factorisation has its own limit: a row there is a chain of a dozen dependent trips to
// L2 (one per entry of its lower part), ~12 us, and a workgroup's 16 wavefronts take the rows of a level in turns — the
// 4 001 levels of ~120 rows of the natural-order pressure-mass factor at 600x200 took 474 ms in ONE workgroup, 43 % of
// the GPU time of config 5's bench run.  One launch per level spreads a level's rows over the chip.
constexpr int kFactorSerialThreshold = 48;

void build_schedule(const std::vector<int> &lvl_ptr, std::vector<TriSolve::Step> &sched, int threshold = kSerialThreshold) {
  sched.clear();
  const int nl = (int)lvl_ptr.size() - 1;
  int l = 0;
  while (l < nl) {
    const int sz = lvl_ptr[l + 1] - lvl_ptr[l];
    if (sz >= threshold) {
      sched.push_back(TriSolve::Step{0, l, l + 1, lvl_ptr[l], sz});
      ++l;
    } else {
      int e = l;
      while (e < nl && lvl_ptr[e + 1] - lvl_ptr[e] < threshold) ++e;
      sched.push_back(TriSolve::Step{1, l, e, lvl_ptr[l], lvl_ptr[e] - lvl_ptr[l]});
      l = e;
    }
  }
}

void level_lists(const std::vector<int> &level, int n_levels, std::vector<int> &lvl_ptr, std::vector<int> &rows) {
  const int n = (int)level.size();
  lvl_ptr.assign(n_levels + 1, 0);
  for (int i = 0; i < n; ++i) ++lvl_ptr[level[i] + 1];
  for (int l = 0; l < n_levels; ++l) lvl_ptr[l + 1] += lvl_ptr[l];
  rows.resize(n);
  std::vector<int> cur(lvl_ptr.begin(), lvl_ptr.end() - 1);
  for (int i = 0; i < n; ++i) rows[cur[level[i]]++] = i;  // ascending row id inside a level
}
}  // namespace

// greedy distance-1 colouring on the graph of G + G^T, vertices visited in natural order
int greedy_color(int nv, const std::vector<int> &grp, const int *gcol, std::vector<int> &color) {
  const int64_t ne = grp[nv];
  // A structurally symmetric graph with sorted rows (finite-element patterns) is its own transpose: checked in parallel
  // (one binary search per edge), which is cheaper than the serial transposition it saves (1200x400: 210 M edges)
  bool symmetric = true;
#pragma omp parallel for schedule(dynamic, 4096) reduction(&& : symmetric)
  for (int i = 0; i < nv; ++i) {
    bool ok = true;
    for (int k = grp[i]; k < grp[i + 1] && ok; ++k) {
      const int j = gcol[k];
      if (k > grp[i] && gcol[k - 1] >= j) { ok = false; break; }   // rows must be sorted for the searches
      ok = std::binary_search(gcol + grp[j], gcol + grp[j + 1], i);
    }
    symmetric = symmetric && ok;
  }
  std::vector<int> trp, tcol;
  if (!symmetric) {
    trp.assign(nv + 1, 0);
    for (int64_t k = 0; k < ne; ++k) ++trp[gcol[k] + 1];
    for (int i = 0; i < nv; ++i) trp[i + 1] += trp[i];
    tcol.resize((size_t)ne);
    std::vector<int> cur(trp.begin(), trp.end() - 1);
    for (int i = 0; i < nv; ++i)
      for (int k = grp[i]; k < grp[i + 1]; ++k) tcol[cur[gcol[k]]++] = i;
  }
  color.assign(nv, -1);
  std::vector<int> mark;
  int ncol = 0;
  for (int i = 0; i < nv; ++i) {
    auto visit = [&](int j) {
      const int cj = color[j];
      if (cj >= 0) {
        if (cj >= (int)mark.size()) mark.resize(cj + 1, -1);
        mark[cj] = i;
      }
    };
    for (int k = grp[i]; k < grp[i + 1]; ++k) if (gcol[k] != i) visit(gcol[k]);
    if (!symmetric)
      for (int k = trp[i]; k < trp[i + 1]; ++k) if (tcol[k] != i) visit(tcol[k]);
    int cc = 0;
    while (cc < (int)mark.size() && mark[cc] == i) ++cc;
    color[i] = cc;
    ncol = std::max(ncol, cc + 1);
  }
  return ncol;
}

// Line groups: up to g items (rows, or velocity nodes) that follow each other on a line of constant y of their support
// points, each a graph neighbour of the one before, never across emulated sub-domains.  A group is coloured as ONE vertex
// (its members are solved one after the other inside a workgroup), which needs fewer colours than colouring the items:
// on the reference's lattices 12 instead of 17-18 node colours for F (pairs) and 17 instead of 29-31 for the Schur
// complement (triples) at the same or lower inner iteration counts (CPU study in DESIGN.md) — the flow runs along x,
// and so does the natural ordering the ILU quality comes from.  Without support points, or on meshes without such
// lines, every group has one member and this is the plain greedy colouring.
// Output: members of group q are grp_items[grp_ptr[q] .. grp_ptr[q+1]) in +x order; groups ordered by first member.
static void line_groups(int ni, const std::vector<int> &irp, const int *icol, const double *xy, int xy_stride,
                        const int *shard_of_item, int g, std::vector<int> &grp_ptr, std::vector<int> &grp_items) {
  grp_ptr.clear();
  grp_items.clear();
  std::vector<int> grp_of((size_t)ni, -1), nxt((size_t)ni, -1);   // nxt: the following member of an item's group
  std::vector<char> head((size_t)ni, 1);
  if (g > 1 && xy && ni > 1) {
    double ymin = xy[1], ymax = xy[1];
    for (int i = 1; i < ni; ++i) {
      const double y = xy[(size_t)i * xy_stride + 1];
      ymin = std::min(ymin, y);
      ymax = std::max(ymax, y);
    }
    const double q = (ymax - ymin) * 1e-7 + 1e-300;
    std::vector<std::pair<std::pair<int64_t, double>, int>> key((size_t)ni);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < ni; ++i)
      key[i] = {{(int64_t)std::llround((xy[(size_t)i * xy_stride + 1] - ymin) / q), xy[(size_t)i * xy_stride]}, i};
    std::sort(key.begin(), key.end());
    // column rank of every item: index of its x among the distinct x values.  Groups are cut at multiples of g of this
    // rank, so that the groups of neighbouring lines sit on top of each other (lines that start behind the obstacle would
    // otherwise pair with the other parity, and the ILU of such a brick pattern is measurably worse)
    std::vector<int> xrank((size_t)ni);
    {
      double xmin = xy[0], xmax = xy[0];
      for (int i = 1; i < ni; ++i) {
        xmin = std::min(xmin, xy[(size_t)i * xy_stride]);
        xmax = std::max(xmax, xy[(size_t)i * xy_stride]);
      }
      const double qx = (xmax - xmin) * 1e-7 + 1e-300;
      std::vector<std::pair<int64_t, int>> xs((size_t)ni);
#pragma omp parallel for schedule(static)
      for (int i = 0; i < ni; ++i) xs[i] = {(int64_t)std::llround((xy[(size_t)i * xy_stride] - xmin) / qx), i};
      std::sort(xs.begin(), xs.end());
      int rank = 0;
      for (int k = 0; k < ni; ++k) {
        if (k > 0 && xs[k].first != xs[k - 1].first) ++rank;
        xrank[xs[k].second] = rank;
      }
    }
    auto adjacent = [&](int a, int b) {
      for (int k = irp[a]; k < irp[a + 1]; ++k)
        if (icol[k] == b) return true;
      return false;
    };
    int len = 1;
    for (int k = 1; k < ni; ++k) {
      const int a = key[k - 1].second, b = key[k].second;
      const bool same_line = key[k].first.first == key[k - 1].first.first;
      const bool same_shard = !shard_of_item || shard_of_item[a] == shard_of_item[b];
      const bool same_cell = xrank[b] == xrank[a] + 1 && xrank[b] / g == xrank[a] / g;
      if (same_line && same_shard && same_cell && len < g && adjacent(a, b) && adjacent(b, a)) {
        nxt[a] = b;
        head[b] = 0;
        ++len;
      } else len = 1;
    }
  }
  // groups in the order of their first member's index: the order the greedy colouring visits them in
  grp_ptr.push_back(0);
  for (int i = 0; i < ni; ++i) {
    if (!head[i]) continue;
    for (int m = i; m >= 0; m = nxt[m]) grp_items.push_back(m);
    grp_ptr.push_back((int)grp_items.size());
  }
}

// Dispatch order for the single-launch (sync-free) kernels.  Workgroup b of a grid lands on XCD b % 8 and
// workgroups start in index order.  Colours follow each other in dependency order (ascending for the lower
// half, descending for the upper half), each padded with empty runs to a multiple of 8, and inside a colour
// position 8 q + k holds the colour's run k * per + q: XCD k then owns the k-th eighth of every colour, i.e.
// the same slice of the lattice colour after colour, so most gathers find lines its own L2 already holds
// (and values produced on the same XCD are read back without polling).
static std::vector<int4> sf_dispatch_order(const std::vector<int4> &desc, const std::vector<int> &first, bool lower) {
  const int nc = (int)first.size() - 1;
  std::vector<int4> out;
  out.reserve(desc.size() + 8 * (size_t)nc);
  for (int cc = 0; cc < nc; ++cc) {
    const int c = lower ? cc : nc - 1 - cc;
    const int b0 = first[c], nb = first[c + 1] - first[c];
    const int per = (nb + 7) / 8;
    for (int p = 0; p < 8 * per; ++p) {
      const int logical = (p & 7) * per + (p >> 3);
      out.push_back(logical < nb ? desc[(size_t)b0 + logical] : make_int4(0, 0, 0, 0));
    }
  }
  return out;
}

// Host-only part of the analysis: restricted pattern, node structure, line groups, colouring, permutation.
void TriOrdering::build(int n, const int *rp, const int *cl, int ordering, const std::vector<int> &sub_off, bool want_block2,
                        const double *xy, int group) {
  static const bool chatty = [] { const char *e = getenv("NSK_VERBOSE"); return e && atoi(e) > 1; }();
  double t_last = omp_get_wtime();
  auto tick = [&](const char *what) {   // NSK_VERBOSE=2: inside the ordering
    if (!chatty) return;
    const double t = omp_get_wtime();
    fprintf(stderr, "[nsk]     ordering: %-28s %9.1f ms\n", what, 1e3 * (t - t_last));
    t_last = t;
  };

  // emulated-rank id of every row (additive Schwarz, overlap 0, inside this GPU)
  shard.clear();
  sharded = sub_off.size() > 2;
  if (sharded) {
    shard.resize(n);
    for (size_t s = 0; s + 1 < sub_off.size(); ++s)
      for (int i = sub_off[s]; i < sub_off[s + 1]; ++i) shard[i] = (int)s;
  }
  auto keep = [&](int i, int cc) { return cc < n && (!sharded || shard[cc] == shard[i]); };

  // restricted pattern R (local square block, cross-shard couplings dropped) and where each entry sits in A.  One rank
  // without sub-domains and without ghost columns drops nothing: R is the block's own pattern and is not copied (3.4 GB of
  // freshly mapped memory for F at 1200x400).
  identity = !sharded;
  if (identity) {
    bool all_local = true;
#pragma omp parallel for schedule(static) reduction(&& : all_local)
    for (int i = 0; i < n; ++i) {
      bool ok = true;
      for (int k = rp[i]; k < rp[i + 1]; ++k) ok = ok && cl[k] < n;
      all_local = all_local && ok;
    }
    identity = all_local;
  }
  if (identity) {
    rrp.assign(rp, rp + n + 1);
    nnz = rrp[n];
    UVec<int>().swap(rcol);
    UVec<int>().swap(rpos);
    rc = cl;
    rpo = nullptr;
  } else {
    rrp.assign(n + 1, 0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      int cnt = 0;
      for (int k = rp[i]; k < rp[i + 1]; ++k) cnt += keep(i, cl[k]) ? 1 : 0;
      rrp[i + 1] = cnt;
    }
    for (int i = 0; i < n; ++i) rrp[i + 1] += rrp[i];
    nnz = rrp[n];
    rcol.resize((size_t)nnz);
    rpos.resize((size_t)nnz);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      int w = rrp[i];
      for (int k = rp[i]; k < rp[i + 1]; ++k)
        if (keep(i, cl[k])) { rcol[w] = cl[k]; rpos[w] = k; ++w; }
    }
    rc = rcol.data();
    rpo = rpos.data();
  }

  tick("restricted pattern");
  perm.clear();
  pcolor.clear();
  n_colors = 0;
  // 2x2 node structure of the restricted pattern (both velocity components of a node share their columns,
  // columns come in aligned pairs): colour NODES instead of DoFs, keep the two rows of a node adjacent
  block2 = want_block2 && ordering == ORDER_MULTICOLOR && n > 0 && n % 2 == 0;
  if (block2) {
    bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
    for (int r = 0; r < n / 2; ++r) {
      const int a0 = rrp[2 * r], a1 = rrp[2 * r + 1], len = a1 - a0;
      bool good = (rrp[2 * r + 2] - a1 == len) && (len % 2 == 0);
      for (int k = 0; good && k < len; k += 2) {
        const int cc = rc[a0 + k];
        good = (cc % 2 == 0) && rc[a0 + k + 1] == cc + 1 && rc[a1 + k] == cc && rc[a1 + k + 1] == cc + 1;
      }
      ok = ok && good;
    }
    block2 = ok;
  }
  tick("node structure check");
  // chain position / length of every permuted ITEM (row, or velocity node when block2) inside its line group
  cpos.clear();
  clen.clear();
  gmax = 1;
  if (ordering == ORDER_MULTICOLOR) {
    // item graph: velocity nodes (block2) or rows
    const int ni = block2 ? n / 2 : n;
    std::vector<int> irp;
    UVec<int> icol_own;
    const int *icol = rc;
    if (block2) {
      irp.assign(ni + 1, 0);
      for (int r = 0; r < ni; ++r) irp[r + 1] = irp[r] + (rrp[2 * r + 1] - rrp[2 * r]) / 2;
      icol_own.resize((size_t)irp[ni]);
#pragma omp parallel for schedule(static)
      for (int r = 0; r < ni; ++r)
        for (int k = 0; k < irp[r + 1] - irp[r]; ++k) icol_own[(size_t)irp[r] + k] = rc[rrp[2 * r] + 2 * k] / 2;
      icol = icol_own.data();
    } else irp.assign(rrp.begin(), rrp.end());
    std::vector<int> ishard;
    if (sharded) {
      ishard.resize(ni);
      for (int r = 0; r < ni; ++r) ishard[r] = shard[block2 ? 2 * r : r];
    }
    tick("item graph");
    // line groups (one member each without support points or with group == 1)
    std::vector<int> gptr, gitems;
    line_groups(ni, irp, icol, xy, block2 ? 4 : 2, sharded ? ishard.data() : nullptr, std::max(1, std::min(group, kTriGroupMax)),
                gptr, gitems);
    const int ng = (int)gptr.size() - 1;
    tick("line groups");
    std::vector<int> gcolor;
    if (ng == ni) {
      n_colors = greedy_color(ni, irp, icol, gcolor);   // gitems is the identity then
    } else {
      // quotient graph: groups adjacent when any of their members are
      std::vector<int> gof((size_t)ni);
      for (int q = 0; q < ng; ++q)
        for (int k = gptr[q]; k < gptr[q + 1]; ++k) gof[gitems[k]] = q;
      std::vector<int> qrp(ng + 1, 0);
      std::vector<std::vector<int>> qrows((size_t)ng);
#pragma omp parallel for schedule(dynamic, 1024)
      for (int q = 0; q < ng; ++q) {
        std::vector<int> &v = qrows[q];
        for (int k = gptr[q]; k < gptr[q + 1]; ++k) {
          const int it = gitems[k];
          for (int e = irp[it]; e < irp[it + 1]; ++e)
            if (gof[icol[e]] != q) v.push_back(gof[icol[e]]);
        }
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
      }
      for (int q = 0; q < ng; ++q) qrp[q + 1] = qrp[q] + (int)qrows[q].size();
      UVec<int> qcol((size_t)qrp[ng]);
#pragma omp parallel for schedule(static)
      for (int q = 0; q < ng; ++q) std::copy(qrows[q].begin(), qrows[q].end(), qcol.begin() + qrp[q]);
      std::vector<std::vector<int>>().swap(qrows);
      n_colors = greedy_color(ng, qrp, qcol.data(), gcolor);
    }
    tick("greedy colouring (+ symmetry check)");
    // perm: colours ascending; inside a colour the groups in their order, members in +x order; a node's two rows adjacent
    std::vector<int> cptr(n_colors + 1, 0);
    for (int q = 0; q < ng; ++q) cptr[gcolor[q] + 1] += gptr[q + 1] - gptr[q];
    for (int q = 0; q < n_colors; ++q) cptr[q + 1] += cptr[q];
    std::vector<int> iperm_item((size_t)ni);   // permuted position -> item
    cpos.resize(ni);
    clen.resize(ni);
    for (int q = 0; q < ng; ++q) {
      const int len = gptr[q + 1] - gptr[q];
      gmax = std::max(gmax, len);
      for (int k = 0; k < len; ++k) {
        const int w = cptr[gcolor[q]]++;
        iperm_item[w] = gitems[gptr[q] + k];
        cpos[w] = (unsigned char)k;
        clen[w] = (unsigned char)len;
      }
    }
    perm.resize(n);
    pcolor.resize(n);
    {
      std::vector<int> color_of_item((size_t)ni);
      for (int q = 0; q < ng; ++q)
        for (int k = gptr[q]; k < gptr[q + 1]; ++k) color_of_item[gitems[k]] = gcolor[q];
      for (int w = 0; w < ni; ++w) {
        const int it = iperm_item[w];
        if (block2) {
          perm[2 * w] = 2 * it; perm[2 * w + 1] = 2 * it + 1;
          pcolor[2 * w] = pcolor[2 * w + 1] = color_of_item[it];
        } else { perm[w] = it; pcolor[w] = color_of_item[it]; }
      }
    }
    tick("permutation");
  }
}

void TriSolve::analyze(Ctx *c, const Csr &A, int kind_, int ordering_, const std::vector<int> &sub_off,
                       bool want_block2, const double *xy, int group) {
  ctx = c;
  n = A.n_rows;
  kind = kind_;
  ordering = ordering_;
  if ((int)A.h_rowptr.size() != n + 1) throw Error(-30, "TriSolve::analyze: host pattern missing");
  static const bool chatty = [] { const char *e = getenv("NSK_VERBOSE"); return e && atoi(e) != 0; }();
  double t_last = omp_get_wtime();
  auto tick = [&](const char *what) {   // NSK_VERBOSE=1: where the host-side analysis spends its time
    if (!chatty) return;
    const double t = omp_get_wtime();
    fprintf(stderr, "[nsk]   analysis: %-30s %9.1f ms\n", what, 1e3 * (t - t_last));
    t_last = t;
  };
  TriOrdering O;
  O.build(n, A.h_rowptr.data(), A.h_col.data(), ordering, sub_off, want_block2, xy, group);
  tick("ordering (colouring)");
  nnz = O.nnz;
  n_colors = O.n_colors;
  gmax = O.gmax;
  grouped = gmax > 1;
  perm = O.perm;
  const std::vector<int> &rrp = O.rrp, &pcolor = O.pcolor;
  const int *const rcol = O.rc, *const rpos = O.rpo;   // (rpos == nullptr: entry k of the restricted pattern is entry k of A)
  const std::vector<unsigned char> &cpos = O.cpos, &clen = O.clen;
  const bool block2 = O.block2;

  // permuted CSR (sorted columns), source position of every entry, diagonal positions
  std::vector<int> iperm;
  if (!perm.empty()) {
    iperm.resize(n);
    for (int i = 0; i < n; ++i) iperm[perm[i]] = i;
  }
  std::vector<int> prp(n + 1, 0), pdiag;
  UVec<int> pcol, psrc;
  int maxw = 0;
  for (int i = 0; i < n; ++i) {
    const int r = perm.empty() ? i : perm[i];
    prp[i + 1] = prp[i] + (rrp[r + 1] - rrp[r]);
    maxw = std::max(maxw, rrp[r + 1] - rrp[r]);
  }
  // Multicolour factors without line groups, nothing dropped from the block: the permuted pattern and the split halves
  // are built ON THE DEVICE from the block's own pattern, which is there already (nsk_setup_kernels.hip) — the host
  // builds no array of the factor's size.  NSK_HOST_ANALYSIS=1: the host path below (A/B, and what the tests compare with)
  static const bool host_only = [] { const char *e = getenv("NSK_HOST_ANALYSIS"); return e && atoi(e) != 0; }();
  const bool dev = !host_only && !host_analysis && !perm.empty() && gmax == 1 && O.identity && A.rowptr.p && A.col.p && (int64_t)A.nnz == (int64_t)nnz &&
                   maxw <= 448 && n > 0;
  hipStream_t s = ctx->stream;
  bool missing_diag = false;
  if (dev) {
    DBuf<int> d_iperm, d_err;
    d_perm.upload(perm, s);
    d_iperm.upload(iperm, s);
    d_err.alloc(1);
    NSK_HIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
    rowptr.upload(prp, s);
    col.alloc((size_t)nnz);
    srcpos.alloc((size_t)nnz);
    diag.alloc((size_t)n);
    setup_permute_rows(s, n, A.rowptr.p, A.col.p, d_perm.p, d_iperm.p, rowptr.p, col.p, srcpos.p, diag.p, std::max(1, maxw), d_err.p);
    int herr = 0;
    NSK_HIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    ctx->sync();
    missing_diag = herr != 0;
  } else {
  pdiag.assign(n, -1);
  pcol.resize((size_t)nnz);   // (not zeroed: filled in the parallel loop below)
  psrc.resize((size_t)nnz);
#pragma omp parallel
  {
    std::vector<std::pair<int, int>> buf;
    int lmax = 0;
    bool lmiss = false;
#pragma omp for schedule(static)
    for (int i = 0; i < n; ++i) {
      const int r = perm.empty() ? i : perm[i];
      buf.clear();
      for (int k = rrp[r]; k < rrp[r + 1]; ++k)
        buf.emplace_back(perm.empty() ? rcol[k] : iperm[rcol[k]], rpos ? rpos[k] : k);
      std::sort(buf.begin(), buf.end());
      int w = prp[i];
      for (auto &e : buf) {
        pcol[w] = e.first;
        psrc[w] = e.second;
        if (e.first == i) pdiag[i] = w;
        ++w;
      }
      lmax = std::max(lmax, (int)buf.size());
      if (pdiag[i] < 0) lmiss = true;
    }
#pragma omp critical
    {
      maxw = std::max(maxw, lmax);
      missing_diag = missing_diag || lmiss;
    }
  }
  }
  if (missing_diag) throw Error(-31, "TriSolve::analyze: a row has no diagonal entry");
  max_row_nnz = maxw;
  if (max_row_nnz > 448) throw Error(-32, "TriSolve::analyze: row too long for the LDS-staged ILU kernel");

  tick("permuted pattern");
  // level schedule of the lower and upper dependency DAGs
  std::vector<int> levL(n, 0), levU(n, 0);
  n_levels_L = n_levels_U = 0;
  if (!perm.empty() && block2) {
    // node colours: inside a colour the members of a line group follow each other (chain position), and the second
    // row of a node depends on the first (lower) / the first on the second (upper)
    for (int i = 0; i < n; ++i) {
      const int q = cpos[i / 2], len = clen[i / 2];
      levL[i] = 2 * (pcolor[i] * gmax + q) + (i & 1);
      levU[i] = 2 * ((n_colors - 1 - pcolor[i]) * gmax + (len - 1 - q)) + (1 - (i & 1));
    }
    n_levels_L = n_levels_U = 2 * n_colors * gmax;
  } else if (!perm.empty()) {
    // colour classes are independent sets of line groups: level = (colour, chain position) is a valid schedule for
    // both DAGs
    for (int i = 0; i < n; ++i) {
      levL[i] = pcolor[i] * gmax + cpos[i];
      levU[i] = (n_colors - 1 - pcolor[i]) * gmax + (clen[i] - 1 - cpos[i]);
    }
    n_levels_L = n_levels_U = n_colors * gmax;
  } else
  for (int i = 0; i < n; ++i) {
    int l = 0;
    for (int k = prp[i]; k < pdiag[i]; ++k) l = std::max(l, levL[pcol[k]] + 1);
    levL[i] = l;
    n_levels_L = std::max(n_levels_L, l + 1);
  }
  if (perm.empty())
  for (int i = n - 1; i >= 0; --i) {
    int l = 0;
    for (int k = pdiag[i] + 1; k < prp[i + 1]; ++k) l = std::max(l, levU[pcol[k]] + 1);
    levU[i] = l;
    n_levels_U = std::max(n_levels_U, l + 1);
  }
  if (n == 0) n_levels_L = n_levels_U = 0;
  std::vector<int> hLp, hLr, hUp, hUr;
  level_lists(levL, n_levels_L, hLp, hLr);
  level_lists(levU, n_levels_U, hUp, hUr);
  build_schedule(hLp, schedL);
  build_schedule(hUp, schedU);
  build_schedule(hLp, schedN, kFactorSerialThreshold);

  const double mean_half = n > 0 ? 0.5 * (double)nnz / n : 0.0;
  lpr = mean_half <= 6 ? 4 : (mean_half <= 14 ? 8 : (mean_half <= 48 ? 16 : 32));

  tick("level schedule");
  block2_ready = false;
  stream_ready = false;
  sf_armed = false;
  if (!perm.empty() && block2) {
    // node rows in colour order; 2x2 blocks towards earlier (L) / later (U) colours, column ids = caller-order node ids
    const int nn = n / 2;
    std::vector<int> lrp(nn + 1, 0), urp(nn + 1, 0);
    // (blocks towards the other members of the node's own line group are not part of the streamed lists: the kernels
    //  apply them from `cpl` once the member before has been solved)
    auto own_group = [&](int r, int m) { return m >= r - (int)cpos[r] && m <= r + ((int)clen[r] - 1 - (int)cpos[r]); };
    if (dev) {   // counts on the device, row pointers by a host prefix sum (the run plans below need them here anyway)
      DBuf<int> cl_, cu_;
      cl_.alloc((size_t)nn);
      cu_.alloc((size_t)nn);
      setup_blk_count(s, nn, rowptr.p, col.p, std::max(1, maxw / 2), cl_.p, cu_.p);
      NSK_HIP(hipMemcpyAsync(lrp.data() + 1, cl_.p, sizeof(int) * (size_t)nn, hipMemcpyDeviceToHost, s));
      NSK_HIP(hipMemcpyAsync(urp.data() + 1, cu_.p, sizeof(int) * (size_t)nn, hipMemcpyDeviceToHost, s));
      ctx->sync();
      for (int r = 0; r < nn; ++r) { lrp[r + 1] += lrp[r]; urp[r + 1] += urp[r]; }
    } else
    for (int r = 0; r < nn; ++r) {
      int nl = 0, nu = 0;
      for (int k = prp[2 * r]; k < prp[2 * r + 1]; k += 2) {
        const int m = pcol[k] / 2;
        if (own_group(r, m)) continue;
        if (m < r) ++nl; else ++nu;
      }
      lrp[r + 1] = lrp[r] + nl;
      urp[r + 1] = urp[r] + nu;
    }
    nnzL = (int64_t)lrp[nn] * 4;
    nnzU = (int64_t)urp[nn] * 4;
    UVec<int> lcol, lsrc, ucol, usrc;
    std::vector<int> hpermn, isrc;  // per node row: caller-order node; positions of l10, u01, d0, d1
    // couplings inside a line group: per node row and half up to kTriGroupMax - 1 blocks (nearest member first), -1: none
    constexpr int CW = (kTriGroupMax - 1) * 4;
    std::vector<int> lcs, ucs;
    std::vector<unsigned char> hchain;
    if (dev) {
      Lrp.upload(lrp, s);
      Urp.upload(urp, s);
      Lcol.alloc((size_t)lrp[nn]);
      Lsrc.alloc((size_t)lrp[nn] * 4);
      Ucol.alloc((size_t)urp[nn]);
      Usrc.alloc((size_t)urp[nn] * 4);
      permn.alloc((size_t)nn);
      intra_src.alloc((size_t)nn * 4);
      setup_blk_fill(s, nn, rowptr.p, col.p, d_perm.p, x_layout, std::max(1, maxw / 2), Lrp.p, Urp.p, Lcol.p, Lsrc.p, Ucol.p, Usrc.p,
                     intra_src.p, permn.p);
    } else {
    lcol.resize((size_t)lrp[nn]); lsrc.resize((size_t)lrp[nn] * 4); ucol.resize((size_t)urp[nn]); usrc.resize((size_t)urp[nn] * 4);
    hpermn.resize(nn); isrc.resize((size_t)nn * 4);
    lcs.assign((size_t)nn * CW, -1); ucs.assign((size_t)nn * CW, -1);
    hchain.resize((size_t)nn);
    for (int r = 0; r < nn; ++r) hchain[r] = (unsigned char)(cpos[r] | (clen[r] << 4));
#pragma omp parallel
    {
      std::vector<std::pair<int, int>> bl, bu;  // (caller-order node id, k offset inside the row)
#pragma omp for schedule(static)
      for (int r = 0; r < nn; ++r) {
        const int i0 = 2 * r, i1 = 2 * r + 1, a0 = prp[i0], a1 = prp[i1];
        hpermn[r] = perm[i0] / 2;
        bl.clear();
        bu.clear();
        for (int k = 0; k < prp[i0 + 1] - a0; k += 2) {
          const int m = pcol[a0 + k] / 2;
          if (m != r && own_group(r, m)) {   // coupling to another member of the line group
            const int t = (m < r ? r - m : m - r) - 1;
            int *dst = (m < r ? lcs.data() : ucs.data()) + (size_t)r * CW + 4 * t;
            dst[0] = a0 + k; dst[1] = a0 + k + 1; dst[2] = a1 + k; dst[3] = a1 + k + 1;
            continue;
          }
          // block-column id: caller-order node id, or the colour-order one for the colour-ordered working vector
          if (m < r) bl.emplace_back(x_layout ? m : perm[pcol[a0 + k]] / 2, k);
          else if (m > r) bu.emplace_back(x_layout ? m : perm[pcol[a0 + k]] / 2, k);
          else {  // the node's own 2x2 diagonal block
            isrc[4 * (size_t)r + 0] = a1 + k;      // l10 = (i1, i0)
            isrc[4 * (size_t)r + 1] = a0 + k + 1;  // u01 = (i0, i1)
            isrc[4 * (size_t)r + 2] = a0 + k;      // d0
            isrc[4 * (size_t)r + 3] = a1 + k + 1;  // d1
          }
        }
        std::sort(bl.begin(), bl.end());
        std::sort(bu.begin(), bu.end());
        size_t w = (size_t)lrp[r];
        for (auto &e : bl) {
          lcol[w] = e.first;
          lsrc[4 * w + 0] = a0 + e.second; lsrc[4 * w + 1] = a0 + e.second + 1;
          lsrc[4 * w + 2] = a1 + e.second; lsrc[4 * w + 3] = a1 + e.second + 1;
          ++w;
        }
        w = (size_t)urp[r];
        for (auto &e : bu) {
          ucol[w] = e.first;
          usrc[4 * w + 0] = a0 + e.second; usrc[4 * w + 1] = a0 + e.second + 1;
          usrc[4 * w + 2] = a1 + e.second; usrc[4 * w + 3] = a1 + e.second + 1;
          ++w;
        }
      }
    }
    }
    // node-colour boundaries in node rows
    std::vector<int> ncuts, cstart(n_colors + 1, nn);
    for (int r = nn - 1; r >= 0; --r) cstart[pcolor[2 * r]] = r;
    for (int c = 1; c < n_colors; ++c) ncuts.push_back(cstart[c]);
    ncuts.push_back(nn);
    std::vector<int> lb, ub;
    std::vector<unsigned char> glue((size_t)nn);   // a run never splits a line group
    for (int r = 0; r < nn; ++r) glue[r] = cpos[r] > 0;
    if (build_rowblocks(lrp.data(), nullptr, nn, kBlkMax, &ncuts, lb, grouped ? glue.data() : nullptr) &&
        build_rowblocks(urp.data(), nullptr, nn, kBlkMax, &ncuts, ub, grouped ? glue.data() : nullptr)) {
      auto first_block_of = [&](const std::vector<int> &blk, std::vector<int> &out) {
        out.assign(n_colors + 1, 0);
        size_t b = 0;
        for (int c = 0; c <= n_colors; ++c) {
          const int row = c < n_colors ? cstart[c] : nn;
          while (b + 1 < blk.size() && blk[b] < row) ++b;
          out[c] = (int)b;
        }
      };
      first_block_of(lb, LB);
      first_block_of(ub, UB);
      auto make_desc = [](const std::vector<int> &blk, const std::vector<int> &rp_) {
        std::vector<int4> d(blk.size() - 1);
        for (size_t b = 0; b + 1 < blk.size(); ++b) d[b] = make_int4(blk[b], blk[b + 1], rp_[blk[b]], rp_[blk[b + 1]]);
        return d;
      };
      const std::vector<int4> ld = make_desc(lb, lrp), ud = make_desc(ub, urp);
      const std::vector<int4> lsf = sf_dispatch_order(ld, LB, true), usf = sf_dispatch_order(ud, UB, false);
      n_Lsf = (int)lsf.size();
      n_Usf = (int)usf.size();
      if (!dev) {
        Lrp.upload(lrp, s); Lcol.upload(lcol, s); Lsrc.upload(lsrc, s);
        Urp.upload(urp, s); Ucol.upload(ucol, s); Usrc.upload(usrc, s);
        permn.upload(hpermn, s);
        intra_src.upload(isrc, s);
      }
      Ldesc.upload(ld, s); Lsf.upload(lsf, s);
      Udesc.upload(ud, s); Usf.upload(usf, s);
      Lval.alloc((size_t)nnzL);
      Uval.alloc((size_t)nnzU);
      intra.alloc((size_t)nn * 4);
      if (grouped) {
        chain.upload(hchain, s);
        Lcpl_src.upload(lcs, s); Ucpl_src.upload(ucs, s);
        Lcpl.alloc(lcs.size()); Ucpl.alloc(ucs.size());
      }
      ctx->sync();
      block2_ready = true;
    }
  }
  if (!perm.empty() && !block2) {
    // strict-lower / strict-upper CSR halves with every colour a contiguous run of rows
    std::vector<int> lrp(n + 1, 0), urp(n + 1, 0);
    // (entries towards the other members of the row's own line group: applied from `cpl`, not streamed)
    auto own_group = [&](int r, int m) { return m >= r - (int)cpos[r] && m <= r + ((int)clen[r] - 1 - (int)cpos[r]); };
    constexpr int CW = kTriGroupMax - 1;
    std::vector<int> lcs, ucs;
    std::vector<unsigned char> hchain;
    if (dev) {   // strict lower / upper counts from the diagonal positions the device found
      pdiag.resize(n);
      NSK_HIP(hipMemcpyAsync(pdiag.data(), diag.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
      ctx->sync();
      for (int i = 0; i < n; ++i) { lrp[i + 1] = pdiag[i] - prp[i]; urp[i + 1] = prp[i + 1] - pdiag[i] - 1; }
    } else {
    lcs.assign((size_t)n * CW, -1);
    ucs.assign((size_t)n * CW, -1);
    hchain.resize((size_t)n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      hchain[i] = (unsigned char)(cpos[i] | (clen[i] << 4));
      int nl = 0, nu = 0;
      for (int k = prp[i]; k < prp[i + 1]; ++k) {
        const int m = pcol[k];
        if (m == i) continue;
        if (own_group(i, m)) { (m < i ? lcs : ucs)[(size_t)i * CW + ((m < i ? i - m : m - i) - 1)] = k; continue; }
        if (m < i) ++nl; else ++nu;
      }
      lrp[i + 1] = nl;
      urp[i + 1] = nu;
    }
    }
    for (int i = 0; i < n; ++i) { lrp[i + 1] += lrp[i]; urp[i + 1] += urp[i]; }
    nnzL = lrp[n];
    nnzU = urp[n];
    UVec<int> lcol, lsrc, ucol, usrc;
    // column ids go back to the caller's numbering, sorted, so that a row's gathers are runs of neighbours
    if (dev) {
      // (8 spare entries behind the index arrays: see below)
      Lrp.upload(lrp, s);
      Urp.upload(urp, s);
      Lcol.alloc((size_t)nnzL + 8);
      Lsrc.alloc((size_t)nnzL);
      Ucol.alloc((size_t)nnzU + 8);
      Usrc.alloc((size_t)nnzU);
      NSK_HIP(hipMemsetAsync(Lcol.p + nnzL, 0, 8 * sizeof(int), s));
      NSK_HIP(hipMemsetAsync(Ucol.p + nnzU, 0, 8 * sizeof(int), s));
      setup_csr_fill(s, n, rowptr.p, col.p, diag.p, d_perm.p, std::max(1, maxw), Lrp.p, Urp.p, Lcol.p, Lsrc.p, Ucol.p, Usrc.p);
    } else {
    lcol.resize((size_t)nnzL); lsrc.resize((size_t)nnzL); ucol.resize((size_t)nnzU); usrc.resize((size_t)nnzU);
#pragma omp parallel
    {
      std::vector<std::pair<int, int>> buf;
#pragma omp for schedule(static)
      for (int i = 0; i < n; ++i) {
        buf.clear();
        for (int k = prp[i]; k < pdiag[i]; ++k)
          if (!own_group(i, pcol[k])) buf.emplace_back(perm[pcol[k]], k);
        std::sort(buf.begin(), buf.end());
        int w = lrp[i];
        for (auto &e : buf) { lcol[w] = e.first; lsrc[w] = e.second; ++w; }
        buf.clear();
        for (int k = pdiag[i] + 1; k < prp[i + 1]; ++k)
          if (!own_group(i, pcol[k])) buf.emplace_back(perm[pcol[k]], k);
        std::sort(buf.begin(), buf.end());
        w = urp[i];
        for (auto &e : buf) { ucol[w] = e.first; usrc[w] = e.second; ++w; }
      }
    }
    }
    // colour boundaries (the rows are sorted by colour)
    std::vector<int> cfirst(n_colors + 1, n), cuts;
    for (int i = n - 1; i >= 0; --i) cfirst[pcolor[i]] = i;
    for (int c = n_colors - 1; c >= 0; --c) cfirst[c] = std::min(cfirst[c], cfirst[c + 1]);
    for (int c = 1; c <= n_colors; ++c) cuts.push_back(cfirst[c]);
    std::vector<int> lb, ub;
    std::vector<unsigned char> glue((size_t)n);   // a run never splits a line group
    for (int i = 0; i < n; ++i) glue[i] = cpos[i] > 0;
    if (build_rowblocks(lrp.data(), nullptr, n, kStreamNnz, &cuts, lb, grouped ? glue.data() : nullptr) &&
        build_rowblocks(urp.data(), nullptr, n, kStreamNnz, &cuts, ub, grouped ? glue.data() : nullptr)) {
      auto first_block_of = [&](const std::vector<int> &blk, std::vector<int> &out) {
        out.assign(n_colors + 1, 0);
        size_t b = 0;
        for (int c = 0; c <= n_colors; ++c) {
          const int row = c < n_colors ? cfirst[c] : n;
          while (b + 1 < blk.size() && blk[b] < row) ++b;
          out[c] = (int)b;
        }
      };
      first_block_of(lb, LB);
      first_block_of(ub, UB);
      auto make_desc = [](const std::vector<int> &blk, const std::vector<int> &rp_) {
        std::vector<int4> d(blk.size() - 1);
        for (size_t b = 0; b + 1 < blk.size(); ++b) d[b] = make_int4(blk[b], blk[b + 1], rp_[blk[b]], rp_[blk[b + 1]]);
        return d;
      };
      const std::vector<int4> ld = make_desc(lb, lrp), ud = make_desc(ub, urp);
      const std::vector<int4> lsf = sf_dispatch_order(ld, LB, true), usf = sf_dispatch_order(ud, UB, false);
      n_Lsf = (int)lsf.size();
      n_Usf = (int)usf.size();
      // (8 spare entries behind the index and value arrays: the wide loads of the single-launch kernels take a thread's 8
      //  consecutive entries at once and may read past a run's — the array's — last one)
      if (!dev) {
        lcol.resize(lcol.size() + 8, 0);
        ucol.resize(ucol.size() + 8, 0);
        Lrp.upload(lrp, s); Lcol.upload(lcol, s); Lsrc.upload(lsrc, s);
        Urp.upload(urp, s); Ucol.upload(ucol, s); Usrc.upload(usrc, s);
      }
      Ldesc.upload(ld, s); Lsf.upload(lsf, s);
      Udesc.upload(ud, s); Usf.upload(usf, s);
      Lval.alloc((size_t)nnzL + 8);
      Uval.alloc((size_t)nnzU + 8);
      NSK_HIP(hipMemsetAsync(Lval.p + nnzL, 0, 8 * sizeof(double), s));
      NSK_HIP(hipMemsetAsync(Uval.p + nnzU, 0, 8 * sizeof(double), s));
      dinv.alloc((size_t)n);
      if (grouped) {
        chain.upload(hchain, s);
        Lcpl_src.upload(lcs, s); Ucpl_src.upload(ucs, s);
        Lcpl.alloc(lcs.size()); Ucpl.alloc(ucs.size());
      }
      ctx->sync();
      stream_ready = true;
    }
  }
  tick("split factors, runs, uploads");
  // Natural ordering: per-(pass, wavefront) records for the LDS-ring solve (nsk_kernels.h).  A pass = at most kRingRows
  // rows of one level (independent), sorted by length; positions = the order the passes walk the rows in; an entry names
  // the ring slot of its column's POSITION.
  ring_ready = false;
  if (perm.empty() && n >= 4096 && n < kRingMaxRows) {
    struct RingPlan {
      std::vector<int> pos, order, pass_first;   // position of a row; row at a position; first position of a pass (+ end)
      std::vector<int> wave_first;               // per pass kRingWaves + 1 positions: the rows of its wavefronts
    };
    auto plan_ring = [&](const std::vector<int> &asap, bool lower, RingPlan &P) -> bool {
      // Pass order.  The earliest level of a row (asap) can lie far before its consumers' — rows behind the obstacle
      // are ready at once and needed hundreds of levels later — and a value must not wait that long in the ring.  So
      // every row starts as LATE as its consumers allow: t(row) = min over its consumers of t(consumer) - 1, rows
      // nobody consumes keep their earliest level.  Consumers of a row come later in the index order (lower half) or
      // earlier (upper half), so one sweep against that order settles it.  Measured on the pressure-mass pattern at
      // 600x200: longest wait 54 113 positions with the earliest levels, 1 069 with these.
      std::vector<int> tt((size_t)n), tmin((size_t)n, INT32_MAX);
      for (int q = 0; q < n; ++q) {
        const int i = lower ? n - 1 - q : q;
        tt[i] = tmin[i] == INT32_MAX ? asap[i] : tmin[i];
        const int kb = lower ? prp[i] : pdiag[i] + 1, ke = lower ? pdiag[i] : prp[i + 1];
        if (ke - kb > kRingLpr * kRingRegs) return false;
        for (int e = kb; e < ke; ++e) tmin[pcol[e]] = std::min(tmin[pcol[e]], tt[i] - 1);
      }
      int nl = 0;
      for (int i = 0; i < n; ++i) nl = std::max(nl, tt[i] + 1);
      std::vector<int> lp, lr;
      level_lists(tt, nl, lp, lr);
      auto regs_of = [&](int i) { return ((lower ? pdiag[i] - prp[i] : prp[i + 1] - pdiag[i] - 1) + kRingLpr - 1) / kRingLpr; };
      P.pos.assign((size_t)n, 0);
      P.order.clear();
      P.order.reserve((size_t)n);
      P.pass_first.clear();
      P.wave_first.clear();
      for (int l = 0; l < nl; ++l)
        for (int b = lp[l]; b < lp[l + 1]; b += kRingRows) {
          const int cnt = std::min(kRingRows, lp[l + 1] - b);
          const int o = (int)P.order.size();
          P.pass_first.push_back(o);
          P.order.insert(P.order.end(), lr.begin() + b, lr.begin() + b + cnt);
          // longest rows first (ties: row order); a wavefront's chunk is padded to its longest row.  Rows of one length CAN
          // get wavefronts of their own while the pass has wavefronts to spare (NSK_RING_BY_CLASS=1: no padding at all) —
          // measured slower, 2.86 against 2.79 ms per application at 600x200: seven thinly filled wavefronts per level
          // issue more loads than five full ones, and the loads are what a level costs
          std::stable_sort(P.order.begin() + o, P.order.end(), [&](int x, int y) { return regs_of(x) > regs_of(y); });
          int waves_by_class = 0;
          for (int p = o; p < o + cnt;) {
            int e = p;
            while (e < o + cnt && regs_of(P.order[e]) == regs_of(P.order[p])) ++e;
            waves_by_class += (e - p + kRingRowsPerWave - 1) / kRingRowsPerWave;
            p = e;
          }
          static const bool want_by_class = [] { const char *e = getenv("NSK_RING_BY_CLASS"); return e && atoi(e) != 0; }();
          const bool by_class = want_by_class && waves_by_class <= kRingWaves;
          int p = o;
          for (int w = 0; w < kRingWaves; ++w) {
            P.wave_first.push_back(p);
            int e = std::min(o + cnt, p + kRingRowsPerWave);
            if (by_class)
              for (int k = p; k < e; ++k) if (regs_of(P.order[k]) != regs_of(P.order[p])) { e = k; break; }
            p = e;
          }
          P.wave_first.push_back(p);   // (== o + cnt: by class when that needs no more than kRingWaves wavefronts, else 32 rows each)
        }
      P.pass_first.push_back((int)P.order.size());
      for (int p = 0; p < n; ++p) P.pos[P.order[p]] = p;
      return true;
    };
    auto build_ring = [&](const RingPlan &P, const RingPlan &other, bool lower, Ring &Rg) -> bool {
      const int np_real = (int)P.pass_first.size() - 1;
      const int np = (np_real + kRingStep - 1) / kRingStep * kRingStep, np_alloc = np + kRingStep + 8;
      auto first_pos = [&](int q) { return P.pass_first[std::min(q, np_real)]; };
      // longest dependency (in positions), then the epoch: behind the barrier in front of epoch k the slots of epoch
      // k + 1 are set back to NaN — their old occupants (kRingSlots positions earlier) must have been read for the last
      // time before epoch k: two epochs + the longest dependency fit the ring
      int maxback = 0;
      bool ok = true;
#pragma omp parallel for schedule(static) reduction(max : maxback) reduction(&& : ok)
      for (int i = 0; i < n; ++i) {
        const int kb = lower ? prp[i] : pdiag[i] + 1, ke = lower ? pdiag[i] : prp[i + 1];
        for (int e = kb; e < ke; ++e) {
          const int back = P.pos[i] - P.pos[pcol[e]];
          if (back <= 0) ok = false;
          maxback = std::max(maxback, back);
        }
      }
      if (!ok) return false;
      int epoch = 0;
      for (int B = kRingMaxEpoch; B >= kRingStep && !epoch; B -= kRingStep) {
        bool fits = true;
        for (int q = 0; q < np && fits; q += B) fits = first_pos(q + 2 * B) - first_pos(q) + maxback <= kRingSlots;
        if (fits) epoch = B;
      }
      if (!epoch) return false;
      std::vector<int2> rearm((size_t)np / epoch + 2, make_int2(0, 0));
      for (int k = 1; k * epoch < np; ++k) rearm[k] = make_int2(first_pos((k + 1) * epoch), first_pos((k + 2) * epoch) - first_pos((k + 1) * epoch));
      auto slot_off = [](int p) { return (unsigned)(8 + 8 * (p & (kRingSlots - 1))); };
      // headers and entries: a wavefront's chunk = regs x (2 x rows) entries, register-major
      std::vector<uint4> hdr((size_t)np_alloc * kRingWaves, make_uint4(0, 0, 0, 0));
      std::vector<int> es;
      std::vector<unsigned> eo;
      es.reserve((size_t)nnz / 2 + (size_t)n);
      eo.reserve(es.capacity());
      for (int q = 0; q < np_real; ++q)
        for (int w = 0; w < kRingWaves; ++w) {
          const int r0 = P.wave_first[(size_t)q * (kRingWaves + 1) + w], r1 = P.wave_first[(size_t)q * (kRingWaves + 1) + w + 1];
          int regs = 0;
          for (int p = r0; p < r1; ++p) {
            const int i = P.order[p];
            regs = std::max(regs, ((lower ? pdiag[i] - prp[i] : prp[i + 1] - pdiag[i] - 1) + kRingLpr - 1) / kRingLpr);
          }
          const size_t ebase = es.size();
          if (ebase >= (size_t)UINT32_MAX - 4096) return false;
          for (int r = 0; r < regs; ++r)
            for (int p = r0; p < r1; ++p) {
              const int i = P.order[p];
              const int kb = lower ? prp[i] : pdiag[i] + 1, ke = lower ? pdiag[i] : prp[i + 1];
              for (int l = 0; l < kRingLpr; ++l) {
                const int e = kb + kRingLpr * r + l;   // entry 2 r + l of the row: walker lane (2 r + l) % 8
                if (e < ke) { es.push_back(e); eo.push_back(slot_off(P.pos[pcol[e]])); }
                else { es.push_back(-1); eo.push_back(0u); }   // (behind a row's last entry: 0.0 times LDS word 0 = 0.0)
              }
            }
          hdr[(size_t)q * kRingWaves + w] = make_uint4((unsigned)ebase, (unsigned)r0 | (unsigned)(r1 - r0) << 26, (unsigned)regs, 0u);
        }
      std::vector<char> ent(es.size() * 12 + 16, 0), rowrec((size_t)n * 16 + 16, 0);
      for (size_t k = 0; k < es.size(); ++k) memcpy(&ent[k * 12 + 8], &eo[k], 4);
      std::vector<int> ds((size_t)n);
      for (int p = 0; p < n; ++p) {
        const int i = P.order[p];
        ds[p] = pdiag[i];
        // the lower half hands its result to the upper half in THAT half's position order, the upper half writes the caller's
        const unsigned m[2] = {8u * (unsigned)(lower ? other.pos[i] : i), slot_off(p)};
        memcpy(&rowrec[(size_t)p * 16 + 8], m, 8);
      }
      Rg.n_pass = np;
      Rg.epoch = epoch;
      Rg.n_ent = (long)es.size();
      Rg.hdr.upload(hdr, s);
      Rg.ent.upload(ent, s);
      Rg.rowrec.upload(rowrec, s);
      Rg.esrc.upload(es, s);
      Rg.dsrc.upload(ds, s);
      Rg.rowid.upload(P.order, s);
      Rg.rearm.upload(rearm, s);
      Rg.own.alloc((size_t)n + 2);   // (the kernel's 16-byte load of a row's own value takes the word behind it along)
      ctx->sync();
      return true;
    };
    RingPlan PL, PU;
    ring_ready = plan_ring(levL, true, PL) && plan_ring(levU, false, PU) && build_ring(PL, PU, true, ringL) &&
                 build_ring(PU, PL, false, ringU);
  }
  if (ring_ready) tick("ring records");
  if (!dev) {
    rowptr.upload(prp, s);
    col.upload(pcol, s);
    srcpos.upload(psrc, s);
    diag.upload(pdiag, s);
    if (!perm.empty()) d_perm.upload(perm, s);
  }
  lvlL_ptr.upload(hLp, s);
  lvlL_rows.upload(hLr, s);
  lvlU_ptr.upload(hUp, s);
  lvlU_rows.upload(hUr, s);
  val.alloc((size_t)nnz);
  y.alloc((size_t)n + 1);
  ctx->sync();  // host staging vectors die at scope exit
  tick("combined factor upload");
}


double TriSolve::format_bytes() const {
  if (stream_ready)   // CSR halves + per run: descriptor; per row: rowptr x2, perm x2, rhs, dinv, y (store + load), x (fill + store)
    return 12.0 * (double)(nnzL + nnzU) + 16.0 * (double)(n_Lsf + n_Usf) + (8.0 + 8.0 + 8.0 + 8.0 + 16.0 + 16.0 + 16.0) * (double)n;
  if (block2_ready)   // 2x2 blocks with one int32 block column + per node row: descriptor share, intra, rhs, y, x
    return 36.0 * (double)(nnzL + nnzU) / 4.0 + 8.0 * (double)(n / 2) + (32.0 + 4.0 + 48.0) * (double)(n / 2);
  return (double)apply_bytes();
}

void TriSolve::numeric(const double *a_val_dev) {
  hipStream_t s = ctx->stream;
  vec_gather(s, (int)nnz, srcpos.p, a_val_dev, val.p);
  if (kind == 0) {
    for (const Step &st : schedN) {
      if (st.serial) ilu0_factor_serial(s, lvlL_ptr.p, lvlL_rows.p, st.l0, st.l1, rowptr.p, diag.p, col.p, val.p, max_row_nnz);
      else ilu0_factor_level(s, st.nrows, lvlL_rows.p + st.row_off, rowptr.p, diag.p, col.p, val.p, max_row_nnz);
    }
  }
  if (block2_ready) {
    vec_gather(s, (int)nnzL, Lsrc.p, val.p, Lval.p);
    vec_gather(s, (int)nnzU, Usrc.p, val.p, Uval.p);
    vec_gather(s, 2 * n, intra_src.p, val.p, intra.p);   // per node: l10, u01, d0, d1
    invert_node_diagonals(s, n / 2, intra.p);            // d0, d1 -> 1/d0, 1/d1
  }
  if (stream_ready) {
    vec_gather(s, (int)nnzL, Lsrc.p, val.p, Lval.p);
    vec_gather(s, (int)nnzU, Usrc.p, val.p, Uval.p);
    vec_gather(s, n, diag.p, val.p, dinv.p);
    vec_recip(s, n, dinv.p, dinv.p);
  }
  if (grouped && (block2_ready || stream_ready)) {
    vec_gather_or_zero(s, (long)Lcpl.n, Lcpl_src.p, val.p, Lcpl.p);
    vec_gather_or_zero(s, (long)Ucpl.n, Ucpl_src.p, val.p, Ucpl.p);
  }
  if (ring_ready)
    for (Ring *Rg : {&ringL, &ringU}) {
      ring_fill_values(s, Rg->n_ent, Rg->esrc.p, val.p, Rg->ent.p, 12);
      ring_fill_values(s, n, Rg->dsrc.p, val.p, Rg->rowrec.p, 16);
    }
}

void TriSolve::apply(const double *b, double *x) {
  hipStream_t s = ctx->stream;
  // Tiny factors (a few MB: they sit in one XCD's L2) are latency-bound on the ~5 us per level launch:
  // one 1024-thread workgroup walking all levels with __syncthreads in between is faster there.
  const bool tiny = (double)nnz * 12.0 < tiny_bytes && !schedL.empty();
  // y stays armed with the sentinel only while consecutive applies go through the single-launch CSR kernels
  if (!(sync_free && use_stream && !tiny && (stream_ready || block2_ready))) sf_armed = false;
  if (sync_free && use_stream && !tiny && stream_ready) {
    // scalar factor: lower half into y (pre-filled with the sentinel), upper half into x; each half is ONE launch
    if (!sf_err.p) {
      sf_err.alloc(1);
      NSK_HIP(hipMemsetAsync(sf_err.p, 0, sizeof(int), s));
    }
    const TriHalf L{Lrp.p, Lcol.p, Lval.p, Lsf.p}, U{Urp.p, Ucol.p, Uval.p, Usf.p};
    // the lower half arms x for the upper half, the upper half re-arms y for the next call: no fill launches
    if (!sf_armed) { vec_fill_sentinel(s, n, y.p); sf_armed = true; }
    const TriChain cl{gmax, chain.p, Lcpl.p}, cu{gmax, chain.p, Ucpl.p};
    tri_stream_syncfree(s, L, n_Lsf, 1, kind, kStreamNnz, 0, dinv.p, d_perm.p, b, nullptr, y.p, x, sf_err.p, sf_dbg, cl);
    tri_stream_syncfree(s, U, n_Usf, 0, kind, kStreamNnz, sf_fault ? 1 : 0, dinv.p, d_perm.p, nullptr, y.p, x, y.p, sf_err.p,
                        sf_dbg ? sf_dbg + (size_t)n_Lsf * 16 : nullptr, cu);
    ++ctx->st.tri_applies;
    ctx->st.tri_bytes += (double)apply_bytes();
    return;
  }
  if (stream_ready && use_stream && !tiny && !grouped) {   // (line groups: the per-colour kernels do not know them)
    // x doubles as the intermediate vector: rows not yet solved hold L^-1 b, solved rows hold the result
    const TriHalf L{Lrp.p, Lcol.p, Lval.p, Ldesc.p}, U{Urp.p, Ucol.p, Uval.p, Udesc.p};
    for (int c = 0; c < n_colors; ++c) tri_stream_level(s, L, LB[c], LB[c + 1], 1, kind, kStreamNnz, dinv.p, d_perm.p, b, x);
    for (int c = n_colors - 1; c >= 0; --c) tri_stream_level(s, U, UB[c], UB[c + 1], 0, kind, kStreamNnz, dinv.p, d_perm.p, nullptr, x);
    ++ctx->st.tri_applies;
    ctx->st.tri_bytes += (double)apply_bytes();
    return;
  }
  if (sync_free && use_stream && !tiny && block2_ready) {
    if (!sf_err.p) {
      sf_err.alloc(1);
      NSK_HIP(hipMemsetAsync(sf_err.p, 0, sizeof(int), s));
    }
    // lower half into y, upper half into x (xc); each half is ONE launch.  The lower half arms the upper half's
    // vector with the sentinel, the upper half re-arms y for the next call: no fill launches
    if (!sf_armed) { vec_fill_sentinel(s, n, y.p); sf_armed = true; }
    const TriBlk L{Lrp.p, Lcol.p, Lval.p, Lsf.p}, U{Urp.p, Ucol.p, Uval.p, Usf.p};
    const TriChain cl{gmax, chain.p, Lcpl.p}, cu{gmax, chain.p, Ucpl.p};
    if (x_layout) {  // colour-ordered working vectors y, xc; the upper half also writes the caller-order result
      if (xc.n != (size_t)n + 1) xc.alloc((size_t)n + 1);
      tri_blk_syncfree(s, L, n_Lsf, 1, kind, 1, 0, intra.p, permn.p, b, nullptr, y.p, nullptr, xc.p, sf_err.p, cl);
      tri_blk_syncfree(s, U, n_Usf, 0, kind, 1, sf_fault ? 1 : 0, intra.p, permn.p, nullptr, y.p, xc.p, x, y.p, sf_err.p, cu);
    } else {
      tri_blk_syncfree(s, L, n_Lsf, 1, kind, 0, 0, intra.p, permn.p, b, nullptr, y.p, nullptr, x, sf_err.p, cl);
      tri_blk_syncfree(s, U, n_Usf, 0, kind, 0, sf_fault ? 1 : 0, intra.p, permn.p, nullptr, y.p, x, nullptr, y.p, sf_err.p, cu);
    }
    ++ctx->st.tri_applies;
    ctx->st.tri_bytes += (double)apply_bytes();
    return;
  }
  if (block2_ready && x_layout && use_stream && !tiny && !grouped)
    throw Error(-33, "the colour-ordered layout of the blocked factor needs the single-launch solves");
  if (block2_ready && use_stream && !tiny && !grouped) {
    const TriBlk L{Lrp.p, Lcol.p, Lval.p, Ldesc.p}, U{Urp.p, Ucol.p, Uval.p, Udesc.p};
    for (int c = 0; c < n_colors; ++c) tri_blk_level(s, L, LB[c], LB[c + 1], 1, kind, intra.p, permn.p, b, x);
    for (int c = n_colors - 1; c >= 0; --c) tri_blk_level(s, U, UB[c], UB[c + 1], 0, kind, intra.p, permn.p, nullptr, x);
    ++ctx->st.tri_applies;
    ctx->st.tri_bytes += (double)apply_bytes();
    return;
  }
  if (ring_ready && use_stream && !tiny) {   // the caller's order: one workgroup, passes through an LDS ring
    // a half's records into the memory-side cache with the whole chip, right before its one workgroup starts
    // (NSK_RING_PREFETCH=0: off): in a solver gigabytes have streamed through that cache since the last application,
    // and one CU fetching from HBM is what then bounds the solve (3.8 against 2.7 ms at 600x200).  Half
    // by half, so that factors of up to ~200 MB per half still find room in its 256 MB
    static const bool prefetch = [] { const char *e = getenv("NSK_RING_PREFETCH"); return !e || atoi(e) != 0; }();
    auto touch = [&](const Ring &Rg) {
      if (!prefetch) return;
      if (!touch_sink.p) touch_sink.alloc(1);
      const TouchRanges R{{Rg.ent.p, Rg.rowrec.p, (const char *)Rg.hdr.p, nullptr, nullptr, nullptr},
                          {(size_t)Rg.n_ent * 12, (size_t)n * 16, (size_t)Rg.n_pass * kRingWaves * 16, 0, 0, 0}};
      mem_touch(s, R, touch_sink.p);
    };
    vec_gather(s, n, ringL.rowid.p, b, ringL.own.p);              // the right-hand side in the lower half's position order
    touch(ringL);
    tri_ring(s, ringL.view(), 1, kind, ringL.own.p, ringU.own.p);   // (its result lands in the upper half's order)
    touch(ringU);
    tri_ring(s, ringU.view(), 0, kind, ringU.own.p, x);
    ++ctx->st.ring_applies;
    ++ctx->st.tri_applies;
    ctx->st.tri_bytes += (double)apply_bytes();
    return;
  }
  const TriView T = view();
  for (const Step &st : schedL) {
    if (st.serial) tri_lower_serial(s, T, kind, lvlL_ptr.p, lvlL_rows.p, st.l0, st.l1, b, y.p);
    else tri_lower_level(s, T, kind, lpr, lvlL_rows.p + st.row_off, st.nrows, b, y.p);
  }
  for (const Step &st : schedU) {
    if (st.serial) tri_upper_serial(s, T, kind, lvlU_ptr.p, lvlU_rows.p, st.l0, st.l1, y.p, x);
    else tri_upper_level(s, T, kind, lpr, lvlU_rows.p + st.row_off, st.nrows, y.p, x);
  }
  ++ctx->st.tri_applies;
  ctx->st.tri_bytes += (double)apply_bytes();
}

}  // namespace nsk
