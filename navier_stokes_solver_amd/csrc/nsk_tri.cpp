// nsk_tri.cpp — host symbolic analysis + device numeric/apply of ILU(0) and SGS.
#include "nsk_tri.hpp"

#include <algorithm>
#include <cstdlib>
#include <numeric>

namespace nsk {

namespace {
constexpr int kSerialThreshold = 1024;  // levels smaller than this are fused into single-workgroup runs

void build_schedule(const std::vector<int> &lvl_ptr, std::vector<TriSolve::Step> &sched) {
  sched.clear();
  const int nl = (int)lvl_ptr.size() - 1;
  int l = 0;
  while (l < nl) {
    const int sz = lvl_ptr[l + 1] - lvl_ptr[l];
    if (sz >= kSerialThreshold) {
      sched.push_back(TriSolve::Step{0, l, l + 1, lvl_ptr[l], sz});
      ++l;
    } else {
      int e = l;
      while (e < nl && lvl_ptr[e + 1] - lvl_ptr[e] < kSerialThreshold) ++e;
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
int greedy_color(int nv, const std::vector<int> &grp, const std::vector<int> &gcol, std::vector<int> &color) {
  const int64_t ne = grp[nv];
  std::vector<int> trp(nv + 1, 0);
  for (int64_t k = 0; k < ne; ++k) ++trp[gcol[k] + 1];
  for (int i = 0; i < nv; ++i) trp[i + 1] += trp[i];
  std::vector<int> tcol((size_t)ne), cur(trp.begin(), trp.end() - 1);
  for (int i = 0; i < nv; ++i)
    for (int k = grp[i]; k < grp[i + 1]; ++k) tcol[cur[gcol[k]]++] = i;
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
    for (int k = trp[i]; k < trp[i + 1]; ++k) if (tcol[k] != i) visit(tcol[k]);
    int cc = 0;
    while (cc < (int)mark.size() && mark[cc] == i) ++cc;
    color[i] = cc;
    ncol = std::max(ncol, cc + 1);
  }
  return ncol;
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

void TriSolve::analyze(Ctx *c, const Csr &A, int kind_, int ordering_, const std::vector<int> &sub_off,
                       bool want_block2) {
  ctx = c;
  n = A.n_rows;
  kind = kind_;
  ordering = ordering_;
  const std::vector<int> &rp = A.h_rowptr, &cl = A.h_col;
  if ((int)rp.size() != n + 1) throw Error(-30, "TriSolve::analyze: host pattern missing");

  // emulated-rank id of every row (additive Schwarz, overlap 0, inside this GPU)
  std::vector<int> shard;
  const bool sharded = sub_off.size() > 2;
  if (sharded) {
    shard.resize(n);
    for (size_t s = 0; s + 1 < sub_off.size(); ++s)
      for (int i = sub_off[s]; i < sub_off[s + 1]; ++i) shard[i] = (int)s;
  }
  auto keep = [&](int i, int cc) { return cc < n && (!sharded || shard[cc] == shard[i]); };

  // restricted pattern R (local square block, cross-shard couplings dropped) and where each entry sits in A
  std::vector<int> rrp(n + 1, 0);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    for (int k = rp[i]; k < rp[i + 1]; ++k) cnt += keep(i, cl[k]) ? 1 : 0;
    rrp[i + 1] = cnt;
  }
  for (int i = 0; i < n; ++i) rrp[i + 1] += rrp[i];
  nnz = rrp[n];
  std::vector<int> rcol((size_t)nnz), rpos((size_t)nnz);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    int w = rrp[i];
    for (int k = rp[i]; k < rp[i + 1]; ++k)
      if (keep(i, cl[k])) { rcol[w] = cl[k]; rpos[w] = k; ++w; }
  }

  perm.clear();
  std::vector<int> pcolor;
  n_colors = 0;
  // 2x2 node structure of the restricted pattern (both velocity components of a node share their columns,
  // columns come in aligned pairs): colour NODES instead of DoFs, keep the two rows of a node adjacent
  bool block2 = want_block2 && ordering == ORDER_MULTICOLOR && n > 0 && n % 2 == 0;
  if (block2) {
    bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
    for (int r = 0; r < n / 2; ++r) {
      const int a0 = rrp[2 * r], a1 = rrp[2 * r + 1], len = a1 - a0;
      bool good = (rrp[2 * r + 2] - a1 == len) && (len % 2 == 0);
      for (int k = 0; good && k < len; k += 2) {
        const int cc = rcol[a0 + k];
        good = (cc % 2 == 0) && rcol[a0 + k + 1] == cc + 1 && rcol[a1 + k] == cc && rcol[a1 + k + 1] == cc + 1;
      }
      ok = ok && good;
    }
    block2 = ok;
  }
  if (ordering == ORDER_MULTICOLOR) {
    std::vector<int> color;
    if (block2) {
      const int nn = n / 2;
      std::vector<int> nrp(nn + 1, 0);
      for (int r = 0; r < nn; ++r) nrp[r + 1] = nrp[r] + (rrp[2 * r + 1] - rrp[2 * r]) / 2;
      std::vector<int> ncol((size_t)nrp[nn]);
#pragma omp parallel for schedule(static)
      for (int r = 0; r < nn; ++r)
        for (int k = 0; k < nrp[r + 1] - nrp[r]; ++k) ncol[(size_t)nrp[r] + k] = rcol[rrp[2 * r] + 2 * k] / 2;
      std::vector<int> ncolor;
      n_colors = greedy_color(nn, nrp, ncol, ncolor);
      color.resize(n);
      for (int i = 0; i < n; ++i) color[i] = ncolor[i / 2];
    } else {
      std::vector<int> rrp32(rrp.begin(), rrp.end());
      n_colors = greedy_color(n, rrp32, rcol, color);
    }
    // perm: colours ascending, natural order inside a colour (counting sort, stable; a node's two rows stay adjacent)
    std::vector<int> cptr(n_colors + 1, 0);
    for (int i = 0; i < n; ++i) ++cptr[color[i] + 1];
    for (int q = 0; q < n_colors; ++q) cptr[q + 1] += cptr[q];
    perm.resize(n);
    pcolor.resize(n);
    for (int i = 0; i < n; ++i) { const int w = cptr[color[i]]++; perm[w] = i; pcolor[w] = color[i]; }
  }

  // permuted CSR (sorted columns), source position of every entry, diagonal positions
  std::vector<int> iperm;
  if (!perm.empty()) {
    iperm.resize(n);
    for (int i = 0; i < n; ++i) iperm[perm[i]] = i;
  }
  std::vector<int> prp(n + 1, 0), pcol((size_t)nnz), psrc((size_t)nnz), pdiag(n, -1);
  for (int i = 0; i < n; ++i) {
    const int r = perm.empty() ? i : perm[i];
    prp[i + 1] = prp[i] + (rrp[r + 1] - rrp[r]);
  }
  int maxw = 0;
  bool missing_diag = false;
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
        buf.emplace_back(perm.empty() ? rcol[k] : iperm[rcol[k]], rpos[k]);
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
  if (missing_diag) throw Error(-31, "TriSolve::analyze: a row has no diagonal entry");
  max_row_nnz = maxw;
  if (max_row_nnz > 448) throw Error(-32, "TriSolve::analyze: row too long for the LDS-staged ILU kernel");

  // level schedule of the lower and upper dependency DAGs
  std::vector<int> levL(n, 0), levU(n, 0);
  n_levels_L = n_levels_U = 0;
  if (!perm.empty() && block2) {
    // node colours: the second row of a node depends on the first (lower) / the first on the second (upper)
    for (int i = 0; i < n; ++i) {
      levL[i] = 2 * pcolor[i] + (i & 1);
      levU[i] = 2 * (n_colors - 1 - pcolor[i]) + (1 - (i & 1));
    }
    n_levels_L = n_levels_U = 2 * n_colors;
  } else if (!perm.empty()) {
    // colour classes are independent sets: level = colour is a valid schedule for both DAGs
    // and keeps every level one contiguous run of rows
    for (int i = 0; i < n; ++i) { levL[i] = pcolor[i]; levU[i] = n_colors - 1 - pcolor[i]; }
    n_levels_L = n_levels_U = n_colors;
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

  const double mean_half = n > 0 ? 0.5 * (double)nnz / n : 0.0;
  lpr = mean_half <= 6 ? 4 : (mean_half <= 14 ? 8 : (mean_half <= 48 ? 16 : 32));

  hipStream_t s = ctx->stream;
  block2_ready = false;
  stream_ready = false;
  sf_armed = false;
  if (!perm.empty() && block2) {
    // node rows in colour order; 2x2 blocks towards earlier (L) / later (U) colours, column ids = caller-order node ids
    const int nn = n / 2;
    std::vector<int> lrp(nn + 1, 0), urp(nn + 1, 0);
    for (int r = 0; r < nn; ++r) {
      int nl = 0, nu = 0;
      for (int k = prp[2 * r]; k < prp[2 * r + 1]; k += 2) {
        const int m = pcol[k] / 2;
        if (m < r) ++nl; else if (m > r) ++nu;
      }
      lrp[r + 1] = lrp[r] + nl;
      urp[r + 1] = urp[r] + nu;
    }
    nnzL = (int64_t)lrp[nn] * 4;
    nnzU = (int64_t)urp[nn] * 4;
    std::vector<int> lcol((size_t)lrp[nn]), lsrc((size_t)lrp[nn] * 4), ucol((size_t)urp[nn]), usrc((size_t)urp[nn] * 4);
    std::vector<int> hpermn(nn), isrc((size_t)nn * 4);  // per node row: positions of l10, u01, d0, d1
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
    // node-colour boundaries in node rows
    std::vector<int> ncuts, cstart(n_colors + 1, nn);
    for (int r = nn - 1; r >= 0; --r) cstart[pcolor[2 * r]] = r;
    for (int c = 1; c < n_colors; ++c) ncuts.push_back(cstart[c]);
    ncuts.push_back(nn);
    std::vector<int> lb, ub;
    if (build_rowblocks(lrp.data(), nullptr, nn, kBlkMax, &ncuts, lb) &&
        build_rowblocks(urp.data(), nullptr, nn, kBlkMax, &ncuts, ub)) {
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
      Lrp.upload(lrp, s); Lcol.upload(lcol, s); Lsrc.upload(lsrc, s); Ldesc.upload(ld, s); Lsf.upload(lsf, s);
      Urp.upload(urp, s); Ucol.upload(ucol, s); Usrc.upload(usrc, s); Udesc.upload(ud, s); Usf.upload(usf, s);
      permn.upload(hpermn, s);
      intra_src.upload(isrc, s);
      Lval.alloc((size_t)nnzL);
      Uval.alloc((size_t)nnzU);
      intra.alloc((size_t)nn * 4);
      ctx->sync();
      block2_ready = true;
    }
  }
  if (!perm.empty() && !block2) {
    // strict-lower / strict-upper CSR halves with every colour a contiguous run of rows
    std::vector<int> lrp(n + 1, 0), urp(n + 1, 0);
    for (int i = 0; i < n; ++i) {
      lrp[i + 1] = lrp[i] + (pdiag[i] - prp[i]);
      urp[i + 1] = urp[i] + (prp[i + 1] - pdiag[i] - 1);
    }
    nnzL = lrp[n];
    nnzU = urp[n];
    std::vector<int> lcol((size_t)nnzL), lsrc((size_t)nnzL), ucol((size_t)nnzU), usrc((size_t)nnzU);
    // column ids go back to the caller's numbering, sorted, so that a row's gathers are runs of neighbours
#pragma omp parallel
    {
      std::vector<std::pair<int, int>> buf;
#pragma omp for schedule(static)
      for (int i = 0; i < n; ++i) {
        buf.clear();
        for (int k = prp[i]; k < pdiag[i]; ++k) buf.emplace_back(perm[pcol[k]], k);
        std::sort(buf.begin(), buf.end());
        int w = lrp[i];
        for (auto &e : buf) { lcol[w] = e.first; lsrc[w] = e.second; ++w; }
        buf.clear();
        for (int k = pdiag[i] + 1; k < prp[i + 1]; ++k) buf.emplace_back(perm[pcol[k]], k);
        std::sort(buf.begin(), buf.end());
        w = urp[i];
        for (auto &e : buf) { ucol[w] = e.first; usrc[w] = e.second; ++w; }
      }
    }
    std::vector<int> cuts(hLp.begin() + 1, hLp.end());  // colour boundaries (levL = colour, rows ascending)
    std::vector<int> lb, ub;
    if (build_rowblocks(lrp.data(), nullptr, n, kStreamNnz, &cuts, lb) &&
        build_rowblocks(urp.data(), nullptr, n, kStreamNnz, &cuts, ub)) {
      auto first_block_of = [&](const std::vector<int> &blk, std::vector<int> &out) {
        out.assign(n_colors + 1, 0);
        size_t b = 0;
        for (int c = 0; c <= n_colors; ++c) {
          const int row = c < n_colors ? hLp[c] : n;
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
      Lrp.upload(lrp, s); Lcol.upload(lcol, s); Lsrc.upload(lsrc, s); Ldesc.upload(ld, s); Lsf.upload(lsf, s);
      Urp.upload(urp, s); Ucol.upload(ucol, s); Usrc.upload(usrc, s); Udesc.upload(ud, s); Usf.upload(usf, s);
      Lval.alloc((size_t)nnzL);
      Uval.alloc((size_t)nnzU);
      dinv.alloc((size_t)n);
      ctx->sync();
      stream_ready = true;
    }
  }
  rowptr.upload(prp, s);
  col.upload(pcol, s);
  srcpos.upload(psrc, s);
  diag.upload(pdiag, s);
  if (!perm.empty()) d_perm.upload(perm, s);
  lvlL_ptr.upload(hLp, s);
  lvlL_rows.upload(hLr, s);
  lvlU_ptr.upload(hUp, s);
  lvlU_rows.upload(hUr, s);
  val.alloc((size_t)nnz);
  y.alloc((size_t)n + 1);
  ctx->sync();  // host staging vectors die at scope exit
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
    for (const Step &st : schedL) {
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
    tri_stream_syncfree(s, L, n_Lsf, 1, kind, kStreamNnz, 0, dinv.p, d_perm.p, b, nullptr, y.p, x, sf_err.p, sf_dbg);
    tri_stream_syncfree(s, U, n_Usf, 0, kind, kStreamNnz, sf_fault ? 1 : 0, dinv.p, d_perm.p, nullptr, y.p, x, y.p, sf_err.p,
                        sf_dbg ? sf_dbg + (size_t)n_Lsf * 16 : nullptr);
    ++ctx->st.tri_applies;
    ctx->st.tri_bytes += (double)apply_bytes();
    return;
  }
  if (stream_ready && use_stream && !tiny) {
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
    if (x_layout) {  // colour-ordered working vectors y, xc; the upper half also writes the caller-order result
      if (xc.n != (size_t)n + 1) xc.alloc((size_t)n + 1);
      tri_blk_syncfree(s, L, n_Lsf, 1, kind, 1, 0, intra.p, permn.p, b, nullptr, y.p, nullptr, xc.p, sf_err.p);
      tri_blk_syncfree(s, U, n_Usf, 0, kind, 1, sf_fault ? 1 : 0, intra.p, permn.p, nullptr, y.p, xc.p, x, y.p, sf_err.p);
    } else {
      tri_blk_syncfree(s, L, n_Lsf, 1, kind, 0, 0, intra.p, permn.p, b, nullptr, y.p, nullptr, x, sf_err.p);
      tri_blk_syncfree(s, U, n_Usf, 0, kind, 0, sf_fault ? 1 : 0, intra.p, permn.p, nullptr, y.p, x, nullptr, y.p, sf_err.p);
    }
    ++ctx->st.tri_applies;
    ctx->st.tri_bytes += (double)apply_bytes();
    return;
  }
  if (block2_ready && x_layout && use_stream && !tiny)
    throw Error(-33, "the colour-ordered layout of the blocked factor needs the single-launch solves");
  if (block2_ready && use_stream && !tiny) {
    const TriBlk L{Lrp.p, Lcol.p, Lval.p, Ldesc.p}, U{Urp.p, Ucol.p, Uval.p, Udesc.p};
    for (int c = 0; c < n_colors; ++c) tri_blk_level(s, L, LB[c], LB[c + 1], 1, kind, intra.p, permn.p, b, x);
    for (int c = n_colors - 1; c >= 0; --c) tri_blk_level(s, U, UB[c], UB[c + 1], 0, kind, intra.p, permn.p, nullptr, x);
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
