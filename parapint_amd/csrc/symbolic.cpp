// Host-side symbolic analysis: static-pivot ordering, panel structure, task schedule.
// See plan.hpp for the model.  Pure C++ (no HIP), so it is unit-tested on CPU.
#include "plan.hpp"
#include <thread>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <tuple>
#include <utility>

namespace pp {
namespace {

}  // namespace

static int build_plan_ordered(int n, int nc, int nnzK, const int* rowK, const int* colK, int nnzB, const int* rowB,
                              const int* colB, const double* vals, const PlanOptions& opt, Plan& P);

int build_plan(int n, int nc, int nnzK, const int* rowK, const int* colK, int nnzB, const int* rowB,
               const int* colB, const double* vals, const PlanOptions& opt, Plan& P) {
  if (opt.order_mode != 2) return build_plan_ordered(n, nc, nnzK, rowK, colK, nnzB, rowB, colB, vals, opt, P);
  // both elimination orders, the cheaper schedule kept: a level costs its launches (two for the factorisation, one
  // in each solve sweep, at a floor of 5-8 us each on one MI355X), an entry of the factor its trips through HBM --
  // about 2500 entries to the level at a thousand instances
  // (side by side on host threads: they share nothing but their read-only inputs; opt.order_candidates: other tolerance
  // windows of the minimum-degree order besides, plan.hpp tune_for_mapped_group)
  std::vector<PlanOptions> variants;
  for (int c = 0; c <= std::max(0, std::min(opt.order_candidates, 2)); ++c) {
    for (int mode = 1; mode >= 0; --mode) {
      PlanOptions o = opt;
      o.order_mode = mode;
      if (c == 1) { o.md_delta_abs = 1; o.md_delta_rel = 0.2; }
      if (c == 2) { o.md_delta_abs = 8; o.md_delta_rel = 1.0; }
      variants.push_back(o);
    }
  }
  std::vector<Plan> plans(variants.size());
  std::vector<int> rcs(variants.size(), 0);
  {
    std::vector<std::thread> others;
    for (size_t v = 1; v < variants.size(); ++v)
      others.emplace_back([&, v]() { rcs[v] = build_plan_ordered(n, nc, nnzK, rowK, colK, nnzB, rowB, colB, vals, variants[v], plans[v]); });
    rcs[0] = build_plan_ordered(n, nc, nnzK, rowK, colK, nnzB, rowB, colB, vals, variants[0], plans[0]);
    for (auto& t : others) t.join();
  }
  auto cost = [](const Plan& x) { return (int64_t)x.n_levels * 2500 + x.usize; };
  size_t best = 0;                      // (ties: the configured window, the round-based order first)
  for (size_t v = 1; v < variants.size(); ++v)
    if (rcs[v] == 0 && (rcs[best] != 0 || cost(plans[v]) < cost(plans[best]))) best = v;
  P = std::move(plans[best]);
  P.opt.order_candidates = opt.order_candidates;
  return rcs[best];
}

static int build_plan_ordered(int n, int nc, int nnzK, const int* rowK, const int* colK, int nnzB, const int* rowB,
                              const int* colB, const double* vals, const PlanOptions& opt, Plan& P) {
  P = Plan();
  P.n = n; P.nc = nc; P.opt = opt; P.ncan = nnzK + nnzB;
  P.numeric_ordering = (vals != nullptr);
  if (opt.tile != 8) { P.error = "tile must be 8"; return 3; }
  if (opt.sn_wmax < 1 || opt.sn_wmax > PP_WMAX) { P.error = "sn_wmax out of range"; return 3; }
  if (n <= 0) { P.error = "empty block"; return 3; }

  // ---- 1. rows of the augmented matrix (K nodes 0..n-1, coupling nodes n..n+nc-1) with the
  // representative values; without values every entry counts as 1 and only diagonals that are
  // present in the pattern count as usable.
  typedef std::pair<int, double> Ent;
  std::vector<std::vector<Ent>> row(n);   // off-diagonal entries of the current Schur complement
  std::vector<double> diag(n, 0.0);
  for (int e = 0; e < nnzK; ++e) {
    int i = rowK[e], j = colK[e];
    if (i < 0 || i >= n || j < 0 || j > i) { P.error = "K entry outside the lower triangle"; return 3; }
    double v = vals ? vals[e] : 1.0;
    if (i == j) diag[i] += v;
    else { row[i].push_back({j, v}); row[j].push_back({i, v}); }
  }
  for (int e = 0; e < nnzB; ++e) {
    int c = rowB[e], j = colB[e];
    if (c < 0 || c >= nc || j < 0 || j >= n) { P.error = "border entry out of range"; return 3; }
    row[j].push_back({n + c, vals ? vals[nnzK + e] : 1.0});
  }
  for (auto& r : row) {
    std::sort(r.begin(), r.end(), [](const Ent& a, const Ent& b) { return a.first < b.first; });
    size_t o = 0;
    for (size_t t = 0; t < r.size(); ++t) {
      if (o > 0 && r[o - 1].first == r[t].first) r[o - 1].second += r[t].second;
      else r[o++] = r[t];
    }
    r.resize(o);
  }
  const double u_thr = opt.pivot_threshold;
  // MA27-style threshold test on the current row of K_i.  Coupling entries are left out: the
  // reference's sub-block solvers pivot on K_i alone (mpi_explicit_schur_complement.py:294),
  // the border only ever appears as right-hand sides.
  auto is_strong = [&](int x) -> bool {
    double m = 0.0;
    for (auto& e : row[x]) { if (e.first >= n) break; m = std::max(m, std::fabs(e.second)); }
    double d = std::fabs(diag[x]);
    return d > 0.0 && d >= u_thr * m;
  };
  std::vector<uint8_t> strong(n);
  for (int i = 0; i < n; ++i) strong[i] = is_strong(i);

  // ---- 2. constrained minimum-degree elimination, carried out numerically on the
  // representative instance, with static 1x1 / 2x2 pivots.  Height-aware: among the nodes whose
  // degree is within a tolerance of the minimum, a strong node with the smallest prospective
  // etree level is taken (chains are then eliminated nested-dissection-like and the level
  // schedule stays shallow); if the window holds no strong node the minimum-degree weak node is
  // eliminated in a 2x2 pivot with its largest off-diagonal neighbour.
  std::vector<uint8_t> eliminated(n, 0);
  std::vector<int> height(n, 0);  // level the node would get if eliminated now
  typedef std::tuple<int, int, int> Key;  // (degree, height, node)
  std::set<Key> q_strong, q_weak;
  auto key_of = [&](int x) { return Key((int)row[x].size(), height[x], x); };
  auto q_insert = [&](int x) { (strong[x] ? q_strong : q_weak).insert(key_of(x)); };
  auto q_erase = [&](int x) { (strong[x] ? q_strong : q_weak).erase(key_of(x)); };
  for (int i = 0; i < n; ++i) q_insert(i);
  auto pick = [&](const std::set<Key>& q, int dmin, int dmax) -> int {
    int best = -1, best_h = 0;
    for (int d = dmin; d <= dmax; ++d) {
      auto it = q.lower_bound(Key(d, -1, -1));
      if (it == q.end()) break;
      if (std::get<0>(*it) > dmax) break;
      if (std::get<0>(*it) != d) { d = std::get<0>(*it) - 1; continue; }
      int h = std::get<1>(*it);
      if (best < 0 || h < best_h) { best = std::get<2>(*it); best_h = h; }
    }
    return best;
  };

  std::vector<int> order; order.reserve(n);
  std::vector<std::vector<int>> pstruct;  // node ids at elimination time
  std::vector<double> lu(n + nc, 0.0), lv(n + nc, 0.0);
  std::vector<Ent> tmp;
  std::vector<int> N;
  std::vector<int> sp_start, sp_w;   // sub-pivots (1x1 / 2x2) in elimination order
  std::vector<int> sp_cluster;       // rounds: cluster of the sub-pivot (consecutive sub-pivots of one cluster form one block pivot)
  int n_clusters = 0;
  // numeric elimination of the sub-pivot (u) or (u, v) from the current Schur complement; leaves its structure in N
  auto eliminate = [&](int u, int v) {
    N.clear();
    for (auto& e : row[u]) if (e.first != v) { N.push_back(e.first); lu[e.first] = e.second; }
    double b_uv = 0.0;
    if (v >= 0) {
      for (auto& e : row[v]) {
        if (e.first == u) { b_uv = e.second; continue; }
        if (lu[e.first] == 0.0 && !std::binary_search(N.begin(), N.end(), e.first)) N.push_back(e.first);
        lv[e.first] = e.second;
      }
      std::sort(N.begin(), N.end());
      N.erase(std::unique(N.begin(), N.end()), N.end());
    }
    // inverse of the pivot block
    double i00, i01 = 0.0, i11 = 0.0;
    if (v < 0) {
      i00 = (diag[u] != 0.0) ? 1.0 / diag[u] : 0.0;
    } else {
      double det = diag[u] * diag[v] - b_uv * b_uv;
      if (det == 0.0) det = 1.0;
      i00 = diag[v] / det; i01 = -b_uv / det; i11 = diag[u] / det;
    }
    const int lvl = (v >= 0) ? std::max(height[u], height[v]) : height[u];
    q_erase(u); eliminated[u] = 1;
    if (v >= 0) { q_erase(v); eliminated[v] = 1; }
    for (int x : N) {
      if (x >= n) continue;
      q_erase(x);
      // w = inv(P) * l_x ; new row_x = row_x - sum_y (l_y . w) over y in N
      const double wx0 = i00 * lu[x] + i01 * lv[x];
      const double wx1 = i01 * lu[x] + i11 * lv[x];
      tmp.clear();
      tmp.reserve(row[x].size() + N.size());
      size_t i = 0, j = 0;
      const auto& rx = row[x];
      while (i < rx.size() || j < N.size()) {
        int a = (i < rx.size()) ? rx[i].first : INT32_MAX;
        int bnode = (j < N.size()) ? N[j] : INT32_MAX;
        if (a == u || a == v) { ++i; continue; }
        if (bnode == x) { ++j; continue; }
        if (a < bnode) { tmp.push_back(rx[i]); ++i; }
        else {
          double upd = -(lu[bnode] * wx0 + lv[bnode] * wx1);
          if (a == bnode) { tmp.push_back({a, rx[i].second + upd}); ++i; ++j; }
          else { tmp.push_back({bnode, upd}); ++j; }
        }
      }
      row[x].swap(tmp);
      diag[x] -= lu[x] * wx0 + lv[x] * wx1;
      height[x] = std::max(height[x], lvl + 1);
      strong[x] = is_strong(x);
      q_insert(x);
    }
    for (int x : N) { lu[x] = 0.0; lv[x] = 0.0; }
    sp_start.push_back((int)order.size());
    sp_w.push_back(v >= 0 ? 2 : 1);
    sp_cluster.push_back(n_clusters);
    order.push_back(u);
    if (v >= 0) { order.push_back(v); P.n_2x2++; }
    pstruct.push_back(N);
    std::vector<Ent>().swap(row[u]);
    if (v >= 0) std::vector<Ent>().swap(row[v]);
  };
  // largest off-diagonal K neighbour of a weak node that is allowed as its 2x2 partner (or -1)
  auto partner_of = [&](int uw, const std::vector<int>* blocked, int round) -> int {
    int vw = -1;
    double best = -1.0;
    for (auto& e : row[uw]) {
      if (e.first >= n) continue;
      double a = std::fabs(e.second);
      if (a > best) { best = a; vw = e.first; }
    }
    if (vw >= 0) {
      double det = diag[uw] * diag[vw] - best * best;
      double ref = std::max(std::fabs(diag[uw] * diag[vw]), best * best);
      if (!(std::fabs(det) > 1e-12 * ref)) vw = -1;  // numerically singular pair
    }
    if (vw >= 0 && blocked && (*blocked)[vw] == round) vw = -1;
    return vw;
  };
  auto pair_degree = [&](int uw, int vw) -> int {   // |N(u) u N(v)| - 2
    size_t i = 0, j = 0, cnt = 0;
    const auto &ru = row[uw], &rv = row[vw];
    while (i < ru.size() || j < rv.size()) {
      int a = (i < ru.size()) ? ru[i].first : INT_MAX, bb = (j < rv.size()) ? rv[j].first : INT_MAX;
      if (a == bb) { ++i; ++j; } else if (a < bb) ++i; else ++j;
      ++cnt;
    }
    return (int)cnt - 2;
  };
  // one step of the sequential rule: the best 1x1 / 2x2 candidate of the whole remaining graph
  auto pick_sequential = [&](int& u, int& v) {
    u = -1; v = -1;
    // best 1x1 candidate: a strong node within the tolerance window above the minimum strong
    // degree, lowest prospective level first
    int d_s = INT_MAX;
    if (!q_strong.empty()) {
      d_s = std::get<0>(*q_strong.begin());
      u = pick(q_strong, d_s, d_s + std::max(opt.md_delta_abs, (int)(opt.md_delta_rel * d_s)));
    }
    // 2x2 candidate: the minimum-degree weak node with its largest off-diagonal neighbour; taken
    // when no strong node is left, or when the pair is no more expensive than the best 1x1
    if (!q_weak.empty() && std::get<0>(*q_weak.begin()) < d_s) {
      int uw = std::get<2>(*q_weak.begin());
      int vw = partner_of(uw, nullptr, 0);
      if (vw >= 0) {
        int d_uv = pair_degree(uw, vw);
        if (u < 0 || d_uv <= d_s) { u = uw; v = vw; }
      } else if (u < 0) {
        u = uw;  // isolated / singular weak node: 1x1, flagged at run time
      }
    }
  };
  if (opt.order_mode == 0) {
    while ((int)order.size() < n) {
      int u, v;
      pick_sequential(u, v);
      eliminate(u, v);
      ++n_clusters;
    }
  } else {
    // Rounds of independent clusters.  A round takes, in order of increasing degree, every node of the degree
    // window that is not adjacent to a cluster already chosen in this round, and grows it along the elimination
    // tree into a cluster of up to sn_wmax columns: the next sub-pivot is a node of the last structure whose own
    // row adds at most sn_tol_rows rows to it (the rule by which step 2b merges a sub-pivot into its parent, so
    // the cluster becomes one block pivot).  The clusters of a round have no edges between them and form one level
    // of the schedule; what a round leaves are the separators.  On a banded coupling (a chain of bandwidth b) a
    // round eliminates sn_wmax of every sn_wmax + b nodes, where taking single nodes of lowest height first (the
    // sequential rule above) eliminates one of b + 1.
    std::vector<int> blocked(n, -1);
    std::vector<std::pair<int, int>> cand;   // (degree, node)
    int round = 0;
    while ((int)order.size() < n) {
      int d_s = q_strong.empty() ? INT_MAX : std::get<0>(*q_strong.begin());
      int d_w = q_weak.empty() ? INT_MAX : std::get<0>(*q_weak.begin());
      const int d0 = std::min(d_s, d_w);
      const int limit = d0 + std::max(opt.md_delta_abs, (int)(opt.md_delta_rel * d0));
      cand.clear();
      for (auto it = q_strong.begin(); it != q_strong.end() && std::get<0>(*it) <= limit; ++it)
        cand.push_back({std::get<0>(*it), std::get<2>(*it)});
      for (auto it = q_weak.begin(); it != q_weak.end() && std::get<0>(*it) <= limit; ++it)
        cand.push_back({std::get<0>(*it), std::get<2>(*it)});
      std::sort(cand.begin(), cand.end());
      // Padding is paid per cluster, a level once: a round with few candidates may pad more (rows a sub-pivot adds
      // to the structure of the cluster it joins) to take fewer rounds; a round with many keeps its clusters narrow,
      // because the panels below a wide block pivot are closed under it (every row of a child that holds one of its
      // columns holds them all).
      const bool sparse_round = (int)cand.size() <= opt.round_relax_pop;
      const int wmax_r = ((int)cand.size() > opt.round_narrow_pop) ? std::min(opt.sn_wmax, opt.round_narrow_wmax) : opt.sn_wmax;
      int taken = 0;
      for (auto& dc : cand) {
        int u = dc.second, v = -1;
        if (eliminated[u] || blocked[u] == round) continue;
        if (!strong[u]) {
          v = partner_of(u, &blocked, round);
          if (v < 0 || pair_degree(u, v) > limit) continue;
        }
        int width = (v >= 0) ? 2 : 1;
        eliminate(u, v);
        ++taken;
        // grow the cluster along the chain of parents
        while (width < wmax_r) {
          const int base = (int)N.size() - 1;
          const int tol_r = sparse_round ? std::max(opt.round_relax_tol_rows, (int)(opt.round_relax_tol_frac * (double)N.size()))
                                         : opt.sn_tol_rows;
          int bu = -1, bv = -1, bnew = INT_MAX;
          for (int q : N) {
            if (q >= n) break;
            if (blocked[q] == round) continue;
            const int add = (int)row[q].size() - base;
            if (add > tol_r || add >= bnew) continue;
            if (strong[q]) { bu = q; bv = -1; bnew = add; continue; }
            if (width + 2 > wmax_r) continue;
            const int q2 = partner_of(q, &blocked, round);
            if (q2 < 0 || !std::binary_search(N.begin(), N.end(), q2)) continue;
            const int add2 = pair_degree(q, q2) - ((int)N.size() - 2);
            if (add2 > tol_r || add2 >= bnew) continue;
            bu = q; bv = q2; bnew = add2;
          }
          if (bu < 0) break;
          eliminate(bu, bv);
          width += (bv >= 0) ? 2 : 1;
        }
        for (int q : N) { if (q >= n) break; blocked[q] = round; }
        ++n_clusters;
        if (pp::env_switch("PP_DEBUG_ROUNDS")) {
          fprintf(stderr, "round %d seed %d width %d N:", round, u, width);
          for (int q : N) fprintf(stderr, " %d", q);
          fprintf(stderr, "\n");
        }
      }
      if (taken == 0) {
        // nothing in the window could be taken (weak nodes without a usable partner): one step of the sequential rule
        int u, v;
        pick_sequential(u, v);
        eliminate(u, v);
        ++n_clusters;
      }
      ++round;
    }
  }
  // ---- 2b. supernodes: a sub-pivot is merged into its elimination-tree parent when their
  // structures agree up to sn_tol_rows padded rows and the merged width stays <= sn_wmax.  The
  // merged block is still eliminated in the static sub-pivot order (pivot.hpp, invert_block), so the
  // arithmetic is that of the sub-pivot sequence; what changes is the schedule: one level per
  // supernode instead of one per sub-pivot (chains of separator nodes collapse).
  const int nsp = (int)sp_w.size();
  sp_start.push_back(n);
  std::vector<int> pos_of_node(n), sp_of_pos(n);
  for (int k = 0; k < n; ++k) pos_of_node[order[k]] = k;
  for (int sidx = 0; sidx < nsp; ++sidx)
    for (int q = 0; q < sp_w[sidx]; ++q) sp_of_pos[sp_start[sidx] + q] = sidx;
  std::vector<std::vector<int>> srows(nsp);   // structure of each sub-pivot: positions (K) and n + c
  for (int sidx = 0; sidx < nsp; ++sidx) {
    auto& r = srows[sidx];
    r.reserve(pstruct[sidx].size());
    for (int x : pstruct[sidx]) r.push_back(x < n ? pos_of_node[x] : x);
    std::sort(r.begin(), r.end());
    std::vector<int>().swap(pstruct[sidx]);
  }
  std::vector<int> sp_parent(nsp, -1);
  std::vector<std::vector<int>> sp_children(nsp);
  for (int sidx = 0; sidx < nsp; ++sidx)
    if (!srows[sidx].empty() && srows[sidx][0] < n) {
      sp_parent[sidx] = sp_of_pos[srows[sidx][0]];
      sp_children[sp_parent[sidx]].push_back(sidx);
    }
  std::vector<int> chain_width(sp_w), merged_child(nsp, -1);   // merged_child[q] = child merged into q
  std::vector<uint8_t> is_merged(nsp, 0);
  // tree heights (children precede parents in elimination order) and the "tail": the heights from the
  // root down that hold at most sn_tail_pop sub-pivots each
  std::vector<int> sp_height(nsp, 0);
  int hmax = 0;
  for (int q = 0; q < nsp; ++q) {
    for (int c : sp_children[q]) sp_height[q] = std::max(sp_height[q], sp_height[c] + 1);
    hmax = std::max(hmax, sp_height[q]);
  }
  int tail_height = hmax + 1;
  {
    std::vector<int> pop(hmax + 1, 0);
    for (int q = 0; q < nsp; ++q) ++pop[sp_height[q]];
    while (opt.sn_tail_pop > 0 && tail_height > 0 && pop[tail_height - 1] <= opt.sn_tail_pop) --tail_height;
  }
  const int tail_wmax = std::min(std::max(opt.sn_tail_wmax, opt.sn_wmax), PP_WMAX);
  if (opt.sn_wmax > 1) {
    for (int q = 0; q < nsp; ++q) {
      int best = -1;
      const int q0 = sp_start[q], q1 = q0 + sp_w[q];
      const bool in_tail = sp_height[q] >= tail_height;
      const int wcap = in_tail ? tail_wmax : opt.sn_wmax;
      const int tol = in_tail ? std::max(opt.sn_tol_rows,
                                         (int)(opt.sn_tail_tol_frac * (double)(srows[q].size() + sp_w[q])))
                              : opt.sn_tol_rows;
      // columns the later sub-pivots of q's cluster will add to this chain (rounds: they are merged unconditionally)
      int later = 0;
      if (opt.order_mode != 0)
        for (int x = q + 1; x < nsp && sp_cluster[x] == sp_cluster[q]; ++x) later += sp_w[x];
      if (opt.order_mode != 0 && q > 0 && sp_cluster[q - 1] == sp_cluster[q] && sp_parent[q - 1] == q) {
        // consecutive sub-pivots of one cluster of the rounds (its growth rule has accepted the padding)
        merged_child[q] = q - 1; is_merged[q - 1] = 1; chain_width[q] = chain_width[q - 1] + sp_w[q];
        continue;
      }
      for (int c : sp_children[q]) {
        if (chain_width[c] + sp_w[q] + later > wcap) continue;
        const auto& rc = srows[c];
        // rows of c beyond q's columns vs rows of q; q's columns inside c's structure
        int have_cols = 0;
        size_t ic = 0;
        while (ic < rc.size() && rc[ic] < q1) { if (rc[ic] >= q0) ++have_cols; ++ic; }
        int missing = sp_w[q] - have_cols;
        size_t iq = 0;
        const auto& rq = srows[q];
        while (iq < rq.size()) {
          while (ic < rc.size() && rc[ic] < rq[iq]) ++ic;
          if (ic >= rc.size() || rc[ic] != rq[iq]) ++missing;
          ++iq;
        }
        if (missing > tol) continue;
        if (best < 0 || srows[c].size() > srows[best].size()) best = c;
      }
      if (best >= 0) { merged_child[q] = best; is_merged[best] = 1; chain_width[q] = chain_width[best] + sp_w[q]; }
    }
  }
  // post-order with chains contiguous: emit(head) = for every member of the chain (bottom up) first
  // the subtrees of its non-merged children, then the chain itself
  std::vector<int> sp_seq; sp_seq.reserve(nsp);
  {
    struct Frame { int head; std::vector<int> members; size_t mi; size_t ci; };
    std::vector<Frame> stack;
    auto make_frame = [&](int head) {
      Frame f; f.head = head; f.mi = 0; f.ci = 0;
      for (int x = head; x >= 0; x = merged_child[x]) f.members.push_back(x);
      std::reverse(f.members.begin(), f.members.end());   // bottom member first
      return f;
    };
    for (int root = 0; root < nsp; ++root) {
      if (sp_parent[root] >= 0 || is_merged[root]) continue;
      stack.push_back(make_frame(root));
      while (!stack.empty()) {
        Frame& f = stack.back();
        bool pushed = false;
        while (f.mi < f.members.size()) {
          const auto& ch = sp_children[f.members[f.mi]];
          while (f.ci < ch.size()) {
            const int c = ch[f.ci++];
            if (is_merged[c]) continue;          // part of this (or handled as) chain
            stack.push_back(make_frame(c));
            pushed = true;
            break;
          }
          if (pushed) break;
          ++f.mi; f.ci = 0;
        }
        if (pushed) continue;
        for (int x : f.members) sp_seq.push_back(x);
        stack.pop_back();
      }
    }
    if ((int)sp_seq.size() != nsp) { P.error = "internal: supernode post-order lost sub-pivots"; return 3; }
  }
  // supernodes = maximal runs of sp_seq linked by merges; new node order
  std::vector<int> new_order; new_order.reserve(n);
  std::vector<std::vector<int>> sn_members;
  for (size_t i = 0; i < sp_seq.size(); ++i) {
    const int x = sp_seq[i];
    if (i > 0 && merged_child[x] == sp_seq[i - 1]) sn_members.back().push_back(x);
    else sn_members.push_back({x});
  }
  // ---- 2c. root front: the top of the elimination tree is a chain of block pivots over (nearly) the same rows --
  // one gather + one scale launch and one launch in each solve sweep per link, all at their floor.  The links are
  // merged into ONE wide block pivot of up to front_max columns (plan.hpp: front_piv): its rows are gathered in
  // column slices of PP_WMAX by the ordinary tasks, its pivot block is inverted by a kernel of its own and its rows
  // are scaled with that explicit inverse.
  if (opt.front_max > PP_WMAX && sn_members.size() >= 2) {
    const int nsn = (int)sn_members.size();
    std::vector<int> sn_of_sp(nsp, -1);
    for (int i = 0; i < nsn; ++i) for (int x : sn_members[(size_t)i]) sn_of_sp[x] = i;
    std::vector<int> sn_parent(nsn, -1), sn_height(nsn, 0);
    std::vector<uint8_t> gone(nsn, 0);
    for (int i = 0; i < nsn; ++i) {
      const int par_sp = sp_parent[sn_members[(size_t)i].back()];
      if (par_sp >= 0) sn_parent[i] = sn_of_sp[par_sp];
    }
    for (int i = 0; i < nsn; ++i)       // (children precede their parents)
      if (sn_parent[i] >= 0) sn_height[sn_parent[i]] = std::max(sn_height[sn_parent[i]], sn_height[i] + 1);
    auto width_of = [&](const std::vector<int>& m) { int c = 0; for (int x : m) c += sp_w[x]; return c; };
    auto rows_of = [&](const std::vector<int>& m, int first_pos) {     // structure beyond the columns (elimination positions / n + c)
      std::vector<int> r;
      for (int x : m) for (int v : srows[x]) if (v >= n || v >= first_pos) r.push_back(v);
      std::sort(r.begin(), r.end());
      r.erase(std::unique(r.begin(), r.end()), r.end());
      return r;
    };
    const int R = nsn - 1;               // the root (last in the post-order) absorbs the chain below it
    int top_w = width_of(sn_members[(size_t)R]);
    const int end_pos = sp_start[sn_members[(size_t)R].back()] + sp_w[sn_members[(size_t)R].back()];
    while (true) {
      // the child on the critical path: strictly the tallest of the root's children (merging any other child, or one
      // of several equally tall ones, pads its columns without taking a level off the schedule)
      int C = -1, second = -1;
      for (int i = 0; i < R; ++i) {
        if (gone[i] || sn_parent[i] != R) continue;
        if (C < 0 || sn_height[i] > sn_height[C]) { second = C; C = i; }
        else if (second < 0 || sn_height[i] > sn_height[second]) second = i;
      }
      if (C < 0 || (second >= 0 && sn_height[second] == sn_height[C])) break;
      const int wc = width_of(sn_members[(size_t)C]);
      if (top_w + wc > opt.front_max) break;
      // rows the child's columns would have to be padded with
      const std::vector<int> rr = rows_of(sn_members[(size_t)R], end_pos), rc = rows_of(sn_members[(size_t)C], end_pos);
      size_t missing = 0, ic = 0;
      for (int v : rr) { while (ic < rc.size() && rc[ic] < v) ++ic; if (ic >= rc.size() || rc[ic] != v) ++missing; }
      if ((double)missing > opt.front_pad_frac * (double)rr.size() + 8.0) break;
      std::vector<int> merged(sn_members[(size_t)C]);
      merged.insert(merged.end(), sn_members[(size_t)R].begin(), sn_members[(size_t)R].end());
      sn_members[(size_t)R].swap(merged);
      gone[C] = 1;
      for (int i = 0; i < R; ++i) if (!gone[i] && sn_parent[i] == C) sn_parent[i] = R;
      top_w += wc;
    }
    {
      // (the other subtrees of the root stay where they are in the post-order: only the absorbed links leave it; a
      // merged root of at most PP_WMAX columns is an ordinary block pivot)
      std::vector<std::vector<int>> kept;
      kept.reserve(sn_members.size());
      for (int i = 0; i < nsn; ++i) if (!gone[i]) kept.push_back(std::move(sn_members[(size_t)i]));
      sn_members.swap(kept);
      if (top_w > PP_WMAX) P.front_piv = (int)sn_members.size() - 1;
    }
  }
  P.npiv = (int)sn_members.size();
  P.piv_start.assign(P.npiv + 1, 0);
  P.piv_w.assign(P.npiv, 0);
  P.piv_sub.assign(P.npiv, 0);
  for (int p = 0; p < P.npiv; ++p) {
    P.piv_start[p] = (int)new_order.size();
    int col = 0;
    for (int x : sn_members[p]) {
      if (sp_w[x] == 2) P.piv_sub[p] |= 1u << col;
      for (int q = 0; q < sp_w[x]; ++q) new_order.push_back(order[sp_start[x] + q]);
      col += sp_w[x];
    }
    P.piv_w[p] = col;
    if (col > PP_WMAX && p != P.front_piv) { P.error = "internal: block pivot wider than PP_WMAX"; return 3; }
  }
  P.piv_start[P.npiv] = n;
  P.perm = new_order;
  P.iperm.assign(n, -1);
  for (int k = 0; k < n; ++k) P.iperm[new_order[k]] = k;
  P.piv_of_col.assign(n, -1);
  for (int p = 0; p < P.npiv; ++p)
    for (int q = 0; q < P.piv_w[p]; ++q) P.piv_of_col[P.piv_start[p] + q] = p;

  // ---- 3. row structures in new indices (union over the members).  With opt.close_supernodes every panel is
  // closed under the block pivots above it (a row structure that holds one column of a block pivot holds them all:
  // explicit zeros, so that every update is a whole-row entry); without it (the default) a panel keeps its own
  // rows and an update into a block pivot of which it holds only some columns is made of single-column entries
  // (plan.hpp, FEntry).  The panels of the leaves -- most of the storage -- then carry no padding at all, however
  // wide the block pivots above them are.
  std::vector<std::vector<int>> rows(P.npiv);
  for (int p = 0; p < P.npiv; ++p) {
    auto& r = rows[p];
    const int pend = P.piv_start[p] + P.piv_w[p];
    for (int x : sn_members[p])
      for (int v : srows[x]) {
        const int nv = (v < n) ? P.iperm[order[v]] : v;
        if (nv >= pend) r.push_back(nv);
        else if (nv < P.piv_start[p]) { P.error = "internal: supernode structure points backwards"; return 3; }
      }
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
  }
  for (auto& v : srows) std::vector<int>().swap(v);
  for (int p = 0; p < P.npiv; ++p) {
    auto& r = rows[p];
    if (opt.close_supernodes) {
      bool added = false;
      size_t m = r.size();
      for (size_t t = 0; t < m; ++t) {
        int c = r[t];
        if (c >= n) break;
        int q = P.piv_of_col[c];
        if (P.piv_w[q] > 1) {
          for (int o = P.piv_start[q]; o < P.piv_start[q] + P.piv_w[q]; ++o)
            if (!std::binary_search(r.begin(), r.begin() + m, o)) { r.push_back(o); added = true; }
        }
      }
      if (added) { std::sort(r.begin(), r.end()); r.erase(std::unique(r.begin(), r.end()), r.end()); }
    }
    // fill closure: the rows beyond the parent's own columns (parent = block pivot of the first row) must appear
    // in the parent's structure (the union over the members of a block pivot and the padding above add rows)
    if (!r.empty() && r[0] < n) {
      int par = P.piv_of_col[r[0]];
      int pend = P.piv_start[par] + P.piv_w[par];
      auto& rp = rows[par];
      size_t before = rp.size();
      for (int c : r) if (c >= pend && !std::binary_search(rp.begin(), rp.begin() + before, c)) rp.push_back(c);
      if (rp.size() != before) { std::sort(rp.begin(), rp.end()); rp.erase(std::unique(rp.begin(), rp.end()), rp.end()); }
    }
  }
  P.piv_rowptr.assign(P.npiv + 1, 0);
  P.piv_uoff.assign(P.npiv, 0);
  P.piv_doff.assign(P.npiv, 0);
  for (int p = 0; p < P.npiv; ++p) {
    P.piv_doff[p] = P.dsize;
    P.dsize += P.piv_w[p] * (P.piv_w[p] + 1) / 2;
    P.piv_rowptr[p + 1] = P.piv_rowptr[p] + (int)rows[p].size();
    P.piv_uoff[p] = P.usize;
    P.usize += (int64_t)(P.piv_w[p] + (int64_t)rows[p].size()) * P.piv_w[p];
    P.nnz_L += (int64_t)rows[p].size() * P.piv_w[p];
  }
  P.piv_cslot0.assign(P.npiv, 0);
  P.piv_ncrow.assign(P.npiv, 0);
  P.piv_boff.assign(P.npiv, 0);
  for (int p = 0; p < P.npiv; ++p) {
    const auto& r = rows[p];
    size_t first = r.size();
    while (first > 0 && r[first - 1] >= n) --first;
    P.piv_cslot0[p] = P.piv_w[p] + (int)first;
    P.piv_ncrow[p] = (int)(r.size() - first);
    P.piv_boff[p] = P.bsize;
    P.bsize += P.piv_w[p] * P.piv_w[p];
  }
  P.rowidx.reserve(P.piv_rowptr[P.npiv]);
  for (int p = 0; p < P.npiv; ++p) P.rowidx.insert(P.rowidx.end(), rows[p].begin(), rows[p].end());

  auto slot_in_panel = [&](int p, int r) -> int {  // slot of new row r in panel p, or -1
    int p0 = P.piv_start[p], w = P.piv_w[p];
    if (r >= p0 && r < p0 + w) return r - p0;
    auto it = std::lower_bound(rows[p].begin(), rows[p].end(), r);
    if (it == rows[p].end() || *it != r) return -1;
    return w + (int)(it - rows[p].begin());
  };

  // ---- 4. scatter positions of the canonical input entries
  P.pos_of_can.assign(P.ncan, -1);
  for (int e = 0; e < nnzK; ++e) {
    int i = P.iperm[rowK[e]], j = P.iperm[colK[e]];
    if (i < j) std::swap(i, j);
    int p = P.piv_of_col[j];
    int s = slot_in_panel(p, i);
    if (s < 0) { P.error = "internal: K entry outside the symbolic structure"; return 3; }
    P.pos_of_can[e] = P.piv_uoff[p] + (int64_t)s * P.piv_w[p] + (j - P.piv_start[p]);
  }
  for (int e = 0; e < nnzB; ++e) {
    int j = P.iperm[colB[e]];
    int p = P.piv_of_col[j];
    int s = slot_in_panel(p, n + rowB[e]);
    if (s < 0) { P.error = "internal: border entry outside the symbolic structure"; return 3; }
    P.pos_of_can[nnzK + e] = P.piv_uoff[p] + (int64_t)s * P.piv_w[p] + (j - P.piv_start[p]);
  }

  // ---- 5. row patterns (which earlier panels hold rows of pivot p, and which of its columns) and levels
  struct RowPat { int k, mslot, cnt; int qs[PP_FRONT_MAX]; };   // rows mslot .. mslot + cnt - 1 of panel k are the columns qs[] of p
  std::vector<std::vector<RowPat>> rowpat(P.npiv);
  for (int k = 0; k < P.npiv; ++k) {
    const auto& r = rows[k];
    for (size_t t = 0; t < r.size() && r[t] < n;) {
      const int p = P.piv_of_col[r[t]];
      RowPat rp; rp.k = k; rp.mslot = P.piv_w[k] + (int)t; rp.cnt = 0;
      while (t < r.size() && r[t] < n && P.piv_of_col[r[t]] == p) { rp.qs[rp.cnt++] = r[t] - P.piv_start[p]; ++t; }
      if (opt.close_supernodes && rp.cnt != P.piv_w[p]) { P.error = "internal: pivot rows not closed"; return 3; }
      rowpat[p].push_back(rp);
    }
  }
  P.piv_level.assign(P.npiv, 0);
  for (int p = 0; p < P.npiv; ++p) {
    int lv = 0;
    for (auto& km : rowpat[p]) lv = std::max(lv, P.piv_level[km.k] + 1);
    P.piv_level[p] = lv;
    P.n_levels = std::max(P.n_levels, lv + 1);
  }
  {
    std::vector<int> cnt(P.n_levels + 1, 0);
    for (int p = 0; p < P.npiv; ++p) cnt[P.piv_level[p] + 1]++;
    for (int l = 0; l < P.n_levels; ++l) cnt[l + 1] += cnt[l];
    P.lvl_ptr = cnt;
    P.lvl_piv.assign(P.npiv, 0);
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    for (int p = 0; p < P.npiv; ++p) P.lvl_piv[fill[P.piv_level[p]]++] = p;
  }

  // ---- 5b. chain fronts (plan.hpp, PlanOptions::chain_fronts): chains p -> q in which the rows of p are exactly the
  // columns and the rows of q (q = the block pivot of p's first row; with the fill closure of step 3 the sizes decide)
  P.piv_chain.assign(P.npiv, -1);
  P.piv_flevel = P.piv_level;
  P.chain_ptr.assign(1, 0);
  if (opt.chain_fronts && (opt.batch_hint <= 0 || opt.batch_hint >= opt.chain_min_batch)) {
    std::vector<int> next(P.npiv, -1), prev(P.npiv, -1);
    for (int p = 0; p < P.npiv; ++p) {
      const auto& r = rows[p];
      if (r.empty() || r[0] >= n || P.piv_w[p] > PP_WMAX) continue;
      const int q = P.piv_of_col[r[0]];
      if (P.piv_w[q] > PP_WMAX || prev[q] >= 0) continue;
      if (r.size() != (size_t)P.piv_w[q] + rows[q].size()) continue;
      next[p] = q; prev[q] = p;
    }
    std::vector<int> ch;
    for (int h = 0; h < P.npiv; ++h) {
      if (prev[h] >= 0 || next[h] < 0) continue;            // heads of chains
      ch.clear();
      for (int q = h; q >= 0; q = next[q]) ch.push_back(q);
      size_t i = 0;
      while (i < ch.size()) {                               // cut into fronts that fit the width and LDS bounds
        const int64_t m0 = P.piv_w[ch[i]] + (int64_t)rows[ch[i]].size();
        int W = 0;
        size_t j = i;
        while (j < ch.size() && W + P.piv_w[ch[j]] <= opt.chain_wmax &&
               m0 * (W + P.piv_w[ch[j]] + 1) + 6 * m0 + 64 <= opt.chain_lds_doubles) { W += P.piv_w[ch[j]]; ++j; }
        if (j == i) { ++i; continue; }                      // (this panel alone does not fit: it stays an ordinary panel)
        if ((int)(j - i) >= opt.chain_min_panels && m0 >= opt.chain_min_rows) {
          const int c = (int)P.chain_m.size();
          int col0 = 0;
          for (size_t t = i; t < j; ++t) {
            P.chain_piv.push_back(ch[t]); P.chain_col0.push_back(col0); col0 += P.piv_w[ch[t]]; P.piv_chain[ch[t]] = c;
          }
          P.chain_ptr.push_back((int)P.chain_piv.size());
          P.chain_m.push_back((int)m0); P.chain_w.push_back(W); P.chain_level.push_back(0);
        }
        i = j;
      }
    }
    // factor levels: the panels of a front share the level at which everything OUTSIDE the front that any of them
    // gathers from is complete.  (A pivot that depends on a member is a later member or follows the last one, and a source
    // of a member from outside precedes it: pivots in ascending order see every level they need.)
    const int nfr = (int)P.chain_m.size();
    if (nfr > 0) {
      std::vector<int>& fl = P.piv_flevel;
      for (int p = 0; p < P.npiv; ++p) {
        const int c = P.piv_chain[p];
        int lv = 0;
        for (auto& km : rowpat[p]) {
          if (c >= 0 && P.piv_chain[km.k] == c) continue;
          if (fl[km.k] < 0) { P.error = "internal: chain front level order"; return 3; }
          lv = std::max(lv, fl[km.k] + 1);
        }
        if (c < 0) { fl[p] = lv; continue; }
        P.chain_level[c] = std::max(P.chain_level[c], lv);
        fl[p] = -1;
        if (p == P.chain_piv[P.chain_ptr[c + 1] - 1])
          for (int t = P.chain_ptr[c]; t < P.chain_ptr[c + 1]; ++t) fl[P.chain_piv[t]] = P.chain_level[c];
      }
      // fronts in level order
      std::vector<int> ord(nfr);
      for (int c = 0; c < nfr; ++c) ord[c] = c;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return P.chain_level[a] < P.chain_level[b]; });
      std::vector<int> nptr(1, 0), npiv, ncol0, nm, nw, nlv;
      for (int c2 = 0; c2 < nfr; ++c2) {
        const int c = ord[c2];
        for (int t = P.chain_ptr[c]; t < P.chain_ptr[c + 1]; ++t) {
          npiv.push_back(P.chain_piv[t]); ncol0.push_back(P.chain_col0[t]); P.piv_chain[P.chain_piv[t]] = c2;
        }
        nptr.push_back((int)npiv.size());
        nm.push_back(P.chain_m[c]); nw.push_back(P.chain_w[c]); nlv.push_back(P.chain_level[c]);
      }
      P.chain_ptr = nptr; P.chain_piv = npiv; P.chain_col0 = ncol0; P.chain_m = nm; P.chain_w = nw; P.chain_level = nlv;
    }
  }
  P.n_flevels = 0;
  for (int p = 0; p < P.npiv; ++p) P.n_flevels = std::max(P.n_flevels, P.piv_flevel[p] + 1);
  P.chain_lvl_ptr.assign(P.n_levels + 1, 0);
  for (size_t c = 0; c < P.chain_level.size(); ++c) P.chain_lvl_ptr[P.chain_level[c] + 1]++;
  for (int l = 0; l < P.n_levels; ++l) P.chain_lvl_ptr[l + 1] += P.chain_lvl_ptr[l];

  // tail = trailing run of levels that each hold few pivots
  P.tail_level0 = P.n_levels;
  for (int l = P.n_levels - 1; l >= 0; --l) {
    if (P.lvl_ptr[l + 1] - P.lvl_ptr[l] > opt.tail_piv_max) break;
    P.tail_level0 = l;
  }
  if (P.n_levels - P.tail_level0 < 3) P.tail_level0 = P.n_levels;

  // ---- 6. factor tasks in L form (see plan.hpp): pure (U, L) gathers, fused small panels,
  // gather + scale chunks for big panels
  {
    std::vector<std::pair<int64_t, int>> can_by_pos(P.ncan);
    for (int e = 0; e < P.ncan; ++e) can_by_pos[e] = {P.pos_of_can[e], e};
    std::sort(can_by_pos.begin(), can_by_pos.end());
    size_t can_cursor = 0;  // panels are visited in increasing U position
    struct TmpTask { FTask t; int level; int nent = 0; };
    std::vector<TmpTask> gtasks, sctasks, tgtasks;
    std::vector<std::vector<FEntry>> row_ents;   // per destination row (slot) of panel p (of its current column slice)
    // tile tasks: the sources of the current slice -- which of its rows hold which destination slots, which hold the columns
    struct SrcInfo { int k, wk; int lrow[PP_WMAX]; std::vector<std::pair<int, int>> rows; };   // rows: (destination slot, row of k)
    std::vector<SrcInfo> srcs;
    std::vector<std::vector<int>> row_src;       // per entry of row_ents: its source (index into srcs), -1 for an initial value
    std::vector<std::pair<int64_t, int>> pan_can; // canonical entries located in panel p: (position in the panel, entry)
    for (int p = 0; p < P.npiv; ++p) {
      const int w = P.piv_w[p], p0 = P.piv_start[p];
      const int f = w + (int)rows[p].size();
      const int64_t u0 = P.piv_uoff[p], u1 = u0 + (int64_t)f * w;
      pan_can.clear();
      while (can_cursor < can_by_pos.size() && can_by_pos[can_cursor].first < u1) {
        if (can_by_pos[can_cursor].first >= u0) pan_can.push_back({can_by_pos[can_cursor].first - u0, can_by_pos[can_cursor].second});
        ++can_cursor;
      }
      // a wide panel (the root front) is gathered in column slices of PP_WMAX: the tasks of a slice are ordinary
      // gather tasks that store at (row * w + qoff + q); every other panel is one slice
      const bool wide = w > PP_WMAX;
      const int nslice = wide ? (w + PP_WMAX - 1) / PP_WMAX : 1;
      const bool in_tail = P.piv_level[p] >= P.tail_level0;
      const int cap_e = in_tail ? opt.tail_task_entries : opt.max_task_entries;
      bool fused = false;
      const int chain = P.piv_chain[p];         // member of a chain front: contributions of earlier members are the front kernel's
      for (int sl = 0; sl < nslice; ++sl) {
        const int qoff = sl * PP_WMAX, ws = wide ? std::min(PP_WMAX, w - qoff) : w;
        row_ents.assign((size_t)f, {});
        // tile tasks: always for the panels of a chain front (dense by construction); for any other panel (tile_panels)
        // when they request clearly fewer operands than the row tasks would (decided below, once the entries are known)
        const bool tile_sure = chain >= 0 && opt.chain_tiles && !wide;
        bool tiled = tile_sure || (opt.tile_panels && !wide);
        if (tiled) { row_src.assign((size_t)f, {}); srcs.clear(); }
        int64_t total = 0;
        // initial values
        for (auto& pc : pan_can) {
          const int col = (int)(pc.first % w);
          if (col < qoff || col >= qoff + ws) continue;
          row_ents[(size_t)(pc.first / w)].push_back({-1 - pc.second, -1, 0, col - qoff});
          if (tiled) row_src[(size_t)(pc.first / w)].push_back(-1);
          ++total;
        }
        for (auto& km : rowpat[p]) {
          const int k = km.k, mslot = km.mslot, wk = P.piv_w[k];
          if (chain >= 0 && P.piv_chain[k] == chain) {       // (inside the front: every row of p, all its columns)
            P.flops_factor += (int64_t)f * w * wk;
            continue;
          }
          int j0 = 0, j1;
          while (j0 < km.cnt && km.qs[j0] < qoff) ++j0;
          j1 = j0;
          while (j1 < km.cnt && km.qs[j1] < qoff + ws) ++j1;
          if (j1 == j0) continue;                  // panel k holds no column of this slice
          int nruns = 0;
          const auto& rk = rows[k];
          size_t tp = 0;
          if (tiled) {
            SrcInfo si; si.k = k; si.wk = wk;
            for (int q = 0; q < PP_WMAX; ++q) si.lrow[q] = -1;
            for (int j = j0; j < j1; ++j) si.lrow[km.qs[j] - qoff] = mslot + j;
            srcs.push_back(si);
          }
          for (size_t t = (size_t)(mslot - wk); t < rk.size(); ++t) {
            const int r = rk[t];
            int d;
            if (r < p0 + w) d = r - p0;
            else {
              while (tp < rows[p].size() && rows[p][tp] < r) ++tp;
              if (tp >= rows[p].size() || rows[p][tp] != r) { P.error = "internal: fill closure violated"; return 3; }
              d = w + (int)tp;
            }
            const int srow = wk + (int)t;
            if (tiled) srcs.back().rows.push_back({d, srow});
            for (int tt = 0; tt < wk; ++tt) {
              const int upos = (int)(P.piv_uoff[k] + (int64_t)srow * wk + tt);
              // one entry per run of consecutive columns of the slice that panel k holds (consecutive rows of k)
              for (int j = j0; j < j1;) {
                int je = j + 1;
                while (je < j1 && km.qs[je] == km.qs[je - 1] + 1) ++je;
                row_ents[(size_t)d].push_back({upos, (int)(P.piv_uoff[k] + (int64_t)(mslot + j) * wk + tt), wk,
                                               (km.qs[j] - qoff) | ((je - j) << 4)});
                if (tiled) row_src[(size_t)d].push_back((int)srcs.size() - 1);
                ++nruns;
                j = je;
              }
            }
            P.flops_factor += (int64_t)wk * (j1 - j0);
            total += nruns;
            nruns = 0;
          }
        }
        auto emit = [&](int r0, int r1, int kind, int piece = 0, int npieces = 1, int e0 = -1, int e1 = -1) {
          TmpTask tt;
          tt.level = P.piv_flevel[p];
          tt.t.piv = p; tt.t.r0 = r0; tt.t.r1 = r1; tt.t.kind = kind; tt.t.dptr0 = (int)P.fdst_ptr.size();
          tt.t.piece = piece; tt.t.npieces = npieces; tt.t.qoff = qoff; tt.t.ws = ws;
          const int first = (int)P.fentries.size();
          if (e0 >= 0) {                                   // entries [e0, e1) of the single row r0
            P.fdst_ptr.push_back(first);
            const auto& de = row_ents[(size_t)r0];
            P.fentries.insert(P.fentries.end(), de.begin() + e0, de.begin() + e1);
          } else {
            for (int rr = r0; rr < r1; ++rr) {
              P.fdst_ptr.push_back((int)P.fentries.size());
              const auto& de = row_ents[(size_t)rr];
              P.fentries.insert(P.fentries.end(), de.begin(), de.end());
            }
          }
          P.fdst_ptr.push_back((int)P.fentries.size());
          // a piece counts with its whole row: a level that holds split rows must not take the lean (one wave per
          // task, no combine) kernel
          tt.nent = (e0 >= 0) ? (int)row_ents[(size_t)r0].size() : (int)P.fentries.size() - first;
          gtasks.push_back(tt);
        };
        if (tiled && !tile_sure) {
          // operand requests of the row tasks (1 + columns per entry) against those of tile tasks (8 per source column of
          // every (tile, source panel) pair); small panels stay with the row tasks (fused with their scaling, or one launch)
          int64_t row_loads = 0, tile_loads = 0, prod = 0;
          for (int rr = 0; rr < f; ++rr)
            for (auto& fe : row_ents[(size_t)rr])
              if (fe.u >= 0) { row_loads += 1 + ((fe.q >> 4) & 15); ++prod; }
          for (auto& si : srcs) {
            int last_tile = -1;
            for (auto& rw : si.rows) {
              const int tl = rw.first / PP_TILE_ROWS;
              if (tl != last_tile) { tile_loads += 8 * (int64_t)si.wk; last_tile = tl; }
            }
          }
          tiled = prod >= opt.tile_min_entries && (double)tile_loads <= opt.tile_load_ratio * (double)row_loads;
        }
        if (tiled) {
          // tile tasks: PP_TILE_ROWS consecutive slots against the source panels that hold any of them, cut into pieces
          // of tile_task_records sources; every piece carries the entries of its rows from its own sources (piece 0 the
          // initial values as well)
          std::vector<int> ts;
          for (int r0 = 0; r0 < f; r0 += PP_TILE_ROWS) {
            const int r1 = std::min(f, r0 + PP_TILE_ROWS);
            ts.clear();
            for (size_t si = 0; si < srcs.size(); ++si) {
              const auto& rw = srcs[si].rows;
              auto it = std::lower_bound(rw.begin(), rw.end(), std::make_pair(r0, -1));
              if (it != rw.end() && it->first < r1) ts.push_back((int)si);
            }
            const int nrec = (int)ts.size();
            const int np = std::max(1, std::min(PP_QUAD, (nrec + opt.tile_task_records - 1) / std::max(1, opt.tile_task_records)));
            const int per = (nrec + np - 1) / np;
            for (int j = 0; j < np; ++j) {
              const int a = std::min(nrec, j * per), b = std::min(nrec, (j + 1) * per);
              TmpTask tt;
              tt.level = P.piv_flevel[p];
              tt.t.piv = p; tt.t.r0 = r0; tt.t.r1 = r1; tt.t.kind = 4; tt.t.dptr0 = (int)P.fdst_ptr.size();
              tt.t.piece = j; tt.t.npieces = np; tt.t.qoff = qoff; tt.t.ws = ws;
              tt.t.te0 = (int)(P.trec.size() / PP_TREC_INTS);
              for (int x = a; x < b; ++x) {
                const SrcInfo& s = srcs[(size_t)ts[(size_t)x]];
                int rec[PP_TREC_INTS] = {s.wk, -1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0};
                auto it = std::lower_bound(s.rows.begin(), s.rows.end(), std::make_pair(r0, -1));
                for (; it != s.rows.end() && it->first < r1; ++it)
                  rec[1 + it->first - r0] = (int)(P.piv_uoff[s.k] + (int64_t)it->second * s.wk);
                for (int q = 0; q < ws; ++q)
                  if (s.lrow[q] >= 0) rec[5 + q] = (int)(P.piv_uoff[s.k] + (int64_t)s.lrow[q] * s.wk);
                P.trec.insert(P.trec.end(), rec, rec + PP_TREC_INTS);
              }
              tt.t.te1 = (int)(P.trec.size() / PP_TREC_INTS);
              const int s_lo = (a < b) ? ts[(size_t)a] : 0, s_hi = (a < b) ? ts[(size_t)b - 1] : -1;
              const int first = (int)P.fentries.size();
              for (int rr = r0; rr < r1; ++rr) {
                P.fdst_ptr.push_back((int)P.fentries.size());
                const auto& de = row_ents[(size_t)rr];
                const auto& ds = row_src[(size_t)rr];
                for (size_t e = 0; e < de.size(); ++e)
                  if ((ds[e] < 0 && j == 0) || (ds[e] >= s_lo && ds[e] <= s_hi)) P.fentries.push_back(de[e]);
              }
              P.fdst_ptr.push_back((int)P.fentries.size());
              tt.nent = (int)P.fentries.size() - first;
              tgtasks.push_back(tt);
            }
          }
        } else if (!wide && chain < 0 && total <= opt.fuse_task_entries) {
          emit(0, f, 1);                                   // fused small panel
          fused = true;
        } else {
          const int cap_row = std::max(cap_e, (int)(opt.row_split_factor * cap_e));
          int r = 0;
          while (r < f) {                                  // gather chunks over all slots
            const int sz = (int)row_ents[(size_t)r].size();
            if (sz > cap_row) {                            // long row: pieces for the waves of one quad
              const int np = std::min(PP_QUAD, (sz + cap_e - 1) / cap_e), per = (sz + np - 1) / np;
              for (int j = 0; j < np; ++j) emit(r, r + 1, 0, j, np, std::min(sz, j * per), std::min(sz, (j + 1) * per));
              ++r;
              continue;
            }
            int nent = 0, r_end = r;
            while (r_end < f) {
              const int add = (int)row_ents[(size_t)r_end].size();
              if (add > cap_row || (r_end > r && nent + add > cap_e)) break;
              nent += add;
              ++r_end;
            }
            emit(r, r_end, 0);
            r = r_end;
          }
        }
      }
      if (!fused && chain < 0) {
        // scale chunks over the rows below the block (the root front: chunks for the workgroups of k_scale_wide)
        const int R = wide ? opt.front_scale_rows : std::max(1, opt.scale_task_rows / w);
        for (int r0 = w; r0 < f || r0 == w; r0 += R) {
          TmpTask tt;
          tt.level = P.piv_flevel[p];
          tt.t.piv = p; tt.t.r0 = r0; tt.t.r1 = std::min(f, r0 + R); tt.t.kind = wide ? 3 : 2; tt.t.dptr0 = -1;
          tt.t.ws = w;
          if (wide) P.wtasks.push_back(tt.t); else sctasks.push_back(tt);
          if (r0 + R >= f) break;
        }
      }
    }
    auto by_level = [](const TmpTask& a, const TmpTask& b) { return a.level < b.level; };
    std::stable_sort(gtasks.begin(), gtasks.end(), by_level);
    std::stable_sort(sctasks.begin(), sctasks.end(), by_level);
    P.flevel_ptr.assign(P.n_levels + 1, 0);
    P.slevel_ptr.assign(P.n_levels + 1, 0);
    P.flevel_maxent.assign(P.n_levels, 0);
    P.flevel_nsplit.assign(P.n_levels, 0);
    // per level: split rows first (each fills one quad, padded with no-ops), then the other tasks packed PP_QUAD
    // per quad; the level's task count is a whole number of quads
    {
      FTask noop; noop.piv = 0; noop.r0 = 0; noop.r1 = 0; noop.dptr0 = 0; noop.kind = -1;
      size_t i = 0;
      while (i < gtasks.size()) {
        const int lvl = gtasks[i].level;
        size_t j = i;
        while (j < gtasks.size() && gtasks[j].level == lvl) ++j;
        const size_t before = P.ftasks.size();
        for (size_t q = i; q < j; ++q) {
          if (gtasks[q].t.npieces <= 1) continue;
          P.ftasks.push_back(gtasks[q].t);
          if (gtasks[q].t.piece == 0) P.flevel_nsplit[lvl]++;
          if (gtasks[q].t.piece == gtasks[q].t.npieces - 1)
            for (int pad = gtasks[q].t.npieces; pad < PP_QUAD; ++pad) {
              FTask z = noop; z.npieces = gtasks[q].t.npieces; z.piece = pad;   // (keeps the quad uniform: all split)
              P.ftasks.push_back(z);
            }
        }
        // unsplit tasks longest first: the waves of a quad then carry similar loads, and the longest tasks of the
        // level -- which decide when the launch ends -- start first
        std::vector<size_t> singles;
        for (size_t q = i; q < j; ++q)
          if (gtasks[q].t.npieces <= 1) singles.push_back(q);
        if (opt.task_order == 0)
          std::stable_sort(singles.begin(), singles.end(), [&](size_t a, size_t b) { return gtasks[a].nent > gtasks[b].nent; });
        else
          std::stable_sort(singles.begin(), singles.end(), [&](size_t a, size_t b) {
            if (gtasks[a].t.piv != gtasks[b].t.piv) return gtasks[a].t.piv < gtasks[b].t.piv;
            return gtasks[a].t.r0 < gtasks[b].t.r0;
          });
        for (size_t q : singles) P.ftasks.push_back(gtasks[q].t);
        while ((P.ftasks.size() - before) % PP_QUAD != 0) P.ftasks.push_back(noop);
        P.flevel_ptr[lvl + 1] += (int)(P.ftasks.size() - before);
        for (size_t q = i; q < j; ++q) P.flevel_maxent[lvl] = std::max(P.flevel_maxent[lvl], gtasks[q].nent);
        i = j;
      }
    }
    for (auto& t : sctasks) { P.stasks.push_back(t.t); P.slevel_ptr[t.level + 1]++; }
    // tile tasks by level: split tiles fill one quad each (padded), the others are packed PP_QUAD per quad
    P.tlevel_ptr.assign(P.n_levels + 1, 0);
    {
      std::stable_sort(tgtasks.begin(), tgtasks.end(), by_level);
      FTask noop; noop.piv = 0; noop.r0 = 0; noop.r1 = 0; noop.dptr0 = 0; noop.kind = -1;
      size_t i = 0;
      while (i < tgtasks.size()) {
        const int lvl = tgtasks[i].level;
        size_t j = i;
        while (j < tgtasks.size() && tgtasks[j].level == lvl) ++j;
        const size_t before = P.ttasks.size();
        for (size_t q = i; q < j; ++q) {
          if (tgtasks[q].t.npieces <= 1) continue;
          P.ttasks.push_back(tgtasks[q].t);
          if (tgtasks[q].t.piece == tgtasks[q].t.npieces - 1)
            for (int pad = tgtasks[q].t.npieces; pad < PP_QUAD; ++pad) {
              FTask z = noop; z.npieces = tgtasks[q].t.npieces; z.piece = pad;
              P.ttasks.push_back(z);
            }
        }
        std::vector<size_t> singles;
        for (size_t q = i; q < j; ++q)
          if (tgtasks[q].t.npieces <= 1) singles.push_back(q);
        std::stable_sort(singles.begin(), singles.end(), [&](size_t a, size_t b) { return tgtasks[a].t.te1 - tgtasks[a].t.te0 > tgtasks[b].t.te1 - tgtasks[b].t.te0; });
        for (size_t q : singles) P.ttasks.push_back(tgtasks[q].t);
        while ((P.ttasks.size() - before) % PP_QUAD != 0) P.ttasks.push_back(noop);
        P.tlevel_ptr[lvl + 1] += (int)(P.ttasks.size() - before);
        i = j;
      }
      for (int l = 0; l < P.n_levels; ++l) P.tlevel_ptr[l + 1] += P.tlevel_ptr[l];
    }
    for (int l = 0; l < P.n_levels; ++l) { P.flevel_ptr[l + 1] += P.flevel_ptr[l]; P.slevel_ptr[l + 1] += P.slevel_ptr[l]; }
  }

  // ---- 7. solve schedules: independent scalar rows / columns by level (L form)
  {
    // forward: y_c = b_c - sum L[c, k-columns] * y[k-columns]
    P.sfwd_eptr.assign(n + 1, 0);
    for (int p = 0; p < P.npiv; ++p)
      for (int q = 0; q < P.piv_w[p]; ++q) {
        const int c = P.piv_start[p] + q;
        for (auto& km : rowpat[p]) {
          const int k = km.k, wk = P.piv_w[k];
          // (a column of a chain front gets the contributions of the front's earlier panels from the front's own sweep kernel)
          if (opt.chain_sweeps && P.piv_chain[p] >= 0 && P.piv_chain[k] == P.piv_chain[p]) continue;
          int j = 0;
          while (j < km.cnt && km.qs[j] != q) ++j;
          if (j == km.cnt) continue;            // panel k has no row for this column of p
          for (int t = 0; t < wk; ++t) {
            P.sfwd_upos.push_back((int)(P.piv_uoff[k] + (int64_t)(km.mslot + j) * wk + t));
            P.sfwd_zcol.push_back(P.piv_start[k] + t);
          }
        }
        P.sfwd_eptr[c + 1] = (int)P.sfwd_upos.size();
      }
    {
      // levels of the sweeps: the dependency levels -- or, with chain fronts whose own kernels run their panels one after
      // the other (chain_sweeps), the levels of the factor schedule: a front's columns share one.  Within a level the
      // columns outside fronts come first (the backward sweep launches only those: clevel_nchain counts the others).
      const bool cs = opt.chain_sweeps && !P.chain_m.empty();
      const std::vector<int>& lv = cs ? P.piv_flevel : P.piv_level;
      std::vector<int> cnt(P.n_levels + 1, 0);
      for (int p = 0; p < P.npiv; ++p) cnt[lv[p] + 1] += P.piv_w[p];
      for (int l = 0; l < P.n_levels; ++l) cnt[l + 1] += cnt[l];
      P.clevel_ptr = cnt;
      P.clevel_col.assign(n, 0);
      P.clevel_nchain.assign(P.n_levels, 0);
      std::vector<int> fill(cnt.begin(), cnt.end() - 1);
      for (int pass = 0; pass < 2; ++pass)
        for (int p = 0; p < P.npiv; ++p) {
          const bool in_front = cs && P.piv_chain[p] >= 0;
          if ((pass == 1) != in_front) continue;
          for (int q = 0; q < P.piv_w[p]; ++q) P.clevel_col[fill[lv[p]]++] = P.piv_start[p] + q;
          if (in_front) P.clevel_nchain[lv[p]] += P.piv_w[p];
        }
      P.chain_sweeps_on = cs;
    }
    std::vector<std::vector<std::pair<int, int>>> cr(nc);
    for (int k = 0; k < P.npiv; ++k) {
      const auto& r = rows[k];
      const int wk = P.piv_w[k];
      for (size_t t = r.size(); t-- > 0;) {
        if (r[t] < n) break;
        for (int tt = 0; tt < wk; ++tt)
          cr[r[t] - n].push_back({(int)(P.piv_uoff[k] + (int64_t)(wk + (int)t) * wk + tt), P.piv_start[k] + tt});
      }
    }
    P.crow_eptr.assign(nc + 1, 0);
    for (int c = 0; c < nc; ++c) {
      for (auto& e : cr[c]) { P.crow_upos.push_back(e.first); P.crow_zcol.push_back(e.second); }
      P.crow_eptr[c + 1] = (int)P.crow_upos.size();
    }
  }

  // ---- 8. Schur tiles: S -= U_c,p inv(P_p) U_c,p^T over coupling rows, T x T tiles
  {
    const int T = opt.tile;
    std::map<std::pair<int, int>, std::vector<STileRec>> by_tile;
    for (int p = 0; p < P.npiv; ++p) {
      const auto& r = rows[p];
      size_t first = r.size();
      while (first > 0 && r[first - 1] >= n) --first;
      if (first == r.size()) continue;
      // slot lists per touched tile row
      std::vector<int> tl;           // tile indices
      std::vector<STileRec> dummy;
      std::vector<std::vector<int>> slots;
      for (size_t t = first; t < r.size(); ++t) {
        int c = r[t] - n, ti = c / T;
        if (tl.empty() || tl.back() != ti) { tl.push_back(ti); slots.push_back(std::vector<int>(T, -1)); }
        slots.back()[c % T] = P.piv_w[p] + (int)t;
      }
      for (size_t a = 0; a < tl.size(); ++a)
        for (size_t b = 0; b <= a; ++b) {
          STileRec rec;
          rec.piv = p;
          for (int q = 0; q < T; ++q) { rec.slotA[q] = slots[a][q]; rec.slotB[q] = slots[b][q]; }
          by_tile[{tl[a], tl[b]}].push_back(rec);
          int na = 0, nb = 0;
          for (int q = 0; q < T; ++q) { na += slots[a][q] >= 0; nb += slots[b][q] >= 0; }
          P.flops_schur += (int64_t)na * nb * P.piv_w[p];
        }
    }
    P.stile_ptr.push_back(0);
    for (auto& kv : by_tile) {
      P.stile_a.push_back(kv.first.first);
      P.stile_b.push_back(kv.first.second);
      P.stile_rec.insert(P.stile_rec.end(), kv.second.begin(), kv.second.end());
      P.stile_ptr.push_back((int)P.stile_rec.size());
    }
  }
  return 0;
}

}  // namespace pp
