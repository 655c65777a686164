// The pairs of opposite perimeter points of the keyed scoring prefilter (mg_score.hip), their order in the bound table
// (mg_tables.hip: mg_score_pair_table / mg_score_pairs) and the plan of the prefilter's LDS reads -- one definition,
// evaluated at compile time on the device side (per radius) and at run time by the host table.
//
// A perimeter (the reference's midpoint walk, utils.py:433-465) consists of pairs (p, -p): both points have the same
// radial direction mod pi and share one bound table.  Round 4 order: first the pairs of the top / bottom arcs
// (|d_row| > |d_col|), grouped so that the (up to four) pairs whose first points lie in one 4-byte stretch T of a
// horizontal run -- and whose opposite points therefore lie in one stretch B of the mirrored run -- are consecutive;
// then the pairs of the side arcs in walk order.  A group costs the prefilter two 4-byte LDS reads instead of eight
// byte reads, and its table entries are consecutive scalar loads.
#pragma once

#include "mg_common.h"

struct MgPairPlan {
  int n;                                              // pairs
  int dr[MG_SCORE_MAX_PAIRS], dc[MG_SCORE_MAX_PAIRS];  // first point of pair k, in table order
  int n_h;                                            // pairs [0, n_h) belong to groups, [n_h, n) are read byte by byte
  int n_sg;                                           // groups
  int sg_first[MG_SCORE_MAX_PAIRS], sg_n[MG_SCORE_MAX_PAIRS];
  int sg_trow[MG_SCORE_MAX_PAIRS], sg_tcol[MG_SCORE_MAX_PAIRS];  // 4 bytes at (row, col .. col + 3): the first points
  int sg_urow[MG_SCORE_MAX_PAIRS], sg_ucol[MG_SCORE_MAX_PAIRS];  // ... the opposite points
  int tb[MG_SCORE_MAX_PAIRS], ub[MG_SCORE_MAX_PAIRS];            // byte of pair k's first / opposite point in T / B

  static constexpr int iabs(int v) { return v < 0 ? -v : v; }

  constexpr MgPairPlan(int r)
      : n(0), dr{}, dc{}, n_h(0), n_sg(0), sg_first{}, sg_n{}, sg_trow{}, sg_tcol{}, sg_urow{}, sg_ucol{}, tb{}, ub{} {
    // the walk
    int wr[MG_SCORE_MAX_PAIRS] = {}, wc[MG_SCORE_MAX_PAIRS] = {}, wn = 0;
    auto put = [&](int a, int b) {
      if (wn < MG_SCORE_MAX_PAIRS) wr[wn] = a, wc[wn] = b;
      ++wn;
    };
    put(0, -r);
    put(-r, 0);
    int x = 1, y = -r;
    while (x < -y) {
      put(x, y);
      put(y, x);
      put(-x, y);
      put(-y, x);
      if (x * x + y * y - r * r <= 0) {
        ++x;
      } else {
        ++y;
        ++x;
      }
    }
    if (y == -x) {
      put(x, y);
      put(-x, y);
    }
    if (wn > MG_SCORE_MAX_PAIRS) {  // (radii beyond MG_SCORE_MAX_R: no plan)
      n = wn;
      return;
    }
    // is (row, col) a perimeter point of the top / bottom arcs?
    auto on_arc = [&](int row, int col) {
      if (iabs(row) <= iabs(col)) return false;
      for (int k = 0; k < wn; ++k)
        if ((wr[k] == row && wc[k] == col) || (-wr[k] == row && -wc[k] == col)) return true;
      return false;
    };
    // the 4-byte stretch of its run that holds (row, col): first column and byte
    auto stretch = [&](int row, int col, int& start, int& byte) {
      int c0 = col;
      while (on_arc(row, c0 - 1)) --c0;
      start = c0 + 4 * ((col - c0) / 4);
      byte = (col - c0) % 4;
    };
    bool placed[MG_SCORE_MAX_PAIRS] = {};
    for (int k = 0; k < wn; ++k) {
      if (placed[k] || iabs(wr[k]) <= iabs(wc[k])) continue;
      int ts = 0, tbyte = 0, us = 0, ubyte = 0;
      stretch(wr[k], wc[k], ts, tbyte);
      stretch(-wr[k], -wc[k], us, ubyte);
      const int sg = n_sg++;
      sg_first[sg] = n;
      sg_trow[sg] = wr[k], sg_tcol[sg] = ts, sg_urow[sg] = -wr[k], sg_ucol[sg] = us;
      for (int j = k; j < wn; ++j) {  // every pair with the same two stretches (k itself first)
        if (placed[j] || iabs(wr[j]) <= iabs(wc[j]) || wr[j] != wr[k]) continue;
        int ts2 = 0, tb2 = 0, us2 = 0, ub2 = 0;
        stretch(wr[j], wc[j], ts2, tb2);
        stretch(-wr[j], -wc[j], us2, ub2);
        if (ts2 != ts || us2 != us) continue;
        placed[j] = true;
        dr[n] = wr[j], dc[n] = wc[j], tb[n] = tb2, ub[n] = ub2;
        ++n;
        ++sg_n[sg];
      }
    }
    n_h = n;
    for (int k = 0; k < wn; ++k)
      if (!placed[k]) {
        dr[n] = wr[k], dc[n] = wc[k];
        ++n;
      }
  }
};
