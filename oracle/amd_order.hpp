// oracle/amd_order.hpp -- TEST INFRASTRUCTURE (part of the CPU oracle; never linked by the product).
//
// The fill-reducing ordering Eigen::SimplicialLDLT applies by default (AMDOrdering) before it factors S^T at
// /root/reference/include/ekf_vio/TightlyCoupledEKF.cpp:577-578.  Eigen is a third-party dependency that is absent from
// /root/reference and from this image (version unpinned upstream, 3.2.9x / 3.3.x era), so this restates the published algorithm:
// Eigen/src/OrderingMethods/Amd.h `internal::minimum_degree_ordering` is a port of CSparse's cs_amd (T. Davis, "Direct Methods for
// Sparse Linear Systems", SIAM 2006, ch. 7; order = 1, i.e. the pattern of A + A^T) and this file follows that routine step by step:
// quotient graph with element absorption, approximate external degrees, mass elimination, hash-based supervariable detection, dense
// rows (degree > max(16, 10 sqrt(n)), capped at n - 2) ordered last, post-ordered assembly tree.
// Two things Eigen does around it, as its sources had them (SimplicialCholesky_impl.h `ordering`, Ordering.h `AMDOrdering`):
//   * the pattern handed over is that of the full symmetric matrix, row indices ascending in every column, and -- the
//     `prune(keep_diag())` that would drop the diagonal is commented out in Ordering.h -- WITH the diagonal entries
//     (keep_diagonal = true; false gives textbook cs_amd, listed as a variant in profiles/r06_oracle_variant_spread.txt);
//   * the routine's result P (P[k] = the k-th pivot's original index) becomes m_Pinv, the matrix factored is A(P, P).
// Parity unpinned against Eigen itself (no Eigen here); pinned by construction properties (tests/test_oracle_amd_cpu.py: a
// permutation; identity when every row is dense; a tree's leaves before its root; fill no worse than natural order on arrow matrices).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace ekf_oracle {

inline int amd_flip(int i) { return -i - 2; }

// pattern: n columns, col_ptr[n + 1], row_idx ascending per column, symmetric.  Returns P (size n).
inline std::vector<int> amd_order(int n, const std::vector<int>& col_ptr, const std::vector<int>& row_idx, bool keep_diagonal = true) {
    std::vector<int> P(n + 1, 0);
    if (n == 0) return {};
    // --- C: the pattern, with or without its diagonal, plus elbow room ---
    std::vector<int> Cp(n + 1), Ci;
    for (int j = 0; j < n; j++) {
        Cp[j] = (int)Ci.size();
        for (int p = col_ptr[j]; p < col_ptr[j + 1]; p++)
            if (keep_diagonal || row_idx[p] != j) Ci.push_back(row_idx[p]);
    }
    Cp[n] = (int)Ci.size();
    int cnz = Cp[n];
    int dense = std::max(16, (int)(10 * std::sqrt((double)n)));
    dense = std::min(n - 2, dense);
    const int t = cnz + cnz / 5 + 2 * n;
    Ci.resize((size_t)t + 1, 0);
    int nzmax = t;
    std::vector<int> len(n + 1), nv(n + 1), next(n + 1), head(n + 1), elen(n + 1), degree(n + 1), w(n + 1), hhead(n + 1);
    std::vector<int>& last = P;  // (P is the workspace for `last` until the post-ordering)
    // --- initialise the quotient graph ---
    for (int k = 0; k < n; k++) len[k] = Cp[k + 1] - Cp[k];
    len[n] = 0;
    for (int i = 0; i <= n; i++) {
        head[i] = -1;
        last[i] = -1;
        next[i] = -1;
        hhead[i] = -1;
        nv[i] = 1;
        w[i] = 1;
        elen[i] = 0;
        degree[i] = len[i];
    }
    auto wclear = [&](int mark, int lemax) {
        if (mark < 2 || (mark + lemax < 0)) {
            for (int k = 0; k < n; k++)
                if (w[k] != 0) w[k] = 1;
            mark = 2;
        }
        return mark;
    };
    int mark = wclear(0, 0);
    elen[n] = -2;
    Cp[n] = -1;
    w[n] = 0;
    int nel = 0, mindeg = 0, lemax = 0;
    // --- initialise the degree lists ---
    for (int i = 0; i < n; i++) {
        const int d = degree[i];
        if (d == 0) {  // empty node
            elen[i] = -2;
            nel++;
            Cp[i] = -1;
            w[i] = 0;
        } else if (d > dense) {  // dense node: absorbed into element n, ordered last
            nv[i] = 0;
            elen[i] = -1;
            nel++;
            Cp[i] = amd_flip(n);
            nv[n]++;
        } else {
            if (head[d] != -1) last[head[d]] = i;
            next[i] = head[d];
            head[d] = i;
        }
    }
    while (nel < n) {
        // --- select a node of minimum approximate degree ---
        int k;
        for (k = -1; mindeg < n && (k = head[mindeg]) == -1; mindeg++) {
        }
        if (next[k] != -1) last[next[k]] = -1;
        head[mindeg] = next[k];
        const int elenk = elen[k];
        int nvk = nv[k];
        nel += nvk;
        // --- garbage collection ---
        if (elenk > 0 && cnz + mindeg >= nzmax) {
            for (int j = 0; j < n; j++) {
                int p;
                if ((p = Cp[j]) >= 0) {
                    Cp[j] = Ci[p];
                    Ci[p] = amd_flip(j);
                }
            }
            int q = 0;
            for (int p = 0; p < cnz;) {
                int j;
                if ((j = amd_flip(Ci[p++])) >= 0) {
                    Ci[q] = Cp[j];
                    Cp[j] = q++;
                    for (int k3 = 0; k3 < len[j] - 1; k3++) Ci[q++] = Ci[p++];
                }
            }
            cnz = q;
        }
        // --- construct the new element ---
        int dk = 0;
        nv[k] = -nvk;
        int p = Cp[k];
        const int pk1 = (elenk == 0) ? p : cnz;
        int pk2 = pk1;
        for (int k1 = 1; k1 <= elenk + 1; k1++) {
            int e, pj, ln;
            if (k1 > elenk) {
                e = k;
                pj = p;
                ln = len[k] - elenk;
            } else {
                e = Ci[p++];
                pj = Cp[e];
                ln = len[e];
            }
            for (int k2 = 1; k2 <= ln; k2++) {
                const int i = Ci[pj++];
                int nvi;
                if ((nvi = nv[i]) <= 0) continue;  // dead, or seen already
                dk += nvi;
                nv[i] = -nvi;
                Ci[pk2++] = i;
                if (next[i] != -1) last[next[i]] = last[i];
                if (last[i] != -1) next[last[i]] = next[i];
                else head[degree[i]] = next[i];
            }
            if (e != k) {
                Cp[e] = amd_flip(k);
                w[e] = 0;
            }
        }
        if (elenk != 0) cnz = pk2;
        degree[k] = dk;
        Cp[k] = pk1;
        len[k] = pk2 - pk1;
        elen[k] = -2;
        // --- set differences ---
        mark = wclear(mark, lemax);
        for (int pk = pk1; pk < pk2; pk++) {
            const int i = Ci[pk];
            int eln;
            if ((eln = elen[i]) <= 0) continue;
            const int nvi = -nv[i];
            const int wnvi = mark - nvi;
            for (int pp = Cp[i]; pp <= Cp[i] + eln - 1; pp++) {
                const int e = Ci[pp];
                if (w[e] >= mark) w[e] -= nvi;
                else if (w[e] != 0) w[e] = degree[e] + wnvi;
            }
        }
        // --- degree update ---
        for (int pk = pk1; pk < pk2; pk++) {
            const int i = Ci[pk];
            const int p1 = Cp[i];
            const int p2 = p1 + elen[i] - 1;
            int pn = p1;
            long long h = 0;
            int d = 0;
            for (int pp = p1; pp <= p2; pp++) {
                const int e = Ci[pp];
                if (w[e] != 0) {
                    const int dext = w[e] - mark;
                    if (dext > 0) {
                        d += dext;
                        Ci[pn++] = e;
                        h += e;
                    } else {
                        Cp[e] = amd_flip(k);  // aggressive absorption
                        w[e] = 0;
                    }
                }
            }
            elen[i] = pn - p1 + 1;
            const int p3 = pn;
            const int p4 = p1 + len[i];
            for (int pp = p2 + 1; pp < p4; pp++) {
                const int j = Ci[pp];
                int nvj;
                if ((nvj = nv[j]) <= 0) continue;
                d += nvj;
                Ci[pn++] = j;
                h += j;
            }
            if (d == 0) {  // mass elimination
                Cp[i] = amd_flip(k);
                const int nvi = -nv[i];
                dk -= nvi;
                nvk += nvi;
                nel += nvi;
                nv[i] = 0;
                elen[i] = -1;
            } else {
                degree[i] = std::min(degree[i], d);
                Ci[pn] = Ci[p3];
                Ci[p3] = Ci[p1];
                Ci[p1] = k;
                len[i] = pn - p1 + 1;
                h = ((h < 0) ? (-h) : h) % n;
                next[i] = hhead[h];
                hhead[h] = i;
                last[i] = (int)h;
            }
        }
        degree[k] = dk;
        lemax = std::max(lemax, dk);
        mark = wclear(mark + lemax, lemax);
        // --- supervariable detection ---
        for (int pk = pk1; pk < pk2; pk++) {
            int i = Ci[pk];
            if (nv[i] >= 0) continue;
            const int h = last[i];
            i = hhead[h];
            hhead[h] = -1;
            for (; i != -1 && next[i] != -1; i = next[i], mark++) {
                const int ln = len[i];
                const int eln = elen[i];
                for (int pp = Cp[i] + 1; pp <= Cp[i] + ln - 1; pp++) w[Ci[pp]] = mark;
                int jlast = i;
                for (int j = next[i]; j != -1;) {
                    bool ok = (len[j] == ln) && (elen[j] == eln);
                    for (int pp = Cp[j] + 1; ok && pp <= Cp[j] + ln - 1; pp++)
                        if (w[Ci[pp]] != mark) ok = false;
                    if (ok) {
                        Cp[j] = amd_flip(i);
                        nv[i] += nv[j];
                        nv[j] = 0;
                        elen[j] = -1;
                        j = next[j];
                        next[jlast] = j;
                    } else {
                        jlast = j;
                        j = next[j];
                    }
                }
            }
        }
        // --- finalise the new element ---
        int pf = pk1;
        for (int pk = pk1; pk < pk2; pk++) {
            const int i = Ci[pk];
            int nvi;
            if ((nvi = -nv[i]) <= 0) continue;
            nv[i] = nvi;
            int d = degree[i] + dk - nvi;
            d = std::min(d, n - nel - nvi);
            if (head[d] != -1) last[head[d]] = i;
            next[i] = head[d];
            last[i] = -1;
            head[d] = i;
            mindeg = std::min(mindeg, d);
            degree[i] = d;
            Ci[pf++] = i;
        }
        nv[k] = nvk;
        if ((len[k] = pf - pk1) == 0) {
            Cp[k] = -1;
            w[k] = 0;
        }
        if (elenk != 0) cnz = pf;
    }
    // --- post-ordering of the assembly tree ---
    for (int i = 0; i < n; i++) Cp[i] = amd_flip(Cp[i]);
    for (int j = 0; j <= n; j++) head[j] = -1;
    for (int j = n; j >= 0; j--) {  // unordered nodes into their parents' lists
        if (nv[j] > 0) continue;
        next[j] = head[Cp[j]];
        head[Cp[j]] = j;
    }
    for (int e = n; e >= 0; e--) {  // elements into their parents' lists
        if (nv[e] <= 0) continue;
        if (Cp[e] != -1) {
            next[e] = head[Cp[e]];
            head[Cp[e]] = e;
        }
    }
    std::vector<int> post(n + 1, 0);
    std::vector<int>& stack = w;
    int kk = 0;
    for (int i = 0; i <= n; i++) {
        if (Cp[i] != -1) continue;
        int top = 0;  // depth-first search from root i (cs_tdfs)
        stack[0] = i;
        while (top >= 0) {
            const int pnode = stack[top];
            const int c = head[pnode];
            if (c == -1) {
                top--;
                post[kk++] = pnode;
            } else {
                head[pnode] = next[c];
                stack[++top] = c;
            }
        }
    }
    post.resize(n);
    return post;
}

}  // namespace ekf_oracle
