"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the rectangular linear-sum-assignment solver the
reference's matcher calls (criterion.py:19,207 -> scipy.optimize.linear_sum_assignment).

The algorithm lives in a third-party dependency that is not under /root/reference: scipy, pinned ``scipy==1.5.1`` in the
reference's requirements.txt:9 (scipy/optimize/rectangular_lsap/rectangular_lsap.cpp).  Its published algorithm is the
shortest-augmenting-path method of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4),
2016, with two implementation rules that decide the result when several assignments share the optimal cost:

  * the set of unscanned columns is the array ``remaining`` filled in REVERSE order (remaining[it] = nc-1-it) and a column
    is removed by moving the last entry into its slot;
  * the scan keeps the first column that reaches a new strict minimum of the shortest-path cost, and on an exact tie
    replaces it by a later column only if that column is unassigned (row4col == -1).

Both are restated below (``index`` selection) because the reference's targets repeat every ground-truth box ``repeat_num``
= 5 times (criterion.py:513-600): duplicated columns make cost ties structural, not accidental.

PINNED against the scipy installed in this image (1.15.3; same algorithm as 1.5.1): tests/test_oracle_criterion.py compares
the two on random, integer (tie-heavy), constant, duplicated-column, tall, wide and empty matrices, row for row.
"""
import numpy as np


def linear_sum_assignment(cost, count_steps=False):
    """Returns (row_ind, col_ind) exactly as scipy.optimize.linear_sum_assignment(cost) (minimisation).  With
    ``count_steps`` also the number of inner scans (one scan = one pass over the remaining columns)."""
    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    if nr == 0 or nc == 0:
        e = np.zeros(0, dtype=np.int64)
        return (e, e, 0) if count_steps else (e, e)
    transpose = nc < nr          # a tall matrix is solved transposed
    if transpose:
        cost = np.ascontiguousarray(cost.T)
        nr, nc = nc, nr
    if np.isnan(cost).any() or np.isneginf(cost).any():
        raise ValueError("matrix contains invalid numeric entries")
    u = np.zeros(nr)
    v = np.zeros(nc)
    path = np.full(nc, -1, dtype=np.int64)
    col4row = np.full(nr, -1, dtype=np.int64)
    row4col = np.full(nc, -1, dtype=np.int64)
    steps = 0
    for cur_row in range(nr):
        # ---- shortest augmenting path from cur_row
        min_val = 0.0
        remaining = np.arange(nc - 1, -1, -1, dtype=np.int64)
        num_remaining = nc
        sr = []
        sc = []
        spc = np.full(nc, np.inf)
        i = cur_row
        sink = -1
        while sink == -1:
            steps += 1
            sr.append(i)
            rem = remaining[:num_remaining]
            r = ((min_val + cost[i, rem]) - u[i]) - v[rem]          # same evaluation order as the C++ expression
            better = r < spc[rem]
            path[rem[better]] = i
            spc[rem[better]] = r[better]
            cand = spc[rem]
            lowest = cand.min()
            if lowest == np.inf:
                raise ValueError("cost matrix is infeasible")
            ties = np.nonzero(cand == lowest)[0]
            free = ties[row4col[rem[ties]] == -1]
            index = free[-1] if free.size else ties[0]
            min_val = lowest
            j = rem[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            sc.append(j)
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        # ---- dual update
        u[cur_row] += min_val
        for i in sr:
            if i != cur_row:
                u[i] += min_val - spc[col4row[i]]
        sc = np.asarray(sc, dtype=np.int64)
        v[sc] -= min_val - spc[sc]
        # ---- augment
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur_row:
                break
    if transpose:
        order = np.argsort(col4row, kind="stable")
        a, b = col4row[order], order.astype(np.int64)
    else:
        a, b = np.arange(nr, dtype=np.int64), col4row
    return (a, b, steps) if count_steps else (a, b)
