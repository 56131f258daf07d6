// cluster.cpp -- agglomerative clustering merge order of the guide tree (praline/util/cluster.py:27-114), host code.
//
// The reference rebuilds the whole cluster-by-cluster linkage table from the N x N distances every round (O(N^4) element
// reads over a run).  Here, as in component.merge_order (the numpy statement of the same algorithm, which the tests
// compare this with), the table lives across rounds and a merge touches one row and one column: min / max of the two old
// rows for single / complete linkage, and for average linkage the float64 SUMS of the member distances divided by the
// member count (what `a.mean()` evaluates, cluster.py:99-114); the first minimum in cluster-id order is found from
// per-row (value, column) minima that are recomputed only for the rows whose minimum pointed at a merged cluster.
// N = 4096 (BASELINE C4): 2.9 s in numpy, tens of milliseconds here - next to a 0.44 s distance stage.
// No HIP in here: also built into the scheduler test library for the CPU tests.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

// The tables are stored in 64 x 64 tiles: a merge rewrites one row AND one column, and in a row-major N x N float64
// matrix a column of N = 4096 elements touches 4096 different pages (the first version spent 2.2 s on TLB misses, as
// numpy does); tiled, a column touches N / 64 tiles of 8 pages each.
namespace {
constexpr size_t TS = 64;
struct Tiled {
    size_t nt = 0;               // tiles per side
    std::vector<double> v;
    void init(size_t n_pad, double fill) { nt = n_pad / TS; v.assign(n_pad * n_pad, fill); }
    double &at(size_t i, size_t j) { return v[((i / TS) * nt + j / TS) * (TS * TS) + (i % TS) * TS + (j % TS)]; }
    double *row_tile(size_t i, size_t tj) { return &v[((i / TS) * nt + tj) * (TS * TS) + (i % TS) * TS]; }       // 64 contiguous
    double *col_tile(size_t ti, size_t j) { return &v[(ti * nt + j / TS) * (TS * TS) + (j % TS)]; }                 // stride TS
};
}  // namespace

// SYMMETRIC distances (what the guide tree produces, tree.py:142-145): the linkage table stays symmetric, the new row
// IS the new column, and no column is ever written: a row is rewritten, whole and contiguous, only when its cluster
// absorbs another one; entry (r, c) of an older row r is stale exactly for the clusters c that changed after r was
// written, and the fresh value then sits in row c (r has not changed since, or it would have been rewritten).  Rows are
// freshened from that event list before they are read.  Same values, same first-minimum order as the general version.
static int merge_order_symmetric(size_t N, const double *dist, int linkage, int32_t *order)
{
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> link(dist, dist + N * N), sums;
    for (size_t i = 0; i < N; ++i) link[i * N + i] = inf;
    if (linkage == 2) sums.assign(dist, dist + N * N);
    std::vector<double> size(N, 1.0), row_val(N), new_row(N);
    std::vector<char> alive(N, 1);
    std::vector<int64_t> row_col(N), written(N, -1);
    std::vector<int32_t> events;   // events[r] = the cluster that absorbed another one in round r
    events.reserve(N);
    auto freshen = [&](size_t r) {
        for (size_t e = (size_t)(written[r] + 1); e < events.size(); ++e) {
            const size_t c = (size_t)events[e];
            if (!alive[c] || c == r) continue;
            link[r * N + c] = link[c * N + r];
            if (linkage == 2) sums[r * N + c] = sums[c * N + r];
        }
        written[r] = (int64_t)events.size() - 1;
    };
    auto rescan = [&](size_t i) {
        freshen(i);
        const double *r = &link[i * N];
        double v = inf;
        int64_t c = -1;
        for (size_t j = 0; j < N; ++j) {
            const double x = (alive[j] && j != i) ? r[j] : inf;
            if (c < 0 || x < v) { v = x; c = (int64_t)j; }   // first minimum of the row = lowest cluster id
        }
        row_val[i] = v;
        row_col[i] = c;
    };
    for (size_t i = 0; i < N; ++i) rescan(i);
    for (size_t round = 0; round + 1 < N; ++round) {
        size_t one = 0;
        for (size_t i = 1; i < N; ++i)
            if (row_val[i] < row_val[one]) one = i;
        const size_t two = (size_t)row_col[one];
        order[2 * round] = (int32_t)one;
        order[2 * round + 1] = (int32_t)two;
        freshen(one);
        freshen(two);
        if (linkage == 2) {
            double *a = &sums[one * N];
            const double *b = &sums[two * N];
            for (size_t j = 0; j < N; ++j) a[j] += b[j];
            a[one] += a[two];   // the column half of the numpy statement at the one entry both halves touch (masked below)
            size[one] += size[two];
            for (size_t j = 0; j < N; ++j) new_row[j] = a[j] / (size[one] * size[j]);
        } else {
            const double *a = &link[one * N], *b = &link[two * N];
            for (size_t j = 0; j < N; ++j) new_row[j] = linkage == 0 ? std::fmin(a[j], b[j]) : std::fmax(a[j], b[j]);
        }
        alive[two] = 0;
        for (size_t j = 0; j < N; ++j)
            if (!alive[j]) new_row[j] = inf;
        new_row[one] = inf;
        for (size_t j = 0; j < N; ++j) link[one * N + j] = new_row[j];
        events.push_back((int32_t)one);
        written[one] = (int64_t)events.size() - 1;
        row_val[two] = inf;
        for (size_t i = 0; i < N; ++i) {
            if (!alive[i]) continue;
            if (i == one) { rescan(i); continue; }
            if (row_col[i] == (int64_t)one || row_col[i] == (int64_t)two) {
                // single linkage: the new entry min(old (i, one), old (i, two)) is the old row minimum or smaller, and
                // one < two (the first row holding the global minimum): the row's first minimum is now at column one
                if (linkage == 0) { row_val[i] = new_row[i]; row_col[i] = (int64_t)one; }
                else rescan(i);
                continue;
            }
            const double v = new_row[i];
            if (v < row_val[i] || (v == row_val[i] && (int64_t)one < row_col[i])) { row_val[i] = v; row_col[i] = (int64_t)one; }
        }
    }
    return 0;
}

extern "C" int praline_merge_order(int64_t n, const double *dist, int linkage /* 0 single, 1 complete, 2 average */,
                                   int32_t *order /* [n - 1][2] */)
{
    if (n < 0 || (n > 0 && !dist) || (n > 1 && !order) || linkage < 0 || linkage > 2) return -1;
    if (n < 2) return 0;
    {
        bool symmetric = true;   // compared tile by tile: a plain column walk of the transpose misses the cache on every element
        for (int64_t i0 = 0; i0 < n && symmetric; i0 += 64)
            for (int64_t j0 = i0; j0 < n && symmetric; j0 += 64)
                for (int64_t i = i0; i < std::min<int64_t>(i0 + 64, n) && symmetric; ++i)
                    for (int64_t j = std::max(j0, i + 1); j < std::min<int64_t>(j0 + 64, n); ++j)
                        if (!(dist[i * n + j] == dist[j * n + i])) { symmetric = false; break; }
        if (symmetric) return merge_order_symmetric((size_t)n, dist, linkage, order);
    }
    const double inf = std::numeric_limits<double>::infinity();
    const size_t N = (size_t)n, NP = (N + TS - 1) / TS * TS, NT = NP / TS;
    Tiled link, sums;
    link.init(NP, inf);
    if (linkage == 2) sums.init(NP, 0.0);
    for (size_t i = 0; i < N; ++i)
        for (size_t j = 0; j < N; ++j) {
            if (i != j) link.at(i, j) = dist[i * N + j];
            if (linkage == 2) sums.at(i, j) = dist[i * N + j];
        }
    std::vector<double> size(NP, 1.0);
    std::vector<char> alive(NP, 0);
    for (size_t i = 0; i < N; ++i) alive[i] = 1;
    std::vector<double> row_val(NP, inf), new_row(NP), new_col(NP);
    std::vector<int64_t> row_col(NP, 0);
    auto rescan = [&](size_t i) {
        double v = inf;
        int64_t c = 0;
        bool first = true;
        for (size_t tj = 0; tj < NT; ++tj) {
            const double *r = link.row_tile(i, tj);
            for (size_t jj = 0; jj < TS; ++jj)
                if (first || r[jj] < v) { v = r[jj]; c = (int64_t)(tj * TS + jj); first = false; }   // first minimum = lowest id
        }
        row_val[i] = v;
        row_col[i] = c;
    };
    for (size_t i = 0; i < N; ++i) rescan(i);
    for (int64_t round = 0; round + 1 < n; ++round) {
        size_t one = 0;
        for (size_t i = 1; i < N; ++i)
            if (row_val[i] < row_val[one]) one = i;        // first row holding the smallest value
        const size_t two = (size_t)row_col[one];
        order[2 * round] = (int32_t)one;
        order[2 * round + 1] = (int32_t)two;
        if (linkage == 2) {
            // sums[one, :] += sums[two, :], THEN sums[:, one] += sums[:, two] (the order of the numpy statement)
            for (size_t tj = 0; tj < NT; ++tj) {
                double *a = sums.row_tile(one, tj);
                const double *b = sums.row_tile(two, tj);
                for (size_t jj = 0; jj < TS; ++jj) a[jj] += b[jj];
            }
            for (size_t ti = 0; ti < NT; ++ti) {
                double *a = sums.col_tile(ti, one);
                const double *b = sums.col_tile(ti, two);
                for (size_t ii = 0; ii < TS; ++ii) a[ii * TS] += b[ii * TS];
            }
            size[one] += size[two];
            for (size_t tj = 0; tj < NT; ++tj) {
                const double *a = sums.row_tile(one, tj);
                for (size_t jj = 0; jj < TS; ++jj) new_row[tj * TS + jj] = a[jj] / (size[one] * size[tj * TS + jj]);
            }
            for (size_t ti = 0; ti < NT; ++ti) {
                const double *a = sums.col_tile(ti, one);
                for (size_t ii = 0; ii < TS; ++ii) new_col[ti * TS + ii] = a[ii * TS] / (size[ti * TS + ii] * size[one]);
            }
        } else {
            for (size_t tj = 0; tj < NT; ++tj) {
                const double *a = link.row_tile(one, tj), *b = link.row_tile(two, tj);
                for (size_t jj = 0; jj < TS; ++jj)
                    new_row[tj * TS + jj] = linkage == 0 ? std::fmin(a[jj], b[jj]) : std::fmax(a[jj], b[jj]);
            }
            for (size_t ti = 0; ti < NT; ++ti) {
                const double *a = link.col_tile(ti, one), *b = link.col_tile(ti, two);
                for (size_t ii = 0; ii < TS; ++ii)
                    new_col[ti * TS + ii] = linkage == 0 ? std::fmin(a[ii * TS], b[ii * TS]) : std::fmax(a[ii * TS], b[ii * TS]);
            }
        }
        alive[two] = 0;
        for (size_t j = 0; j < NP; ++j)
            if (!alive[j]) { new_row[j] = inf; new_col[j] = inf; }
        new_row[one] = new_col[one] = inf;
        for (size_t tj = 0; tj < NT; ++tj) {
            double *a = link.row_tile(one, tj), *b = link.row_tile(two, tj);
            for (size_t jj = 0; jj < TS; ++jj) { a[jj] = new_row[tj * TS + jj]; b[jj] = inf; }
        }
        for (size_t ti = 0; ti < NT; ++ti) {
            double *a = link.col_tile(ti, one), *b = link.col_tile(ti, two);
            for (size_t ii = 0; ii < TS; ++ii) { a[ii * TS] = new_col[ti * TS + ii]; b[ii * TS] = inf; }
        }
        row_val[two] = inf;
        // rows whose minimum sat in a column that changed or died: rescan; any other row can only have gained a new
        // minimum in column `one`
        for (size_t i = 0; i < N; ++i) {
            if (!alive[i]) continue;
            if (i == one || row_col[i] == (int64_t)one || row_col[i] == (int64_t)two) { rescan(i); continue; }
            const double v = new_col[i];
            if (v < row_val[i] || (v == row_val[i] && (int64_t)one < row_col[i])) { row_val[i] = v; row_col[i] = (int64_t)one; }
        }
    }
    return 0;
}
