// Closest-layer grid (host side, no HIP): a uniform grid over the track area that tells, for every cell, WHICH reference-line layers can
// be the closest one to ANY point of the cell -- at most two index intervals. The path kernel's phase 1 (closest reference-line layer per
// obstacle position, get_intersec_edges.py:40-51 = np.argmin of the squared distances over all layers) then evaluates the reference's
// exact fp64 distances on those few layers instead of on all of them (round 5: on the 400-layer C3 oval the full scan was 400 dependent
// iterations per position batch, two L2 loads each, and most of the kernel's wait cycles).
//
// Conservativeness. Cell with centre c and half-diagonal r (closed, plus a slack for the rounding of the cell look-up); for a layer l with
// reference point q_l and D_l = |c - q_l|: every point p of the cell has  D_l - r <= |p - q_l| <= D_l + r.  With U = min_l (D_l + r), the
// distance of p to ITS closest layer is <= U, and a layer that attains p's minimum -- or ties with it -- has D_l - r <= |p - q_l| <= U.
// So the candidate set { l : D_l - r <= U (1 + eps) + eps } contains every layer that can be np.argmin's answer for any p of the cell,
// all layers it could tie with included; the kernel scans the candidates in ascending layer order with the strict '<' of the full scan
// and finds the same first minimum. The set is stored as at most two ascending intervals (a closed track's closest layers wrap from L - 1
// to 0; a cell between two parts of the track sees both): gaps are closed, smallest first, until two intervals remain -- a superset is
// as good. Cells whose candidates cover more than a quarter of the line (far from the track: the middle of an oval) are marked
// "full scan" -- as is every position outside the grid: the kernel then runs the reference's scan over all layers for that scenario.
// tests/test_layer_grid.py checks the property on the real lattices (random and adversarial points against the brute-force argmin).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ltplgrid {

struct Grid {
    double x0 = 0.0, y0 = 0.0, inv_cell = 0.0, cell = 0.0;
    int nx = 0, ny = 0;
    std::vector<int32_t> cells;      // 4 ints per cell: first layer and length of interval 1 and of interval 2 (length 0: absent); length -1: full scan
};

// margin: how far beyond the reference line's bounding box the grid reaches (track width + obstacles next to the track)
inline Grid build(int L, const double* rx, const double* ry, double margin = 40.0, int max_cells = 1 << 16)
{
    Grid g;
    if (L < 1) return g;
    double xmin = rx[0], xmax = rx[0], ymin = ry[0], ymax = ry[0];
    for (int l = 1; l < L; ++l) { xmin = std::fmin(xmin, rx[l]); xmax = std::fmax(xmax, rx[l]); ymin = std::fmin(ymin, ry[l]); ymax = std::fmax(ymax, ry[l]); }
    xmin -= margin; ymin -= margin; xmax += margin; ymax += margin;
    const double w = xmax - xmin, h = ymax - ymin;
    // (the construction costs cells x layers distance evaluations on the host: bounded to ~4e8, i.e. well under a second at ltpl_create,
    //  by a coarser grid for lattices with very many layers)
    if ((double)max_cells * (double)L > 4.0e8) max_cells = (int)(4.0e8 / (double)L);
    if (max_cells < 1024) max_cells = 1024;
    double cell = std::sqrt(w * h / (double)max_cells);
    if (cell < 2.0) cell = 2.0;
    g.cell = cell; g.inv_cell = 1.0 / cell; g.x0 = xmin; g.y0 = ymin;
    g.nx = (int)std::ceil(w / cell) + 1; g.ny = (int)std::ceil(h / cell) + 1;
    g.cells.assign((size_t)g.nx * g.ny * 4, 0);
    // slack of the cell look-up: the kernel computes floor((p - x0) * inv_cell) in fp64 -- a point may land in the neighbouring cell by a
    // rounding of ~1e-12 relative of the coordinate offset; 1e-6 m covers it for any track
    const double r = cell * std::sqrt(0.5) * (1.0 + 1e-9) + 1e-6;
    std::vector<double> D((size_t)L);
    std::vector<char> cand((size_t)L);
    std::vector<int> lo, hi;
    for (int iy = 0; iy < g.ny; ++iy)
        for (int ix = 0; ix < g.nx; ++ix) {
            const double cx = xmin + (ix + 0.5) * cell, cy = ymin + (iy + 0.5) * cell;
            double U = INFINITY;
            for (int l = 0; l < L; ++l) { const double dx = rx[l] - cx, dy = ry[l] - cy; D[(size_t)l] = std::sqrt(dx * dx + dy * dy); U = std::fmin(U, D[(size_t)l] + r); }
            const double thr = U * (1.0 + 1e-9) + 1e-9;
            for (int l = 0; l < L; ++l) cand[(size_t)l] = (D[(size_t)l] - r <= thr) ? 1 : 0;
            int32_t* rec = &g.cells[((size_t)iy * g.nx + ix) * 4];
            lo.clear(); hi.clear();
            for (int l = 0; l < L; ++l)
                if (cand[(size_t)l] && (l == 0 || !cand[(size_t)l - 1])) { lo.push_back(l); int e = l; while (e + 1 < L && cand[(size_t)e + 1]) ++e; hi.push_back(e); }
            while (lo.size() > 2) {                    // close the smallest gap (the result stays a superset)
                size_t best = 0; int gap = lo[1] - hi[0];
                for (size_t k = 1; k + 1 < lo.size(); ++k) if (lo[k + 1] - hi[k] < gap) { gap = lo[k + 1] - hi[k]; best = k; }
                hi[best] = hi[best + 1]; lo.erase(lo.begin() + (long)best + 1); hi.erase(hi.begin() + (long)best + 1);
            }
            int total = 0;
            for (size_t k = 0; k < lo.size(); ++k) total += hi[k] - lo[k] + 1;
            if (lo.empty() || total > std::max(L / 4, 8)) { rec[0] = 0; rec[1] = -1; rec[2] = 0; rec[3] = 0; continue; }
            rec[0] = lo[0]; rec[1] = hi[0] - lo[0] + 1;
            if (lo.size() > 1) { rec[2] = lo[1]; rec[3] = hi[1] - lo[1] + 1; }
        }
    return g;
}

}  // namespace ltplgrid
