// Edge capsules of the obstacle mask's two-sided cull (host side, no HIP): per edge the chord between its first and last
// sample, the largest distance of a sample from that chord and the largest gap between consecutive sample projections.
// The decisions the kernel derives from them (paths_team.hpp, phase 2) must be CONSERVATIVE with respect to the reference's
// exact test (GraphBase.py:626-643: any sample with d^2 <= threshold^2); tests/test_capsule_cull.py checks that property on
// the real lattices with the kernel's fp32 arithmetic restated in NumPy.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace ltplcap {

// 8 floats per edge: (Ax, Ay, ABx, ABy), (1 / |AB|^2, dev, hg2, first sample | #samples << 24 as bit pattern; #samples = 0:
// the sample range does not fit the packing and is looked up in samp_ptr). Returns the slack the kernel adds to BOTH decisions.
inline float build(int n_edges, const int32_t* samp_ptr, const double* sx, const double* sy, int n_samples, float* out)
{
    double maxabs = 1.0;
    for (int k = 0; k < n_samples; ++k) maxabs = std::fmax(maxabs, std::fmax(std::fabs(sx[k]), std::fabs(sy[k])));
    // fp32 effects of the cull's distance: query position and chord are rounded to fp32 (relative 2^-24 of the coordinate
    // magnitude each), the arithmetic adds a few ulps of the same magnitude
    const float slack = (float)(maxabs * 16.0 * 5.960464477539063e-08 + 2.0e-4);
    std::vector<double> along;
    for (int e = 0; e < n_edges; ++e) {
        const int k0 = samp_ptr[e], k1 = samp_ptr[e + 1], ns = k1 - k0;
        float* r = out + (size_t)e * 8;
        for (int i = 0; i < 8; ++i) r[i] = 0.0f;
        if (ns <= 0) continue;
        const double ax = sx[k0], ay = sy[k0], bx = sx[k1 - 1], by = sy[k1 - 1];
        // the kernel works with the fp32-rounded chord: deviation and gaps are measured against THAT chord
        const float axf = (float)ax, ayf = (float)ay, abxf = (float)(bx - ax), abyf = (float)(by - ay);
        const double cx = axf, cy = ayf, vx = abxf, vy = abyf, len2 = vx * vx + vy * vy, len = std::sqrt(len2);
        double dev = 0.0;
        along.clear();
        for (int k = k0; k < k1; ++k) {
            const double ux = sx[k] - cx, uy = sy[k] - cy;
            double t = len2 > 0.0 ? (ux * vx + uy * vy) / len2 : 0.0;
            t = std::fmin(std::fmax(t, 0.0), 1.0);
            const double dx = ux - t * vx, dy = uy - t * vy;
            dev = std::fmax(dev, std::sqrt(dx * dx + dy * dy));
            along.push_back(t * len);
        }
        std::sort(along.begin(), along.end());
        // the rounded chord ends are not samples: a_end = distance from a chord end to the nearest projection (~1e-5 m); it joins
        // dev (foot of the query clamped at a chord end: nearest sample within d + a_end + dev) and the gaps
        const double a_end = std::fmax(along.front(), len - along.back());
        double g = 2.0 * a_end;
        for (size_t i = 1; i < along.size(); ++i) g = std::fmax(g, along[i] - along[i - 1]);
        const unsigned packed = (k0 < (1 << 24) && ns <= 255) ? ((unsigned)k0 | ((unsigned)ns << 24)) : 0u;
        r[0] = axf; r[1] = ayf; r[2] = abxf; r[3] = abyf;
        r[4] = len2 > 0.0 ? (float)(1.0 / len2) : 0.0f;
        r[5] = std::nextafter((float)((dev + a_end) * (1.0 + 1.0e-6)), INFINITY);
        r[6] = std::nextafter((float)(0.25 * g * g * (1.0 + 1.0e-6)), INFINITY);
        std::memcpy(&r[7], &packed, sizeof(float));
    }
    return slack;
}

}   // namespace ltplcap
