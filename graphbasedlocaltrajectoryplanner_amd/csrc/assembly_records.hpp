// Path assembly records (host side, built once per lattice by ltpl_create; exported for their test by ltpl_assembly_records).
// What the assembly of a path gathers per node / per edge, as ONE record each, so that the chain of dependent global round trips behind the
// backtrack is node record -> edge record (main_online_path_gen.py:260-328: nodes -> edges -> spline sample coordinates and headings; the
// arrays it was read from one hop at a time: layer_off -> in_ptr -> edge_src -> samp_ptr -> samp_x / samp_y / samp_psi).
//   node record  (16 B): first in-edge (CSC id) of the node, then the source nodes (index inside their layer) of its first
//                        LTPL_NODE_REC_SRC in-edges as bytes, 0xff = none. The in-edges of a node are sorted by source.
//   edge record  (LTPL_EDGE_REC doubles): [0] first sample | #samples << 32 (bit pattern), [1] edge length, [2] [3] x, y of the first
//                        sample, [4] [5] x, y of the last, [6] [7] sin, cos of the first sample's heading, [8] [9] of the last's.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define LTPL_EDGE_REC 10
#define LTPL_NODE_REC_SRC 12

namespace ltplrec {

inline void build(int V, int E, const int* in_ptr, const int* edge_src, const double* edge_len, const int* samp_ptr, const double* samp_x,
                  const double* samp_y, const double* samp_psi, std::vector<int32_t>& node_rec, std::vector<double>& edge_rec)
{
    node_rec.assign((size_t)V * 4, 0);
    for (int v = 0; v < V; ++v) {
        const int e0 = in_ptr[v], e1 = in_ptr[v + 1];
        unsigned char b[LTPL_NODE_REC_SRC];
        for (int k = 0; k < LTPL_NODE_REC_SRC; ++k) b[k] = e0 + k < e1 ? (unsigned char)edge_src[e0 + k] : (unsigned char)0xff;
        node_rec[(size_t)v * 4] = e0;
        memcpy(&node_rec[(size_t)v * 4 + 1], b, LTPL_NODE_REC_SRC);
    }
    edge_rec.assign((size_t)E * LTPL_EDGE_REC, 0.0);
    for (int e = 0; e < E; ++e) {
        const int k0 = samp_ptr[e], k1 = samp_ptr[e + 1];
        double* r = &edge_rec[(size_t)e * LTPL_EDGE_REC];
        const unsigned long long w = (unsigned long long)(unsigned)k0 | ((unsigned long long)(unsigned)(k1 - k0) << 32);
        memcpy(&r[0], &w, 8);
        r[1] = edge_len[e];
        if (k1 > k0) {
            r[2] = samp_x[k0]; r[3] = samp_y[k0]; r[4] = samp_x[k1 - 1]; r[5] = samp_y[k1 - 1];
            r[6] = sin(samp_psi[k0]); r[7] = cos(samp_psi[k0]); r[8] = sin(samp_psi[k1 - 1]); r[9] = cos(samp_psi[k1 - 1]);
        }
    }
}

}  // namespace ltplrec
