// Decision tables of the S^3 topology shared by the host engine (topology.cpp) and the device engine (topo_dev.hip).
//
// Reference behaviour: _assign_indices, s_cube.py:1188-1536 (which node of a new child is taken from which same-level
// leaf neighbour / earlier sibling, in which order), restated as tables; s3t_selfcheck (topology.cpp) verifies them
// geometrically.  Directions / slots: s_cube.py:188-194, 22-26.
#pragma once

#include <cstdint>

namespace s3topo {

constexpr int LEAF = -1;      // Cell.children is None
constexpr int INVALID = -2;   // Cell.children == []

// node-sharing rules of _assign_indices.  For child i the entries are processed in order; an entry either looks the
// node up in same-level leaf neighbours (first hit wins, else a new node is appended) or copies it from an earlier
// sibling.  {node, n_cand, {slot, nb_node}...} / {node, -1, {sibling, sibling_node}}
struct NodeRule { int8_t node; int8_t n_cand; int8_t cand[3][2]; };
constexpr int N_RULES_2D = 3, N_RULES_3D = 7;     // rules per child (every node but the one inherited from the parent)

struct NbEntry { int8_t pslot; int8_t target; };   // pslot < 0: sibling `target`; else parent's neighbour slot + its child

// slots: w0 nw1 n2 ne3 e4 se5 s6 sw7 | wl8 nwl9 nl10 nel11 el12 sel13 sl14 swl15 cl16 | wu17 nwu18 nu19 neu20 eu21 seu22
// su23 swu24 cu25 ; nodes: swu0 nwu1 neu2 seu3 swl4 nwl5 nel6 sel7
#define S3_NODE_RULES_2D_INIT \
{ \
    /* child 0 */ {{1, 1, {{0, 2}}}, {2, 0, {}}, {3, 1, {{6, 2}}}}, \
    /* child 1 */ {{2, 1, {{2, 3}}}, {0, -1, {{0, 1}}}, {3, -1, {{0, 2}}}}, \
    /* child 2 */ {{3, 1, {{4, 0}}}, {0, -1, {{0, 2}}}, {1, -1, {{1, 2}}}}, \
    /* child 3 */ {{0, -1, {{0, 3}}}, {1, -1, {{0, 2}}}, {2, -1, {{2, 3}}}}, \
}

#define S3_NODE_RULES_3D_INIT \
{ \
    /* child 0 */ {{1, 3, {{0, 2}, {17, 6}, {25, 5}}}, {2, 1, {{25, 6}}}, {3, 3, {{6, 2}, {23, 6}, {25, 7}}}, \
                   {4, 3, {{0, 7}, {7, 6}, {6, 5}}}, {5, 1, {{0, 6}}}, {6, 0, {}}, {7, 1, {{6, 6}}}}, \
    /* child 1 */ {{2, 3, {{2, 3}, {19, 7}, {25, 6}}}, {5, 3, {{0, 6}, {1, 7}, {2, 4}}}, {6, 1, {{2, 7}}}, \
                   {0, -1, {{0, 1}}}, {3, -1, {{0, 2}}}, {4, -1, {{0, 5}}}, {7, -1, {{0, 6}}}}, \
    /* child 2 */ {{3, 3, {{4, 0}, {21, 4}, {25, 7}}}, {6, 3, {{4, 5}, {3, 4}, {2, 7}}}, {7, 1, {{4, 4}}}, \
                   {0, -1, {{0, 2}}}, {1, -1, {{1, 2}}}, {4, -1, {{0, 6}}}, {5, -1, {{1, 6}}}}, \
    /* child 3 */ {{7, 3, {{4, 4}, {5, 5}, {6, 6}}}, {0, -1, {{0, 3}}}, {1, -1, {{0, 2}}}, {2, -1, {{2, 3}}}, \
                   {4, -1, {{0, 7}}}, {5, -1, {{0, 6}}}, {6, -1, {{2, 7}}}}, \
    /* child 4 */ {{5, 3, {{0, 6}, {8, 2}, {16, 1}}}, {6, 1, {{16, 2}}}, {7, 3, {{6, 6}, {14, 2}, {16, 3}}}, \
                   {0, -1, {{0, 4}}}, {1, -1, {{0, 5}}}, {2, -1, {{0, 6}}}, {3, -1, {{0, 7}}}}, \
    /* child 5 */ {{6, 3, {{2, 7}, {10, 3}, {16, 2}}}, {0, -1, {{1, 4}}}, {1, -1, {{1, 5}}}, {2, -1, {{1, 6}}}, \
                   {3, -1, {{1, 7}}}, {4, -1, {{4, 5}}}, {7, -1, {{4, 6}}}}, \
    /* child 6 */ {{7, 3, {{4, 4}, {12, 0}, {16, 3}}}, {0, -1, {{2, 4}}}, {1, -1, {{2, 5}}}, {2, -1, {{2, 6}}}, \
                   {3, -1, {{2, 7}}}, {4, -1, {{5, 7}}}, {5, -1, {{5, 6}}}}, \
    /* child 7 */ {{0, -1, {{3, 4}}}, {1, -1, {{3, 5}}}, {2, -1, {{3, 6}}}, {3, -1, {{3, 7}}}, {4, -1, {{4, 7}}}, \
                   {5, -1, {{4, 6}}}, {6, -1, {{6, 7}}}}, \
}

// transient encodings of a node id while a batch is assembled in parallel (final ids are >= 0)
constexpr int64_t REF_BASE = (int64_t)1 << 20;

}  // namespace s3topo
