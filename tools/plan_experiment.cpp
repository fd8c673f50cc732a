// dev helper (offline, no GPU): staged-row count of tile-packing strategies for the planned interpolation kernel.
// input: raw files written from tools/dump_plan_inputs.py output (centers f64 [nc][3], idx i32 [nc][k]).
//   g++ -O2 -std=c++17 -o gpurun_out/plan_experiment tools/plan_experiment.cpp && gpurun_out/plan_experiment <centers> <idx> <k>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <unordered_map>
#include <vector>

static uint64_t spread3(uint64_t v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

// Skilling's transpose <-> Hilbert index (3-D, b bits per axis)
static uint64_t hilbert3(uint32_t x0, uint32_t x1, uint32_t x2, int b) {
    uint32_t X[3] = {x0, x1, x2};
    const uint32_t M = 1u << (b - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    uint64_t h = 0;
    for (int bit = b - 1; bit >= 0; --bit)
        for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> bit) & 1u);
    return h;
}

struct Result { int64_t tiles, rows; };

static std::vector<int32_t> g_mult;     // tiles holding each source row (filled by pack)
static Result pack(const std::vector<int32_t> &perm, const std::vector<int32_t> &idx, int k, int tc, int ucap) {
    std::unordered_map<int32_t, int> seen;
    std::fill(g_mult.begin(), g_mult.end(), 0);
    int64_t tiles = 0, rows = 0;
    int cells = 0;
    for (size_t pos = 0; pos < perm.size(); ++pos) {
        const int32_t *ci = &idx[(size_t)perm[pos] * k];
        int fresh = 0;
        for (int m = 0; m < k; ++m) fresh += !seen.count(ci[m]);
        // (repeats inside a row are negligible for statistics)
        if (cells == tc || (int)seen.size() + fresh > ucap) {
            ++tiles; rows += seen.size(); for (auto &kv : seen) ++g_mult[kv.first]; seen.clear(); cells = 0;
        }
        for (int m = 0; m < k; ++m) seen.emplace(ci[m], 0);
        ++cells;
    }
    if (cells) { ++tiles; rows += seen.size(); for (auto &kv : seen) ++g_mult[kv.first]; }
    int64_t hist[6] = {0, 0, 0, 0, 0, 0};
    for (int32_t v : g_mult) if (v) hist[std::min(v, 5)] += v;
    printf("    staged rows by tile multiplicity 1/2/3/4/5+: %lld %lld %lld %lld %lld\n", (long long)hist[1], (long long)hist[2], (long long)hist[3], (long long)hist[4], (long long)hist[5]);
    return {tiles, rows};
}

// window greedy: the next tile member is the cell, among the first W unassigned cells of the curve order, that adds
// the fewest new rows (ties: curve order)
static Result pack_window(const std::vector<int32_t> &perm, const std::vector<int32_t> &idx, int k, int tc, int ucap, int W) {
    const int64_t nc = perm.size();
    std::vector<int32_t> win;          // unassigned cells, in curve order
    int64_t next = 0;
    std::unordered_map<int32_t, int> seen;
    int64_t tiles = 0, rows = 0;
    int cells = 0;
    while (true) {
        while ((int)win.size() < W && next < nc) win.push_back(perm[next++]);
        if (win.empty()) break;
        int best = -1, best_fresh = 1 << 30;
        if (cells == 0) { best = 0; best_fresh = k; }
        else
            for (int i = 0; i < (int)win.size(); ++i) {
                const int32_t *ci = &idx[(size_t)win[i] * k];
                int fresh = 0;
                for (int m = 0; m < k; ++m) fresh += !seen.count(ci[m]);
                if (fresh < best_fresh) { best_fresh = fresh; best = i; }
            }
        if (cells == tc || (int)seen.size() + best_fresh > ucap) {
            ++tiles; rows += seen.size(); seen.clear(); cells = 0;
            continue;
        }
        const int32_t *ci = &idx[(size_t)win[best] * k];
        for (int m = 0; m < k; ++m) seen.emplace(ci[m], 0);
        win.erase(win.begin() + best);
        ++cells;
    }
    if (cells) { ++tiles; rows += seen.size(); }
    return {tiles, rows};
}

// block-aligned packing: cells in octree-aligned Hilbert order; a "block" = the cells sharing the ancestor two levels up
// (64 cells when fully refined).  Tiles are closed at block boundaries whenever the next whole block does not fit.
static Result pack_blocks(const std::vector<int32_t> &perm, const std::vector<uint64_t> &block, const std::vector<int32_t> &idx,
                          int k, int tc, int ucap) {
    std::unordered_map<int32_t, int> seen;
    int64_t tiles = 0, rows = 0;
    int cells = 0;
    const size_t n = perm.size();
    size_t pos = 0;
    while (pos < n) {
        size_t end = pos;                                   // extent of the block starting at pos
        while (end < n && block[perm[end]] == block[perm[pos]]) ++end;
        const int bsize = (int)(end - pos);
        if (cells > 0 && cells + bsize > tc) { ++tiles; rows += seen.size(); seen.clear(); cells = 0; }
        for (size_t p = pos; p < end; ++p) {
            const int32_t *ci = &idx[(size_t)perm[p] * k];
            int fresh = 0;
            for (int m = 0; m < k; ++m) fresh += !seen.count(ci[m]);
            if (cells == tc || (int)seen.size() + fresh > ucap) { ++tiles; rows += seen.size(); seen.clear(); cells = 0; }
            for (int m = 0; m < k; ++m) seen.emplace(ci[m], 0);
            ++cells;
        }
        pos = end;
    }
    if (cells) { ++tiles; rows += seen.size(); }
    return {tiles, rows};
}

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    const int k = atoi(argv[3]);
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); const long nb = ftell(f); fseek(f, 0, SEEK_SET);
    const int64_t nc = nb / 24;
    std::vector<double> ctr(nc * 3); if (fread(ctr.data(), 8, nc * 3, f) != (size_t)nc * 3) return 2; fclose(f);
    std::vector<int32_t> idx(nc * k);
    f = fopen(argv[2], "rb"); if (fread(idx.data(), 4, nc * k, f) != (size_t)nc * k) return 2; fclose(f);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t c = 0; c < nc; ++c) for (int j = 0; j < 3; ++j) { lo[j] = std::min(lo[j], ctr[c*3+j]); hi[j] = std::max(hi[j], ctr[c*3+j]); }
    double ext = 0; for (int j = 0; j < 3; ++j) ext = std::max(ext, hi[j] - lo[j]);
    if (argc > 4) {                                          // octree-aligned experiment: levels file + root origin/width
        std::vector<int8_t> lev(nc);
        f = fopen(argv[4], "rb"); if (fread(lev.data(), 1, nc, f) != (size_t)nc) return 2; fclose(f);
        const double org[3] = {atof(argv[5]), atof(argv[6]), atof(argv[7])}, width = atof(argv[8]);
        const int LMAX = 12;
        std::vector<uint64_t> key(nc), block(nc);
        std::vector<int32_t> perm(nc);
        std::iota(perm.begin(), perm.end(), 0);
        for (int64_t c = 0; c < nc; ++c) {
            uint32_t q[3];
            for (int j = 0; j < 3; ++j) q[j] = (uint32_t)((ctr[c*3+j] - org[j]) / width * (1 << LMAX));
            key[c] = hilbert3(q[0], q[1], q[2], LMAX);
            block[c] = key[c] >> (3 * (LMAX - lev[c] + 2));
            block[c] = block[c] * 16 + (uint64_t)lev[c];   // blocks of different levels are different blocks
        }
        std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b_) { return key[a] < key[b_]; });
        g_mult.assign(*std::max_element(idx.begin(), idx.end()) + 1, 0);
        Result r0 = pack(perm, idx, k, 64, 496);
        printf("octree-aligned hilbert, greedy : tiles %7lld staged rows %9lld\n", (long long)r0.tiles, (long long)r0.rows);
        Result r1 = pack_blocks(perm, block, idx, k, 64, 496);
        printf("octree-aligned hilbert, blocks : tiles %7lld staged rows %9lld\n", (long long)r1.tiles, (long long)r1.rows);
        Result r2 = pack_blocks(perm, block, idx, k, 64, 560);
        printf("  (ucap 560)                   : tiles %7lld staged rows %9lld\n", (long long)r2.tiles, (long long)r2.rows);
        return 0;
    }
    std::vector<int32_t> perm(nc);
    g_mult.assign(*std::max_element(idx.begin(), idx.end()) + 1, 0);
    for (const char *order : {"creation", "morton", "hilbert"}) {
        std::iota(perm.begin(), perm.end(), 0);
        std::vector<uint64_t> key(nc, 0);
        const int b = 16;
        const double scale = ((1 << b) - 1) / ext;
        for (int64_t c = 0; c < nc; ++c) {
            uint32_t q[3]; for (int j = 0; j < 3; ++j) q[j] = (uint32_t)((ctr[c*3+j] - lo[j]) * scale);
            if (order[0] == 'm') key[c] = spread3(q[0]) | spread3(q[1]) << 1 | spread3(q[2]) << 2;
            if (order[0] == 'h') key[c] = hilbert3(q[0], q[1], q[2], b);
        }
        if (order[0] != 'c') std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b_) { return key[a] < key[b_]; });
        if (order[0] == 'h')
            for (int W : std::vector<int>{}) {
                Result r = pack_window(perm, idx, k, 64, 496, W);
                printf("%-9s window %4d tc=64: tiles %7lld staged rows %9lld (%.3f per cell)\n", order, W, (long long)r.tiles, (long long)r.rows, (double)r.rows / nc);
            }
        if (order[0] == 'h')
            for (int ucap : {400, 448, 560, 600, 700}) {
                Result r = pack(perm, idx, k, 64, ucap);
                printf("%-9s tc= 64 ucap=%4d: tiles %7lld staged rows %9lld\n", order, ucap, (long long)r.tiles, (long long)r.rows);
            }
        for (int tc : {64, 128}) {
            const int ucap = tc == 64 ? 496 : 1024;
            Result r = pack(perm, idx, k, tc, ucap);
            printf("%-9s tc=%3d ucap=%4d: tiles %7lld staged rows %9lld (%.3f per cell)\n", order, tc, ucap, (long long)r.tiles, (long long)r.rows, (double)r.rows / nc);
        }
    }
    return 0;
}
