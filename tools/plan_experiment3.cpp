// dev helper (offline, no GPU): row-segment fetch counts when ONE workgroup walks a run of R consecutive tiles per column chunk
// and keeps the rows of the previous tiles in LDS (optimal replacement: the plan is static), for tile sizes / LDS row
// capacities / run lengths (HISTORY 5.1).
//   g++ -O2 -std=c++17 -o /tmp/pe3 tools/plan_experiment3.cpp && /tmp/pe3 gpurun_out/c3_centers.f64 gpurun_out/c3_idx.i32 26
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <unordered_map>
#include <unordered_set>
#include <vector>
static uint64_t hilbert3(uint32_t x0, uint32_t x1, uint32_t x2, int b) {
    uint32_t X[3] = {x0, x1, x2};
    const uint32_t M = 1u << (b - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) { const uint32_t P = Q - 1;
        for (int i = 0; i < 3; ++i) { if (X[i] & Q) X[0] ^= P; else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; } } }
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    uint64_t h = 0;
    for (int bit = b - 1; bit >= 0; --bit) for (int i = 0; i < 3; ++i) h = (h << 1) | ((X[i] >> bit) & 1u);
    return h;
}
template <typename T> static std::vector<T> slurp(const char *f) {
    FILE *fp = fopen(f, "rb"); if (!fp) { perror(f); exit(1); }
    fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T)); if (fread(v.data(), sizeof(T), v.size(), fp) != v.size()) exit(1); fclose(fp); return v;
}
int main(int argc, char **argv) {
    auto centers = slurp<double>(argv[1]); auto idx = slurp<int32_t>(argv[2]); const int k = atoi(argv[3]);
    const int64_t nc = idx.size() / k;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < nc; ++i) for (int j = 0; j < 3; ++j) { lo[j] = std::min(lo[j], centers[i * 3 + j]); hi[j] = std::max(hi[j], centers[i * 3 + j]); }
    double ext = 0; for (int j = 0; j < 3; ++j) ext = std::max(ext, hi[j] - lo[j]);
    const double scale = 65535.0 / ext;
    std::vector<uint64_t> key(nc);
    for (int64_t i = 0; i < nc; ++i) key[i] = hilbert3((uint32_t)((centers[i * 3] - lo[0]) * scale), (uint32_t)((centers[i * 3 + 1] - lo[1]) * scale), (uint32_t)((centers[i * 3 + 2] - lo[2]) * scale), 16);
    std::vector<int32_t> perm(nc); std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
    for (int tc : {64, 32, 16}) for (int cap : {496, 640, 1100}) {
        // tiles: greedy, <= tc cells, <= min(cap, 496)-ish distinct rows (needed set must fit in cap)
        std::vector<std::vector<int32_t>> tiles; std::vector<int> tcells;
        { std::unordered_set<int32_t> s; int cells = 0;
          for (int64_t p = 0; p < nc; ++p) { const int32_t *ci = &idx[(size_t)perm[p] * k]; int add = 0; for (int m = 0; m < k; ++m) add += !s.count(ci[m]);
            if (cells && (cells == tc || (int)s.size() + add > std::min(cap, 496))) { tiles.emplace_back(s.begin(), s.end()); tcells.push_back(cells); s.clear(); cells = 0; }
            for (int m = 0; m < k; ++m) s.insert(ci[m]); ++cells; }
          if (cells) { tiles.emplace_back(s.begin(), s.end()); tcells.push_back(cells); } }
        int64_t base = 0; for (auto &t : tiles) base += t.size();
        for (int R : {1, 2, 4, 8, 16, 64}) {
            int64_t fetch = 0;
            for (size_t r0 = 0; r0 < tiles.size(); r0 += R) {
                const size_t r1 = std::min(tiles.size(), r0 + R);
                // next-use lists
                std::unordered_map<int32_t, std::vector<int>> uses;
                for (size_t j = r0; j < r1; ++j) for (int32_t r : tiles[j]) uses[r].push_back((int)j);
                std::unordered_map<int32_t, int> lds;   // row -> index into uses[r] of next use
                for (size_t j = r0; j < r1; ++j) {
                    std::unordered_set<int32_t> need(tiles[j].begin(), tiles[j].end());
                    int miss = 0; for (int32_t r : tiles[j]) miss += !lds.count(r);
                    fetch += miss;
                    // evict until room: candidates not in need, farthest next use first
                    int over = (int)lds.size() + miss - cap;
                    if (over > 0) {
                        std::vector<std::pair<int, int32_t>> cand;
                        for (auto &e : lds) if (!need.count(e.first)) { auto &u = uses[e.first]; auto it = std::upper_bound(u.begin(), u.end(), (int)j); cand.push_back({it == u.end() ? 1 << 30 : *it, e.first}); }
                        std::sort(cand.begin(), cand.end(), [](auto &a, auto &b) { return a.first > b.first; });
                        for (int i = 0; i < over && i < (int)cand.size(); ++i) lds.erase(cand[i].second);
                    }
                    for (int32_t r : tiles[j]) lds[r] = 0;
                    // drop rows never used again in run
                    for (auto it = lds.begin(); it != lds.end();) { auto &u = uses[it->first]; if (u.back() <= (int)j) it = lds.erase(it); else ++it; }
                }
            }
            printf("tc %3d cap %4d tiles %6zu base %8lld  run %3d: fetched %8lld (%.3f of base, %.3f of 4181599)\n", tc, cap, tiles.size(), (long long)base, R, (long long)fetch, (double)fetch / base, fetch / 4181599.0);
        }
    }
}
