// dev helper (offline, no GPU): row-segment fetch counts of tile-packing strategies with a joint LDS budget
// (rows * seg_bytes + cells * k * 10 <= budget) and with rows used by a single cell of the tile bypassing LDS.
//   g++ -O2 -std=c++17 -o /tmp/pe2 tools/plan_experiment2.cpp && /tmp/pe2 gpurun_out/c3_centers.f64 gpurun_out/c3_idx.i32 26
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <unordered_map>
#include <vector>

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

template <typename T> static std::vector<T> slurp(const char *f) {
    FILE *fp = fopen(f, "rb");
    if (!fp) { perror(f); exit(1); }
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T));
    if (fread(v.data(), sizeof(T), v.size(), fp) != v.size()) exit(1);
    fclose(fp);
    return v;
}

struct Res { int64_t tiles = 0, lds_rows = 0, bypass_rows = 0, cells_hist[9] = {0}; };

// greedy along the curve.  A tile is closed when adding the next cell would break: cells <= tc_max and
// lds_rows * seg + cells * k * 10 <= budget, where lds_rows = rows with >= min_share cells of the tile (others bypass).
static Res pack(const std::vector<int32_t> &perm, const std::vector<int32_t> &idx, int k, int tc_max, int budget, int seg,
                int min_share, int row_cap) {
    Res r;
    std::unordered_map<int32_t, int> cnt;
    int cells = 0, shared = 0;          // shared = rows with count >= min_share
    auto close = [&]() {
        ++r.tiles;
        r.lds_rows += shared;
        r.bypass_rows += (int64_t)cnt.size() - shared;
        r.cells_hist[std::min(8, cells / 32)]++;
        cnt.clear(); cells = 0; shared = 0;
    };
    for (size_t pos = 0; pos < perm.size(); ++pos) {
        const int32_t *ci = &idx[(size_t)perm[pos] * k];
        int add = 0;
        for (int m = 0; m < k; ++m) {
            auto it = cnt.find(ci[m]);
            const int c = it == cnt.end() ? 0 : it->second;
            if (c + 1 == min_share) ++add;
        }
        if (cells > 0 && (cells == tc_max || (int64_t)(shared + add) * seg + (int64_t)(cells + 1) * k * 10 > budget || shared + add > row_cap))
            { close(); add = min_share == 1 ? k : 0; }
        for (int m = 0; m < k; ++m) {
            int &c = cnt[ci[m]];
            ++c;
            if (c == min_share) ++shared;
        }
        ++cells;
    }
    if (cells) close();
    return r;
}

int main(int argc, char **argv) {
    if (argc < 4) return 1;
    auto centers = slurp<double>(argv[1]);
    auto idx = slurp<int32_t>(argv[2]);
    const int k = atoi(argv[3]);
    const int64_t nc = idx.size() / k;
    printf("cells %lld k %d\n", (long long)nc, k);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < nc; ++i) for (int j = 0; j < 3; ++j) { lo[j] = std::min(lo[j], centers[i * 3 + j]); hi[j] = std::max(hi[j], centers[i * 3 + j]); }
    double ext = 0; for (int j = 0; j < 3; ++j) ext = std::max(ext, hi[j] - lo[j]);
    const double scale = 65535.0 / ext;
    std::vector<uint64_t> key(nc);
    for (int64_t i = 0; i < nc; ++i)
        key[i] = hilbert3((uint32_t)((centers[i * 3] - lo[0]) * scale), (uint32_t)((centers[i * 3 + 1] - lo[1]) * scale), (uint32_t)((centers[i * 3 + 2] - lo[2]) * scale), 16);
    std::vector<int32_t> perm(nc);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
    { std::vector<int32_t> u(idx); std::sort(u.begin(), u.end()); printf("unique rows %lld\n", (long long)(std::unique(u.begin(), u.end()) - u.begin())); }

    auto show = [&](const char *name, Res r) {
        printf("%-58s tiles %7lld lds_rows %8lld bypass %8lld total %8lld  cells/32 hist:", name, (long long)r.tiles, (long long)r.lds_rows,
               (long long)r.bypass_rows, (long long)(r.lds_rows + r.bypass_rows));
        for (int i = 0; i < 9; ++i) printf(" %lld", (long long)r.cells_hist[i]);
        printf("\n");
    };
    const int B80 = 80 * 1024, B160 = 160 * 1024;
    show("r01: tc64 rows<=496 seg128 (fixed w/loc area)", pack(perm, idx, k, 64, 496 * 128 + 64 * k * 10, 128, 1, 496));
    show("joint 80K tc64", pack(perm, idx, k, 64, B80, 128, 1, 1 << 30));
    show("joint 80K tc128", pack(perm, idx, k, 128, B80, 128, 1, 1 << 30));
    show("joint 80K tc256", pack(perm, idx, k, 256, B80, 128, 1, 1 << 30));
    show("joint 160K tc128", pack(perm, idx, k, 128, B160, 128, 1, 1 << 30));
    show("joint 160K tc256", pack(perm, idx, k, 256, B160, 128, 1, 1 << 30));
    show("joint 160K tc512", pack(perm, idx, k, 512, B160, 128, 1, 1 << 30));
    show("bypass-single joint 80K tc64", pack(perm, idx, k, 64, B80, 128, 2, 1 << 30));
    show("bypass-single joint 80K tc128", pack(perm, idx, k, 128, B80, 128, 2, 1 << 30));
    show("bypass-single joint 80K tc256", pack(perm, idx, k, 256, B80, 128, 2, 1 << 30));
    show("bypass-single joint 160K tc256", pack(perm, idx, k, 256, B160, 128, 2, 1 << 30));
    show("bypass-single joint 160K tc512", pack(perm, idx, k, 512, B160, 128, 2, 1 << 30));
    show("seg64 joint 80K tc128", pack(perm, idx, k, 128, B80, 64, 1, 1 << 30));
    show("seg64 joint 80K tc256", pack(perm, idx, k, 256, B80, 64, 1, 1 << 30));
    show("seg64 bypass joint 80K tc256", pack(perm, idx, k, 256, B80, 64, 2, 1 << 30));
    show("no w/loc in LDS: 80K tc256", pack(perm, idx, k, 256, B80 + 256 * k * 10, 128, 1, 640));
    show("no w/loc in LDS: 160K tc512", pack(perm, idx, k, 512, B160 + 512 * k * 10, 128, 1, 1280));
    show("unbounded tc1024", pack(perm, idx, k, 1024, 1 << 30, 128, 1, 1 << 30));
    show("unbounded tc4096", pack(perm, idx, k, 4096, 1 << 30, 128, 1, 1 << 30));
    return 0;
}
