// decode rate of GsBgzfReader alone (no file pipeline) for several thread counts (developer tool):
//   g++ -O2 -std=c++17 -pthread -o /tmp/bgzf_bench tools/bgzf_decode_bench.cpp -lz && /tmp/bgzf_bench
#include "../genestrip_amd/csrc/gs_inflate.h"
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
static std::vector<uint8_t> fastq(size_t n) {
    std::mt19937_64 rng(1);
    std::vector<uint8_t> v;
    const char *b = "ACGT";
    unsigned long long id = 0;
    while (v.size() < n) {
        char d[64];
        int m = snprintf(d, 64, "@r%llu\n", id++);
        v.insert(v.end(), d, d + m);
        for (int i = 0; i < 150; i++) v.push_back(b[rng() & 3]);
        v.push_back('\n'); v.push_back('+'); v.push_back('\n');
        v.insert(v.end(), 150, 'I');
        v.push_back('\n');
    }
    v.resize(n);
    return v;
}
static std::vector<uint8_t> bgzf(const std::vector<uint8_t> &in) {
    std::vector<uint8_t> out;
    const size_t B = 65280;
    for (size_t at = 0;; at += B) {
        const size_t n = at < in.size() ? std::min(B, in.size() - at) : 0;
        z_stream z{};
        deflateInit2(&z, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        std::vector<uint8_t> body(deflateBound(&z, n) + 16);
        z.next_in = (Bytef *)in.data() + (n ? at : 0); z.avail_in = (uInt)n; z.next_out = body.data(); z.avail_out = (uInt)body.size();
        deflate(&z, Z_FINISH);
        body.resize(z.total_out);
        deflateEnd(&z);
        const uint32_t bsize = (uint32_t)(18 + body.size() + 8 - 1), crc = (uint32_t)crc32(crc32(0, Z_NULL, 0), in.data() + (n ? at : 0), (uInt)n), isz = (uint32_t)n;
        const uint8_t h[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
        out.insert(out.end(), h, h + 18);
        out.insert(out.end(), body.begin(), body.end());
        for (int i = 0; i < 4; i++) out.push_back((uint8_t)(crc >> (8 * i)));
        for (int i = 0; i < 4; i++) out.push_back((uint8_t)(isz >> (8 * i)));
        if (n == 0) break;
    }
    return out;
}
int main() {
    const auto text = fastq((size_t)400 << 20);
    const auto packed = bgzf(text);
    std::vector<uint8_t> out((size_t)8 << 20);
    for (int threads : {1, 4, 8, 16, 24, 32}) {
        double best = 1e9;
        for (int rep = 0; rep < 2; rep++) {
            GsBgzfReader br(packed.data(), packed.size(), threads);
            const auto t0 = std::chrono::steady_clock::now();
            bool done = false;
            size_t total = 0;
            while (!done) {
                size_t p = 0;
                if (!br.read(out.data(), out.size(), &p, &done)) return 1;
                total += p;
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (total != text.size()) return 2;
            best = std::min(best, dt);
        }
        printf("threads %2d: %.2f GB/s of text\n", threads, text.size() / best / 1e9);
    }
    return 0;
}
