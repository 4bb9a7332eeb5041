// Sanitizer harness for the threaded writers of the host layer (tests/test_host_cpu.py builds it with -fsanitize=thread
// and with -fsanitize=address,undefined): a FormatPool fills per-thread buffers for many small and large "chunks", the
// OutFile threads write them to a plain and a gzip file in order, and the files are read back.
#include "../../genestrip_amd/csrc/gs_ingest.h"

#include <cstdio>
#include <string>
#include <vector>

using namespace gs_host;

static std::vector<uint8_t> slurp(const std::string &p, bool gz) {
    std::vector<uint8_t> out;
    char buf[1 << 16];
    if (gz) {
        gzFile f = gzopen(p.c_str(), "rb");
        int n;
        while ((n = gzread(f, buf, sizeof buf)) > 0) out.insert(out.end(), buf, buf + n);
        gzclose(f);
    } else {
        FILE *f = fopen(p.c_str(), "rb");
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
        fclose(f);
    }
    return out;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    int fails = 0;
    for (int threads : {1, 3, 8}) {
        const std::string plain = dir + "/w.txt", packed = dir + "/w.txt.gz";
        std::vector<uint8_t> want;
        {
            OutFile a, b, unused;
            if (!a.open(plain.c_str()) || !b.open(packed.c_str()) || !unused.open(nullptr)) return 99;
            FormatPool pool(threads);
            int64_t line_no = 0;
            for (int chunk = 0; chunk < 60; chunk++) {
                const int64_t n = chunk % 7 == 0 ? 40000 : 100 + 37 * chunk;  // big chunks use the pool, small ones run inline
                std::vector<std::vector<uint8_t>> parts((size_t)pool.threads());
                const int64_t base = line_no;
                pool.run(n, [&](int t, int64_t lo, int64_t hi) {
                    std::vector<uint8_t> &p = parts[(size_t)t];
                    p = a.take();
                    for (int64_t i = lo; i < hi; i++) {
                        char tmp[32];
                        const int m = snprintf(tmp, sizeof tmp, "line %lld\n", (long long)(base + i));
                        p.insert(p.end(), tmp, tmp + m);
                    }
                });
                line_no += n;
                for (auto &p : parts) {
                    want.insert(want.end(), p.begin(), p.end());
                    std::vector<uint8_t> copy = p;  // big parts become gzip members on this side, small ones in the writer
                    const bool packed = copy.size() >= 65536 && b.pack(copy);
                    b.write(std::move(copy), packed);
                    a.write(std::move(p));
                    unused.write(std::vector<uint8_t>(3, 'x'));  // an inactive file swallows its buffers
                }
            }
            if (!a.close() || !b.close()) fails++;
        }
        if (slurp(plain, false) != want) fails++;
        if (slurp(packed, true) != want) fails++;
    }
    {   // a gzip file nothing was written to is still a gzip file
        const std::string empty = dir + "/e.gz";
        {
            OutFile e;
            if (!e.open(empty.c_str()) || !e.close()) fails++;
        }
        gzFile f = gzopen(empty.c_str(), "rb");
        char c;
        if (!f || gzread(f, &c, 1) != 0 || !gzeof(f)) fails++;
        if (f) gzclose(f);
        FILE *raw = fopen(empty.c_str(), "rb");
        if (!raw || fgetc(raw) != 0x1f) fails++;
        if (raw) fclose(raw);
    }
    {   // a file that cannot take the data reports it at close()
        OutFile full;
        if (full.open("/dev/full")) {
            full.write(std::vector<uint8_t>(1 << 20, 'y'));
            if (full.close()) fails++;
        }
    }
    printf("fails %d\n", fails);
    return fails;
}
