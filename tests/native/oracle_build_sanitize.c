/* the DB-construction restatement (oracle/gs_oracle.c: orc_build_*) under AddressSanitizer / UBSan: a few regions with shared
 * material, invalid bytes and lower case; fill, optimize, update, fetch.  Prints the number of stored k-mers. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/gs_oracle.h"

int main(void) {
    const int32_t parent[7] = {-1, 0, 1, 1, 2, 4, 0};
    enum { R = 40, LEN = 700 };
    uint8_t *seq = (uint8_t *)malloc((size_t)R * LEN);
    uint64_t off[R + 1];
    int32_t node[R];
    uint64_t x = 88172645463325252ULL;
    const char alphabet[] = "ACGTacgtN\r";
    for (int r = 0; r < R; r++) {
        off[r] = (uint64_t)r * LEN;
        node[r] = r % 7;
        for (int i = 0; i < LEN; i++) {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            seq[r * LEN + i] = (uint8_t)alphabet[(x >> 20) % ((x >> 40) % 16 ? 4 : 10)];
        }
        if (r % 3 == 0 && r) memcpy(seq + r * LEN + 100, seq + 100, 300); /* shared material */
    }
    off[R] = (uint64_t)R * LEN;
    for (int k = 1; k <= 31; k += 6) {
        orc_build *b = orc_build_begin(k, 7, parent, 1, k == 7 ? 3 : 1);
        orc_build_fill(b, seq, off, node, R / 2);
        const int64_t n = orc_build_optimize(b);
        orc_build_update(b, seq, off, node, R);
        int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
        int32_t *vals = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
        orc_build_fetch(b, keys, vals);
        for (int64_t i = 1; i < n; i++)
            if (keys[i] <= keys[i - 1] || vals[i] < 0 || vals[i] > 6) return 2;
        printf("k=%d stored=%lld\n", k, (long long)n);
        free(keys);
        free(vals);
        orc_build_destroy(b);
    }
    free(seq);
    return 0;
}
