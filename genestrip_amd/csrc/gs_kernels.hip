// gs_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the match / filter hot path.
//
// match: ONE WAVE PER READ.  Lane p owns k-mer start position p; an "iteration" covers 128 positions as two
// sub-rounds of 64, so a 150 bp read at k=31 (120 k-mers) is exactly one iteration.  Per iteration:
//   1. coalesced byte loads of the bases, three 64-lane ballots per 64 bases (code-hi, code-lo, invalid)
//   2. per lane: funnel-shift the ballot planes -> forward k-mer, bit-reverse -> reverse complement,
//      representative orientation (gs_rep_planes), gs_mix_planes -> bucket + remainder
//   3. one 64-byte bucket line per k-mer (4 x dwordx4 per lane, both sub-rounds in flight together)
//   4. per-read reduce inside the wave: contig boundaries by neighbour compare, a wave-uniform loop over
//      the few contig "events" (flush stats / start contig / path merge), unique bitmap by test-then-
//      atomicOr, KrakenUniq-style vote + LCA with O(1) pre-order interval ancestor tests
//   5. per-taxid counters privatised in LDS per workgroup, flushed once with 64-bit global atomics
// The semantics follow C/match/FastqKMerMatcher.java:327-535; the structure does
// not: the reference's rolling/jumping state machine is replaced by its closed form
//   node(p) = INVALID if the window [p,p+k) holds a non-CGAT byte, else store lookup,
//   #INVALID iterations = popcount(bad bases at positions <= max-2) + (any bad base at >= max-1).
// Reads with more than 128 k-mer positions are queued and drained by gs_match_long_kernel, the same code
// iterating with carried state and per-wave scratch counters in HBM instead of in-register lists.
//
// filter: one wave per read, lanes = k-mer positions, the reference's exact Bloom hashes; accept is the
// closed form  #(member k-mers) >= max(posThreshold, 1)  of FastqBloomFilter.isAcceptRead (:120-161).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "gs_layout.h"
#include "gs_params.h"

// developer ablations (tools/ablate.sh): 1 = no per-read reduce, 2 = no record / table probe, 4 = no gate either,
// 8 = no statistics atomics
#ifndef GS_ABLATE
#define GS_ABLATE 0
#endif

#define GS_NODE_MISS (-1)
#define GS_NODE_INVALID (-2)
#define GS_NODE_NONE (-3)
#define GS_LONG_CHUNK 64           // entries of the long-read queue a wave reserves at a time (one atomic per chunk)
#define GS_LONG_NONE 0xffffffffu   // padding of a chunk (a batch holds at most 2^32 - 1 reads)

typedef unsigned long long u64;

__device__ __forceinline__ int gs_lane() { return (int)__lane_id(); }
__device__ __forceinline__ int gs_readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int gs_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// developer instrumentation (tools/phase_times.sh: -DGS_PHASE=1): wave cycles per phase of the short-read path, summed over all
// waves (s_memtime at the phase boundaries; `dep` is a value the phase produces, so the stamp sits behind the wait for it)
#ifndef GS_PHASE
#define GS_PHASE 0
#endif
#if GS_PHASE
__device__ unsigned long long gs_phase_acc[16];
__device__ __forceinline__ unsigned long long *gs_phase_row() {
    __shared__ unsigned long long ph[4][16];
    return ph[threadIdx.x >> 6];
}
__device__ __forceinline__ void gs_stamp(int i) {
    const unsigned long long t = __builtin_readcyclecounter();
    if (gs_lane() == 0) {
        unsigned long long *p = gs_phase_row();
        const unsigned long long last = p[15];
        p[15] = t;
        if (i >= 0) p[i] += t - last;
    }
}
#define GS_STAMP(i, dep)                \
    {                                   \
        asm volatile("" ::"v"(dep));    \
        gs_stamp(i);                    \
    }
extern "C" int gs_debug_phase(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(gs_phase_acc), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gs_phase_acc), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define GS_STAMP(i, dep) {}
#endif

// bits [s, s+64) of the 128-bit string (b:a), s in 0..63
__device__ __forceinline__ u64 gs_funnel(u64 a, u64 b, int s) { return (a >> s) | ((b << 1) << (63 - s)); }

// ASCII bases -> ballot planes of the reference's 2-bit codes (C0 G1 A2 T3, upper case only: CGAT.java:66-69), from a byte that was
// loaded earlier; lanes beyond the read hold GS_FILL ('C': code 0, valid -- planes 0 and not bad, without a range test).
// Straight-line, every ballot one integer compare: x = bits 1..2 of the byte (A 0, C 1, T 2, G 3), d = byte ^ the one upper-case
// letter with that x (a byte permute), code-hi = x even, code-lo = bit 2 of the byte; a byte that is not its letter (N, lower
// case, CR, NUL) has d != 0: planes 0 -- every window that holds such a byte is INVALID whatever its planes say -- and bad.
#define GS_FILL 0x43u
__device__ __forceinline__ void gs_word_from_byte(uint32_t c, u64 &hi, u64 &lo, u64 &bad) {
    const uint32_t x = (c >> 1) & 3u;
    const uint32_t d = c ^ __builtin_amdgcn_perm(0u, 0x47544341u, x | 0x0c0c0c00u);
    hi = __ballot(((x & 1u) | d) == 0u);
    lo = __ballot((((c ^ 4u) & 4u) | d) == 0u);
    bad = __ballot(d != 0u);
}

// one 64-base word of a read -> ballot planes
__device__ __forceinline__ void gs_load_word(const uint8_t *rd, int L, int w, int lane, u64 &hi, u64 &lo, u64 &bad) {
    const int j = 64 * w + lane;
    const uint32_t c = j < L ? rd[j] : GS_FILL;
    gs_word_from_byte(c, hi, lo, bad);
}

// ---------------------------------------------------------------------------------------------------
// statistics sinks: LDS-privatised (n_values <= GS_NV_LDS) or direct global atomics
// ---------------------------------------------------------------------------------------------------
// a wave's place in the record array: it takes 64 slots at a time with one atomic and fills them read by read.  The
// cursor lives in LDS (the scalar registers of the read loop are all taken): [0] = base of the chunk, [1] = slots used
// (64: no chunk in hand)
typedef unsigned long long GsRecCursor[2];

struct GsStats {
    u64 *sums;      // [nv][GS_N_SUMS]
    u64 *maxk;      // [nv]
    double *dsums;  // [nv][GS_N_DCOLS]
    // the taxonomy next to the counters: LDS copies when the counters are in LDS (a tree walk is a chain of dependent
    // loads; from HBM/L2 each link costs the better part of a microsecond), the store's arrays otherwise
    const int32_t *parent, *tin, *tout;
    __device__ __forceinline__ void add(int vi, int col, u64 v) const {
        if (!(GS_ABLATE & 8)) atomicAdd(&sums[(size_t)vi * GS_N_SUMS + col], v);
    }
    __device__ __forceinline__ void max(int vi, u64 key) const {
        if (!(GS_ABLATE & 8)) atomicMax(&maxk[vi], key);
    }
    __device__ __forceinline__ void dadd(int vi, int col, double v) const {
        if (!(GS_ABLATE & 8)) atomicAdd(&dsums[(size_t)vi * GS_N_DCOLS + col], v);
    }
    // contig flush (FastqKMerMatcher.java:396-411 / :457-471)
    __device__ __forceinline__ void contig(int vi, int len, u64 key_lo) const {
        add(vi, GS_S_KMERS, (u64)len);
        add(vi, GS_S_CONTIGS, 1);
        add(vi, GS_S_CONTIG_LEN_SQ_SUM, (u64)len * (u64)len);
        max(vi, ((u64)len << 40) | key_lo);
    }
};

// ---------------------------------------------------------------------------------------------------
// store probe
// ---------------------------------------------------------------------------------------------------
typedef unsigned long long gs_u64x2 __attribute__((ext_vector_type(2)));

// The fused kernels look at the first half of a bucket first (two loads, four compares, half the registers) and run
// at 8 waves per SIMD with 64 VGPRs.  Measured against whole-bucket probes at 6 waves: bench stream equal (10.26 ms),
// miss-only stream 5.94 -> 5.56 ms, 47 M-k-mer store 16.3 -> 16.0 ms.
#ifndef GS_HALF_BUCKETS
#define GS_HALF_BUCKETS 1
#endif
#ifndef GS_WAVES
#define GS_WAVES 8
#endif

// (Software pipelining across the reads of a wave -- offsets two reads ahead, bases one read ahead -- was measured on
// MI355X in round 1 and again in round 2 on the record layout, also on the HBM-resident 47 M-k-mer store: within noise
// (7.67 / 7.79 ms, 10.45 / 10.53 ms) while costing 4 more spilled VGPRs; the code is gone.)

#ifndef GS_NT_TABLE
// Measured on MI355X: non-temporal bucket loads stop the four dwordx4 loads of one 64-byte line from sharing a
// single L2 request (the match kernel ran 2.2x slower), so the default is plain loads.
#define GS_NT_TABLE 0
#endif

struct GsBucket {
    gs_u64x2 q[4];
};

__device__ __forceinline__ void gs_load_bucket(const u64 *table, u64 bkt, GsBucket &b) {
    const gs_u64x2 *p = reinterpret_cast<const gs_u64x2 *>(table + bkt * GS_SLOTS_PER_BUCKET);
#if GS_NT_TABLE
    b.q[0] = __builtin_nontemporal_load(p);
    b.q[1] = __builtin_nontemporal_load(p + 1);
    b.q[2] = __builtin_nontemporal_load(p + 2);
    b.q[3] = __builtin_nontemporal_load(p + 3);
#else
    b.q[0] = p[0];
    b.q[1] = p[1];
    b.q[2] = p[2];
    b.q[3] = p[3];
#endif
}

// returns true when the probe is finished (hit, or miss proven by a non-full bucket).
// slot = rem << (vbits+3) | disp << (vbits+1) | (vi+1) << 1 | seen, with vbits+3 < 32: a match has an equal high
// dword and a low dword that differs from `want` (value and seen fields 0) by x = 2(vi+1) + seen, i.e.
// 0 <= x-2 < 2*vmask.  On a hit vs = 2*vi + seen.  Buckets fill front to back, so "full" is "slot 7 is occupied".
__device__ __forceinline__ bool gs_match_bucket(const GsBucket &b, u64 want, uint32_t vmask2, int &vs, int &slot) {
    const u64 s[8] = {b.q[0].x, b.q[0].y, b.q[1].x, b.q[1].y, b.q[2].x, b.q[2].y, b.q[3].x, b.q[3].y};
    // two instructions per slot: mask the value/seen field away and compare all 64 bits.  An empty slot (0) can only
    // "match" want == 0 and is told apart afterwards by its zero value field.
    const u64 keep = ~(u64)(vmask2 + 1u);  // vmask2 + 1 = 2^(vbits+1) - 1: the (vi+1, seen) field
    int hit = -1;
    uint32_t low = 0;
#pragma unroll
    for (int j = 7; j >= 0; j--) {  // buckets fill front to back: the lowest match is the real entry
        const bool m = (s[j] & keep) == want;
        low = m ? (uint32_t)s[j] : low;
        hit = m ? j : hit;
    }
    const int x = (int)(low & (vmask2 + 1u)) - 2;  // 2 vi + seen, or < 0 for the empty slot
    if (hit >= 0 && x >= 0) {
        vs = x;
        slot = hit;
        return true;
    }
    return s[7] == 0;
}

// Half a bucket (slots 0-3 or 4-7, 32 bytes).  Buckets fill front to back and hold 1.5 .. 3 keys on average, so the
// first half answers most probes with two load instructions, four slot compares and half the registers.
struct GsHalf {
    gs_u64x2 q[2];
};

__device__ __forceinline__ void gs_load_half(const u64 *table, u64 bkt, int half, GsHalf &b) {
    const gs_u64x2 *p = reinterpret_cast<const gs_u64x2 *>(table + bkt * GS_SLOTS_PER_BUCKET) + 2 * half;
    b.q[0] = p[0];
    b.q[1] = p[1];
}

// as gs_match_bucket over four slots; finished = hit, or the half's last slot is empty (nothing can follow it)
__device__ __forceinline__ bool gs_match_half(const GsHalf &b, u64 want, uint32_t vmask2, int &vs, int &slot) {
    const u64 s[4] = {b.q[0].x, b.q[0].y, b.q[1].x, b.q[1].y};
    const u64 keep = ~(u64)(vmask2 + 1u);
    int hit = -1;
    uint32_t low = 0;
#pragma unroll
    for (int j = 3; j >= 0; j--) {
        const bool m = (s[j] & keep) == want;
        low = m ? (uint32_t)s[j] : low;
        hit = m ? j : hit;
    }
    const int x = (int)(low & (vmask2 + 1u)) - 2;
    if (hit >= 0 && x >= 0) {
        vs = x;
        slot = hit;
        return true;
    }
    return s[3] == 0;
}

// table hash of the k-mer whose forward planes are (fhi, flo): representative orientation, then the Feistel mix
__device__ __forceinline__ u64 gs_kmer_hash(uint32_t fhi, uint32_t flo, int k, uint32_t kmask) {
    uint32_t a, b;
    gs_rep_planes(fhi, flo, k, kmask, a, b);
    return gs_mix_planes(a, b);
}

// ---------------------------------------------------------------------------------------------------
// tree helpers (value index == node id; tin/tout pre-order interval)
// ---------------------------------------------------------------------------------------------------
// is `a` an ancestor-or-self of `x`  (SmallTaxTree.isAncestorOf(x, a), SmallTaxTree.java:242-252)
__device__ __forceinline__ bool gs_anc_or_self(int a_tin, int a_tout, int x_tin) { return a_tin <= x_tin && x_tin < a_tout; }

template <typename Tree>
__device__ __forceinline__ int gs_lca(const Tree &t, int a, int b) {  // SmallTaxTree.java:263-289, -1 == null
    if (a == b) return a;
    if (a < 0 || b < 0) return -1;
    const int tb = t.tin[b];
    while (a >= 0 && !gs_anc_or_self(t.tin[a], t.tout[a], tb)) a = t.parent[a];
    return a;
}

// per-wave scratch of the long-read kernel, always accessed at agent scope (L1 bypass)
__device__ __forceinline__ int gs_sc_load(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gs_sc_store(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---------------------------------------------------------------------------------------------------
// 128 k-mer positions [base, base+128) of one read: planes -> minimizer -> gate -> record line (or bucket line) -> node.
// node[s] for position base + 64 s + lane: value index (hit), GS_NODE_MISS, GS_NODE_INVALID (window holds a
// non-CGAT byte) or GS_NODE_NONE (position >= max).  Hits are marked on the spot (mk): the unique-k-mer "seen" bit of
// the slot / record offset that was just read (KMerUniqueCounterBits.putInlined; a k-mer that is already marked costs
// nothing, a stale copy only repeats the atomic) and the per-k-mer hit counter (maxKMerResCounts > 0).
// ---------------------------------------------------------------------------------------------------
struct GsMark {
    int count_unique;
    uint32_t *hit_counts;  // [table slots | GS_REC_SLOTS per record bucket] or nullptr
    // striped store: the store's memory is read-only (foreign stripes belong to other GPUs); the seen bits are this run's
    uint32_t *tab_seen;    // one bit per table slot
    uint32_t *rec_seen;    // one word per record bucket
};

// where a striped store keeps the pointers of its stripes: behind the two hash rows of the wave's LDS block
// (GS_STRIPE_TABLE): GS_MAX_STRIPES record stripes, then GS_MAX_STRIPES table stripes, all biased (gs_layout.h)
#define GS_ROW 160  // one LDS row: 128 + 16 positions, padded
#define GS_STRIPE_WORDS (4 * GS_MAX_STRIPES)
__device__ __forceinline__ const u64 *gs_rec_stripe(const GsDbDev &db, const uint32_t *wave_g, uint32_t bucket) {
    return reinterpret_cast<const u64 *const *>(wave_g + 2 * GS_ROW)[(bucket * db.n_parts) >> db.rec_bits];
}
__device__ __forceinline__ const u64 *gs_tab_stripe(const GsDbDev &db, const uint32_t *wave_g, uint32_t bucket) {
    return reinterpret_cast<const u64 *const *>(wave_g + 2 * GS_ROW)[GS_MAX_STRIPES + (((bucket >> 2) * db.n_parts) >> (db.bucket_bits - 2))];
}

template <bool STRIPED>
__device__ __forceinline__ void gs_take_hit(const GsDbDev &db, uint32_t slot, int vs, int &node, const GsMark &mk) {
    node = vs >> 1;
    if (STRIPED) {
        if (mk.count_unique) {
            uint32_t *w = mk.tab_seen + (slot >> 5);
            const uint32_t bit = 1u << (slot & 31);
            if ((*w & bit) == 0) atomicOr(w, bit);
        }
    } else if (mk.count_unique && (vs & 1) == 0)
        atomicOr(const_cast<u64 *>(db.table) + slot, 1ULL);
    if (mk.hit_counts != nullptr) atomicAdd(mk.hit_counts + slot, 1u);
}

// the rest of a table lookup for the lanes in `pending` (their k-mer was not decided by the first half of its home
// bucket, or that half has not been looked at yet: FIRST): second half, then the displaced buckets
template <bool FIRST, bool STRIPED>
__device__ __forceinline__ void gs_lookup_rest(const GsDbDev &db, uint32_t bkt, u64 want, uint32_t vmask2, bool pending, int &node,
                                               const GsMark &mk, const uint32_t *wave_g) {
    const uint32_t bmask = (uint32_t)db.bucket_mask;
#pragma unroll
    for (int half = FIRST ? 0 : 1; half < 2; half++) {
        if (__ballot(pending) != 0) {  // (second half: more than four entries in the home bucket)
            if (pending) {
                GsHalf t;
                gs_load_half(STRIPED ? gs_tab_stripe(db, wave_g, bkt) : db.table, bkt, half, t);
                int vs = -1, sl = 0;
                const bool done = gs_match_half(t, want, vmask2, vs, sl);
                if (vs >= 0) gs_take_hit<STRIPED>(db, bkt * GS_SLOTS_PER_BUCKET + 4 * half + sl, vs, node, mk);
                pending = !done;
            }
        }
    }
    // rare: home bucket full without a match -> walk the displaced buckets
    for (int disp = 1; disp <= GS_MAX_DISP && __ballot(pending) != 0; disp++) {
        if (pending) {
            const uint32_t b2 = (bkt + (uint32_t)disp) & bmask;
            GsBucket t;
            gs_load_bucket(STRIPED ? gs_tab_stripe(db, wave_g, b2) : db.table, b2, t);
            int vs = -1, sl = 0;
            const bool done = gs_match_bucket(t, want | ((u64)disp << (db.vbits + 1)), vmask2, vs, sl);
            if (vs >= 0) gs_take_hit<STRIPED>(db, b2 * GS_SLOTS_PER_BUCKET + sl, vs, node, mk);
            pending = !done;
        }
    }
}

// minimizer of the k-mers at positions base + 64 s + lane through the wave's LDS row: rank of every 15-mer of
// positions base .. base+143, then per lane the minimum over its k-14 positions (one min3 chain; the rank carries the
// row index, so the minimum is the position as well).  Returns the offset of the chosen 15-mer inside the lane's k-mer.
// wave_g: two rows of GS_ROW words per wave -- the ranks, and behind them (canonical 15-mer << 1 | strand) of every position,
// which the lane that picks a position reads back instead of recomputing it.  cf[s] = that word for the lane's minimizer.
// NS: sub-rounds of 64 positions per trip (2: the kernels' iteration of 128 positions; 3, 4: gs_match_wide_kernel, reads of up to
// 192 / 256 positions in ONE trip).  The rank carries its row index in eight bits, so beyond two sub-rounds there are TWO rank rows:
// A for positions 0 .. 143 (read by sub-rounds 0 and 1), B for positions 128 .. 64 NS + 15 with the index counted from 128 (read by
// sub-rounds 2 and 3); the 16 positions both need are written twice.  Layout of the wave's words: [A: GS_ROW][B: GS_ROW, NS > 2 only]
// [canonical 15-mers of all positions: 64 NS + 16].
#define GS_WIDE_WORDS(NS) (2 * GS_ROW + 64 * (NS) + 32)
template <int KC, int NS = 2>
__device__ __forceinline__ void gs_wave_minimizers(const u64 (&Bhi)[NS + 1], const u64 (&Blo)[NS + 1], const uint32_t (&fhi)[NS],
                                                   const uint32_t (&flo)[NS], const uint32_t (&rhi)[NS], const uint32_t (&rlo)[NS], int k,
                                                   int lane, uint32_t *wave_g, int (&p)[NS], uint32_t (&cf)[NS]) {
    static_assert(NS >= 2 && NS <= 4, "two rank rows of 144 positions");
    constexpr bool TWO = NS > 2;
    constexpr int ROWB = GS_ROW, CAN = TWO ? 2 * GS_ROW : GS_ROW;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        // gs_lmer_canon of the k-mer's first 15-mer; its reverse complement is the top 15 bases of the k-mer's (rhi, rlo: the
        // orientation step needs them anyway), and "the smaller of f and r, low bit = f was it" is min(2f + 1, 2r)
        const uint32_t f = ((fhi[s] & 0x7fffu) << GS_MIN_L) | (flo[s] & 0x7fffu);
        const uint32_t rr = ((rhi[s] >> (k - GS_MIN_L)) << GS_MIN_L) | (rlo[s] >> (k - GS_MIN_L));
        const uint32_t c = min((f << 1) | 1u, rr << 1);
        const uint32_t h = gs_canon_hash(c >> 1);
        if (!TWO || s < 2) wave_g[64 * s + lane] = gs_lmer_rank(h, (uint32_t)(64 * s + lane));
        if (TWO && s >= 2) wave_g[ROWB + 64 * (s - 2) + lane] = gs_lmer_rank(h, (uint32_t)(64 * (s - 2) + lane));
        if (TWO && s == 2 && lane < 16) wave_g[128 + lane] = gs_lmer_rank(h, (uint32_t)(128 + lane));
        wave_g[CAN + 64 * s + lane] = c;
    }
    if (lane < 16) {
        const uint32_t c = gs_lmer_canon((uint32_t)(Bhi[NS] >> lane) & 0x7fffu, (uint32_t)(Blo[NS] >> lane) & 0x7fffu);
        const uint32_t h = gs_canon_hash(c >> 1);
        if (!TWO)
            wave_g[64 * NS + lane] = gs_lmer_rank(h, (uint32_t)(64 * NS + lane));
        else
            wave_g[ROWB + 64 * (NS - 2) + lane] = gs_lmer_rank(h, (uint32_t)(64 * (NS - 2) + lane));
        wave_g[CAN + 64 * NS + lane] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int w = k - GS_MIN_L + 1;
    // The ranks of a lane's k - 14 positions are read in ONE batch per sub-round (the loads are independent; left to itself the
    // compiler waits after every ds_read2 because of the 64-register budget: nine LDS round trips in a row per sub-round), then
    // folded with min3.
    constexpr int ND = KC ? KC - GS_MIN_L + 1 : 32 - GS_MIN_L;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const uint32_t *row = (TWO && s >= 2) ? wave_g + ROWB + 64 * (s - 2) + lane : wave_g + 64 * s + lane;
        uint32_t g[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) g[d] = (KC || d < w) ? row[d] : 0xffffffffu;
        __builtin_amdgcn_sched_group_barrier(0x100, (ND + 1) / 2, 0);  // the DS reads first ...
        uint32_t mn = g[0];
#pragma unroll
        for (int d = 1; d < ND; d++) mn = g[d] < mn ? g[d] : mn;
        __builtin_amdgcn_sched_group_barrier(0x002, ND, 0);            // ... then the min chain
        const int idx = (int)(mn & 0xffu) + ((TWO && s >= 2) ? 128 : 0);
        cf[s] = wave_g[CAN + idx];
        p[s] = idx - (64 * s + lane);
    }
    __builtin_amdgcn_wave_barrier();  // the rows are rewritten by the next iteration / read
}

#define GS_ACT(m) __builtin_amdgcn_inverse_ballot_w64(m)  // a wave-level mask as a per-lane condition
// CTX: is the gate keyed by gs_gate_ctx_key?  0 never (the launcher has looked), 1 always, 2 ask the store (GsDbDev::mgate_ctx)
// AGG: the unique-k-mer marks of the k-mers that share a record line leave as ONE atomic per line.  A device-scope atomic is
// executed on the memory side of the L2s (eight XCDs, eight L2s): every one is a request to the fabric, and on a store that does not
// fit the caches they were HALF of the kernel's fabric requests (473 M k-mers: 30 read + 36 write requests per read, 61 writes per
// read from the store -- one per first-seen k-mer; tools/huge_lines.py), with the fabric's line rate the bound (0.89).  The lanes of
// a minimizer run OR their offset bits into a word of the wave's LDS rows (free again after the minimizer scan), the lane that owns
// the lowest bit sends the word: ~13 atomics per read from the store instead of ~60.
template <int KC, bool STRIPED, int CTX = 2, bool AGG = false, int NS = 2>
__device__ __forceinline__ void gs_probe_planes(const GsDbDev &db, const u64 (&Bhi)[NS + 1], const u64 (&Blo)[NS + 1],
                                                const u64 (&Bbad)[NS + 1], int base, int max, int lane, int (&node)[NS],
                                                uint32_t *wave_g, const GsMark &mk) {
    static_assert(NS == 2 || !AGG, "the seen-bit accumulators are laid out for two sub-rounds");
    const int k = KC ? KC : db.k;  // KC = compile-time k of the specialised kernels (0: any k)
    const uint32_t kmask = (1u << k) - 1u;
    const uint32_t vmask2 = 2u * ((1u << db.vbits) - 1u);
    const int shift_rem = (int)db.vbits + 3;
    const uint32_t bmask = (uint32_t)db.bucket_mask;  // n_buckets <= 2^29
    bool any_bad = false;
#pragma unroll
    for (int w = 0; w <= NS; w++) any_bad = any_bad || Bbad[w] != 0;
    uint32_t fhi[NS], flo[NS];
    // Which lanes hold a live k-mer is kept as a wave-level MASK (a scalar register pair), not as a per-lane flag: "position < max"
    // is a mask the scalar unit builds from the read length, every later condition is one vector compare whose result is a
    // mask already, and a mask conditions a lane directly (inverse ballot) -- no flag is materialised in a vector register and
    // compared again.
    u64 act[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const int nv = max - base - 64 * s;  // valid positions of this sub-round
        const u64 vm = nv >= 64 ? ~0ULL : (nv <= 0 ? 0ULL : ((1ULL << nv) - 1ULL));
        fhi[s] = (uint32_t)gs_funnel(Bhi[s], Bhi[s + 1], lane) & kmask;
        flo[s] = (uint32_t)gs_funnel(Blo[s], Blo[s + 1], lane) & kmask;
        act[s] = vm;
        node[s] = GS_ACT(vm) ? GS_NODE_MISS : GS_NODE_NONE;
        if (any_bad) {  // almost every read is clean: the per-lane window test sits behind a wave-uniform branch
            const u64 wb = __ballot(((uint32_t)gs_funnel(Bbad[s], Bbad[s + 1], lane) & kmask) != 0u) & vm;
            act[s] = vm & ~wb;
            node[s] = GS_ACT(wb) ? GS_NODE_INVALID : node[s];
        }
    }
    if (db.mgate != nullptr) {
        // minimizer gate: lanes that share a minimizer read the same gate word -> one request
        int mp[NS];
        uint32_t cf[NS];
        GS_STAMP(2, fhi[1] ^ flo[1])
        uint32_t rhi[NS], rlo[NS];  // reverse complement of the k-mer
#pragma unroll
        for (int s = 0; s < NS; s++) {
            rhi[s] = __brev(fhi[s]) >> (32 - k);
            rlo[s] = (__brev(flo[s]) >> (32 - k)) ^ kmask;
        }
        gs_wave_minimizers<KC, NS>(Bhi, Blo, fhi, flo, rhi, rlo, k, lane, wave_g, mp, cf);
        GS_STAMP(3, cf[0] ^ cf[1])
        uint32_t gh[NS], ohi[NS], olo[NS];
        int j[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) gs_min_oriented_cf(cf[s], fhi[s], flo[s], rhi[s], rlo[s], k, mp[s], gh[s], ohi[s], olo[s], j[s]);
        // Lanes that look into their minimizer's second bucket as well.  Every store carries the hint bits (gs_mgate_hint); they
        // are USED where a record line is dear -- a context-keyed store (hundreds of millions of k-mers, lines from HBM) and a
        // striped one (lines over xGMI) --: on a store whose records sit in the caches the second line costs less than the
        // test (measured on configs[1]: 6.74 ms without, 6.94 ms with it; 47 M store 8.66 / 8.68; 473 M store 15.39 / 15.03).
        const bool use_hint = STRIPED || CTX == 1 || (CTX == 2 && db.mgate_ctx);
        u64 two[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) two[s] = ~0ULL;
        if ((GS_ABLATE & 4) == 0) {
            // the gate words of both sub-rounds are requested together (lanes without a live k-mer read word 0): one round
            // trip, no exec-mask regions
            uint32_t gw[NS];
            uint32_t gk[NS];  // what the gate is keyed by: the minimizer, or (big stores) the minimizer + four bases next to it
#pragma unroll
            for (int s = 0; s < NS; s++) gk[s] = (CTX == 1 || (CTX == 2 && db.mgate_ctx)) ? gs_gate_ctx_key(gh[s], ohi[s], olo[s], j[s], k) : gh[s];
#pragma unroll
            for (int s = 0; s < NS; s++)
                gw[s] = db.mgate[GS_ACT(act[s]) ? ((CTX == 1 || (CTX == 2 && db.mgate_ctx)) ? gs_mgate_word_ctx(gk[s], db.mgate_bits) : gs_mgate_word(gk[s], db.mgate_bits)) : 0u];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const uint32_t bits = gs_mgate_bits(gk[s]);
                act[s] &= __ballot((gw[s] & bits) == bits);  // no false negatives
                if (use_hint) two[s] = __ballot((gw[s] & gs_mgate_hint(gk[s])) != 0u);  // a window of this key may sit in the second bucket
            }
            GS_STAMP(4, gw[0] ^ gw[1])
        }
        if (GS_ABLATE & 6) {  // keep the values alive, look nothing up
#pragma unroll
            for (int s = 0; s < NS; s++)
                if (GS_ACT(act[s]) && (gh[s] ^ ohi[s] ^ olo[s] ^ (uint32_t)j[s]) == 0x12345u) node[s] = 0;
            return;
        }
        if (STRIPED || db.rec != nullptr) {
            // ---- super-k-mer records: per candidate bucket of the minimizer one 16-byte load of the window planes + the
            // 8-byte word that holds this offset's value.  The first bucket always; the second one only where the gate word
            // carries the key's hint bit (two windows in three sit in their first bucket: a third fewer line requests, which
            // is what bounds a store that does not fit the caches).  Both loads of a sub-round are in flight together and the
            // compares are straight-line code (bitwise, no short-circuit branches: every branch is an exec-mask save /
            // restore on the scalar unit, which this kernel keeps as busy as the vector unit)
            u64 any_act = 0;
#pragma unroll
            for (int s = 0; s < NS; s++) any_act |= act[s];
            if (any_act == 0) return;  // nothing passed the gate: a read that is not from the store
            if (AGG && mk.count_unique) {  // accumulators of the seen bits: 80 minimizer positions x 2 buckets per sub-round
#pragma unroll
                for (int q = 0; q < 5; q++) wave_g[64 * q + lane] = 0u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // (Tried in round 3 and dropped, twice: touching sub-round 1's lines -- one dword each -- before sub-round 0's loads go
            // out, so that the second record round trip (3 400 + 3 200 of 24 900 wave cycles per read on the 473 M-k-mer store,
            // tools/phase_times.sh huge) is an L2 hit.  With plain loads, which the compiler is free to sink: configs[1] 6.80 ->
            // 7.05 ms, 47 M store 8.72 -> 9.08, 473 M store 15.0 -> 15.2.  With loads the compiler does not see (inline asm, issued
            // first for certain; big stores only): 473 M store 15.0 -> 15.2 ms, striped 7.24 -> 7.40.  The round trip that was
            // taken out does not come off the kernel's time.)
#pragma unroll
            for (int s = 0; s < NS; s++) {
                bool pending = false;
                if (GS_ACT(act[s])) {
                    const uint32_t jj = (uint32_t)j[s];
                    const int jw = (int)(jj * 11u) >> 5;  // j / 3 for j <= 16
                    const uint32_t b0 = gs_rec_bucket(gh[s], db.rec_bits, 0), b1 = gs_rec_bucket(gh[s], db.rec_bits, 1);
                    // the stripe's (biased) base pointer from the wave's copy of the table, behind the hash rows
                    const u64 *r0 = (STRIPED ? gs_rec_stripe(db, wave_g, b0) : db.rec) + (u64)b0 * GS_REC_WORDS;
                    gs_u64x2 A1 = {0, 0};  // (no valid bit: matches nothing)
                    u64 V1 = 0;
                    if (!use_hint || GS_ACT(two[s])) {
                        const u64 *r1 = (STRIPED ? gs_rec_stripe(db, wave_g, b1) : db.rec) + (u64)b1 * GS_REC_WORDS;
                        A1 = *reinterpret_cast<const gs_u64x2 *>(r1);
                        V1 = r1[2 + jw];
                    }
                    const gs_u64x2 A0 = *reinterpret_cast<const gs_u64x2 *>(r0);
                    const u64 V0 = r0[2 + jw];
                    // window bits [j, j + k) of a plane: j + k <= 2k - 15, so the seen / valid bits above the window stay out
                    // of the low k bits of the shifted word; one 32-bit funnel shift (j <= 16) per plane
                    const uint32_t x0 = __builtin_amdgcn_alignbit((uint32_t)(A0.x >> 32), (uint32_t)A0.x, jj);
                    const uint32_t y0 = __builtin_amdgcn_alignbit((uint32_t)(A0.y >> 32), (uint32_t)A0.y, jj);
                    const uint32_t x1 = __builtin_amdgcn_alignbit((uint32_t)(A1.x >> 32), (uint32_t)A1.x, jj);
                    const uint32_t y1 = __builtin_amdgcn_alignbit((uint32_t)(A1.y >> 32), (uint32_t)A1.y, jj);
                    const uint32_t fbit = 1u << (GS_REC_WIN_BITS - 32 + jj);  // seen (w0) / valid (w1) bit of offset j, high dword
                    const bool ok0 = ((((x0 ^ ohi[s]) | (y0 ^ olo[s])) & kmask) == 0) & (((uint32_t)(A0.y >> 32) & fbit) != 0);
                    const bool ok1 = ((((x1 ^ ohi[s]) | (y1 ^ olo[s])) & kmask) == 0) & (((uint32_t)(A1.y >> 32) & fbit) != 0);
                    const bool hit = ok0 | ok1;  // (an eligible k-mer is filed in exactly one window)
                    const u64 V = ok0 ? V0 : V1;
                    const uint32_t seen_hi = ok0 ? (uint32_t)(A0.x >> 32) : (uint32_t)(A1.x >> 32);
                    const uint32_t rb = ok0 ? b0 : b1;
                    const int val = (int)((V >> (GS_REC_VAL_BITS * ((int)jj - 3 * jw))) & (GS_REC_MAX_VALUES - 1));
                    node[s] = hit ? val : node[s];
                    if (hit) {
                        if (STRIPED && AGG) {
                            if (mk.count_unique && ((mk.rec_seen[rb] >> jj) & 1u) == 0) {
                                uint32_t *acc = wave_g + GS_ROW * s + 2 * (mp[s] + lane) + (ok0 ? 0 : 1);
                                atomicOr(acc, 1u << jj);
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                const uint32_t w = *(volatile uint32_t *)acc;
                                if ((w & (0u - w)) == (1u << jj)) atomicOr(mk.rec_seen + rb, w);
                            }
                        } else if (STRIPED) {
                            if (mk.count_unique && ((mk.rec_seen[rb] >> jj) & 1u) == 0) atomicOr(mk.rec_seen + rb, 1u << jj);
                        } else if (AGG) {
                            if (mk.count_unique && (seen_hi & fbit) == 0) {
                                // (the lanes of a minimizer occurrence share both buckets; occurrence = mp + lane inside the sub-round, < 80)
                                uint32_t *acc = wave_g + GS_ROW * s + 2 * (mp[s] + lane) + (ok0 ? 0 : 1);
                                atomicOr(acc, 1u << jj);
                                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                                const uint32_t w = *(volatile uint32_t *)acc;
                                if ((w & (0u - w)) == (1u << jj))  // the owner of the lowest bit sends the run's bits
                                    atomicOr(const_cast<u64 *>(db.rec) + (u64)rb * GS_REC_WORDS, (u64)w << GS_REC_WIN_BITS);
                            }
                        } else if (mk.count_unique && (seen_hi & fbit) == 0)
                            atomicOr(const_cast<u64 *>(db.rec) + (u64)rb * GS_REC_WORDS, 1ULL << (GS_REC_WIN_BITS + jj));
                        if (mk.hit_counts != nullptr)
                            atomicAdd(mk.hit_counts + ((u64)(bmask + 1u) * GS_SLOTS_PER_BUCKET + (u64)rb * GS_REC_SLOTS + (u64)jj), 1u);
                    }
                    pending = !hit & (((V0 | V1) & GS_REC_MORE) != 0);
                }
                GS_STAMP(8 + 2 * s, node[s])  // (phase 8 / 10: the record lines of sub-round 0 / 1)
                // a k-mer whose window found no bucket (or that has two strand views) lives in the table
                if (__ballot(pending) != 0) {
                    const u64 h = gs_kmer_hash(ohi[s], olo[s], k, kmask);
                    gs_lookup_rest<true, STRIPED>(db, (uint32_t)h & bmask, (h >> db.bucket_bits) << shift_rem, vmask2, pending, node[s], mk, wave_g);
                }
                GS_STAMP(9 + 2 * s, node[s])  // (phase 9 / 11: its walk of the overflow table)
            }
            return;
        }
    }
    // ---- the ordinary table: one 64-byte bucket line per k-mer (stores without records: k < 19, partitions, > 2^21 values)
    uint32_t bkt[NS];
    u64 want[NS];
    GsHalf bk[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
        const u64 h = gs_kmer_hash(fhi[s], flo[s], k, kmask);
        bkt[s] = (uint32_t)h & bmask;
        want[s] = (h >> db.bucket_bits) << shift_rem;
        if (db.mgate == nullptr && db.gate != nullptr) {  // word gate (L2-resident), no false negatives
            bool pass = false;
            if (GS_ACT(act[s])) {
                const u64 gbits = gs_gate_field_bits((uint32_t)(h >> GS_GATE_FIELD_SHIFT));
                pass = (db.gate[(h >> db.bucket_bits) & db.gate_mask] & gbits) == gbits;
            }
            act[s] = __ballot(pass);
        }
        if (GS_ACT(act[s])) gs_load_half(db.table, bkt[s], 0, bk[s]);  // both sub-rounds' loads in flight together
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        bool pending = false;
        if (GS_ACT(act[s])) {
            int vs = -1, sl = 0;
            const bool done = gs_match_half(bk[s], want[s], vmask2, vs, sl);
            if (vs >= 0) gs_take_hit<false>(db, bkt[s] * GS_SLOTS_PER_BUCKET + sl, vs, node[s], mk);
            pending = !done;
        }
        gs_lookup_rest<false, false>(db, bkt[s], want[s], vmask2, pending, node[s], mk, wave_g);
    }
}

// ---------------------------------------------------------------------------------------------------
// one read on one wave.  LONG = false: max <= 128 (one iteration, distinct nodes kept in registers).
// LONG = true: any length; tag/cnt are this wave's scratch rows of n_values ints, serial its read tag.
// WIDE = true: maxClassificationPaths in 65..128 (C/GSConfigKey.java:350 allows 1..128): candidate path i lives in
// lane i & 63 of register set i >> 6; with WIDE = false there is one set and lane = path.
// ---------------------------------------------------------------------------------------------------
template <bool LONG, bool FROM_NODES, int KC, bool WIDE, bool REC, bool STRIPED, int CTX = 2>
__device__ __forceinline__ void gs_process_read(const GsMatchParams &P, const GsStats &st, int64_t r, u64 off, int L,
                                                int lane, int (*s_dvi)[128], int (*s_dcnt)[128], int wave_in_block,
                                                int32_t *tag, int32_t *cnt, int serial, const uint32_t (&pre)[3],
                                                uint32_t *wave_g, unsigned long long *cur) {
    const GsDbDev &db = P.db;
    const int k = KC ? KC : db.k;
    const int max = L - k + 1;
    const uint8_t *rd = P.seq + off;
    int out_class = -1;
    int out_flags = 0;

    if (max > 0) {
        const int64_t read_no = P.first_read_no + r;
        const u64 key_lo = ((1ULL << 40) - 1) - ((u64)read_no & ((1ULL << 40) - 1));
        const int n_iter = LONG ? (max + 127) >> 7 : 1;
        // per-read state
        bool found = false;
        int n_miss = 0, bad_lo = 0;
        bool bad_hi = false;
        int carry_last = GS_NODE_NONE;  // node of the last position of the previous iteration
        int cur_start = 0;
        int dviA = -1, dviB = -1, dcntA = 0, dcntB = 0, nd = 0;  // !LONG: distinct hit nodes (cap 128 >= #contigs)
        int one_vi = -1, one_cnt = 0;                              // !LONG: the first of them, wave-uniform
        bool deferred = false;                                     // REC: its statistics go into a GsStatRec (P.stat_recs)
        int def_contigs = 0, def_max = 0, def_sq = 0;              // (at most 128 positions: 128^2 fits)
        int def_counted = 0, def_read_kmers = 0, def_tax_err = 0;
        constexpr int NP = WIDE ? 2 : 1;
        int path[NP], ptin[NP], ptout[NP], used = 0;            // candidate paths: path i in lane i & 63 of set i >> 6
#pragma unroll
        for (int h = 0; h < NP; h++) {
            path[h] = -1;
            ptin[h] = 0;
            ptout[h] = 0;
        }

        for (int it = 0; it < n_iter; it++) {
            const int base = it << 7;  // first k-mer position of this iteration
            // ---- 1. bases -> ballot planes (3 words cover 128 + k - 1 <= 158 bases)
            u64 Bhi[3], Blo[3], Bbad[3];
#pragma unroll
            for (int w = 0; w < 3; w++) {
                if (LONG)
                    gs_load_word(rd, L, 2 * it + w, lane, Bhi[w], Blo[w], Bbad[w]);
                else
                    gs_word_from_byte(pre[w], Bhi[w], Blo[w], Bbad[w]);
            }
            {   // bad-base census for the INVALID-iteration closed form; word 2 belongs to the next iteration
                const int q = max - 1;
                const int nw = (it == n_iter - 1) ? 3 : 2;
#pragma unroll
                for (int w = 0; w < 3; w++) {
                    if (w < nw) {
                        const int lo_bits = q - (base + 64 * w);  // positions of this word that are < q
                        const u64 m_lo = lo_bits >= 64 ? ~0ULL : (lo_bits <= 0 ? 0ULL : ((1ULL << lo_bits) - 1));
                        bad_lo += __popcll(Bbad[w] & m_lo);
                        bad_hi = bad_hi || ((Bbad[w] & ~m_lo) != 0);
                    }
                }
            }
            if (!LONG) GS_STAMP(1, (uint32_t)(Bhi[0] ^ Bhi[2]))
            // ---- 2/3. k-mers + probe, both sub-rounds in flight
            int node[2];
            if (FROM_NODES) {
                // DB-partitioned mode: the node of every position was looked up by the owner of its k-mer
                // (gs_probe_keys_kernel on the owning rank) and routed back; INVALID windows carry GS_NODE_INVALID
                const u64 pb = P.pos_off[r] + (u64)base;
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    const int p = base + 64 * s + lane;
                    int v = p < max ? P.nodes[pb + 64 * s + lane] : GS_NODE_NONE;
                    // the stream comes from other ranks: a value outside the store's range must not index anything
                    if (p < max && (v >= db.n_values || v < GS_NODE_INVALID)) v = GS_NODE_MISS;
                    node[s] = v;
                }
            } else {
                // (4a. unique k-mers and per-k-mer hit counters are marked by the probe itself)
                const GsMark mk = {P.count_unique, P.hit_counts, STRIPED ? P.bitmap : nullptr,
                                   STRIPED ? P.bitmap + ((db.bucket_mask + 1) * GS_SLOTS_PER_BUCKET >> 5) : nullptr};
#ifndef GS_AGG_ALL
#define GS_AGG_ALL 0
#endif
                gs_probe_planes<KC, STRIPED, CTX, (REC || GS_AGG_ALL) && !LONG>(db, Bhi, Blo, Bbad, base, max, lane, node, wave_g, mk);
            }

            if (!LONG) GS_STAMP(5, node[0] ^ node[1])
            const u64 hit0 = __ballot(node[0] >= 0), hit1 = __ballot(node[1] >= 0);
            found = found || ((hit0 | hit1) != 0);
            n_miss += __popcll(__ballot(node[0] == GS_NODE_MISS)) + __popcll(__ballot(node[1] == GS_NODE_MISS));

            // ---- 4b. contigs and distinct nodes.  Contig statistics are lane parallel: every lane that starts a hit
            // contig finds the next node change in the ballot masks and books the contig itself (:391-413).  Only
            // the DISTINCT hit nodes are walked one by one, in order of first appearance: reads1KMer counts a
            // (read, tax id) pair once (:434-439) and mergeReadTaxidPath (:568-586) is idempotent for a node it
            // has already seen (paths only ever move down the tree), so repeats need no work.  The per-node vote
            // count of the read (incCount, :380-388) is the number of positions holding that node.
            // the first distinct hit node; in ~90 % of the reads that hit a store of species-specific k-mers it is the
            // only one: its id and count stay in scalar registers (one_vi, one_cnt)
            u64 m0 = hit0, m1 = hit1;
#define GS_FIRST_NODE()                                                                                   \
    {                                                                                                     \
        const int j = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);                                \
        one_vi = j < 64 ? gs_readlane(node[0], j) : gs_readlane(node[1], j - 64);                         \
        const u64 e0 = __ballot(node[0] == one_vi), e1 = __ballot(node[1] == one_vi);                     \
        one_cnt = __popcll(e0) + __popcll(e1);                                                            \
        m0 &= ~e0;                                                                                        \
        m1 &= ~e1;                                                                                        \
        nd = 1;                                                                                           \
    }
            if (REC && (hit0 | hit1) != 0) {  // (needed before the contigs are booked: one record instead of atomics?)
                GS_FIRST_NODE()
                deferred = (m0 | m1) == 0 && P.stat_recs != nullptr;  // one tax id, counters in HBM: no atomics, one record
            }
            if ((GS_ABLATE & 1) == 0 && ((hit0 | hit1) != 0 || carry_last >= 0)) {
                int prev[2];
                {
                    const int up0 = __shfl_up(node[0], 1);
                    const int up1 = __shfl_up(node[1], 1);
                    const int last0 = gs_readlane(node[0], 63);
                    prev[0] = lane == 0 ? carry_last : up0;
                    prev[1] = lane == 0 ? last0 : up1;
                }
                // (positions >= max hold NONE: the first of them differs from its predecessor and is masked away; wave-level masks
                // throughout, as in gs_probe_planes)
                const int nv0 = max - base, nv1 = max - base - 64;
                const u64 vm0 = nv0 >= 64 ? ~0ULL : (nv0 <= 0 ? 0ULL : ((1ULL << nv0) - 1ULL));
                const u64 vm1 = nv1 >= 64 ? ~0ULL : (nv1 <= 0 ? 0ULL : ((1ULL << nv1) - 1ULL));
                const u64 chg0 = __ballot(node[0] != prev[0]) & vm0;
                const u64 chg1 = __ballot(node[1] != prev[1]) & vm1;
                // a hit contig left open by the previous iteration ends at the first change of this one
                if (LONG && carry_last >= 0 && (chg0 | chg1) != 0) {
                    const int q = chg0 ? __builtin_ctzll(chg0) : 64 + __builtin_ctzll(chg1);
                    if (lane == 0) st.contig(carry_last, base + q - cur_start, key_lo);
                }
                {   // hit contigs that start here and end before the iteration does
                    const u64 a0 = (chg0 >> 1) >> lane, a1 = (chg1 >> 1) >> lane;  // changes above this lane
                    const int e1 = chg1 ? 64 + __builtin_ctzll(chg1) : -1;
                    const int end0 = a0 ? lane + 1 + __builtin_ctzll(a0) : e1;
                    const int end1 = a1 ? 64 + lane + 1 + __builtin_ctzll(a1) : -1;
                    // a lane books a contig if it starts a hit contig (hit and change) and a later change closes it: in word 0
                    // any change of word 1 or a change above the lane, in word 1 a change above the lane
                    const u64 c0m = hit0 & chg0 & (chg1 ? ~0ULL : (chg0 ? (1ULL << (63 - __builtin_clzll(chg0))) - 1ULL : 0ULL));
                    const u64 c1m = hit1 & chg1 & (chg1 ? (1ULL << (63 - __builtin_clzll(chg1))) - 1ULL : 0ULL);
                    const bool c0 = GS_ACT(c0m), c1 = GS_ACT(c1m);
                    if (REC && deferred) {  // every contig belongs to one_vi: add them up in scalar registers
                        const int len0 = c0 ? end0 - lane : 0, len1 = c1 ? end1 - 64 - lane : 0;
                        for (u64 sm = c0m; sm; sm &= sm - 1) {
                            const int l = gs_readlane(len0, __builtin_ctzll(sm));
                            def_contigs++;
                            def_sq += l * l;
                            def_max = l > def_max ? l : def_max;
                        }
                        for (u64 sm = c1m; sm; sm &= sm - 1) {
                            const int l = gs_readlane(len1, __builtin_ctzll(sm));
                            def_contigs++;
                            def_sq += l * l;
                            def_max = l > def_max ? l : def_max;
                        }
                    } else {
                        if (c0) st.contig(node[0], end0 - lane, key_lo);
                        if (c1) st.contig(node[1], end1 - 64 - lane, key_lo);
                    }
                }
                // start of the contig that is still open at the end of this iteration
                if (chg1)
                    cur_start = base + 127 - __builtin_clzll(chg1);
                else if (chg0)
                    cur_start = base + 63 - __builtin_clzll(chg0);
                // distinct hit nodes of this iteration, in order of first appearance (the first one is known already;
                // the candidate-path state is set up only if a second node follows)
                if (!LONG) {
                    if (!REC) GS_FIRST_NODE()
                    if (!(REC && deferred) && lane == 0) st.add(one_vi, GS_S_READS_1KMER, 1);  // first k-mer of this tax id in the read (:434-439)
                    if ((m0 | m1) != 0) {
                        if (lane == 0) {
                            dviA = one_vi;
                            dcntA = one_cnt;
                        }
                        if (P.classify) {  // mergeReadTaxidPath into the empty list (max_paths >= 1)
                            if (lane == 0) {
                                path[0] = one_vi;
                                ptin[0] = st.tin[one_vi];
                                ptout[0] = st.tout[one_vi];
                            }
                            used = 1;
                        }
                    }
                }
                while ((m0 | m1) != 0) {
                    const int j = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
                    const int nvj = j < 64 ? gs_readlane(node[0], j) : gs_readlane(node[1], j - 64);
                    const u64 e0 = __ballot(node[0] == nvj), e1 = __ballot(node[1] == nvj);
                    const int c = __popcll(e0) + __popcll(e1);
                    m0 &= ~e0;
                    m1 &= ~e1;
                    bool first = true;
                    if (LONG) {
                        int f = 0;
                        if (lane == 0) {
                            f = gs_sc_load(tag + nvj) != serial;
                            if (f) gs_sc_store(tag + nvj, serial);
                            gs_sc_store(cnt + nvj, f ? c : gs_sc_load(cnt + nvj) + c);
                        }
                        first = gs_rfl(f) != 0;
                    } else {  // one iteration per read: every node of the walk is new
                        if (nd < 64) {
                            if (lane == nd) {
                                dviA = nvj;
                                dcntA = c;
                            }
                        } else if (lane == nd - 64) {
                            dviB = nvj;
                            dcntB = c;
                        }
                        nd++;
                    }
                    if (first) {
                        // first k-mer of this tax id in the read (:434-439)
                        if (lane == 0) st.add(nvj, GS_S_READS_1KMER, 1);
                        if (P.classify) {  // mergeReadTaxidPath (:568-586)
                            const int ntin = st.tin[nvj], ntout = st.tout[nvj];
                            bool related = false;  // the first path (in path order) that is an ancestor or a descendant
#pragma unroll
                            for (int h = 0; h < NP; h++) {
                                if (!related) {
                                    const bool mine = 64 * h + lane < used;
                                    const bool a = mine && gs_anc_or_self(ptin[h], ptout[h], ntin);  // path anc-or-self of node
                                    const bool b = mine && gs_anc_or_self(ntin, ntout, ptin[h]);     // node anc-or-self of path
                                    const u64 m = __ballot(a || b);
                                    if (m) {
                                        const int i = __builtin_ctzll(m);
                                        if (lane == i && a) {
                                            path[h] = nvj;
                                            ptin[h] = ntin;
                                            ptout[h] = ntout;
                                        }
                                        related = true;
                                    }
                                }
                            }
                            if (!related && used < P.max_paths) {
#pragma unroll
                                for (int h = 0; h < NP; h++) {
                                    if (64 * h + lane == used) {
                                        path[h] = nvj;
                                        ptin[h] = ntin;
                                        ptout[h] = ntout;
                                    }
                                }
                                used++;
                            }
                        }
                    }
                }
            }
            if (!LONG) GS_STAMP(6, node[0])
            {   // carry: node of the last valid position of this iteration
                const int last_p = (max - 1 < base + 127) ? max - 1 : base + 127;
                const int ls = (last_p - base) >> 6, ll = (last_p - base) & 63;
                carry_last = gs_readlane(ls ? node[1] : node[0], ll);
            }
        }

        if ((GS_ABLATE & 1) == 0 && found) {
            out_flags = GS_F_FOUND | GS_F_RETURNED;
            // tail flush (:455-473): the last contig if it is a hit contig
            if (carry_last >= 0) {
                const int len = max - cur_start;
                if (REC && deferred) {
                    def_contigs++;
                    def_sq += len * len;
                    def_max = len > def_max ? len : def_max;
                } else if (lane == 0)
                    st.contig(carry_last, len, key_lo);
            }
            // ---- 4c. classification (:474-531)
            if (P.classify) {
                const int tax_err = n_miss + bad_lo + (bad_hi ? 1 : 0);
                const double m = P.max_read_tax_err;
                const bool disabled = m >= 0 && ((m >= 1 && (double)tax_err > m) || ((double)tax_err > m * (double)max));
                if (!disabled) {
                    int cn = -1, first_node = -1, best = 0;
                    // one distinct node v with c positions: the only candidate path is v, its sum is c, and the threshold
                    // mapping (:488-492) keeps v if c >= threshold and finds nothing above it otherwise
                    const bool single = !LONG && nd == 1;
                    if (single) {
                        best = one_cnt;
                        cn = (P.threshold <= 1 || one_cnt >= P.threshold) ? one_vi : -1;
                    } else {
                    // sumCounts per candidate path (SmallTaxTree.java:184-193): lanes = paths
                    int sum[NP];
#pragma unroll
                    for (int h = 0; h < NP; h++) sum[h] = 0;
                    if (LONG) {
#pragma unroll
                        for (int h = 0; h < NP; h++)
                            if (64 * h + lane < used)
                                for (int x = path[h]; x >= 0; x = st.parent[x])
                                    if (gs_sc_load(tag + x) == serial) sum[h] += gs_sc_load(cnt + x);
                    } else {
                        for (int dd = 0; dd < nd; dd++) {
                            const int v = dd < 64 ? gs_readlane(dviA, dd) : gs_readlane(dviB, dd - 64);
                            const int c = dd < 64 ? gs_readlane(dcntA, dd) : gs_readlane(dcntB, dd - 64);
#pragma unroll
                            for (int h = 0; h < NP; h++)
                                if (64 * h + lane < used && gs_anc_or_self(st.tin[v], st.tout[v], ptin[h])) sum[h] += c;
                        }
                    }
                    // max + ties exactly as the in-place scan (:476-487); tie order = path order
                    u64 tie_mask[NP];
#pragma unroll
                    for (int h = 0; h < NP; h++) tie_mask[h] = 0;
                    for (int i = 0; i < used; i++) {
                        const int si = (!WIDE || i < 64) ? gs_readlane(sum[0], i) : gs_readlane(sum[NP - 1], i - 64);
                        if (si > best) {
                            best = si;
#pragma unroll
                            for (int h = 0; h < NP; h++) tie_mask[h] = 0;
                        }
                        if (si >= best) {
                            if (!WIDE || i < 64)
                                tie_mask[0] |= 1ULL << i;
                            else
                                tie_mask[NP - 1] |= 1ULL << (i - 64);
                        }
                    }
                    int cand[NP];  // per tied lane: the node entering the LCA fold
#pragma unroll
                    for (int h = 0; h < NP; h++) cand[h] = path[h];
                    if (P.threshold > 1) {
                        // lowestNodeWhereSumAboveThreshold per tied path (SmallTaxTree.java:208-221)
                        if (!LONG) {
                            s_dvi[wave_in_block][lane] = dviA;
                            s_dcnt[wave_in_block][lane] = dcntA;
                            s_dvi[wave_in_block][64 + lane] = dviB;
                            s_dcnt[wave_in_block][64 + lane] = dcntB;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        }
#pragma unroll
                        for (int h = 0; h < NP; h++) {
                            int mapped = -1;
                            if ((tie_mask[h] >> lane) & 1ULL) {
                                int acc = 0;
                                for (int x = path[h]; x >= 0 && mapped < 0; x = st.parent[x]) {
                                    if (LONG) {
                                        if (gs_sc_load(tag + x) == serial) {
                                            acc += gs_sc_load(cnt + x);
                                            if (acc >= P.threshold) mapped = x;
                                        }
                                    } else {
                                        for (int dd = 0; dd < nd; dd++)
                                            if (s_dvi[wave_in_block][dd] == x) {
                                                acc += s_dcnt[wave_in_block][dd];
                                                if (acc >= P.threshold) mapped = x;
                                            }
                                    }
                                }
                            }
                            cand[h] = mapped;
                        }
                    }
                    {
                        bool first = true;
#pragma unroll
                        for (int h = 0; h < NP; h++) {
                            u64 tm = tie_mask[h];
                            while (tm) {
                                const int i = __builtin_ctzll(tm);
                                tm &= tm - 1;
                                const int x = gs_readlane(cand[h], i);
                                if (first) {
                                    cn = x;
                                    first_node = x;
                                    first = false;
                                } else
                                    cn = gs_lca(st, cn, x);
                            }
                        }
                    }
                    }  // !single
                    out_class = cn;
                    if (cn < 0) {
                        out_flags &= ~GS_F_RETURNED;  // "return false" (:497-500)
                    } else {
                        int read_kmers = best;
                        if (P.threshold > 1 && !single) {  // sumCounts(readTaxIdNode[0]) after the promotion (:506-507)
                            read_kmers = 0;
                            if (LONG) {
                                for (int x = first_node; x >= 0; x = st.parent[x])
                                    if (gs_sc_load(tag + x) == serial) read_kmers += gs_sc_load(cnt + x);
                            } else {
                                const int ft = st.tin[first_node];
                                for (int dd = 0; dd < nd; dd++) {
                                    const int v = dd < 64 ? gs_readlane(dviA, dd) : gs_readlane(dviB, dd - 64);
                                    const int c = dd < 64 ? gs_readlane(dcntA, dd) : gs_readlane(dcntB, dd - 64);
                                    if (gs_anc_or_self(st.tin[v], st.tout[v], ft)) read_kmers += c;
                                }
                            }
                        }
                        const int class_err = max - read_kmers;
                        const double mc = P.max_read_class_err;
                        if (mc < 0 || (mc >= 1 && (double)class_err <= mc) || ((double)class_err <= mc * (double)max)) {
                            out_flags |= GS_F_COUNTED;
                            if (REC && deferred) {  // (cn == one_vi: the only candidate)
                                def_counted = 1;
                                def_read_kmers = read_kmers;
                                def_tax_err = tax_err;
                            } else if (lane == 0) {
                                const double err = (double)tax_err / (double)max;
                                const double cerr = (double)class_err / (double)max;
                                st.add(cn, GS_S_READS, 1);
                                st.add(cn, GS_S_READS_KMERS, (u64)read_kmers);
                                st.add(cn, GS_S_READS_BPS, (u64)L);
                                st.dadd(cn, GS_D_ERR_SUM, err);
                                st.dadd(cn, GS_D_ERR_SQ_SUM, err * err);
                                st.dadd(cn, GS_D_CLASS_ERR_SUM, cerr);
                                st.dadd(cn, GS_D_CLASS_ERR_SQ_SUM, cerr * cerr);
                            }
                        }
                    }
                }
            }
        }
        if (REC && deferred && lane == 0) {
            u64 rbase = cur[0], used = cur[1];
            if (used == 64) {  // a fresh chunk of 64 records for this wave
                rbase = atomicAdd(P.stat_rec_count, 64ULL);
                used = 0;
                cur[0] = rbase;
            }
            cur[1] = used + 1;
            GsStatRec rec;
            rec.vi = one_vi;
            rec.contigs = def_contigs;
            rec.kmers = one_cnt;
            rec.sq = def_sq;
            rec.max_key = ((u64)def_max << 40) | key_lo;
            rec.counted = def_counted;
            rec.read_kmers = def_read_kmers;
            rec.read_len = L;
            rec.pad = 0;
            rec.err = (double)def_tax_err / (double)max;
            rec.cerr = (double)(max - def_read_kmers) / (double)max;
            P.stat_recs[rbase + used] = rec;
        }
    }
    if (lane == 0) {
        if (P.class_vi) P.class_vi[r] = out_class;
        if (P.flags) P.flags[r] = (uint8_t)out_flags;
    }
    if (!LONG) GS_STAMP(7, out_class)
}

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
// dynamic LDS: [nv*GS_N_SUMS u64 sums][nv u64 max keys][nv*GS_N_DCOLS doubles][3*nv int32 tree] -- sized to the store's value count
// so that small taxonomies do not cap the occupancy
#define GS_STATS_PROLOGUE()                                                                           \
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_dyn_lds[];                        \
    const int nv = P.db.n_values;                                                                     \
    u64 *s_sums = reinterpret_cast<u64 *>(gs_dyn_lds);                                                \
    u64 *s_max = s_sums + (LDS_STATS ? nv * GS_N_SUMS : 0);                                           \
    double *s_d = reinterpret_cast<double *>(s_max + (LDS_STATS ? nv : 0));                           \
    int32_t *s_tree = reinterpret_cast<int32_t *>(s_d + (LDS_STATS ? nv * GS_N_DCOLS : 0));           \
    if (LDS_STATS) {                                                                                  \
        for (int i = threadIdx.x; i < nv * GS_N_SUMS; i += blockDim.x) s_sums[i] = 0;                 \
        for (int i = threadIdx.x; i < nv; i += blockDim.x) s_max[i] = 0;                              \
        for (int i = threadIdx.x; i < nv * GS_N_DCOLS; i += blockDim.x) s_d[i] = 0.0;                 \
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {                                          \
            s_tree[i] = P.db.parent[i];                                                               \
            s_tree[nv + i] = P.db.tin[i];                                                             \
            s_tree[2 * nv + i] = P.db.tout[i];                                                        \
        }                                                                                             \
        __syncthreads();                                                                              \
    }                                                                                                 \
    const bool tree_lds = !LDS_STATS && nv <= GS_NV_TREE_LDS; /* counters too big for LDS, the tree is not */ \
    if (tree_lds) {                                                                                   \
        for (int i = threadIdx.x; i < nv; i += blockDim.x) {                                          \
            s_tree[i] = P.db.parent[i];                                                               \
            s_tree[nv + i] = P.db.tin[i];                                                             \
            s_tree[2 * nv + i] = P.db.tout[i];                                                        \
        }                                                                                             \
        __syncthreads();                                                                              \
    }                                                                                                 \
    GsStats st;                                                                                       \
    if (LDS_STATS) {                                                                                  \
        st.sums = s_sums;                                                                             \
        st.maxk = s_max;                                                                              \
        st.dsums = s_d;                                                                               \
        st.parent = s_tree;                                                                           \
        st.tin = s_tree + nv;                                                                         \
        st.tout = s_tree + 2 * nv;                                                                    \
    } else {                                                                                          \
        const size_t copy = P.stat_copies > 1 ? (size_t)(blockIdx.x % (unsigned)P.stat_copies) : 0;   \
        st.sums = (u64 *)P.sums + copy * (size_t)nv * GS_N_SUMS;                                      \
        st.maxk = (u64 *)P.max_keys + copy * (size_t)nv;                                              \
        st.dsums = P.dsums + copy * (size_t)nv * GS_N_DCOLS;                                          \
        st.parent = tree_lds ? s_tree : P.db.parent;                                                  \
        st.tin = tree_lds ? s_tree + nv : P.db.tin;                                                   \
        st.tout = tree_lds ? s_tree + 2 * nv : P.db.tout;                                             \
    }

#define GS_STATS_EPILOGUE()                                                                           \
    if (LDS_STATS) {                                                                                  \
        __syncthreads();                                                                              \
        const size_t ecopy = P.stat_copies > 1 ? (size_t)(blockIdx.x % (unsigned)P.stat_copies) : 0;  \
        for (int i = threadIdx.x; i < nv * GS_N_SUMS; i += blockDim.x)                                \
            if (s_sums[i]) atomicAdd((u64 *)P.sums + ecopy * (size_t)nv * GS_N_SUMS + i, s_sums[i]);  \
        for (int i = threadIdx.x; i < nv; i += blockDim.x)                                            \
            if (s_max[i]) atomicMax((u64 *)P.max_keys + ecopy * (size_t)nv + i, s_max[i]);            \
        for (int i = threadIdx.x; i < nv * GS_N_DCOLS; i += blockDim.x)                               \
            if (s_d[i] != 0.0) atomicAdd(P.dsums + ecopy * (size_t)nv * GS_N_DCOLS + i, s_d[i]);      \
    }

// STRIPED: the record table is split over several devices (GsDbDev::rec_biased); every wave keeps the stripe pointers in
// LDS behind its hash rows, where a lane picks the one of its bucket with a single ds_read
#define GS_STRIPE_TABLE(kernarg)                                                                                        \
    if (STRIPED) {                                                                                                       \
        if (lane < 2 * GS_MAX_STRIPES)                                                                                   \
            reinterpret_cast<const u64 **>(s_g[wave_in_block] + 2 * GS_ROW)[lane] =                                      \
                lane < GS_MAX_STRIPES ? (kernarg)->db.rec_biased[lane] : (kernarg)->db.tab_biased[lane - GS_MAX_STRIPES]; \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                           \
        __builtin_amdgcn_wave_barrier();                                                                                 \
    }

// ---------------------------------------------------------------------------------------------------
// Which kernel takes which read of a batch with an offsets array: up to 128 k-mer positions gs_match_kernel (which skips the others);
// 129 .. 192 / 193 .. 256 positions gs_match_wide_kernel<3 / 4> where wide_mask says it serves this run (queues 1 / 2); more
// gs_match_long_kernel (queue 0); huge_min positions and more the kernels that cut a read into chunks over many waves (the first
// huge_slots of a batch).  A queue is a dense list of read numbers.  One thread per read and round, GS_CLS_READS reads per workgroup:
// first the workgroup counts its reads per class (in LDS), ONE atomic per class reserves their places, then every read is written to
// its place.  (Queued by gs_match_kernel itself -- one atomic per 64 reads and wave, all on one address -- 9.4 M reads of 159 bp cost
// 1.9 ms: atomics on one address execute one after the other, about 11 ns each.)
// ---------------------------------------------------------------------------------------------------
#define GS_CLS_THREADS 1024
#define GS_CLS_ROUNDS 8
#define GS_CLS_READS (GS_CLS_THREADS * GS_CLS_ROUNDS)
__global__ __launch_bounds__(GS_CLS_THREADS) void gs_classify_kernel(GsMatchParams P, int from_nodes) {
    __shared__ unsigned int s_n[4], s_base[4];
    if (P.skip != nullptr && *P.skip != 0) return;
    if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
    __syncthreads();
    const int lane = gs_lane();
    const int k = P.db.k;
    const int64_t b0 = (int64_t)blockIdx.x * GS_CLS_READS;
    unsigned int place[GS_CLS_ROUNDS];  // class << 30 | place among the workgroup's reads of that class
#pragma unroll
    for (int j = 0; j < GS_CLS_ROUNDS; j++) {
        const int64_t r = b0 + (int64_t)j * GS_CLS_THREADS + threadIdx.x;
        int cls = 0;  // 0: gs_match_kernel; 1 / 2: wide queues; 3: long-read queue (queue 0); huge reads take a slot or class 3
        if (r < P.n_reads) {
            int L;
            if (P.off_stride == 0)
                L = P.fixed_len;
            else {
                const uint64_t *po = P.off + r * P.off_stride;
                L = (int)(po[1] - po[0]);
            }
            const int pos = L - k + 1;
            if (pos > 128) {
                cls = (pos <= 192 && (P.wide_mask & 1)) ? 1 : ((pos <= 256 && (P.wide_mask & 2)) ? 2 : 3);
                if (!from_nodes && P.huge_count != nullptr && pos >= P.huge_min) {
                    const unsigned int slot = atomicAdd(P.huge_count, 1u);  // (rare: contigs, chromosomes)
                    if (slot < (unsigned int)P.huge_slots) {
                        P.huge_list[slot] = (uint32_t)r;
                        cls = 0;
                    } else
                        cls = 3;
                }
            }
        }
        unsigned int pl = 0;
#pragma unroll
        for (int c = 1; c <= 3; c++) {
            const u64 m = __ballot(cls == c);
            if (m != 0) {  // (wave-uniform)
                unsigned int wb = 0;
                if (lane == 0) wb = atomicAdd(&s_n[c], (unsigned int)__popcll(m));
                wb = (unsigned int)gs_rfl((int)wb);
                if (cls == c) pl = ((unsigned int)c << 30) | (wb + (unsigned int)__popcll(m & ((1ULL << lane) - 1ULL)));
            }
        }
        place[j] = pl;
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= 3 && s_n[threadIdx.x] != 0) s_base[threadIdx.x] = atomicAdd(P.long_count + 2 * (threadIdx.x % 3), s_n[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GS_CLS_ROUNDS; j++) {
        const unsigned int c = place[j] >> 30;
        if (c != 0) P.long_list[(size_t)(c % 3) * (size_t)P.long_cap + s_base[c] + (place[j] & 0x3fffffffu)] = (uint32_t)(b0 + (int64_t)j * GS_CLS_THREADS + threadIdx.x);
    }
}
extern "C" hipError_t gs_launch_classify(const GsMatchParams *P, hipStream_t stream) {
    if (P->n_reads <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_classify_kernel, dim3((unsigned)((P->n_reads + GS_CLS_READS - 1) / GS_CLS_READS)), dim3(GS_CLS_THREADS), 0, stream, *P, P->nodes != nullptr ? 1 : 0);
    return hipGetLastError();
}

template <bool LDS_STATS, bool FROM_NODES, int KC, bool WIDE = false, bool STRIPED = false, int CTX = 0>
__global__ __launch_bounds__(GS_BLOCK) __attribute__((amdgpu_waves_per_eu(GS_WAVES, GS_WAVES))) void gs_match_kernel(GsMatchParams P) {
    GS_STATS_PROLOGUE()
    __shared__ int s_dvi[GS_BLOCK / 64][128];  // distinct-node list copy, threshold > 1 only
    __shared__ int s_dcnt[GS_BLOCK / 64][128];
    // 15-mer order hashes of the wave's current 144 positions (+ the stripe pointers)
    __shared__ __attribute__((aligned(8))) uint32_t s_g[GS_BLOCK / 64][2 * GS_ROW + (STRIPED ? GS_STRIPE_WORDS : 0)];
    const int lane = gs_lane();
    const int wave_in_block = gs_rfl((int)(threadIdx.x >> 6));  // wave-uniform: per-read bookkeeping runs on the scalar unit
    const int64_t wave_id = (int64_t)blockIdx.x * (GS_BLOCK / 64) + wave_in_block;
    const int64_t n_waves = (int64_t)gs_rfl((int)gridDim.x) * (GS_BLOCK / 64);
    const int k = KC ? KC : P.db.k;
    // The ~40 launch parameters do not fit the scalar register file next to the ballot planes; kept live across
    // the loop they are spilled to VGPR lanes and read back with v_readlane on every use.  Re-reading them from
    // the kernarg segment (scalar cache) once per read is cheaper: the empty asm hides from the compiler that the
    // pointer is loop invariant, so the s_loads stay inside the loop.
    typedef const __attribute__((address_space(4))) GsMatchParams *GsKernargPtr;
    const GsKernargPtr kp0 = (GsKernargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    // text mode: a chunk that the device-side record scan refused is skipped as a whole (gs_text.hip)
    const int64_t n_reads = (P.skip != nullptr && *P.skip != 0) ? 0 : P.n_reads;
    __shared__ GsRecCursor s_cur[GS_BLOCK / 64];
    if (lane == 0) {
        s_cur[wave_in_block][0] = 0;
        s_cur[wave_in_block][1] = 64;
    }
    GS_STRIPE_TABLE((const GsMatchParams *)kp0)
#if GS_PHASE
    if (lane < 16) gs_phase_row()[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    gs_stamp(-1);
#endif
    // (Software pipelines over the wave's reads were measured again in round 3, after the LDS and gate round trips had been batched:
    // (1) the bases of the next read into a second LDS buffer by LDS-DMA and the offsets two reads ahead, requested at the top of
    // the iteration: 7.59 -> 7.77 ms on configs[1] -- requests OLDER than the gate loads are waited for with them (in-order
    // counter), the latency only moves; (2) the same requests issued behind the gate loads, the youngest at every later wait: the
    // waits for offsets and bases shrink from 3 250 to 1 650 wave cycles per read (tools/phase_times.sh) and every other phase
    // grows by as much: 7.62 ms, 47 M-k-mer store 9.22 -> 9.41 ms.  The SIMD is short of issue slots, not of overlap.)
    const int64_t po_step = n_waves * P.off_stride;        // stride 1: running offsets; 2: (start, end) pairs
    const uint64_t *po = P.off + wave_id * P.off_stride;   // offsets of the current read
    for (int64_t r = wave_id; r < n_reads; r += n_waves, po += po_step) {
        GsKernargPtr kp = kp0;
#ifndef GS_KERNARG_HOISTED  // (experiment, DESIGN 8.1 "kernarg re-reads": let the compiler keep the parameters live across the loop)
        asm volatile("" : "+s"(kp));
#endif
        const GsMatchParams &Q = *(const GsMatchParams *)kp;
        u64 off;
        int L;
        if (Q.off_stride == 0) {  // reads of one length, back to back: nothing to load
            L = Q.fixed_len;
            off = (u64)r * (u64)(uint32_t)L;
        } else {
            off = po[0];
            L = (int)(po[1] - off);
        }
        GS_STAMP(0, L)
        if (L - k + 1 > 128) continue;  // another kernel's: gs_classify_kernel has put it into that kernel's queue
        uint32_t pre[3];
        const uint8_t *rd = Q.seq + off;
#pragma unroll
        for (int w = 0; w < 3; w++) pre[w] = 64 * w + lane < L ? rd[64 * w + lane] : GS_FILL;
        gs_process_read<false, FROM_NODES, KC, WIDE, !LDS_STATS, STRIPED, CTX>(Q, st, r, off, L, lane, s_dvi, s_dcnt, wave_in_block, nullptr, nullptr, 0, pre,
                                                     s_g[wave_in_block], s_cur[wave_in_block]);
    }
#if GS_PHASE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 15) atomicAdd(&gs_phase_acc[lane], gs_phase_row()[lane]);
#endif
    if (!LDS_STATS && P.stat_recs != nullptr) {  // the rest of the wave's last chunk
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const u64 base = s_cur[wave_in_block][0], used = s_cur[wave_in_block][1];
        if (used < 64 && (u64)lane >= used) P.stat_recs[base + (u64)lane].vi = -1;
    }
    GS_STATS_EPILOGUE()
}

// ---------------------------------------------------------------------------------------------------
// Reads of 129 .. 64 NS k-mer positions in ONE trip (NS = 3: up to 192 positions, 159 .. 222 bp at k = 31, 150 bp at k < 23).  The
// long-read path takes such a read in two iterations of 128 positions, each a chain of dependent round trips (bases, gate word,
// record lines) that costs what a whole short read does however few of its positions are live: 158 bp 233 Gbp/s, 159 bp 129.  Here
// the NS sub-rounds of the read go through the probe together -- their gate words and record lines are in flight side by side --
// and the read is reduced like a short read (one pass, distinct tax ids in registers); the price is registers (planes, k-mers and
// nodes of NS sub-rounds), i.e. fewer waves per SIMD.  Same results as gs_process_read by construction: the same probe, the same
// closed forms, and every rule of matchRead (C/match/FastqKMerMatcher.java:330-531) restated for NS words of positions.
// ---------------------------------------------------------------------------------------------------
template <int NS, int KC, int CTX>
__device__ __forceinline__ void gs_process_read_wide(const GsMatchParams &P, const GsStats &st, int64_t r, u64 off, int L, int lane, uint32_t *wave_g) {
    // (the list of distinct tax ids goes through the wave's rows when the threshold rule needs it in memory: the probe is through with them)
    int *s_dvi = reinterpret_cast<int *>(wave_g), *s_dcnt = reinterpret_cast<int *>(wave_g) + 64 * NS;
    static_assert(GS_WIDE_WORDS(NS) >= 2 * 64 * NS, "room for the list");
    const GsDbDev &db = P.db;
    const int k = KC ? KC : db.k;
    const int max = L - k + 1;
    const uint8_t *rd = P.seq + off;
    int out_class = -1, out_flags = 0;
    if (max > 0) {
        const u64 key_lo = ((1ULL << 40) - 1) - ((u64)(P.first_read_no + r) & ((1ULL << 40) - 1));
        u64 Bhi[NS + 1], Blo[NS + 1], Bbad[NS + 1];
#pragma unroll
        for (int w = 0; w <= NS; w++) gs_load_word(rd, L, w, lane, Bhi[w], Blo[w], Bbad[w]);
        int bad_lo = 0;
        bool bad_hi = false;
        {   // bad-base census for the closed form of the INVALID steps (as gs_process_read; every word belongs to this one trip)
            const int q = max - 1;
#pragma unroll
            for (int w = 0; w <= NS; w++) {
                const int lo_bits = q - 64 * w;
                const u64 m_lo = lo_bits >= 64 ? ~0ULL : (lo_bits <= 0 ? 0ULL : ((1ULL << lo_bits) - 1));
                bad_lo += __popcll(Bbad[w] & m_lo);
                bad_hi = bad_hi || ((Bbad[w] & ~m_lo) != 0);
            }
        }
        int node[NS];
        const GsMark mk = {P.count_unique, P.hit_counts, nullptr, nullptr};
        gs_probe_planes<KC, false, CTX, false, NS>(db, Bhi, Blo, Bbad, 0, max, lane, node, wave_g, mk);
        u64 hit[NS], any_hit = 0;
        int n_miss = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            hit[s] = __ballot(node[s] >= 0);
            any_hit |= hit[s];
            n_miss += __popcll(__ballot(node[s] == GS_NODE_MISS));
        }
        if (any_hit != 0) {
            out_flags = GS_F_FOUND | GS_F_RETURNED;
            // ---- contigs (:390-413, :455-473): a lane that starts a run of a hit node books it; the run ends at the next change of
            // node, in its own word or a later one, or with the read
            u64 chg[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const int up = __shfl_up(node[s], 1);
                int before = GS_NODE_NONE;  // (in front of position 0: nothing -- the first position starts a run)
                if (s > 0) before = gs_readlane(node[s - 1], 63);
                const int prev = lane == 0 ? before : up;
                const int nvs = max - 64 * s;
                const u64 vm = nvs >= 64 ? ~0ULL : (nvs <= 0 ? 0ULL : ((1ULL << nvs) - 1ULL));
                chg[s] = __ballot(node[s] != prev) & vm;
            }
            int later = max;  // first change in the words behind the current one (or the read's end)
#pragma unroll
            for (int s = NS - 1; s >= 0; s--) {
                const u64 above = (chg[s] >> 1) >> lane;
                const int end = above ? 64 * s + lane + 1 + __builtin_ctzll(above) : later;
                if (GS_ACT(hit[s] & chg[s])) st.contig(node[s], end - (64 * s + lane), key_lo);
                if (chg[s]) later = 64 * s + __builtin_ctzll(chg[s]);
            }
            // ---- the distinct hit nodes in order of first appearance: reads1KMer (:434-439), mergeReadTaxidPath (:568-586); entry i
            // of the list (node, votes) lives in lane i & 63 of register set i >> 6
            int dvi[NS], dcnt[NS], nd = 0;
            int path = -1, ptin = 0, ptout = 0, used = 0;
#pragma unroll
            for (int h = 0; h < NS; h++) {
                dvi[h] = -1;
                dcnt[h] = 0;
            }
            u64 m[NS];
#pragma unroll
            for (int s = 0; s < NS; s++) m[s] = hit[s];
            for (;;) {
                int j = -1;
#pragma unroll
                for (int s = NS - 1; s >= 0; s--)
                    if (m[s]) j = 64 * s + __builtin_ctzll(m[s]);
                if (j < 0) break;
                int nvj = 0;
#pragma unroll
                for (int s = 0; s < NS; s++)
                    if ((j >> 6) == s) nvj = gs_readlane(node[s], j & 63);
                int c = 0;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    const u64 e = __ballot(node[s] == nvj);
                    c += __popcll(e);
                    m[s] &= ~e;
                }
#pragma unroll
                for (int h = 0; h < NS; h++)
                    if ((nd >> 6) == h && lane == (nd & 63)) {
                        dvi[h] = nvj;
                        dcnt[h] = c;
                    }
                nd++;
                if (lane == 0) st.add(nvj, GS_S_READS_1KMER, 1);
                if (P.classify) {
                    const int ntin = st.tin[nvj], ntout = st.tout[nvj];
                    const bool mine = lane < used;
                    const bool a = mine && gs_anc_or_self(ptin, ptout, ntin);  // path anc-or-self of node
                    const bool b = mine && gs_anc_or_self(ntin, ntout, ptin);  // node anc-or-self of path
                    const u64 rel = __ballot(a || b);
                    if (rel) {
                        if (lane == __builtin_ctzll(rel) && a) {
                            path = nvj;
                            ptin = ntin;
                            ptout = ntout;
                        }
                    } else if (used < P.max_paths) {
                        if (lane == used) {
                            path = nvj;
                            ptin = ntin;
                            ptout = ntout;
                        }
                        used++;
                    }
                }
            }
            // ---- classification (:474-531)
            if (P.classify) {
                const int tax_err = n_miss + bad_lo + (bad_hi ? 1 : 0);
                const double mt = P.max_read_tax_err;
                const bool disabled = mt >= 0 && ((mt >= 1 && (double)tax_err > mt) || ((double)tax_err > mt * (double)max));
                if (!disabled) {
                    int cn = -1, first_node = -1, best = 0;
                    // sumCounts per candidate path (SmallTaxTree.java:184-193): lanes = paths
                    int sum = 0;
                    for (int dd = 0; dd < nd; dd++) {
                        int v = 0, c = 0;
#pragma unroll
                        for (int h = 0; h < NS; h++)
                            if ((dd >> 6) == h) {
                                v = gs_readlane(dvi[h], dd & 63);
                                c = gs_readlane(dcnt[h], dd & 63);
                            }
                        if (lane < used && gs_anc_or_self(st.tin[v], st.tout[v], ptin)) sum += c;
                    }
                    u64 tie_mask = 0;  // max + ties exactly as the in-place scan (:476-487); tie order = path order
                    for (int i = 0; i < used; i++) {
                        const int si = gs_readlane(sum, i);
                        if (si > best) {
                            best = si;
                            tie_mask = 0;
                        }
                        if (si >= best) tie_mask |= 1ULL << i;
                    }
                    int cand = path;
                    if (P.threshold > 1) {  // lowestNodeWhereSumAboveThreshold per tied path (SmallTaxTree.java:208-221)
#pragma unroll
                        for (int h = 0; h < NS; h++) {
                            s_dvi[64 * h + lane] = dvi[h];
                            s_dcnt[64 * h + lane] = dcnt[h];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        int mapped = -1;
                        if ((tie_mask >> lane) & 1ULL) {
                            int acc = 0;
                            for (int x = path; x >= 0 && mapped < 0; x = st.parent[x])
                                for (int dd = 0; dd < nd; dd++)
                                    if (s_dvi[dd] == x) {
                                        acc += s_dcnt[dd];
                                        if (acc >= P.threshold) mapped = x;
                                    }
                        }
                        cand = mapped;
                        __builtin_amdgcn_wave_barrier();
                    }
                    {
                        bool first = true;
                        for (u64 tm = tie_mask; tm; tm &= tm - 1) {
                            const int x = gs_readlane(cand, __builtin_ctzll(tm));
                            if (first) {
                                cn = x;
                                first_node = x;
                                first = false;
                            } else
                                cn = gs_lca(st, cn, x);
                        }
                    }
                    out_class = cn;
                    if (cn < 0) {
                        out_flags &= ~GS_F_RETURNED;  // "return false" (:497-500)
                    } else {
                        int read_kmers = best;
                        if (P.threshold > 1) {  // sumCounts(readTaxIdNode[0]) after the promotion (:506-507)
                            read_kmers = 0;
                            const int ft = st.tin[first_node];
                            for (int dd = 0; dd < nd; dd++) {
                                int v = 0, c = 0;
#pragma unroll
                                for (int h = 0; h < NS; h++)
                                    if ((dd >> 6) == h) {
                                        v = gs_readlane(dvi[h], dd & 63);
                                        c = gs_readlane(dcnt[h], dd & 63);
                                    }
                                if (gs_anc_or_self(st.tin[v], st.tout[v], ft)) read_kmers += c;
                            }
                        }
                        const int class_err = max - read_kmers;
                        const double mc = P.max_read_class_err;
                        if (mc < 0 || (mc >= 1 && (double)class_err <= mc) || ((double)class_err <= mc * (double)max)) {
                            out_flags |= GS_F_COUNTED;
                            if (lane == 0) {
                                const double err = (double)tax_err / (double)max;
                                const double cerr = (double)class_err / (double)max;
                                st.add(cn, GS_S_READS, 1);
                                st.add(cn, GS_S_READS_KMERS, (u64)read_kmers);
                                st.add(cn, GS_S_READS_BPS, (u64)L);
                                st.dadd(cn, GS_D_ERR_SUM, err);
                                st.dadd(cn, GS_D_ERR_SQ_SUM, err * err);
                                st.dadd(cn, GS_D_CLASS_ERR_SUM, cerr);
                                st.dadd(cn, GS_D_CLASS_ERR_SQ_SUM, cerr * cerr);
                            }
                        }
                    }
                }
            }
        }
    }
    if (lane == 0) {
        if (P.class_vi) P.class_vi[r] = out_class;
        if (P.flags) P.flags[r] = (uint8_t)out_flags;
    }
}

// waves per SIMD, three sub-rounds: 8 (64 VGPRs, a few spilled) beats 7 / 6 (80 VGPRs, none spilled) / 5 / 4 on reads of 159, 190
// and 222 bp -- 168 / 168 / 161 / 149 / 105 Gbp/s at 159 bp; four sub-rounds: 6 beats 8 / 7 / 5 -- 196 / 169 / 189 / 179 Gbp/s at
// 250 bp (the long-read path: 180)
#ifndef GS_WIDE_WAVES
#define GS_WIDE_WAVES 8
#endif
#ifndef GS_WIDE4_WAVES
#define GS_WIDE4_WAVES 6
#endif
#ifndef GS_WIDE_ANYK_WAVES  // (k as a run-time value; a store without records keeps half a bucket line per sub-round in registers: k = 16 table-only 112.6 Gbp/s at 7, 91.8 at 8, 112.0 at 6; a k = 25 record store: 7 = 8 > 6)
#define GS_WIDE_ANYK_WAVES 7
#endif
#define GS_WIDE_WAVES_OF(NS, KC) ((NS) == 4 ? GS_WIDE4_WAVES : ((KC) ? GS_WIDE_WAVES : GS_WIDE_ANYK_WAVES))
// The reads of queue NS - 2 (filled by gs_match_kernel: 129 .. 192 positions in queue 1, 193 .. 256 in queue 2, GsMatchParams::
// wide_mask), or -- long_list == nullptr: reads of one length, gs_match_submit_fixed, the launcher has looked -- every read of the batch.
template <bool LDS_STATS, int NS, int KC>
__global__ __launch_bounds__(GS_BLOCK) __attribute__((amdgpu_waves_per_eu(GS_WIDE_WAVES_OF(NS, KC), GS_WIDE_WAVES_OF(NS, KC)))) void gs_match_wide_kernel(GsMatchParams P) {
    const bool all = P.long_list == nullptr;
    unsigned int *qc = P.long_count + 2 * (NS - 2);  // [0] entries, [1] the cursor the waves draw chunks of GS_LONG_CHUNK entries from
    if (!all && qc[0] == 0) return;
    GS_STATS_PROLOGUE()
    __shared__ __attribute__((aligned(8))) uint32_t s_g[GS_BLOCK / 64][GS_WIDE_WORDS(NS)];
    const int lane = gs_lane();
    const int wave_in_block = gs_rfl((int)(threadIdx.x >> 6));
    const u64 n_q = all ? (u64)P.n_reads : (u64)qc[0];
    const uint32_t *list = all ? nullptr : P.long_list + (size_t)(NS - 2) * (size_t)P.long_cap;
    for (;;) {
        __builtin_amdgcn_wave_barrier();  // (as gs_match_long_kernel: no lane-0 test threaded through the back edge)
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(qc + 1, 1u);
        c = (uint32_t)gs_rfl((int)c);
        if ((u64)c * GS_LONG_CHUNK >= n_q) break;
        const u64 at = (u64)c * GS_LONG_CHUNK + (u64)lane;
        const uint32_t mine = at < n_q ? (all ? (uint32_t)at : list[at]) : GS_LONG_NONE;
        // the offsets of the chunk's reads with one gather (every lane its own read's), not one dependent load per read
        u64 my_off = 0;
        int my_len = 0;
        if (P.off_stride != 0 && mine != GS_LONG_NONE) {
            const uint64_t *po = P.off + (int64_t)mine * P.off_stride;
            my_off = po[0];
            my_len = (int)(po[1] - my_off);
        }
        for (u64 todo = __ballot(mine != GS_LONG_NONE); todo; todo &= todo - 1) {
            const int i = __builtin_ctzll(todo);
            const int64_t r = (int64_t)(uint32_t)gs_readlane((int)mine, i);
            u64 off;
            int L;
            if (P.off_stride == 0) {
                L = P.fixed_len;
                off = (u64)r * (u64)(uint32_t)L;
            } else {
                off = ((u64)(uint32_t)gs_readlane((int)(my_off >> 32), i) << 32) | (u64)(uint32_t)gs_readlane((int)(uint32_t)my_off, i);
                L = gs_readlane(my_len, i);
            }
            gs_process_read_wide<NS, KC, 2>(P, st, r, off, L, lane, s_g[wave_in_block]);
        }
    }
    GS_STATS_EPILOGUE()
}

// waves per SIMD of the long-read kernel: left alone the compiler takes 80-90 VGPRs (5 waves); at 8 waves (64 VGPRs, 8 of them
// spilled) reads of 1000 bp run at 162 instead of 134 Gbp/s, at 6 waves (77 VGPRs, no spills) at 140 (tools/long_read_rate.py)
#ifndef GS_LONG_WAVES
#define GS_LONG_WAVES GS_WAVES
#endif
#if GS_LONG_WAVES
#define GS_LONG_ATTR __attribute__((amdgpu_waves_per_eu(GS_LONG_WAVES, GS_LONG_WAVES)))
#else
#define GS_LONG_ATTR
#endif
template <bool LDS_STATS, bool FROM_NODES, bool WIDE = false, bool STRIPED = false, int KC = 0>
__global__ __launch_bounds__(GS_BLOCK) GS_LONG_ATTR void gs_match_long_kernel(GsMatchParams P, int32_t *scratch, uint32_t *serials) {
    // long_list == nullptr: EVERY read of the batch is a long one (reads of one length, gs_match_submit_fixed: the launcher has looked) --
    // no queue was written, chunk c holds the reads 64 c .. 64 c + 63
    const bool all_long = P.long_list == nullptr;
    if (!all_long && P.long_count[0] == 0) return;  // (a batch of short reads: nothing queued, no counters to set up and flush)
    GS_STATS_PROLOGUE()
    const int lane = gs_lane();
    const int wave_in_block = gs_rfl((int)(threadIdx.x >> 6));  // wave-uniform: per-read bookkeeping runs on the scalar unit
    const int64_t wave_id = (int64_t)blockIdx.x * (GS_BLOCK / 64) + wave_in_block;
    __shared__ __attribute__((aligned(8))) uint32_t s_g[GS_BLOCK / 64][2 * GS_ROW + (STRIPED ? GS_STRIPE_WORDS : 0)];
    GS_STRIPE_TABLE(&P)
    // The queue (written by gs_classify_kernel on the same stream) is a dense list; the waves of this kernel draw chunks of
    // GS_LONG_CHUNK entries from a shared cursor
    // (long_count[1]) until the queue is empty: reads of very different lengths spread over the waves by themselves, and
    // every wave leaves the loop with the first chunk index beyond the end.
    const u64 n_long = all_long ? (u64)P.n_reads : (u64)P.long_count[0];
    // per-wave vote rows (tag = serial of the read that last touched a node, cnt = its votes in that read): in LDS for the
    // taxonomies whose counters are in LDS as well (every distinct node of every iteration reads and writes them: two
    // dependent round trips to HBM otherwise), else this wave's rows of `scratch`, which persist from launch to launch
    __shared__ int32_t s_votes[LDS_STATS ? GS_BLOCK / 64 : 1][LDS_STATS ? 2 * GS_NV_LDS : 1];
    int32_t *tag = LDS_STATS ? s_votes[wave_in_block] : scratch + (size_t)wave_id * 2 * (size_t)nv;
    int32_t *cnt = tag + (LDS_STATS ? GS_NV_LDS : nv);
    uint32_t serial = LDS_STATS ? 0u : serials[wave_id];
    if (LDS_STATS) {
        for (int i = lane; i < GS_NV_LDS; i += 64) tag[i] = 0;  // (serials start at 1)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    for (;;) {
        // (keeps the compiler from threading a lane-0 test at the end of an iteration into this one through the back edge: lane 0
        // and the other lanes would go around the loop separately and read their own `c` -- seen in gs_inflate_dev.hip)
        __builtin_amdgcn_wave_barrier();
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(P.long_count + 1, 1u);
        c = (uint32_t)gs_rfl((int)c);
        if ((u64)c * GS_LONG_CHUNK >= n_long) break;
        const u64 at = (u64)c * GS_LONG_CHUNK + (u64)lane;
        const uint32_t mine = at < n_long ? (all_long ? (uint32_t)at : P.long_list[at]) : GS_LONG_NONE;  // (the queue is a dense list)
        for (u64 todo = __ballot(mine != GS_LONG_NONE); todo; todo &= todo - 1) {
            serial++;
            if (serial == 0) {  // wrap after 2^32 - 1 long reads on this wave: old tags could alias, so the wave's tag row starts over
                for (int i = lane; i < nv; i += 64) gs_sc_store(tag + i, 0);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                serial = 1;
            }
            const int64_t r = (int64_t)(uint32_t)gs_readlane((int)mine, __builtin_ctzll(todo));
            u64 off;
            int L;
            if (P.off_stride == 0) {
                L = P.fixed_len;
                off = (u64)r * (u64)(uint32_t)L;
            } else {
                const uint64_t *po = P.off + r * P.off_stride;
                off = po[0];
                L = (int)(po[1] - off);
            }
            const uint32_t none[3] = {0, 0, 0};
            gs_process_read<true, FROM_NODES, KC, WIDE, false, STRIPED>(P, st, r, off, L, lane, nullptr, nullptr, wave_in_block, tag,
                                                       cnt, (int)serial, none, s_g[wave_in_block], nullptr);
        }
    }
    if (!LDS_STATS && lane == 0) serials[wave_id] = serial;
    GS_STATS_EPILOGUE()
}

// ---------------------------------------------------------------------------------------------------
// Reads of GS_HUGE_MIN positions and more, over many waves (a chromosome on ONE wave runs at 30 Mbp/s).
// A read is cut into chunks of whole iterations (GS_HUGE_CHUNK_MIN positions and more, at most GS_HUGE_MAX_CHUNKS chunks); gs_match_huge_kernel gives every chunk a
// wave, which walks it exactly as the long-read path walks a read -- same probe, same closed form for windows with a bad base, the
// contigs INSIDE the chunk booked by the lanes that start them -- and leaves behind what a single wave would have carried across:
//   * per read and node, atomically: the positions that hold the node (the votes) and the FIRST of them;
//   * per read: misses, bad bases, "some k-mer hit";
//   * per chunk: the run it starts with and the run that is open at its end (neither is booked: a run may span chunks).
// gs_match_huge_finish_kernel, one wave per read, then does what is sequential in matchRead and cheap: closes the runs across the
// seams (FastqKMerMatcher.java:390-413, :455-473), walks the distinct nodes in order of first appearance -- reads1KMer (:434-439),
// mergeReadTaxidPath (:568-586) --, classifies (:474-531) from the vote counts, and clears the read's rows for the next batch.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gs_huge_chunk_positions(int max, int chunk_min) {
    int c = ((max + GS_HUGE_MAX_CHUNKS - 1) / GS_HUGE_MAX_CHUNKS + 127) & ~127;
    return c < chunk_min ? chunk_min : c;
}

__device__ __forceinline__ void gs_huge_read(const GsMatchParams &P, int64_t r, u64 &off, int &L) {
    if (P.off_stride == 0) {
        L = P.fixed_len;
        off = (u64)r * (u64)(uint32_t)L;
    } else {
        const uint64_t *po = P.off + r * P.off_stride;
        off = po[0];
        L = (int)(po[1] - off);
    }
}

// votes of one node from one chunk into the read's rows; the wave that brings the first ones puts the node on the read's list
__device__ __forceinline__ void gs_huge_vote(uint32_t *cnt, uint32_t *first, uint32_t *touch, GsHugeHead *h0, int v, uint32_t c, uint32_t pos) {
    if (atomicAdd(cnt + v, c) == 0) touch[atomicAdd(&h0->n_touch, 1u)] = (uint32_t)v;  // (once per copy: the list may hold a node several times)
    atomicMin(first + v, pos);
}

// (waves per SIMD: a chunk is a chain of dependent iterations, and a spilled register is a round trip to memory inside it)
#ifndef GS_HUGE_WAVES
#define GS_HUGE_WAVES GS_LONG_WAVES
#endif
#if GS_HUGE_WAVES
#define GS_HUGE_ATTR __attribute__((amdgpu_waves_per_eu(GS_HUGE_WAVES, GS_HUGE_WAVES)))
#else
#define GS_HUGE_ATTR
#endif
template <bool LDS_STATS, bool STRIPED, int KC = 0>
__global__ __launch_bounds__(GS_BLOCK) GS_HUGE_ATTR void gs_match_huge_kernel(GsMatchParams P) {
    const unsigned int n_huge_all = P.huge_count[0];
    if (n_huge_all == 0) return;
    const int n_huge = (int)(n_huge_all < (unsigned int)P.huge_slots ? n_huge_all : (unsigned int)P.huge_slots);
    GS_STATS_PROLOGUE()
    const int lane = gs_lane();
    const int wave_in_block = gs_rfl((int)(threadIdx.x >> 6));
    const int64_t wave_id = (int64_t)blockIdx.x * (GS_BLOCK / 64) + wave_in_block;
    const int64_t n_waves = (int64_t)gridDim.x * (GS_BLOCK / 64);
    __shared__ __attribute__((aligned(8))) uint32_t s_g[GS_BLOCK / 64][2 * GS_ROW + (STRIPED ? GS_STRIPE_WORDS : 0)];
    GS_STRIPE_TABLE(&P)
    uint32_t *wave_g = s_g[wave_in_block];
    const GsDbDev &db = P.db;
    const int k = KC ? KC : db.k;
    const GsMark mk = {P.count_unique, P.hit_counts, STRIPED ? P.bitmap : nullptr, STRIPED ? P.bitmap + ((db.bucket_mask + 1) * GS_SLOTS_PER_BUCKET >> 5) : nullptr};
    // the chunks of all reads in one row: s_first[slot] = number of the read's first chunk (GS_HUGE_SLOTS = GS_BLOCK: a thread per slot)
    __shared__ int s_first[GS_HUGE_SLOTS + 1];
    {
        int mine = 0;
        if ((int)threadIdx.x < n_huge) {
            u64 off;
            int L;
            gs_huge_read(P, (int64_t)P.huge_list[threadIdx.x], off, L);
            const int max = L - k + 1, C = gs_huge_chunk_positions(max, P.huge_chunk_min);
            mine = (max + C - 1) / C;
        }
        s_first[threadIdx.x + 1] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            int sum = 0;
            s_first[0] = 0;
            for (int i = 1; i <= n_huge; i++) s_first[i] = (sum += s_first[i]);
        }
        __syncthreads();
    }
    const int n_all = s_first[n_huge];
    for (int64_t g = wave_id; g < n_all; g += n_waves) {
        int slot = 0;
        {   // the last slot whose first chunk is <= g (reads have at least one chunk each)
            int lo = 0, hi = n_huge - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_first[mid] <= (int)g)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            slot = gs_rfl(lo);
        }
        const int c = (int)g - s_first[slot];
        const int64_t r = (int64_t)P.huge_list[slot];
        u64 off;
        int L;
        gs_huge_read(P, r, off, L);
        const int max = L - k + 1;
        const uint8_t *rd = P.seq + off;
        const int C = gs_huge_chunk_positions(max, P.huge_chunk_min);
        const int n_iter_all = (max + 127) >> 7;
        const u64 key_lo = ((1ULL << 40) - 1) - ((u64)(P.first_read_no + r) & ((1ULL << 40) - 1));
        // (GS_HUGE_COPIES copies of the read's rows and counters, chunk c on copy c mod GS_HUGE_COPIES: the chunks of a chromosome vote
        // for the same handful of tax ids, and atomics on ONE address execute one after the other -- 4 883 chunks x 2 atomics per tax id
        // were most of the kernel's 0.17 ms; the finish kernel folds the copies)
        const size_t copy = (size_t)(c & (GS_HUGE_COPIES - 1)), row = ((size_t)slot * GS_HUGE_COPIES + copy) * (size_t)nv;
        uint32_t *cnt = P.huge_cnt + row, *first = P.huge_first + row;
        uint32_t *touch = P.huge_touch + (size_t)slot * GS_HUGE_COPIES * (size_t)nv;
        GsHugeHead *h = P.huge_head + (size_t)slot * GS_HUGE_COPIES + copy;
        GsHugeHead *h0 = P.huge_head + (size_t)slot * GS_HUGE_COPIES;  // (the touch list's length lives in copy 0)
        const int p0 = c * C, p1 = p0 + C < max ? p0 + C : max;
        int carry_last = GS_NODE_NONE, cur_start = p0, head_node = GS_NODE_NONE, head_len = 0;
        bool head_open = true, found = false, bad_hi = false;
        int n_miss = 0, bad_lo = 0;
        // the chunk's votes per node, one node per lane (a chunk of a genome meets a handful; the 65th goes to the rows at once)
        int c_node = -1, n_cached = 0;
        uint32_t c_cnt = 0, c_first = 0;
        for (int it = p0 >> 7; it < (p1 + 127) >> 7; it++) {
            const int base = it << 7;
            u64 Bhi[3], Blo[3], Bbad[3];
#pragma unroll
            for (int w = 0; w < 3; w++) gs_load_word(rd, L, 2 * it + w, lane, Bhi[w], Blo[w], Bbad[w]);
            {   // bad-base census as gs_process_read: words 2 it and 2 it + 1 belong to this iteration, word 2 it + 2 to the read's last
                const int q = max - 1;
                const int nw = (it == n_iter_all - 1) ? 3 : 2;
#pragma unroll
                for (int w = 0; w < 3; w++) {
                    if (w < nw) {
                        const int lo_bits = q - (base + 64 * w);
                        const u64 m_lo = lo_bits >= 64 ? ~0ULL : (lo_bits <= 0 ? 0ULL : ((1ULL << lo_bits) - 1));
                        bad_lo += __popcll(Bbad[w] & m_lo);
                        bad_hi = bad_hi || ((Bbad[w] & ~m_lo) != 0);
                    }
                }
            }
            int node[2];
            // (the marks of first-seen k-mers as ONE atomic per record line: a chromosome of the store's own species is nothing but
            // first-seen k-mers -- 5 M device-scope atomics, each a request to the fabric, were 0.09 of the chunk kernel's 0.13 ms)
            gs_probe_planes<KC, STRIPED, 2, true>(db, Bhi, Blo, Bbad, base, max, lane, node, wave_g, mk);
            const u64 hit0 = __ballot(node[0] >= 0), hit1 = __ballot(node[1] >= 0);
            found = found || ((hit0 | hit1) != 0);
            n_miss += __popcll(__ballot(node[0] == GS_NODE_MISS)) + __popcll(__ballot(node[1] == GS_NODE_MISS));
            if (base == p0) {  // the chunk's first position opens its head run (no change in front of it)
                carry_last = gs_readlane(node[0], 0);
                head_node = carry_last;
            }
            int prev[2];
            {
                const int up0 = __shfl_up(node[0], 1);
                const int up1 = __shfl_up(node[1], 1);
                const int last0 = gs_readlane(node[0], 63);
                prev[0] = lane == 0 ? carry_last : up0;
                prev[1] = lane == 0 ? last0 : up1;
            }
            const int nv0 = max - base, nv1 = max - base - 64;
            const u64 vm0 = nv0 >= 64 ? ~0ULL : (nv0 <= 0 ? 0ULL : ((1ULL << nv0) - 1ULL));
            const u64 vm1 = nv1 >= 64 ? ~0ULL : (nv1 <= 0 ? 0ULL : ((1ULL << nv1) - 1ULL));
            const u64 chg0 = __ballot(node[0] != prev[0]) & vm0;
            const u64 chg1 = __ballot(node[1] != prev[1]) & vm1;
            if ((chg0 | chg1) != 0) {  // the run that was open ends at the first change
                const int q = chg0 ? __builtin_ctzll(chg0) : 64 + __builtin_ctzll(chg1);
                if (head_open) {
                    head_len = base + q - p0;
                    head_open = false;
                } else if (carry_last >= 0 && lane == 0)
                    st.contig(carry_last, base + q - cur_start, key_lo);
            }
            {   // hit contigs that start at a change of this iteration and end before the iteration does
                const u64 a0 = (chg0 >> 1) >> lane, a1 = (chg1 >> 1) >> lane;
                const int e1 = chg1 ? 64 + __builtin_ctzll(chg1) : -1;
                const int end0 = a0 ? lane + 1 + __builtin_ctzll(a0) : e1;
                const int end1 = a1 ? 64 + lane + 1 + __builtin_ctzll(a1) : -1;
                const u64 c0m = hit0 & chg0 & (chg1 ? ~0ULL : (chg0 ? (1ULL << (63 - __builtin_clzll(chg0))) - 1ULL : 0ULL));
                const u64 c1m = hit1 & chg1 & (chg1 ? (1ULL << (63 - __builtin_clzll(chg1))) - 1ULL : 0ULL);
                if (GS_ACT(c0m)) st.contig(node[0], end0 - lane, key_lo);
                if (GS_ACT(c1m)) st.contig(node[1], end1 - 64 - lane, key_lo);
            }
            if (chg1)
                cur_start = base + 127 - __builtin_clzll(chg1);
            else if (chg0)
                cur_start = base + 63 - __builtin_clzll(chg0);
            // the distinct hit nodes of this iteration
            u64 m0 = hit0, m1 = hit1;
            while ((m0 | m1) != 0) {
                const int j = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
                const int nvj = j < 64 ? gs_readlane(node[0], j) : gs_readlane(node[1], j - 64);
                const u64 e0 = __ballot(node[0] == nvj), e1 = __ballot(node[1] == nvj);
                m0 &= ~e0;
                m1 &= ~e1;
                const uint32_t votes = (uint32_t)(__popcll(e0) + __popcll(e1));
                const u64 has = __ballot(c_node == nvj);
                if (has) {
                    if (lane == __builtin_ctzll(has)) c_cnt += votes;
                } else if (n_cached < 64) {
                    if (lane == n_cached) {
                        c_node = nvj;
                        c_cnt = votes;
                        c_first = (uint32_t)(base + j);
                    }
                    n_cached++;
                } else if (lane == 0)
                    gs_huge_vote(cnt, first, touch, h0, nvj, votes, (uint32_t)(base + j));
            }
            {
                const int last_p = (max - 1 < base + 127) ? max - 1 : base + 127;
                const int ls = (last_p - base) >> 6, ll = (last_p - base) & 63;
                carry_last = gs_readlane(ls ? node[1] : node[0], ll);
            }
        }
        if (c_node >= 0) gs_huge_vote(cnt, first, touch, h0, c_node, c_cnt, c_first);
        if (lane == 0) {
            GsHugeChunk rec;
            rec.head_node = head_node;
            rec.head_len = head_open ? p1 - p0 : head_len;
            rec.tail_node = head_open ? head_node : carry_last;
            rec.tail_len = head_open ? -1 : p1 - cur_start;
            P.huge_chunks[(size_t)slot * GS_HUGE_MAX_CHUNKS + (size_t)c] = rec;
            if (n_miss) atomicAdd(&h->n_miss, (unsigned int)n_miss);
            if (bad_lo) atomicAdd(&h->bad_lo, (unsigned int)bad_lo);
            if (found || bad_hi) atomicOr(&h->flags, (found ? 1u : 0u) | (bad_hi ? 2u : 0u));
        }
    }
    GS_STATS_EPILOGUE()
}

template <bool LDS_STATS>
__global__ __launch_bounds__(GS_BLOCK) void gs_match_huge_finish_kernel(GsMatchParams P) {
    const unsigned int n_huge_all = P.huge_count[0];
    if (n_huge_all == 0) return;
    const int n_huge = (int)(n_huge_all < (unsigned int)P.huge_slots ? n_huge_all : (unsigned int)P.huge_slots);
    GS_STATS_PROLOGUE()
    const int lane = gs_lane();
    const int wave_in_block = gs_rfl((int)(threadIdx.x >> 6));
    const int slot = (int)blockIdx.x * (GS_BLOCK / 64) + wave_in_block;
    if (slot < n_huge) {
        const GsDbDev &db = P.db;
        const int k = db.k;
        const int64_t r = (int64_t)P.huge_list[slot];
        u64 off;
        int L;
        gs_huge_read(P, r, off, L);
        const int max = L - k + 1;
        const int C = gs_huge_chunk_positions(max, P.huge_chunk_min);
        const int n_chunks = (max + C - 1) / C;
        const u64 key_lo = ((1ULL << 40) - 1) - ((u64)(P.first_read_no + r) & ((1ULL << 40) - 1));
        uint32_t *cnt_c = P.huge_cnt + (size_t)slot * GS_HUGE_COPIES * (size_t)nv, *first_c = P.huge_first + (size_t)slot * GS_HUGE_COPIES * (size_t)nv;
        uint32_t *touch = P.huge_touch + (size_t)slot * GS_HUGE_COPIES * (size_t)nv;
        uint32_t *cnt = P.huge_fold + (size_t)slot * 2 * (size_t)nv, *first = cnt + nv;  // the copies folded: votes and first position per node
        GsHugeHead *hd = P.huge_head + (size_t)slot * GS_HUGE_COPIES;
        unsigned int hflags = 0, h_miss = 0, h_bad = 0;
        if (lane < GS_HUGE_COPIES) {
            hflags = hd[lane].flags;
            h_miss = hd[lane].n_miss;
            h_bad = hd[lane].bad_lo;
        }
#pragma unroll
        for (int o = GS_HUGE_COPIES / 2; o >= 1; o >>= 1) {
            hflags |= (unsigned int)__shfl_xor((int)hflags, o);
            h_miss += (unsigned int)__shfl_xor((int)h_miss, o);
            h_bad += (unsigned int)__shfl_xor((int)h_bad, o);
        }
        hflags = (unsigned int)gs_rfl((int)hflags);
        const bool found = (hflags & 1u) != 0;
        const int tax_err = gs_rfl((int)h_miss) + gs_rfl((int)h_bad) + ((hflags & 2u) ? 1 : 0);
        const int n_touch = (int)hd[0].n_touch;
        for (int i = lane; i < n_touch; i += 64) {  // (a node that is on the list twice is folded twice, to the same values)
            const int v = (int)touch[i];
            uint32_t total = 0, fmin = 0xffffffffu;
            uint32_t cv[GS_HUGE_COPIES], fv[GS_HUGE_COPIES];  // (all loads on their way before the first is waited for)
#pragma unroll
            for (int cp = 0; cp < GS_HUGE_COPIES; cp++) {
                cv[cp] = cnt_c[(size_t)cp * nv + v];
                fv[cp] = first_c[(size_t)cp * nv + v];
            }
#pragma unroll
            for (int cp = 0; cp < GS_HUGE_COPIES; cp++) {
                total += cv[cp];
                fmin = fv[cp] < fmin ? fv[cp] : fmin;
            }
            cnt[v] = total;
            first[v] = fmin;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int out_class = -1, out_flags = 0;
        // ---- the runs across the seams, 64 chunks at a time, one per lane.  With link(c) = "chunk c starts with the node chunk c - 1
        // ends with" and whole(c) = "no change inside c", the length of the run that is open at the end of chunk c is
        //     out(c) = whole(c) && link(c) ? out(c - 1) + head_len(c) : (whole(c) ? head_len(c) : tail_len(c))
        // -- a segmented sum --, the run that is open at the end of c - 1 is closed at the seam if !link(c) (length out(c - 1)), and the
        // head run of c is closed inside c if !whole(c) (length head_len(c), plus out(c - 1) if link(c)).
        {
            const GsHugeChunk *ch = P.huge_chunks + (size_t)slot * GS_HUGE_MAX_CHUNKS;
            int carry_node = GS_NODE_NONE, carry_out = 0;  // tail node and out() of the chunk before this round's first
            const GsHugeChunk none = {GS_NODE_NONE, 0, GS_NODE_NONE, -1};
            GsHugeChunk next = lane < n_chunks ? ch[lane] : none;
            for (int c0 = 0; c0 < n_chunks; c0 += 64) {
                const bool valid = c0 + lane < n_chunks;
                const GsHugeChunk mine = next;
                next = c0 + 64 + lane < n_chunks ? ch[c0 + 64 + lane] : none;  // (on its way while this round is booked)
                const int up_node = __shfl_up(mine.tail_node, 1);
                const int prev_node = lane == 0 ? carry_node : up_node;
                const bool link = valid && (c0 + lane > 0) && mine.head_node == prev_node;
                const bool whole = mine.tail_len < 0;
                // segmented inclusive sum: a lane that does not continue its predecessor's run starts a segment
                int val = whole ? mine.head_len : mine.tail_len;
                bool cont = whole && link;  // "my value is added to the one before me"
                if (lane == 0 && cont) {
                    val += carry_out;
                    cont = false;
                }
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int pv = __shfl_up(val, d);
                    const int pc = __shfl_up((int)cont, d);
                    if (lane >= d && cont) {
                        val += pv;
                        cont = pc != 0;
                    }
                }
                const int up_out = __shfl_up(val, 1);
                const int prev_out = lane == 0 ? carry_out : up_out;
                if (valid && c0 + lane > 0 && !link && prev_node >= 0) st.contig(prev_node, prev_out, key_lo);
                if (valid && !whole && mine.head_node >= 0) st.contig(mine.head_node, mine.head_len + (link ? prev_out : 0), key_lo);
                const int last = (n_chunks - c0 < 64 ? n_chunks - c0 : 64) - 1;
                carry_node = gs_readlane(mine.tail_node, last);
                carry_out = gs_readlane(val, last);
            }
            if (found && carry_node >= 0 && lane == 0) st.contig(carry_node, carry_out, key_lo);  // tail flush (:455-473)
        }
        if (found) {
            out_flags = GS_F_FOUND | GS_F_RETURNED;
            // ---- the distinct nodes in order of first appearance: reads1KMer and the candidate paths (two register sets: up to 128)
            int path[2] = {-1, -1}, ptin[2] = {0, 0}, ptout[2] = {0, 0}, used = 0;
            uint32_t last_pos = 0;
            for (int step = 0; step < n_touch; step++) {
                // the node with the smallest first position beyond the last one taken (one node per position: a tie is the same node twice)
                uint32_t best = 0xffffffffu;
                int best_v = -1;
                for (int i = lane; i < n_touch; i += 64) {
                    const int v = (int)touch[i];
                    const uint32_t fp = first[v];
                    if ((step == 0 || fp > last_pos) && fp < best) {
                        best = fp;
                        best_v = v;
                    }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const uint32_t ob = (uint32_t)__shfl_xor((int)best, o);
                    const int ov = __shfl_xor(best_v, o);
                    if (ob < best) {
                        best = ob;
                        best_v = ov;
                    }
                }
                best = (uint32_t)gs_rfl((int)best);
                const int nvj = gs_rfl(best_v);
                if (nvj < 0) break;
                last_pos = best;
                if (lane == 0) st.add(nvj, GS_S_READS_1KMER, 1);
                if (P.classify) {  // mergeReadTaxidPath (:568-586), as gs_process_read
                    const int ntin = st.tin[nvj], ntout = st.tout[nvj];
                    bool related = false;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        if (!related) {
                            const bool mine = 64 * h + lane < used;
                            const bool a = mine && gs_anc_or_self(ptin[h], ptout[h], ntin);
                            const bool b = mine && gs_anc_or_self(ntin, ntout, ptin[h]);
                            const u64 m = __ballot(a || b);
                            if (m) {
                                const int i = __builtin_ctzll(m);
                                if (lane == i && a) {
                                    path[h] = nvj;
                                    ptin[h] = ntin;
                                    ptout[h] = ntout;
                                }
                                related = true;
                            }
                        }
                    }
                    if (!related && used < P.max_paths) {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            if (64 * h + lane == used) {
                                path[h] = nvj;
                                ptin[h] = ntin;
                                ptout[h] = ntout;
                            }
                        }
                        used++;
                    }
                }
            }
            // ---- classification (:474-531) from the vote counts, as the long-read path does it from its rows
            if (P.classify) {
                const double m = P.max_read_tax_err;
                const bool disabled = m >= 0 && ((m >= 1 && (double)tax_err > m) || ((double)tax_err > m * (double)max));
                if (!disabled) {
                    int cn = -1, first_node = -1, best = 0;
                    int sum[2] = {0, 0};
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        if (64 * h + lane < used)
                            for (int x = path[h]; x >= 0; x = st.parent[x]) sum[h] += (int)cnt[x];
                    u64 tie_mask[2] = {0, 0};
                    for (int i = 0; i < used; i++) {
                        const int si = i < 64 ? gs_readlane(sum[0], i) : gs_readlane(sum[1], i - 64);
                        if (si > best) {
                            best = si;
                            tie_mask[0] = tie_mask[1] = 0;
                        }
                        if (si >= best) tie_mask[i >> 6] |= 1ULL << (i & 63);
                    }
                    int cand[2] = {path[0], path[1]};
                    if (P.threshold > 1) {  // lowestNodeWhereSumAboveThreshold per tied path (SmallTaxTree.java:208-221)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            int mapped = -1;
                            if ((tie_mask[h] >> lane) & 1ULL) {
                                int acc = 0;
                                for (int x = path[h]; x >= 0 && mapped < 0; x = st.parent[x]) {
                                    const int cx = (int)cnt[x];
                                    if (cx != 0) {
                                        acc += cx;
                                        if (acc >= P.threshold) mapped = x;
                                    }
                                }
                            }
                            cand[h] = mapped;
                        }
                    }
                    {
                        bool firstc = true;
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            u64 tm = tie_mask[h];
                            while (tm) {
                                const int i = __builtin_ctzll(tm);
                                tm &= tm - 1;
                                const int x = gs_readlane(cand[h], i);
                                if (firstc) {
                                    cn = x;
                                    first_node = x;
                                    firstc = false;
                                } else
                                    cn = gs_lca(st, cn, x);
                            }
                        }
                    }
                    out_class = cn;
                    if (cn < 0) {
                        out_flags &= ~GS_F_RETURNED;
                    } else {
                        int read_kmers = best;
                        if (P.threshold > 1) {
                            read_kmers = 0;
                            for (int x = first_node; x >= 0; x = st.parent[x]) read_kmers += (int)cnt[x];
                        }
                        const int class_err = max - read_kmers;
                        const double mc = P.max_read_class_err;
                        if (mc < 0 || (mc >= 1 && (double)class_err <= mc) || ((double)class_err <= mc * (double)max)) {
                            out_flags |= GS_F_COUNTED;
                            if (lane == 0) {
                                const double err = (double)tax_err / (double)max;
                                const double cerr = (double)class_err / (double)max;
                                st.add(cn, GS_S_READS, 1);
                                st.add(cn, GS_S_READS_KMERS, (u64)read_kmers);
                                st.add(cn, GS_S_READS_BPS, (u64)L);
                                st.dadd(cn, GS_D_ERR_SUM, err);
                                st.dadd(cn, GS_D_ERR_SQ_SUM, err * err);
                                st.dadd(cn, GS_D_CLASS_ERR_SUM, cerr);
                                st.dadd(cn, GS_D_CLASS_ERR_SQ_SUM, cerr * cerr);
                            }
                        }
                    }
                }
            }
        }
        if (lane == 0) {
            if (P.class_vi) P.class_vi[r] = out_class;
            if (P.flags) P.flags[r] = (uint8_t)out_flags;
        }
        // the read's rows, clean for the next batch (every lane has read what it needs of them: the writes come last)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n_touch; i += 64) {
            const int v = (int)touch[i];
            cnt[v] = 0;
            for (int cp = 0; cp < GS_HUGE_COPIES; cp++) {
                cnt_c[(size_t)cp * nv + v] = 0;
                first_c[(size_t)cp * nv + v] = 0xffffffffu;
            }
        }
        if (lane < GS_HUGE_COPIES) hd[lane].n_miss = hd[lane].bad_lo = hd[lane].flags = hd[lane].n_touch = 0;
    }
    GS_STATS_EPILOGUE()
}

// ---------------------------------------------------------------------------------------------------
// the deferred statistics (GsStatRec) of a batch into the counters: one LDS table per workgroup for the value indices
// [lo, hi) -- this kernel has the whole LDS for it --, flushed with one global atomic per touched counter and workgroup
// ---------------------------------------------------------------------------------------------------
#define GS_REDUCE_VALUES 640
// the value indices of the records as an array of their own: with more than 640 values the reduce runs once per 640 of them, and a
// pass that reads 4 bytes per record instead of touching every 64-byte record is 16 times lighter
__global__ __launch_bounds__(256) void gs_stat_vi_kernel(const GsStatRec *recs, const u64 *count, int32_t *vi) {
    const int64_t n = (int64_t)*count;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) vi[i] = recs[i].vi;
}

__global__ __launch_bounds__(1024) void gs_stat_reduce_kernel(const GsStatRec *recs, const int32_t *vis, const u64 *count, int lo, int hi, u64 *sums,
                                                              u64 *maxk, double *dsums) {
    const int64_t n = (int64_t)*count;  // (0 for a refused text chunk: the match kernel handed out no records)
    if (n == 0) return;
    __shared__ u64 s_sums[GS_REDUCE_VALUES * GS_N_SUMS];
    __shared__ u64 s_max[GS_REDUCE_VALUES];
    __shared__ double s_d[GS_REDUCE_VALUES * GS_N_DCOLS];
    const int nvl = hi - lo;
    for (int i = threadIdx.x; i < nvl * GS_N_SUMS; i += blockDim.x) s_sums[i] = 0;
    for (int i = threadIdx.x; i < nvl; i += blockDim.x) s_max[i] = 0;
    for (int i = threadIdx.x; i < nvl * GS_N_DCOLS; i += blockDim.x) s_d[i] = 0.0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int vi = vis ? vis[i] : recs[i].vi;
        if (vi < lo || vi >= hi) continue;
        const GsStatRec rc = recs[i];
        u64 *row = s_sums + (size_t)(vi - lo) * GS_N_SUMS;
        atomicAdd(&row[GS_S_READS_1KMER], 1ULL);
        atomicAdd(&row[GS_S_KMERS], (u64)rc.kmers);
        atomicAdd(&row[GS_S_CONTIGS], (u64)rc.contigs);
        atomicAdd(&row[GS_S_CONTIG_LEN_SQ_SUM], (u64)rc.sq);
        atomicMax(&s_max[vi - lo], rc.max_key);
        if (rc.counted) {
            double *dr = s_d + (size_t)(vi - lo) * GS_N_DCOLS;
            atomicAdd(&row[GS_S_READS], 1ULL);
            atomicAdd(&row[GS_S_READS_KMERS], (u64)rc.read_kmers);
            atomicAdd(&row[GS_S_READS_BPS], (u64)rc.read_len);
            atomicAdd(&dr[GS_D_ERR_SUM], rc.err);
            atomicAdd(&dr[GS_D_ERR_SQ_SUM], rc.err * rc.err);
            atomicAdd(&dr[GS_D_CLASS_ERR_SUM], rc.cerr);
            atomicAdd(&dr[GS_D_CLASS_ERR_SQ_SUM], rc.cerr * rc.cerr);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nvl * GS_N_SUMS; i += blockDim.x)
        if (s_sums[i]) atomicAdd(&sums[(size_t)lo * GS_N_SUMS + i], s_sums[i]);
    for (int i = threadIdx.x; i < nvl; i += blockDim.x)
        if (s_max[i]) atomicMax(&maxk[lo + i], s_max[i]);
    for (int i = threadIdx.x; i < nvl * GS_N_DCOLS; i += blockDim.x)
        if (s_d[i] != 0.0) atomicAdd(&dsums[(size_t)lo * GS_N_DCOLS + i], s_d[i]);
}

// n_values <= GS_STAT_REC_MAX_VALUES: a pass of gs_stat_reduce_kernel per 640 values
// n_max: upper bound of the record count (the kernels read the real one from *count)
// vi_scratch: room for n_max value indices, used when the reduce takes more than one pass (nullptr: every pass reads the records)
extern "C" hipError_t gs_launch_stat_reduce(const GsStatRec *recs, const void *count, int64_t n_max, int n_values, void *sums, void *maxk,
                                             void *dsums, int32_t *vi_scratch, hipStream_t stream) {
    if (n_max <= 0) return hipSuccess;
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n_max + 8191) / 8192, 512));
    const int32_t *vis = nullptr;
    if (vi_scratch != nullptr && n_values > GS_REDUCE_VALUES) {
        hipLaunchKernelGGL(gs_stat_vi_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n_max + 1023) / 1024, 4096))), dim3(256), 0, stream,
                           recs, (const u64 *)count, vi_scratch);
        vis = vi_scratch;
    }
    for (int lo = 0; lo < n_values; lo += GS_REDUCE_VALUES)
        hipLaunchKernelGGL(gs_stat_reduce_kernel, dim3(grid), dim3(1024), 0, stream, recs, vis, (const u64 *)count, lo,
                           std::min(n_values, lo + GS_REDUCE_VALUES), (u64 *)sums, (u64 *)maxk, (double *)dsums);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// DB-partitioned mode: the fused kernel is split into
//   gs_encode_kernel      reads -> mixed key h of every k-mer position (GS_KEY_INVALID for windows with a bad base,
//                         GS_KEY_MISS for k-mers the store-wide minimizer gate rules out)
//   gs_probe_keys_kernel  keys -> node (value index / miss), run by the rank that OWNS the key's table partition
//   gs_match_kernel<.., FROM_NODES = true>   per-read reduce over the routed-back node stream
// with two all-to-all exchanges in between (genestrip_amd/distributed.py).
// ---------------------------------------------------------------------------------------------------
#define GS_KEY_INVALID (~0ULL)
#define GS_KEY_MISS (~0ULL - 1)  // the minimizer gate already says "not stored": never routed, node = MISS
#define GS_KEY_ROUTED(h) ((h) < GS_KEY_MISS)

template <int KC>
__global__ __launch_bounds__(GS_BLOCK) void gs_encode_kernel(GsEncodeParams P) {
    __shared__ uint32_t s_g[GS_BLOCK / 64][2 * GS_ROW];  // 15-mer order hashes of the wave's current 128 + 16 positions
    const int lane = gs_lane();
    uint32_t *wave_g = s_g[threadIdx.x >> 6];
    const int64_t wave_id = (int64_t)blockIdx.x * (GS_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (GS_BLOCK / 64);
    const int k = KC ? KC : P.k;  // KC = compile-time k (0: any k)
    const uint32_t kmask = (1u << k) - 1u;
    // software pipeline over the wave's reads: the first 192 bases of the next read are loaded while the current
    // one is hashed (a read is a chain of dependent loads: offsets -> bases -> gate word)
    u64 off = 0, pb = 0;
    int L = 0;
    uint32_t c[3] = {0, 0, 0};
    if (wave_id < P.n_reads) {
        off = P.off[wave_id];
        L = (int)(P.off[wave_id + 1] - off);
        pb = P.pos_off[wave_id];
#pragma unroll
        for (int i = 0; i < 3; i++) c[i] = 64 * i + lane < L ? P.seq[off + 64 * i + lane] : GS_FILL;
    }
    for (int64_t r = wave_id; r < P.n_reads; r += n_waves) {
        const int max = L - k + 1;
        const uint8_t *rd = P.seq + off;
        const int64_t rn = r + n_waves;
        u64 offN = 0, pbN = 0;
        int LN = 0;
        if (rn < P.n_reads) {
            offN = P.off[rn];
            LN = (int)(P.off[rn + 1] - offN);
            pbN = P.pos_off[rn];
        }
        uint32_t cN[3] = {0, 0, 0};
        for (int base = 0; base < max || base == 0; base += 128) {
            u64 Bhi[3], Blo[3], Bbad[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                if (base == 0)
                    gs_word_from_byte(c[i], Bhi[i], Blo[i], Bbad[i]);
                else
                    gs_load_word(rd, L, (base >> 6) + i, lane, Bhi[i], Blo[i], Bbad[i]);
            }
            if (base == 0 && rn < P.n_reads) {
#pragma unroll
                for (int i = 0; i < 3; i++) cN[i] = 64 * i + lane < LN ? P.seq[offN + 64 * i + lane] : GS_FILL;
            }
            if (max <= 0) break;
            u64 key[2];
            uint32_t fhi[2], flo[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                fhi[s] = (uint32_t)gs_funnel(Bhi[s], Bhi[s + 1], lane) & kmask;
                flo[s] = (uint32_t)gs_funnel(Blo[s], Blo[s + 1], lane) & kmask;
                const uint32_t wbad = (uint32_t)gs_funnel(Bbad[s], Bbad[s + 1], lane) & kmask;
                key[s] = wbad ? GS_KEY_INVALID : gs_kmer_hash(fhi[s], flo[s], k, kmask);
            }
            if (P.mgate != nullptr) {
                // the store's minimizer gate (which covers the keys of every partition) as in gs_probe_planes: k-mers
                // it rules out are not routed at all
                int mp[2];
                uint32_t cf[2];
                uint32_t rhi[2], rlo[2];
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    rhi[s] = __brev(fhi[s]) >> (32 - k);
                    rlo[s] = (__brev(flo[s]) >> (32 - k)) ^ kmask;
                }
                gs_wave_minimizers<KC>(Bhi, Blo, fhi, flo, rhi, rlo, k, lane, wave_g, mp, cf);
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    if (base + 64 * s + lane < max && key[s] != GS_KEY_INVALID) {
                        const uint32_t gh = gs_canon_hash(cf[s] >> 1);  // only the minimizer's hash is needed
                        const uint32_t bits = gs_mgate_bits(gh);
                        if ((P.mgate[gs_mgate_word(gh, P.mgate_bits)] & bits) != bits) key[s] = GS_KEY_MISS;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int p = base + 64 * s + lane;
                if (p < max) P.keys[pb + (u64)p] = key[s];
            }
        }
        off = offN;
        L = LN;
        pb = pbN;
#pragma unroll
        for (int i = 0; i < 3; i++) c[i] = cN[i];
    }
}

// ---- encode and routing in one pass: what gs_encode_kernel + gs_route_count_kernel + gs_route_scatter_kernel do through
// an 8-byte key per k-mer POSITION that is written once and read twice.  Here a wave appends the keys of its reads
// straight to the owners' send regions: it holds one chunk of GS_ROUTE_CHUNK slots per owner (cursor in LDS), takes a
// new one with a single global atomic when the chunk cannot take a sub-round's keys (the rest of the old chunk becomes
// sentinels), and pads its open chunks at the end.  Keys leave in no particular order inside an owner's region.
template <int KC>
__global__ __launch_bounds__(GS_BLOCK) void gs_encode_route_kernel(GsEncodeParams P, GsRouteParams R) {
    __shared__ uint32_t s_g[GS_BLOCK / 64][2 * GS_ROW];
    __shared__ unsigned long long s_base[GS_BLOCK / 64][64];  // per wave and owner: base of the chunk in hand
    __shared__ uint32_t s_used[GS_BLOCK / 64][64];            // slots used in it (GS_ROUTE_CHUNK: none in hand)
    const int lane = gs_lane();
    const int wv = threadIdx.x >> 6;
    uint32_t *wave_g = s_g[wv];
    const int64_t wave_id = (int64_t)blockIdx.x * (GS_BLOCK / 64) + wv;
    const int64_t n_waves = (int64_t)gridDim.x * (GS_BLOCK / 64);
    const int k = KC ? KC : P.k;
    const uint32_t kmask = (1u << k) - 1u;
    const int n_parts = R.n_parts;
    if (lane < n_parts) {
        s_base[wv][lane] = 0;
        s_used[wv][lane] = GS_ROUTE_CHUNK;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int64_t r = wave_id; r < P.n_reads; r += n_waves) {
        const u64 off = P.off[r];
        const int L = (int)(P.off[r + 1] - off);
        const u64 pb = P.pos_off[r];
        const int max = L - k + 1;
        const uint8_t *rd = P.seq + off;
        for (int base = 0; base < max; base += 128) {
            u64 Bhi[3], Blo[3], Bbad[3];
#pragma unroll
            for (int i = 0; i < 3; i++) gs_load_word(rd, L, (base >> 6) + i, lane, Bhi[i], Blo[i], Bbad[i]);
            u64 key[2];
            uint32_t fhi[2], flo[2];
#pragma unroll
            for (int s = 0; s < 2; s++) {
                fhi[s] = (uint32_t)gs_funnel(Bhi[s], Bhi[s + 1], lane) & kmask;
                flo[s] = (uint32_t)gs_funnel(Blo[s], Blo[s + 1], lane) & kmask;
                const uint32_t wbad = (uint32_t)gs_funnel(Bbad[s], Bbad[s + 1], lane) & kmask;
                key[s] = wbad ? GS_KEY_INVALID : gs_kmer_hash(fhi[s], flo[s], k, kmask);
            }
            if (P.mgate != nullptr) {
                int mp[2];
                uint32_t cf[2];
                uint32_t rhi[2], rlo[2];
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    rhi[s] = __brev(fhi[s]) >> (32 - k);
                    rlo[s] = (__brev(flo[s]) >> (32 - k)) ^ kmask;
                }
                gs_wave_minimizers<KC>(Bhi, Blo, fhi, flo, rhi, rlo, k, lane, wave_g, mp, cf);
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    if (base + 64 * s + lane < max && key[s] != GS_KEY_INVALID) {
                        const uint32_t gh = gs_canon_hash(cf[s] >> 1);
                        const uint32_t bits = gs_mgate_bits(gh);
                        if ((P.mgate[gs_mgate_word(gh, P.mgate_bits)] & bits) != bits) key[s] = GS_KEY_MISS;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int p = base + 64 * s + lane;
                const bool in = p < max;
                const bool routed = in && GS_KEY_ROUTED(key[s]);
                if (in) R.nodes[pb + (u64)p] = key[s] == GS_KEY_INVALID ? GS_NODE_INVALID : GS_NODE_MISS;  // (routed: overwritten later)
                const int owner = routed ? (int)((key[s] >> GS_OWNER_SHIFT) % (u64)n_parts) : -1;
                u64 todo = __ballot(routed);
                while (todo) {  // one owner after the other
                    const int o = gs_readlane(owner, __builtin_ctzll(todo));
                    const u64 mine = __ballot(owner == o);
                    const uint32_t cnt = (uint32_t)__popcll(mine);
                    uint32_t used = s_used[wv][o];
                    u64 cbase = s_base[wv][o];
                    if (used + cnt > GS_ROUTE_CHUNK) {
                        // the rest of the chunk in hand becomes sentinels; a new chunk
                        if (used < GS_ROUTE_CHUNK && cbase + GS_ROUTE_CHUNK <= R.cap)
                            for (uint32_t j = used + (uint32_t)lane; j < GS_ROUTE_CHUNK; j += 64) {
                                R.send_keys[(u64)o * R.cap + cbase + j] = GS_KEY_INVALID;
                                R.send_idx[(u64)o * R.cap + cbase + j] = 0xffffffffu;
                            }
                        u64 nb = 0;
                        if (lane == 0) nb = atomicAdd(&R.cursors[o], (u64)GS_ROUTE_CHUNK);
                        cbase = ((u64)(uint32_t)gs_rfl((int)(nb >> 32)) << 32) | (uint32_t)gs_rfl((int)nb);
                        used = 0;
                        if (lane == 0) s_base[wv][o] = cbase;
                    }
                    const bool fits = cbase + GS_ROUTE_CHUNK <= R.cap;  // (else: the region is full, the host falls back)
                    if (!fits && lane == 0) R.cursors[64] = 1;
                    if (owner == o && fits) {
                        const u64 at = (u64)o * R.cap + cbase + used + (u64)__popcll(mine & ((1ULL << lane) - 1));
                        R.send_keys[at] = key[s];
                        R.send_idx[at] = (uint32_t)(pb + (u64)p);
                    }
                    if (lane == 0) s_used[wv][o] = used + cnt;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    todo &= ~mine;
                }
            }
        }
    }
    // pad the chunks that are still open
    for (int o = 0; o < n_parts; o++) {
        const uint32_t used = s_used[wv][o];
        const u64 cbase = s_base[wv][o];
        if (used < GS_ROUTE_CHUNK && cbase + GS_ROUTE_CHUNK <= R.cap)
            for (uint32_t j = used + (uint32_t)lane; j < GS_ROUTE_CHUNK; j += 64) {
                R.send_keys[(u64)o * R.cap + cbase + j] = GS_KEY_INVALID;
                R.send_idx[(u64)o * R.cap + cbase + j] = 0xffffffffu;
            }
    }
}

// scatter of the answers of one owner region: nodes[idx[i]] = back[i] (sentinel slots skipped)
__global__ __launch_bounds__(256) void gs_unroute_region_kernel(const uint32_t *idx, const int32_t *back, int64_t n, int32_t *nodes) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t at = idx[i];
        if (at != 0xffffffffu) nodes[at] = back[i];
    }
}

extern "C" hipError_t gs_launch_encode_route(const GsEncodeParams *P, const GsRouteParams *R, int grid, hipStream_t stream) {
    if (P->k == 31)
        hipLaunchKernelGGL(gs_encode_route_kernel<31>, dim3(grid), dim3(GS_BLOCK), 0, stream, *P, *R);
    else
        hipLaunchKernelGGL(gs_encode_route_kernel<0>, dim3(grid), dim3(GS_BLOCK), 0, stream, *P, *R);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_unroute_region(const uint32_t *idx, const int32_t *back, int64_t n, int32_t *nodes, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_unroute_region_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 256 * 16)), dim3(256), 0, stream, idx, back, n, nodes);
    return hipGetLastError();
}

// ---- routing of the keys to their owner ranks (counting sort by owner; order inside an owner group is arbitrary,
// idx remembers where every routed key came from)
__global__ __launch_bounds__(256) void gs_route_count_kernel(const u64 *keys, int64_t n, int n_parts, u64 *counts) {
    __shared__ unsigned int s_cnt[64];
    const int lane = gs_lane();
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL; base < n; base += stride) {
        const int64_t i = base + lane;
        const u64 h = i < n ? keys[i] : GS_KEY_INVALID;
        const int owner = GS_KEY_ROUTED(h) ? (int)((h >> GS_OWNER_SHIFT) % (u64)n_parts) : -1;
        u64 todo = __ballot(owner >= 0);
        while (todo) {  // one LDS atomic per wave and distinct owner
            const int o = gs_readlane(owner, __builtin_ctzll(todo));
            const u64 mine = __ballot(owner == o);
            if (lane == 0) atomicAdd(&s_cnt[o], (unsigned int)__popcll(mine));
            todo &= ~mine;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_parts && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (u64)s_cnt[threadIdx.x]);
}

// tiles of 4096 keys per workgroup: ranks inside the tile come from LDS counters (one LDS atomic per wave and owner),
// then ONE global atomic per owner and tile reserves the output range -- a single hot cursor would serialise the chip
#define GS_ROUTE_T 16
// nodes (may be NULL): the positions that are not routed get their node here already (miss / invalid window), so that
// gs_unroute_kernel only has to scatter what comes back
__global__ __launch_bounds__(256) void gs_route_scatter_kernel(const u64 *keys, int64_t n, int n_parts, u64 *cursors,
                                                              u64 *send_keys, uint32_t *idx, int32_t *nodes) {
    __shared__ unsigned int s_cnt[64];
    __shared__ u64 s_base[64];
    const int lane = gs_lane();
    const int64_t tile = 256 * GS_ROUTE_T;
    for (int64_t t0 = (int64_t)blockIdx.x * tile; t0 < n; t0 += (int64_t)gridDim.x * tile) {
        if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        u64 h[GS_ROUTE_T];
        int owner[GS_ROUTE_T];
        unsigned int lp[GS_ROUTE_T];
#pragma unroll
        for (int j = 0; j < GS_ROUTE_T; j++) {
            const int64_t i = t0 + (int64_t)j * 256 + threadIdx.x;
            h[j] = i < n ? keys[i] : GS_KEY_INVALID;
            owner[j] = GS_KEY_ROUTED(h[j]) ? (int)((h[j] >> GS_OWNER_SHIFT) % (u64)n_parts) : -1;
            if (nodes != nullptr && i < n && owner[j] < 0) nodes[i] = h[j] == GS_KEY_MISS ? GS_NODE_MISS : GS_NODE_INVALID;
            lp[j] = 0;
            u64 todo = __ballot(owner[j] >= 0);
            while (todo) {
                const int first = __builtin_ctzll(todo);
                const int o = gs_readlane(owner[j], first);
                const u64 mine = __ballot(owner[j] == o);
                unsigned int b = 0;
                if (lane == first) b = atomicAdd(&s_cnt[o], (unsigned int)__popcll(mine));
                b = (unsigned int)gs_readlane((int)b, first);
                if (owner[j] == o) lp[j] = b + (unsigned int)__popcll(mine & ((1ULL << lane) - 1));
                todo &= ~mine;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < n_parts) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], (u64)s_cnt[threadIdx.x]) : 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < GS_ROUTE_T; j++) {
            if (owner[j] >= 0) {
                const u64 p = s_base[owner[j]] + lp[j];
                send_keys[p] = h[j];
                idx[p] = (uint32_t)(t0 + (int64_t)j * 256 + threadIdx.x);
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void gs_unroute_kernel(const u64 *keys, const uint32_t *idx, const int32_t *back,
                                                        int64_t n_routed, int32_t *nodes, int64_t n_keys, int phase) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (phase == 0)  // positions that were never routed: windows with a bad base, k-mers the gate ruled out
                     // (only when gs_route_scatter_kernel has not written them already)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += stride)
            nodes[i] = keys[i] == GS_KEY_MISS ? GS_NODE_MISS : GS_NODE_INVALID;
    else
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_routed; i += stride) nodes[idx[i]] = back[i];
}

// one lane per key; marks the slot's seen bit like the fused kernel does
__global__ __launch_bounds__(256) void gs_probe_keys_kernel(GsDbDev db, const u64 *keys, int64_t n, int32_t *nodes,
                                                           int count_unique) {
    const uint32_t vmask2 = 2u * ((1u << db.vbits) - 1u);
    const int shift_rem = (int)db.vbits + 3;
    const uint32_t bmask = (uint32_t)db.bucket_mask;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 h = keys[i];
        int node = h == GS_KEY_MISS ? GS_NODE_MISS : GS_NODE_INVALID;
        if (GS_KEY_ROUTED(h)) {
            node = GS_NODE_MISS;
            bool cand = true;
            if (db.gate != nullptr) {
                const u64 g = gs_gate_bits(h);
                cand = (db.gate[(h >> db.bucket_bits) & db.gate_mask] & g) == g;
            }
            const uint32_t home = (uint32_t)h & bmask;
            const u64 want = (h >> db.bucket_bits) << shift_rem;
            for (int disp = 0; cand && disp <= GS_MAX_DISP; disp++) {
                const uint32_t b = (home + (uint32_t)disp) & bmask;
                GsBucket bk;
                gs_load_bucket(db.table, b, bk);
                int vs = -1, sl = 0;
                const bool done = gs_match_bucket(bk, want | ((u64)disp << (db.vbits + 1)), vmask2, vs, sl);
                if (vs >= 0) {
                    node = vs >> 1;
                    if (count_unique && (vs & 1) == 0)
                        atomicOr(const_cast<u64 *>(db.table) + (size_t)b * GS_SLOTS_PER_BUCKET + sl, 1ULL);
                }
                if (done) break;
            }
        }
        nodes[i] = node;
    }
}

// ---------------------------------------------------------------------------------------------------
// Kraken-style segments (FastqKMerMatcher.printKrakenStyleOut, :597-611 / :391-394 / :452-454): the maximal runs
// of equal node over the k-mer positions of a read.  WRITE = false counts them per read; WRITE = true stores
// (code, start) of every run at seg_off[r] + i (code = value index, -1 miss "0", -2 INVALID "A").
// ---------------------------------------------------------------------------------------------------
// iterations [it0, it1) of one read: runs that start there, written behind out_base when WRITE; carry_last: node in front of it0
template <bool WRITE, bool STRIPED>
__device__ __forceinline__ uint32_t gs_segments_span(const GsSegParams &P, const uint8_t *rd, int L, int max, int it0, int it1, int &carry_last, u64 out_base,
                                                     int lane, uint32_t *wave_g, int &first_node) {
    const GsMark nomark = {0, nullptr, nullptr, nullptr};
    uint32_t nseg = 0;
    for (int it = it0; it < it1; it++) {
        const int base = it << 7;
        u64 Bhi[3], Blo[3], Bbad[3];
#pragma unroll
        for (int w = 0; w < 3; w++) gs_load_word(rd, L, 2 * it + w, lane, Bhi[w], Blo[w], Bbad[w]);
        int node[2];
        gs_probe_planes<0, STRIPED>(P.db, Bhi, Blo, Bbad, base, max, lane, node, wave_g, nomark);
        if (it == it0) first_node = gs_readlane(node[0], 0);
        const int up0 = __shfl_up(node[0], 1), up1 = __shfl_up(node[1], 1);
        const int last0 = gs_readlane(node[0], 63);
        const int prev[2] = {lane == 0 ? carry_last : up0, lane == 0 ? last0 : up1};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const bool head = node[s] != GS_NODE_NONE && node[s] != prev[s];
            const u64 H = __ballot(head);
            if (WRITE && head) {
                const u64 idx = out_base + nseg + (u64)__popcll(H & ((1ULL << lane) - 1));
                P.seg_code[idx] = node[s];
                P.seg_start[idx] = base + 64 * s + lane;
            }
            nseg += (uint32_t)__popcll(H);
        }
        const int last_p = (max - 1 < base + 127) ? max - 1 : base + 127;
        carry_last = gs_readlane(((last_p - base) >> 6) ? node[1] : node[0], (last_p - base) & 63);
    }
    return nseg;
}

template <bool WRITE, bool STRIPED>
__global__ __launch_bounds__(GS_BLOCK) void gs_segments_kernel(GsSegParams P) {
    __shared__ __attribute__((aligned(8))) uint32_t s_g[GS_BLOCK / 64][2 * GS_ROW + (STRIPED ? GS_STRIPE_WORDS : 0)];
    const GsDbDev &db = P.db;
    const int lane = gs_lane();
    const int wave_in_block = (int)(threadIdx.x >> 6);
    GS_STRIPE_TABLE(&P)
    const int64_t wave_id = (int64_t)blockIdx.x * (GS_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (GS_BLOCK / 64);
    const int k = db.k;
    if (P.pieces != nullptr) {  // the pieces of the reads that were left out
        for (int64_t i = wave_id; i < P.n_pieces; i += n_waves) {
            const GsSegPiece pc = P.pieces[i];
            const uint64_t *po = P.off + (int64_t)pc.read * P.off_stride;
            const u64 off = po[0];
            const int L = (int)(po[1] - off);
            int carry_last = WRITE ? pc.carry : GS_NODE_NONE, first_node = GS_NODE_NONE;
            const uint32_t nseg = gs_segments_span<WRITE, STRIPED>(P, P.seq + off, L, L - k + 1, pc.it0, pc.it0 + pc.n_iter, carry_last,
                                                                   WRITE ? P.seg_off[pc.read] + pc.out_off : 0, lane, s_g[wave_in_block], first_node);
            if (!WRITE && lane == 0) {
                GsSegPieceOut o = {first_node, carry_last, nseg, 0};
                P.piece_out[i] = o;
            }
        }
        return;
    }
    for (int64_t r = wave_id; r < P.n_reads; r += n_waves) {
        const uint64_t *po = P.off + r * P.off_stride;
        const u64 off = po[0];
        const int L = (int)(po[1] - off);
        const int max = L - k + 1;
        if (max >= P.huge_min) {
            if (!WRITE && lane == 0) P.seg_count[r] = GS_SEG_HUGE;
            continue;
        }
        int carry_last = GS_NODE_NONE, first_node;
        const uint32_t nseg = gs_segments_span<WRITE, STRIPED>(P, P.seq + off, L, max, 0, max > 0 ? (max + 127) >> 7 : 0, carry_last, WRITE ? P.seg_off[r] : 0, lane,
                                                               s_g[wave_in_block], first_node);
        if (!WRITE && lane == 0) P.seg_count[r] = nseg;
    }
}

// ---------------------------------------------------------------------------------------------------
// unique k-mers per value: scan the bitmap, look the set slots up in the table
// (KMerUniqueCounterBits.getUniqueKmerCounts, C/store/KMerUniqueCounterBits.java:146-163)
// ---------------------------------------------------------------------------------------------------
#define GS_UNIQ_LDS 8192  // value indices whose unique counters are privatised per workgroup
__global__ __launch_bounds__(256) void gs_unique_count_kernel(const u64 *table, const uint32_t *bitmap, int64_t n_slots,
                                                             uint32_t vbits, int32_t n_values, u64 *unique) {
    __shared__ unsigned int s_cnt[GS_UNIQ_LDS];
    const bool lds = n_values <= GS_UNIQ_LDS;
    if (lds) {
        for (int i = threadIdx.x; i < n_values; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
    }
    const u64 vmask = (1ULL << vbits) - 1;
    const int64_t n_words = (n_slots + 31) / 32;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t bits = bitmap[w];
        while (bits) {
            const int b = __builtin_ctz(bits);
            bits &= bits - 1;
            const u64 s = table[w * 32 + b];
            const int vi = (int)((s >> 1) & vmask) - 1;
            if (vi >= 0) {
                if (lds)
                    atomicAdd(&s_cnt[vi], 1u);
                else
                    atomicAdd(&unique[vi], 1ULL);
            }
        }
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_values; i += blockDim.x)
            if (s_cnt[i]) atomicAdd(&unique[i], (u64)s_cnt[i]);
    }
}

// seen bits of the slots -> compact bitmap (bit i = slot i); one wave per 64 slots
__global__ __launch_bounds__(256) void gs_bitmap_extract_kernel(const u64 *table, int64_t n_slots, uint32_t *bitmap) {
    const int lane = gs_lane();
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL; base < n_slots;
         base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = base + lane;
        const u64 m = __ballot(i < n_slots && (table[i] & 1ULL));
        if (lane == 0) bitmap[base >> 5] = (uint32_t)m;
        if (lane == 1 && base + 32 < n_slots) bitmap[(base >> 5) + 1] = (uint32_t)(m >> 32);
    }
}

__global__ __launch_bounds__(256) void gs_clear_seen_kernel(u64 *table, int64_t n_slots) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 s = table[i];
        if (s & 1ULL) table[i] = s & ~1ULL;
    }
}

// ---- the same three sweeps over the super-k-mer records: the compact bitmap holds one 32-bit word per record bucket
// (bit j = seen bit of offset j) behind the words of the table slots.  The sweeps stream the record lines with 16-byte
// loads of consecutive threads (a thread per line and 8 bytes of it costs one 64-byte request per lane: ~4x slower on a
// 9 GB table); thread 4 b holds words 0 and 1 of bucket b.
__global__ __launch_bounds__(256) void gs_rec_bitmap_extract_kernel(const u64 *rec, int64_t n_rec, uint32_t *bitmap_rec) {
    const int64_t n = n_rec * (GS_REC_WORDS / 2);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const gs_u64x2 v = reinterpret_cast<const gs_u64x2 *>(rec)[t];
        if ((t & 3) == 0) bitmap_rec[t >> 2] = (uint32_t)(v.x >> GS_REC_WIN_BITS);
    }
}

// 8 lanes per bucket: lane g holds word g of the line (one coalesced 64-byte request per bucket, only for buckets with a
// seen bit); lanes 2..7 look at the three value fields of their word
__global__ __launch_bounds__(256) void gs_rec_unique_count_kernel(const u64 *rec, const uint32_t *bitmap_rec, int64_t n_rec,
                                                                 int32_t n_values, u64 *unique) {
    __shared__ unsigned int s_cnt[GS_UNIQ_LDS];
    const bool lds = n_values <= GS_UNIQ_LDS;
    if (lds) {
        for (int i = threadIdx.x; i < n_values; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
    }
    const int g = (int)(threadIdx.x & 7);
    const int64_t groups = ((int64_t)gridDim.x * blockDim.x) >> 3;
    // eight lanes per bucket; four buckets per group and iteration with their loads in flight together (the loop is bound by
    // the latency of bitmap word -> record line, not by bytes)
    for (int64_t b0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; b0 < n_rec; b0 += 4 * groups) {
        uint32_t bits[4];
        u64 w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t bb = b0 + u * groups;
            bits[u] = bb < n_rec ? bitmap_rec[bb] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = bits[u] ? rec[(b0 + u * groups) * GS_REC_WORDS + g] : 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u64 w1 = __shfl(w[u], 1, 8);
            const uint32_t bt = bits[u] & (uint32_t)(w1 >> GS_REC_WIN_BITS);  // only offsets that hold a k-mer (a merged bitmap comes from other ranks)
            if (g >= 2 && bt) {
                // the three k-mers of a value word usually carry the same value index: one atomic for them together
                int v[3], c[3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int vi = (int)((w[u] >> (GS_REC_VAL_BITS * i)) & (GS_REC_MAX_VALUES - 1));
                    v[i] = (((bt >> (3 * (g - 2) + i)) & 1u) && vi < n_values) ? vi : -1;
                    c[i] = 1;
                }
                if (v[1] >= 0 && v[1] == v[0]) {
                    c[0]++;
                    v[1] = -1;
                }
                if (v[2] >= 0 && v[2] == v[0]) {
                    c[0]++;
                    v[2] = -1;
                } else if (v[2] >= 0 && v[2] == v[1]) {
                    c[1]++;
                    v[2] = -1;
                }
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    if (v[i] >= 0) {
                        if (lds)
                            atomicAdd(&s_cnt[v[i]], (unsigned int)c[i]);
                        else
                            atomicAdd(&unique[v[i]], (u64)c[i]);
                    }
                }
            }
        }
    }
    if (lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_values; i += blockDim.x)
            if (s_cnt[i]) atomicAdd(&unique[i], (u64)s_cnt[i]);
    }
}

__global__ __launch_bounds__(256) void gs_rec_clear_seen_kernel(u64 *rec, int64_t n_rec) {
    const u64 M47 = (1ULL << GS_REC_WIN_BITS) - 1;
    const int64_t n = n_rec * (GS_REC_WORDS / 2);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const gs_u64x2 v = reinterpret_cast<const gs_u64x2 *>(rec)[t];
        if ((t & 3) == 0 && (v.x >> GS_REC_WIN_BITS)) rec[2 * t] = v.x & M47;
    }
}

__global__ __launch_bounds__(256) void gs_bitmap_or_kernel(uint32_t *dst, const uint32_t *parts, int64_t n_words, int64_t n_parts) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = dst[w];
        for (int64_t p = 0; p < n_parts; p++) v |= parts[p * n_words + w];
        dst[w] = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// filter
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 gs_spread32(uint32_t v) {  // bit i -> bit 2i
    u64 x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}

// canonical k-mer in the REFERENCE's interleaved encoding (the value the Bloom hashes see)
__device__ __forceinline__ int64_t gs_canonical_java(uint32_t fhi, uint32_t flo, int k, uint32_t kmask) {
    const uint32_t rhi = __brev(fhi) >> (32 - k);
    const uint32_t rlo = __brev(flo) >> (32 - k);
    const u64 fwd = (gs_spread32(rhi) << 1) | gs_spread32(rlo);                     // first base in the top bits
    const u64 rev = (gs_spread32(fhi) << 1) | gs_spread32((flo ^ kmask) & kmask);   // complement, reversed
    return (int64_t)(fwd > rev ? fwd : rev);
}

// |v| mod d for the reference's Math.abs(v % bits) (XORKMerBloomFilter.java:57-59); d < 2^63
__device__ __forceinline__ u64 gs_absmod(int64_t v, u64 d, u64 magic, int shift) {
    const u64 n = v < 0 ? (u64)0 - (u64)v : (u64)v;
    if (shift == 0) return 0;  // d == 1
    const u64 t = __umul64hi(magic, n);
    const u64 q = (t + ((n - t) >> 1)) >> (shift - 1);
    return n - q * d;
}

__device__ __forceinline__ int64_t gs_murmur64(int64_t data_, int64_t base) {  // MurmurHash3DropIn.java:60-87
    const u64 data = (u64)data_;
    u64 hash = (u64)base;
    u64 kk = __builtin_bswap64(data);
    kk *= 0x87c37b91114253d5ULL;
    kk = (kk << 31) | (kk >> 33);
    kk *= 0x4cf5ad432745937fULL;
    hash ^= kk;
    hash = ((hash << 27) | (hash >> 37)) * 5 + 0x52dce729ULL;
    hash ^= 8;
    hash ^= hash >> 33;
    hash *= 0xff51afd7ed558ccdULL;
    hash ^= hash >> 33;
    hash *= 0xc4ceb9fe1a85ec53ULL;
    hash ^= hash >> 33;
    return (int64_t)(hash ^ data);
}

// bit `i`-th hash of `key` in the filter (XOR or Murmur kind); factors come from the block's LDS copy
__device__ __forceinline__ bool gs_filter_bit(const GsFilterParams &P, const uint32_t *words32, const int64_t *factors,
                                              int64_t key, int i) {
    const int64_t f = factors[i];
    const int64_t h = P.kind == GS_BLOOM_XOR ? (f ^ key) : gs_murmur64(key, f);
    const u64 idx = gs_absmod(h, P.bits, P.magic, P.magic_shift);
    return ((words32[(uint32_t)(idx >> 5)] >> (idx & 31)) & 1u) != 0;
}

#define GS_FILTER_MAX_HASHES 128

__global__ __launch_bounds__(GS_BLOCK) __attribute__((amdgpu_waves_per_eu(GS_WAVES, GS_WAVES))) void gs_filter_kernel(GsFilterParams P) {
    __shared__ u64 s_fkey[GS_BLOCK / 64][16];
    __shared__ u64 s_fcand[GS_BLOCK / 64][64];
    __shared__ int64_t s_factors[GS_FILTER_MAX_HASHES];
    for (int i = threadIdx.x; i < P.n_hashes && i < GS_FILTER_MAX_HASHES; i += blockDim.x) s_factors[i] = P.factors[i];
    __syncthreads();
    const int lane = gs_lane();
    const int wave_in_block = threadIdx.x >> 6;
    const int64_t wave_id = (int64_t)blockIdx.x * (GS_BLOCK / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (GS_BLOCK / 64);
    const int k = P.k;
    const uint32_t kmask = (1u << k) - 1u;
    // bit i of the filter = bit (i & 31) of 32-bit word i >> 5 (little endian): a dword load with a scalar base and a
    // 32-bit lane offset is the cheapest scattered load there is (XOR / Murmur filters are limited to 2^37 bits)
    const uint32_t *words32 = reinterpret_cast<const uint32_t *>(P.words);
    const int64_t n_reads = (P.skip != nullptr && *P.skip != 0) ? 0 : P.n_reads;
    for (int64_t r = wave_id; r < n_reads; r += n_waves) {
        const uint64_t *po = P.off + r * P.off_stride;
        const u64 off = po[0];
        const int L = (int)(po[1] - off);
        const int max = L - k + 1;
        const uint8_t *rd = P.seq + off;
        int accept = 0;
        if (max > 0) {
            const int pos_thr = P.min_pos_count > 0 ? P.min_pos_count : (int)((double)max * P.positive_ratio);
            const int need = pos_thr > 1 ? pos_thr : 1;
            int members = 0;
            u64 hi0, lo0, bad0;
            gs_load_word(rd, L, 0, lane, hi0, lo0, bad0);
            for (int round = 0; round * 64 < max && members < need; round++) {
                u64 hi1, lo1, bad1;
                gs_load_word(rd, L, round + 1, lane, hi1, lo1, bad1);
                const int p = 64 * round + lane;
                const uint32_t fhi = (uint32_t)gs_funnel(hi0, hi1, lane) & kmask;
                const uint32_t flo = (uint32_t)gs_funnel(lo0, lo1, lane) & kmask;
                const uint32_t wbad = (uint32_t)gs_funnel(bad0, bad1, lane) & kmask;
                const bool valid_lane = p < max && wbad == 0;
                const int64_t key = gs_canonical_java(fhi, flo, k, kmask);
                // A read that comes from the indexed genomes is accepted by its first few k-mers, so the first round
                // looks at every 8th position before it pays for all of them: an accepted read then costs ~60 filter
                // lines instead of ~250, a rejected one pays one extra round trip.  Every position is still examined
                // at most once and accept <=> #members >= need is unchanged.
                for (int pass = round == 0 ? 0 : 1; pass < 2 && members < need; pass++) {
                    bool alive = valid_lane && (round != 0 || (((lane & 7) == 0) == (pass == 0)));
                    if (P.kind == GS_BLOOM_BLOCKED) {
                        if (alive) {  // BlockedKMerBloomFilter.containsLong :181-199
                            const int64_t h0 = P.factors[0] ^ key;
                            const u64 start = gs_absmod(h0, P.bits, P.magic, P.magic_shift);
                            u64 uh = (u64)h0;
                            uh ^= (uh << 32) | (uh >> 32);
                            const int64_t sh = (int64_t)uh;
                            const u64 m1 = (1ULL << (sh & 63)) | (1ULL << ((sh >> 6) & 63));
                            const u64 m2 = (1ULL << ((sh >> 12) & 63)) | (1ULL << ((sh >> 18) & 63));
                            const u64 a = P.words[start];
                            const u64 b = P.words[start + 1 + (uh >> 60)];
                            alive = ((m1 & a) == m1) && ((m2 & b) == m2);
                        }
                    } else {
                        // AbstractKMerBloomFilter.containsLong :209-216.  accept <=> #members >= need, and a member is a
                        // k-mer whose n_hashes bits are all set; in which order the bits are looked at is free.  A
                        // scattered load costs the CU about the same whether 1 or 64 lanes take part, so the work is
                        // arranged to keep the lanes of every load busy:
                        //   1. hashes 0..2 per lane (drops ~88 % of the non-members of a half-full filter),
                        //   2. the survivors, 16 at a time, are spread over the wave: 4 lanes per survivor look at
                        //      hashes 3..6 in ONE load (keys travel through LDS),
                        //   3. what is still alive is a candidate: the lanes look at the next 8 hashes of one
                        //      candidate in one load (a false candidate rarely gets further), then at all the rest,
                        //      candidate after candidate until `need` members are confirmed.
                        // Measured on the 47 M-key index filter: 58 -> 12 load instructions and 153 -> ~180 filter
                        // lines per read, 59.8 -> 38.3 ms per 10 M reads; the kernel now runs at the fabric's
                        // random-line rate instead of at the CU's rate of (mostly empty) load instructions.
                        const int nh = P.n_hashes;
                        const int S = 3;  // per-lane steps
                        for (int i = 0; i < S && i < nh && __ballot(alive) != 0; i++) {
                            if (alive) alive = gs_filter_bit(P, words32, s_factors, key, i);
                        }
                        u64 rem = __ballot(alive);
                        int confirmed = 0;
                        if (nh <= S) {
                            confirmed = __popcll(rem);
                            rem = 0;
                        }
                        const int T = nh - S < 4 ? nh - S : 4;
                        int nc = 0, cdone = 0;
                        u64 *fkey = s_fkey[wave_in_block], *fcand = s_fcand[wave_in_block];
                        while (rem != 0 && members + confirmed < need) {
                            const int rank = __popcll(rem & ((1ULL << lane) - 1));
                            const bool in_batch = ((rem >> lane) & 1ULL) && rank < 16;
                            const u64 batch = __ballot(in_batch);
                            const int nb = __popcll(batch);
                            if (in_batch) fkey[rank] = (u64)key;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            const int j = lane >> 2, t = lane & 3;
                            const bool active = j < nb && t < T;
                            bool bit = true;
                            u64 jkey = 0;
                            if (j < nb) jkey = fkey[j];
                            if (active) bit = gs_filter_bit(P, words32, s_factors, (int64_t)jkey, S + t);
                            const u64 okm = __ballot(bit);
                            u64 pass4 = okm & (okm >> 1) & (okm >> 2) & (okm >> 3) & 0x1111111111111111ULL;
                            pass4 &= nb >= 16 ? ~0ULL : ((1ULL << (4 * nb)) - 1);  // groups beyond the batch are idle
                            if (((pass4 >> lane) & 1ULL) != 0) fcand[nc + __popcll(pass4 & ((1ULL << lane) - 1))] = jkey;
                            nc += __popcll(pass4);
                            rem &= ~batch;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            while (cdone < nc && members + confirmed < need) {
                                const int64_t ckey = (int64_t)fcand[cdone++];
                                bool all = true;
                                {
                                    const int i = S + T + lane;  // a first slice of 8 hashes
                                    const bool b = (lane < 8 && i < nh) ? gs_filter_bit(P, words32, s_factors, ckey, i) : true;
                                    all = __ballot(b) == ~0ULL;
                                }
                                for (int base = S + T + 8; base < nh && all; base += 64) {
                                    const int i = base + lane;
                                    const bool b = i < nh ? gs_filter_bit(P, words32, s_factors, ckey, i) : true;
                                    all = __ballot(b) == ~0ULL;
                                }
                                confirmed += all ? 1 : 0;
                            }
                            __builtin_amdgcn_wave_barrier();  // fkey / fcand are rewritten by the next batch
                        }
                        members += confirmed;
                        alive = false;  // already counted
                    }
                    members += __popcll(__ballot(alive));
                }
                hi0 = hi1;
                lo0 = lo1;
                bad0 = bad1;
            }
            accept = members >= need;
        }
        if (lane == 0) P.accept[r] = (uint8_t)accept;
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers (called from gs_api.cpp)
// ---------------------------------------------------------------------------------------------------
static size_t gs_stats_lds_bytes(int n_values) {
    if (n_values <= GS_NV_LDS) return (size_t)n_values * ((GS_N_SUMS + 1 + GS_N_DCOLS) * 8 + 3 * 4);
    return n_values <= GS_NV_TREE_LDS ? (size_t)n_values * 3 * 4 : 0;  // the tree alone
}

// GS_FORCE_GLOBAL_STATS=1 (developer knob): the kernels for stores with more values than the LDS counters hold, on any store
static bool gs_force_global_stats() {
    static const bool on = [] {
        const char *e = getenv("GS_FORCE_GLOBAL_STATS");
        return e != nullptr && atoi(e) != 0;
    }();
    return on;
}

template <bool FROM_NODES, int KC, bool WIDE, bool STRIPED, int CTX>
static void gs_launch_match_t(const GsMatchParams *P, int grid, size_t lds, bool lds_stats, hipStream_t stream) {
    if (lds_stats)
        hipLaunchKernelGGL((gs_match_kernel<true, FROM_NODES, KC, WIDE, STRIPED, CTX>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
    else
        hipLaunchKernelGGL((gs_match_kernel<false, FROM_NODES, KC, WIDE, STRIPED, CTX>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
}

// k = 31 (the reference's default and maximum) runs kernels with k folded in at compile time; whether the gate is keyed by
// minimizer + context (big stores, GsDbDev::mgate_ctx) is folded in as well -- the branch alone costs the common kernels 3 % --
// except in the 128-path variants, which ask the store
extern "C" hipError_t gs_launch_match(const GsMatchParams *P, int grid, hipStream_t stream) {
    const size_t lds = gs_stats_lds_bytes(P->db.n_values);
    const bool lds_stats = P->db.n_values <= GS_NV_LDS && !gs_force_global_stats();
    const bool ctx = P->db.mgate_ctx != 0;
    if (P->db.n_parts > 1) {  // striped store (always probed locally: nodes == nullptr)
        if (P->nodes != nullptr) return hipErrorInvalidValue;
        if (P->max_paths > 64)
            gs_launch_match_t<false, 0, true, true, 2>(P, grid, lds, lds_stats, stream);
        else if (P->db.k == 31)
            ctx ? gs_launch_match_t<false, 31, false, true, 1>(P, grid, lds, lds_stats, stream)
                : gs_launch_match_t<false, 31, false, true, 0>(P, grid, lds, lds_stats, stream);
        else
            ctx ? gs_launch_match_t<false, 0, false, true, 1>(P, grid, lds, lds_stats, stream)
                : gs_launch_match_t<false, 0, false, true, 0>(P, grid, lds, lds_stats, stream);
        return hipGetLastError();
    }
    if (P->nodes != nullptr) {  // DB-partitioned mode: the nodes come from the owners, nothing is probed here
        if (P->max_paths > 64)
            gs_launch_match_t<true, 0, true, false, 0>(P, grid, lds, lds_stats, stream);
        else
            gs_launch_match_t<true, 0, false, false, 0>(P, grid, lds, lds_stats, stream);
        return hipGetLastError();
    }
    if (P->max_paths > 64)  // two candidate paths per lane (the reference allows up to 128, C/GSConfigKey.java:350)
        gs_launch_match_t<false, 0, true, false, 2>(P, grid, lds, lds_stats, stream);
    else if (P->db.k == 31)
        ctx ? gs_launch_match_t<false, 31, false, false, 1>(P, grid, lds, lds_stats, stream)
            : gs_launch_match_t<false, 31, false, false, 0>(P, grid, lds, lds_stats, stream);
    else
        ctx ? gs_launch_match_t<false, 0, false, false, 1>(P, grid, lds, lds_stats, stream)
            : gs_launch_match_t<false, 0, false, false, 0>(P, grid, lds, lds_stats, stream);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_match_long(const GsMatchParams *P, int grid, int32_t *scratch, uint32_t *serial,
                                           hipStream_t stream) {
    const size_t lds = gs_stats_lds_bytes(P->db.n_values);
    const bool lds_stats = P->db.n_values <= GS_NV_LDS && !gs_force_global_stats();
    if (P->db.n_parts > 1) {
        if (P->nodes != nullptr) return hipErrorInvalidValue;
        if (P->max_paths > 64) {
            if (lds_stats)
                hipLaunchKernelGGL((gs_match_long_kernel<true, false, true, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
            else
                hipLaunchKernelGGL((gs_match_long_kernel<false, false, true, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
        } else {
            if (lds_stats)
                hipLaunchKernelGGL((gs_match_long_kernel<true, false, false, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
            else
                hipLaunchKernelGGL((gs_match_long_kernel<false, false, false, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
        }
        return hipGetLastError();
    }
    if (P->max_paths > 64) {
        if (P->nodes == nullptr) {
            if (lds_stats)
                hipLaunchKernelGGL((gs_match_long_kernel<true, false, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
            else
                hipLaunchKernelGGL((gs_match_long_kernel<false, false, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
        } else {
            if (lds_stats)
                hipLaunchKernelGGL((gs_match_long_kernel<true, true, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
            else
                hipLaunchKernelGGL((gs_match_long_kernel<false, true, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
        }
        return hipGetLastError();
    }
    if (P->nodes == nullptr) {
        if (P->db.k == 31) {  // (as gs_match_kernel: k folded in at compile time)
            if (lds_stats)
                hipLaunchKernelGGL((gs_match_long_kernel<true, false, false, false, 31>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
            else
                hipLaunchKernelGGL((gs_match_long_kernel<false, false, false, false, 31>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
        } else if (lds_stats)
            hipLaunchKernelGGL((gs_match_long_kernel<true, false>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
        else
            hipLaunchKernelGGL((gs_match_long_kernel<false, false>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
    } else {
        if (lds_stats)
            hipLaunchKernelGGL((gs_match_long_kernel<true, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
        else
            hipLaunchKernelGGL((gs_match_long_kernel<false, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P, scratch, serial);
    }
    return hipGetLastError();
}

// the copies 1 .. copies - 1 of the run's counters into copy 0 (sums and double sums added, max keys by maximum), the copies zeroed:
// one launch (one per copy and array -- 45 for 16 copies -- cost 0.2 ms per job)
__global__ __launch_bounds__(256) void gs_fold_stats_kernel(long long *sums, unsigned long long *maxk, double *dsums, long long nv, int copies) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n_s = nv * GS_N_SUMS, n_d = nv * GS_N_DCOLS;
    if (i < n_s) {
        long long a = sums[i];
        for (int c = 1; c < copies; c++) {
            a += sums[(long long)c * n_s + i];
            sums[(long long)c * n_s + i] = 0;
        }
        sums[i] = a;
    }
    if (i < nv) {
        unsigned long long m = maxk[i];
        for (int c = 1; c < copies; c++) {
            const unsigned long long x = maxk[(long long)c * nv + i];
            m = x > m ? x : m;
            maxk[(long long)c * nv + i] = 0;
        }
        maxk[i] = m;
    }
    if (i < n_d) {
        double a = dsums[i];
        for (int c = 1; c < copies; c++) {
            a += dsums[(long long)c * n_d + i];
            dsums[(long long)c * n_d + i] = 0.0;
        }
        dsums[i] = a;
    }
}
extern "C" hipError_t gs_launch_fold_stats(long long *sums, unsigned long long *maxk, double *dsums, long long nv, int copies, hipStream_t stream) {
    const long long n = nv * (GS_N_SUMS > GS_N_DCOLS ? GS_N_SUMS : GS_N_DCOLS);
    hipLaunchKernelGGL(gs_fold_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sums, maxk, dsums, nv, copies);
    return hipGetLastError();
}

// which of the wide kernels (bit 0: three sub-rounds, reads of 129 .. 192 positions; bit 1: four, 193 .. 256) serve this store and run
extern "C" int gs_match_wide_mask(const GsMatchParams *P) {
    if (P->nodes != nullptr || P->db.n_parts > 1 || P->max_paths > 64) return 0;
    return 3;
}
// ns = 3 / 4: the reads of that queue -- or all reads of the batch when P->long_list is null -- each in one trip; grid: the device's CUs
extern "C" hipError_t gs_launch_match_wide(const GsMatchParams *P, int ns, int n_cu, hipStream_t stream) {
    if ((ns != 3 && ns != 4) || ((gs_match_wide_mask(P) >> (ns - 3)) & 1) == 0) return hipErrorNotSupported;
    const size_t lds = gs_stats_lds_bytes(P->db.n_values);
    const bool lds_stats = P->db.n_values <= GS_NV_LDS && !gs_force_global_stats();
    const int v = (lds_stats ? 4 : 0) | (ns == 4 ? 2 : 0) | (P->db.k == 31 ? 1 : 0);
    void (*kern)(GsMatchParams) = nullptr;
    switch (v) {
        case 0: kern = gs_match_wide_kernel<false, 3, 0>; break;
        case 1: kern = gs_match_wide_kernel<false, 3, 31>; break;
        case 2: kern = gs_match_wide_kernel<false, 4, 0>; break;
        case 3: kern = gs_match_wide_kernel<false, 4, 31>; break;
        case 4: kern = gs_match_wide_kernel<true, 3, 0>; break;
        case 5: kern = gs_match_wide_kernel<true, 3, 31>; break;
        case 6: kern = gs_match_wide_kernel<true, 4, 0>; break;
        default: kern = gs_match_wide_kernel<true, 4, 31>; break;
    }
    static int occ_of[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // resident workgroups per CU (dynamic LDS aside: small against the static part)
    int occ = occ_of[v];
    if (occ == 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, GS_BLOCK, lds) != hipSuccess || occ < 1) occ = 1;
        occ_of[v] = occ;
    }
    hipLaunchKernelGGL(kern, dim3(n_cu * occ), dim3(GS_BLOCK), lds, stream, *P);
    return hipGetLastError();
}

// reads of GS_HUGE_MIN positions and more that the match kernel handed over: chunks over the whole device, then one wave per read
extern "C" hipError_t gs_launch_match_huge(const GsMatchParams *P, int grid, hipStream_t stream) {
    if (P->huge_count == nullptr || P->nodes != nullptr) return hipSuccess;
    const size_t lds = gs_stats_lds_bytes(P->db.n_values);
    const bool lds_stats = P->db.n_values <= GS_NV_LDS && !gs_force_global_stats();
    const bool striped = P->db.n_parts > 1;
    if (lds_stats) {
        if (striped)
            hipLaunchKernelGGL((gs_match_huge_kernel<true, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
        else if (P->db.k == 31)
            hipLaunchKernelGGL((gs_match_huge_kernel<true, false, 31>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
        else
            hipLaunchKernelGGL((gs_match_huge_kernel<true, false>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
        hipLaunchKernelGGL((gs_match_huge_finish_kernel<true>), dim3(GS_HUGE_SLOTS / (GS_BLOCK / 64)), dim3(GS_BLOCK), lds, stream, *P);
    } else {
        if (striped)
            hipLaunchKernelGGL((gs_match_huge_kernel<false, true>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
        else if (P->db.k == 31)
            hipLaunchKernelGGL((gs_match_huge_kernel<false, false, 31>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
        else
            hipLaunchKernelGGL((gs_match_huge_kernel<false, false>), dim3(grid), dim3(GS_BLOCK), lds, stream, *P);
        hipLaunchKernelGGL((gs_match_huge_finish_kernel<false>), dim3(GS_HUGE_SLOTS / (GS_BLOCK / 64)), dim3(GS_BLOCK), lds, stream, *P);
    }
    return hipGetLastError();
}

extern "C" int gs_match_occupancy(int n_values) {
    int n = 0;
    hipError_t e = n_values <= GS_NV_LDS
                       ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gs_match_kernel<true, false, 31>, GS_BLOCK, gs_stats_lds_bytes(n_values))
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gs_match_kernel<false, false, 31>, GS_BLOCK, gs_stats_lds_bytes(n_values));
    return e == hipSuccess ? n : 0;
}

// workgroups of the long-read kernel a CU holds at once (the plain variant stands for all of them)
extern "C" int gs_match_long_occupancy(int n_values) {
    int n = 0;
    hipError_t e = n_values <= GS_NV_LDS
                       ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gs_match_long_kernel<true, false, true, true>, GS_BLOCK, gs_stats_lds_bytes(n_values))
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gs_match_long_kernel<false, false, true, true>, GS_BLOCK, gs_stats_lds_bytes(n_values));
    return e == hipSuccess ? n : 0;
}

extern "C" int gs_filter_occupancy() {
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gs_filter_kernel, GS_BLOCK, 0);
    return e == hipSuccess ? n : 0;
}

// the compact bitmap: (n_slots + 31) / 32 words for the table slots, then one word per record bucket (rec may be NULL)
extern "C" hipError_t gs_launch_unique_count(const u64 *table, const uint32_t *bitmap, int64_t n_slots, uint32_t vbits,
                                              int32_t n_values, u64 *unique, const u64 *rec, int64_t n_rec, hipStream_t stream) {
    int64_t n_words = (n_slots + 31) / 32;
    int grid = (int)((n_words + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gs_unique_count_kernel, dim3(grid), dim3(256), 0, stream, table, bitmap, n_slots, vbits, n_values, unique);
    if (rec != nullptr && n_rec > 0) {
        grid = (int)std::min<int64_t>((n_rec * 8 + 255) / 256, 2048);
        hipLaunchKernelGGL(gs_rec_unique_count_kernel, dim3(grid), dim3(256), 0, stream, rec, bitmap + n_words, n_rec, n_values, unique);
    }
    return hipGetLastError();
}

// one stripe of a striped store: `rec` = the stripe's lines, `bitmap_rec` = the bitmap words of ITS buckets
extern "C" hipError_t gs_launch_rec_unique_count(const u64 *rec, const uint32_t *bitmap_rec, int64_t n_rec, int32_t n_values,
                                                  u64 *unique, hipStream_t stream) {
    if (n_rec <= 0) return hipSuccess;
    const int grid = (int)std::min<int64_t>((n_rec * 8 + 255) / 256, 2048);
    hipLaunchKernelGGL(gs_rec_unique_count_kernel, dim3(grid), dim3(256), 0, stream, rec, bitmap_rec, n_rec, n_values, unique);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_bitmap_extract(const u64 *table, int64_t n_slots, uint32_t *bitmap, const u64 *rec, int64_t n_rec,
                                                hipStream_t stream) {
    hipLaunchKernelGGL(gs_bitmap_extract_kernel, dim3(4096), dim3(256), 0, stream, table, n_slots, bitmap);
    if (rec != nullptr && n_rec > 0)
        hipLaunchKernelGGL(gs_rec_bitmap_extract_kernel, dim3((int)std::min<int64_t>((n_rec * 4 + 255) / 256, 8192)), dim3(256), 0, stream,
                           rec, n_rec, bitmap + (n_slots + 31) / 32);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_clear_seen(u64 *table, int64_t n_slots, u64 *rec, int64_t n_rec, hipStream_t stream) {
    hipLaunchKernelGGL(gs_clear_seen_kernel, dim3(4096), dim3(256), 0, stream, table, n_slots);
    if (rec != nullptr && n_rec > 0)
        hipLaunchKernelGGL(gs_rec_clear_seen_kernel, dim3((int)std::min<int64_t>((n_rec * 4 + 255) / 256, 8192)), dim3(256), 0, stream, rec, n_rec);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_bitmap_or(uint32_t *dst, const uint32_t *parts, int64_t n_words, int64_t n_parts,
                                           hipStream_t stream) {
    int grid = (int)((n_words + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gs_bitmap_or_kernel, dim3(grid), dim3(256), 0, stream, dst, parts, n_words, n_parts);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_encode(const GsEncodeParams *P, int grid, hipStream_t stream) {
    if (P->k == 31)
        hipLaunchKernelGGL(gs_encode_kernel<31>, dim3(grid), dim3(GS_BLOCK), 0, stream, *P);
    else
        hipLaunchKernelGGL(gs_encode_kernel<0>, dim3(grid), dim3(GS_BLOCK), 0, stream, *P);
    return hipGetLastError();
}

static int gs_stream_grid(int64_t n) {
    int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
    return grid < 1 ? 1 : grid;
}

extern "C" hipError_t gs_launch_route_count(const u64 *keys, int64_t n, int n_parts, u64 *counts, hipStream_t stream) {
    hipLaunchKernelGGL(gs_route_count_kernel, dim3(gs_stream_grid(n)), dim3(256), 0, stream, keys, n, n_parts, counts);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_route_scatter(const u64 *keys, int64_t n, int n_parts, u64 *cursors, u64 *send_keys,
                                               uint32_t *idx, int32_t *nodes, hipStream_t stream) {
    int grid = (int)std::min<int64_t>((n + 256 * GS_ROUTE_T - 1) / (256 * GS_ROUTE_T), 256 * 8);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gs_route_scatter_kernel, dim3(grid), dim3(256), 0, stream, keys, n, n_parts, cursors, send_keys, idx,
                       nodes);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_unroute(const u64 *keys, const uint32_t *idx, const int32_t *back, int64_t n_routed,
                                         int32_t *nodes, int64_t n_keys, hipStream_t stream) {
    if (keys != nullptr)  // (NULL: gs_route_scatter_kernel has filled in the unrouted positions)
        hipLaunchKernelGGL(gs_unroute_kernel, dim3(gs_stream_grid(n_keys)), dim3(256), 0, stream, keys, idx, back, n_routed,
                           nodes, n_keys, 0);
    if (n_routed > 0)
        hipLaunchKernelGGL(gs_unroute_kernel, dim3(gs_stream_grid(n_routed)), dim3(256), 0, stream, keys, idx, back,
                           n_routed, nodes, n_keys, 1);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_probe_keys(const GsDbDev *db, const u64 *keys, int64_t n, int32_t *nodes, int count_unique,
                                            hipStream_t stream) {
    int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 24);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gs_probe_keys_kernel, dim3(grid), dim3(256), 0, stream, *db, keys, n, nodes, count_unique);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_segments(const GsSegParams *P, int write, int grid, hipStream_t stream) {
    if (P->db.n_parts > 1) {
        if (write)
            hipLaunchKernelGGL((gs_segments_kernel<true, true>), dim3(grid), dim3(GS_BLOCK), 0, stream, *P);
        else
            hipLaunchKernelGGL((gs_segments_kernel<false, true>), dim3(grid), dim3(GS_BLOCK), 0, stream, *P);
    } else if (write)
        hipLaunchKernelGGL((gs_segments_kernel<true, false>), dim3(grid), dim3(GS_BLOCK), 0, stream, *P);
    else
        hipLaunchKernelGGL((gs_segments_kernel<false, false>), dim3(grid), dim3(GS_BLOCK), 0, stream, *P);
    return hipGetLastError();
}

extern "C" hipError_t gs_launch_filter(const GsFilterParams *P, int grid, hipStream_t stream) {
    hipLaunchKernelGGL(gs_filter_kernel, dim3(grid), dim3(GS_BLOCK), 0, stream, *P);
    return hipGetLastError();
}
