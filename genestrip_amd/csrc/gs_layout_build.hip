// gs_layout_build.hip -- the device layout of a store (gs_layout.h: super-k-mer records, overflow table, minimizer gate)
// built ON the device from the (k-mer, value index) arrays.  The host builder (gs_api.cpp: db_create_impl) stays the
// reference for the layout rules and serves striped / partition stores and stores without records; this file does the same
// steps with one thread per key / minimizer / window:
//   per key        planes, both strand views' minimizers, eligible for a record or not           gs_lb_perkey_kernel
//   sort           eligible entries by (minimizer hash, offset, planes)                          rocPRIM radix sort
//   per minimizer  greedy clustering of its entries into windows, the two fullest are kept       gs_lb_cluster_kernel
//   cuckoo         bidding rounds (atomic min: the lowest window number wins or keeps a bucket)    gs_lb_bid / gs_lb_resolve_kernel
//   lines          window planes + valid bits, values OR-ed in per entry, `more` bits             gs_lb_lines*/gs_lb_more_kernel
//   table          what found no record: sorted by home bucket, slot = rank, leftovers pass by pass gs_lb_tplace_kernel
//   gate           two bits per minimizer                                                          gs_lb_gate_kernel
// Every step is independent of timing (total sort orders, atomic min / max / or only): two builds from the same arrays give the
// same bytes, which the merge of runs on separately built replicas relies on (the unique-k-mer bitmap is indexed by slot).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gs_layout.h"

typedef unsigned long long u64;

#define GS_LB_EMPTY 0xffffffffu
#define GS_LB_MAXWIN 8  // windows a minimizer's entries are clustered into; what fits none of them goes to the table

enum { GS_LB_N_E = 0, GS_LB_N_T, GS_LB_N_M, GS_LB_N_H, GS_LB_IN_REC, GS_LB_OVERFLOW, GS_LB_MAX_DISP, GS_LB_N_WIN, GS_LB_N_CTX, GS_LB_N_HINT, GS_LB_COUNTERS };

// every lane of the wave calls this; returns the slot of the lanes with `have` in a list that grows by one atomic per wave
__device__ __forceinline__ u64 gs_lb_append(bool have, u64 *counter, int lane) {
    const u64 m = __ballot(have);
    if (m == 0) return 0;
    const int leader = __builtin_ctzll(m);
    u64 base = 0;
    if (lane == leader) base = atomicAdd(counter, (u64)__popcll(m));
    base = __shfl(base, leader);
    return base + (u64)__popcll(m & ((1ULL << lane) - 1));
}

// ---- per key (db_create_impl "keys"): reachable (the reference only ever queries max(fwd, revcomp)), node exists, both strand
// views' minimizer; one view = one record entry, two views = a table key reachable from both minimizers' buckets
__global__ __launch_bounds__(256) void gs_lb_perkey_kernel(const int64_t *kmers, const int32_t *vidx, int64_t n, int k, const int32_t *parent,
                                                           uint32_t *e_gh, uint32_t *e_ohi, uint32_t *e_olo, uint32_t *e_vj, u64 *e_sort,
                                                           u64 *e_sort2, u64 *t_key, int32_t *t_val, uint32_t *m_gh, uint32_t *h_gh, uint32_t *h_ctx,
                                                           u64 *cnt) {
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t kmask = (1u << k) - 1u;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL; i0 < n; i0 += stride) {
        const int64_t i = i0 + lane;
        bool ok = false, same = false;
        uint32_t gh1 = 0, gh2 = 0, ohi1 = 0, olo1 = 0, fhi = 0, flo = 0, ck1 = 0, ck2 = 0;
        int j1 = 0;
        int32_t v = 0;
        if (i < n) {
            const u64 x = (u64)kmers[i];
            v = vidx[i];
            u64 rc = 0;
            for (int b = 0; b < k; b++) {  // reference encoding (first base in the top bits) -> planes, and the reverse complement
                const uint32_t c = (uint32_t)(x >> (2 * (k - 1 - b))) & 3u;
                fhi |= (c >> 1) << b;
                flo |= (c & 1u) << b;
                rc |= (u64)(c ^ 1u) << (2 * b);
            }
            ok = x >= rc && parent[v] != -2;
        }
        if (ok) {
            const uint32_t rhi = gs_brev32(fhi) >> (32 - k), rlo = (gs_brev32(flo) >> (32 - k)) ^ kmask;
            uint32_t best1 = 0xffffffffu, best2 = 0xffffffffu;
            const int last = k - GS_MIN_L;
            for (int d = 0; d <= last; d++) {  // one pass over the order hashes serves both views (mirrored offsets)
                const uint32_t h = gs_lmer_hash((fhi >> d) & 0x7fffu, (flo >> d) & 0x7fffu);
                const uint32_t x1 = gs_lmer_rank(h, (uint32_t)d), x2 = gs_lmer_rank(h, (uint32_t)(last - d));
                best1 = x1 < best1 ? x1 : best1;
                best2 = x2 < best2 ? x2 : best2;
            }
            uint32_t ohi2, olo2;
            int j2;
            gs_min_oriented(fhi, flo, rhi, rlo, k, (int)(best1 & 0xffu), gh1, ohi1, olo1, j1);
            gs_min_oriented(rhi, rlo, fhi, flo, k, (int)(best2 & 0xffu), gh2, ohi2, olo2, j2);
            same = gh1 == gh2 && j1 == j2 && ohi1 == ohi2 && olo1 == olo2;
            if (h_ctx != nullptr) {  // the context keys of both strand views (gs_gate_ctx_key), k >= GS_CTX_MIN_K
                ck1 = gs_gate_ctx_key(gh1, ohi1, olo1, j1, k);
                ck2 = gs_gate_ctx_key(gh2, ohi2, olo2, j2, k);
            }
        }
        const u64 he = gs_lb_append(ok, cnt + GS_LB_N_H, lane);
        if (ok) h_gh[he] = gh1;
        const u64 he2 = gs_lb_append(ok && gh2 != gh1, cnt + GS_LB_N_H, lane);
        if (ok && gh2 != gh1) h_gh[he2] = gh2;
        if (h_ctx != nullptr) {
            const u64 ce = gs_lb_append(ok, cnt + GS_LB_N_CTX, lane);
            if (ok) h_ctx[ce] = ck1;
            const u64 ce2 = gs_lb_append(ok && ck2 != ck1, cnt + GS_LB_N_CTX, lane);
            if (ok && ck2 != ck1) h_ctx[ce2] = ck2;
        }
        const u64 ee = gs_lb_append(ok && same, cnt + GS_LB_N_E, lane);
        if (ok && same) {
            e_gh[ee] = gh1;
            e_ohi[ee] = ohi1;
            e_olo[ee] = olo1;
            e_vj[ee] = ((uint32_t)v << 5) | (uint32_t)j1;
            e_sort[ee] = ((u64)gh1 << 32) | ((u64)j1 << 27) | (u64)(ohi1 >> 4);
            e_sort2[ee] = ((u64)(ohi1 & 15u) << 32) | (u64)olo1;  // the rest of (ohi, olo): together a total order
        }
        const u64 te = gs_lb_append(ok && !same, cnt + GS_LB_N_T, lane);
        const u64 me = gs_lb_append(ok && !same, cnt + GS_LB_N_M, lane);
        const u64 me2 = gs_lb_append(ok && !same, cnt + GS_LB_N_M, lane);
        if (ok && !same) {
            uint32_t phi, plo;
            gs_rep_planes(fhi, flo, k, kmask, phi, plo);
            t_key[te] = gs_mix_planes(phi, plo);
            t_val[te] = v;
            m_gh[me] = gh1;
            m_gh[me2] = gh2;
        }
    }
}

__global__ __launch_bounds__(256) void gs_lb_gather_kernel(const uint32_t *perm, int64_t n, const uint32_t *a0, const uint32_t *a1, const uint32_t *a2,
                                                           const uint32_t *a3, uint32_t *b0, uint32_t *b1, uint32_t *b2, uint32_t *b3) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t p = perm[i];
        b0[i] = a0[p];
        b1[i] = a1[p];
        b2[i] = a2[p];
        b3[i] = a3[p];
    }
}

__global__ __launch_bounds__(256) void gs_lb_iota_kernel(uint32_t *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void gs_lb_dec_kernel(uint32_t *p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] -= 1u;
}

// head[i] = 1 where a new minimizer starts in the sorted entries
__global__ __launch_bounds__(256) void gs_lb_heads_kernel(const uint32_t *gh, int64_t n, uint32_t *head) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        head[i] = (i == 0 || gh[i] != gh[i - 1]) ? 1u : 0u;
}

// group[i] (inclusive scan of head, minus one) -> start of every group; g_start[n_groups] = n
__global__ __launch_bounds__(256) void gs_lb_starts_kernel(const uint32_t *head, const uint32_t *group, int64_t n, int64_t n_groups, uint32_t *g_start) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (head[i]) g_start[group[i]] = (uint32_t)i;
    if (blockIdx.x == 0 && threadIdx.x == 0) g_start[n_groups] = (uint32_t)n;
}

// ---- one thread per minimizer: its entries (sorted by offset) join the first window that agrees on every base both know and
// has their offset free, as in db_create_impl's cluster_chunk; of the windows the two fullest get a number (2 g, 2 g + 1)
__global__ __launch_bounds__(128) void gs_lb_cluster_kernel(const uint32_t *gh, const uint32_t *ohi, const uint32_t *olo, const uint32_t *vj,
                                                            const uint32_t *g_start, int64_t n_groups, int k, uint8_t *assign, u64 *w_hi,
                                                            u64 *w_lo, uint32_t *w_valid, uint32_t *w_gh, u64 *cnt) {
    const u64 kmask = (1ULL << k) - 1;
    int64_t n_win_mine = 0;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t s = g_start[g], t = g_start[g + 1];
        u64 whi[GS_LB_MAXWIN], wlo[GS_LB_MAXWIN], known[GS_LB_MAXWIN];
        uint32_t valid[GS_LB_MAXWIN];
        int nw = 0;
        for (uint32_t e = s; e < t; e++) {
            const int j = (int)(vj[e] & 31u);
            const u64 eh = (u64)ohi[e] << j, el = (u64)olo[e] << j, km = kmask << j;
            int w = 0;
            for (; w < nw; w++)
                if (!((valid[w] >> j) & 1u) && ((whi[w] ^ eh) & known[w] & km) == 0 && ((wlo[w] ^ el) & known[w] & km) == 0) break;
            if (w == nw) {
                if (nw == GS_LB_MAXWIN) {
                    assign[e] = 0xff;
                    continue;
                }
                whi[nw] = wlo[nw] = known[nw] = 0;
                valid[nw] = 0;
                nw++;
            }
            whi[w] |= eh;
            wlo[w] |= el;
            known[w] |= km;
            valid[w] |= 1u << j;
            assign[e] = (uint8_t)w;
        }
        int b0 = 0, b1 = -1;
        for (int w = 1; w < nw; w++)
            if (__popc(valid[w]) > __popc(valid[b0])) b0 = w;
        for (int w = 0; w < nw; w++)
            if (w != b0 && (b1 < 0 || __popc(valid[w]) > __popc(valid[b1]))) b1 = w;
        w_hi[2 * g] = whi[b0];
        w_lo[2 * g] = wlo[b0];
        w_valid[2 * g] = valid[b0];
        w_gh[2 * g] = gh[s];
        w_gh[2 * g + 1] = gh[s];
        if (b1 >= 0) {
            w_hi[2 * g + 1] = whi[b1];
            w_lo[2 * g + 1] = wlo[b1];
            w_valid[2 * g + 1] = valid[b1];
        } else {
            w_hi[2 * g + 1] = 0;
            w_lo[2 * g + 1] = 0;
            w_valid[2 * g + 1] = 0;
        }
        n_win_mine += 1 + (b1 >= 0);
        for (uint32_t e = s; e < t; e++) {
            const int a = assign[e];
            assign[e] = a == b0 ? 0 : (a == b1 ? 1 : 0xff);
        }
    }
    if (n_win_mine) atomicAdd(cnt + GS_LB_N_WIN, (u64)n_win_mine);
}

// ---- cuckoo placement of the windows (number w, minimizer hash w_gh[w], empty if w_valid[w] == 0), the same whatever the
// timing: in every round the windows without a bucket bid for one of their two buckets (alternating), a bucket goes to the
// LOWEST number among its holder and the bidders (atomic min), a holder that lost its bucket bids for its other one in the next
// round.  Bucket numbers only ever decrease, so the rounds end; a window that loses both its buckets to lower numbers again
// and again stays without one (its k-mers go to the table).  Windows are numbered by minimizer hash, the fuller of a
// minimizer's two windows first.
//   state[w]: bit 31 = holds a bucket, bit 0 = the choice (0 / 1) it holds or bids for next
__global__ __launch_bounds__(256) void gs_lb_bid_kernel(const uint32_t *w_valid, const uint32_t *w_gh, const uint32_t *state, int64_t n_w,
                                                        uint32_t rec_bits, uint32_t *slot) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_w; w += (int64_t)gridDim.x * blockDim.x) {
        if (!w_valid[w] || (state[w] & 0x80000000u)) continue;
        atomicMin(&slot[gs_rec_bucket(w_gh[w], rec_bits, (int)(state[w] & 1u))], (uint32_t)w);
    }
}

__global__ __launch_bounds__(256) void gs_lb_resolve_kernel(const uint32_t *w_valid, const uint32_t *w_gh, uint32_t *state, int64_t n_w, uint32_t rec_bits,
                                                            const uint32_t *slot, u64 *changes) {
    u64 mine = 0;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_w; w += (int64_t)gridDim.x * blockDim.x) {
        if (!w_valid[w]) continue;
        const uint32_t st = state[w];
        const bool mine_now = slot[gs_rec_bucket(w_gh[w], rec_bits, (int)(st & 1u))] == (uint32_t)w;
        if (st & 0x80000000u) {
            if (!mine_now) {  // a lower number took the bucket: the other one next
                state[w] = (st & 1u) ^ 1u;
                mine++;
            }
        } else if (mine_now) {
            state[w] = st | 0x80000000u;
            mine++;
        } else
            state[w] = (st & 1u) ^ 1u;
    }
    if (mine) atomicAdd(changes, mine);
}

// After the bidding has settled, the windows without a bucket look one step further (the first step of a cuckoo eviction walk,
// without the walk's randomness): if the holder h of one of w's buckets could move to ITS other bucket because that one is
// empty, w claims the empty bucket (atomic min over the claimants: the lowest number wins it), and the winners carry the move
// out -- h into the empty bucket, w into h's old one.  A holder is asked by one winner at most (all who ask it aim at the same
// empty bucket) and no move touches a bucket of another move, so a round is independent of timing.  Sibling windows are not asked
// to move (a minimizer's two windows stay where they are).
__device__ __forceinline__ bool gs_lb_plan(uint32_t w, const uint32_t *w_gh, uint32_t rec_bits, const uint32_t *slot, int &c, uint32_t &b, uint32_t &h,
                                           uint32_t &alt) {
    const uint32_t gh = w_gh[w];
    for (c = 0; c < 2; c++) {
        b = gs_rec_bucket(gh, rec_bits, c);
        h = slot[b];
        if (h == GS_LB_EMPTY) {  // (free after all: take it as it is)
            alt = b;
            return true;
        }
        const uint32_t hg = w_gh[h];
        if (hg == gh) continue;
        const uint32_t h0 = gs_rec_bucket(hg, rec_bits, 0), h1 = gs_rec_bucket(hg, rec_bits, 1);
        if (h0 == h1) continue;
        alt = h0 == b ? h1 : h0;
        if (slot[alt] == GS_LB_EMPTY) return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void gs_lb_claim_kernel(const uint32_t *w_valid, const uint32_t *w_gh, const uint32_t *state, int64_t n_w,
                                                          uint32_t rec_bits, const uint32_t *slot, uint32_t *claim) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_w; w += (int64_t)gridDim.x * blockDim.x) {
        if (!w_valid[w] || (state[w] & 0x80000000u)) continue;
        int c;
        uint32_t b, h, alt;
        if (gs_lb_plan((uint32_t)w, w_gh, rec_bits, slot, c, b, h, alt)) atomicMin(&claim[alt], (uint32_t)w);
    }
}

// (reads slot as it was before the round for its plan: the writes of other winners touch other buckets)
__global__ __launch_bounds__(256) void gs_lb_move_kernel(const uint32_t *w_valid, const uint32_t *w_gh, uint32_t *state, int64_t n_w, uint32_t rec_bits,
                                                         const uint32_t *slot_in, uint32_t *slot_out, const uint32_t *claim, u64 *changes) {
    u64 mine = 0;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_w; w += (int64_t)gridDim.x * blockDim.x) {
        if (!w_valid[w] || (state[w] & 0x80000000u)) continue;
        int c;
        uint32_t b, h, alt;
        if (!gs_lb_plan((uint32_t)w, w_gh, rec_bits, slot_in, c, b, h, alt) || claim[alt] != (uint32_t)w) continue;
        if (h != GS_LB_EMPTY) {
            slot_out[alt] = h;
            state[h] = 0x80000000u | (gs_rec_bucket(w_gh[h], rec_bits, 1) == alt ? 1u : 0u);
        }
        slot_out[b] = (uint32_t)w;
        state[w] = 0x80000000u | (uint32_t)c;
        mine++;
    }
    if (mine) atomicAdd(changes, mine);
}

// win_bucket[w] = its bucket, from the slots
__global__ __launch_bounds__(256) void gs_lb_buckets_kernel(const uint32_t *slot, int64_t n_rec, uint32_t *win_bucket) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_rec; b += (int64_t)gridDim.x * blockDim.x)
        if (slot[b] != GS_LB_EMPTY) win_bucket[slot[b]] = (uint32_t)b;
}

__global__ __launch_bounds__(256) void gs_lb_lines_kernel(const uint32_t *slot, int64_t n_rec, const u64 *w_hi, const u64 *w_lo, const uint32_t *w_valid,
                                                          u64 *rec) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_rec; b += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w = slot[b];
        if (w == GS_LB_EMPTY) continue;
        rec[b * GS_REC_WORDS] = w_hi[w];
        rec[b * GS_REC_WORDS + 1] = w_lo[w] | ((u64)w_valid[w] << GS_REC_WIN_BITS);
    }
}

// every sorted entry: value into its window's line, or -- window without a bucket, entry in no kept window -- into the table
// list, with its minimizer on the `more` list
__global__ __launch_bounds__(256) void gs_lb_values_kernel(const uint32_t *gh, const uint32_t *ohi, const uint32_t *olo, const uint32_t *vj,
                                                           const uint32_t *group, const uint8_t *assign, const uint32_t *win_bucket, int64_t n, int k,
                                                           u64 *rec, u64 *t_key, int32_t *t_val, uint32_t *m_gh, u64 *cnt) {
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t kmask = (1u << k) - 1u;
    u64 mine = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL; i0 < n; i0 += stride) {
        const int64_t i = i0 + lane;
        bool homeless = false;
        if (i < n) {
            const int a = assign[i];
            const uint32_t bkt = a < 2 ? win_bucket[2 * (u64)group[i] + (u64)a] : GS_LB_EMPTY;
            if (bkt != GS_LB_EMPTY) {
                const int j = (int)(vj[i] & 31u);
                atomicOr(&rec[(u64)bkt * GS_REC_WORDS + 2 + j / 3], (u64)(vj[i] >> 5) << (GS_REC_VAL_BITS * (j % 3)));
                mine++;
            } else
                homeless = true;
        }
        const u64 te = gs_lb_append(homeless, cnt + GS_LB_N_T, lane);
        const u64 me = gs_lb_append(homeless, cnt + GS_LB_N_M, lane);
        if (homeless) {
            uint32_t phi, plo;
            gs_rep_planes(ohi[i], olo[i], k, kmask, phi, plo);
            t_key[te] = gs_mix_planes(phi, plo);
            t_val[te] = (int32_t)(vj[i] >> 5);
            m_gh[me] = gh[i];
        }
    }
    if (mine) atomicAdd(cnt + GS_LB_IN_REC, mine);
}

__global__ __launch_bounds__(256) void gs_lb_more_kernel(const uint32_t *m_gh, int64_t n, uint32_t rec_bits, u64 *rec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        for (int ch = 0; ch < 2; ch++) {
            u64 *rp = rec + (u64)gs_rec_bucket(m_gh[i], rec_bits, ch) * GS_REC_WORDS;
            if (rp[2] & GS_REC_MORE) continue;  // (whoever set it sets the others too)
            for (int x = 2; x < GS_REC_WORDS; x++) atomicOr(&rp[x], GS_REC_MORE);
        }
}

// ---- overflow table, the same whatever the timing.  The keys are sorted by (home bucket, rest of the hash); pass d = 0..3
// looks at the keys that are still to be placed -- all of one home bucket are neighbours and aim at bucket home + d --: the
// r-th of them takes slot fill[target] + r if that is below 8, the others go on to pass d + 1.  A bucket gets keys from one home
// bucket per pass, buckets fill front to back, and a key is displaced past a bucket only if that bucket is full: the invariants
// the probe relies on.  What is left after pass GS_MAX_DISP overflows (the caller retries with twice the buckets).
__global__ __launch_bounds__(256) void gs_lb_rot_kernel(const u64 *t_key, int64_t n, int b, u64 *rot) {  // home bucket into the top bits
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        rot[i] = (t_key[i] << (64 - b)) | (t_key[i] >> b);
}

// start[i] = i where item i is the first of its home bucket, else 0 (an inclusive max-scan turns it into "first of my group")
__global__ __launch_bounds__(256) void gs_lb_tstart_kernel(const u64 *rot, int64_t n, int b, uint32_t *start) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        start[i] = (i > 0 && (rot[i] >> (64 - b)) != (rot[i - 1] >> (64 - b))) ? (uint32_t)i : 0u;
}

__global__ __launch_bounds__(256) void gs_lb_tplace_kernel(const u64 *rot, const int32_t *val, const uint32_t *start, int64_t n, int b, int vbits, int d,
                                                           const uint32_t *fill_in, uint32_t *fill_out, u64 *table, uint32_t *left) {
    const u64 mask = (1ULL << b) - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 home = rot[i] >> (64 - b), rem = rot[i] & ((1ULL << (64 - b)) - 1);  // rem = hash >> b
        const u64 bk = (home + (u64)d) & mask;
        const uint32_t r = (uint32_t)i - start[i], at = fill_in[bk] + r;
        if (at < GS_SLOTS_PER_BUCKET) {
            table[bk * GS_SLOTS_PER_BUCKET + at] = (rem << (vbits + 3)) | ((u64)d << (vbits + 1)) | ((u64)(val[i] + 1) << 1);
            atomicMax(&fill_out[bk], at + 1u);
            left[i] = 0;
        } else {
            atomicMax(&fill_out[bk], (uint32_t)GS_SLOTS_PER_BUCKET);
            left[i] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void gs_lb_tcompact_kernel(const u64 *rot, const int32_t *val, const uint32_t *left, const uint32_t *pos, int64_t n, u64 *rot2,
                                                             int32_t *val2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (left[i]) {
            rot2[pos[i]] = rot[i];
            val2[pos[i]] = val[i];
        }
}

__global__ __launch_bounds__(256) void gs_lb_gate_kernel(const uint32_t *h_gh, int64_t n, int ctx, uint32_t mgate_bits, uint32_t *mgate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t *w = mgate + (ctx ? gs_mgate_word_ctx(h_gh[i], mgate_bits) : gs_mgate_word(h_gh[i], mgate_bits));
        const uint32_t bits = gs_mgate_bits(h_gh[i]);
        if ((*w & bits) != bits) atomicOr(w, bits);
    }
}

// windows that sit in their minimizer's SECOND candidate bucket: (minimizer, the window's two context values) on the hint list
__global__ __launch_bounds__(256) void gs_lb_hint_collect_kernel(const uint32_t *w_valid, const uint32_t *w_gh, const u64 *w_hi, const u64 *w_lo,
                                                                 const uint32_t *win_bucket, int64_t n_w, uint32_t rec_bits, int k, uint32_t *hint_gh,
                                                                 uint32_t *hint_cx, u64 *cnt) {
    const int lane = (int)(threadIdx.x & 63);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL; i0 < n_w; i0 += stride) {
        const int64_t w = i0 + lane;
        bool second = false;
        if (w < n_w && w_valid[w] != 0) {
            const uint32_t bkt = win_bucket[w];
            second = bkt != GS_LB_EMPTY && bkt != gs_rec_bucket(w_gh[w], rec_bits, 0);
        }
        const u64 at = gs_lb_append(second, cnt + GS_LB_N_HINT, lane);
        if (second) {
            hint_gh[at] = w_gh[w];
            hint_cx[at] = k >= GS_CTX_MIN_K ? gs_window_ctx(w_hi[w], w_lo[w], k, true) | (gs_window_ctx(w_hi[w], w_lo[w], k, false) << 16) : 0u;
        }
    }
}

// ... and their hint bits into the finished gate (keyed by the minimizer, or by both context keys of the window)
__global__ __launch_bounds__(256) void gs_lb_hint_kernel(const uint32_t *hint_gh, const uint32_t *hint_cx, int64_t n, int ctx, uint32_t mgate_bits,
                                                         uint32_t *mgate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t gh = hint_gh[i];
        for (int side = 0; side < (ctx ? 2 : 1); side++) {
            const uint32_t key = ctx ? gs_gate_ctx_key_raw(gh, (hint_cx[i] >> (16 * side)) & 0xffffu) : gh;
            uint32_t *w = mgate + (ctx ? gs_mgate_word_ctx(key, mgate_bits) : gs_mgate_word(key, mgate_bits));
            const uint32_t bit = gs_mgate_hint(key);
            if ((*w & bit) == 0) atomicOr(w, bit);
        }
    }
}

// number of distinct values in a SORTED array
__global__ __launch_bounds__(256) void gs_lb_distinct_kernel(const uint32_t *a, int64_t n, u64 *out) {
    u64 mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) mine += (i == 0 || a[i] != a[i - 1]);
    if (mine) atomicAdd(out, mine);
}

static int lb_grid(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g > 16384) g = 16384;
    return g < 1 ? 1 : (int)g;
}

#define LB_LAUNCH(kernel, n, ...)                                                                   \
    do {                                                                                            \
        if ((n) > 0) hipLaunchKernelGGL(kernel, dim3(lb_grid(n)), dim3(256), 0, stream, __VA_ARGS__); \
    } while (0)

extern "C" hipError_t gs_lb_perkey(const int64_t *kmers, const int32_t *vidx, int64_t n, int k, const int32_t *parent, uint32_t *e_gh, uint32_t *e_ohi,
                                   uint32_t *e_olo, uint32_t *e_vj, u64 *e_sort, u64 *e_sort2, u64 *t_key, int32_t *t_val, uint32_t *m_gh, uint32_t *h_gh,
                                   uint32_t *h_ctx, u64 *cnt, hipStream_t stream) {
    LB_LAUNCH(gs_lb_perkey_kernel, n, kmers, vidx, n, k, parent, e_gh, e_ohi, e_olo, e_vj, e_sort, e_sort2, t_key, t_val, m_gh, h_gh, h_ctx, cnt);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gs_lb_gather64_kernel(const uint32_t *perm, int64_t n, const u64 *a, u64 *b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[perm[i]];
}

static hipError_t lb_sort_pairs(u64 *keys, u64 *keys_alt, uint32_t *vals, uint32_t *vals_alt, int64_t n, int bits, u64 **k_out, uint32_t **v_out,
                                hipStream_t stream) {
    rocprim::double_buffer<u64> dk(keys, keys_alt);
    rocprim::double_buffer<uint32_t> dv(vals, vals_alt);
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, (size_t)n, 0, (unsigned)bits, stream);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, (size_t)n, 0, (unsigned)bits, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    *k_out = dk.current();
    *v_out = dv.current();
    return e;
}

// sorts the n entries by (e_sort, e_sort2) -- a total order: the entries came in whatever order the waves appended them -- and
// writes them to the s_* arrays.  Two stable radix sorts, least significant key first.  sort_alt, perm, perm_alt: scratch of n.
extern "C" hipError_t gs_lb_sort_entries(u64 *e_sort, u64 *e_sort2, u64 *sort_alt, uint32_t *perm, uint32_t *perm_alt, int64_t n, const uint32_t *e_gh,
                                         const uint32_t *e_ohi, const uint32_t *e_olo, const uint32_t *e_vj, uint32_t *s_gh, uint32_t *s_ohi,
                                         uint32_t *s_olo, uint32_t *s_vj, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    LB_LAUNCH(gs_lb_iota_kernel, n, perm, n);
    u64 *k2 = nullptr, *k1 = nullptr;
    uint32_t *p2 = nullptr, *p1 = nullptr;
    hipError_t e = lb_sort_pairs(e_sort2, sort_alt, perm, perm_alt, n, 36, &k2, &p2, stream);
    if (e != hipSuccess) return e;
    // the primary keys in the order of the first sort (k2's buffer pair is free now: its other half takes them)
    u64 *g1 = k2 == e_sort2 ? sort_alt : e_sort2;
    LB_LAUNCH(gs_lb_gather64_kernel, n, p2, n, e_sort, g1);
    uint32_t *p_other = p2 == perm ? perm_alt : perm;
    e = lb_sort_pairs(g1, e_sort, p2, p_other, n, 64, &k1, &p1, stream);
    if (e != hipSuccess) return e;
    LB_LAUNCH(gs_lb_gather_kernel, n, p1, n, e_gh, e_ohi, e_olo, e_vj, s_gh, s_ohi, s_olo, s_vj);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    return e;
}

// head / group (n each), g_start (n + 1): *n_groups minimizers
extern "C" hipError_t gs_lb_groups(const uint32_t *s_gh, int64_t n, uint32_t *head, uint32_t *group, uint32_t *g_start, int64_t *n_groups,
                                   hipStream_t stream) {
    *n_groups = 0;
    if (n <= 0) return hipSuccess;
    LB_LAUNCH(gs_lb_heads_kernel, n, s_gh, n, head);
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::inclusive_scan(nullptr, tmp_bytes, head, group, (size_t)n, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(tmp, tmp_bytes, head, group, (size_t)n, rocprim::plus<uint32_t>(), stream);
    uint32_t last = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&last, group + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    if (e != hipSuccess) return e;
    *n_groups = (int64_t)last;
    LB_LAUNCH(gs_lb_dec_kernel, n, group, n);  // group ids from 0: the scan counted from 1
    LB_LAUNCH(gs_lb_starts_kernel, n, head, group, n, *n_groups, g_start);
    return hipGetLastError();
}

extern "C" hipError_t gs_lb_cluster(const uint32_t *s_gh, const uint32_t *s_ohi, const uint32_t *s_olo, const uint32_t *s_vj, const uint32_t *g_start,
                                    int64_t n_groups, int k, uint8_t *assign, u64 *w_hi, u64 *w_lo, uint32_t *w_valid, uint32_t *w_gh, u64 *cnt,
                                    hipStream_t stream) {
    if (n_groups <= 0) return hipSuccess;
    hipLaunchKernelGGL(gs_lb_cluster_kernel, dim3(lb_grid(n_groups, 128)), dim3(128), 0, stream, s_gh, s_ohi, s_olo, s_vj, g_start, n_groups, k, assign,
                       w_hi, w_lo, w_valid, w_gh, cnt);
    return hipGetLastError();
}

// slot / slot2 / claim: n_rec entries (the last two scratch), state / win_bucket: n_w entries, changes: one device counter.  At most
// max_rounds bidding rounds and as many move rounds.
extern "C" hipError_t gs_lb_place(const uint32_t *w_valid, const uint32_t *w_gh, int64_t n_w, uint32_t rec_bits, int max_rounds, uint32_t *slot,
                                  uint32_t *slot2, uint32_t *claim, uint32_t *state, uint32_t *win_bucket, u64 *changes, int *rounds_done,
                                  hipStream_t stream) {
    const int64_t n_rec = (int64_t)1 << rec_bits;
    *rounds_done = 0;
    hipError_t e = hipMemsetAsync(slot, 0xff, (size_t)n_rec * sizeof(uint32_t), stream);
    if (e == hipSuccess) e = hipMemsetAsync(state, 0, (size_t)(n_w > 0 ? n_w : 1) * sizeof(uint32_t), stream);
    if (e == hipSuccess) e = hipMemsetAsync(win_bucket, 0xff, (size_t)(n_w > 0 ? n_w : 1) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    int quiet = 0;
    for (int r = 0; r < max_rounds; r++) {
        e = hipMemsetAsync(changes, 0, sizeof(u64), stream);
        if (e != hipSuccess) return e;
        LB_LAUNCH(gs_lb_bid_kernel, n_w, w_valid, w_gh, state, n_w, rec_bits, slot);
        LB_LAUNCH(gs_lb_resolve_kernel, n_w, w_valid, w_gh, state, n_w, rec_bits, slot, changes);
        u64 ch = 0;
        e = hipMemcpyAsync(&ch, changes, sizeof(u64), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        *rounds_done = r + 1;
        quiet = ch == 0 ? quiet + 1 : 0;
        if (quiet == 2) break;  // nobody won or lost a bucket with either choice: the bidders that are left lose every time
    }
    for (int r = 0; r < max_rounds; r++) {  // one-step moves for those still without a bucket
        e = hipMemsetAsync(changes, 0, sizeof(u64), stream);
        if (e == hipSuccess) e = hipMemsetAsync(claim, 0xff, (size_t)n_rec * sizeof(uint32_t), stream);
        if (e == hipSuccess) e = hipMemcpyAsync(slot2, slot, (size_t)n_rec * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        LB_LAUNCH(gs_lb_claim_kernel, n_w, w_valid, w_gh, state, n_w, rec_bits, slot, claim);
        LB_LAUNCH(gs_lb_move_kernel, n_w, w_valid, w_gh, state, n_w, rec_bits, slot2, slot, claim, changes);
        u64 ch = 0;
        e = hipMemcpyAsync(&ch, changes, sizeof(u64), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        *rounds_done += 1;
        if (ch == 0) break;
    }
    LB_LAUNCH(gs_lb_buckets_kernel, n_rec, slot, n_rec, win_bucket);
    return hipGetLastError();
}

extern "C" hipError_t gs_lb_lines(const uint32_t *slot, uint32_t rec_bits, const u64 *w_hi, const u64 *w_lo, const uint32_t *w_valid, const uint32_t *s_gh,
                                  const uint32_t *s_ohi, const uint32_t *s_olo, const uint32_t *s_vj, const uint32_t *group, const uint8_t *assign,
                                  const uint32_t *win_bucket, int64_t n_e, int k, u64 *rec, u64 *t_key, int32_t *t_val, uint32_t *m_gh, u64 *cnt,
                                  hipStream_t stream) {
    const int64_t n_rec = (int64_t)1 << rec_bits;
    hipError_t e = hipMemsetAsync(rec, 0, (size_t)n_rec * GS_REC_WORDS * sizeof(u64), stream);
    if (e != hipSuccess) return e;
    LB_LAUNCH(gs_lb_lines_kernel, n_rec, slot, n_rec, w_hi, w_lo, w_valid, rec);
    LB_LAUNCH(gs_lb_values_kernel, n_e, s_gh, s_ohi, s_olo, s_vj, group, assign, win_bucket, n_e, k, rec, t_key, t_val, m_gh, cnt);
    return hipGetLastError();
}

extern "C" hipError_t gs_lb_more(const uint32_t *m_gh, int64_t n_m, uint32_t rec_bits, u64 *rec, hipStream_t stream) {
    LB_LAUNCH(gs_lb_more_kernel, n_m, m_gh, n_m, rec_bits, rec);
    return hipGetLastError();
}

static hipError_t lb_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, bool max_op, bool exclusive, hipStream_t stream) {
    size_t tmp_bytes = 0;
    void *tmp = nullptr;
    hipError_t e;
    if (max_op)
        e = rocprim::inclusive_scan(nullptr, tmp_bytes, in, out, (size_t)n, rocprim::maximum<uint32_t>(), stream);
    else if (exclusive)
        e = rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
    else
        e = rocprim::inclusive_scan(nullptr, tmp_bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    if (max_op)
        e = rocprim::inclusive_scan(tmp, tmp_bytes, in, out, (size_t)n, rocprim::maximum<uint32_t>(), stream);
    else if (exclusive)
        e = rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
    else
        e = rocprim::inclusive_scan(tmp, tmp_bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    return e;
}

// t_key / t_val: the n_t table keys in any order (only read).  rot_a / rot_b (u64), val_b / val_c (i32), start / left / pos / perm /
// perm_alt (u32): scratch of n_t each; fill_a / fill_b: 2^b counters; table: 2^b * 8 slots.  *overflow = keys that found no slot (the caller
// retries with b + 1), *max_disp = the largest displacement used.
extern "C" hipError_t gs_lb_table(const u64 *t_key, const int32_t *t_val, int64_t n_t, int b, int vbits, u64 *rot_a, u64 *rot_b, int32_t *val_b,
                                  int32_t *val_c, uint32_t *start, uint32_t *left, uint32_t *pos, uint32_t *perm, uint32_t *perm_alt, uint32_t *fill_a,
                                  uint32_t *fill_b, u64 *table, int64_t *overflow, int *max_disp, hipStream_t stream) {
    const size_t nb = (size_t)1 << b;
    *overflow = 0;
    *max_disp = 0;
    hipError_t e = hipMemsetAsync(fill_a, 0, nb * sizeof(uint32_t), stream);
    if (e == hipSuccess) e = hipMemsetAsync(table, 0, nb * GS_SLOTS_PER_BUCKET * sizeof(u64), stream);
    if (e != hipSuccess || n_t <= 0) return e;
    // sorted by (home bucket, rest): rot_a holds the rotated hashes; the values follow through a permutation
    LB_LAUNCH(gs_lb_rot_kernel, n_t, t_key, n_t, b, rot_a);
    LB_LAUNCH(gs_lb_iota_kernel, n_t, perm, n_t);
    u64 *ks = nullptr;
    uint32_t *ps = nullptr;
    e = lb_sort_pairs(rot_a, rot_b, perm, perm_alt, n_t, 64, &ks, &ps, stream);
    if (e != hipSuccess) return e;
    u64 *cur_rot = ks, *other_rot = ks == rot_a ? rot_b : rot_a;
    // values in sorted order: t_val -> val_b through ps (reinterpreted as 32-bit words)
    LB_LAUNCH(gs_lb_gather_kernel, n_t, ps, n_t, (const uint32_t *)t_val, (const uint32_t *)t_val, (const uint32_t *)t_val, (const uint32_t *)t_val,
              (uint32_t *)val_b, (uint32_t *)val_b, (uint32_t *)val_b, (uint32_t *)val_b);
    int32_t *cur_val = val_b, *other_val = val_c;
    uint32_t *fill_in = fill_a, *fill_out = fill_b;
    int64_t n = n_t;
    for (int d = 0; d <= GS_MAX_DISP && n > 0; d++) {
        e = hipMemcpyAsync(fill_out, fill_in, nb * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        LB_LAUNCH(gs_lb_tstart_kernel, n, cur_rot, n, b, start);
        e = lb_scan_u32(start, start, n, true, false, stream);
        if (e != hipSuccess) return e;
        LB_LAUNCH(gs_lb_tplace_kernel, n, cur_rot, cur_val, start, n, b, vbits, d, fill_in, fill_out, table, left);
        e = lb_scan_u32(left, pos, n, false, true, stream);
        if (e != hipSuccess) return e;
        uint32_t last_pos = 0, last_left = 0;
        e = hipMemcpyAsync(&last_pos, pos + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&last_left, left + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        const int64_t n_left = (int64_t)last_pos + (int64_t)last_left;
        if (n_left < n) *max_disp = d;
        if (n_left > 0) {
            LB_LAUNCH(gs_lb_tcompact_kernel, n, cur_rot, cur_val, left, pos, n, other_rot, other_val);
            std::swap(cur_rot, other_rot);
            std::swap(cur_val, other_val);
        }
        std::swap(fill_in, fill_out);
        n = n_left;
    }
    *overflow = n;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    return e;
}

// h_gh is sorted in place (h_alt: scratch of n_h); *distinct = number of different minimizers
extern "C" hipError_t gs_lb_distinct(uint32_t *h_gh, uint32_t *h_alt, int64_t n_h, u64 *d_scratch, int64_t *distinct, uint32_t **sorted,
                                     hipStream_t stream) {
    *distinct = 0;
    *sorted = h_gh;
    if (n_h <= 0) return hipSuccess;
    rocprim::double_buffer<uint32_t> dk(h_gh, h_alt);
    size_t tmp_bytes = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, tmp_bytes, dk, (size_t)n_h, 0, 32, stream);
    if (e != hipSuccess) return e;
    void *tmp = nullptr;
    e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 1);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_keys(tmp, tmp_bytes, dk, (size_t)n_h, 0, 32, stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_scratch, 0, sizeof(u64), stream);
    if (e == hipSuccess) {
        LB_LAUNCH(gs_lb_distinct_kernel, n_h, dk.current(), n_h, d_scratch);
        e = hipGetLastError();
    }
    u64 d = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&d, d_scratch, sizeof(u64), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(tmp);
    *distinct = (int64_t)d;
    *sorted = dk.current();
    return e;
}

extern "C" hipError_t gs_lb_hint_collect(const uint32_t *w_valid, const uint32_t *w_gh, const u64 *w_hi, const u64 *w_lo, const uint32_t *win_bucket,
                                         int64_t n_w, uint32_t rec_bits, int k, uint32_t *hint_gh, uint32_t *hint_cx, u64 *cnt, hipStream_t stream) {
    LB_LAUNCH(gs_lb_hint_collect_kernel, n_w, w_valid, w_gh, w_hi, w_lo, win_bucket, n_w, rec_bits, k, hint_gh, hint_cx, cnt);
    return hipGetLastError();
}

extern "C" hipError_t gs_lb_hint(const uint32_t *hint_gh, const uint32_t *hint_cx, int64_t n, int ctx, uint32_t mgate_bits, uint32_t *mgate,
                                 hipStream_t stream) {
    LB_LAUNCH(gs_lb_hint_kernel, n, hint_gh, hint_cx, n, ctx, mgate_bits, mgate);
    return hipGetLastError();
}

extern "C" hipError_t gs_lb_gate(const uint32_t *h_gh, int64_t n_h, int ctx, uint32_t mgate_bits, uint32_t *mgate, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(mgate, 0, ((size_t)1 << mgate_bits) * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    LB_LAUNCH(gs_lb_gate_kernel, n_h, h_gh, n_h, ctx, mgate_bits, mgate);
    return hipGetLastError();
}
