// Compact bilinear pooling, forward: the raw Gram X X^T and the count-sketch binning in ONE kernel - G never reaches HBM.
//
// Round 2 ran Gram (writes G: 67 MB at B = 64) -> cbp_rowscatter_kernel (reads it back) -> two finishing launches:
// 187 MB moved for 27 MB of algorithmic traffic, 76 us at B = 64 and 52 us at the yaml batch of 16, where the Gram
// launched 128 workgroups on 256 CUs.  Here a workgroup computes 64x64 tiles (I, J), J >= I, of one sample's Gram on
// the panel-resident MFMA loop of bcnn_fast.hip (A panel = row block I, full K, resident in LDS), parks each finished
// tile in LDS and bins it from there into a PRIVATE copy of the D output bins, also in LDS; the per-workgroup partial
// bin vectors (24 KB each) are added in fixed order by cbp_finish_kernel (cbp.hip), which also normalises.
//
// Binning a tile deterministically, without atomics, with four waves in parallel: it is a GATHER.  For every ordered
// pair of 64-row blocks (I, J) the plan holds the 4096 entries (i in I, j in J) -> bin (h1_i + h2_j) mod D, sign
// s1_i s2_j, sorted by bin and cut into FOUR runs, one per binning wave: wave w owns the bins [w D / 4, (w + 1) D / 4) in
// every list, so no bin is ever touched by two waves, whatever their relative progress.  A run has 18 x 64 = 1152
// (step, lane) slots (>= 1024 + 4.6 sigma of the run length; padded with entries that add 0 to a dump bin; the plan build
// fails over to the unfused path if a run does not fit), and in step e every lane adds one entry:
//     c[bin] += +-T[idx]           T = the tile in LDS
// The plan build deals the entries so that the 64 entries of a step (and, for the paired form, of a pair of steps) hit
// distinct bins - and, round 5, so that the 32 addresses a half wave presents per LDS access fall on different banks;
// later steps of a wave may hit the same bin again - LDS instructions of one wave execute in order, and a step's (pair of
// steps') read-modify-write is complete in the instruction stream before the next one's read is issued.  A computed
// tile (I, J), I < J, feeds two lists: (I, J) and (J, I) (G_ji = G_ij, other bin); a diagonal tile one.  Per bin the
// additions happen in a fixed order (tile order of the workgroup, position within a list): bit-reproducible.
//
// LDS: A panel 50 KB + ONE column panel 50 KB (the next one waits in registers during the tile's MFMA loop and is
// written behind it - a second panel buffer does not fit next to the bins) + two tile buffers of 17 KB (stored
// transposed, pitch 68: 16-byte stores without bank conflicts) + CBF_DMAX bins 24 KB = 160 KB, one static array.
// Work split: `items` per sample, each one or two runs of tiles (row block, first column block, count) - B = 64: the
// balanced pairs {p, nb - 1 - p} (nb + 1 tiles, one workgroup per CU); smaller batches: rows cut into runs of <= 3 tiles
// so that B x items fills the chip (B = 16, C = 512: 16 items per sample = 256 workgroups).
#pragma once
#include <vector>
#include <array>
#include <algorithm>
#include "hk_gram_tile.h"

namespace hk {

constexpr int CBF_LSTEPS = 18;                     // entries per lane and list (even: the steps go in pairs)
constexpr int CBF_WLEN = CBF_LSTEPS * 64;          // entries of a wave's run (padded)
constexpr int CBF_LLEN = 4 * CBF_WLEN;             // words of one list
constexpr int CBF_TP = 68;                         // pitch of the transposed tile T[j][i]
constexpr int CBF_TSZ = 64 * CBF_TP + 4;           // + a zero slot (index 64 * 68) for the padding entries
constexpr int CBF_MAXITEMS = 40;
constexpr int CBF_DMAX = 6144;                     // floats of LDS reserved for the bins: D + 1 <= CBF_DMAX

struct CbfItem {                                   // up to two runs: row block I, column blocks J0 .. J0 + cnt - 1
    unsigned char I[2], J0[2], cnt[2];
};
struct CbfSchedule {
    int nitems;
    CbfItem it[CBF_MAXITEMS];
};

// entry word: bits 0..14 BYTE offset into T, bits 16..30 BYTE offset of the bin, bit 31 negative sign - every field is
// usable after one instruction (v_and / v_bfe / v_and): the binning waves share their SIMDs with the MFMA waves, and
// on gfx950 an fp32 MFMA runs at the vector-FMA rate: every VALU instruction of the binning is time taken from the
// matrix pipe (tools/cbf_lab.py: 28 cycles per v_mov next to the MFMA stream)
__host__ __device__ __forceinline__ unsigned cbf_pack(int idx, int bin, int neg) { return (unsigned)(4 * idx) | ((unsigned)(4 * bin) << 16) | ((unsigned)neg << 31); }
__host__ __device__ __forceinline__ int cbf_bin_of(unsigned w) { return (int)((w >> 16) & 0x7fffu) / 4; }

// Host: the nb x nb lists of (C, D, hashes).  Returns 0 when the fused path cannot be used for these hashes (a run that
// does not fit its padding, an entry no step can take, D too large for the entry word), 1 when it can, 2 when in addition
// the two steps of every PAIR (2 p, 2 p + 1) of a wave hit disjoint bins - the kernel then issues the reads of a pair
// before its writes.  How the entries of a run are dealt to its (step, lane) slots: see the loop over the waves below.
static inline int cbf_build_lists(const int* h1, const float* s1, const int* h2, const float* s2, int C, int D,
                                  std::vector<unsigned>& out) {
    if (C % 64 != 0 || D + 1 > CBF_DMAX) return 0;
    const int nb = C / 64;
    out.assign((size_t)nb * nb * CBF_LLEN, cbf_pack(64 * CBF_TP, D, 0));       // padding: + T[zero slot] to the dump bin
    struct E { int bin, idx, neg; };
    std::vector<E> es(4096);
    bool pairs_ok = true;
    for (int I = 0; I < nb; ++I)
        for (int J = 0; J < nb; ++J) {
            // list (I, J): entries G[i][j], i in block I, j in block J.  The value is read from the tile the workgroup
            // computed: (I, J) itself when I <= J - stored transposed, T[jl][il] - else tile (J, I), in which G_ij = G_ji
            // sits at T[il][jl].
            for (int a = 0; a < 64; ++a)
                for (int b = 0; b < 64; ++b) {
                    const int i = I * 64 + a, j = J * 64 + b;
                    E e;
                    e.bin = (h1[i] + h2[j]) % D;
                    e.neg = (s1[i] * s2[j] < 0.f) ? 1 : 0;
                    e.idx = (I <= J) ? b * CBF_TP + a : a * CBF_TP + b;
                    es[a * 64 + b] = e;
                }
            std::stable_sort(es.begin(), es.end(), [](const E& p, const E& q) { return p.bin < q.bin; });
            // four runs, one per wave: wave w owns the bins [ceil(w D / 4), ceil((w + 1) D / 4)) in EVERY list - the waves
            // walk the lists of a workgroup's tiles at their own pace, so a bin must never change hands between lists
            int cut[5] = {0, 0, 0, 0, 4096};
            for (int w = 1; w < 4; ++w) {
                const int b0 = (int)(((long long)w * D + 3) / 4);
                int c = cut[w - 1];
                while (c < 4096 && es[c].bin < b0) ++c;
                cut[w] = c;
            }
            unsigned* L = out.data() + ((size_t)I * nb + J) * CBF_LLEN;
            for (int w = 0; w < 4; ++w) {
                const int n = cut[w + 1] - cut[w];
                if (n > CBF_WLEN || n < 0) return 0;
                // The run's n entries are dealt to its 18 x 64 (step, lane) slots; word (step q, lane t) at (w * 18 + q) * 64 + t.
                // Any deal is correct as long as one instruction never touches a bin twice (a step's 64 read-modify-writes;
                // for the paired form also the partner step's) - a bin's additions then happen in step order, fixed by the
                // plan.  Round 5 uses that freedom against LDS bank conflicts (round 3 dealt sorted runs of 18 to the lanes
                // with a stride-7 step pattern: the 32 addresses a half wave presents per 4-byte LDS access fell on random
                // banks, 3.3-way on average - 45 % of the kernel's LDS cycles were conflict cycles).  An entry has a gather
                // bank (idx mod 32) and a bin bank (bin mod 32): the entries of one half-step are chosen as a MATCHING of
                // the bipartite multigraph gather banks <-> bin banks (Kuhn's augmenting paths, busiest banks first), i.e.
                // 32 entries whose gather banks are all different AND whose bin banks are all different (two entries of one
                // bin share a bin bank, so a matching never holds a bin twice); bins already in the other half of the step
                // or in the partner step are excluded.  What the 36 matchings leave over goes to the free slots with the
                // fewest collisions.
                {
                    const E* el = &es[cut[w]];
                    std::vector<char> left(n, 1);
                    std::vector<int> slot(CBF_LSTEPS * 64, -1);                  // entry index per (step, lane)
                    std::vector<std::vector<int>> binsOf(CBF_LSTEPS);            // bins placed per step
                    auto in_step = [&](int q, int bin) {
                        for (int b_ : binsOf[q]) if (b_ == bin) return true;
                        return false;
                    };
                    for (int q = 0; q < CBF_LSTEPS; ++q)
                        for (int half = 0; half < 2; ++half) {
                            // candidate edges by gather bank; degrees for the "busiest first" order
                            std::vector<int> adj[32];
                            int degT[32] = {}, degB[32] = {};
                            std::vector<int> remBin(n, 0);                       // entries of the same bin still to place (sorted: neighbours)
                            for (int e = 0; e < n;) {
                                int f = e, c = 0;
                                while (f < n && el[f].bin == el[e].bin) { c += left[f]; ++f; }
                                for (int g_ = e; g_ < f; ++g_) remBin[g_] = c;
                                e = f;
                            }
                            for (int e = 0; e < n; ++e) {
                                if (!left[e]) continue;
                                degT[el[e].idx & 31]++;
                                degB[el[e].bin & 31]++;
                                if (in_step(q, el[e].bin) || in_step(q ^ 1, el[e].bin)) continue;
                                adj[el[e].idx & 31].push_back(e);
                            }
                            // edge order of a gather bank: entries of the bins with the most entries still to place first (a
                            // bin with c entries needs c steps no two of which are partners: the heavy ones must not be left
                            // for the end), then the busier bin bank
                            for (int tb = 0; tb < 32; ++tb)
                                std::stable_sort(adj[tb].begin(), adj[tb].end(), [&](int a_, int b_) {
                                    if (remBin[a_] != remBin[b_]) return remBin[a_] > remBin[b_];
                                    return degB[el[a_].bin & 31] > degB[el[b_].bin & 31];
                                });
                            int order[32];
                            for (int tb = 0; tb < 32; ++tb) order[tb] = tb;
                            std::stable_sort(order, order + 32, [&](int a_, int b_) { return degT[a_] > degT[b_]; });
                            int matchB[32], matchT[32];                          // bin bank -> entry, gather bank -> entry
                            for (int k = 0; k < 32; ++k) matchB[k] = matchT[k] = -1;
                            bool seen[32];
                            auto augment = [&](auto&& self, int tb) -> bool {
                                for (int e : adj[tb]) {
                                    const int bb = el[e].bin & 31;
                                    if (seen[bb]) continue;
                                    seen[bb] = true;
                                    if (matchB[bb] < 0 || self(self, el[matchB[bb]].idx & 31)) {
                                        matchB[bb] = e;
                                        matchT[tb] = e;
                                        return true;
                                    }
                                }
                                return false;
                            };
                            for (int k = 0; k < 32; ++k) {
                                for (int z = 0; z < 32; ++z) seen[z] = false;
                                augment(augment, order[k]);
                            }
                            int lane = 32 * half;
                            for (int tb = 0; tb < 32; ++tb) {
                                const int e = matchT[tb];
                                if (e < 0 || matchB[el[e].bin & 31] != e) continue;          // (an augmenting path may have re-matched it)
                                slot[q * 64 + lane++] = e;
                                left[e] = 0;
                                binsOf[q].push_back(el[e].bin);
                            }
                        }
                    // leftovers: the free slot with the fewest bank collisions whose step (and partner step) does not hold the bin
                    for (int e = 0; e < n; ++e) {
                        if (!left[e]) continue;
                        int best = -1;
                        long long bc = 0;
                        for (int q = 0; q < CBF_LSTEPS; ++q) {
                            if (in_step(q, el[e].bin) || in_step(q ^ 1, el[e].bin)) continue;
                            for (int half = 0; half < 2; ++half) {
                                int freeLane = -1, coll = 0;
                                for (int t = 32 * half; t < 32 * half + 32; ++t) {
                                    const int o = slot[q * 64 + t];
                                    if (o < 0) { if (freeLane < 0) freeLane = t; continue; }
                                    coll += ((el[o].idx & 31) == (el[e].idx & 31)) + ((el[o].bin & 31) == (el[e].bin & 31));
                                }
                                if (freeLane >= 0 && (best < 0 || coll < bc)) { best = q * 64 + freeLane; bc = coll; }
                            }
                        }
                        if (best < 0) {
                            // every admissible step is full: move one of its occupants to a free slot IT may take, and take its place
                            auto drop_bin = [&](int q, int bin) {
                                for (size_t z = 0; z < binsOf[q].size(); ++z)
                                    if (binsOf[q][z] == bin) { binsOf[q].erase(binsOf[q].begin() + z); return; }
                            };
                            for (int q = 0; q < CBF_LSTEPS && best < 0; ++q) {
                                if (in_step(q, el[e].bin) || in_step(q ^ 1, el[e].bin)) continue;
                                for (int t = 0; t < 64 && best < 0; ++t) {
                                    const int o = slot[q * 64 + t];
                                    if (o < 0) continue;
                                    for (int q2 = 0; q2 < CBF_LSTEPS && best < 0; ++q2) {
                                        if (q2 == q || in_step(q2, el[o].bin)) continue;
                                        if (q2 != (q ^ 1) && in_step(q2 ^ 1, el[o].bin)) continue;
                                        if (q2 == (q ^ 1)) continue;             // (o itself sits in q: its bin would meet itself across the pair)
                                        for (int t2 = 0; t2 < 64; ++t2)
                                            if (slot[q2 * 64 + t2] < 0) {
                                                slot[q2 * 64 + t2] = o;
                                                binsOf[q2].push_back(el[o].bin);
                                                drop_bin(q, el[o].bin);
                                                slot[q * 64 + t] = -1;
                                                best = q * 64 + t;
                                                break;
                                            }
                                    }
                                }
                            }
                        }
                        if (best < 0) {                                          // only steps that break the PAIRED form are left: take one
                            for (int q = 0; q < CBF_LSTEPS && best < 0; ++q) {
                                if (in_step(q, el[e].bin)) continue;
                                for (int t = 0; t < 64; ++t)
                                    if (slot[q * 64 + t] < 0) { best = q * 64 + t; break; }
                            }
                            if (best < 0) return 0;
                        }
                        slot[best] = e;
                        left[e] = 0;
                        binsOf[best / 64].push_back(el[e].bin);
                    }
                    for (int q = 0; q < CBF_LSTEPS; ++q)
                        for (int t = 0; t < 64; ++t)
                            if (slot[q * 64 + t] >= 0) {
                                const E& e = el[slot[q * 64 + t]];
                                L[(w * CBF_LSTEPS + q) * 64 + t] = cbf_pack(e.idx, e.bin, e.neg);
                            }
                }
                // the 64 entries of a step must hit distinct bins (the dump bin may repeat); two consecutive steps too
                // for the pipelined form
                auto distinct = [&](int st0, int nst) {
                    int seen[128], ns = 0;
                    for (int st = st0; st < st0 + nst; ++st)
                        for (int t = 0; t < 64; ++t) {
                            const int bin = cbf_bin_of(L[(w * CBF_LSTEPS + st) * 64 + t]);
                            if (bin == D) continue;
                            for (int u = 0; u < ns; ++u)
                                if (seen[u] == bin) return false;
                            seen[ns++] = bin;
                        }
                    return true;
                };
                for (int st = 0; st < CBF_LSTEPS; ++st)
                    if (!distinct(st, 1)) return 0;
                for (int st = 0; st + 1 < CBF_LSTEPS; st += 2)
                    if (!distinct(st, 2)) pairs_ok = false;
            }
        }
    return pairs_ok ? 2 : 1;
}

// Host: the work items of one sample for a batch of B (target: B x items >= the CU count, runs of <= 3 tiles)
static inline CbfSchedule cbf_schedule(int nb, int B, int ncu = 256) {
    CbfSchedule s;
    s.nitems = 0;
    auto add = [&](int I0, int J00, int c0, int I1 = 0, int J01 = 0, int c1 = 0) {
        CbfItem& it = s.it[s.nitems++];
        it.I[0] = (unsigned char)I0; it.J0[0] = (unsigned char)J00; it.cnt[0] = (unsigned char)c0;
        it.I[1] = (unsigned char)I1; it.J0[1] = (unsigned char)J01; it.cnt[1] = (unsigned char)c1;
    };
    const int pairs = (nb + 1) / 2;
    if ((long long)B * pairs >= ncu - ncu / 8 || nb * (nb + 1) / 2 <= pairs) {      // balanced pairs {p, nb - 1 - p}
        for (int p_ = 0; p_ < pairs; ++p_) {
            const int q = nb - 1 - p_;
            if (q != p_) add(p_, p_, nb - p_, q, q, nb - q);
            else add(p_, p_, nb - p_);
        }
        return s;
    }
    // rows cut into runs of at most `len` tiles: the len with the shortest makespan = rounds of workgroups (one per CU:
    // the kernel takes nearly all of a CU's LDS) x (tiles per run + 1 for the run's prologue); ties: the longer runs
    int best = nb;
    long long best_cost = -1;
    for (int len = nb; len >= 1; --len) {
        int n = 0;
        for (int I = 0; I < nb; ++I) n += (nb - I + len - 1) / len;
        if (n > CBF_MAXITEMS) break;
        const long long cost = (((long long)B * n + ncu - 1) / ncu) * (len + 1);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = len; }
    }
    for (int I = 0; I < nb; ++I) {
        const int total = nb - I, runs = (total + best - 1) / best;
        int J = I;
        for (int r = 0; r < runs; ++r) {                  // as even as possible: the first (total % runs) runs one longer
            const int c = total / runs + (r < total % runs ? 1 : 0);
            add(I, J, c);
            J += c;
        }
    }
    return s;
}

// 512 threads: waves 0-3 ("M") run the MFMA loops and stage the panels, waves 4-7 ("N") bin the finished tiles - one M
// and one N wave per SIMD, so the binning's LDS round trips (34 dependent read-modify-write steps per tile and lane) and
// its VALU work issue in the shadow of the other wave's MFMAs without any hand interleaving.  Two tile buffers; per
// tile t every wave passes two barriers:
//     M: MFMA loop t -> store T[t & 1] -> B1 -> store the next panel (sB, or sA for a new row run) -> B2 -> loop t + 1
//     N:                                  B1 ->                                                      B2 -> bin tile t
// T[t & 1] is written before B1(t), read after B2(t) and until the N waves arrive at B1(t + 1), rewritten after
// B2(t + 1); sB / sA are rewritten between B1 and B2, when no M wave reads them.
// PIPE: the plan guarantees that two consecutive steps of a wave hit disjoint bins (cbf_build_lists = 2): the reads of
// steps 2 p, 2 p + 1 are issued before their writes - 9 dependent LDS round trips per list instead of 17.

template <int HW, bool PIPE>
__global__ __launch_bounds__(512, 2) void cbp_fused_kernel(const float* __restrict__ x, const unsigned* __restrict__ lists,
                                                           float* __restrict__ part, int C, int nb, int B, int D,
                                                           const CbfSchedule sch) {
    constexpr int PANEL = 64 * HW;
    constexpr int N4 = PANEL / 4;
    constexpr int NST = (N4 + 255) / 256;
    // LDS: the two tile buffers first, then the bins - everything the binning waves address sits below 64 KB at
    // compile-time bases, so a tile element is `ds_read vaddr offset:T` and a bin `ds_read / ds_write vaddr offset:bins`
    // with the entry's fields as the address registers
    // (STATIC, sized for D < CBF_DMAX: against a dynamic LDS array every address carries the array's link-time base and
    //  the compiler adds it - zero - with a VALU op per gather)
    __shared__ __attribute__((aligned(16))) float lds[2 * CBF_TSZ + CBF_DMAX + 2 * PANEL];
    float* sT = lds;                                // 2 x ([64][68] transposed tile + zero slot)
    float* sc = sT + 2 * CBF_TSZ;                   // [D + 1] bins (+ dump)
    float* sA = sc + CBF_DMAX;
    float* sB = sA + PANEL;

    int b, w;
    if (!xcd_map(blockIdx.x, B, sch.nitems, b, w)) return;
    const CbfItem& it = sch.it[w];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_m = wave < 4;                     // wave-uniform role
    const int mt = tid & 255;                       // thread index within the role
    const int wq = wave & 3;
    const int wm = wq >> 1, wn = wq & 1, l31 = lane & 31, lh = lane >> 5;
    const float* xb = x + (long long)b * C * HW;
    // the workgroup's tiles in order: tile t = (row block, column block); the first tile of a run starts a new row panel
    // (the item's six bytes are read ONCE into scalars: `sch` lives in the kernarg segment behind a dynamic index, every
    //  mention of a field is a global byte load - inside the tile loops that was a memory latency per tile and role)
    const int iI0 = __builtin_amdgcn_readfirstlane((int)it.I[0]), iI1 = __builtin_amdgcn_readfirstlane((int)it.I[1]);
    const int iJ0 = __builtin_amdgcn_readfirstlane((int)it.J0[0]), iJ1 = __builtin_amdgcn_readfirstlane((int)it.J0[1]);
    const int c0 = __builtin_amdgcn_readfirstlane((int)it.cnt[0]);
    const int ntile = c0 + __builtin_amdgcn_readfirstlane((int)it.cnt[1]);
    auto tile_I = [&](int t) { return t < c0 ? iI0 : iI1; };
    auto tile_J = [&](int t) { return t < c0 ? iJ0 + t : iJ1 + (t - c0); };

    for (int k = tid; k <= D; k += 512) sc[k] = 0.f;
    if (tid < 4) { sT[64 * CBF_TP + tid] = 0.f; sT[CBF_TSZ + 64 * CBF_TP + tid] = 0.f; }

    auto load_panel = [&](f32x4 (&st)[NST], int blk) {
        const f32x4* src = reinterpret_cast<const f32x4*>(xb + (long long)blk * PANEL);
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int f = mt + 256 * u;
            st[u] = src[f < N4 ? f : N4 - 1];
        }
    };
    auto store_panel = [&](const f32x4 (&st)[NST], float* dstp) {
        f32x4* dst = reinterpret_cast<f32x4*>(dstp);
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int f = mt + 256 * u;
            if (f < N4) dst[f] = st[u];
        }
    };
    {   // prologue: the first run's row panel (M waves) and, when it starts off the diagonal, its first column panel (N waves)
        f32x4 s0[NST];
        load_panel(s0, is_m ? iI0 : iJ0);
        if (is_m) store_panel(s0, sA);
        else if (iJ0 != iI0) store_panel(s0, sB);
    }
    __syncthreads();

    if (is_m) {
        for (int t = 0; t < ntile; ++t) {
            const int I = tile_I(t), J = tile_J(t);
            float* T = sT + (t & 1) * CBF_TSZ;
            // the panel needed next: the column panel of the next tile, or the row panel when the next tile starts a run
            const bool more = t + 1 < ntile;
            const bool nxt_is_row = more && t + 1 == c0;
            const int nxt = more ? (nxt_is_row ? tile_I(t + 1) : tile_J(t + 1)) : J;
            f32x4 st[NST];
            load_panel(st, nxt);                     // (unconditional: the staging registers stay registers)
            // (pinned: the scheduler otherwise sinks the 13 loads below the MFMA loop - 52 fewer live registers - and the
            //  panel store behind the first barrier starts by waiting for memory)
            __builtin_amdgcn_sched_barrier(0);

            const float* Ap = sA + (wm * 32 + l31) * HW + 4 * lh;
            const float* Bp = (J == I ? sA : sB) + (wn * 32 + l31) * HW + 4 * lh;
            f32x16 acc0, acc1, dummy;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; dummy[i] = 0.f; }
            GramEpi<1> ep;
            ep.yb = nullptr; ep.C = 0; ep.i0 = ep.j0 = ep.offdiag = 0; ep.inv = ep.inv_m = 0.f; ep.l31 = l31; ep.lh = lh;
            gram_tile<HW, false, NST>(Ap, Bp, acc0, acc1, dummy, ep, lh, st, reinterpret_cast<f32x4*>(sB), false, mt);
            const f32x16 t16 = acc0 + acc1;
            // the tile, transposed: T[j][i], four consecutive i per 16-byte store (C layout of the 32x32 MFMA:
            // col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 q4 = (f32x4){t16[4 * g], t16[4 * g + 1], t16[4 * g + 2], t16[4 * g + 3]};
                *reinterpret_cast<f32x4*>(T + (wn * 32 + l31) * CBF_TP + wm * 32 + 8 * g + 4 * lh) = q4;
            }
            HK_LDS_BARRIER();                       // B1
            // (a row panel that starts off the diagonal cannot follow: only the first run of an item may, and that one is
            //  loaded by the prologue)
            if (more) store_panel(st, nxt_is_row ? sA : sB);
            HK_LDS_BARRIER();                       // B2
        }
    } else {
        // (the younger wave of a SIMD loses the issue arbitration against the MFMA stream of its partner)
        __builtin_amdgcn_s_setprio(2);
        // The list words of a tile are requested ONE TILE AHEAD (they come from L2: 1.1 MB of lists, and the gathers cannot
        // start without them) into the register set the tile after next does not use: tiles alternate between sets A and
        // B - and between the two tile buffers - so nothing is copied and every LDS base is an immediate.
        auto ldlist = [&](unsigned (&e)[CBF_LSTEPS], int li) {
            const unsigned* L = lists + (long long)li * CBF_LLEN + (wq * CBF_LSTEPS) * 64 + lane;
#pragma unroll
            for (int q = 0; q < CBF_LSTEPS; ++q) e[q] = L[q * 64];
        };
        const char* scb = reinterpret_cast<const char*>(sc);
        // one list of the tile in buffer TB into the bins: per entry v_and (T offset), ds_read, v_and + v_xor (sign),
        // v_bfe (bin offset), ds_read, v_add, ds_write
        auto bin_list = [&](int TB, const unsigned (&e)[CBF_LSTEPS]) {
            const char* Tb = reinterpret_cast<const char*>(sT + TB * CBF_TSZ);
            float v[CBF_LSTEPS];
#pragma unroll
            for (int q = 0; q < CBF_LSTEPS; ++q) {                                // the gathers do not depend on the bins
                const float g_ = *HK_LDS_CONST(Tb + (e[q] & 0x7fffu));
                v[q] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, g_) ^ (e[q] & 0x80000000u));
            }
            // volatile: the read-modify-write steps stay in program order in the instruction stream (lanes of this wave
            // may hit the same bin in later steps; the LDS executes a wave's instructions in order)
#pragma unroll
            for (int q = 0; q < CBF_LSTEPS; q += 2) {
                auto c0p = HK_LDS_VOLATILE(scb + ((e[q] >> 16) & 0x7fffu));
                auto c1p = HK_LDS_VOLATILE(scb + ((e[q + 1] >> 16) & 0x7fffu));
                if (PIPE) {
                    const float r0 = *c0p, r1 = *c1p;
                    *c0p = r0 + v[q];
                    *c1p = r1 + v[q + 1];
                } else {
                    *c0p = *c0p + v[q];
                    *c1p = *c1p + v[q + 1];
                }
            }
        };
        unsigned a0[CBF_LSTEPS], a1[CBF_LSTEPS], b0[CBF_LSTEPS], b1[CBF_LSTEPS];
        ldlist(a0, tile_I(0) * nb + tile_J(0));
        ldlist(a1, tile_J(0) * nb + tile_I(0));
        // one tile: both barriers, then the requests for tile t + 1 into the other register set (pinned: left alone, the
        // scheduler sinks the 36 loads below the binning and the next tile starts by waiting for them), then the binning
#define HK_CBF_NTILE(t_, TB_, CUR0, CUR1, NXT0, NXT1)                                                  \
        do {                                                                                          \
            const int I_ = tile_I(t_), J_ = tile_J(t_);                                               \
            HK_LDS_BARRIER();                       /* B1 */                                          \
            HK_LDS_BARRIER();                       /* B2 */                                          \
            const int tn_ = (t_) + 1 < ntile ? (t_) + 1 : (t_);                                       \
            ldlist(NXT0, tile_I(tn_) * nb + tile_J(tn_));                                             \
            ldlist(NXT1, tile_J(tn_) * nb + tile_I(tn_));                                             \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            bin_list(TB_, CUR0);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            if (J_ != I_) bin_list(TB_, CUR1);                                                        \
        } while (0)
        for (int t = 0; t < ntile; t += 2) {
            HK_CBF_NTILE(t, 0, a0, a1, b0, b1);
            if (t + 1 < ntile) HK_CBF_NTILE(t + 1, 1, b0, b1, a0, a1);
        }
#undef HK_CBF_NTILE
    }
    __syncthreads();
    float* pp = part + ((long long)b * sch.nitems + w) * D;
    for (int k = tid; k < D; k += 512) pp[k] = sc[k];
}

static inline size_t cbf_lds_bytes(int HW) { return ((size_t)2 * 64 * HW + 2 * CBF_TSZ + CBF_DMAX) * sizeof(float); }

// HK_ERR_UNSUPPORTED when the shape is not covered (the caller takes the unfused path)
static inline int cbf_launch(const float* x, const unsigned* lists, float* part, int B, int C, int HW, int D,
                             const CbfSchedule& sch, bool pipe, hipStream_t st) {
    if (C % 64 != 0 || !aligned16(x)) return HK_ERR_UNSUPPORTED;
    if (cbf_lds_bytes(HW) > 160 * 1024 || D + 1 > CBF_DMAX) return HK_ERR_UNSUPPORTED;
    const int nb = C / 64;
    const dim3 grid(xcd_grid(B, sch.nitems));
#define HK_CBF_GO(H)                                                                                              \
    case H:                                                                                                       \
        if (pipe) {                                                                                               \
                        hipLaunchKernelGGL((cbp_fused_kernel<H, true>), grid, dim3(512), 0, st, x, lists, part, C, nb, B, D, sch);  \
        } else {                                                                                                  \
                        hipLaunchKernelGGL((cbp_fused_kernel<H, false>), grid, dim3(512), 0, st, x, lists, part, C, nb, B, D, sch); \
        }                                                                                                         \
        break;
    switch (HW) {
        HK_CBF_GO(196)
        HK_CBF_GO(144)
        HK_CBF_GO(100)
        HK_CBF_GO(64)
        default: return HK_ERR_UNSUPPORTED;
    }
#undef HK_CBF_GO
    HK_LAUNCH_CHECK();
    return HK_OK;
}

}  // namespace hk
