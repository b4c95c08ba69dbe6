// k_quadtree.hip -- ORBextractor::DistributeOctTree (ORBextractor.cc:539-763) on the device.
//
// The reference walks a std::list of nodes, pushes children to the FRONT of the list and
// erases parents.  Two facts make an exact data-parallel restatement possible:
//   (1) every new node goes to the front and nodes never move, so the list is always sorted
//       by creation time, newest first.  "Position in the list" and "creation order" are the
//       same thing, and the tie-break of the final phase (equal-sized nodes: newest first --
//       the deterministic rule this build documents) is "smaller position";
//   (2) a child's box depends only on its parent's box, so every key can route itself.
// One breadth pass of the reference (:594-666) therefore becomes: histogram keys into the four
// children of their node, prefix-sum the non-empty child counts over the nodes in list order,
// and write the new list as [children of the last parent ... children of the first parent]
// followed by the untouched single-key nodes.  The final phase (:671-737) sorts the expandable
// nodes by (size desc, position asc), prefix-sums how many nodes each split adds, cuts where
// the total reaches N, and rebuilds the list the same way.
//
// One workgroup per (image, level); node tables live in LDS, keys and their node index in HBM
// scratch (L2-resident: a level holds a few thousand candidates).
#include "gfo_internal.h"
#include <stdlib.h>

#define QT_MAX_THREADS 1024
#define QT_THREADS ((int)blockDim.x)   // 256 (small quotas) or 1024 (large ones): see gfo_launch_quadtree

struct QtBox {
    short ulx, uly, urx, bry;
};

// exclusive scan of vals[0..n) in place (LDS), returns the total; all threads call it.
// Per-thread chunk sums are scanned inside each wave with lane shuffles; only the four wave totals
// cross a workgroup barrier (3 barriers per scan).
__device__ __forceinline__ int qt_wave_incl_scan(int v)
{
    // inclusive scan over the 64 lanes in the DPP network (row shifts, then the row_bcast:15 / row_bcast:31
    // steps): six dependent VALU operations instead of six LDS round trips of a shuffle-based scan
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2 and 3
    return v;
}

__device__ int qt_scan(int* vals, int n, int* part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + QT_THREADS - 1) / QT_THREADS;
    const int b = tid * chunk, e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; i++) s += vals[i];
    const int incl = qt_wave_incl_scan(s);
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < QT_THREADS / 64; w++) {
        const int t = part[w];
        if (w < wave) wbase += t;
        total += t;
    }
    int run = wbase + incl - s;
    for (int i = b; i < e; i++) {
        const int v = vals[i];
        vals[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

// Two exclusive scans over the same index range in one pass: the values ride in the two halves of a word
// (totals stay below 2^16: the caller checks 4 * ncap < 65536, otherwise it scans twice).
__device__ int2 qt_scan2(int* a, int* b2, int n, int* part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + QT_THREADS - 1) / QT_THREADS;
    const int b = tid * chunk, e = min(b + chunk, n);
    unsigned s = 0;
    for (int i = b; i < e; i++) s += (unsigned)a[i] | ((unsigned)b2[i] << 16);
    const unsigned incl = (unsigned)qt_wave_incl_scan((int)s);
    if (lane == 63) part[wave] = (int)incl;
    __syncthreads();
    unsigned wbase = 0, total = 0;
    for (int w = 0; w < QT_THREADS / 64; w++) {
        const unsigned t = (unsigned)part[w];
        if (w < wave) wbase += t;
        total += t;
    }
    unsigned run = wbase + incl - s;
    for (int i = b; i < e; i++) {
        const unsigned v = (unsigned)a[i] | ((unsigned)b2[i] << 16);
        a[i] = (int)(run & 0xFFFFu);
        b2[i] = (int)(run >> 16);
        run += v;
    }
    __syncthreads();
    return make_int2((int)(total & 0xFFFFu), (int)(total >> 16));
}

__device__ __forceinline__ int qt_quadrant(uint32_t key, QtBox b)
{
    // ExtractorNode::DivideNode, ORBextractor.cc:483-484,515-525
    const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
    const int midx = b.ulx + (int)ceilf((float)(b.urx - b.ulx) / 2);
    const int midy = b.uly + (int)ceilf((float)(b.bry - b.uly) / 2);
    return (x < midx ? 0 : 1) + (y < midy ? 0 : 2);  // 0=n1 1=n2 2=n3 3=n4
}

__device__ __forceinline__ QtBox qt_child_box(QtBox b, int q)
{
    const short midx = (short)(b.ulx + (int)ceilf((float)(b.urx - b.ulx) / 2));
    const short midy = (short)(b.uly + (int)ceilf((float)(b.bry - b.uly) / 2));
    QtBox c;
    c.ulx = (q & 1) ? midx : b.ulx;
    c.urx = (q & 1) ? b.urx : midx;
    c.uly = (q & 2) ? midy : b.uly;
    c.bry = (q & 2) ? b.bry : midy;
    return c;
}

__global__ __launch_bounds__(QT_MAX_THREADS) void k_quadtree(const GfoGeom* __restrict__ gp,
                                                         const uint32_t* __restrict__ cand,
                                                         const int* __restrict__ cand_cnt,
                                                         uint16_t* __restrict__ node_of_all,
                                                         uint32_t* __restrict__ sel, int* __restrict__ sel_cnt,
                                                         int* __restrict__ flags, int ncap, int klds,
                                                         unsigned long long* __restrict__ dbg_ts)
{
    // debugging aid (GFO_QT_TIMING=1): thread 0 of block (0, 0) stamps the constant-rate clock at phase boundaries
    int ts_i = 0;
#define QT_TS(label)                                                                              \
    if (dbg_ts && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && ts_i < 120)          \
    dbg_ts[ts_i++] = (wall_clock64() << 8) | (unsigned)(label)

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const GfoGeom& g = *gp;
    const int level = blockIdx.y, img = blockIdx.x, tid = threadIdx.x;  // level-major dispatch: the heavy level-0 blocks start first
    const GfoLevel& L = g.lv[level];
    const int N = L.quota;
    const int K = min(cand_cnt[(img * g.nlevels + level) * GFO_CNT_STRIDE], L.cand_cap);
    const uint32_t* gkeys = cand + (long long)img * g.cand_img_stride + L.cand_off;
    uint16_t* gnode = node_of_all + (long long)img * g.cand_img_stride + L.cand_off;
    int* out_cnt = sel_cnt + img * g.nlevels + level;
    uint32_t* out = sel + (long long)img * g.total_sel_cap + L.sel_off;
    if (K == 0) {
        if (tid == 0) *out_cnt = 0;
        return;
    }
    QT_TS(1);
    // LDS carve-up (ncap entries each unless noted)
    unsigned long long* srt = reinterpret_cast<unsigned long long*>(lds);   // sort keys / best keys, pow2(ncap)
    int p2 = 1;
    while (p2 < ncap) p2 <<= 1;
    QtBox* boxA = reinterpret_cast<QtBox*>(srt + p2);
    QtBox* boxB = boxA + ncap;
    int* cntA = reinterpret_cast<int*>(boxB + ncap);
    int* cntB = cntA + ncap;
    int* cc = cntB + ncap;            // 4*ncap child counts
    int* cpos = cc + 4 * ncap;        // 4*ncap child positions in the new list
    int* sc1 = cpos + 4 * ncap;       // scan buffers
    int* sc2 = sc1 + ncap;
    int* npos = sc2 + ncap;           // new position of an unsplit node
    int* part = npos + ncap;          // QT_THREADS
    __shared__ int s_cut, s_misc;
    // Keys and their node index live in LDS when the level's candidate list fits (the usual case: a few
    // thousand keys); every pass then runs at LDS latency.  Larger lists stay in HBM/L2 (same code, generic
    // pointers).
    uint32_t* lkeys = reinterpret_cast<uint32_t*>(part + QT_MAX_THREADS);
    uint16_t* lnode = reinterpret_cast<uint16_t*>(lkeys + klds);
    const bool in_lds = K <= klds;
    const uint32_t* keys = in_lds ? lkeys : gkeys;
    uint16_t* node_of = in_lds ? lnode : gnode;
    if (in_lds) {
        for (int k = tid; k < K; k += QT_THREADS) lkeys[k] = gkeys[k];
    }

    // ---- roots (:543-585) ----
    const int nini = L.n_ini;
    const float hx = L.hx;
    for (int i = tid; i < nini; i += QT_THREADS) {
        QtBox b;
        b.ulx = (short)(int)(hx * (float)i);
        b.urx = (short)(int)(hx * (float)(i + 1));
        b.uly = 0;
        b.bry = (short)(L.max_by - GFO_MIN_BORDER);
        boxB[i] = b;
        cntB[i] = 0;
    }
    __syncthreads();
    for (int k = tid; k < K; k += QT_THREADS) {
        const int x = keys[k] & 0xFFF;
        int r = (int)((float)x / hx);
        r = min(r, nini - 1);
        node_of[k] = (uint16_t)r;
        atomicAdd(&cntB[r], 1);
    }
    __syncthreads();
    for (int i = tid; i < nini; i += QT_THREADS) sc1[i] = cntB[i] > 0 ? 1 : 0;
    __syncthreads();
    int n = qt_scan(sc1, nini, part);
    for (int i = tid; i < nini; i += QT_THREADS)
        if (cntB[i] > 0) {
            boxA[sc1[i]] = boxB[i];
            cntA[sc1[i]] = cntB[i];
        }
    __syncthreads();
    for (int k = tid; k < K; k += QT_THREADS) node_of[k] = (uint16_t)sc1[node_of[k]];
    __syncthreads();

    QT_TS(2);
    QtBox* box = boxA;
    int* cnt = cntA;
    QtBox* boxn = boxB;
    int* cntn = cntB;
    bool phase2 = false;
    int xlen = 0;  // phase 2: the expandable nodes are among positions [0, xlen)

    for (int iter = 0; iter < 64; iter++) {
        // ---- child histograms of the nodes that may be split in this pass ----
        QT_TS(3);
        const int lim = phase2 ? xlen : n;
        for (int i = tid; i < 4 * lim; i += QT_THREADS) cc[i] = 0;
        __syncthreads();
        for (int k = tid; k < K; k += QT_THREADS) {
            const int i = node_of[k];
            if (i < lim && cnt[i] > 1) atomicAdd(&cc[4 * i + qt_quadrant(keys[k], box[i])], 1);
        }
        __syncthreads();
        QT_TS(4);
        int m_split;  // number of nodes split in this pass
        if (!phase2) {
            // every node with more than one key is split, in list order (:606-665)
            for (int i = tid; i < n; i += QT_THREADS) {
                int nch = 0;
                if (cnt[i] > 1) nch = (cc[4 * i] > 0) + (cc[4 * i + 1] > 0) + (cc[4 * i + 2] > 0) + (cc[4 * i + 3] > 0);
                sc1[i] = nch;
                sc2[i] = cnt[i] > 1 ? 0 : 1;
            }
            __syncthreads();
            int ctot, nsingle;
            if (4 * ncap < 65536) {
                const int2 t2 = qt_scan2(sc1, sc2, n, part);
                ctot = t2.x;
                nsingle = t2.y;
            } else {
                ctot = qt_scan(sc1, n, part);
                nsingle = qt_scan(sc2, n, part);
            }
            QT_TS(5);
            if (ctot + nsingle > ncap) {
                if (tid == 0) { atomicOr(&flags[0], 2); *out_cnt = 0; }
                return;
            }
            for (int i = tid; i < n; i += QT_THREADS) {
                if (cnt[i] > 1) {
                    int r = 0;
                    for (int q = 0; q < 4; q++)
                        if (cc[4 * i + q] > 0) {
                            const int pos = ctot - 1 - (sc1[i] + r);
                            boxn[pos] = qt_child_box(box[i], q);
                            cntn[pos] = cc[4 * i + q];
                            cpos[4 * i + q] = pos;
                            r++;
                        }
                } else {
                    const int pos = ctot + sc2[i];
                    boxn[pos] = box[i];
                    cntn[pos] = cnt[i];
                    npos[i] = pos;
                }
            }
            __syncthreads();
            QT_TS(7);
            for (int k = tid; k < K; k += QT_THREADS) {
                const int i = node_of[k];
                node_of[k] = (uint16_t)(cnt[i] > 1 ? cpos[4 * i + qt_quadrant(keys[k], box[i])] : npos[i]);
            }
            __syncthreads();
            QT_TS(8);
            const int prev = n;
            n = ctot + nsingle;
            xlen = ctot;
            QtBox* tb = box; box = boxn; boxn = tb;
            int* tc = cnt; cnt = cntn; cntn = tc;
            // nToExpand = new children holding more than one key (:618-664)
            if (tid == 0) s_misc = 0;
            __syncthreads();
            int e = 0;
            for (int i = tid; i < xlen; i += QT_THREADS) e += cnt[i] > 1 ? 1 : 0;
            if (e) atomicAdd(&s_misc, e);
            __syncthreads();
            const int n_to_expand = s_misc;
            __syncthreads();
            if (n >= N || n == prev) break;              // :667-670
            if (n + n_to_expand * 3 > N) phase2 = true;  // :671
            continue;
        }
        // ---- final phase: split the largest nodes first until N is reached (:673-735) ----
        // Order the expandable nodes by (size desc, position asc).  Keys are distinct (the position is part of
        // them), so the sorted slot of a key is simply the number of larger keys.  For the few hundred nodes of
        // the usual quotas every thread counts them directly -- xlen broadcast reads of LDS, one barrier -- which
        // takes a fifth of the time of the 45 dependent compare-exchange rounds of a bitonic network (each round
        // is a full LDS round trip); large lists (1080p, 4000 features) keep the network.
        const int per_thread = (xlen + QT_THREADS - 1) / QT_THREADS;
        if (xlen * per_thread <= 2048 && K < 65536) {
            // 32-bit keys (size << 16 | inverted position): sizes are bounded by the candidate count, and a level
            // with 65536 or more candidates takes the 64-bit network below
            // (cpos: 4*ncap ints, free until the rebuild; rounded up to a 16-byte boundary for the wide reads)
            unsigned* keyv = reinterpret_cast<unsigned*>((reinterpret_cast<uintptr_t>(cpos) + 15) & ~(uintptr_t)15);
            const int xlen8 = (xlen + 7) & ~7;   // entries [xlen, xlen8) are zero: never larger than a key
            for (int i = tid; i < xlen8; i += QT_THREADS)
                keyv[i] = i < xlen && cnt[i] > 1 ? ((unsigned)cnt[i] << 16) | (unsigned)(0xFFFF - i) : 0u;
            for (int i = tid; i < p2; i += QT_THREADS) srt[i] = 0;
            __syncthreads();
            // two own keys per sweep; the sweep reads eight list entries at a time (two 16-byte broadcast loads in
            // flight together -- one load per iteration would make every step a full LDS round trip)
            const uint4* kv4 = reinterpret_cast<const uint4*>(keyv);
            for (int i = tid; i < xlen; i += 2 * QT_THREADS) {
                const unsigned m0 = keyv[i], m1 = i + QT_THREADS < xlen ? keyv[i + QT_THREADS] : 0u;
                int r0 = 0, r1 = 0;
                for (int j = 0; j < xlen8 / 4; j += 2) {
                    const uint4 a = kv4[j], b = kv4[j + 1];
                    const unsigned kj[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        r0 += kj[u] > m0 ? 1 : 0;
                        r1 += kj[u] > m1 ? 1 : 0;
                    }
                }
                // the consumers read 64-bit entries: size << 16 | inverted position, same layout
                if (m0) srt[r0] = m0;
                if (m1) srt[r1] = m1;
            }
            __syncthreads();
        } else {
        for (int i = tid; i < p2; i += QT_THREADS) {
            unsigned long long key = 0;
            if (i < xlen && cnt[i] > 1) key = ((unsigned long long)(unsigned)cnt[i] << 16) | (unsigned)(0xFFFF - i);
            srt[i] = key;
        }
        __syncthreads();
        for (int k2 = 2; k2 <= p2; k2 <<= 1)
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < p2; i += QT_THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long a = srt[i], b = srt[ixj];
                        const bool desc = (i & k2) == 0;
                        if (desc ? a < b : a > b) { srt[i] = b; srt[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        }
        QT_TS(6);
        // sorted descending: expandable nodes first, (size desc, position asc)
        for (int i = tid; i < xlen; i += QT_THREADS) {
            const unsigned long long key = srt[i];
            int nch = 0;
            if (key) {
                const int node = 0xFFFF - (int)(key & 0xFFFF);
                nch = (cc[4 * node] > 0) + (cc[4 * node + 1] > 0) + (cc[4 * node + 2] > 0) + (cc[4 * node + 3] > 0);
            }
            sc1[i] = nch;                 // children created by the t-th split
            sc2[i] = key ? nch - 1 : 0;   // growth of the list
        }
        __syncthreads();
        if (4 * ncap < 65536) {
            qt_scan2(sc1, sc2, xlen, part);
        } else {
            qt_scan(sc1, xlen, part);
            qt_scan(sc2, xlen, part);  // exclusive: growth before split t
        }
        if (tid == 0) s_cut = xlen, s_misc = 0;
        __syncthreads();
        // first t whose split brings the list to >= N (:730); splits t = 0..cut inclusive happen
        for (int i = tid; i < xlen; i += QT_THREADS) {
            const unsigned long long key = srt[i];
            if (key) {
                const int node = 0xFFFF - (int)(key & 0xFFFF);
                const int nch = (cc[4 * node] > 0) + (cc[4 * node + 1] > 0) + (cc[4 * node + 2] > 0) + (cc[4 * node + 3] > 0);
                if (n + sc2[i] + nch - 1 >= N) atomicMin(&s_cut, i);
                atomicAdd(&s_misc, 1);
            }
        }
        __syncthreads();
        const int nexp = s_misc;
        m_split = min(s_cut + 1, nexp);
        __syncthreads();
        // totals over the split prefix
        int ctot = 0, growth = 0;
        if (m_split > 0) {
            const unsigned long long key = srt[m_split - 1];
            const int node = 0xFFFF - (int)(key & 0xFFFF);
            const int nch = (cc[4 * node] > 0) + (cc[4 * node + 1] > 0) + (cc[4 * node + 2] > 0) + (cc[4 * node + 3] > 0);
            ctot = sc1[m_split - 1] + nch;
            growth = sc2[m_split - 1] + nch - 1;
        }
        if (n + growth > ncap) {
            if (tid == 0) { atomicOr(&flags[0], 2); *out_cnt = 0; }
            return;
        }
        // mark split nodes, rank the others in old order
        for (int i = tid; i < n; i += QT_THREADS) npos[i] = 1;  // 1 = kept
        __syncthreads();
        for (int t = tid; t < m_split; t += QT_THREADS) npos[0xFFFF - (int)(srt[t] & 0xFFFF)] = 0;
        __syncthreads();
        // children of the t-th split go to ctot-1-(P_t + r)
        for (int t = tid; t < m_split; t += QT_THREADS) {
            const int node = 0xFFFF - (int)(srt[t] & 0xFFFF);
            int r = 0;
            for (int q = 0; q < 4; q++)
                if (cc[4 * node + q] > 0) {
                    const int pos = ctot - 1 - (sc1[t] + r);
                    boxn[pos] = qt_child_box(box[node], q);
                    cntn[pos] = cc[4 * node + q];
                    cpos[4 * node + q] = pos;
                    r++;
                }
        }
        __syncthreads();
        for (int i = tid; i < n; i += QT_THREADS) sc2[i] = npos[i];
        __syncthreads();
        qt_scan(sc2, n, part);
        for (int i = tid; i < n; i += QT_THREADS)
            if (npos[i]) {
                const int pos = ctot + sc2[i];
                boxn[pos] = box[i];
                cntn[pos] = cnt[i];
                sc1[i] = pos;  // sc1 is free again: new position of kept node i
            } else sc1[i] = -1;
        __syncthreads();
        QT_TS(7);
        for (int k = tid; k < K; k += QT_THREADS) {
            const int i = node_of[k];
            node_of[k] = (uint16_t)(sc1[i] >= 0 ? sc1[i] : cpos[4 * i + qt_quadrant(keys[k], box[i])]);
        }
        __syncthreads();
        QT_TS(8);
        const int prev = n;
        n = n + growth;
        xlen = ctot;
        QtBox* tb = box; box = boxn; boxn = tb;
        int* tc = cnt; cnt = cntn; cntn = tc;
        if (n >= N || n == prev) break;  // :733-734
    }

    // ---- keep the best response of every node, first in the reference's key order on ties
    //      (:740-760).  Key order = cell-major then row-major: rank = (cell_i, cell_j, y, x).
    QT_TS(9);
    for (int i = tid; i < n; i += QT_THREADS) srt[i] = 0;
    __syncthreads();
    for (int k = tid; k < K; k += QT_THREADS) {
        const uint32_t key = keys[k];
        const unsigned x = key & 0xFFF, y = (key >> 12) & 0xFFF, s = key >> 24;
        const unsigned ci = (y - 3) / (unsigned)L.hcell, cj = (x - 3) / (unsigned)L.wcell;
        const unsigned long long rank = ((unsigned long long)ci << 36) | ((unsigned long long)cj << 24) | (y << 12) | x;
        const unsigned long long v = ((unsigned long long)(s + 1) << 48) | (0xFFFFFFFFFFFFull ^ rank);
        atomicMax(&srt[node_of[k]], v);
    }
    __syncthreads();
    if (n > L.sel_cap) {
        if (tid == 0) { atomicOr(&flags[0], 4); *out_cnt = 0; }
        return;
    }
    for (int i = tid; i < n; i += QT_THREADS) {
        const unsigned long long v = srt[i];
        const unsigned long long rank = 0xFFFFFFFFFFFFull ^ (v & 0xFFFFFFFFFFFFull);
        const unsigned s = (unsigned)(v >> 48) - 1;
        out[i] = (uint32_t)(rank & 0xFFFFFF) | (s << 24);
    }
    QT_TS(10);
    if (tid == 0) *out_cnt = n;
#undef QT_TS
}

size_t gfo_quadtree_lds_bytes(int ncap, int klds)
{
    int p2 = 1;
    while (p2 < ncap) p2 <<= 1;
    return (size_t)p2 * 8 + (size_t)ncap * (2 * sizeof(QtBox) + 2 * 4 + 4 * 4 + 4 * 4 + 3 * 4) + QT_MAX_THREADS * 4 + (size_t)klds * 6 + 64;
}

void gfo_launch_quadtree(gfo_ctx* c, int nimg)
{
    int ncap = 0;
    for (int l = 0; l < c->g.nlevels; l++) ncap = c->g.lv[l].node_cap > ncap ? c->g.lv[l].node_cap : ncap;
    // keys kept in LDS (6 B each).  Measured on MI355X: 0 (keys in L2, 4 workgroups per CU) beats 6144
    // (LDS-resident keys, 2 workgroups per CU) -- the kernel is barrier-bound, not load-bound.
    static const int klds_env = getenv("GFO_QT_KLDS") ? atoi(getenv("GFO_QT_KLDS")) : 0;
    int klds = klds_env;
    while (klds > 0 && gfo_quadtree_lds_bytes(ncap, klds) > 150 * 1024) klds -= 1024;
    const size_t lds = gfo_quadtree_lds_bytes(ncap, klds);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(nimg, c->g.nlevels);
    // the per-pass key loops are latency-bound inside a workgroup: large quotas (1080p @4000 features) get
    // 1024 threads per (image, level), the 752x480 @2000 case runs best with 256
    static const int nt_env = getenv("GFO_QT_THREADS") ? atoi(getenv("GFO_QT_THREADS")) : 0;
    const int nthreads = nt_env ? nt_env : (c->g.lv[0].quota >= 600 ? 1024 : 256);
    static const bool timing = getenv("GFO_QT_TIMING") != nullptr;
    unsigned long long* d_ts = nullptr;
    if (timing && hipMalloc(&d_ts, 128 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemsetAsync(d_ts, 0, 128 * sizeof(unsigned long long), c->stream);
    gfo_prof_begin(c, ST_QUADTREE);
    hipLaunchKernelGGL(k_quadtree, grid, dim3(nthreads), lds, c->stream, c->d_geom, c->d_cand, c->d_cand_cnt,
                       c->d_node_of, c->d_sel, c->d_sel_cnt, c->d_flags, ncap, klds, d_ts);
    if (d_ts) {   // debugging aid: blocks until the kernel is done and prints the phase times of block (0, 0)
        unsigned long long ts[128];
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpy(ts, d_ts, sizeof ts, hipMemcpyDeviceToHost);
        (void)hipFree(d_ts);
        fprintf(stderr, "[gfo] quadtree block (0,0), 10 ns ticks since start:");
        for (int i = 0; i < 128 && ts[i]; i++) fprintf(stderr, " %llu:%llu", ts[i] & 255, (ts[i] >> 8) - (ts[0] >> 8));
        fprintf(stderr, "\n");
    }
    gfo_prof_end(c);
}
