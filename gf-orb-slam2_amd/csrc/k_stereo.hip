// k_stereo.hip -- Frame::PrepareStereoCandidates + ComputeStereoMatches_Undistorted(false)
// (Frame.h:230-263, Frame.cc:1167-1316; ALTER_STEREO_MATCHING path), plus
// ORBmatcher::DescriptorDistance (ORBmatcher.cc:1768-1784) as XOR + popcount.
//
// The reference builds a row -> right-keypoint table and scans one row's list per left keypoint.
// Membership of right keypoint iR in row `r` is the pure predicate minr(iR) <= r <= maxr(iR), and
// the scan keeps the FIRST minimum in iR order, i.e. the lexicographic minimum of (distance, iR).
// Device form:
//   k_stereo_bucket : one workgroup per pair counting-sorts the right keypoints by floor(y)
//                     (LDS histogram + scan) and writes them as a compact SoA in bucket order
//                     (x, y, octave|iR, 32-byte descriptor), so the matcher reads contiguous memory;
//   k_stereo_match_rows : the association around ROWS (round 3).  k_stereo_bucket also counting-sorts the LEFT keypoints
//                     by their row; a workgroup owns a band of R rows of one pair, stages -- once, coalesced -- the
//                     right-side records of rows [r0 - W, r1 + W] (band, octave | iR, x, 32-byte descriptor) and the
//                     left keypoints of its rows in LDS, and every 32-lane half-wave then sweeps LDS for one left
//                     keypoint at a time: band / octave / disparity-window predicates from one 16-byte LDS read per
//                     candidate, the 256-bit Hamming distance (v_bcnt) of the survivors, half-wave reduction of
//                     min(dist << 16 | iR)
//                     (W = ceil(2 * max scale) + 1 covers every band that can contain a row).  Before, left keypoints
//                     arrived in level / list order, every half-wave chased four dependent global loads per candidate
//                     and shared nothing with its neighbours: 80 % of the wave cycles were waits and the kernel moved
//                     1.9 x its algorithmic bytes;
//   k_stereo_match  : that earlier form -- one half-wave per left keypoint straight from the bucketed arrays -- kept
//                     for windows too wide for the LDS row table and as GFO_STEREO_ROWS=0;
//   k_stereo_cut    : the outlier cut (:1290-1313) needs only the (ndi/2)-th order statistic of the
//                     accepted distances: a 128-bin LDS histogram gives it exactly (no sort).
#include "gfo_internal.h"
#include <stdlib.h>

#define TH_HIGH 100  // ORBmatcher.cc:57
#define TH_LOW 50    // ORBmatcher.cc:58

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

struct StereoArgs {
    const gfo_keypoint* kl;
    const uint8_t* dl;
    const gfo_keypoint* kr;
    const uint8_t* dr;
    const int* cnt_dev;       // [2*pairs] (left, right) counts, or null -> host counts
    int nl_host, nr_host;
    long long pair_stride;    // keypoints between consecutive pairs in kl/kr (and x32 bytes in dl/dr)
    const float* scale;
    gfo_stereo_params p;
    const float* min_d;
    const float* max_d;
    long long win_stride;     // floats between consecutive pairs in min_d / max_d (0: one pair)
    GfoStereoDev out;
    int out_stride;
    // bucketed right side, [pairs][sort_stride]
    float* sx;
    unsigned* sband;          // row band minr | maxr << 16 of the bucketed right keypoint
    unsigned* soi;            // octave << 16 | iR
    uint4* sdesc;             // 2 x uint4 per keypoint
    int* row_start;           // [pairs][n_rows + 1]
    int* lorder;              // left keypoints with a row inside the image, sorted by row: original indices
    int* lrow_start;          // [pairs][n_rows + 1]
    int sort_stride;
    int window;               // W
    int nlevels;              // entries of `scale`
};

#include "k_wave.inc"   // st_row_min / st_half_min / st_wave_min / st_wave_sum / st_wave_incl_scan / st_block_incl_scan

// blockIdx.y = 0: the right keypoints of the pair by floor(y); blockIdx.y = 1 (row form only): the left ones by row.
// The two counting sorts are independent -- a workgroup each, side by side (one after the other in one workgroup: 20 us
// per 64 pairs instead of 16 for the right side alone; the launch is 64 workgroups of latency chains either way).
__global__ __launch_bounds__(1024) void k_stereo_bucket(StereoArgs a)
{
    extern __shared__ int s_hist[];  // n_rows + 256
    const int pair = blockIdx.x, tid = threadIdx.x;
    const bool left = blockIdx.y == 1;
    const int nRows = a.p.n_rows;
    int* s_part = s_hist + nRows;
    const int n = a.cnt_dev ? a.cnt_dev[2 * pair + (left ? 0 : 1)] : (left ? a.nl_host : a.nr_host);
    const gfo_keypoint* kp = (left ? a.kl : a.kr) + pair * a.pair_stride;
    const int NT = blockDim.x;   // 1024: the count / scatter loops are chains of dependent global loads, so more
                                 // threads means fewer serial round trips (8 -> 2 for 2000 keypoints)
    for (int r = tid; r < nRows; r += NT) s_hist[r] = 0;
    __syncthreads();
    const long long oo = (long long)pair * a.out_stride;
    for (int i = tid; i < n; i += NT) {
        const float y = kp[i].y;
        if (!left) {
            atomicAdd(&s_hist[min(max((int)floorf(y), 0), nRows - 1)], 1);
        } else if (y < 0 || y > (float)(nRows - 1)) {
            // Frame.cc:1204-1211: row = (int)vL, a left keypoint with vL outside [0, nRows - 1] is skipped: the defaults
            a.out.u_right[oo + i] = -1.0f;
            a.out.depth[oo + i] = -1.0f;
            a.out.best_dist[oo + i] = -1;
            a.out.best_idx[oo + i] = -1;
            a.out.counted[oo + i] = 0;
        } else {
            atomicAdd(&s_hist[(int)y], 1);
        }
    }
    __syncthreads();
    // exclusive scan over nRows entries
    const int chunk = (nRows + NT - 1) / NT;
    const int b0 = min(tid * chunk, nRows), e0 = min(b0 + chunk, nRows);
    int s = 0;
    for (int r = b0; r < e0; r++) s += s_hist[r];
    int total;
    int run = st_block_incl_scan(s, s_part, &total) - s;
    int* rs = (left ? a.lrow_start : a.row_start) + (long long)pair * (nRows + 1);
    for (int r = b0; r < e0; r++) {
        const int v = s_hist[r];
        rs[r] = run;
        s_hist[r] = run;  // fill cursor
        run += v;
    }
    if (tid == 0) rs[nRows] = total;
    __syncthreads();
    const long long so = (long long)pair * a.sort_stride;
    if (left) {
        for (int i = tid; i < n; i += NT) {
            const float y = kp[i].y;
            if (!(y < 0 || y > (float)(nRows - 1))) a.lorder[so + atomicAdd(&s_hist[(int)y], 1)] = i;
        }
        return;
    }
    const uint4* dr = reinterpret_cast<const uint4*>(a.dr + pair * a.pair_stride * 32);
    for (int i = tid; i < n; i += NT) {
        const gfo_keypoint k = kp[i];
        const int b = min(max((int)floorf(k.y), 0), nRows - 1);
        const int pos = atomicAdd(&s_hist[b], 1);
        a.sx[so + pos] = k.x;
        // the row band of this right keypoint (Frame.h:248-256), once here instead of once per left keypoint that visits it
        const float rr = 2.0f * a.scale[k.octave];
        int maxr = (int)fminf((float)(nRows - 1), ceilf(k.y + rr));
        int minr = (int)fmaxf(0.0f, floorf(k.y - rr));
        // a keypoint whose band misses the image (host arrays may carry any y: maxr < 0 or minr > nRows - 1) enters no
        // row in the reference -- `for (yi = minr; yi <= maxr; yi++)` does not run, Frame.h:256 -- and must not wrap
        // around in the packed form: stored as the empty band (1, 0)
        if (minr > maxr) { minr = 1; maxr = 0; }
        a.sband[so + pos] = (unsigned)minr | ((unsigned)maxr << 16);   // n_rows <= 65535 (checked by the launcher)
        a.soi[so + pos] = ((unsigned)k.octave << 16) | (unsigned)i;
        a.sdesc[2 * (so + pos)] = dr[2 * i];
        a.sdesc[2 * (so + pos) + 1] = dr[2 * i + 1];
    }
}

// Two left keypoints per wavefront, one per 32-lane half: the work per keypoint is a short chain of
// dependent loads (keypoint -> bucket range -> ~80 candidates), so the kernel is latency-bound and two
// independent chains per wave halve the waves to retire.
__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs a)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform: keep it scalar
    const int half = lane >> 5, hl = lane & 31;
    const int pair = blockIdx.y;
    const int nl = a.cnt_dev ? a.cnt_dev[2 * pair] : a.nl_host;
    const int iL0 = (blockIdx.x * (int)(blockDim.x >> 6) + wave) * 2;
    if (iL0 >= nl) return;                         // wave-uniform
    const bool act = iL0 + half < nl;
    const int iL = min(iL0 + half, nl - 1);        // an idle second half redoes the last keypoint, writes nothing
    const gfo_keypoint* kl = a.kl + pair * a.pair_stride;
    const uint8_t* dl = a.dl + pair * a.pair_stride * 32;
    const long long o = (long long)pair * a.out_stride + iL;
    const long long so = (long long)pair * a.sort_stride;
    float res_u = -1.0f, res_depth = -1.0f;
    int res_dist = -1, res_idx = -1, counted = 0;

    const gfo_keypoint L = kl[iL];
    const float vL = L.y, uL = L.x;
    const int nRows = a.p.n_rows;
    const bool in_rows = !(vL < 0 || vL > (float)(nRows - 1));  // Frame.cc:1208
    const int row = in_rows ? (int)vL : 0;
    float minD = 0.f, maxD = a.p.mbf / a.p.mb;  // :1199-1200 (minZ = mb)
    if (a.min_d && a.max_d) {                  // :1220-1231 flattened by the adapter
        minD = a.min_d[pair * a.win_stride + iL];
        maxD = a.max_d[pair * a.win_stride + iL];
    }
    const float minU = uL - maxD, maxU = uL - minD;
    const uint4* dlp = reinterpret_cast<const uint4*>(dl + (long long)iL * 32);
    const uint4 a0 = dlp[0], a1 = dlp[1];
    const int* rs = a.row_start + (long long)pair * (nRows + 1);
    // every entry of row_start[0..nRows] is written by k_stereo_bucket of this launch sequence (values 0..nr)
    const int jb = rs[max(row - a.window, 0)], je = in_rows ? rs[min(row + a.window + 1, nRows)] : jb;
    unsigned best = ((unsigned)TH_HIGH << 16);  // bestDist = TH_HIGH, iR = 0: only dist < TH_HIGH replaces it
    bool any = false;
    for (int j = jb + hl; j < je; j += 32) {
        const unsigned band = a.sband[so + j];   // row band of this right keypoint, packed by k_stereo_bucket
        const unsigned oi = a.soi[so + j];
        const int oct = (int)(oi >> 16);
        if (row < (int)(band & 0xFFFF) || row > (int)(band >> 16)) continue;
        any = true;
        if (oct < L.octave - 1 || oct > L.octave + 1) continue;  // :1250
        const float rx = a.sx[so + j];
        if (rx >= minU && rx <= maxU) {                          // :1255
            const unsigned dist = (unsigned)hamming256(a0, a1, a.sdesc[2 * (so + j)], a.sdesc[2 * (so + j) + 1]);
            best = min(best, (dist << 16) | (oi & 0xFFFF));      // first minimum in iR order (:1260)
        }
    }
    const unsigned long long anym = __builtin_amdgcn_ballot_w64(any);
    const bool have_cands = ((anym >> (32 * half)) & 0xFFFFFFFFull) != 0;
    best = st_half_min(best);   // inside the half
    if (in_rows && have_cands && !(maxU < a.p.min_x)) {  // :1213, :1236
        counted = 1;
        const int bestDist = (int)(best >> 16);
        const int bestIdxR = (int)(best & 0xFFFF);
        if (bestDist < (TH_HIGH + TH_LOW) / 2) {  // :1269
            float bestuR = (a.kr + pair * a.pair_stride)[bestIdxR].x;
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01f;
                    bestuR = (float)((double)uL - 0.01);   // `uL-0.01` is double arithmetic in the reference (Frame.cc:1278)
                }
                res_depth = a.p.mbf / disparity;
                res_u = bestuR;
                res_dist = bestDist;
                res_idx = bestIdxR;
            }
        }
    }
    if (hl == 0 && act) {
        a.out.u_right[o] = res_u;
        a.out.depth[o] = res_depth;
        a.out.best_dist[o] = res_dist;
        a.out.best_idx[o] = res_idx;
        a.out.counted[o] = (unsigned char)counted;  // summed per pair by k_stereo_cut (no same-line atomics)
    }
}

// ---- one to three pairs: the association straight from the extractor's arrays, no bucketing launch (round 5) ----
// The per-frame path (Frame::Frame: ONE stereo pair per call) is a chain of launches of 7-12 us each; k_stereo_bucket was one of
// them (9-12 us: two workgroups of latency chains) only to hand k_stereo_match a row-sorted copy.  Here every workgroup first
// packs ALL right keypoints into LDS as 8-byte records {row band minr | maxr << 16 (Frame.h:248-256), x} -- nr x 28 bytes out
// of L2, ~15 vector operations per keypoint and thread -- and each wavefront then sweeps that table for its left keypoints:
// the band test (97 % of the table fails it) costs one 8-byte LDS read and two compares per candidate, octave / disparity
// window / Hamming only for the band's members, read from the arrays where the extractor left them.  Candidates are visited
// in iR order by construction (lane j, j + 64, ...; the wave minimum of dist << 16 | iR is the first minimum, :1260).
// Survivors of band, octave and disparity window are compacted (ballot) into a per-wave list and scored on full lanes, so
// the sweep itself holds no global load.  Measured per stereo frame (profiles/stereo_direct_r05.txt): staging 4.6 us +
// sweep and scoring 5.6 us = 10.2 us against 9.2 (bucket) + 7.3 (match); ONE left keypoint per wavefront and eight
// wavefronts per workgroup: a lone wave per SIMD retires a dependent instruction every ~9 cycles, so instructions per wave --
// not total work -- set the time (four keypoints per wave, swept together: 24 us; two, one after the other: 20-30 us).
#define SD_KP 1        // left keypoints per wavefront
#define SD_WAVES 8     // wavefronts per workgroup (512 threads: 251 workgroups for 2008 left keypoints)
#define SD_NT (64 * SD_WAVES)
#define SD_LIST 128    // survivor slots per wavefront and keypoint (flushed when fewer than 64 are free)
__global__ __launch_bounds__(SD_NT) void k_stereo_match_direct(StereoArgs a)
{
    extern __shared__ uint4 s_rec[];   // [nr] {minr | maxr << 16, x, octave, -}
    __shared__ float s_scale[GFO_MAX_LEVELS];
    __shared__ unsigned short s_list[SD_WAVES][SD_KP][SD_LIST];
    const int pair = blockIdx.y, tid = threadIdx.x;
    const int nl = a.cnt_dev ? a.cnt_dev[2 * pair] : a.nl_host;
    const int nr = a.cnt_dev ? a.cnt_dev[2 * pair + 1] : a.nr_host;
    if ((int)blockIdx.x * SD_WAVES * SD_KP >= nl) return;     // workgroup-uniform, before any barrier
    const int nRows = a.p.n_rows;
    const gfo_keypoint* kr = a.kr + pair * a.pair_stride;
    if (tid < GFO_MAX_LEVELS) s_scale[tid] = tid < a.nlevels ? a.scale[tid] : 0.f;
    // this wave's left keypoints, requested before the staging below so that their round trips overlap it
    struct LeftKp { int iL; bool valid, in_rows; float vL, uL, minD, maxD; int oct; uint4 a0, a1; } Lv[SD_KP];
    {
        const gfo_keypoint* kl = a.kl + pair * a.pair_stride;
        const uint8_t* dl = a.dl + pair * a.pair_stride * 32;
        const int w_ = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int q = 0; q < SD_KP; q++) {
            const int iL = ((int)blockIdx.x * SD_WAVES + w_) * SD_KP + q;
            Lv[q].valid = iL < nl;
            Lv[q].iL = min(iL, nl - 1);                  // an idle slot redoes the last keypoint, writes nothing
            const gfo_keypoint L = kl[Lv[q].iL];
            Lv[q].vL = L.y; Lv[q].uL = L.x; Lv[q].oct = L.octave;
            Lv[q].in_rows = Lv[q].valid && !(L.y < 0 || L.y > (float)(nRows - 1));  // Frame.cc:1208
            Lv[q].minD = 0.f; Lv[q].maxD = a.p.mbf / a.p.mb;                         // :1199-1200
            if (a.min_d && a.max_d) {                                                // :1220-1231 flattened by the adapter
                Lv[q].minD = a.min_d[pair * a.win_stride + Lv[q].iL];
                Lv[q].maxD = a.max_d[pair * a.win_stride + Lv[q].iL];
            }
            const uint4* dlp = reinterpret_cast<const uint4*>(dl + (long long)Lv[q].iL * 32);
            Lv[q].a0 = dlp[0]; Lv[q].a1 = dlp[1];
        }
    }
    // the right keypoints' x / y / octave, four per thread in flight (the loop body is a chain of L2 round trips otherwise)
    for (int j0 = 0; j0 < nr; j0 += 4 * SD_NT) {
        float kx[4], ky[4];
        int ko[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = min(j0 + u * SD_NT + tid, nr - 1);
            kx[u] = kr[j].x; ky[u] = kr[j].y; ko[u] = kr[j].octave;
        }
        if (j0 == 0) __syncthreads();                   // s_scale
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * SD_NT + tid;
            if (j >= nr) continue;
            const float rr = 2.0f * s_scale[ko[u] & (GFO_MAX_LEVELS - 1)];
            int maxr = (int)fminf((float)(nRows - 1), ceilf(ky[u] + rr));
            int minr = (int)fmaxf(0.0f, floorf(ky[u] - rr));
            if (minr > maxr) { minr = 1; maxr = 0; }    // a band that misses the image enters no row (Frame.h:256)
            s_rec[j] = make_uint4((unsigned)minr | ((unsigned)maxr << 16), __float_as_uint(kx[u]), (unsigned)ko[u], 0u);
        }
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const uint4* dr = reinterpret_cast<const uint4*>(a.dr + pair * a.pair_stride * 32);
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));     // lanes below this one
    // the wave's SD_KP left keypoints are swept TOGETHER: one LDS read of a right record serves all of them, and every
    // dependent access (the record, later the survivors' descriptors) is a round trip paid once per wave, not per keypoint
    unsigned row[SD_KP], nsurv[SD_KP], best[SD_KP];
    float minU[SD_KP], maxU[SD_KP];
    int octL[SD_KP];
    bool any[SD_KP], inr[SD_KP];
#pragma unroll
    for (int q = 0; q < SD_KP; q++) {
        row[q] = __builtin_amdgcn_readfirstlane(Lv[q].in_rows ? (int)Lv[q].vL : 0);
        inr[q] = Lv[q].in_rows;
        minU[q] = Lv[q].uL - Lv[q].maxD; maxU[q] = Lv[q].uL - Lv[q].minD;
        octL[q] = Lv[q].oct;
        nsurv[q] = 0; best[q] = ((unsigned)TH_HIGH << 16); any[q] = false;
    }
    auto score = [&](int q) {
        for (unsigned k = lane; k < nsurv[q]; k += 64) {
            const unsigned j = s_list[wave][q][k];
            const unsigned dist = (unsigned)hamming256(Lv[q].a0, Lv[q].a1, dr[2 * j], dr[2 * j + 1]);
            best[q] = min(best[q], (dist << 16) | j);
        }
        nsurv[q] = 0;
    };
    for (int j = lane; j - lane < nr; j += 64) {
        uint4 r = make_uint4(1u, 0u, 0u, 0u);          // an empty band: matches no row
        if (j < nr) r = s_rec[j];
        const unsigned minr = r.x & 0xFFFFu, maxr = r.x >> 16;
        const float rx = __uint_as_float(r.y);
        const int oct = (int)r.z;
#pragma unroll
        for (int q = 0; q < SD_KP; q++) {
            const bool band = inr[q] && !(row[q] < minr || row[q] > maxr);
            any[q] = any[q] || band;
            const bool keep = band && !(oct < octL[q] - 1 || oct > octL[q] + 1) && rx >= minU[q] && rx <= maxU[q];   // :1250, :1255
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (m) {
                if (keep) s_list[wave][q][nsurv[q] + __builtin_popcountll(m & lt)] = (unsigned short)j;
                nsurv[q] += __builtin_popcountll(m);
                if (nsurv[q] > SD_LIST - 64) score(q);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SD_KP; q++) {
        if (!Lv[q].valid) continue;                     // wave-uniform
        if (nsurv[q]) score(q);
        const long long o = (long long)pair * a.out_stride + Lv[q].iL;
        float res_u = -1.0f, res_depth = -1.0f;
        int res_dist = -1, res_idx = -1, counted = 0;
        const bool have_cands = __builtin_amdgcn_ballot_w64(any[q]) != 0;
        const unsigned bst = st_wave_min(best[q]);
        if (inr[q] && have_cands && !(maxU[q] < a.p.min_x)) {  // :1213, :1236
            counted = 1;
            const int bestDist = (int)(bst >> 16);
            const int bestIdxR = (int)(bst & 0xFFFF);
            if (bestDist < (TH_HIGH + TH_LOW) / 2) {  // :1269
                float bestuR = __uint_as_float(s_rec[bestIdxR].y);
                float disparity = Lv[q].uL - bestuR;
                if (disparity >= Lv[q].minD && disparity < Lv[q].maxD) {
                    if (disparity <= 0) {
                        disparity = 0.01f;
                        bestuR = (float)((double)Lv[q].uL - 0.01);   // `uL-0.01` is double arithmetic in the reference (Frame.cc:1278)
                    }
                    res_depth = a.p.mbf / disparity;
                    res_u = bestuR;
                    res_dist = bestDist;
                    res_idx = bestIdxR;
                }
            }
        }
        if (lane == 0) {
            a.out.u_right[o] = res_u;
            a.out.depth[o] = res_depth;
            a.out.best_dist[o] = res_dist;
            a.out.best_idx[o] = res_idx;
            a.out.counted[o] = (unsigned char)counted;
        }
    }
}

// ---- the association around rows ----
#define SR_ROWS 72    // rows of the row table a workgroup needs: R + 2 W + 2
#define SR_CL 64      // left records per pass (a band of 8 rows holds ~35)
#ifndef SR_THREADS
#define SR_THREADS 256   // (128: 52 us, 512: 49 us against 38 per 64 pairs)
#endif

// cr = right records the workgroup can stage (dynamic LDS: 48 bytes each); a band whose buckets hold more is read in place
// xcd8: the grid is (8 * bands, ceil(pairs / 8)) and blockIdx.x & 7 picks the pair inside a group of eight, so that ALL bands of a
// pair run on ONE XCD (workgroups are dealt round-robin over the eight XCDs by their linear index, as in k_fast): the five 4-byte
// result stores per left keypoint -- scattered over the pair's output arrays, because the bands walk the keypoints in row order and the
// arrays are in the extractor's order -- then meet in one L2, which merges them into whole lines before they leave for memory.  With
// the plain (bands, pairs) grid a pair's bands sat on all eight XCDs, every L2 held a few dirty bytes of every line and wrote its own
// 32-byte sector: 12 MB of write traffic per 64 pairs for 2.2 MB of results (profiles/traffic_r03.json: 2.1-3.4 x the algorithmic
// bytes).  The overlapping right-side records that neighbouring bands stage are fetched through the fabric once instead of up to
// eight times, too.  k_stereo_bucket and k_stereo_cut handle pair p in workgroup p: the same XCD when pairs come in multiples of 8.
__global__ __launch_bounds__(SR_THREADS) void k_stereo_match_rows(StereoArgs a, int R, int cr, int xcd8, int npairs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sr_lds[];
    uint4* r_rec = reinterpret_cast<uint4*>(sr_lds);    // {first row of the band, octave << 16 | iR, x, band height | 1 << (16 + octave)}: the three filters from ONE 16-byte read
    uint4* r_d0 = r_rec + cr;                           // descriptor halves apart: 16-byte reads of consecutive records, no bank conflicts
    uint4* r_d1 = r_d0 + cr;
    __shared__ float l_x[SR_CL], l_minD[SR_CL], l_maxD[SR_CL];
    __shared__ int l_ro[SR_CL], l_i[SR_CL];             // row | octave << 16, original index
    __shared__ uint4 l_d0[SR_CL], l_d1[SR_CL];
    __shared__ int s_rs[SR_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, hl = lane & 31, hw = tid >> 5;
    const int pair = xcd8 ? (int)(blockIdx.y * 8 + (blockIdx.x & 7)) : (int)blockIdx.y;
    if (pair >= npairs) return;                         // the last group of eight may be partly empty
    const int nRows = a.p.n_rows, W = a.window;
    const int r0 = (int)(xcd8 ? blockIdx.x >> 3 : blockIdx.x) * R, r1 = min(r0 + R, nRows);
    if (r0 >= nRows) return;
    const int* lrs = a.lrow_start + (long long)pair * (nRows + 1);
    const int lb = lrs[r0], le = lrs[r1];
    if (lb == le) return;                               // no left keypoint in these rows (workgroup-uniform)
    const int* rs = a.row_start + (long long)pair * (nRows + 1);
    const int rlo = max(r0 - W, 0), rhi = min(r1 + W, nRows);      // right buckets [rlo, rhi) can hold candidates of rows [r0, r1)
    for (int t = tid; t <= rhi - rlo; t += SR_THREADS) s_rs[t] = rs[rlo + t];
    const int jb0 = rs[rlo], nR = rs[rhi] - jb0;
    const bool staged = nR <= cr;
    const long long so = (long long)pair * a.sort_stride;
    if (staged) {
        for (int t = tid; t < nR; t += SR_THREADS) {
            const long long j = so + jb0 + t;
            // the band as (first row, height) and the octave as a bit: "row inside the band" is one subtraction and one unsigned
            // compare, "octave within one of the left keypoint's" one AND against the left keypoint's three-bit mask -- 5 vector
            // operations per candidate where the (lo, hi) / octave-number form took 9; computed once per staged record, used by every
            // left keypoint whose window holds it (round 4: 59 -> 55 us came from the XCD placement, 55 -> see DESIGN from this)
            const unsigned band = a.sband[j], oi = a.soi[j];
            r_rec[t] = make_uint4(band & 0xFFFFu, oi, __float_as_uint(a.sx[j]), ((band >> 16) - (band & 0xFFFFu)) | (0x10000u << (oi >> 16)));
            r_d0[t] = a.sdesc[2 * j];
            r_d1[t] = a.sdesc[2 * j + 1];
        }
    }
#if defined(SR_STOP) && SR_STOP == 1
    if (nR != 123456) return;
#endif
    const gfo_keypoint* kl = a.kl + pair * a.pair_stride;
    const uint4* dl = reinterpret_cast<const uint4*>(a.dl + pair * a.pair_stride * 32);
    const bool win = a.min_d && a.max_d;
    const float maxD0 = a.p.mbf / a.p.mb;               // Frame.cc:1199-1200 (minZ = mb)
    for (int c0 = lb; c0 < le; c0 += SR_CL) {
        const int nc = min(SR_CL, le - c0);
        __syncthreads();                                // the previous pass has been consumed
        for (int t = tid; t < nc; t += SR_THREADS) {
            const int iL = a.lorder[so + c0 + t];
            const gfo_keypoint* L = kl + iL;
            l_x[t] = L->x;
            l_ro[t] = (int)L->y | (L->octave << 16);    // 0 <= (int)y < nRows <= 65535
            l_i[t] = iL;
            l_minD[t] = win ? a.min_d[pair * a.win_stride + iL] : 0.f;        // :1220-1231 flattened by the adapter
            l_maxD[t] = win ? a.max_d[pair * a.win_stride + iL] : maxD0;
            l_d0[t] = dl[2 * iL];
            l_d1[t] = dl[2 * iL + 1];
        }
        __syncthreads();
#if defined(SR_STOP) && SR_STOP == 2
        if (nR != 123456) return;
#endif
        for (int t = hw; t < nc; t += SR_THREADS / 32) {   // one half-wave per left keypoint
            const int ro = l_ro[t], row = ro & 0xFFFF, octL = ro >> 16;
            const float uL = l_x[t], minD = l_minD[t], maxD = l_maxD[t];
            const float minU = uL - maxD, maxU = uL - minD;
            const uint4 a0 = l_d0[t], a1 = l_d1[t];
            const unsigned octm = ((7u << octL) >> 1) << 16;      // octaves octL - 1 .. octL + 1 (:1250), in the upper half like the records' bit
            const int jb = s_rs[max(row - W, 0) - rlo] - jb0, je = s_rs[min(row + W + 1, nRows) - rlo] - jb0;
            unsigned best = ((unsigned)TH_HIGH << 16);  // bestDist = TH_HIGH, iR = 0: only dist < TH_HIGH replaces it
            float bx = 0.f;
            bool any = false;
            if (staged) {
                // two candidates per lane and trip: both records are requested before either is looked at (a trip is one LDS
                // round trip instead of two; the kernel is bound by those chains, not by instruction issue).
                // Measured and dropped: testing first and scoring the compacted survivors on full lanes (a ring per half-wave:
                // 48 us against 38 -- the ring's own LDS round trips lengthen the chain it was meant to shorten).
                for (int j0 = jb + hl; j0 < je; j0 += 64) {
                    const int j1 = j0 + 32;
                    const bool in1 = j1 < je;
                    const uint4 rec0 = r_rec[j0], rec1 = r_rec[in1 ? j1 : j0];
                    const bool band0 = (unsigned)(row - (int)rec0.x) <= (rec0.w & 0xFFFFu);
                    const bool band1 = in1 && (unsigned)(row - (int)rec1.x) <= (rec1.w & 0xFFFFu);
                    any = any || band0 || band1;
                    const float rx0 = __uint_as_float(rec0.z), rx1 = __uint_as_float(rec1.z);
                    const bool pass0 = band0 && (rec0.w & octm) != 0 && rx0 >= minU && rx0 <= maxU;   // :1250, :1255
                    const bool pass1 = band1 && (rec1.w & octm) != 0 && rx1 >= minU && rx1 <= maxU;
                    if (pass0) {
                        const unsigned key = ((unsigned)hamming256(a0, a1, r_d0[j0], r_d1[j0]) << 16) | (rec0.y & 0xFFFF);
                        if (key < best) { best = key; bx = rx0; }            // first minimum in iR order (:1260)
                    }
                    if (pass1) {
                        const unsigned key = ((unsigned)hamming256(a0, a1, r_d0[j1], r_d1[j1]) << 16) | (rec1.y & 0xFFFF);
                        if (key < best) { best = key; bx = rx1; }
                    }
                }
            } else {
                for (int j = jb + hl; j < je; j += 32) {
                    const long long g = so + jb0 + j;
                    const unsigned band = a.sband[g], oi = a.soi[g];
                    if (row < (int)(band & 0xFFFF) || row > (int)(band >> 16)) continue;
                    any = true;
                    const int oct = (int)(oi >> 16);
                    if (oct < octL - 1 || oct > octL + 1) continue;
                    const float rx = a.sx[g];
                    if (rx >= minU && rx <= maxU) {
                        const unsigned key = ((unsigned)hamming256(a0, a1, a.sdesc[2 * g], a.sdesc[2 * g + 1]) << 16) | (oi & 0xFFFF);
                        if (key < best) { best = key; bx = rx; }
                    }
                }
            }
            const unsigned long long anym = __builtin_amdgcn_ballot_w64(any);
            const bool have_cands = ((anym >> (32 * half)) & 0xFFFFFFFFull) != 0;
            unsigned red = best;
            red = st_half_min(red);   // inside the half
            // the x of the winner: keys are distinct (iR is part of them), so exactly one lane of the half holds it
            const unsigned long long own = __builtin_amdgcn_ballot_w64(best == red && red < ((unsigned)TH_HIGH << 16));
            const unsigned ownh = (unsigned)(own >> (32 * half));
            float bestuR = __shfl(bx, 32 * half + (ownh ? __ffs((int)ownh) - 1 : 0));
            float res_u = -1.0f, res_depth = -1.0f;
            int res_dist = -1, res_idx = -1, counted = 0;
            if (have_cands && !(maxU < a.p.min_x)) {     // :1213, :1236 (the row is inside the image: the keypoint is in lorder)
                counted = 1;
                const int bestDist = (int)(red >> 16);
                if (bestDist < (TH_HIGH + TH_LOW) / 2) {  // :1269
                    float disparity = uL - bestuR;
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = 0.01f;
                            bestuR = (float)((double)uL - 0.01);   // `uL-0.01` is double arithmetic in the reference (Frame.cc:1278)
                        }
                        res_depth = a.p.mbf / disparity;
                        res_u = bestuR;
                        res_dist = bestDist;
                        res_idx = (int)(red & 0xFFFF);
                    }
                }
            }
#if defined(SR_STOP) && SR_STOP == 3
            if (res_dist != 123456) continue;
#endif
            if (hl == 0) {
                const long long o = (long long)pair * a.out_stride + l_i[t];
                a.out.u_right[o] = res_u;
                a.out.depth[o] = res_depth;
                a.out.best_dist[o] = res_dist;
                a.out.best_idx[o] = res_idx;
                a.out.counted[o] = (unsigned char)counted;  // summed per pair by k_stereo_cut (no same-line atomics)
            }
        }
    }
}

// the outlier cut of one pair by one workgroup (any size); h_* (optional): host-mapped copies of the final outputs
__device__ __forceinline__ void stereo_cut_pair(const int* __restrict__ cnt_dev, int nl_host, const GfoStereoDev& out, int out_stride, int pair,
                                                int* hist, int* s_part, int* s_med, int* s_drop, int* s_cnt,
                                                float* __restrict__ h_u, float* __restrict__ h_d, int* __restrict__ h_nm)
{
    const int tid = threadIdx.x, NT = blockDim.x;
    const int nl = cnt_dev ? cnt_dev[2 * pair] : nl_host;
    const long long o = (long long)pair * out_stride;
    if (tid < 128) hist[tid] = 0;
    if (tid == 0) *s_drop = 0, *s_cnt = 0, *s_med = 1 << 30;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < nl; i += NT) {
        const int d = out.best_dist[o + i];
        if (d >= 0) atomicAdd(&hist[d], 1);
        mine += out.counted[o + i];
    }
    if (mine) atomicAdd(s_cnt, mine);
    __syncthreads();
    // median = the distance whose running count first exceeds ndi / 2 (the element of rank ndi/2 in the sorted
    // list, :1297): a workgroup scan over the 128 bins instead of one thread walking them
    int ndi;
    int h = 0;
    h = tid < 128 ? hist[tid] : 0;                // (NT >= 128 everywhere this is launched: one bin per thread)
    const int incl = st_block_incl_scan(h, s_part, &ndi);
    if (tid < 128 && ndi > 0 && incl > ndi / 2) atomicMin(s_med, tid);
    __syncthreads();
    if (ndi == 0) {
        if (h_u)
            for (int i = tid; i < nl; i += NT) { h_u[o + i] = out.u_right[o + i]; h_d[o + i] = out.depth[o + i]; }
        if (tid == 0) {
            out.nmatched[pair] = *s_cnt;
            if (h_nm) h_nm[pair] = *s_cnt;
        }
        return;
    }
    const float thDist = 1.5f * 1.4f * (float)*s_med;  // :1298
    int drop = 0;
    for (int i = tid; i < nl; i += NT) {
        const int d = out.best_dist[o + i];
        const bool cut = d >= 0 && !((float)d < thDist);
        if (cut) {
            out.u_right[o + i] = -1.0f;
            out.depth[o + i] = -1.0f;
            drop++;
        }
        if (h_u) {
            h_u[o + i] = cut ? -1.0f : out.u_right[o + i];
            h_d[o + i] = cut ? -1.0f : out.depth[o + i];
        }
    }
    if (drop) atomicAdd(s_drop, drop);
    __syncthreads();
    if (tid == 0) {
        out.nmatched[pair] = *s_cnt - *s_drop;
        if (h_nm) h_nm[pair] = *s_cnt - *s_drop;
    }
}

__global__ __launch_bounds__(1024) void k_stereo_cut(const int* __restrict__ cnt_dev, int nl_host, GfoStereoDev out,
                                                     int out_stride)
{
    __shared__ int hist[128];
    __shared__ int s_part[16];
    __shared__ int s_med, s_drop, s_cnt;
    stereo_cut_pair(cnt_dev, nl_host, out, out_stride, blockIdx.x, hist, s_part, &s_med, &s_drop, &s_cnt, nullptr, nullptr, nullptr);
}

// k_pack_results with the cut in it (GfoPack::cut_pairs > 0): workgroups [0, cut_pairs) cut one pair each, the others copy
__global__ __launch_bounds__(1024) void k_pack_results_cut(GfoPack p)
{
    if ((int)blockIdx.x < p.cut_pairs) {
        __shared__ int hist[128];
        __shared__ int s_part[16];
        __shared__ int s_med, s_drop, s_cnt;
        stereo_cut_pair(p.cut_cnt_dev, p.cut_nl_host, p.cut_out, p.cut_out_stride, blockIdx.x, hist, s_part, &s_med, &s_drop, &s_cnt,
                        p.h_u_right, p.h_depth, p.h_nmatched);
        return;
    }
    const int t = ((int)blockIdx.x - p.cut_pairs) * 1024 + threadIdx.x, stride = ((int)gridDim.x - p.cut_pairs) * 1024;
    for (int s = 0; s < p.nseg; s++)
        for (int i = t; i < p.n16[s]; i += stride) p.dst[s][i] = p.src[s][i];
}

void gfo_launch_pack_cut(gfo_ctx* c, const GfoPack& p, hipStream_t st)
{
    int total = 0;
    for (int s_ = 0; s_ < p.nseg; s_++) total += p.n16[s_];
    // 1024 threads: the cut is two passes of dependent loads over the pair's ~2000 entries -- two trips per thread instead of eight
    const int copy_blocks = (total + 4095) / 4096 > 0 ? (total + 4095) / 4096 : 1;
    GFO_LAUNCH(c, k_pack_results_cut, dim3(p.cut_pairs + copy_blocks), dim3(1024), 0, st, p);
}

// ---------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (Frame.cc:889-1078), the SAD sub-pixel variant the reference compiles out with
// ALTER_STEREO_MATCHING: same row-band Hamming search (on mvKeys / mvKeysRight, maxU < 0 instead of
// < mnMinX), then an 11x11 SAD over 11 horizontal shifts on the keypoint's pyramid level of BOTH images,
// parabola fit, and the same 2.1 x median cut on the SAD values.  One wavefront per left keypoint: lanes
// sweep the candidates, then split the 121 window pixels; the eleven shift sums are wave-reduced.
// All SAD terms are integers (differences of u8), so the float/double accumulation of the reference
// (cv::norm on CV_32F) is exact and integer accumulation gives the same numbers.
// ---------------------------------------------------------------------------------------------
struct SadArgs {
    StereoArgs s;
    const GfoGeom* g;
    GfoInput in;
    const uint8_t* pyr;
    const float* inv_scale;
};

__global__ __launch_bounds__(256) void k_stereo_match_sad(SadArgs A)
{
    const StereoArgs& a = A.s;
    const GfoGeom& g = *A.g;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform: keep it scalar
    const int pair = blockIdx.y;
    const int iL = blockIdx.x * 4 + wave;
    const int nl = a.cnt_dev[2 * pair];
    if (iL >= nl) return;
    const gfo_keypoint* kl = a.kl + pair * a.pair_stride;
    const gfo_keypoint* kr = a.kr + pair * a.pair_stride;
    const uint8_t* dl = a.dl + pair * a.pair_stride * 32;
    const long long o = (long long)pair * a.out_stride + iL;
    const long long so = (long long)pair * a.sort_stride;
    float res_u = -1.0f, res_depth = -1.0f;
    int res_dist = -1, res_idx = -1;
    const gfo_keypoint L = kl[iL];
    const float vL = L.y, uL = L.x;
    const int nRows = a.p.n_rows;
    const int row = (int)vL;
    const float minD = 0.f, maxD = a.p.mbf / a.p.mb;
    const float minU = uL - maxD, maxU = uL - minD;
    if (row >= 0 && row <= nRows - 1 && !(maxU < 0)) {
        const uint4* dlp = reinterpret_cast<const uint4*>(dl + (long long)iL * 32);
        const uint4 a0 = dlp[0], a1 = dlp[1];
        const int* rs = a.row_start + (long long)pair * (nRows + 1);
        const int jb = rs[max(row - a.window, 0)], je = rs[min(row + a.window + 1, nRows)];
        unsigned best = ((unsigned)TH_HIGH << 16);
        for (int j = jb + lane; j < je; j += 64) {
            const unsigned band = a.sband[so + j];   // :910-916 (clamped), packed by k_stereo_bucket
            const unsigned oi = a.soi[so + j];
            const int oct = (int)(oi >> 16);
            if (row < (int)(band & 0xFFFF) || row > (int)(band >> 16)) continue;
            if (oct < L.octave - 1 || oct > L.octave + 1) continue;
            const float rx = a.sx[so + j];
            if (rx >= minU && rx <= maxU) {
                const unsigned dist = (unsigned)hamming256(a0, a1, a.sdesc[2 * (so + j)], a.sdesc[2 * (so + j) + 1]);
                best = min(best, (dist << 16) | (oi & 0xFFFF));
            }
        }
        best = st_wave_min(best);
        const int bestDist = (int)(best >> 16), bestIdxR = (int)(best & 0xFFFF);
        if (bestDist < (TH_HIGH + TH_LOW) / 2) {  // :973
            const int lvl = L.octave;
            const float uR0 = kr[bestIdxR].x;
            const float sfac = A.inv_scale[lvl];
            const float scaleduL = roundf(uL * sfac), scaledvL = roundf(vL * sfac), scaleduR0 = roundf(uR0 * sfac);
            const int w = 5, LL = 5;
            int pl, prr;
            const uint8_t* IL = gfo_level_ptr(g, A.in, A.pyr, lvl, 2 * pair, &pl);
            const uint8_t* IR = gfo_level_ptr(g, A.in, A.pyr, lvl, 2 * pair + 1, &prr);
            const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
            const float iniu = scaleduR0 + LL - w, endu = scaleduR0 + LL + w + 1;
            if (!(iniu < 0 || endu >= (float)g.lv[lvl].w)) {  // :1004
                const int ilc = IL[(long long)cv * pl + cu];
                int acc[11];
#pragma unroll
                for (int k = 0; k < 11; k++) acc[k] = 0;
                for (int p0 = 0; p0 < 121; p0 += 64) {
                    const int p = p0 + lane;
                    if (p < 121) {
                        const int dy = p / 11 - w, dx = p - (p / 11) * 11 - w;
                        const int il = (int)IL[(long long)(cv + dy) * pl + cu + dx] - ilc;
                        const uint8_t* rrow = IR + (long long)(cv + dy) * prr + cr + dx;
                        const uint8_t* rcen = IR + (long long)cv * prr + cr;
#pragma unroll
                        for (int k = 0; k < 11; k++) {
                            const int ir = (int)rrow[k - LL] - (int)rcen[k - LL];
                            acc[k] += abs(il - ir);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 11; k++) acc[k] = st_wave_sum(acc[k]);
                int sadBest = 2147483647, bestinc = 0;
#pragma unroll
                for (int k = 0; k < 11; k++)
                    if ((float)acc[k] < (float)sadBest) { sadBest = acc[k]; bestinc = k - LL; }  // :1019 (float < int compare)
                if (!(bestinc == -LL || bestinc == LL)) {
                    float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
                    for (int k = 1; k < 10; k++)
                        if (k - LL == bestinc) { d1 = (float)acc[k - 1]; d2 = (float)acc[k]; d3 = (float)acc[k + 1]; }
                    const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
                    if (!(deltaR < -1 || deltaR > 1)) {
                        float bestuR = g.lv[lvl].scale * ((float)scaleduR0 + (float)bestinc + deltaR);
                        float disparity = uL - bestuR;
                        if (disparity >= minD && disparity < maxD) {
                            if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }   // Frame.cc:1053-1054
                            res_depth = a.p.mbf / disparity;
                            res_u = bestuR;
                            res_dist = sadBest;
                            res_idx = bestIdxR;
                        }
                    }
                }
            }
        }
    }
    if (lane == 0) {
        a.out.u_right[o] = res_u;
        a.out.depth[o] = res_depth;
        a.out.best_dist[o] = res_dist;
        a.out.best_idx[o] = res_idx;
        a.out.counted[o] = res_dist >= 0 ? 1 : 0;
    }
}

// exact rank-(n/2) order statistic of the SAD values (< 2^16) by a two-level 256-bin histogram, then the cut
__global__ __launch_bounds__(256) void k_stereo_cut_sad(const int* __restrict__ cnt_dev, GfoStereoDev out, int out_stride)
{
    __shared__ int hist[256];
    __shared__ int s_hi, s_rank, s_med, s_drop, s_n;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nl = cnt_dev[2 * pair];
    const long long o = (long long)pair * out_stride;
    hist[tid] = 0;
    if (tid == 0) s_drop = 0, s_n = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < nl; i += 256) {
        const int d = out.best_dist[o + i];
        if (d >= 0) { atomicAdd(&hist[min(d >> 8, 255)], 1); mine++; }
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    const int ndi = s_n;
    if (ndi == 0) { if (tid == 0) out.nmatched[pair] = 0; return; }
    if (tid == 0) {
        int acc = 0, b = 0;
        for (; b < 256; b++) { if (acc + hist[b] > ndi / 2) break; acc += hist[b]; }
        s_hi = b;
        s_rank = ndi / 2 - acc;  // rank inside the bin
    }
    __syncthreads();
    const int hi = s_hi;
    __syncthreads();
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 256) {
        const int d = out.best_dist[o + i];
        if (d >= 0 && min(d >> 8, 255) == hi) atomicAdd(&hist[d & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, b = 0;
        for (; b < 256; b++) { acc += hist[b]; if (acc > s_rank) break; }
        s_med = (hi << 8) | b;
    }
    __syncthreads();
    const float thDist = 1.5f * 1.4f * (float)s_med;  // :1061
    int drop = 0;
    for (int i = tid; i < nl; i += 256) {
        const int d = out.best_dist[o + i];
        if (d >= 0 && !((float)d < thDist)) {
            out.u_right[o + i] = -1.0f;
            out.depth[o + i] = -1.0f;
            drop++;
        }
    }
    if (drop) atomicAdd(&s_drop, drop);
    __syncthreads();
    if (tid == 0) out.nmatched[pair] = ndi - s_drop;
}

void gfo_launch_stereo_sad(gfo_ctx* c, const GfoStereoLaunch& s, const GfoInput& in, const float* d_inv_scale)
{
    SadArgs A{};
    StereoArgs& a = A.s;
    a.kl = s.kl; a.dl = s.dl; a.kr = s.kr; a.dr = s.dr;
    a.cnt_dev = s.cnt_dev; a.nl_host = 0; a.nr_host = 0;
    a.pair_stride = s.pair_stride_kp;
    a.scale = s.d_scale;
    a.p = s.p;
    a.min_d = nullptr; a.max_d = nullptr; a.win_stride = 0;
    a.out = s.out; a.out_stride = s.out_stride;
    a.sx = s.sort.sx; a.sband = reinterpret_cast<unsigned*>(s.sort.sy); a.soi = s.sort.soi; a.sdesc = reinterpret_cast<uint4*>(s.sort.sdesc);
    a.row_start = s.sort.row_start;
    a.lorder = nullptr; a.lrow_start = nullptr;      // this variant sweeps per left keypoint in list order
    a.sort_stride = s.sort_stride;
    a.window = s.window;
    A.g = c->d_geom;
    A.in = in;
    A.pyr = c->d_pyr;
    A.inv_scale = d_inv_scale;
    gfo_prof_begin(c, ST_STEREO_BUCKET);
    GFO_LAUNCH(c, k_stereo_bucket, dim3(s.npairs), dim3(1024), (size_t)(s.p.n_rows + 256) * sizeof(int), c->stream, a);
    gfo_prof_end(c);
    gfo_prof_begin(c, ST_STEREO);
    GFO_LAUNCH(c, k_stereo_match_sad, dim3((s.out_stride + 3) / 4, s.npairs), dim3(256), 0, c->stream, A);
    gfo_prof_end(c);
    gfo_prof_begin(c, ST_STEREO_CUT);
    GFO_LAUNCH(c, k_stereo_cut_sad, dim3(s.npairs), dim3(256), 0, c->stream, s.cnt_dev, s.out, s.out_stride);
    gfo_prof_end(c);
}

int gfo_stereo_window(const float* scale, int nlevels)
{
    float mx = 0.f;
    for (int l = 0; l < nlevels; l++) mx = scale[l] > mx ? scale[l] : mx;
    return (int)ceilf(2.0f * mx) + 1;
}

void gfo_launch_stereo(gfo_ctx* c, const GfoStereoLaunch& s)
{
    StereoArgs a{};
    a.kl = s.kl; a.dl = s.dl; a.kr = s.kr; a.dr = s.dr;
    a.cnt_dev = s.cnt_dev; a.nl_host = s.nl_host; a.nr_host = s.nr_host;
    a.pair_stride = s.pair_stride_kp;
    a.scale = s.d_scale;
    a.p = s.p;
    a.min_d = s.min_d; a.max_d = s.max_d; a.win_stride = s.win_stride;
    a.out = s.out; a.out_stride = s.out_stride;
    a.sx = s.sort.sx; a.sband = reinterpret_cast<unsigned*>(s.sort.sy); a.soi = s.sort.soi; a.sdesc = reinterpret_cast<uint4*>(s.sort.sdesc);
    a.row_start = s.sort.row_start;
    a.sort_stride = s.sort_stride;
    a.window = s.window;
    a.nlevels = s.nlevels;
    const int max_nl = s.cnt_dev ? s.out_stride : s.nl_host;
    if (max_nl <= 0) {
        (void)hipMemsetAsync(s.out.nmatched, 0, sizeof(int) * s.npairs, c->stream);
        return;
    }
    // rows per workgroup of the row form (64 pairs of 480 rows: 2 / 4 / 6 / 8 / 16 / 32 rows -> 51 / 40 / 39 / 38 / 42 / 51 us)
    const int rows_env = getenv("GFO_STEREO_ROWS") ? atoi(getenv("GFO_STEREO_ROWS")) : -1;   // 0 = the per-keypoint form, R > 0 = that many rows (read per call: the tests run both forms in one process)
    int R = rows_env > 0 ? rows_env : 8;
    if (R > 32) R = 32;
    // from four pairs on; one to three pairs -- the per-frame path -- spread better as one half-wave per left keypoint
    // (a stereo frame: 0.207 ms against 0.212 with 120 row workgroups)
    const bool rows_form = rows_env != 0 && (s.npairs >= 4 || rows_env > 0) && s.sort.lorder && s.p.n_rows <= 65535 && R + 2 * s.window + 2 <= SR_ROWS;
    // one to three pairs (the per-frame path): no bucketing launch at all, k_stereo_match_direct (GFO_STEREO_DIRECT=0, or any
    // GFO_STEREO_ROWS setting, keeps the bucketed forms: the tests run all of them in one process)
    const char* direct_env = getenv("GFO_STEREO_DIRECT");
    const int max_nr = s.cnt_dev ? s.out_stride : s.nr_host;
    const bool direct = !rows_form && rows_env < 0 && !(direct_env && direct_env[0] == '0') && s.npairs <= 3 && max_nr > 0 && max_nr <= 3584 && s.nlevels > 0 &&   /* 16 B per right keypoint + 4 KB of lists: inside the 64 KB a launch gets without asking */
                        s.p.n_rows <= 65535;
    if (direct) {
        gfo_prof_begin(c, ST_STEREO);
        GFO_LAUNCH(c, k_stereo_match_direct, dim3((max_nl + SD_WAVES * SD_KP - 1) / (SD_WAVES * SD_KP), s.npairs), dim3(SD_NT), (size_t)max_nr * sizeof(uint4), c->stream, a);
        gfo_prof_end(c);
        if (!s.cut_in_pack) {
            gfo_prof_begin(c, ST_STEREO_CUT);
            GFO_LAUNCH(c, k_stereo_cut, dim3(s.npairs), dim3(1024), 0, c->stream, s.cnt_dev, s.nl_host, s.out, s.out_stride);
            gfo_prof_end(c);
        }
        return;
    }
    a.lorder = rows_form ? s.sort.lorder : nullptr;
    a.lrow_start = rows_form ? s.sort.lrow_start : nullptr;
    gfo_prof_begin(c, ST_STEREO_BUCKET);
    // threads of the two small per-pair kernels (bucket, cut): 1024 alone is fastest (fewer serial round trips), but a
    // 1024-thread workgroup needs sixteen free wave slots in ONE CU at once, and in the running pipeline it waits for them behind
    // the other context's kernels (the launches stretch 4-6 x, profiles/overlap_trace_r03.txt): 512 / 256 threads from 32 pairs on
    // (1024 / 1024: 278-280 k frames/s, 512 / 256: 281-281.5 k, 256 / 256: 276.5 k; a stereo frame alone keeps 1024)
    static const int bt_env = getenv("GFO_STEREO_BUCKET_THREADS") ? atoi(getenv("GFO_STEREO_BUCKET_THREADS")) : 0;
    static const int ct_env = getenv("GFO_STEREO_CUT_THREADS") ? atoi(getenv("GFO_STEREO_CUT_THREADS")) : 0;
    const int bucket_threads = bt_env == 256 || bt_env == 512 || bt_env == 1024 ? bt_env : (s.npairs >= 32 ? 512 : 1024);
    const int cut_threads = ct_env == 256 || ct_env == 512 || ct_env == 1024 ? ct_env : (s.npairs >= 32 ? 256 : 1024);
    GFO_LAUNCH(c, k_stereo_bucket, dim3(s.npairs, rows_form ? 2 : 1), dim3(bucket_threads), (size_t)(s.p.n_rows + 256) * sizeof(int), c->stream, a);
    gfo_prof_end(c);
    gfo_prof_begin(c, ST_STEREO);
    if (rows_form) {
        // right records staged per workgroup: twice what an even spread of the keypoints over the rows puts into R + 2 W buckets
        // (128: most bands overflow into the read-in-place path, 53 us; 512: 41 us -- LDS nobody needs)
        static const int cr_env = getenv("GFO_STEREO_CR") ? atoi(getenv("GFO_STEREO_CR")) : 0;
        const int per_row8 = (8 * max_nl + s.p.n_rows - 1) / s.p.n_rows;      // keypoints per 8 rows
        int cr = cr_env > 0 ? cr_env : ((2 * per_row8 * (R + 2 * s.window) / 8 + 63) & ~63);
        cr = cr < 64 ? 64 : (cr > 1024 ? 1024 : cr);
        static const int xcd_env = getenv("GFO_STEREO_XCD") ? atoi(getenv("GFO_STEREO_XCD")) : 1;
        const int xcd8 = xcd_env && s.npairs >= 8 ? 1 : 0;      // below eight pairs seven of eight workgroups would be empty
        const unsigned bands = (unsigned)((s.p.n_rows + R - 1) / R);
        const dim3 grid = xcd8 ? dim3(bands * 8u, (unsigned)(s.npairs + 7) / 8u) : dim3(bands, (unsigned)s.npairs);
        GFO_LAUNCH(c, k_stereo_match_rows, grid, dim3(SR_THREADS), (size_t)cr * 48, c->stream, a, R, cr, xcd8, s.npairs);
    } else {
        static const int snw_env = getenv("GFO_STEREO_WAVES") ? atoi(getenv("GFO_STEREO_WAVES")) : 4;   // waves per workgroup, 2 left keypoints each
        const int snw = snw_env < 1 ? 1 : (snw_env > 4 ? 4 : snw_env);                                     // the kernel is built for <= 256 threads
        dim3 grid((max_nl + 2 * snw - 1) / (2 * snw), s.npairs);
        GFO_LAUNCH(c, k_stereo_match, grid, dim3(64 * snw), 0, c->stream, a);
    }
    gfo_prof_end(c);
    if (s.cut_in_pack) return;
    gfo_prof_begin(c, ST_STEREO_CUT);
    GFO_LAUNCH(c, k_stereo_cut, dim3(s.npairs), dim3(cut_threads), 0, c->stream, s.cnt_dev, s.nl_host, s.out, s.out_stride);
    gfo_prof_end(c);
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_stereo(std::vector<const void*>& v)
{
    v.push_back((const void*)k_stereo_bucket); v.push_back((const void*)k_stereo_match); v.push_back((const void*)k_stereo_match_rows);
    v.push_back((const void*)k_stereo_cut); v.push_back((const void*)k_stereo_match_sad); v.push_back((const void*)k_stereo_cut_sad);
    v.push_back((const void*)k_stereo_match_direct); v.push_back((const void*)k_pack_results_cut);
}
