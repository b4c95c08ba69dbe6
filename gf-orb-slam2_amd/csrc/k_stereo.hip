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
//   k_stereo_match  : one 32-lane half-wave per left keypoint sweeps the buckets [row-W, row+W]
//                     (W = ceil(2*max scale)+1 covers every band that can contain `row`), applies the
//                     exact band / octave / disparity-window predicates, takes the 256-bit Hamming
//                     distance with v_bcnt and wave-reduces min(dist << 16 | iR);
//   k_stereo_cut    : the outlier cut (:1290-1313) needs only the (ndi/2)-th order statistic of the
//                     accepted distances: a 128-bin LDS histogram gives it exactly (no sort).
#include "gfo_internal.h"

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
    GfoStereoDev out;
    int out_stride;
    // bucketed right side, [pairs][sort_stride]
    float* sx;
    float* sy;
    unsigned* soi;            // octave << 16 | iR
    uint4* sdesc;             // 2 x uint4 per keypoint
    int* row_start;           // [pairs][n_rows + 1]
    int sort_stride;
    int window;               // W
};

__global__ __launch_bounds__(256) void k_stereo_bucket(StereoArgs a)
{
    extern __shared__ int s_hist[];  // n_rows + 256
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nRows = a.p.n_rows;
    int* s_part = s_hist + nRows;
    const int nr = a.cnt_dev ? a.cnt_dev[2 * pair + 1] : a.nr_host;
    const gfo_keypoint* kr = a.kr + pair * a.pair_stride;
    const uint4* dr = reinterpret_cast<const uint4*>(a.dr + pair * a.pair_stride * 32);
    for (int r = tid; r < nRows; r += 256) s_hist[r] = 0;
    __syncthreads();
    for (int i = tid; i < nr; i += 256) {
        const int b = min(max((int)floorf(kr[i].y), 0), nRows - 1);
        atomicAdd(&s_hist[b], 1);
    }
    __syncthreads();
    // exclusive scan over nRows entries
    const int chunk = (nRows + 255) / 256;
    const int b0 = tid * chunk, e0 = min(b0 + chunk, nRows);
    int s = 0;
    for (int r = b0; r < e0; r++) s += s_hist[r];
    s_part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - s;
    int* rs = a.row_start + (long long)pair * (nRows + 1);
    for (int r = b0; r < e0; r++) {
        const int v = s_hist[r];
        rs[r] = run;
        s_hist[r] = run;  // fill cursor
        run += v;
    }
    if (tid == 255) rs[nRows] = s_part[255];
    __syncthreads();
    const long long so = (long long)pair * a.sort_stride;
    for (int i = tid; i < nr; i += 256) {
        const gfo_keypoint k = kr[i];
        const int b = min(max((int)floorf(k.y), 0), nRows - 1);
        const int pos = atomicAdd(&s_hist[b], 1);
        a.sx[so + pos] = k.x;
        a.sy[so + pos] = k.y;
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
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, hl = lane & 31;
    const int pair = blockIdx.y;
    const int nl = a.cnt_dev ? a.cnt_dev[2 * pair] : a.nl_host;
    const int iL0 = (blockIdx.x * 4 + wave) * 2;
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
        minD = a.min_d[iL];
        maxD = a.max_d[iL];
    }
    const float minU = uL - maxD, maxU = uL - minD;
    const uint4* dlp = reinterpret_cast<const uint4*>(dl + (long long)iL * 32);
    const uint4 a0 = dlp[0], a1 = dlp[1];
    const int* rs = a.row_start + (long long)pair * (nRows + 1);
    const int jb = rs[max(row - a.window, 0)], je = in_rows ? rs[min(row + a.window + 1, nRows)] : jb;
    unsigned best = ((unsigned)TH_HIGH << 16);  // bestDist = TH_HIGH, iR = 0: only dist < TH_HIGH replaces it
    bool any = false;
    for (int j = jb + hl; j < je; j += 32) {
        const float ry = a.sy[so + j];
        const unsigned oi = a.soi[so + j];
        const int oct = (int)(oi >> 16);
        // Frame.h:248-256 row band of this right keypoint
        const float r = 2.0f * a.scale[oct];
        const int maxr = (int)fminf((float)(nRows - 1), ceilf(ry + r));
        const int minr = (int)fmaxf(0.0f, floorf(ry - r));
        if (row < minr || row > maxr) continue;
        any = true;
        if (oct < L.octave - 1 || oct > L.octave + 1) continue;  // :1250
        const float rx = a.sx[so + j];
        if (rx >= minU && rx <= maxU) {                          // :1255
            const unsigned dist = (unsigned)hamming256(a0, a1, a.sdesc[2 * (so + j)], a.sdesc[2 * (so + j) + 1]);
            best = min(best, (dist << 16) | (oi & 0xFFFF));      // first minimum in iR order (:1260)
        }
    }
    const unsigned long long anym = __ballot(any);
    const bool have_cands = ((anym >> (32 * half)) & 0xFFFFFFFFull) != 0;
#pragma unroll
    for (int s = 16; s > 0; s >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, s));  // inside the half
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
                    bestuR = uL - 0.01f;
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

__global__ __launch_bounds__(256) void k_stereo_cut(const int* __restrict__ cnt_dev, int nl_host, GfoStereoDev out,
                                                    int out_stride)
{
    __shared__ int hist[128];
    __shared__ int s_med, s_drop, s_cnt;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nl = cnt_dev ? cnt_dev[2 * pair] : nl_host;
    const long long o = (long long)pair * out_stride;
    if (tid < 128) hist[tid] = 0;
    if (tid == 0) s_drop = 0, s_cnt = 0;
    __syncthreads();
    int mine = 0;
    for (int i = tid; i < nl; i += 256) {
        const int d = out.best_dist[o + i];
        if (d >= 0) atomicAdd(&hist[d], 1);
        mine += out.counted[o + i];
    }
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (tid == 0) {
        int ndi = 0;
        for (int d = 0; d < 128; d++) ndi += hist[d];
        int med = -1;
        if (ndi > 0) {
            int acc = 0;
            for (int d = 0; d < 128; d++) {
                acc += hist[d];
                if (acc > ndi / 2) { med = d; break; }  // element of rank ndi/2 in the sorted list (:1297)
            }
        }
        s_med = med;
    }
    __syncthreads();
    if (s_med < 0) {
        if (tid == 0) out.nmatched[pair] = s_cnt;
        return;
    }
    const float thDist = 1.5f * 1.4f * (float)s_med;  // :1298
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
    if (tid == 0) out.nmatched[pair] = s_cnt - s_drop;
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
    a.min_d = s.min_d; a.max_d = s.max_d;
    a.out = s.out; a.out_stride = s.out_stride;
    a.sx = s.sort.sx; a.sy = s.sort.sy; a.soi = s.sort.soi; a.sdesc = reinterpret_cast<uint4*>(s.sort.sdesc);
    a.row_start = s.sort.row_start;
    a.sort_stride = s.sort_stride;
    a.window = s.window;
    const int max_nl = s.cnt_dev ? s.out_stride : s.nl_host;
    if (max_nl <= 0) {
        (void)hipMemsetAsync(s.out.nmatched, 0, sizeof(int) * s.npairs, c->stream);
        return;
    }
    gfo_prof_begin(c, ST_STEREO_BUCKET);
    hipLaunchKernelGGL(k_stereo_bucket, dim3(s.npairs), dim3(256), (size_t)(s.p.n_rows + 256) * sizeof(int), c->stream, a);
    gfo_prof_end(c);
    dim3 grid((max_nl + 7) / 8, s.npairs);  // 4 waves x 2 left keypoints per workgroup
    gfo_prof_begin(c, ST_STEREO);
    hipLaunchKernelGGL(k_stereo_match, grid, dim3(256), 0, c->stream, a);
    gfo_prof_end(c);
    gfo_prof_begin(c, ST_STEREO_CUT);
    hipLaunchKernelGGL(k_stereo_cut, dim3(s.npairs), dim3(256), 0, c->stream, s.cnt_dev, s.nl_host, s.out, s.out_stride);
    gfo_prof_end(c);
}
