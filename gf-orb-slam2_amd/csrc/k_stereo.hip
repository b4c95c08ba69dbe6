// k_stereo.hip -- Frame::PrepareStereoCandidates + ComputeStereoMatches_Undistorted(false)
// (Frame.h:230-263, Frame.cc:1167-1316; ALTER_STEREO_MATCHING path), plus
// ORBmatcher::DescriptorDistance (ORBmatcher.cc:1768-1784) as XOR + popcount.
//
// The reference builds a row -> right-keypoint table and scans one row's list per left
// keypoint.  Membership of right keypoint iR in row `r` is the pure predicate
// minr(iR) <= r <= maxr(iR), and the scan keeps the FIRST minimum in iR order, i.e. the
// lexicographic minimum of (distance, iR).  So one wavefront per left keypoint sweeps all
// right keypoints (64 per step), applies the band / octave / disparity-window predicates,
// takes the 256-bit Hamming distance with v_bcnt, and wave-reduces min(dist << 16 | iR):
// no table, no ordering problem, identical indices.
// The outlier cut (:1290-1313) needs only the (ndi/2)-th order statistic of the accepted
// distances, which a 128-bin LDS histogram gives exactly (no sort).
#include "gfo_internal.h"

#define TH_HIGH 100  // ORBmatcher.cc:57
#define TH_LOW 50    // ORBmatcher.cc:58

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4* __restrict__ b)
{
    const uint4 b0 = b[0], b1 = b[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ __launch_bounds__(256) void k_stereo_match(const gfo_keypoint* __restrict__ kl_all,
                                                      const uint8_t* __restrict__ dl_all,
                                                      const gfo_keypoint* __restrict__ kr_all,
                                                      const uint8_t* __restrict__ dr_all,
                                                      const int* __restrict__ cnt_dev, int nl_host, int nr_host,
                                                      long long pair_stride, const float* __restrict__ scale,
                                                      gfo_stereo_params p, const float* __restrict__ min_d,
                                                      const float* __restrict__ max_d, GfoStereoDev out,
                                                      int out_stride)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.y;
    const int iL = blockIdx.x * 4 + wave;
    const int nl = cnt_dev ? cnt_dev[2 * pair] : nl_host;
    const int nr = cnt_dev ? cnt_dev[2 * pair + 1] : nr_host;
    if (iL >= nl) return;
    const gfo_keypoint* kl = kl_all + pair * pair_stride;
    const gfo_keypoint* kr = kr_all + pair * pair_stride;
    const uint8_t* dl = dl_all + pair * pair_stride * 32;
    const uint8_t* dr = dr_all + pair * pair_stride * 32;
    const long long o = (long long)pair * out_stride + iL;
    float res_u = -1.0f, res_depth = -1.0f;
    int res_dist = -1, res_idx = -1, counted = 0;

    const gfo_keypoint L = kl[iL];
    const float vL = L.y, uL = L.x;
    const int nRows = p.n_rows;
    if (!(vL < 0 || vL > (float)(nRows - 1))) {  // Frame.cc:1208
        const int row = (int)vL;
        float minD = 0.f, maxD = p.mbf / p.mb;      // :1199-1200 (minZ = mb)
        if (min_d && max_d) {                      // :1220-1231 flattened by the adapter
            minD = min_d[iL];
            maxD = max_d[iL];
        }
        const float minU = uL - maxD, maxU = uL - minD;
        const uint4* dlp = reinterpret_cast<const uint4*>(dl + (long long)iL * 32);
        const uint4 a0 = dlp[0], a1 = dlp[1];
        unsigned best = ((unsigned)TH_HIGH << 16);  // bestDist = TH_HIGH, strict < below
        bool any = false;
        for (int iR = lane; iR < nr; iR += 64) {
            const gfo_keypoint R = kr[iR];
            // Frame.h:248-256 row band of this right keypoint
            const float r = 2.0f * scale[R.octave];
            const int maxr = (int)fminf((float)(nRows - 1), ceilf(R.y + r));
            const int minr = (int)fmaxf(0.0f, floorf(R.y - r));
            if (row < minr || row > maxr) continue;
            any = true;
            if (R.octave < L.octave - 1 || R.octave > L.octave + 1) continue;  // :1250
            if (R.x >= minU && R.x <= maxU) {                                  // :1255
                const unsigned dist = (unsigned)hamming256(a0, a1, reinterpret_cast<const uint4*>(dr + (long long)iR * 32));
                const unsigned cand = (dist << 16) | (unsigned)iR;
                if (dist < (best >> 16) || cand < best) best = min(best, cand);
            }
        }
        // NOTE: `cand < best` alone is the rule (lexicographic (dist, iR)); the initial best carries iR = 0
        // with dist = TH_HIGH, and only dist < TH_HIGH may replace it, which the first clause guarantees.
        const bool have_cands = __any(any);
        if (have_cands && !(maxU < p.min_x)) {  // :1213, :1236
            counted = 1;
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, s));
            const int bestDist = (int)(best >> 16);
            const int bestIdxR = (int)(best & 0xFFFF);
            if (bestDist < (TH_HIGH + TH_LOW) / 2) {  // :1269
                float bestuR = kr[bestIdxR].x;
                float disparity = uL - bestuR;
                if (disparity >= minD && disparity < maxD) {
                    if (disparity <= 0) {
                        disparity = 0.01f;
                        bestuR = uL - 0.01f;
                    }
                    res_depth = p.mbf / disparity;
                    res_u = bestuR;
                    res_dist = bestDist;
                    res_idx = bestIdxR;
                }
            }
        }
    }
    if (lane == 0) {
        out.u_right[o] = res_u;
        out.depth[o] = res_depth;
        out.best_dist[o] = res_dist;
        out.best_idx[o] = res_idx;
        if (counted) atomicAdd(&out.nmatched[pair], 1);
    }
}

__global__ __launch_bounds__(256) void k_stereo_cut(const int* __restrict__ cnt_dev, int nl_host, GfoStereoDev out,
                                                    int out_stride)
{
    __shared__ int hist[128];
    __shared__ int s_med, s_drop;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nl = cnt_dev ? cnt_dev[2 * pair] : nl_host;
    const long long o = (long long)pair * out_stride;
    if (tid < 128) hist[tid] = 0;
    if (tid == 0) s_drop = 0;
    __syncthreads();
    for (int i = tid; i < nl; i += 256) {
        const int d = out.best_dist[o + i];
        if (d >= 0) atomicAdd(&hist[d], 1);
    }
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
    if (s_med < 0) return;
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
    if (tid == 0 && s_drop) out.nmatched[pair] -= s_drop;
}

void gfo_launch_stereo(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, const int* cnt_dev, int nl_host,
                       const gfo_keypoint* kr, const uint8_t* dr, const int* /*unused*/, int nr_host,
                       long long pair_stride_kp, int npairs, const float* d_scale, const gfo_stereo_params& p,
                       const float* min_d, const float* max_d, GfoStereoDev out, int out_stride)
{
    (void)hipMemsetAsync(out.nmatched, 0, sizeof(int) * npairs, c->stream);
    const int max_nl = cnt_dev ? out_stride : nl_host;
    if (max_nl > 0) {
        dim3 grid((max_nl + 3) / 4, npairs);
        gfo_prof_begin(c, ST_STEREO);
        hipLaunchKernelGGL(k_stereo_match, grid, dim3(256), 0, c->stream, kl, dl, kr, dr, cnt_dev, nl_host, nr_host,
                           pair_stride_kp, d_scale, p, min_d, max_d, out, out_stride);
        gfo_prof_end(c);
        gfo_prof_begin(c, ST_STEREO_CUT);
        hipLaunchKernelGGL(k_stereo_cut, dim3(npairs), dim3(256), 0, c->stream, cnt_dev, nl_host, out, out_stride);
        gfo_prof_end(c);
    }
}
