// k_project.hip -- Frame::AssignFeaturesToGrid / GetFeaturesInArea (Frame.cc:461-476, 593-658) and
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORBmatcher.cc:155-249).
//
// The reference is ORDER-DEPENDENT: map points are visited in vector order, every accepted match
// writes F.mvpMapPoints[bestIdx], and later map points skip keypoints already taken by a point with
// observations (ORBmatcher.cc:197-199, 233).  The device version reproduces the serial result exactly
// with a Jacobi fixed-point iteration:
//   state  = each map point's tentative pick;
//   update = every map point i re-evaluates its candidates treating keypoint k as taken iff it was
//            taken on entry or some map point j < i (with observations) currently picks k;
//   the serial answer is the unique fixed point, map point i is final after at most (its rank among
//   the matching points) rounds, and "nothing changed in a round" proves convergence.
// Points whose best distance exceeds TH_HIGH under the entry state can never match (more blocking only
// removes candidates), so only the few thousand "live" points are re-evaluated after round 0.
//
// Candidate order matters only for ties (strict '<' keeps the first candidate, :212-225).  The
// reference iterates cells (ix, iy) then the cell's list in keypoint order, so "first" is the minimum
// of (dist, ix, iy, idx): candidates can be scanned in any order with that 64-bit key, and the
// best / second-best pair is the two smallest keys (a stable sort by distance).
#include "gfo_internal.h"

#define GRID_COLS 64   // FRAME_GRID_COLS, Frame.h:92
#define GRID_ROWS 48   // FRAME_GRID_ROWS, Frame.h:93
#define NCELL (GRID_COLS * GRID_ROWS)
#define TH_HIGH 100    // ORBmatcher.cc:57

struct ProjArgs {
    const gfo_keypoint* kp;
    const uint8_t* desc;
    const float* u_right;     // may be null
    const uint8_t* taken0;    // may be null
    int n;
    gfo_frame_bounds fb;
    float inv_w, inv_h;
    const gfo_proj_query* q;  // one per projected map point, in the reference's visiting order
    const uint8_t* mp_desc;
    int m;
    int use_ratio;
    float nn_ratio;
    int th_dist;
    const float* kp_angle;    // rotation check only
    int* rot_bin;             // [m] histogram bin of an accepted query, -1 otherwise
    // grid
    int* cell_start;          // [NCELL+1]
    int* cell_items;          // [n]
    unsigned short* kp_cell;  // [n] ix<<8|iy, 0xFFFF = outside
    // state
    int* pick;                // [m] keypoint index or -1
    int* pick_dist;           // [m]
    int* live;                // [m] compacted indices of live map points
    int* counters;            // [0] n_live, [1] changed, [2] nmatches
    int* block_by;            // [n] lowest live map point (with observations) currently picking k
    int* out_mp;              // [n]
    int* out_score;           // [n]
};

// one workgroup builds the whole grid (N <= 65535)
__global__ __launch_bounds__(1024) void k_grid_build(ProjArgs a)
{
    __shared__ int s_cnt[NCELL];
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    for (int c = tid; c < NCELL; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < a.n; i += 1024) {
        // Frame::PosInGrid, Frame.cc:648-658 (round half away from zero)
        const int px = (int)roundf((a.kp[i].x - a.fb.min_x) * a.inv_w);
        const int py = (int)roundf((a.kp[i].y - a.fb.min_y) * a.inv_h);
        unsigned short cell = 0xFFFF;
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) {
            cell = (unsigned short)((px << 8) | py);
            atomicAdd(&s_cnt[px * GRID_ROWS + py], 1);
        }
        a.kp_cell[i] = cell;
    }
    __syncthreads();
    // exclusive scan of 3072 counters: 3 per thread
    int loc[3], s = 0;
    for (int k = 0; k < 3; k++) { loc[k] = s_cnt[tid * 3 + k]; s += loc[k]; }
    s_part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - s;
    for (int k = 0; k < 3; k++) {
        a.cell_start[tid * 3 + k] = run;
        s_cnt[tid * 3 + k] = run;   // becomes the fill cursor
        run += loc[k];
    }
    if (tid == 1023) a.cell_start[NCELL] = run;
    __syncthreads();
    for (int i = tid; i < a.n; i += 1024) {
        const unsigned short cell = a.kp_cell[i];
        if (cell != 0xFFFF) a.cell_items[atomicAdd(&s_cnt[(cell >> 8) * GRID_ROWS + (cell & 0xFF)], 1)] = i;
    }
}

__device__ __forceinline__ int hamming_u4(const uint4 a0, const uint4 a1, const uint4* __restrict__ b)
{
    const uint4 b0 = b[0], b1 = b[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// One wavefront evaluates one map point.  round0: all map points, builds the live list.
template <bool ROUND0>
__global__ __launch_bounds__(256) void k_project_eval(ProjArgs a)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform: keep it scalar
    const int slot = blockIdx.x * 4 + wave;
    int iMP;
    if (ROUND0) {
        if (slot >= a.m) return;
        iMP = slot;
    } else {
        if (slot >= a.counters[0]) return;
        iMP = a.live[slot];
    }
    const gfo_proj_query mp = a.q[iMP];
    int new_pick = -1, new_dist = 256;
    bool is_live = false;
    if (mp.flags & 1) {  // visible and usable (mbTrackInView && !isBad(), :163-167 / :1467-1490)
        const float rs = mp.radius;
        // GetFeaturesInArea(u, v, rs, minLevel, maxLevel), Frame.cc:593-646
        const float x = mp.u, y = mp.v;
        const int minLevel = mp.min_level, maxLevel = mp.max_level;
        int cx0 = max(0, (int)floorf((x - a.fb.min_x - rs) * a.inv_w));
        int cx1 = min(GRID_COLS - 1, (int)ceilf((x - a.fb.min_x + rs) * a.inv_w));
        int cy0 = max(0, (int)floorf((y - a.fb.min_y - rs) * a.inv_h));
        int cy1 = min(GRID_ROWS - 1, (int)ceilf((y - a.fb.min_y + rs) * a.inv_h));
        if (!(cx0 >= GRID_COLS || cx1 < 0 || cy0 >= GRID_ROWS || cy1 < 0)) {
            const bool check_levels = (minLevel > 0) || (maxLevel >= 0);
            const uint4* dmp = reinterpret_cast<const uint4*>(a.mp_desc + (long long)iMP * 32);
            const uint4 a0 = dmp[0], a1 = dmp[1];
            unsigned long long k1 = ~0ull, k2 = ~0ull;  // two smallest (dist, ix, iy, idx) keys
            for (int ix = cx0; ix <= cx1; ix++) {
                // cells (ix, cy0..cy1) are contiguous in the CSR
                const int beg = max(a.cell_start[ix * GRID_ROWS + cy0], 0);
                const int end = min(a.cell_start[ix * GRID_ROWS + cy1 + 1], a.n);  // bounded by the keypoint count whatever the table holds
                for (int j = beg + lane; j < end; j += 64) {
                    const int i = a.cell_items[j];
                    const gfo_keypoint kp = a.kp[i];
                    if (check_levels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    if (!(fabsf(kp.x - x) < rs && fabsf(kp.y - y) < rs)) continue;
                    // F.mvpMapPoints[idx] with Observations() > 0, :197-199
                    bool blocked = a.taken0 && a.taken0[i];
                    if (!ROUND0) blocked = blocked || a.block_by[i] < iMP;
                    if (blocked) continue;
                    if (a.u_right && a.u_right[i] > 0) {  // :201-206
                        const float er = fabsf(mp.ur - a.u_right[i]);
                        if (er > rs) continue;
                    }
                    const unsigned dist = (unsigned)hamming_u4(a0, a1, reinterpret_cast<const uint4*>(a.desc + (long long)i * 32));
                    const unsigned iy = a.kp_cell[i] & 0xFF;
                    const unsigned long long key = ((unsigned long long)dist << 32) | ((unsigned long long)ix << 26) |
                                                   ((unsigned long long)iy << 20) | (unsigned)i;
                    if (key < k1) { k2 = k1; k1 = key; }
                    else if (key < k2) k2 = key;
                }
            }
            // wave merge of the per-lane (k1, k2) pairs
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
                const unsigned long long lo = k1 < o1 ? k1 : o1;
                const unsigned long long hi = k1 < o1 ? o1 : k1;
                const unsigned long long s2 = k2 < o2 ? k2 : o2;
                k1 = lo;
                k2 = hi < s2 ? hi : s2;
            }
            if (k1 != ~0ull) {
                const int bestDist = (int)(k1 >> 32);
                const int bestIdx = (int)(k1 & 0xFFFFF);
                if (bestDist <= a.th_dist) {  // :228 / :1536
                    is_live = true;
                    bool accept = true;
                    if (a.use_ratio && k2 != ~0ull) {
                        const int bestDist2 = (int)(k2 >> 32);
                        const int idx2 = (int)(k2 & 0xFFFFF);
                        const int bestLevel = a.kp[bestIdx].octave, bestLevel2 = a.kp[idx2].octave;
                        if (bestLevel == bestLevel2 && (float)bestDist > a.nn_ratio * (float)bestDist2) accept = false;  // :230
                    }
                    // (no second candidate: bestLevel2 = -1 != bestLevel, bestDist2 = 256 -> always accepted)
                    if (accept) {
                        new_pick = bestIdx;
                        new_dist = bestDist;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        if (ROUND0) {
            a.pick[iMP] = new_pick;
            a.pick_dist[iMP] = new_dist;
            if (is_live) a.live[atomicAdd(&a.counters[0], 1)] = iMP;
        } else {
            if (a.pick[iMP] != new_pick || a.pick_dist[iMP] != new_dist) {
                atomicOr(&a.counters[1], 1);
            }
            // written to the shadow half, swapped by the host each round (Jacobi: reads see the old state)
            a.pick[a.m + iMP] = new_pick;
            a.pick_dist[a.m + iMP] = new_dist;
        }
    }
}

// block_by[k] = min live map point WITH observations whose current pick is k
__global__ void k_project_claims(ProjArgs a, int nlive)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlive) return;
    const int iMP = a.live[t];
    const int k = a.pick[iMP];
    if (k >= 0 && (a.q[iMP].flags & 4)) atomicMin(&a.block_by[k], iMP);
}

__global__ void k_project_commit(ProjArgs a, int nlive)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlive) return;
    const int iMP = a.live[t];
    a.pick[iMP] = a.pick[a.m + iMP];
    a.pick_dist[iMP] = a.pick_dist[a.m + iMP];
}

// final owner of a keypoint = the LAST accepted map point that picked it (:233 overwrites)
__global__ void k_project_owner(ProjArgs a, int nlive)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlive) return;
    const int iMP = a.live[t];
    const int k = a.pick[iMP];
    if (k >= 0) {
        atomicMax(&a.out_mp[k], iMP);
        atomicAdd(&a.counters[2], 1);
    }
}

__global__ void k_project_score(ProjArgs a, int nlive)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlive) return;
    const int iMP = a.live[t];
    const int k = a.pick[iMP];
    if (k >= 0 && a.out_mp[k] == iMP) a.out_score[k] = a.pick_dist[iMP];
}

static inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

#define HISTO_LENGTH 30  // ORBmatcher.cc:59

// rotation consistency (ORBmatcher.cc:1548-1591): bin of every accepted query, histogram, the reference's
// three-maxima scan, then every query in a discarded bin clears the keypoint it took and costs one match.
__global__ __launch_bounds__(256) void k_project_rotation(ProjArgs a, int nlive)
{
    __shared__ int histo[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int s_drop;
    const int tid = threadIdx.x;
    if (tid < HISTO_LENGTH) histo[tid] = 0;
    if (tid == 0) s_drop = 0;
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    for (int t = tid; t < nlive; t += 256) {
        const int iq = a.live[t];
        const int k = a.pick[iq];
        int bin = -1;
        if (k >= 0) {
            float rot = a.q[iq].angle - a.kp_angle[k];
            if (rot < 0.0f) rot += 360.0f;
            bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            atomicAdd(&histo[bin], 1);
        }
        a.rot_bin[iq] = bin;
    }
    __syncthreads();
    if (tid == 0) {  // ComputeThreeMaxima, ORBmatcher.cc:1723-1764
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = histo[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
        keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
    }
    __syncthreads();
    int drop = 0;
    for (int t = tid; t < nlive; t += 256) {
        const int iq = a.live[t];
        const int b = a.rot_bin[iq];
        if (b >= 0 && b != keep[0] && b != keep[1] && b != keep[2]) {
            a.out_mp[a.pick[iq]] = -1;  // benign race: every writer stores -1
            drop++;
        }
    }
    if (drop) atomicAdd(&s_drop, drop);
    __syncthreads();
    if (tid == 0) a.counters[2] -= s_drop;
}

#define PTRY(c, expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return GFO_ERR_DEVICE;                                                                \
        }                                                                                         \
    } while (0)

extern "C" int gfo_search_by_projection_queries(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc,
                                                const float* u_right, const float* kp_angle, int n,
                                                const gfo_frame_bounds* fb, const gfo_proj_query* queries,
                                                const uint8_t* q_desc, int m, const gfo_proj_mode* mode,
                                                const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score, int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!fb || !mode || !out_mp || !out_score || !nmatches || n < 0 || m < 0 || (n > 0 && (!kp_un || !desc)) ||
        (m > 0 && (!queries || !q_desc)) || (mode->check_orientation && n > 0 && !kp_angle)) {
        c->err = "gfo_search_by_projection: bad argument";
        return GFO_ERR_INVALID;
    }
    if (n > 65535) {
        c->err = "gfo_search_by_projection: more than 65535 keypoints";
        return GFO_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n; i++) { out_mp[i] = -1; out_score[i] = 0; }
    if (n == 0 || m == 0) return GFO_OK;
    PTRY(c, hipSetDevice(c->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = al256(off + bytes); return o; };
    const size_t o_kp = take(sizeof(gfo_keypoint) * n), o_desc = take(32 * (size_t)n), o_ur = take(4 * (size_t)n),
                 o_tk = take(n), o_ang = take(4 * (size_t)n), o_q = take(sizeof(gfo_proj_query) * m),
                 o_mpd = take(32 * (size_t)m), o_cs = take(4 * (NCELL + 1)), o_ci = take(4 * (size_t)n),
                 o_kc = take(2 * (size_t)n), o_pick = take(8 * (size_t)m), o_pd = take(8 * (size_t)m),
                 o_live = take(4 * (size_t)m), o_cnt = take(16), o_bb = take(4 * (size_t)n), o_om = take(4 * (size_t)n),
                 o_os = take(4 * (size_t)n), o_rb = take(4 * (size_t)m);
    if (off > c->scratch_bytes) {
        (void)hipStreamSynchronize(c->stream);
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_bytes = 0;
        PTRY(c, hipMalloc(&c->d_scratch, off));
        c->scratch_bytes = off;
    }
    uint8_t* S = (uint8_t*)c->d_scratch;
    hipStream_t st = c->stream;
    PTRY(c, hipMemcpyAsync(S + o_kp, kp_un, sizeof(gfo_keypoint) * n, hipMemcpyHostToDevice, st));
    PTRY(c, hipMemcpyAsync(S + o_desc, desc, 32 * (size_t)n, hipMemcpyHostToDevice, st));
    if (u_right) PTRY(c, hipMemcpyAsync(S + o_ur, u_right, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    if (kp_taken) PTRY(c, hipMemcpyAsync(S + o_tk, kp_taken, n, hipMemcpyHostToDevice, st));
    if (kp_angle) PTRY(c, hipMemcpyAsync(S + o_ang, kp_angle, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    PTRY(c, hipMemcpyAsync(S + o_q, queries, sizeof(gfo_proj_query) * m, hipMemcpyHostToDevice, st));
    PTRY(c, hipMemcpyAsync(S + o_mpd, q_desc, 32 * (size_t)m, hipMemcpyHostToDevice, st));
    PTRY(c, hipMemsetAsync(S + o_cnt, 0, 16, st));
    PTRY(c, hipMemsetAsync(S + o_om, 0xFF, 4 * (size_t)n, st));
    PTRY(c, hipMemsetAsync(S + o_os, 0, 4 * (size_t)n, st));
    ProjArgs a{};
    a.kp = (const gfo_keypoint*)(S + o_kp);
    a.desc = S + o_desc;
    a.u_right = u_right ? (const float*)(S + o_ur) : nullptr;
    a.taken0 = kp_taken ? S + o_tk : nullptr;
    a.n = n;
    a.fb = *fb;
    a.inv_w = (float)GRID_COLS / (fb->max_x - fb->min_x);  // Frame.cc:129-130
    a.inv_h = (float)GRID_ROWS / (fb->max_y - fb->min_y);
    a.q = (const gfo_proj_query*)(S + o_q);
    a.mp_desc = S + o_mpd;
    a.m = m;
    a.use_ratio = mode->use_ratio;
    a.nn_ratio = mode->nn_ratio;
    a.th_dist = mode->th_dist;
    a.kp_angle = (const float*)(S + o_ang);
    a.rot_bin = (int*)(S + o_rb);
    a.cell_start = (int*)(S + o_cs);
    a.cell_items = (int*)(S + o_ci);
    a.kp_cell = (unsigned short*)(S + o_kc);
    a.pick = (int*)(S + o_pick);
    a.pick_dist = (int*)(S + o_pd);
    a.live = (int*)(S + o_live);
    a.counters = (int*)(S + o_cnt);
    a.block_by = (int*)(S + o_bb);
    a.out_mp = (int*)(S + o_om);
    a.out_score = (int*)(S + o_os);

    gfo_prof_begin(c, ST_PROJECT);
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_project_eval<true>, dim3((m + 3) / 4), dim3(256), 0, st, a);
    int cnt[4] = {0, 0, 0, 0};
    PTRY(c, hipMemcpyAsync(cnt, a.counters, 16, hipMemcpyDeviceToHost, st));
    PTRY(c, hipStreamSynchronize(st));
    const int nlive = cnt[0];
    int rounds = 0;
    if (nlive > 0) {
        const dim3 g1((nlive + 255) / 256), b1(256);
        for (;; rounds++) {
            if (rounds > nlive + 1) {
                c->err = "gfo_search_by_projection: fixed point not reached";
                return GFO_ERR_STATE;
            }
            PTRY(c, hipMemsetAsync(a.block_by, 0x7F, 4 * (size_t)n, st));
            PTRY(c, hipMemsetAsync(a.counters + 1, 0, 4, st));
            hipLaunchKernelGGL(k_project_claims, g1, b1, 0, st, a, nlive);
            hipLaunchKernelGGL(k_project_eval<false>, dim3((nlive + 3) / 4), dim3(256), 0, st, a);
            hipLaunchKernelGGL(k_project_commit, g1, b1, 0, st, a, nlive);
            PTRY(c, hipMemcpyAsync(cnt, a.counters, 16, hipMemcpyDeviceToHost, st));
            PTRY(c, hipStreamSynchronize(st));
            if (!cnt[1]) break;
        }
        hipLaunchKernelGGL(k_project_owner, g1, b1, 0, st, a, nlive);
        hipLaunchKernelGGL(k_project_score, g1, b1, 0, st, a, nlive);
        if (mode->check_orientation) hipLaunchKernelGGL(k_project_rotation, dim3(1), dim3(256), 0, st, a, nlive);
    }
    gfo_prof_end(c);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    PTRY(c, hipGetLastError());
    PTRY(c, hipMemcpyAsync(out_mp, a.out_mp, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    PTRY(c, hipMemcpyAsync(out_score, a.out_score, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    PTRY(c, hipMemcpyAsync(cnt, a.counters, 16, hipMemcpyDeviceToHost, st));
    PTRY(c, hipStreamSynchronize(st));
    *nmatches = cnt[2];
    c->last_project_rounds = rounds + 1;
    return GFO_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th): every map point becomes a query with
// r = RadiusByViewingCos(viewCos) (* th), window r * scale[level], levels [level-1, level] (ORBmatcher.cc:171-180).
extern "C" int gfo_search_by_projection(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right,
                                        int n, const float* sf, int nlevels, const gfo_frame_bounds* fb,
                                        const gfo_map_point* mps, const uint8_t* mp_desc, int m, float th, float nn_ratio,
                                        const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score, int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!sf || nlevels < 1 || nlevels > GFO_MAX_LEVELS || m < 0 || (m > 0 && !mps)) {
        c->err = "gfo_search_by_projection: bad argument";
        return GFO_ERR_INVALID;
    }
    std::vector<gfo_proj_query> q((size_t)(m > 0 ? m : 1));
    const bool bFactor = th != 1.0f;
    for (int i = 0; i < m; i++) {
        const gfo_map_point& p = mps[i];
        gfo_proj_query& d = q[i];
        const int lvl = p.level;
        float r = (double)p.view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, :243-249
        if (bFactor) r *= th;
        d.u = p.proj_x;
        d.v = p.proj_y;
        d.ur = p.proj_xr;
        d.radius = (lvl >= 0 && lvl < nlevels) ? r * sf[lvl] : 0.f;
        d.min_level = lvl - 1;
        d.max_level = lvl;
        d.angle = 0.f;
        d.flags = (((p.flags & 1) && !(p.flags & 2) && lvl >= 0 && lvl < nlevels) ? 1 : 0) | (p.flags & 4);
    }
    gfo_proj_mode mode = {1, nn_ratio, TH_HIGH, 0};
    return gfo_search_by_projection_queries(c, kp_un, desc, u_right, nullptr, n, fb, q.data(), mp_desc, m, &mode, kp_taken,
                                            out_mp, out_score, nmatches);
}
