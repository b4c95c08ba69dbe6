// k_bow.hip -- ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (ORBmatcher.cc:270-404)
// on flattened arrays: the two DBoW2::FeatureVector maps arrive as CSR (node ids ascending).
//
// Keypoints of different vocabulary nodes are disjoint on both sides, so nodes are independent; inside
// a node the reference is sequential (every accepted match blocks its frame keypoint for the following
// keyframe keypoints, :320,343).  Device form: ONE WAVEFRONT PER COMMON NODE walks the node's keyframe
// keypoints in order; for each, the 64 lanes sweep the node's frame keypoints, take the 256-bit
// Hamming distance and wave-reduce the two smallest (distance, position) keys -- "first minimum wins"
// (:326-336) is the smaller position.  The rotation-consistency filter (:349-360, 382-401,
// ComputeThreeMaxima :1723-1764) is one small workgroup: 30-bin histogram, the reference's exact
// three-maxima scan on one lane, then the sweep that clears the other bins.
// The node intersection (the lower_bound walk, :292-380) is a host-side merge of two sorted id lists.
#include "gfo_internal.h"
#include "k_wave.inc"

#define TH_LOW 50        // ORBmatcher.cc:58
#define HISTO_LENGTH 30  // ORBmatcher.cc:59

struct BowArgs {
    const uint8_t* kf_desc;
    const float* kf_angle;
    const uint8_t* kf_valid;
    const int* kf_start;       // CSR of the keyframe feature vector
    const unsigned* kf_items;
    const uint8_t* f_desc;
    const float* f_angle;
    const int* f_start;
    const unsigned* f_items;
    const int2* pairs;         // (kf node slot, frame node slot) of every common node
    int npairs;
    int n_f;
    float nn_ratio;
    int check_ori;
    int* out;                  // [n_f] keyframe keypoint index or -1
    int* rot_bin;              // [n_f]
    int* counters;             // [0] nmatches
    int* h_out;                // host-visible pinned copies of `out` / counters[0], written by k_bow_rotation (null: copied back)
    int* h_count;
    // BUDGETING_FEATURE_MATCHING (gfo_search_by_bow_budget), 0 = off: every accepted match records its node and its ordinal among the
    // node's accepted matches, every node how many it accepted; k_bow_budget then keeps of node j the first min(full_j, max(1, K - before_j))
    int th_low;                // accept bestDist1 <= th_low: TH_LOW (:339), TH_LOW - 1 for the keyframe pair's strict test (:713)
    int max_matches;
    int* ord;                  // [n_f]
    int* node_of;              // [n_f]
    int* node_acc;             // [npairs]
};

// A node with more frame keypoints than the lanes' registers hold (> 64 * BOW_R): every keyframe keypoint sweeps the node's frame
// keypoints from memory, `out` carries the taken state (this wave is its only writer: same-wave program order + the fence below).
__device__ __forceinline__ void bow_node_sweep(const BowArgs& a, int pi, int kb, int ke, int fb, int fe, int lane)
{
    const float factor = 1.0f / HISTO_LENGTH;  // :284 (applied to degrees, as the reference does)
    int accepted = 0;
    for (int ik = kb; ik < ke; ik++) {
        const unsigned realIdxKF = a.kf_items[ik];
        if (!a.kf_valid[realIdxKF]) continue;  // :306-310 (wave-uniform)
        const uint4* dk = reinterpret_cast<const uint4*>(a.kf_desc + (long long)realIdxKF * 32);
        const uint4 k0 = dk[0], k1 = dk[1];
        unsigned b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;  // two smallest (dist << 20 | position)
        for (int j = fb + lane; j < fe; j += 64) {
            const unsigned realIdxF = a.f_items[j];
            if (a.out[realIdxF] >= 0) continue;  // :320 already matched (this wave wrote it: same-wave program order)
            const uint4* df = reinterpret_cast<const uint4*>(a.f_desc + (long long)realIdxF * 32);
            const uint4 f0 = df[0], f1 = df[1];
            const unsigned dist = __popc(k0.x ^ f0.x) + __popc(k0.y ^ f0.y) + __popc(k0.z ^ f0.z) + __popc(k0.w ^ f0.w) +
                                  __popc(k1.x ^ f1.x) + __popc(k1.y ^ f1.y) + __popc(k1.z ^ f1.z) + __popc(k1.w ^ f1.w);
            const unsigned key = (dist << 20) | (unsigned)(j - fb);
            if (key < b1) { b2 = b1; b1 = key; }
            else if (key < b2) b2 = key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned o1 = (unsigned)__shfl_xor((int)b1, o), o2 = (unsigned)__shfl_xor((int)b2, o);
            const unsigned lo = min(b1, o1), hi = max(b1, o1);
            b1 = lo;
            b2 = min(hi, min(b2, o2));
        }
        if (b1 == 0xFFFFFFFFu) continue;
        const int bestDist1 = (int)(b1 >> 20);
        const int bestDist2 = b2 == 0xFFFFFFFFu ? 256 : (int)(b2 >> 20);
        if (bestDist1 <= a.th_low && (float)bestDist1 < a.nn_ratio * (float)bestDist2) {  // :339-341 (:713-715)
            const unsigned bestIdxF = a.f_items[fb + (int)(b1 & 0xFFFFF)];
            if (lane == 0) {
                a.out[bestIdxF] = (int)realIdxKF;
                if (a.check_ori) {
                    float rot = a.kf_angle[realIdxKF] - a.f_angle[bestIdxF];
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    a.rot_bin[bestIdxF] = bin;
                }
                if (a.max_matches > 0) { a.ord[bestIdxF] = accepted; a.node_of[bestIdxF] = pi; }
            }
            accepted++;
            // make lane 0's store visible to the whole wave's next sweep
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (lane == 0 && accepted) atomicAdd(&a.counters[0], accepted);
    if (lane == 0 && a.max_matches > 0) a.node_acc[pi] = accepted;
}


#define BOW_R 4   // frame keypoints of the node a lane keeps in registers (descriptor, index, angle): nodes up to 256 frame keypoints

// One wavefront per common node.  The node's FRAME side is loaded once into registers (lane l holds positions l, l+64, ...), with a
// lane-local "still free" bit per slot -- the node's frame keypoints belong to no other node, so nobody else writes them; the
// KEYFRAME side is loaded 64 keypoints at a time, one per lane, and broadcast keypoint by keypoint with v_readlane.  The walk over
// the keyframe keypoints stays sequential (it IS, :320,343) but an iteration is ~60 register instructions and two DPP reductions
// instead of three dependent trips to memory and a fence: 129 us -> see profiles/NOTEBOOK.md for the EuRoC frame over 92 nodes.
__global__ __launch_bounds__(256) void k_bow_match(BowArgs a)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform: keep it scalar
    const int pi = blockIdx.x * 4 + wave;
    if (pi >= a.npairs) return;
    const int2 pr = a.pairs[pi];
    const int kb = a.kf_start[pr.x], ke = a.kf_start[pr.x + 1];
    const int fb = a.f_start[pr.y], fe = a.f_start[pr.y + 1];
    if (fe - fb > 64 * BOW_R) {
        bow_node_sweep(a, pi, kb, ke, fb, fe, lane);
        return;
    }
    const float factor = 1.0f / HISTO_LENGTH;  // :284 (applied to degrees, as the reference does)
    uint4 f0[BOW_R], f1[BOW_R];
    unsigned fidx[BOW_R];
    float fang[BOW_R];
    unsigned free_m = 0;
#pragma unroll
    for (int r = 0; r < BOW_R; r++) {
        const int j = fb + r * 64 + lane;
        fidx[r] = 0; fang[r] = 0.f;
        f0[r] = make_uint4(0, 0, 0, 0); f1[r] = f0[r];
        if (j < fe) {
            fidx[r] = a.f_items[j];
            const uint4* df = reinterpret_cast<const uint4*>(a.f_desc + (long long)fidx[r] * 32);
            f0[r] = df[0]; f1[r] = df[1];
            if (a.check_ori) fang[r] = a.f_angle[fidx[r]];
            if (a.out[fidx[r]] < 0) free_m |= 1u << r;   // :320 (cleared by the host; a slot this wave fills clears its bit)
        }
    }
    int accepted = 0;
    for (int c0 = kb; c0 < ke; c0 += 64) {
        const int ik = c0 + lane;
        unsigned kidx = 0;
        bool kval = false;
        uint4 k0 = make_uint4(0, 0, 0, 0), k1 = k0;
        float kang = 0.f;
        if (ik < ke) {
            kidx = a.kf_items[ik];
            kval = a.kf_valid[kidx] != 0;   // :306-310
            const uint4* dk = reinterpret_cast<const uint4*>(a.kf_desc + (long long)kidx * 32);
            k0 = dk[0]; k1 = dk[1];
            if (a.check_ori) kang = a.kf_angle[kidx];
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(kval);
        while (todo) {
            const int i = __builtin_ctzll(todo);   // next valid keyframe keypoint, in the node's order
            todo &= todo - 1;
            const unsigned q0 = __builtin_amdgcn_readlane((int)k0.x, i), q1 = __builtin_amdgcn_readlane((int)k0.y, i),
                           q2 = __builtin_amdgcn_readlane((int)k0.z, i), q3 = __builtin_amdgcn_readlane((int)k0.w, i),
                           q4 = __builtin_amdgcn_readlane((int)k1.x, i), q5 = __builtin_amdgcn_readlane((int)k1.y, i),
                           q6 = __builtin_amdgcn_readlane((int)k1.z, i), q7 = __builtin_amdgcn_readlane((int)k1.w, i);
            unsigned b1 = 0xFFFFFFFFu, b2 = 0xFFFFFFFFu;  // two smallest (dist << 20 | position)
#pragma unroll
            for (int r = 0; r < BOW_R; r++) {
                if (free_m & (1u << r)) {
                    const unsigned dist = __popc(q0 ^ f0[r].x) + __popc(q1 ^ f0[r].y) + __popc(q2 ^ f0[r].z) + __popc(q3 ^ f0[r].w) +
                                          __popc(q4 ^ f1[r].x) + __popc(q5 ^ f1[r].y) + __popc(q6 ^ f1[r].z) + __popc(q7 ^ f1[r].w);
                    const unsigned key = (dist << 20) | (unsigned)(r * 64 + lane);
                    if (key < b1) { b2 = b1; b1 = key; }
                    else if (key < b2) b2 = key;
                }
            }
            // the two smallest keys of the wave: keys are distinct (the position is part of them), so the runner-up is the smallest of
            // "every lane's best that is not the winner, and the winner's lane's second"
            const unsigned w1 = st_wave_min(b1);
            if (w1 == 0xFFFFFFFFu) continue;
            const unsigned w2 = st_wave_min(b1 == w1 ? b2 : b1);
            const int bestDist1 = (int)(w1 >> 20);
            const int bestDist2 = w2 == 0xFFFFFFFFu ? 256 : (int)(w2 >> 20);
            if (bestDist1 <= a.th_low && (float)bestDist1 < a.nn_ratio * (float)bestDist2) {  // :339-341 (:713-715)
                const int pos = (int)(w1 & 0xFFFFF);
                const unsigned realIdxKF = (unsigned)__builtin_amdgcn_readlane((int)kidx, i);
                const float ka = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kang), i));
                if (lane == (pos & 63)) {
                    const int r = pos >> 6;
                    unsigned bestIdxF = fidx[0];
                    float fa = fang[0];
#pragma unroll
                    for (int t = 1; t < BOW_R; t++)
                        if (r == t) { bestIdxF = fidx[t]; fa = fang[t]; }
                    free_m &= ~(1u << r);
                    a.out[bestIdxF] = (int)realIdxKF;
                    if (a.check_ori) {
                        float rot = ka - fa;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        a.rot_bin[bestIdxF] = bin;
                    }
                    if (a.max_matches > 0) { a.ord[bestIdxF] = accepted; a.node_of[bestIdxF] = pi; }
                }
                accepted++;
            }
        }
    }
    if (lane == 0 && accepted) atomicAdd(&a.counters[0], accepted);
    if (lane == 0 && a.max_matches > 0) a.node_acc[pi] = accepted;
}

// BUDGETING_FEATURE_MATCHING, ORBmatcher.cc:360-365: `if (nmatches >= MAX_NUM_FEATURE_MATCHING) break;` sits in the loop over ONE node's
// keyframe keypoints.  Nodes are independent (a frame keypoint belongs to one node), so the unbudgeted kernel above has every node's
// accepted matches in order; the budget keeps of node j the first t_j = min(full_j, max(1, K - before_j)) of them, before_j = the
// matches kept in the nodes in front of it (common nodes in ascending id order = pair order).  One workgroup: the serial recurrence
// over the nodes on one lane (a few hundred nodes), then the sweep that un-matches what lies beyond a node's share.
__global__ __launch_bounds__(256) void k_bow_budget(BowArgs a)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        int before = 0;
        for (int j = 0; j < a.npairs; j++) {
            const int full = a.node_acc[j];
            const int want = a.max_matches - before > 1 ? a.max_matches - before : 1;
            const int t = full < want ? full : want;
            a.node_acc[j] = t;
            before += t;
        }
        a.counters[0] = before;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < a.n_f; i += 256) {
        if (a.out[i] < 0) continue;
        if (a.ord[i] >= a.node_acc[a.node_of[i]]) {
            a.out[i] = -1;
            a.rot_bin[i] = -1;
        }
    }
}

__global__ __launch_bounds__(256) void k_bow_rotation(BowArgs a)
{
    __shared__ int histo[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int s_drop;
    const int tid = threadIdx.x;
    if (tid < HISTO_LENGTH) histo[tid] = 0;
    if (tid == 0) s_drop = 0;
    __syncthreads();
    for (int i = tid; i < a.n_f; i += 256) {
        const int b = a.rot_bin[i];
        if (b >= 0) atomicAdd(&histo[b], 1);
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
    for (int i = tid; i < a.n_f; i += 256) {
        const int b = a.rot_bin[i];
        int o = a.out[i];
        if (b >= 0 && b != keep[0] && b != keep[1] && b != keep[2]) {
            a.out[i] = o = -1;
            drop++;
        }
        if (a.h_out) a.h_out[i] = o;   // the call's answer, straight into the caller-side pinned block
    }
    if (drop) atomicAdd(&s_drop, drop);
    __syncthreads();
    if (tid == 0) {
        a.counters[0] -= s_drop;
        if (a.h_count) *a.h_count = a.counters[0];
    }
}

// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:770-935), local mapping's matcher for NEW map points: keypoints without a map point in two
// keyframes, node by node of the shared vocabulary level.  Unlike the overloads above nothing a keypoint finds hides a candidate from the next
// one -- vbMatched2 is declared and tested but never set (:790, :832) -- so every keypoint of the first keyframe is a search of its own: the
// candidate with the smallest distance <= TH_LOW among those that pass the stereo / epipole / epipolar-line gates (:836-860), the LAST such
// candidate of the node's list on a tie (`dist > bestDist` skips, an equal distance replaces).  One wavefront per common node; the lanes take
// the node's second-keyframe keypoints side by side for one first-keyframe keypoint at a time.
struct TriArgs {
    const uint8_t* desc1; const gfo_keypoint* kp1; const uint8_t* flag1;   // flag: bit 0 has a map point, bit 1 mvuRight >= 0
    const int* start1; const unsigned* items1;
    const uint8_t* desc2; const gfo_keypoint* kp2; const uint8_t* flag2;
    const int* start2; const unsigned* items2;
    const int2* pairs; int npairs;
    float f12[9]; float ex, ey;
    const float* scale2; const float* sigma2;   // pKF2->mvScaleFactors, mvLevelSigma2 (device)
    int only_stereo, check_ori, n1;
    int* out; int* rot_bin; int* counters;
};

// CheckDistEpipolarLine (ORBmatcher.cc:251-268): the expressions as written, float, left to right, un-fused; the last comparison in double
__device__ __forceinline__ bool tri_epipolar_ok(const TriArgs& a, float x1, float y1, float x2, float y2, int octave2)
{
    const float la = x1 * a.f12[0] + y1 * a.f12[3] + a.f12[6];
    const float lb = x1 * a.f12[1] + y1 * a.f12[4] + a.f12[7];
    const float lc = x1 * a.f12[2] + y1 * a.f12[5] + a.f12[8];
    const float num = la * x2 + lb * y2 + lc;
    const float den = la * la + lb * lb;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)a.sigma2[octave2];
}

#define TRI_SPLIT 4
__global__ __launch_bounds__(256) void k_bow_triangulate(TriArgs a)
{
    // a common node is shared by TRI_WAVES wavefronts (4 of a block x gridDim.y blocks): the first keyframe's keypoints of the node are
    // independent searches (nothing one finds hides a candidate from the next), dealt round robin
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int2 pr = a.pairs[blockIdx.x];
    const int b1 = a.start1[pr.x], e1 = a.start1[pr.x + 1];
    const int b2 = a.start2[pr.y], e2 = a.start2[pr.y + 1];
    const float factor = 1.0f / HISTO_LENGTH;
    int accepted = 0;
    for (int i1 = b1 + (int)blockIdx.y * 4 + wave; i1 < e1; i1 += 4 * (int)gridDim.y) {
        const unsigned idx1 = a.items1[i1];
        const unsigned fl1 = a.flag1[idx1];
        if (fl1 & 1) continue;                               // pMP1: there is a map point already (:812-815)
        const bool stereo1 = (fl1 & 2) != 0;
        if (a.only_stereo && !stereo1) continue;             // :819-821
        const gfo_keypoint k1 = a.kp1[idx1];
        const uint4* d1 = reinterpret_cast<const uint4*>(a.desc1 + (long long)idx1 * 32);
        const uint4 p0 = d1[0], p1 = d1[1];
        unsigned best = 0xFFFFFFFFu;                         // dist << 20 | (0xFFFFF - position): the smallest distance, the LAST position on a tie
        for (int j = b2 + lane; j < e2; j += 64) {
            const unsigned idx2 = a.items2[j];
            const unsigned fl2 = a.flag2[idx2];
            if (fl2 & 1) continue;                           // vbMatched2[idx2] (never set) || pMP2 (:832-833)
            const bool stereo2 = (fl2 & 2) != 0;
            if (a.only_stereo && !stereo2) continue;
            const uint4* d2 = reinterpret_cast<const uint4*>(a.desc2 + (long long)idx2 * 32);
            const uint4 q0 = d2[0], q1 = d2[1];
            const unsigned dist = __popc(p0.x ^ q0.x) + __popc(p0.y ^ q0.y) + __popc(p0.z ^ q0.z) + __popc(p0.w ^ q0.w) +
                                  __popc(p1.x ^ q1.x) + __popc(p1.y ^ q1.y) + __popc(p1.z ^ q1.z) + __popc(p1.w ^ q1.w);
            if (dist > TH_LOW) continue;                     // :845 (the running `dist > bestDist` only prunes what could not win)
            const gfo_keypoint k2 = a.kp2[idx2];
            const int oct2 = min(max(k2.octave, 0), GFO_MAX_LEVELS - 1);
            if (!stereo1 && !stereo2) {                      // :850-856: too close to the epipole
                const float distex = a.ex - k2.x, distey = a.ey - k2.y;
                if (distex * distex + distey * distey < 100 * a.scale2[oct2]) continue;
            }
            if (!tri_epipolar_ok(a, k1.x, k1.y, k2.x, k2.y, oct2)) continue;
            const unsigned key = (dist << 20) | (0xFFFFFu - (unsigned)(j - b2));
            best = min(best, key);
        }
        best = st_wave_min(best);
        if (best == 0xFFFFFFFFu) continue;
        const unsigned idx2 = a.items2[b2 + (int)(0xFFFFFu - (best & 0xFFFFFu))];
        if (lane == 0) {
            a.out[idx1] = (int)idx2;
            if (a.check_ori) {
                float rot = k1.angle - a.kp2[idx2].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                a.rot_bin[idx1] = bin;
            }
        }
        accepted++;
    }
    if (lane == 0 && accepted) atomicAdd(&a.counters[0], accepted);
}

#define BTRY(c, expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return GFO_ERR_DEVICE;                                                                \
        }                                                                                         \
    } while (0)

// A gfo_feature_vector is the caller's flattening of a DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>): the kernels
// index the descriptor rows with its items and walk its node ranges, so a stale or damaged one (an mFeatVec of another frame, a
// keypoint list that shrank) must be refused here instead of reading beside the arrays on the device.  O(items) on the host.
static const char* bow_check_feature_vector(const gfo_feature_vector* fv, int n)
{
    if (fv->n_nodes < 0) return "negative node count";
    if (fv->n_nodes == 0) return nullptr;
    if (!fv->node_ids || !fv->node_start) return "null node_ids / node_start";
    if (fv->node_start[0] != 0) return "node_start[0] must be 0";
    for (int i = 0; i < fv->n_nodes; i++) {
        if (fv->node_start[i + 1] < fv->node_start[i]) return "node_start must not decrease";
        if (i > 0 && fv->node_ids[i] <= fv->node_ids[i - 1]) return "node_ids must ascend strictly (std::map order)";
    }
    const int items = fv->node_start[fv->n_nodes];
    if (items > 0 && !fv->items) return "null items";
    for (int k = 0; k < items; k++)
        if (fv->items[k] >= (uint32_t)n) return "an item indexes past the keypoints";
    return nullptr;
}

extern "C" int gfo_search_by_bow(gfo_ctx* c, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_mp_valid,
                                 int n_kf, const gfo_feature_vector* kf_fv, const uint8_t* f_desc, const float* f_angle,
                                 int n_f, const gfo_feature_vector* f_fv, float nn_ratio, int check_orientation,
                                 int32_t* out_kf_idx, int* nmatches)
{
    return gfo_search_by_bow_budget(c, kf_desc, kf_angle, kf_mp_valid, n_kf, kf_fv, f_desc, f_angle, n_f, f_fv, nn_ratio, check_orientation, 0,
                                    out_kf_idx, nmatches);
}

#define BOW_UNAVAILABLE 0x7FFFFFFF   // `out` of a second-side keypoint without a usable map point: "taken" from the start (:690-696)

// f_valid (optional): the second side's map-point mask of the keyframe-pair overload; strict: `bestDist1 < TH_LOW` (:713)
static int bow_search(gfo_ctx* c, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_mp_valid,
                      int n_kf, const gfo_feature_vector* kf_fv, const uint8_t* f_desc, const float* f_angle, const uint8_t* f_valid,
                      int n_f, const gfo_feature_vector* f_fv, float nn_ratio, int check_orientation, int max_matches, bool strict,
                      int32_t* out_kf_idx, int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!kf_fv || !f_fv || !out_kf_idx || !nmatches || n_kf < 0 || n_f < 0 || (n_kf > 0 && (!kf_desc || !kf_mp_valid)) ||
        (n_f > 0 && !f_desc) || (check_orientation && n_kf > 0 && n_f > 0 && (!kf_angle || !f_angle))) {
        c->err = "gfo_search_by_bow: bad argument";
        return GFO_ERR_INVALID;
    }
    if (check_orientation) {
        // the rotation histogram is indexed with round((angle_kf - angle_f [+ 360]) / 30) (ORBmatcher.cc:346-355, which asserts the
        // bin): cv::KeyPoint::angle of oriented keypoints, 0..360 -- anything else would index beside the 30 bins
        for (int i = 0; i < n_kf; i++)
            if (!(kf_angle[i] >= 0.f && kf_angle[i] <= 360.f)) { c->err = "gfo_search_by_bow: keyframe keypoint angle outside 0..360"; return GFO_ERR_INVALID; }
        for (int i = 0; i < n_f; i++)
            if (!(f_angle[i] >= 0.f && f_angle[i] <= 360.f)) { c->err = "gfo_search_by_bow: frame keypoint angle outside 0..360"; return GFO_ERR_INVALID; }
    }
    if (const char* why = bow_check_feature_vector(kf_fv, n_kf)) {
        c->err = std::string("gfo_search_by_bow: keyframe feature vector: ") + why;
        return GFO_ERR_INVALID;
    }
    if (const char* why = bow_check_feature_vector(f_fv, n_f)) {
        c->err = std::string("gfo_search_by_bow: frame feature vector: ") + why;
        return GFO_ERR_INVALID;
    }
    *nmatches = 0;
    for (int i = 0; i < n_f; i++) out_kf_idx[i] = -1;
    if (n_kf == 0 || n_f == 0) return GFO_OK;
    // node intersection: merge of two ascending id lists (the lower_bound walk of :292-380)
    std::vector<int2> pairs;
    int max_f_items = 0;
    for (int a = 0, b = 0; a < kf_fv->n_nodes && b < f_fv->n_nodes;) {
        if (kf_fv->node_ids[a] == f_fv->node_ids[b]) {
            pairs.push_back(make_int2(a, b));
            const int nf = f_fv->node_start[b + 1] - f_fv->node_start[b];
            max_f_items = nf > max_f_items ? nf : max_f_items;
            a++; b++;
        } else if (kf_fv->node_ids[a] < f_fv->node_ids[b]) a++;
        else b++;
    }
    if (pairs.empty()) return GFO_OK;
    if (max_f_items >= (1 << 20)) {
        c->err = "gfo_search_by_bow: node with more than 2^20 frame keypoints";
        return GFO_ERR_INVALID;
    }
    BTRY(c, hipSetDevice(c->device));
    const int nk_items = kf_fv->node_start[kf_fv->n_nodes], nf_items = f_fv->node_start[f_fv->n_nodes];
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    const size_t o_kd = take(32 * (size_t)n_kf), o_ka = take(4 * (size_t)n_kf), o_kv = take(n_kf),
                 o_ks = take(4 * (size_t)(kf_fv->n_nodes + 1)), o_ki = take(4 * (size_t)(nk_items > 0 ? nk_items : 1)),
                 o_fd = take(32 * (size_t)n_f), o_fa = take(4 * (size_t)n_f), o_fs = take(4 * (size_t)(f_fv->n_nodes + 1)),
                 o_fi = take(4 * (size_t)(nf_items > 0 ? nf_items : 1)), o_pr = take(sizeof(int2) * pairs.size()),
                 o_rb = take(4 * (size_t)n_f), o_out = take(4 * (size_t)n_f), o_cnt = take(16);
    const bool budget = max_matches > 0;   // the budget's bookkeeping behind everything else: not part of the one copy in, never copied out
    const size_t o_ord = take(budget ? 4 * (size_t)n_f : 0), o_nof = take(budget ? 4 * (size_t)n_f : 0), o_nacc = take(budget ? 4 * pairs.size() : 0);
    const size_t in_bytes = o_cnt + 16;
    if (off > c->scratch_bytes) {
        (void)hipStreamSynchronize(c->stream);
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_bytes = 0;
        BTRY(c, hipMalloc(&c->d_scratch, off));
        c->scratch_bytes = off;
    }
    uint8_t* S = (uint8_t*)c->d_scratch;
    hipStream_t st = c->stream;
    // ten inputs and the three cleared outputs in ONE copy through the pinned mirror of the scratch layout (GfoXfer)
    GfoXfer x(c);
    if (int rc = x.in(in_bytes)) return rc;
    x.put(o_kd, kf_desc, 32 * (size_t)n_kf);
    if (kf_angle) x.put(o_ka, kf_angle, 4 * (size_t)n_kf);
    x.put(o_kv, kf_mp_valid, n_kf);
    x.put(o_ks, kf_fv->node_start, 4 * (size_t)(kf_fv->n_nodes + 1));
    if (nk_items) x.put(o_ki, kf_fv->items, 4 * (size_t)nk_items);
    x.put(o_fd, f_desc, 32 * (size_t)n_f);
    if (f_angle) x.put(o_fa, f_angle, 4 * (size_t)n_f);
    x.put(o_fs, f_fv->node_start, 4 * (size_t)(f_fv->n_nodes + 1));
    if (nf_items) x.put(o_fi, f_fv->items, 4 * (size_t)nf_items);
    x.put(o_pr, pairs.data(), sizeof(int2) * pairs.size());
    memset(x.H + o_rb, 0xFF, o_cnt - o_rb);   // rot_bin and out = -1
    if (f_valid) {
        int* o = reinterpret_cast<int*>(x.H + o_out);
        for (int i = 0; i < n_f; i++) if (!f_valid[i]) o[i] = BOW_UNAVAILABLE;
    }
    memset(x.H + o_cnt, 0, 16);
    BTRY(c, x.up(S, o_cnt + 16, st));
    BowArgs a{};
    a.kf_desc = S + o_kd; a.kf_angle = (const float*)(S + o_ka); a.kf_valid = S + o_kv;
    a.kf_start = (const int*)(S + o_ks); a.kf_items = (const unsigned*)(S + o_ki);
    a.f_desc = S + o_fd; a.f_angle = (const float*)(S + o_fa);
    a.f_start = (const int*)(S + o_fs); a.f_items = (const unsigned*)(S + o_fi);
    a.pairs = (const int2*)(S + o_pr); a.npairs = (int)pairs.size();
    a.n_f = n_f; a.nn_ratio = nn_ratio; a.check_ori = check_orientation ? 1 : 0;
    a.out = (int*)(S + o_out); a.rot_bin = (int*)(S + o_rb); a.counters = (int*)(S + o_cnt);
    a.max_matches = budget ? max_matches : 0;
    a.th_low = strict ? TH_LOW - 1 : TH_LOW;
    a.ord = (int*)(S + o_ord); a.node_of = (int*)(S + o_nof); a.node_acc = (int*)(S + o_nacc);
    if (int rc = x.out(o_cnt + 16 - o_out)) return rc;
    // with the rotation check its kernel is the last one and writes the answer into the pinned block itself; without it, one copy back
    // (out and the counters are neighbours in the scratch)
    const bool direct = a.check_ori && gfo_matcher_host_writes();
    if (direct) { a.h_out = (int*)x.HO; a.h_count = (int*)(x.HO + (o_cnt - o_out)); }
    gfo_prof_begin(c, ST_BOW);
    GFO_LAUNCH(c, k_bow_match, dim3((a.npairs + 3) / 4), dim3(256), 0, st, a);
    if (budget) GFO_LAUNCH(c, k_bow_budget, dim3(1), dim3(256), 0, st, a);
    if (a.check_ori) GFO_LAUNCH(c, k_bow_rotation, dim3(1), dim3(256), 0, st, a);
    gfo_prof_end(c);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    BTRY(c, hipGetLastError());
    if (!direct) BTRY(c, x.down(S + o_out, o_cnt + 16 - o_out, st));
    BTRY(c, hipStreamSynchronize(st));
    memcpy(out_kf_idx, x.HO, 4 * (size_t)n_f);
    if (f_valid) for (int i = 0; i < n_f; i++) if (out_kf_idx[i] == BOW_UNAVAILABLE) out_kf_idx[i] = -1;
    *nmatches = reinterpret_cast<const int*>(x.HO + (o_cnt - o_out))[0];
    return GFO_OK;
}

extern "C" int gfo_search_by_bow_budget(gfo_ctx* c, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_mp_valid,
                                        int n_kf, const gfo_feature_vector* kf_fv, const uint8_t* f_desc, const float* f_angle,
                                        int n_f, const gfo_feature_vector* f_fv, float nn_ratio, int check_orientation, int max_matches,
                                        int32_t* out_kf_idx, int* nmatches)
{
    return bow_search(c, kf_desc, kf_angle, kf_mp_valid, n_kf, kf_fv, f_desc, f_angle, nullptr, n_f, f_fv, nn_ratio, check_orientation, max_matches,
                      false, out_kf_idx, nmatches);
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (ORBmatcher.cc:635-768): the same walk over the
// common nodes with the first keyframe in the (KeyFrame, Frame) overload's keyframe role and the second in its frame role -- a keypoint
// of the second keyframe without a usable map point starts out "taken" (:690-696), the distance test is strict (:713).  Every keypoint of
// either side is matched at most once, so the second side's answer (which keypoint of pKF1 took me) turned around is vpMatches12.
extern "C" int gfo_search_by_bow_keyframes(gfo_ctx* c, const uint8_t* desc1, const float* angle1, const uint8_t* mp_valid1, int n1,
                                           const gfo_feature_vector* fv1, const uint8_t* desc2, const float* angle2, const uint8_t* mp_valid2,
                                           int n2, const gfo_feature_vector* fv2, float nn_ratio, int check_orientation, int32_t* out_idx2,
                                           int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!out_idx2 || n1 < 0 || n2 < 0 || (n2 > 0 && !mp_valid2)) {
        c->err = "gfo_search_by_bow_keyframes: bad argument";
        return GFO_ERR_INVALID;
    }
    std::vector<int32_t> taker((size_t)(n2 > 0 ? n2 : 1));
    const int rc = bow_search(c, desc1, angle1, mp_valid1, n1, fv1, desc2, angle2, mp_valid2, n2, fv2, nn_ratio, check_orientation, 0, true,
                              taker.data(), nmatches);
    if (rc != GFO_OK) return rc;
    for (int i = 0; i < n1; i++) out_idx2[i] = -1;
    for (int j = 0; j < n2; j++)
        if (taker[j] >= 0 && taker[j] < n1) out_idx2[taker[j]] = j;
    return GFO_OK;
}

extern "C" int gfo_search_for_triangulation(gfo_ctx* c, const gfo_keypoint* kp1, const uint8_t* desc1, const uint8_t* has_mp1, const float* u_right1, int n1,
                                            const gfo_feature_vector* fv1, const gfo_keypoint* kp2, const uint8_t* desc2, const uint8_t* has_mp2,
                                            const float* u_right2, int n2, const gfo_feature_vector* fv2, const float* scale_factors2,
                                            const float* level_sigma2_2, int nlevels, const float* f12, float ex, float ey, int only_stereo,
                                            int check_orientation, int32_t* out_idx2, int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!fv1 || !fv2 || !out_idx2 || !nmatches || !f12 || !scale_factors2 || !level_sigma2_2 || nlevels < 1 || nlevels > GFO_MAX_LEVELS || n1 < 0 || n2 < 0 ||
        (n1 > 0 && (!kp1 || !desc1 || !has_mp1)) || (n2 > 0 && (!kp2 || !desc2 || !has_mp2))) {
        c->err = "gfo_search_for_triangulation: bad argument";
        return GFO_ERR_INVALID;
    }
    for (int i = 0; i < n2; i++)
        if (kp2[i].octave < 0 || kp2[i].octave >= nlevels) { c->err = "gfo_search_for_triangulation: keypoint octave outside the level tables"; return GFO_ERR_INVALID; }
    if (check_orientation) {
        for (int i = 0; i < n1; i++)
            if (!(kp1[i].angle >= 0.f && kp1[i].angle <= 360.f)) { c->err = "gfo_search_for_triangulation: keypoint angle outside 0..360"; return GFO_ERR_INVALID; }
        for (int i = 0; i < n2; i++)
            if (!(kp2[i].angle >= 0.f && kp2[i].angle <= 360.f)) { c->err = "gfo_search_for_triangulation: keypoint angle outside 0..360"; return GFO_ERR_INVALID; }
    }
    if (const char* why = bow_check_feature_vector(fv1, n1)) { c->err = std::string("gfo_search_for_triangulation: first feature vector: ") + why; return GFO_ERR_INVALID; }
    if (const char* why = bow_check_feature_vector(fv2, n2)) { c->err = std::string("gfo_search_for_triangulation: second feature vector: ") + why; return GFO_ERR_INVALID; }
    std::vector<int2> pairs;
    for (int a = 0, b = 0; a < fv1->n_nodes && b < fv2->n_nodes;) {
        if (fv1->node_ids[a] == fv2->node_ids[b]) { pairs.push_back(make_int2(a, b)); a++; b++; }
        else if (fv1->node_ids[a] < fv2->node_ids[b]) a++;
        else b++;
    }
    for (const int2& pr : pairs)
        if (fv2->node_start[pr.y + 1] - fv2->node_start[pr.y] >= (1 << 20)) { c->err = "gfo_search_for_triangulation: node with more than 2^20 keypoints"; return GFO_ERR_INVALID; }
    // (nothing above has written to the caller's arrays: a refused call leaves them as they were)
    *nmatches = 0;
    for (int i = 0; i < n1; i++) out_idx2[i] = -1;
    if (n1 == 0 || n2 == 0 || pairs.empty()) return GFO_OK;
    BTRY(c, hipSetDevice(c->device));
    const int it1 = fv1->node_start[fv1->n_nodes], it2 = fv2->node_start[fv2->n_nodes];
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    const size_t o_d1 = take(32 * (size_t)n1), o_k1 = take(sizeof(gfo_keypoint) * (size_t)n1), o_f1 = take(n1), o_s1 = take(4 * (size_t)(fv1->n_nodes + 1)),
                 o_i1 = take(4 * (size_t)(it1 > 0 ? it1 : 1)), o_d2 = take(32 * (size_t)n2), o_k2 = take(sizeof(gfo_keypoint) * (size_t)n2), o_f2 = take(n2),
                 o_s2 = take(4 * (size_t)(fv2->n_nodes + 1)), o_i2 = take(4 * (size_t)(it2 > 0 ? it2 : 1)), o_pr = take(sizeof(int2) * pairs.size()),
                 o_sc = take(4 * GFO_MAX_LEVELS), o_sg = take(4 * GFO_MAX_LEVELS), o_rb = take(4 * (size_t)n1), o_out = take(4 * (size_t)n1), o_cnt = take(16);
    if (off > c->scratch_bytes) {
        (void)hipStreamSynchronize(c->stream);
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_bytes = 0;
        BTRY(c, hipMalloc(&c->d_scratch, off + off / 2));
        c->scratch_bytes = off + off / 2;
    }
    uint8_t* S = (uint8_t*)c->d_scratch;
    hipStream_t st = c->stream;
    GfoXfer x(c);
    if (int rc = x.in(o_cnt + 16)) return rc;
    x.put(o_d1, desc1, 32 * (size_t)n1); x.put(o_k1, kp1, sizeof(gfo_keypoint) * (size_t)n1);
    x.put(o_d2, desc2, 32 * (size_t)n2); x.put(o_k2, kp2, sizeof(gfo_keypoint) * (size_t)n2);
    for (int i = 0; i < n1; i++) x.H[o_f1 + i] = (uint8_t)((has_mp1[i] ? 1 : 0) | (u_right1 && u_right1[i] >= 0 ? 2 : 0));   // bStereo1, :817
    for (int i = 0; i < n2; i++) x.H[o_f2 + i] = (uint8_t)((has_mp2[i] ? 1 : 0) | (u_right2 && u_right2[i] >= 0 ? 2 : 0));
    x.put(o_s1, fv1->node_start, 4 * (size_t)(fv1->n_nodes + 1));
    if (it1) x.put(o_i1, fv1->items, 4 * (size_t)it1);
    x.put(o_s2, fv2->node_start, 4 * (size_t)(fv2->n_nodes + 1));
    if (it2) x.put(o_i2, fv2->items, 4 * (size_t)it2);
    x.put(o_pr, pairs.data(), sizeof(int2) * pairs.size());
    {
        float tab[GFO_MAX_LEVELS] = {0};
        for (int l = 0; l < nlevels; l++) tab[l] = scale_factors2[l];
        x.put(o_sc, tab, sizeof tab);
        for (int l = 0; l < nlevels; l++) tab[l] = level_sigma2_2[l];
        x.put(o_sg, tab, sizeof tab);
    }
    memset(x.H + o_rb, 0xFF, o_cnt - o_rb);   // rot_bin and out = -1
    memset(x.H + o_cnt, 0, 16);
    BTRY(c, x.up(S, o_cnt + 16, st));
    TriArgs a{};
    a.desc1 = S + o_d1; a.kp1 = (const gfo_keypoint*)(S + o_k1); a.flag1 = S + o_f1; a.start1 = (const int*)(S + o_s1); a.items1 = (const unsigned*)(S + o_i1);
    a.desc2 = S + o_d2; a.kp2 = (const gfo_keypoint*)(S + o_k2); a.flag2 = S + o_f2; a.start2 = (const int*)(S + o_s2); a.items2 = (const unsigned*)(S + o_i2);
    a.pairs = (const int2*)(S + o_pr); a.npairs = (int)pairs.size();
    for (int i = 0; i < 9; i++) a.f12[i] = f12[i];
    a.ex = ex; a.ey = ey;
    a.scale2 = (const float*)(S + o_sc); a.sigma2 = (const float*)(S + o_sg);
    a.only_stereo = only_stereo ? 1 : 0; a.check_ori = check_orientation ? 1 : 0; a.n1 = n1;
    a.out = (int*)(S + o_out); a.rot_bin = (int*)(S + o_rb); a.counters = (int*)(S + o_cnt);
    if (int rc = x.out(o_cnt + 16 - o_out)) return rc;
    gfo_prof_begin(c, ST_BOW);
    GFO_LAUNCH(c, k_bow_triangulate, dim3(a.npairs, TRI_SPLIT), dim3(256), 0, st, a);
    if (a.check_ori) {   // the rotation histogram over the FIRST keyframe's keypoints (:876, :913-917): the same kernel, its arrays of length n1
        BowArgs r{};
        r.n_f = n1; r.out = a.out; r.rot_bin = a.rot_bin; r.counters = a.counters; r.check_ori = 1;
        GFO_LAUNCH(c, k_bow_rotation, dim3(1), dim3(256), 0, st, r);
    }
    gfo_prof_end(c);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    BTRY(c, hipGetLastError());
    BTRY(c, x.down(S + o_out, o_cnt + 16 - o_out, st));
    BTRY(c, hipStreamSynchronize(st));
    memcpy(out_idx2, x.HO, 4 * (size_t)n1);
    *nmatches = reinterpret_cast<const int*>(x.HO + (o_cnt - o_out))[0];
    return GFO_OK;
}

// ---------------------------------------------------------------------------------------------
// DBoW2 TemplatedVocabulary<FORB>::transform -- the per-descriptor descent of the vocabulary tree
// (TemplatedVocabulary.h:1231-1272).  One wavefront per descriptor: the children of the current node are
// compared 64 at a time, the wave keeps the FIRST minimum (min of dist << 16 | child rank), and walks down
// until it reaches a leaf.  Depth is ~6, so the kernel is a chain of dependent 32-byte reads: L2/MALL-resident
// for any vocabulary that fits the 256 MiB Infinity Cache (the ORB vocabulary is ~35 MB).
// ---------------------------------------------------------------------------------------------
struct VocDev {
    const int* first_child;
    const int* n_children;
    const uint8_t* descriptors;
    const int* word_id;
    const float* weight;
    const double* weight64;
    int n_nodes, depth;
};

__global__ __launch_bounds__(256) void k_bow_transform(VocDev v, const uint8_t* __restrict__ desc, int n, int levelsup,
                                                       int* __restrict__ word_id, float* __restrict__ weight,
                                                       int* __restrict__ node_id, double* __restrict__ weight64)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform: keep it scalar
    const int i = blockIdx.x * 4 + wave;
    if (i >= n) return;
    const uint4* d = reinterpret_cast<const uint4*>(desc + (long long)i * 32);
    const uint4 a0 = d[0], a1 = d[1];
    const int nid_level = v.depth - levelsup;
    int nid = 0, final_id = 0, level = 0;
    for (int guard = 0; guard < 64; guard++) {  // the tree is at most v.depth deep; the guard bounds a malformed one
        const int nc = v.n_children[final_id];
        if (nc <= 0) break;
        ++level;
        const int fc = v.first_child[final_id];
        unsigned best = 0xFFFFFFFFu;
        for (int c = lane; c < nc; c += 64) {
            const uint4* q = reinterpret_cast<const uint4*>(v.descriptors + (long long)(fc + c) * 32);
            const uint4 b0 = q[0], b1 = q[1];
            const unsigned dist = __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                                  __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
            best = min(best, (dist << 16) | (unsigned)c);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
        final_id = fc + (int)(best & 0xFFFF);
        if (level == nid_level) nid = final_id;
    }
    if (lane == 0) {
        word_id[i] = v.word_id[final_id];
        if (weight) weight[i] = v.weight[final_id];
        if (weight64) weight64[i] = v.weight64[final_id];
        node_id[i] = nid;
    }
}

// ---------------------------------------------------------------------------------------------
// The fold of TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1140-1212): one
// workgroup turns the per-feature (word, weight, node) triples into both maps, flattened in std::map order.
// Two sorts of (key << 32 | feature index) in LDS -- by node for the FeatureVector, by word for the BowVector --
// give every segment its members in feature order, which is the order the reference inserts and ADDS them in, so
// the double sums come out bit for bit (addWeight: v1, then += v2, += v3 ...); the normalisation sum runs over the
// words in ascending order, serially, as BowVector::normalize does.
// ---------------------------------------------------------------------------------------------
struct BowFold {
    const int* word; const double* wt; const int* node; int n;
    int weighting, norm;
    unsigned* bow_words; double* bow_values; unsigned* fv_nodes; int* fv_start; unsigned* fv_items; int* counts;   // counts[0] words, [1] fv nodes
    unsigned long long* gkey; int* gflag;   // k_bow_fold<true>: the sort keys and marks in device memory (more than 8192 descriptors)
    // k_bow_fold<false>: the answer ALSO goes straight into the caller-side pinned block (null: copied back after the kernel)
    unsigned* h_bow_words; double* h_bow_values; unsigned* h_fv_nodes; int* h_fv_start; unsigned* h_fv_items; int* h_counts;
};

// compare-exchange of the bitonic network for the element at index i: keeps the smaller key when the element is the lower one of an
// ascending pair (or the upper one of a descending pair)
__device__ __forceinline__ unsigned long long bow_cx(unsigned long long mine, unsigned long long other, int i, int j, int k2)
{
    const bool lower = (i & j) == 0, up = (i & k2) == 0;
    const bool take_min = lower == up;
    const bool other_smaller = other < mine;
    return (take_min == other_smaller) ? other : mine;
}

// Bitonic sort of p2 keys (ascending) by 1024 threads.  Partners closer than a wavefront (j < 64) are exchanged through the lanes
// with the keys in registers (slot t of a thread is element t * 1024 + tid, so i ^ j is lane ^ j of the same slot); only the steps
// with j >= 64 go through the array, one barrier each.  REG = false (keys in device memory, any p2): every step through the array.
// p2 = 2048: 21 barriers instead of 66.
template <bool REG>
__device__ void bow_bitonic(unsigned long long* key, int p2, int tid)
{
    if (REG && p2 >= 64) {
        const int slots = p2 >> 10 ? p2 >> 10 : 1;   // p2 <= 8192: at most 8; p2 < 1024: threads tid >= p2 idle
        unsigned long long r[8];
        const bool mine = tid < p2;
        auto load = [&]() {
#pragma unroll
            for (int t = 0; t < 8; t++) if (t < slots && mine) r[t] = key[t * 1024 + tid];
        };
        auto store = [&]() {
#pragma unroll
            for (int t = 0; t < 8; t++) if (t < slots && mine) key[t * 1024 + tid] = r[t];
        };
        auto lanes = [&](int k2, int j_from) {       // the steps j_from, j_from / 2, ..., 1 of stage k2 between lanes
            for (int j = j_from; j > 0; j >>= 1) {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    if (t < slots) {
                        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)r[t], j), hi = (unsigned)__shfl_xor((int)(unsigned)(r[t] >> 32), j);
                        r[t] = bow_cx(r[t], ((unsigned long long)hi << 32) | lo, t * 1024 + tid, j, k2);
                    }
                }
            }
        };
        load();
        for (int k2 = 2; k2 <= 64; k2 <<= 1) lanes(k2, k2 >> 1);
        for (int k2 = 128; k2 <= p2; k2 <<= 1) {
            store();
            __syncthreads();
            for (int j = k2 >> 1; j >= 64; j >>= 1) {
                for (int i = tid; i < p2; i += 1024) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long a = key[i], b = key[ixj];
                        const bool up = (i & k2) == 0;
                        if (up ? a > b : a < b) { key[i] = b; key[ixj] = a; }
                    }
                }
                __syncthreads();
            }
            load();
            lanes(k2, 32);
        }
        store();
        __syncthreads();
        return;
    }
    for (int k2 = 2; k2 <= p2; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < p2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = key[i], b = key[ixj];
                    const bool up = (i & k2) == 0;
                    if (up ? a > b : a < b) { key[i] = b; key[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

// exclusive scan of flag[0..n) (ints in LDS) in place; returns the total.  1024 threads: a run of the array per thread, the runs'
// sums scanned in the DPP network (k_wave.inc, two barriers).
__device__ int bow_scan(int* v, int n, int* part, int tid)
{
    const int chunk = (n + 1023) / 1024;
    const int b = min(tid * chunk, n), e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; i++) s += v[i];
    int total;
    int run = st_block_incl_scan(s, part, &total) - s;
    for (int i = b; i < e; i++) { const int x = v[i]; v[i] = run; run += x; }
    __syncthreads();
    return total;
}

// GMEM = false: keys and marks in LDS (p2 <= 8192: 96 KB), the form every ordinary frame takes.  GMEM = true: the same code on
// device memory -- one workgroup bitonic-sorting through L2 is slow (milliseconds at 50 000 descriptors) but it lifts the limit:
// no frame is refused or folded elsewhere (the reference itself has no limit, TemplatedVocabulary.h:1140-1212).
// TWO workgroups: block 0 folds the FeatureVector, block 1 the BowVector -- the two maps share nothing but the inputs, and a fold is
// a chain of ~90 workgroup barriers that one CU walks alone (round 5: 162 us for both in one workgroup on the EuRoC frame).
template <bool GMEM>
__global__ __launch_bounds__(1024) void k_bow_fold(BowFold a, int p2)
{
    extern __shared__ unsigned long long lds_dyn[];          // p2 keys + p2 ints (GMEM = false)
    const bool bow_side = blockIdx.x == 1;
    unsigned long long* lds_key = GMEM ? a.gkey + (bow_side ? p2 : 0) : lds_dyn;
    int* flag = GMEM ? a.gflag + (bow_side ? p2 : 0) : reinterpret_cast<int*>(lds_dyn + p2);
    __shared__ int part[1024];
    __shared__ int s_nvalid;
    const int tid = threadIdx.x, n = a.n;
    // ---- sort the kept features by (node, feature) / (word, feature): every segment then lists its members in feature order ----
    const int* id = bow_side ? a.word : a.node;
    for (int i = tid; i < p2; i += 1024)
        lds_key[i] = i < n && a.wt[i] > 0 ? ((unsigned long long)(unsigned)id[i] << 32) | (unsigned)i : ~0ull;   // w > 0: not stopped (:1169)
    if (tid == 0) s_nvalid = 0;
    __syncthreads();
    bow_bitonic<!GMEM>(lds_key, p2, tid);
    int cnt = 0;
    for (int i = tid; i < n; i += 1024) {
        const bool ok = lds_key[i] != ~0ull;
        cnt += ok;
        flag[i] = ok && (i == 0 || (lds_key[i] >> 32) != (lds_key[i - 1] >> 32)) ? 1 : 0;
    }
    if (cnt) atomicAdd(&s_nvalid, cnt);
    __syncthreads();
    const int nvalid = s_nvalid;
    // the scan is exclusive: keep the segment-start marks aside (bit 31 of the key's index half is free: n <= 2^20)
    for (int i = tid; i < n; i += 1024)
        if (flag[i]) lds_key[i] |= 0x80000000ull;
    __syncthreads();
    const int nseg = bow_scan(flag, n, part, tid);
    if (!bow_side) {
        // ---- FeatureVector ----
        const bool host = !GMEM && a.h_fv_items;
        for (int i = tid; i < nvalid; i += 1024) {
            const unsigned long long k = lds_key[i];
            a.fv_items[i] = (unsigned)(k & 0x7FFFFFFFu);
            if (host) a.h_fv_items[i] = (unsigned)(k & 0x7FFFFFFFu);
            if (k & 0x80000000ull) {
                a.fv_nodes[flag[i]] = (unsigned)(k >> 32);
                a.fv_start[flag[i]] = i;
                if (host) { a.h_fv_nodes[flag[i]] = (unsigned)(k >> 32); a.h_fv_start[flag[i]] = i; }
            }
        }
        if (tid == 0) {
            a.fv_start[nseg] = nvalid; a.counts[1] = nseg;
            if (host) { a.h_fv_start[nseg] = nvalid; a.h_counts[1] = nseg; }
        }
        return;
    }
    // ---- BowVector: one thread per word walks its features in order ----
    const int nwords = nseg;
    const bool tf = a.weighting == 0 || a.weighting == 1;
    auto word_value = [&](int i, unsigned long long k) {
        const unsigned w = (unsigned)(k >> 32);
        double v = a.wt[(int)(k & 0x7FFFFFFFu)];              // insert(id, v)
        if (tf)                                                // addWeight: += in feature order
            for (int j = i + 1; j < nvalid && (unsigned)(lds_key[j] >> 32) == w; j++) v += a.wt[(int)(lds_key[j] & 0x7FFFFFFFu)];
        return v;
    };
    // the values of the words, in word order, where the normalisation below reads them: device memory when the keys are there,
    // otherwise the key array itself (8 bytes a slot, nwords <= nvalid) once every thread has taken what it needs out of it
    double* vals = GMEM ? a.bow_values : reinterpret_cast<double*>(lds_key);
    if (GMEM) {
        for (int i = tid; i < nvalid; i += 1024) {
            const unsigned long long k = lds_key[i];
            if (!(k & 0x80000000ull)) continue;
            a.bow_words[flag[i]] = (unsigned)(k >> 32);
            vals[flag[i]] = word_value(i, k);
        }
        __syncthreads();
        __threadfence_block();   // the stores above are read back below by other threads of this workgroup
        __syncthreads();
    } else {
        double v[8];             // p2 <= 8192: at most eight slots a thread
        int slot[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int i = tid + t * 1024;
            slot[t] = -1;
            v[t] = 0.0;
            if (i < nvalid) {
                const unsigned long long k = lds_key[i];
                if (k & 0x80000000ull) {
                    slot[t] = flag[i];
                    a.bow_words[slot[t]] = (unsigned)(k >> 32);
                    if (a.h_bow_words) a.h_bow_words[slot[t]] = (unsigned)(k >> 32);
                    v[t] = word_value(i, k);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; t++)
            if (slot[t] >= 0) vals[slot[t]] = v[t];
        __syncthreads();
    }
    double scale = 0.0;   // 0: values leave as they are
    bool divide = false;
    if (tf && nwords > 0 && a.norm == 0) {                     // "unnecessary when normalizing" (:1177-1183)
        scale = (double)nwords;
        divide = true;
    }
    if (a.norm != 0) {
        __shared__ double s_norm;
        if (tid == 0) {                                        // BowVector::normalize: ONE running sum over the map order
            double nrm = 0.0;
            int k = 0;
            if (a.norm == 1) {
                for (; k + 8 <= nwords; k += 8) {              // eight loads in flight, the additions in order
                    const double t0 = vals[k], t1 = vals[k + 1], t2 = vals[k + 2], t3 = vals[k + 3], t4 = vals[k + 4], t5 = vals[k + 5],
                                 t6 = vals[k + 6], t7 = vals[k + 7];
                    nrm += fabs(t0); nrm += fabs(t1); nrm += fabs(t2); nrm += fabs(t3);
                    nrm += fabs(t4); nrm += fabs(t5); nrm += fabs(t6); nrm += fabs(t7);
                }
                for (; k < nwords; k++) nrm += fabs(vals[k]);
            } else {
                for (; k + 8 <= nwords; k += 8) {
                    const double t0 = vals[k], t1 = vals[k + 1], t2 = vals[k + 2], t3 = vals[k + 3], t4 = vals[k + 4], t5 = vals[k + 5],
                                 t6 = vals[k + 6], t7 = vals[k + 7];
                    nrm += t0 * t0; nrm += t1 * t1; nrm += t2 * t2; nrm += t3 * t3;
                    nrm += t4 * t4; nrm += t5 * t5; nrm += t6 * t6; nrm += t7 * t7;
                }
                for (; k < nwords; k++) nrm += vals[k] * vals[k];
                nrm = sqrt(nrm);
            }
            s_norm = nrm;
        }
        __syncthreads();
        scale = s_norm;
        divide = scale > 0.0;
    }
    if (!GMEM || divide)
        for (int k = tid; k < nwords; k += 1024) {
            const double v = divide ? vals[k] / scale : vals[k];
            a.bow_values[k] = v;
            if (!GMEM && a.h_bow_values) a.h_bow_values[k] = v;
        }
    if (tid == 0) {
        a.counts[0] = nwords;
        if (!GMEM && a.h_counts) a.h_counts[0] = nwords;
    }
}

extern "C" int gfo_vocabulary_upload(gfo_ctx* c, const gfo_vocabulary* voc)
{
    if (!c) return GFO_ERR_INVALID;
    if (!voc || voc->n_nodes < 1 || !voc->first_child || !voc->n_children || !voc->descriptors || !voc->word_id || !voc->weight) {
        c->err = "gfo_vocabulary_upload: bad argument";
        return GFO_ERR_INVALID;
    }
    for (int i = 0; i < voc->n_nodes; i++) {
        const int nc = voc->n_children[i], fc = voc->first_child[i];
        if (nc < 0 || nc > 65535 || (nc > 0 && (fc <= i || fc + nc > voc->n_nodes))) {
            c->err = "gfo_vocabulary_upload: children must follow their parent as one contiguous range";
            return GFO_ERR_INVALID;
        }
    }
    BTRY(c, hipSetDevice(c->device));
    (void)hipStreamSynchronize(c->stream);
    if (c->d_voc) (void)hipFree(c->d_voc);
    c->d_voc = nullptr;
    const size_t nn = (size_t)voc->n_nodes;
    const size_t bytes = nn * (4 + 4 + 32 + 4 + 4 + 8) + 2048;
    BTRY(c, hipMalloc(&c->d_voc, bytes));
    uint8_t* p = (uint8_t*)c->d_voc;
    c->voc_desc_off = 0;
    c->voc_fc_off = (nn * 32 + 255) / 256 * 256;
    c->voc_nc_off = c->voc_fc_off + nn * 4;
    c->voc_wid_off = c->voc_nc_off + nn * 4;
    c->voc_w_off = c->voc_wid_off + nn * 4;
    c->voc_w64_off = (c->voc_w_off + nn * 4 + 255) / 256 * 256;
    BTRY(c, hipMemcpy(p + c->voc_desc_off, voc->descriptors, nn * 32, hipMemcpyHostToDevice));
    BTRY(c, hipMemcpy(p + c->voc_fc_off, voc->first_child, nn * 4, hipMemcpyHostToDevice));
    BTRY(c, hipMemcpy(p + c->voc_nc_off, voc->n_children, nn * 4, hipMemcpyHostToDevice));
    BTRY(c, hipMemcpy(p + c->voc_wid_off, voc->word_id, nn * 4, hipMemcpyHostToDevice));
    BTRY(c, hipMemcpy(p + c->voc_w_off, voc->weight, nn * 4, hipMemcpyHostToDevice));
    {
        std::vector<double> w64(nn);
        for (size_t i = 0; i < nn; i++) w64[i] = voc->weight64 ? voc->weight64[i] : (double)voc->weight[i];
        BTRY(c, hipMemcpy(p + c->voc_w64_off, w64.data(), nn * 8, hipMemcpyHostToDevice));
    }
    c->voc_nodes = voc->n_nodes;
    c->voc_depth = voc->depth;
    return GFO_OK;
}

// where k_bow_transform reads the call's descriptors: the pinned block (device-visible; up to 1 MB) or, after a copy, the scratch
static const uint8_t* bow_desc_source(gfo_ctx* c, const GfoXfer& x, uint8_t* d_scratch, size_t bytes, hipStream_t st)
{
    static const bool from_host = !(getenv("GFO_BOW_DESC_FROM_HOST") && atoi(getenv("GFO_BOW_DESC_FROM_HOST")) == 0);
    if (from_host && bytes <= (1u << 20)) return x.H;
    if (x.up(d_scratch, bytes, st) != hipSuccess) {
        c->err = "gfo_compute_bow: copying the descriptors failed";
        return nullptr;
    }
    return d_scratch;
}

extern "C" int gfo_bow_transform(gfo_ctx* c, const uint8_t* desc, int n, int levelsup, int32_t* word_id, float* weight,
                                 int32_t* node_id)
{
    if (!c) return GFO_ERR_INVALID;
    if (n < 0 || (n > 0 && (!desc || !word_id || !weight || !node_id))) {
        c->err = "gfo_bow_transform: bad argument";
        return GFO_ERR_INVALID;
    }
    if (!c->d_voc) {
        c->err = "gfo_bow_transform: no vocabulary uploaded";
        return GFO_ERR_STATE;
    }
    if (n == 0) return GFO_OK;
    BTRY(c, hipSetDevice(c->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    const size_t o_d = take(32 * (size_t)n), o_w = take(4 * (size_t)n), o_wt = take(4 * (size_t)n), o_n = take(4 * (size_t)n);
    if (off > c->scratch_bytes) {
        (void)hipStreamSynchronize(c->stream);
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_bytes = 0;
        BTRY(c, hipMalloc(&c->d_scratch, off));
        c->scratch_bytes = off;
    }
    uint8_t* S = (uint8_t*)c->d_scratch;
    uint8_t* V = (uint8_t*)c->d_voc;
    VocDev v{(const int*)(V + c->voc_fc_off), (const int*)(V + c->voc_nc_off), V + c->voc_desc_off,
             (const int*)(V + c->voc_wid_off), (const float*)(V + c->voc_w_off), (const double*)(V + c->voc_w64_off), c->voc_nodes, c->voc_depth};
    hipStream_t st = c->stream;
    GfoXfer x(c);
    if (int rc = x.in(32 * (size_t)n)) return rc;
    x.put(0, desc, 32 * (size_t)n);
    // every wavefront reads its descriptor ONCE (32 bytes): the descent reads them from the pinned block itself -- no copy in
    // (GFO_BOW_DESC_FROM_HOST=0: a copy into the scratch first)
    const uint8_t* d_desc = bow_desc_source(c, x, S + o_d, 32 * (size_t)n, st);
    if (!d_desc) return GFO_ERR_DEVICE;
    gfo_prof_begin(c, ST_BOW);
    GFO_LAUNCH(c, k_bow_transform, dim3((n + 3) / 4), dim3(256), 0, st, v, d_desc, n, levelsup, (int*)(S + o_w),
                       (float*)(S + o_wt), (int*)(S + o_n), (double*)nullptr);
    gfo_prof_end(c);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    BTRY(c, hipGetLastError());
    const size_t out_bytes = o_n + 4 * (size_t)n - o_w;   // word, weight, node: neighbours in the scratch, one copy back
    if (int rc = x.out(out_bytes)) return rc;
    BTRY(c, x.down(S + o_w, out_bytes, st));
    BTRY(c, hipStreamSynchronize(st));
    memcpy(word_id, x.HO, 4 * (size_t)n);
    memcpy(weight, x.HO + (o_wt - o_w), 4 * (size_t)n);
    memcpy(node_id, x.HO + (o_n - o_w), 4 * (size_t)n);
    return GFO_OK;
}


extern "C" int gfo_compute_bow(gfo_ctx* c, const uint8_t* desc, int n, int levelsup, const gfo_bow_mode* mode,
                               uint32_t* bow_words, double* bow_values, int* n_words, uint32_t* fv_node_ids,
                               int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes)
{
    if (!c) return GFO_ERR_INVALID;
    if (n < 0 || !mode || !n_words || !n_fv_nodes || !fv_start || (n > 0 && (!desc || !bow_words || !bow_values || !fv_node_ids || !fv_items)) ||
        mode->weighting < 0 || mode->weighting > 3 || mode->norm < 0 || mode->norm > 2) {
        c->err = "gfo_compute_bow: bad argument";
        return GFO_ERR_INVALID;
    }
    if (!c->d_voc) {
        c->err = "gfo_compute_bow: no vocabulary uploaded";
        return GFO_ERR_STATE;
    }
    *n_words = *n_fv_nodes = 0;
    fv_start[0] = 0;
    if (n == 0) return GFO_OK;
    if (n > (1 << 20)) {
        c->err = "gfo_compute_bow: more than 1048576 descriptors in one call";
        return GFO_ERR_CAPACITY;
    }
    BTRY(c, hipSetDevice(c->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; };
    const size_t N = (size_t)n;
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    const bool gmem = p2 > 8192;      // keys + marks beyond 96 KB of LDS: sorted in device memory
    const size_t o_d = take(32 * N), o_w = take(4 * N), o_wt = take(8 * N), o_n = take(4 * N), o_bw = take(4 * N), o_bv = take(8 * N),
                 o_fn = take(4 * N), o_fs = take(4 * (N + 1)), o_fi = take(4 * N), o_ct = take(16), o_gk = take(gmem ? 16 * (size_t)p2 : 0),
                 o_gf = take(gmem ? 8 * (size_t)p2 : 0);   // (two workgroups, one set each)
    if (off > c->scratch_bytes) {
        (void)hipStreamSynchronize(c->stream);
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_bytes = 0;
        BTRY(c, hipMalloc(&c->d_scratch, off + off / 2));
        c->scratch_bytes = off + off / 2;
    }
    uint8_t* S = (uint8_t*)c->d_scratch;
    uint8_t* V = (uint8_t*)c->d_voc;
    VocDev v{(const int*)(V + c->voc_fc_off), (const int*)(V + c->voc_nc_off), V + c->voc_desc_off,
             (const int*)(V + c->voc_wid_off), (const float*)(V + c->voc_w_off), (const double*)(V + c->voc_w64_off), c->voc_nodes, c->voc_depth};
    hipStream_t st = c->stream;
    GfoXfer x(c);
    if (int rc = x.in(32 * N)) return rc;
    x.put(0, desc, 32 * N);
    const uint8_t* d_desc = bow_desc_source(c, x, S + o_d, 32 * N, st);   // (see gfo_bow_transform)
    if (!d_desc) return GFO_ERR_DEVICE;
    gfo_prof_begin(c, ST_BOW);
    GFO_LAUNCH(c, k_bow_transform, dim3((n + 3) / 4), dim3(256), 0, st, v, d_desc, n, levelsup, (int*)(S + o_w), (float*)nullptr,
                       (int*)(S + o_n), (double*)(S + o_wt));
    // the two vectors and the counters are neighbours in the scratch (o_bw .. o_ct); the pinned block has the same layout.  Up to 8192
    // descriptors the fold writes its answer there itself; beyond (keys in device memory) one copy back instead of six.
    const size_t out_bytes = o_ct + 16 - o_bw;
    if (int rc = x.out(out_bytes)) return rc;
    const bool direct = !gmem && gfo_matcher_host_writes();
    auto twin = [&](size_t o) { return direct ? x.HO + (o - o_bw) : (uint8_t*)nullptr; };   // the pinned twin of S + o
    BowFold f{(const int*)(S + o_w), (const double*)(S + o_wt), (const int*)(S + o_n), n, mode->weighting, mode->norm,
              (unsigned*)(S + o_bw), (double*)(S + o_bv), (unsigned*)(S + o_fn), (int*)(S + o_fs), (unsigned*)(S + o_fi), (int*)(S + o_ct),
              gmem ? (unsigned long long*)(S + o_gk) : nullptr, gmem ? (int*)(S + o_gf) : nullptr,
              (unsigned*)twin(o_bw), (double*)twin(o_bv), (unsigned*)twin(o_fn), (int*)twin(o_fs), (unsigned*)twin(o_fi), (int*)twin(o_ct)};
    if (gmem) {
        GFO_LAUNCH(c, k_bow_fold<true>, dim3(2), dim3(1024), 0, st, f, p2);
    } else {
        const size_t lds = (size_t)p2 * 12;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bow_fold<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        GFO_LAUNCH(c, k_bow_fold<false>, dim3(2), dim3(1024), lds, st, f, p2);
    }
    gfo_prof_end(c);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    BTRY(c, hipGetLastError());
    if (!direct) BTRY(c, x.down(S + o_bw, out_bytes, st));
    BTRY(c, hipStreamSynchronize(st));
    int cnt[4];
    memcpy(cnt, x.HO + (o_ct - o_bw), 16);
    // only what the call produced: n_words entries of the BowVector, n_fv_nodes (+1 offsets) and n items of the FeatureVector
    const size_t nw = (size_t)(cnt[0] < 0 ? 0 : cnt[0] > n ? n : cnt[0]), nn = (size_t)(cnt[1] < 0 ? 0 : cnt[1] > n ? n : cnt[1]);
    memcpy(bow_words, x.HO, 4 * nw);
    memcpy(bow_values, x.HO + (o_bv - o_bw), 8 * nw);
    memcpy(fv_node_ids, x.HO + (o_fn - o_bw), 4 * nn);
    memcpy(fv_start, x.HO + (o_fs - o_bw), 4 * (nn + 1));
    memcpy(fv_items, x.HO + (o_fi - o_bw), 4 * N);
    *n_words = cnt[0];
    *n_fv_nodes = cnt[1];
    return GFO_OK;
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_bow(std::vector<const void*>& v)
{
    v.push_back((const void*)k_bow_match); v.push_back((const void*)k_bow_budget); v.push_back((const void*)k_bow_rotation); v.push_back((const void*)k_bow_triangulate); v.push_back((const void*)k_bow_transform);
    v.push_back((const void*)k_bow_fold<false>); v.push_back((const void*)k_bow_fold<true>);
}
