// k_fast.hip -- per-cell FAST-9/16 with non-max suppression and the per-cell threshold
// fallback of ORBextractor::ComputeKeyPointsOctTree (ORBextractor.cc:786-831), restating
// cv::FAST / FAST_t<16> (in-tree mirror: FAST_NEON.cc:91-287).
//
// One wavefront owns one 30-px cell (+6 px overlap, +3 px ring halo): the cell's pixels are
// staged in LDS once (the tile starts one byte left of the cell, so that the scan columns fall on aligned LDS dwords) and
// the wave runs a filter cascade with WAVE-LEVEL COMPACTION between the stages, so the expensive stages run with all 64
// lanes busy:
//   A  every pixel      : compass test -- a 9-arc of the 16-ring holds a pixel of every opposite pair, so
//                         (ring 0 or 8) and (ring 4 or 12) must both be beyond the threshold (5 LDS reads)
//   B  survivors of A   : the score S = max over the sixteen 9-arcs of min |v - ring| (common sign); a pixel
//                         is a corner at threshold t iff S > t, so the exact test costs nothing extra
//   N  corners          : strict 3x3 local maximum of S inside the cell's scan area
//   E  maxima           : append (x, y, S-1) for S > iniThFAST, or S > minThFAST when the cell
//                         produced nothing at iniThFAST (ORBextractor.cc:811-818)
// S does not depend on the threshold (corner(t) <=> S > t, cornerScore = S - 1: FAST_NEON.cc:231,
// Fast_gpu.cu:196-219), and a neighbour can suppress a corner only if its S is >= the corner's,
// so "compare against the score buffer of corners at threshold t" (FAST_NEON.cc:268-285) is
// "strict local maximum of S" -- one suppression pass serves both thresholds (FAST).
// A cell that finds no maximum at iniThFAST repeats the WHOLE cascade at minThFAST -- half the cells of real indoor
// imagery do (the reference's EuRoC frames: k_fast 166 us per 128 images against 140 on the synthetic stream; 198 against
// 156 before the round-3 instruction cuts below, which pay twice in a repeating cell).  Two ways of sparing such a cell work
// were built earlier in round 3, parity-green, and measured slower on BOTH streams (same-box A/B, tools/ab_fast.sh):
//   * both thresholds tested in the one stage-A pass, the minThFAST-only pass bits kept as an LDS bitmap that the second
//     round merely compacts: +8 vector instructions per stage-A iteration for EVERY cell and 1.5 KB more LDS per wave
//     (27 -> 21 waves per CU): synthetic 156 -> 176 us, EuRoC 198 -> 200 us;
//   * second round's stage A dropping the pixels the first round scored (scores taken with the polarity rule at
//     minThFAST are exact above it and were kept): synthetic 156 -> 164 us (the first round's suppression walks the
//     pixels with minTh < S <= iniTh too), EuRoC 198 -> 205 us.
// The premise was wrong: a cell repeats BECAUSE it has next to no pixels that pass at iniThFAST, so there is nothing to
// avoid re-scoring -- its cost is the second round's own work on ~290 weak candidates (25 % of its pixels pass the
// compass test at threshold 7), which no bookkeeping in the first round reduces; what does reduce it is a cheaper cascade.
// Round 3's cuts (674 -> 514 vector instructions per cell, 156 -> 140 us per 128 images): aligned scan columns (4 stage-A
// passes per cell instead of 5), pass bits tested where the packed arithmetic leaves them, the score from the eight odd
// 8-runs (23 packed operations instead of 40), signed differences by one packed multiply-add per ring word, vote masks kept
// scalar, and the lane -> (row, column) maps precomputed per cell by plan() instead of divided out by every wave.
// Queue entries are 16 bit: px | py << 6, the local-maximum flag in bit 15.
// One cell per (single-wave) workgroup -- a cell's slot is free again the moment its wave ends; all levels of all
// images are ONE launch.
//
// Candidates are appended to the level's list with one atomicAdd per cell; their order in
// memory is unspecified.  The order the reference hands to DistributeOctTree (cell-major,
// row-major inside a cell) matters only as a tie-break on equal response, and it is a pure
// function of (x, y) that the quadtree kernel recomputes.
#include "gfo_internal.h"
#include <stdlib.h>

// Each wave owns its LDS region and LDS operations of one wave execute in issue order, so a
// compiler-level fence is all that is needed between phases (no workgroup barrier: waves of a
// workgroup run their cells independently and may exit early).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef short __attribute__((ext_vector_type(2))) s16x2;
__device__ __forceinline__ unsigned pmin(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ unsigned pmax(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// the two halves swapped: written as a vector shuffle (not v_alignbit) so that instruction selection can fold it into the
// op_sel bits of the packed operation that consumes it -- pmin(a, rot16(a)) is ONE v_pk_min_i16 with op_sel, no rotation
__device__ __forceinline__ unsigned rot16(unsigned a)
{
#ifdef GFO_FAST_ALIGNBIT_ROT
    return __builtin_amdgcn_alignbit(a, a, 16);
#else
    const s16x2 v = __builtin_bit_cast(s16x2, a);
    return __builtin_bit_cast(unsigned, __builtin_shufflevector(v, v, 1, 0));
#endif
}

// Score of a pixel = max over the sixteen 9-arcs of min(sign * d) (sign = +1 dark, -1 bright), computed with two
// ring positions (k, k+8) per register in packed i16: the arc minima for k and k+8 come out of one op.  The ring word
// (r[k], r[k+8]) is one v_lshl_or; the signed difference sign * (v - r) one packed multiply-add.
// Survivors of the compass test: score directly.  corner(t) <=> S > t, so the exact 9-contiguous test and the
// score are one computation (the packed arc minima); the polarity to score comes from the compass pixels
// themselves -- a 9-arc holds a pixel of each opposite pair, so a dark arc needs (r0 or r8) and (r4 or r12)
// below v - t, a bright arc the mirror image.  Where both hold, the all-eight-pairs test picks the one polarity that
// can still be a corner (below).
template <int TP>
__device__ __forceinline__ int corner_score(const uint8_t* __restrict__ tl, int tq, unsigned long long& m_run)
{
    // LDS offsets are unsigned immediates: the ring is addressed from `tl`, one byte left of its top-left corner (so that the
    // caller's address is tile + py * TP + px, nothing added), the centre sits RB bytes further
    constexpr int RB = 3 * TP + 3 + GFO_FAST_XOFF;
#ifdef GFO_FAST_RING64
    // Round 6 experiment (profiles/fast_ring64_r06.txt has the A/B): the ring as SEVEN 8-byte reads, one per ring row
    // from column -3 (bytes of columns -3 .. +4 at fixed positions, whatever the pixel's alignment: gfx950 takes an 8-byte LDS read at
    // any byte address), the ring words cut out by one v_perm_b32 each (selector 0x0C = zero byte; sources: {S1 = bytes 0-3, S0 =
    // bytes 4-7}) -- 7 LDS instructions instead of 17, the same 8 packing operations, one more for the centre.
    struct __attribute__((packed)) row8 { unsigned lo, hi; };
    const row8 Rm3 = *reinterpret_cast<const row8*>(tl + RB - 3 * TP - 3), Rm2 = *reinterpret_cast<const row8*>(tl + RB - 2 * TP - 3),
               Rm1 = *reinterpret_cast<const row8*>(tl + RB - TP - 3), R0 = *reinterpret_cast<const row8*>(tl + RB - 3),
               Rp1 = *reinterpret_cast<const row8*>(tl + RB + TP - 3), Rp2 = *reinterpret_cast<const row8*>(tl + RB + 2 * TP - 3),
               Rp3 = *reinterpret_cast<const row8*>(tl + RB + 3 * TP - 3);
    // column c of a row is byte c + 3 of its 8 bytes: lo holds columns -3 .. 0, hi columns +1 .. +4
#define GFO_SEL(b0, b1) ((unsigned)(b0) | (0x0Cu << 8) | ((unsigned)(b1) << 16) | (0x0Cu << 24))
    const unsigned v2 = __builtin_amdgcn_perm(0u, R0.lo, GFO_SEL(3, 3));   // the centre (column 0 = lo byte 3) in both halves
    unsigned Q[8];
    Q[0] = __builtin_amdgcn_perm(Rm3.lo, Rp3.lo, GFO_SEL(3, 4 + 3));     // (0,+3) | (0,-3): byte 3 of both lo words
    Q[1] = __builtin_amdgcn_perm(Rm3.lo, Rp3.hi, GFO_SEL(0, 4 + 2));     // (+1,+3): hi byte 0 | (-1,-3): lo byte 2
    Q[2] = __builtin_amdgcn_perm(Rm2.lo, Rp2.hi, GFO_SEL(1, 4 + 1));     // (+2,+2): hi byte 1 | (-2,-2): lo byte 1
    Q[3] = __builtin_amdgcn_perm(Rm1.lo, Rp1.hi, GFO_SEL(2, 4 + 0));     // (+3,+1): hi byte 2 | (-3,-1): lo byte 0
    Q[4] = __builtin_amdgcn_perm(R0.lo, R0.hi, GFO_SEL(2, 4 + 0));       // (+3, 0): hi byte 2 | (-3, 0): lo byte 0
    Q[5] = __builtin_amdgcn_perm(Rp1.lo, Rm1.hi, GFO_SEL(2, 4 + 0));     // (+3,-1): hi byte 2 | (-3,+1): lo byte 0
    Q[6] = __builtin_amdgcn_perm(Rp2.lo, Rm2.hi, GFO_SEL(1, 4 + 1));     // (+2,-2): hi byte 1 | (-2,+2): lo byte 1
    Q[7] = __builtin_amdgcn_perm(Rp3.lo, Rm3.hi, GFO_SEL(0, 4 + 2));     // (+1,-3): hi byte 0 | (-1,+3): lo byte 2
#undef GFO_SEL
#else
    const unsigned v = tl[RB];
    const unsigned v2 = v | (v << 16);
    // ring words Q[k] = (r[k], r[k+8]).  (ds_read_u8_d16 / _d16_hi would deliver the pair packed, but with SRAM ECC on -- as
    // on every MI300 / MI355 -- a d16 load clears the other half of its register instead of preserving it: tried in round 3,
    // twice the candidates of the oracle.  One v_lshl_or per word it is.)
    const unsigned r[16] = {tl[RB + 3 * TP], tl[RB + 3 * TP + 1], tl[RB + 2 * TP + 2], tl[RB + TP + 3], tl[RB + 3], tl[RB - TP + 3], tl[RB - 2 * TP + 2], tl[RB - 3 * TP + 1],
                            tl[RB - 3 * TP], tl[RB - 3 * TP - 1], tl[RB - 2 * TP - 2], tl[RB - TP - 3], tl[RB - 3], tl[RB + TP - 3], tl[RB + 2 * TP - 2], tl[RB + 3 * TP - 1]};
    unsigned Q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) Q[k] = r[k] | (r[k + 8] << 16);
#endif
    // compass, on the ring words themselves (no differences yet -- their sign is not known): both halves of A / B are alike,
    //   both pairs (0,8), (4,12) hold a pixel darker than v - t    <=>  max(min(r0, r8), min(r4, r12)) < v - t
    //   both pairs hold a brighter one                              <=>  min(max(r0, r8), max(r4, r12)) > v + t
    // and words with equal halves compare like their halves (x * 0x10001 is monotone, no overflow for 9-bit values)
    const unsigned tq2 = (unsigned)tq * 0x10001u;
    const unsigned A = pmax(pmin(Q[0], rot16(Q[0])), pmin(Q[4], rot16(Q[4])));
    const unsigned B = pmin(pmax(Q[0], rot16(Q[0])), pmax(Q[4], rot16(Q[4])));
    const int vmt = (int)(v2 - tq2), vpt = (int)(v2 + tq2);
    const bool dark = (int)A < vmt, bright = (int)B > vpt;
    // ONE polarity is scored per pixel.  A dark and a bright 9-arc cannot coexist (18 > 16 ring pixels), so a pixel
    // whose compass pixels allow both (saddles, edge crossings: ~5 % of the survivors, i.e. nearly every wave holds
    // one) only needs the polarity that can still be a corner: a dark 9-arc covers at least one member of EVERY
    // opposite pair (k, k+8), so "all eight pairs hold a pixel darker than v - t" is necessary for it -- and when both
    // polarities pass that test no pair has two members of one sign, so neither has a 9-arc and either choice scores
    // <= t.  12 packed ops for the whole wave instead of a second scoring pass.
    // (lane masks in scalar registers from here on: as per-lane bools the compiler carried them through the branch as 0 / 1
    //  integers and paid half a dozen vector selects and compares to turn them back into masks)
    const unsigned long long m_dark = __builtin_amdgcn_ballot_w64(dark), m_bright = __builtin_amdgcn_ballot_w64(bright);
    unsigned long long m_neg = ~m_dark;
    if ((m_dark & m_bright) != 0) {
        // (a tree, not a chain: dependent packed operations back to back cost a wait state each)
        const unsigned e1 = pmin(Q[1], rot16(Q[1])), e2 = pmin(Q[2], rot16(Q[2])), e3 = pmin(Q[3], rot16(Q[3]));
        const unsigned e5 = pmin(Q[5], rot16(Q[5])), e6 = pmin(Q[6], rot16(Q[6])), e7 = pmin(Q[7], rot16(Q[7]));
        const unsigned m8 = pmax(pmax(pmax(e1, e2), pmax(e3, e5)), pmax(pmax(e6, e7), A));   // A: pairs 0 and 4
        const unsigned long long m_all8 = __builtin_amdgcn_ballot_w64((int)m8 < vmt);
        m_neg |= m_bright & ~m_all8;       // dark && bright: bright polarity unless all eight pairs can still be dark
    }
    const bool neg = __builtin_amdgcn_inverse_ballot_w64(m_neg);
    m_run = m_dark | m_bright;   // lanes whose compass pixels allow an arc at all: the others' result is meaningless
    int best = 0;
    {
        // signed differences straight from the ring words: p[k] = sg * (v - r[k]) = (-sg) * r[k] + sg * v is ONE packed
        // multiply-add per ring word (round 2 formed v - r and multiplied by the sign: two)
        const s16x2 nsg = __builtin_bit_cast(s16x2, neg ? 0x00010001u : 0xFFFFFFFFu);
        const s16x2 sv = __builtin_bit_cast(s16x2, v2) * __builtin_bit_cast(s16x2, neg ? 0xFFFFFFFFu : 0x00010001u);
        unsigned P[8];
#pragma unroll
        for (int k = 0; k < 8; k++) P[k] = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, Q[k]) * nsg + sv);   // (p[k], p[k+8])
        // S = max over the sixteen 9-arcs of their minimum.  The arcs starting at s-1 and at s share the eight entries
        // s..s+7 (m8(s)), and max(min(p[s-1], m8), min(m8, p[s+8])) = min(m8, max(p[s-1], p[s+8])): only the EIGHT runs with odd
        // start are needed -- four packed registers (starts s and s+8 in the two halves), built by doubling: 4 + 4 + 4 packed
        // minima, then 3 ops per register (round 2 built all sixteen 4-runs and three ops per arc pair: 40 against 23).
        // rot16() of an operand folds into the op_sel bits of the consuming instruction.
        const unsigned a1 = pmin(P[1], P[2]), a3 = pmin(P[3], P[4]), a5 = pmin(P[5], P[6]), a7 = pmin(P[7], rot16(P[0]));   // 2-runs from 1, 3, 5, 7
        const unsigned b1 = pmin(a1, a3), b3 = pmin(a3, a5), b5 = pmin(a5, a7), b7 = pmin(a7, rot16(a1));                     // 4-runs
        const unsigned c1 = pmin(b1, b5), c3 = pmin(b3, b7), c5 = pmin(b5, rot16(b1)), c7 = pmin(b7, rot16(b3));               // 8-runs
        unsigned bst = pmin(c1, pmax(P[0], rot16(P[1])));
        bst = pmax(bst, pmin(c3, pmax(P[2], rot16(P[3]))));
        bst = pmax(bst, pmin(c5, pmax(P[4], rot16(P[5]))));
        bst = pmax(bst, pmin(c7, pmax(P[6], rot16(P[7]))));
        const int b0 = (int)(short)(bst & 0xFFFF), b1s = (int)(short)(bst >> 16);
        best = max(b0, b1s);
    }
    return best;
}

// Round 6 experiment, NEGATIVE (profiles/fast_prefilter_r06.txt), off by default: a DWORD-level pre-filter in front of stage A of
// the first round.  A pixel passes the compass test only if one of (U, D) and one of (L, R) differs from it by more than t, so for the
// four pixels of a dword  min(max(SAD(C,U), SAD(C,D)), max(SAD(C,L), SAD(C,R))) > t  is necessary for ANY of them to pass: four
// v_sad_u8, three min / max and a compare per dword instead of the 36 packed operations of the test itself; at iniThFAST 70 % of the
// dwords of the synthetic stream (61 % of EuRoC's) fail it.  The survivors' ids go through a 128-entry ring in LDS and the full test
// runs on 64 of them at a time.  Parity-green -- and 5 % fewer vector instructions for 5 % MORE time (256 -> 270 us): the survivors of
// a cell rarely fill whole 64-lane passes (1.25 on average, most cells pay two), the gathered tile reads of the second stage conflict
// where the row-major ones did not (SQ_LDS_BANK_CONFLICT +30 %), and the ring costs two wave fences per pass.
// -DGFO_FAST_PREFILTER=1 builds it.
#ifndef GFO_FAST_PREFILTER
#define GFO_FAST_PREFILTER 0
#endif
#define GFO_FAST_RING_BYTES 0     // the ring lives in the last 128 entries of the pixel queue (a capped queue only: see q_cap below)

// TP / SP: pitch of the pixel tile and of the score map in LDS, compile-time so that every ring and neighbour
// offset is an immediate of the LDS instruction (three buckets cover cells up to 64 px).
// XCD8: workgroups are dealt round-robin over the 8 XCDs (workgroup b runs on XCD b % 8), so with the plain
// (cells, images) grid the four neighbours of a cell -- which share its 6-px overlap rows and columns and the
// 16-byte segments around them -- sit on four OTHER XCDs and every one of them pulls the shared lines through the
// fabric into its own L2 (measured 344 MB fetched per 143 MB of pyramid).  The XCD8 grid is
// (8 * cell blocks, ceil(images / 8)): blockIdx.x & 7 picks the image inside a group of eight, so ALL cells of an
// image run on one XCD and an image's levels (1.1 MB) enter exactly one 4-MB L2.  Speed only: any placement gives
// the same candidates.
template <int TP, int SP, bool XCD8>
__global__ __launch_bounds__(256) void k_fast(const GfoGeom* __restrict__ gp, GfoInput in,
                                              const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand,
                                              int* __restrict__ cand_cnt, int* __restrict__ flags, const int* __restrict__ cell_tab,
                                              int nimg
#ifdef GFO_FAST_DEBUG
                                              , int dbg_stop   // tools/pmc_fast_phases.sh only: truncate after a phase
#endif
                                              )
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const GfoGeom& g = *gp;
    // the wave index is uniform: keep it (and the cell geometry derived from it) in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int img = XCD8 ? (int)(blockIdx.y * 8 + (blockIdx.x & 7)) : (int)blockIdx.y;
    if (XCD8 && img >= nimg) return;   // the last group of 8 may be partly empty
    const int cell = (int)(XCD8 ? blockIdx.x >> 3 : blockIdx.x) * (int)(blockDim.x >> 6) + wave;
    const int tile_bytes = TP * g.fast_tile_rows, smap_bytes = (SP * g.fast_smap_rows + 15) & ~15;
    // one u16 queue, compacted in place.  It holds fast_q_cap entries, not the cell's whole scan area: 768 instead of up to
    // 1224 is what lets 32 waves instead of 27 share a CU's LDS (the kernel is bound by wave latency x occupancy: truncated
    // after the tile load it still takes a third of its time, tools/fast_phase_times.sh).  A cell that would overflow it
    // (noise-like imagery) scores what it has queued before it queues more, and drops the corner list for a dense pass over
    // the score map if even the corners do not fit (below).
#if GFO_FAST_PREFILTER
    // the pre-filter's ring takes the last 128 entries of a CAPPED queue (one that flushes anyway when it fills: 640 entries instead of
    // 768); a queue that holds the whole scan area has no room to give and its cells take the plain loop
    const bool prefilter = g.fast_q_cap < g.fast_npx_max && g.fast_q_cap >= 512;
    const int q_cap = prefilter ? g.fast_q_cap - 128 : g.fast_q_cap;
#else
    const int q_cap = g.fast_q_cap;
#endif
    const bool q_capped = q_cap < g.fast_npx_max;                        // wave-uniform
    const int per_wave = tile_bytes + smap_bytes + 2 * g.fast_q_cap;
    uint8_t* tile = lds + wave * per_wave;
    uint8_t* smap = tile + tile_bytes;
    unsigned short* qa = reinterpret_cast<unsigned short*>(smap + smap_bytes);
    unsigned short* qb = qa;  // stage B1 writes entry j <= i after reading entry i (same wave, in order)
#if GFO_FAST_PREFILTER
    unsigned short* ring = qa + q_cap;   // 128 dword ids waiting for the full compass test (stage A of the first round)
#endif

    if (cell >= g.total_cells) return;
    // cell -> (level, row, column) through the table plan() uploads: one scalar load instead of a search over the
    // level prefix table and an integer division
    const int4 ct = reinterpret_cast<const int4*>(cell_tab)[cell];
    const int ce = __builtin_amdgcn_readfirstlane(ct.x);
    // lane -> (row, column) maps of the tile load and of stage A: n | (64 / n) << 5 | ceil(4096 / n) << 12, from plan()
    const int map_ld = __builtin_amdgcn_readfirstlane(ct.y), map_a = __builtin_amdgcn_readfirstlane(ct.z);
    const int level = ce & 15, ci = (ce >> 4) & 0xFFF, cj = ce >> 16;
    const GfoLevel& L = g.lv[level];
    const int wcell = L.wcell, hcell = L.hcell;
    const int iniX = GFO_MIN_BORDER + cj * wcell, iniY = GFO_MIN_BORDER + ci * hcell;
    const int maxX = min(iniX + wcell + 6, L.max_bx), maxY = min(iniY + hcell + 6, L.max_by);
    if (iniY >= L.max_by - 3 || iniX >= L.max_bx - 6) return;  // ORBextractor.cc:796,805
    const int cw = maxX - iniX, ch = maxY - iniY;
    const int sw = cw - 6, sh = ch - 6;
    if (sw <= 0 || sh <= 0) return;
    // The tile starts ONE byte left of the cell, so that scan column 0 (cell column 3) sits at tile column 4: every lane of
    // stage A then owns four scan pixels of an aligned LDS dword, a 30..32-px cell is 8 dwords wide and 8 rows of it fill the
    // wave -- four stage-A passes for a 30x30 cell.  (Round 2 started the tile at the aligned dword left of the cell: the scan
    // columns began at byte 1 or 3 of a dword, 9 dwords per row, 7 rows per pass, five passes -- 5.05 on average at 752x480,
    // 4.16 now.)  The price is global loads at odd addresses, which the memory pipeline takes: the 16-byte loads below are
    // declared unaligned.
    constexpr int xoff = GFO_FAST_XOFF;
    struct __attribute__((packed)) seg16 { uint32_t a, b, c, d; };
    {
        int pitch;
        const uint8_t* src = gfo_level_ptr(g, in, pyr, level, img, &pitch);
        src += (long long)iniY * pitch + (iniX - xoff);   // wave-uniform: the loads below are scalar base + 32-bit lane offset
        // 16-byte row segments: TP is a multiple of 16, every lane issues its (<= 4) wide loads before the
        // first LDS store; the segment right of the cell may run a few bytes past maxX but stays inside the
        // image row (maxX <= w - 16)
        const int spr = map_ld & 31;  // 16-byte segments per tile row (3 for a 30-px cell, at most 5)
        // fixed lane -> (row of the step, segment) map: 64 / spr rows per step, so a load is an offset increment
        // (two steps for a 30-px cell); all loads of a lane are issued before its first LDS store
        const int rps = (map_ld >> 5) & 127;
        const int lr = (lane * (map_ld >> 12)) >> 12;      // lane / spr
        const int lc = lane - lr * spr;
        const bool ld_on = lr < rps;
        // rows past the cell's last one are clamped to it (loaded, not stored)
        const unsigned o_last = (unsigned)((ch - 1) * pitch + 16 * lc);
        const unsigned o_first = (unsigned)(min(lr, ch - 1) * pitch + 16 * lc);
        const unsigned gstep = (unsigned)(rps * pitch);
        uint4* t128 = reinterpret_cast<uint4*>(tile) + lr * (TP >> 4) + lc;
        const int lstep = rps * (TP >> 4);
        if (2 * rps >= ch) {   // the usual cell: two steps, no per-step bookkeeping (wave-uniform branch)
            const seg16 sa = *reinterpret_cast<const seg16*>(src + o_first);
            const seg16 sb = *reinterpret_cast<const seg16*>(src + (rps + lr < ch ? o_first + gstep : o_last));
            const uint4 va = make_uint4(sa.a, sa.b, sa.c, sa.d), vb = make_uint4(sb.a, sb.b, sb.c, sb.d);
            if (ld_on && lr < ch) t128[0] = va;
            if (ld_on && rps + lr < ch) t128[lstep] = vb;
        } else {
            uint4 v[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (k * rps < ch) {   // wave-uniform
                    const seg16 sk = *reinterpret_cast<const seg16*>(src + (k * rps + lr < ch ? o_first + k * gstep : o_last));
                    v[k] = make_uint4(sk.a, sk.b, sk.c, sk.d);
                }
            }
#pragma unroll
            for (int k = 0; k < 6; k++)
                if (k * rps < ch && ld_on && k * rps + lr < ch) t128[k * lstep] = v[k];
        }
        uint4* sm128 = reinterpret_cast<uint4*>(smap);
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int t = lane; t < ((sh + 2) * SP + 15) >> 4; t += 64) sm128[t] = z4;
    }
    wave_sync();
#ifdef GFO_FAST_DEBUG
    if (dbg_stop == 1) return;
#endif
    // The cascade runs at iniThFAST first; only a cell that yields no maximum there repeats it at
    // minThFAST (ORBextractor.cc:811-818).  Scores left in the map by the first round are true S values,
    // so the second round needs no re-initialisation (a stored S <= threshold never suppresses a corner).
    int total = 0, nb = 0;
    for (int round = 0; round < 2; round++) {
    const int tq = round == 0 ? g.ini_th : g.min_th;
    int na = 0;
    bool dense = false;   // wave-uniform: the corner list was dropped (it outgrew the queue); the suppression pass reads the score map densely
    nb = 0;
    // ---- B: score S of the queued survivors [from, na) (corner <=> S > threshold), written to the cell's score map; the corners
    //         are compacted into qb behind the corners already there ----
    auto score_queued = [&](int from) {
        const unsigned short* qr = qa + from + lane;
        for (int i0 = from; i0 < na; i0 += 64, qr += 64) {
            // which lanes hold a queue entry is known to the scalar unit
            const int rem = na - i0;
            const unsigned long long m_in = rem >= 64 ? ~0ull : (1ull << rem) - 1ull;
            int p = 0;
            if (__builtin_amdgcn_inverse_ballot_w64(m_in)) p = *qr;
            const int py = p >> 6, px = p & 63;
            unsigned long long m_run;
            const int sc = corner_score<TP>(tile + py * TP + px, tq, m_run);   // every lane (idle ones re-score pixel 0: wave votes inside)
            const unsigned long long m = __builtin_amdgcn_ballot_w64(sc > tq) & m_in & m_run;   // corners
            unsigned short* qw = qb + nb;
            if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                smap[(py + 1) * SP + px + 1] = (uint8_t)sc;
                if (!dense) qw[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)p;
            }
            nb += __popcll(m);
        }
    };
    // ---- A: compass test over every scan pixel, compacted into qa ----
    // A lane takes one aligned dword (4 pixels) of a tile row: centre, the rows 3 above/below and the
    // dwords left/right come in as five LDS dword reads, and the "second largest / second smallest of the
    // four compass differences" network runs on packed i16 (even and odd bytes of the dword), so four
    // pixels cost ~40 VALU operations instead of ~100.
    {
        const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile);
        const int tp4 = TP >> 2;
        // scan pixel 0 is tile column 4: the row's dwords 1 .. nq hold the scan pixels, only the last one may run past them
        const int nq = map_a & 31;                     // (sw + 3) >> 2
        // lane -> (row lr of the iteration, dword lq of the row), fixed for the cell: 64 / nq rows per iteration,
        // so the loop body has no index arithmetic beyond one add
        const int rpi = (map_a >> 5) & 127;            // 64 / nq; nq <= 16 (cells up to 64 px)
        const int lr = (lane * (map_a >> 12)) >> 12;   // lane / nq
        const int lq = lane - lr * nq;
        const bool lane_on = lr < rpi;
        const int c0 = 4 * lq;                         // scan column of pixel 0 of the dword
        // A pixel right of scan column sw-1 (last dword of a row) must not pass: its threshold is 0x7FFF, which no difference
        // exceeds -- validity costs nothing inside the loop.  Packed per pixel pair: (pixel 0, pixel 2) and (pixel 1, pixel 3).
        const unsigned tq2 = (unsigned)tq | ((unsigned)tq << 16);
        const int nv = sw - c0;                        // valid pixels of this lane's dword (>= 4: all)
        const unsigned tq_e = (nv > 0 ? (unsigned)tq : 0x7FFFu) | ((nv > 2 ? (unsigned)tq : 0x7FFFu) << 16);
        const unsigned tq_o = (nv > 1 ? (unsigned)tq : 0x7FFFu) | ((nv > 3 ? (unsigned)tq : 0x7FFFu) << 16);
        (void)tq2;
        int idx = (lr + 3) * tp4 + 1 + lq;
        int p0 = (lr << 6) + c0;                       // queue entry of the dword's pixel 0: px | py << 6 (cells are at most 64 px wide)
#if GFO_FAST_PREFILTER
        if (round == 0 && prefilter) {
            // thresholds of a row's LAST dword (pixels right of scan column sw - 1 must not pass) and of every other one
            const int nv_last = sw - 4 * (nq - 1);
            const unsigned tq_full = (unsigned)tq | ((unsigned)tq << 16);
            const unsigned tq_e_last = (nv_last > 0 ? (unsigned)tq : 0x7FFFu) | ((nv_last > 2 ? (unsigned)tq : 0x7FFFu) << 16);
            const unsigned tq_o_last = (nv_last > 1 ? (unsigned)tq : 0x7FFFu) | ((nv_last > 3 ? (unsigned)tq : 0x7FFFu) << 16);
            int wr = 0, rd = 0;                        // ring cursors (scalar): wr - rd dword ids are waiting
            // the full compass test on `cnt` (<= 64) waiting dwords, lane i takes ring entry rd + i; survivors' pixels go to qa
            auto full_pass = [&](int cnt) {
                if (q_capped && na + 256 > q_cap) {
                    wave_sync();
                    score_queued(dense ? 0 : nb);
                    wave_sync();
                    na = dense ? 0 : nb;
                    if (!dense && nb + 256 > q_cap) {
                        dense = true;
                        na = 0;
                    }
                }
                const unsigned long long m_in = cnt >= 64 ? ~0ull : (1ull << cnt) - 1ull;
                unsigned t_e = 0, t_o = 0;
                int pd = 0;
                if (__builtin_amdgcn_inverse_ballot_w64(m_in)) {
                    pd = ring[(rd + lane) & 127];
                    const int py = pd >> 6, px = pd & 63;
                    const int di = (py + 3) * tp4 + 1 + (px >> 2);
                    const unsigned C = t32[di], Wm = t32[di - 1], Wp = t32[di + 1];
                    const unsigned U = t32[di + 3 * tp4], D = t32[di - 3 * tp4];
                    const unsigned Lw = __builtin_amdgcn_alignbyte(C, Wm, 1);
                    const unsigned Rw = __builtin_amdgcn_alignbyte(Wp, C, 3);
                    const bool last = (px >> 2) == nq - 1;
                    const unsigned te_l = last ? tq_e_last : tq_full, to_l = last ? tq_o_last : tq_full;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const unsigned ps = h ? 0x0C030C01u : 0x0C020C00u;
                        const s16x2 vc = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, C, ps));
                        const s16x2 pu = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, U, ps));
                        const s16x2 pr = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, Rw, ps));
                        const s16x2 pdn = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, D, ps));
                        const s16x2 pl = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, Lw, ps));
                        const s16x2 dk = vc - __builtin_elementwise_max(__builtin_elementwise_min(pu, pdn), __builtin_elementwise_min(pr, pl));
                        const s16x2 br = __builtin_elementwise_min(__builtin_elementwise_max(pu, pdn), __builtin_elementwise_max(pr, pl)) - vc;
                        const unsigned t1 = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, h ? to_l : te_l) - __builtin_elementwise_max(dk, br));
                        if (h) t_o = t1; else t_e = t1;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned tw = (j & 1) ? t_o : t_e;
                    unsigned long long m;
                    if (j & 2) m = __builtin_amdgcn_ballot_w64((int)tw < 0);
                    else asm("v_cmp_gt_i16_e64 %0, 0, %1" : "=s"(m) : "v"(tw));
                    unsigned short* qj = qa + na;
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) qj[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)(pd + j);
                    na += __popcll(m);
                }
                rd += cnt;
            };
            for (int y0 = 0; y0 < sh; y0 += rpi, idx += rpi * tp4, p0 += rpi << 6) {
                bool keep = false;
                if (lane_on && lr < sh - y0) {
                    const unsigned C = t32[idx], Wm = t32[idx - 1], Wp = t32[idx + 1];
                    const unsigned U = t32[idx + 3 * tp4], D = t32[idx - 3 * tp4];
                    const unsigned Lw = __builtin_amdgcn_alignbyte(C, Wm, 1);
                    const unsigned Rw = __builtin_amdgcn_alignbyte(Wp, C, 3);
                    const unsigned sv = max(__builtin_amdgcn_sad_u8(C, U, 0u), __builtin_amdgcn_sad_u8(C, D, 0u));
                    const unsigned shz = max(__builtin_amdgcn_sad_u8(C, Lw, 0u), __builtin_amdgcn_sad_u8(C, Rw, 0u));
                    keep = min(sv, shz) > (unsigned)tq;
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                if (__builtin_amdgcn_inverse_ballot_w64(m))
                    ring[(wr + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) & 127] = (unsigned short)p0;
                wr += __popcll(m);
                if (wr - rd >= 64) {
                    wave_sync();      // the ring entries written above are read by other lanes
                    full_pass(64);
                }
            }
            if (wr - rd > 0) {
                wave_sync();
                full_pass(wr - rd);
            }
        } else
#endif
        for (int y0 = 0; y0 < sh; y0 += rpi, idx += rpi * tp4, p0 += rpi << 6) {
            if (q_capped && na + 256 > q_cap) {
                // a pass queues up to 256 pixels: make room first -- score what is queued (its corners stay, compacted, at the
                // front), and if the corners alone leave no room either, drop the list: the score map holds everything the
                // suppression pass needs, at the price of walking it pixel by pixel
                wave_sync();
                score_queued(dense ? 0 : nb);
                wave_sync();
                na = dense ? 0 : nb;
                if (!dense && nb + 256 > q_cap) {
                    dense = true;
                    na = 0;
                }
            }
            // the pass bits stay where the packed arithmetic leaves them: the sign bits of the two halves of t_e (pixels 0, 2)
            // and of t_o (pixels 1, 3) -- round 2 gathered them into one word first (mask, shift, or, mask: six operations)
            unsigned t_e = 0, t_o = 0;
            if (lane_on && lr < sh - y0) {
                const unsigned C = t32[idx], Wm = t32[idx - 1], Wp = t32[idx + 1];
                const unsigned U = t32[idx + 3 * tp4], D = t32[idx - 3 * tp4];
                const unsigned Lw = __builtin_amdgcn_alignbyte(C, Wm, 1);   // pixels at column-3
                const unsigned Rw = __builtin_amdgcn_alignbyte(Wp, C, 3);   // pixels at column+3
#pragma unroll
                for (int h = 0; h < 2; h++) {  // h = 0: bytes 0,2   h = 1: bytes 1,3
                    // one v_perm_b32 per operand spreads bytes (h, h+2) into two u16 (selector 0x0C = zero byte)
                    const unsigned ps = h ? 0x0C030C01u : 0x0C020C00u;
                    const s16x2 vc = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, C, ps));
                    const s16x2 pu = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, U, ps));
                    const s16x2 pr = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, Rw, ps));
                    const s16x2 pd = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, D, ps));
                    const s16x2 pl = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(0u, Lw, ps));
                    // a 9-arc holds at least one pixel of every opposite pair: (U, D) = ring 0/8, (R, L) = ring 4/12.
                    // Both pairs hold a darker pixel  <=>  v - max(min(U, D), min(R, L)) > t;
                    // both pairs hold a brighter one  <=>  min(max(U, D), max(R, L)) - v > t   (12 packed ops per pixel pair)
                    const s16x2 dk = vc - __builtin_elementwise_max(__builtin_elementwise_min(pu, pd), __builtin_elementwise_min(pr, pl));
                    const s16x2 br = __builtin_elementwise_min(__builtin_elementwise_max(pu, pd), __builtin_elementwise_max(pr, pl)) - vc;
                    // sign bit set <=> max(dk, br) > threshold
                    const unsigned t1 = __builtin_bit_cast(unsigned, __builtin_bit_cast(s16x2, h ? tq_o : tq_e) - __builtin_elementwise_max(dk, br));
                    if (h) t_o = t1; else t_e = t1;
                }
            }
            // (measured alternative: gathering the pass bits of all iterations per lane and letting every lane pop its
            //  lowest set bit per round -- survivors cluster, the busiest lane holds 8-12 of them, and the rounds cost
            //  more than these four ballots per iteration: 386 vs 373 vector instructions per cell for stage A)
            // Four compares on the sign bits of the words and of their low halves (the 16-bit compare is written out: the
            // compiler turns the C form into a bit-field extract and a compare).  The slot is the count of passing lanes
            // below this one (v_mbcnt) behind a queue pointer that advances in a scalar register.
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned tw = (j & 1) ? t_o : t_e;
                unsigned long long m;
                if (j & 2) m = __builtin_amdgcn_ballot_w64((int)tw < 0);
                else asm("v_cmp_gt_i16_e64 %0, 0, %1" : "=s"(m) : "v"(tw));
                unsigned short* qj = qa + na;
                if (__builtin_amdgcn_inverse_ballot_w64(m)) qj[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = (unsigned short)(p0 + j);
                na += __popcll(m);
            }
        }
    }
    wave_sync();
#ifdef GFO_FAST_DEBUG
    if (dbg_stop == 2) { if (na == 12345) flags[1] = 1; return; }
#endif
    score_queued(dense ? 0 : nb);   // (after a flush the corners so far sit in [0, nb), the unscored survivors behind them)
    wave_sync();
#ifdef GFO_FAST_DEBUG
    if (dbg_stop == 3) { if (nb == 12345) flags[1] = 1; return; }
    if (dbg_stop == 4 && round == 1) return;
    if (dbg_stop == 9) { if (lane == 0) { atomicAdd(&flags[1], sw * sh); atomicAdd(&flags[2], na); atomicAdd(&flags[3], nb); } }
#endif
    // ---- N: strict local maxima of S among the corners; their keys (x, y, S-1) are compacted into the tile's
    //         LDS (the pixels are no longer needed once a maximum exists: a cell with none leaves the tile
    //         untouched for the second round) ----
    int n_max = 0;
    uint32_t* keyq = reinterpret_cast<uint32_t*>(tile);
    const unsigned short* qn = qb + lane;
    // list form: the corners are in qb; dense form (the list was dropped): one scan row per step, lane = column (sw <= 64),
    // a pixel takes part if its stored score is above the threshold
    const int n_steps = dense ? 64 * sh : nb;
    for (int i0 = 0; i0 < n_steps; i0 += 64, qn += 64) {
        unsigned long long m_in;
        int p = 0;
        if (dense) {
            p = ((i0 >> 6) << 6) | min(lane, sw - 1);
            m_in = sw >= 64 ? ~0ull : (1ull << sw) - 1ull;
        } else {
            const int rem = nb - i0;
            m_in = rem >= 64 ? ~0ull : (1ull << rem) - 1ull;
            if (__builtin_amdgcn_inverse_ballot_w64(m_in)) p = *qn;
        }
        // every lane (idle ones look at pixel 0): no branch around the body, the vote is masked in the scalar unit
        const int py = p >> 6, px = p & 63;
        const uint8_t* q = smap + (py + 1) * SP + px + 1;
        const int s = q[0];
        const int nmax = max(max(max(max((int)q[-1], (int)q[1]), max((int)q[-SP - 1], (int)q[-SP])),
                                 max(max((int)q[-SP + 1], (int)q[SP - 1]), max((int)q[SP], (int)q[SP + 1]))), tq);
        // strictly above the eight neighbours and above the threshold (S >= 2 follows: tq >= 1)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(s > nmax) & m_in;
        const int x = px + 3 + cj * wcell, y = py + 3 + ci * hcell;  // ORBextractor.cc:824-825
        const uint32_t key = (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)(s - 1) << 24);
        uint32_t* kw = keyq + n_max;
        if (__builtin_amdgcn_inverse_ballot_w64(m)) kw[__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = key;
        n_max += __popcll(m);
    }
    wave_sync();
    total = n_max;
    if (total > 0) break;
    }  // round
    if (total == 0) return;
    // ---- E: one atomicAdd reserves the cell's slots in the level's candidate list, then a straight copy ----
    int base = 0;
    int* cnt = cand_cnt + (img * g.nlevels + level) * GFO_CNT_STRIDE;
    if (lane == 0) base = atomicAdd(cnt, total);
    base = __shfl(base, 0);
    if (base + total > L.cand_cap) {
        if (lane == 0) atomicOr(&flags[0], 1);
        return;
    }
    uint32_t* out = cand + (long long)img * g.cand_img_stride + L.cand_off + base;
    const uint32_t* keyq = reinterpret_cast<const uint32_t*>(tile);
    for (int i = lane; i < total; i += 64) out[i] = keyq[i];
}

void gfo_launch_fast(gfo_ctx* c, const GfoInput& in, int nimg)
{
    const GfoGeom& g = c->g;
    if (g.total_cells == 0) return;  // image too small for a single 30-px cell on any level: no candidates
    // cells (waves) per workgroup: one -- a cell's slot is free the moment its wave ends (166 us against 172 with four).
    // In the running three-context pipeline four-wave workgroups measured +0.8 % (206.1k -> 207.8k frames/s, same-box
    // A/B: they leave more of a CU to the other contexts' kernels); not worth 6 us of the kernel's own time.
    static const int nw_env = getenv("GFO_FAST_WAVES") ? atoi(getenv("GFO_FAST_WAVES")) : 0;
    const int nw = nw_env >= 1 && nw_env <= 4 ? nw_env : 1;
    const size_t lds = nw * (size_t)(g.fast_tile_pitch * g.fast_tile_rows + ((g.fast_smap_pitch * g.fast_smap_rows + 15) & ~15) + 2 * g.fast_q_cap);
    const int cell_blocks = (g.total_cells + nw - 1) / nw;
    // one image per XCD from 8 images up (below that, 7 of 8 workgroups would be empty: plain grid)
    static const int xcd_env = getenv("GFO_FAST_XCD") ? atoi(getenv("GFO_FAST_XCD")) : 1;
    const bool xcd8 = xcd_env && nimg >= 8;
    const dim3 grid = xcd8 ? dim3((unsigned)cell_blocks * 8u, (unsigned)(nimg + 7) / 8u) : dim3(cell_blocks, nimg);
    gfo_prof_begin(c, ST_FAST);
#ifdef GFO_FAST_DEBUG
    static const int dbg_stop = getenv("GFO_FAST_STOP") ? atoi(getenv("GFO_FAST_STOP")) : 0;  // instruction-count experiments only
#define GFO_FAST_DBG_ARG , dbg_stop
#else
#define GFO_FAST_DBG_ARG
#endif
#define GFO_FAST_LAUNCH(TP_, SP_)                                                                                       \
    do {                                                                                                                \
        if (xcd8)                                                                                                       \
            GFO_LAUNCH(c, (k_fast<TP_, SP_, true>), grid, dim3(64 * nw), lds, c->stream, c->d_geom, in, c->d_pyr, c->d_cand, \
                               c->d_cand_cnt, c->d_flags, c->d_cell_tab, nimg GFO_FAST_DBG_ARG);                        \
        else                                                                                                            \
            GFO_LAUNCH(c, (k_fast<TP_, SP_, false>), grid, dim3(64 * nw), lds, c->stream, c->d_geom, in, c->d_pyr, c->d_cand, \
                               c->d_cand_cnt, c->d_flags, c->d_cell_tab, nimg GFO_FAST_DBG_ARG);                        \
    } while (0)
    if (g.fast_tile_pitch == 48) GFO_FAST_LAUNCH(48, 44);
    else if (g.fast_tile_pitch == 64) GFO_FAST_LAUNCH(64, 60);
    else GFO_FAST_LAUNCH(80, 76);
#undef GFO_FAST_LAUNCH
#undef GFO_FAST_DBG_ARG
    gfo_prof_end(c);
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_fast(std::vector<const void*>& v)
{
#ifndef GFO_FAST_DEBUG
    v.push_back((const void*)k_fast<48, 44, true>); v.push_back((const void*)k_fast<48, 44, false>);
    v.push_back((const void*)k_fast<64, 60, true>); v.push_back((const void*)k_fast<64, 60, false>);
    v.push_back((const void*)k_fast<80, 76, true>); v.push_back((const void*)k_fast<80, 76, false>);
#else
    (void)v;
#endif
}
