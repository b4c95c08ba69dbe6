// k_orient_desc.hip -- IC_Angle (ORBextractor.cc:76-103), computeOrbDescriptor (:107-146) and
// the keypoint bookkeeping of ComputeKeyPointsOctTree / operator() (:839-849, 1144-1172).
//
// TWO keypoints per wavefront, one per 32-lane half, both of ONE pyramid level (pair table of plan()): level, plane
// pointers, pitch and the output row's prefix are scalar.  Where the time goes (profiles/orient_phase_times_r03.txt,
// profiles/vmem_rate_r03.txt): the eight 16-byte window loads per lane -- a vector load costs the texture path ~2.2 clocks
// per distinct 64-byte chunk it touches, and 68 window rows per keypoint are ~120 of them -- and the rate at which the
// chip starts waves; the arithmetic below is a tenth of the kernel.
//   tables     : the 256 test pairs and the disc's row weights enter LDS once per workgroup (16 bytes per thread);
//   staging    : the 31x31 patch of the level and the 37x37 window of the BLURRED level (the rotated
//                pattern reaches 18 px) go to LDS as aligned 16-byte segments, all loads of a
//                lane in flight before the first LDS store; patch and window share one LDS region, the window's
//                bytes wait in registers while the patch is consumed;
//   angle      : a lane owns a disc ROW: v_dot4_u32_u8 of the row's eight dwords against its moment weights and
//                membership bytes; the two int32 moments are summed over the half by DPP steps and a ds_swizzle;
//                fastAtan2 is the plain-fp32 polynomial of include/gfo_sincos.h, sin / cos its double evaluation;
//   descriptor : each lane evaluates 8 of the 256 pair tests; the half of the wave ballot that belongs
//                to a keypoint in round r IS its descriptor bytes 4r..4r+3 -- sixteen v_writelane move the
//                scalar masks into lanes 0..7 of each half, and the 32 bytes leave as eight 32-bit words.
// Output rows are laid out level by level, inside a level in list order (:1144-1161).
// One workgroup barrier (the tables must be complete before the first wave reads them); otherwise each wave owns its
// LDS window (LDS operations of one wave execute in issue order), and a wave without work stores its share of the
// tables and leaves.
#include "gfo_internal.h"
// The double-precision coefficients of gfo_sincosf come from a table in memory (scalar loads into register pairs that the
// fma takes directly) instead of 64-bit literals, each of which is two v_mov per wave: 15 constants, 24 vector instructions.
// Not const on purpose: a const table's loads fold back into the literals.
#include "../../include/gfo_sincos_coef.h"
__device__ double k_sincos_coef[GFO_SINCOS_NCOEF] = {GFO_SINCOS_COEF_LIST};
#define GFO_SINCOS_TABLE k_sincos_coef
#include "../../include/gfo_sincos.h"

// the 256 test pairs as floats (one 16-byte load per lane and round, no integer -> float conversion in the loop)
// In memory the two x and the two y of a test pair sit side by side, as the packed-fp32 operands want them (three register
// moves per round otherwise); the table file lists (x0, y0, x1, y1), the constructor reorders at compile time.
struct PatQuad {
#ifdef GFO_OD_FILE_ORDER
    float x0, y0, x1, y1;
#else
    float x0, x1, y0, y1;
#endif
    constexpr PatQuad(float ax0, float ay0, float ax1, float ay1) : x0(ax0), x1(ax1), y0(ay0), y1(ay1) {}
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ const PatQuad k_pattern[256] = {
#include "../../include/gfo_pattern.inc"
};
// ORBextractor.cc:451-468: umax[|v|], the half-width of the orientation disc in row v
#define GFO_UMAX_LIST 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3
// Row weights of the disc for v_dot4_u32_u8: one row per |v| = 0 .. 15 (rows +v and -v share umax) and an all-zero row 16 for
// the idle lane, each as 32 bytes for the columns u = -15 .. 16: first the moment weights (u + 16 inside the disc, 0 outside:
// unsigned, the bias comes off as 16 * the row sum), then the membership bytes (1 / 0) -- eight dwords each.
#define OD_AW_ROWS 17
struct AngleRows {
    uint32_t w[OD_AW_ROWS][16];
    constexpr AngleRows() : w()
    {
        const int umax[16] = {GFO_UMAX_LIST};
        for (int av = 0; av < 16; av++) {
            for (int i = 0; i < 32; i++) {
                const int u = i - 15, au = u < 0 ? -u : u;
                if (au <= umax[av]) {
                    w[av][i >> 2] |= (uint32_t)(u + 16) << (8 * (i & 3));
                    w[av][8 + (i >> 2)] |= 1u << (8 * (i & 3));
                }
            }
        }
    }
};
__device__ const AngleRows k_angle_rows = AngleRows();

#define DW 37          // descriptor window (blurred level), rows
#define DWP 48         // its LDS pitch: three 16-byte segments cover 37 px + up to 11 px of alignment slack
#define OW 31          // orientation patch (unblurred level), rows
// The patch's LDS pitch.  Round 6 (profiles/orient_lds_r06.txt): 52 bytes = 13 dwords.  With 48 (12 dwords) the 30 storing lanes of
// a half -- (row, segment) = (hl / 3, hl % 3), dword 12 * row + 4 * segment + k -- fall on EIGHT of the 32 banks a ds_write*_b32
// sees (every start a multiple of 4): a four-way conflict on every one of the sixteen dword stores per lane, and the row reads of
// the angle phase (lane = row, stride 12 dwords) the same on their dword read.  13 is odd: rows 13 dwords apart walk all 32 banks,
// the stores land on 30 different banks but for a few pairs, and the row is read as nine conflict-free dwords.
#ifndef OWP
#define OWP 52
#endif
#define OD_REGION (DW * DWP + 16)   // one keypoint's LDS region and 16 bytes of slack behind it (= in front of the next region)
static_assert(OW * OWP + 16 <= OD_REGION && OWP % 4 == 0 && OWP >= 48, "the patch must fit the window's region");
#define OD_STEPS 4     // 10 rows per step: 4 steps cover the 31-row patch and the 37-row window
// The window's segments are stored where they lie -- 16-byte-aligned ds_write_b128, conflict-free at a 48-byte pitch (eight
// consecutive lanes cover the 32 banks exactly) -- and the row's first pixel sits at byte woff = 0 .. 11 of its LDS row: the
// gathers address bytes, the offset is one more term of their base.  (Round 5 shifted the stores left to keep the pixel at byte
// 0 .. 3: two ds_write2_b32 per segment, four-way conflicts on each.)  -DGFO_OD_WIN_ALIGNED=0: the shifted stores.
#ifndef GFO_OD_WIN_ALIGNED
#define GFO_OD_WIN_ALIGNED 1
#endif

// Sum over the 32 lanes of a half, left in every lane: four DPP steps inside the 16-lane rows (lane ^ 1, lane ^ 2, the other
// quad of the eight, the other eight of the row -- each folds into its v_add) and one ds_swizzle across the two rows.
// (__shfl_xor is a ds_bpermute with its address arithmetic: six vector instructions and an LDS round trip per step.)
__device__ __forceinline__ int half_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    v += __builtin_amdgcn_ds_swizzle(v, 0x401F);                     // bit mode: lane ^ 16
    return v;
}

// Round 5: window loads from 16-BYTE-aligned addresses.  A vector load costs the texture path per distinct 64-byte chunk it
// touches (~2.2 clocks each, profiles/vmem_rate_r03.txt) and a 16-byte load from a dword-aligned address straddles a chunk
// boundary 3 times in 16: 1.19 chunks per load.  The 31-px patch row fits three segments from ANY 16-byte boundary left of it
// (15 + 31 <= 48), the 37-px window row from twelve of the sixteen (11 + 37 <= 48; the other four keep the dword-aligned base):
// 242 -> 209 chunks per keypoint.  The LDS image keeps its layout -- pixel 0 of a row at byte (offset & 3) -- by storing every
// segment (offset & ~3) bytes further left (4-byte-aligned LDS stores: two ds_write2_b32 instead of one ds_write_b128; the region
// has 16 bytes of slack in front).  -DGFO_OD_ALIGN16=0: the dword-aligned bases of rounds 2-4.
// MEASURED (profiles/orient_align16_r05.txt, same-box A/B): 236-237 -> 233 us per 256 images, headline +0.4 % (inside the noise):
// kept because it is never slower, but the chunk count is NOT this kernel's wall -- 14 % fewer chunks bought 1.5 %.
#ifndef GFO_OD_ALIGN16
#define GFO_OD_ALIGN16 1
#endif
typedef unsigned int od_u4a4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte value at a 4-byte-aligned LDS address

#ifndef OD_WAVES
#define OD_WAVES 4   // waves per workgroup, 2 keypoints each (2-wave workgroups measured 5 % slower)
#endif
// What the head of a wave needs before its first pixel load, BY VALUE in the kernel arguments (one scalar round trip): truncated
// after its selection word the kernel still took 97 of its 240 us per 256 images -- the per-level counts were read one dependent
// scalar load after the other through the geometry pointer, and the 16-byte table loads had to land in LDS before the selection
// word was even requested (tools: -DOD_STOP=n, profiles/orient_phase_times_r03.txt).
struct OdK {
    int nlevels, kp_stride;
    long long total_sel_cap;
    int sel_cap[GFO_MAX_LEVELS];
};

__global__ __launch_bounds__(64 * OD_WAVES) void k_orient_desc(const GfoGeom* __restrict__ gp, OdK ok, GfoInput in,
                                                     const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                     const uint32_t* __restrict__ sel, const int* __restrict__ sel_cnt,
                                                     gfo_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                     int* __restrict__ kp_cnt, int* __restrict__ flags, const int* __restrict__ od_tab,
                                                     int nimg)
{
    // ONE LDS region per keypoint, used twice: first the 31-row patch (angle), then -- once the angle's reads are done --
    // the 37-row window, whose bytes wait in registers meanwhile.  14 KB per workgroup instead of 26 KB: LDS no longer
    // caps the kernel at 6 waves per SIMD (the registers allow 8), and the chain of dependent loads at the head of
    // every wave is what the extra waves hide.
    // 16 bytes of slack, then [wave*2 + half][OD_REGION]: a region's slack is its own -- the patch's segments are stored up to 12 bytes
    // to the left of their row (GFO_OD_ALIGN16), which for row 0 is the slack in front of the region, never a neighbour's bytes
    __shared__ __attribute__((aligned(16))) uint8_t s_win_raw[16 + 2 * OD_WAVES * OD_REGION];
    // The two tables every wave needs -- the 256 test pairs (4 KB) and the disc's row weights (1.1 KB) -- are brought into LDS
    // ONCE per workgroup, 16 bytes per thread, instead of twelve 16-byte loads per lane and wave: this kernel is bound by the
    // rate at which a CU's texture path takes vector memory instructions (22 per wave were 62 % of its time; with the row
    // weights fetched per lane as well it went from 128 to 144 us), not by arithmetic.  19.7 KB per workgroup: eight still fit.
    __shared__ __attribute__((aligned(16))) float s_pat[256][4];   // PatQuad rows
    __shared__ __attribute__((aligned(16))) uint32_t s_aw[OD_AW_ROWS][20];   // rows padded to 80 B: 16-byte reads of eight consecutive rows then hit all 32 banks once (64-B rows: four ways)
    static_assert(sizeof(PatQuad) == 16 && sizeof(s_pat) == 4096 && sizeof(k_pattern) == 4096, "pattern table layout");
    uint4 tab_p = make_uint4(0, 0, 0, 0), tab_w = make_uint4(0, 0, 0, 0);
    {
        constexpr int NT = 64 * OD_WAVES;
        static_assert(NT == 256 || NT == 128 || NT == 64, "table staging assumes 64 / 128 / 256 threads");
        // issued first, stored to LDS in front of the first possible exit (every wave of the workgroup must contribute)
        if (NT == 256) {
            tab_p = reinterpret_cast<const uint4*>(k_pattern)[threadIdx.x];
            if (threadIdx.x < 4 * OD_AW_ROWS) tab_w = reinterpret_cast<const uint4*>(k_angle_rows.w)[threadIdx.x];
        }
    }
    const GfoGeom& g = *gp;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform: keep it scalar
    const int half = lane >> 5, hl = lane & 31;
    // XCD-aware placement (speed only): workgroups are dealt round-robin over the 8 XCDs, so workgroup b runs
    // on XCD b % 8.  All workgroups of one image are given the same b % 8, so an image's windows (2.2 MB of
    // pyramid + blurred pyramid, overlapping heavily between keypoints) are fetched into ONE 4-MB L2 instead
    // of all eight.  Any placement gives the same results.
    // The grid is (8 * blocks_per_img, ceil(nimg / 8)): x is a multiple of 8, so the flattened workgroup index
    // modulo 8 -- the XCD -- is blockIdx.x & 7, with no division anywhere.
    const int img_fwd = blockIdx.y * 8 + (blockIdx.x & 7), blk = blockIdx.x >> 3;
    if (img_fwd >= nimg) return;   // the last group of 8 may be partly empty
    // Images in DESCENDING order: the blur before this kernel wrote them ascending, and the level + blurred level of a
    // 128-image batch (286 MB) is a little more than the 256-MB Infinity Cache holds -- walking them in the same
    // direction would find every image just evicted, walking back finds all but the first few still on the die.
#ifdef GFO_OD_FORWARD
    const int img = img_fwd;
#else
    const int img = nimg - 1 - img_fwd;
#endif
    // A wave owns the selection slots (2 * pair, 2 * pair + 1) of ONE level (table of plan(): level | pair << 4), so the
    // level and everything derived from it -- plane pointers, pitch, scale, the output row of slot 0 -- live in scalar
    // registers.  (Round 2 walked the compacted output rows and let each half of the wave search its level in the count
    // prefix: 8 levels x 4 selects per wave, and every address 64-bit per lane.)
    // table of plan(), two ints per wave: level | pair << 4, and the level's first selection slot (sel_off): the selection word
    // can be requested as soon as these and the counts are there, without a look at the level's geometry entry
    const int2 pe2 = reinterpret_cast<const int2*>(od_tab)[blk * OD_WAVES + wave];
    const int pe = __builtin_amdgcn_readfirstlane(pe2.x), sel_off = __builtin_amdgcn_readfirstlane(pe2.y);
    const int level = pe & 15, pair = pe >> 4;
    // per-level counts (wave-uniform); output rows are level by level, list order inside a level (:1144-1161)
    // (counts are clamped to the slots a level owns: a selection that was never written cannot send this kernel
    //  outside its buffers).  All of them are requested together (a loop over g.nlevels compiled into one dependent scalar
    //  load per level: ~2 us at the head of every wave).
    int total = 0, prefix = 0, cnt = 0;
#ifndef GFO_OD_VECTOR_COUNTS
    {   // sixteen scalar loads and clamp / select / add chains: a third of the wave's ~380 scalar instructions
        const int nl = ok.nlevels;
        const int* sc = sel_cnt + img * nl;
        int craw[GFO_MAX_LEVELS];
#pragma unroll
        for (int l = 0; l < GFO_MAX_LEVELS; l++) craw[l] = sc[min(l, nl - 1)];   // (clamped index: never past the image's entries)
#pragma unroll
        for (int l = 0; l < GFO_MAX_LEVELS; l++) {
            const int c = l < nl ? min(max(craw[l], 0), ok.sel_cap[l]) : 0;
            prefix += l < level ? c : 0;
            cnt = l == level ? c : cnt;
            total += c;
        }
    }
#else
    {   // Round 6 experiment (-DGFO_OD_VECTOR_COUNTS): lane l holds level l's count -- one vector load, the clamp, a prefix sum over a DPP
        // row, three readlanes.  530 -> 274 scalar instructions in the kernel's text, +17 vector ones; same-box ABAB: 231-233 us against
        // 230-231, headline 290.6 k against 293.4 k -- the scalar instructions were not what the kernel waits for, and the dependent
        // vector load + DPP chain at the head of every wave is longer than sixteen independent scalar loads.  Off.
        const int nl = ok.nlevels;
        const int ll = min(lane & 15, nl - 1);                                    // (clamped index: never past the image's entries)
        const int craw = sel_cnt[img * nl + ll];
        const int cap = g.lv[ll].sel_cap;
        int c = (lane & 15) < nl ? min(max(craw, 0), cap) : 0;
        const int own = c;
        c += __builtin_amdgcn_update_dpp(0, c, 0x111, 0xF, 0xF, false);  // row_shr:1
        c += __builtin_amdgcn_update_dpp(0, c, 0x112, 0xF, 0xF, false);  // row_shr:2
        c += __builtin_amdgcn_update_dpp(0, c, 0x114, 0xF, 0xF, false);  // row_shr:4
        c += __builtin_amdgcn_update_dpp(0, c, 0x118, 0xF, 0xF, false);  // row_shr:8
        total = __builtin_amdgcn_readlane(c, 15);
        cnt = __builtin_amdgcn_readlane(own, level);
        prefix = __builtin_amdgcn_readlane(c, level) - cnt;
    }
#endif
    if (blk == 0 && wave == 0 && lane == 0) {
        kp_cnt[img] = min(total, ok.kp_stride);
        if (total > ok.kp_stride) atomicOr(&flags[0], 8);
    }
    const int nkp = min(total, ok.kp_stride);
    const int i0 = 2 * pair;
    // the tables' share of this thread goes to LDS: in front of an early exit (every wave of the workgroup must contribute),
    // otherwise not before this wave's own loads are under way
    auto store_tables = [&]() {
        if (64 * OD_WAVES == 256) {
            reinterpret_cast<uint4*>(s_pat)[threadIdx.x] = tab_p;
            if (threadIdx.x < 4 * OD_AW_ROWS) *reinterpret_cast<uint4*>(&s_aw[threadIdx.x >> 2][4 * (threadIdx.x & 3)]) = tab_w;
        } else {
            for (int t = threadIdx.x; t < 256; t += 64 * OD_WAVES) reinterpret_cast<uint4*>(s_pat)[t] = reinterpret_cast<const uint4*>(k_pattern)[t];
            for (int t = threadIdx.x; t < 4 * OD_AW_ROWS; t += 64 * OD_WAVES) *reinterpret_cast<uint4*>(&s_aw[t >> 2][4 * (t & 3)]) = reinterpret_cast<const uint4*>(k_angle_rows.w)[t];
        }
    };
    if (pe < 0 || i0 >= cnt || prefix + i0 >= nkp) {   // wave-uniform
        store_tables();
        return;
    }
    const bool act = i0 + half < cnt && prefix + i0 + half < nkp;   // an odd count leaves the last wave's second half idle: it redoes the first keypoint
    const unsigned idx = act ? i0 + half : i0;
    const int slot = prefix + (int)idx;
    const uint32_t* selb = sel + (long long)img * ok.total_sel_cap + sel_off;   // uniform base, 32-bit lane offset
    const uint32_t key = selb[idx];
    const GfoLevel& L = g.lv[level];
#if defined(OD_STOP) && OD_STOP == 1   // tools/ab_variant.sh "-DOD_STOP=n": truncate after a phase (results are wrong by construction)
    if (key != 0x12345678u) return;
#endif
    const int x = (int)(key & 0xFFF) + GFO_MIN_BORDER, y = (int)((key >> 12) & 0xFFF) + GFO_MIN_BORDER;  // :845-846
    const int score = (int)(key >> 24);

    // ---- stage both windows of this half's keypoint: 16-byte loads, all in flight together ----
    // A row of either window is three 16-byte segments from the aligned dword left of it (31 + 3 and 37 + 3
    // bytes fit in 48; the bytes past the window belong to the same row or, at the right image border, to the
    // next row, which exists because keypoints keep 19 px from every edge).  The lane -> (row, segment) map is
    // fixed -- 10 rows x 3 segments per step -- so a load or store costs an offset increment, not an index
    // decomposition.  (Round 3 tried both windows from their exact first byte, loads declared unaligned: the patch is then
    // two segments a row, six vector memory instructions per wave instead of eight -- and the kernel went from 114 to 145 us:
    // a 16-byte load at an odd address costs the texture path about twice an aligned one.)
    int pitch;
    const uint8_t* lv = gfo_level_ptr(g, in, pyr, level, img, &pitch);   // wave-uniform
    const uint8_t* bl = blur + (long long)img * g.blur_img_stride + L.blur_off;
#if GFO_OD_ALIGN16
    const int ox_al = (x - GFO_HALF_PATCH) & ~15;
    const int wx_al = ((x - 18) & 15) < 12 ? (x - 18) & ~15 : (x - 18) & ~3;
#else
    const int ox_al = (x - GFO_HALF_PATCH) & ~3;
    const int wx_al = (x - 18) & ~3;
#endif
    // pixel 0 of a staged row sits at byte (offset & 3) of its LDS row; the segments are stored (offset & ~3) bytes to the left
#if GFO_OD_ALIGN16 && GFO_OD_WIN_ALIGNED
    const int ooff = ((x - GFO_HALF_PATCH) - ox_al) & 3, woff = (x - 18) - wx_al;   // window: 0 .. 11, the segments are stored where they lie
    const int oshift = ((x - GFO_HALF_PATCH) - ox_al) & ~3;
#else
    const int ooff = ((x - GFO_HALF_PATCH) - ox_al) & 3, woff = ((x - 18) - wx_al) & 3;
#if GFO_OD_ALIGN16
    const int oshift = ((x - GFO_HALF_PATCH) - ox_al) & ~3, wshift = ((x - 18) - wx_al) & ~3;
#endif
#endif
    const int lpitch = L.pitch;
    const int rw = (hl * 11) >> 5, seg = hl - 3 * rw;   // hl / 3, hl % 3 for hl < 32; rw == 10: idle lanes
    const bool ld_on = rw < 10;
    const unsigned po = (unsigned)((y - GFO_HALF_PATCH) * pitch + ox_al + 16 * seg);     // row 0 of the patch, this lane's segment
    const unsigned wo = (unsigned)((y - 18) * lpitch + wx_al + 16 * seg);                // row 0 of the window
    uint8_t* win = s_win_raw + 16 + (wave * 2 + half) * OD_REGION;
    uint8_t* pat = win;   // same bytes, earlier in time
    uint4 vw0, vw1, vw2, vw3;   // the window's bytes, in registers until the patch has been consumed
    {
        uint4 vp[OD_STEPS];
#pragma unroll
        for (int k = 0; k < OD_STEPS; k++) {
            const int rp = min(10 * k + rw, OW - 1);   // clamped: idle lanes re-read a valid row
            vp[k] = *reinterpret_cast<const uint4*>(lv + (po + (unsigned)(rp * pitch)));
        }
        vw0 = *reinterpret_cast<const uint4*>(bl + (wo + (unsigned)(rw * lpitch)));
        vw1 = *reinterpret_cast<const uint4*>(bl + (wo + (unsigned)(min(10 + rw, DW - 1) * lpitch)));
        vw2 = *reinterpret_cast<const uint4*>(bl + (wo + (unsigned)(min(20 + rw, DW - 1) * lpitch)));
        vw3 = *reinterpret_cast<const uint4*>(bl + (wo + (unsigned)(min(30 + rw, DW - 1) * lpitch)));
        store_tables();   // (their loads were the first of the kernel: long landed)
        // steps 0-2 store unconditionally: every row they touch exists, and the two idle lanes (rw == 10) hold
        // exactly the bytes that lane rw == 0 of the next step writes to the same place; only the last step is
        // predicated (row 30)
#if GFO_OD_ALIGN16
        uint8_t* pl = pat + rw * OWP + 16 * seg - oshift;
#pragma unroll
        for (int k = 0; k < OD_STEPS - 1; k++) *reinterpret_cast<od_u4a4*>(pl + k * (10 * OWP)) = od_u4a4{vp[k].x, vp[k].y, vp[k].z, vp[k].w};
        if (ld_on && 10 * (OD_STEPS - 1) + rw < OW)
            *reinterpret_cast<od_u4a4*>(pl + (OD_STEPS - 1) * (10 * OWP)) = od_u4a4{vp[OD_STEPS - 1].x, vp[OD_STEPS - 1].y, vp[OD_STEPS - 1].z, vp[OD_STEPS - 1].w};
#else
        static_assert(OWP % 16 == 0, "without GFO_OD_ALIGN16 the patch rows must be 16-byte aligned");
        uint4* pl = reinterpret_cast<uint4*>(pat + rw * OWP + 16 * seg);
#pragma unroll
        for (int k = 0; k < OD_STEPS - 1; k++) pl[k * (10 * OWP / 16)] = vp[k];
        if (ld_on && 10 * (OD_STEPS - 1) + rw < OW) pl[(OD_STEPS - 1) * (10 * OWP / 16)] = vp[OD_STEPS - 1];
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- IC_Angle on the unblurred patch: lane = disc row v, the row's 31 pixels as eight dwords against the row's weight
    //      bytes -- v_dot4_u32_u8 takes four pixels a time, twice (moment and plain sum): 8 byte shifts + 16 dot products per
    //      row instead of the 90 vector instructions of a column per lane (2 byte adds, 2 selects, 2 multiply-adds per row
    //      pair).  m10 = sum of u * I = sum of (u + 16) * I - 16 * sum of I; m01 = sum over the rows of v * (row sum); integer
    //      sums, so the order is immaterial (the reference pairs rows +v / -v, ORBextractor.cc:88-100).  Lane 31 of a half
    //      reads the region's row 31 (not the patch's) against the all-zero weight row.
    // the tables are complete once every wave that is still running has passed here (a wave that left early stored its
    // share first; finished waves do not count at the barrier)
    __syncthreads();
#if defined(OD_STOP) && OD_STOP == 2
    if (key != 0x12345678u) return;
#endif
    int m10, m01;
    {
        uint32_t aw[16];
        {
            const int av = hl < GFO_HALF_PATCH ? GFO_HALF_PATCH - hl : hl - GFO_HALF_PATCH;   // |v|; 16 for the idle lane
            const uint4* awp = reinterpret_cast<const uint4*>(s_aw[av]);
            const uint4 a0 = awp[0], a1 = awp[1], a2 = awp[2], a3 = awp[3];
            aw[0] = a0.x; aw[1] = a0.y; aw[2] = a0.z; aw[3] = a0.w; aw[4] = a1.x; aw[5] = a1.y; aw[6] = a1.z; aw[7] = a1.w;
            aw[8] = a2.x; aw[9] = a2.y; aw[10] = a2.z; aw[11] = a2.w; aw[12] = a3.x; aw[13] = a3.y; aw[14] = a3.z; aw[15] = a3.w;
        }
#if OWP % 16 == 0
        const uint4* prow = reinterpret_cast<const uint4*>(pat + hl * OWP);
        const uint4 q0 = prow[0], q1 = prow[1];
        const uint32_t q2 = reinterpret_cast<const uint32_t*>(prow)[8];
        // the row starts at byte `ooff` of the staged segment: shift it down (the shift count is the same for the whole half)
        const uint32_t d[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2};
#else
        // a pitch that is odd in dwords: nine dword reads, the lanes of a half on 32 different banks each time
        const uint32_t* prow = reinterpret_cast<const uint32_t*>(pat + hl * OWP);
        uint32_t d[9];
#pragma unroll
        for (int k = 0; k < 9; k++) d[k] = prow[k];
#endif
        uint32_t s1 = 0, s0 = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t px = __builtin_amdgcn_alignbyte(d[k + 1], d[k], (uint32_t)ooff);
            s1 = __builtin_amdgcn_udot4(px, aw[k], s1, false);
            s0 = __builtin_amdgcn_udot4(px, aw[8 + k], s0, false);
        }
        m10 = (int)s1 - 16 * (int)s0;
        m01 = (hl - GFO_HALF_PATCH) * (int)s0;
    }
    m10 = half_sum(m10);
    m01 = half_sum(m01);
    const float angle = gfo_fast_atan2f((float)m01, (float)m10);
#if defined(OD_STOP) && OD_STOP == 3
    if (angle != 12345.f) return;
#endif

    // ---- the patch has been read (LDS operations of a wave execute in issue order): the window takes its place ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
#if GFO_OD_ALIGN16 && !GFO_OD_WIN_ALIGNED
        od_u4a4* wl = reinterpret_cast<od_u4a4*>(win + rw * DWP + 16 * seg - wshift);
        wl[0] = od_u4a4{vw0.x, vw0.y, vw0.z, vw0.w};
        wl[10 * DWP / 16] = od_u4a4{vw1.x, vw1.y, vw1.z, vw1.w};
        wl[2 * (10 * DWP / 16)] = od_u4a4{vw2.x, vw2.y, vw2.z, vw2.w};
        if (ld_on && 30 + rw < DW) wl[3 * (10 * DWP / 16)] = od_u4a4{vw3.x, vw3.y, vw3.z, vw3.w};
#else
        uint4* wl = reinterpret_cast<uint4*>(win + rw * DWP + 16 * seg);
        wl[0] = vw0;
        wl[10 * DWP / 16] = vw1;
        wl[2 * (10 * DWP / 16)] = vw2;
        if (ld_on && 30 + rw < DW) wl[3 * (10 * DWP / 16)] = vw3;
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- descriptor on the blurred window ----
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    gfo_sincosf(angle * factorPI, &b, &a);
    // rounding to nearest-even by the 1.5 * 2^23 constant: the sum's low mantissa bits ARE the integer, biased by
    // 0x4B400000; both biases of a point are folded into the window base pointer (rintf + convert would be two
    // more operations per coordinate)
    const float magic = 12582912.0f;
    const unsigned wbase = (unsigned)(18 * DWP + woff + 18) - 0x4B400000u * (unsigned)(DWP + 1);   // mod 2^32
    unsigned long long mk[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const PatQuad p = *reinterpret_cast<const PatQuad*>(s_pat[r * 32 + hl]);
        // both points of the pair at once: packed-fp32 multiply / add (v_pk_mul_f32, v_pk_add_f32), each lane op
        // still the plain IEEE single operation of the scalar form (no contraction: -ffp-contract=off)
        const f32x2 px = {p.x0, p.x1}, py = {p.y0, p.y1};
        const f32x2 fy = (px * b + py * a) + magic, fx = (px * a - py * b) + magic;
        const unsigned iy0 = __float_as_uint(fy.x), ix0 = __float_as_uint(fx.x), iy1 = __float_as_uint(fy.y), ix1 = __float_as_uint(fx.y);
        const int t0 = win[iy0 * (unsigned)DWP + ix0 + wbase], t1 = win[iy1 * (unsigned)DWP + ix1 + wbase];   // u32 arithmetic: the biases cancel
        const unsigned long long m = __builtin_amdgcn_ballot_w64(t0 < t1);
        mk[r] = m;
    }
    // Tests 32r..32r+31 of each half's keypoint go to lane r of that half: sixteen v_writelane from the scalar masks, after
    // the last round (as a per-lane select of a 64-bit shift of the mask this cost five vector instructions a round).  The
    // instructions are written out (this compiler has no builtin for them), so the hazard between a vector compare that
    // writes a scalar register and a v_writelane that reads it is this code's to keep: the wait states stand in front.
    unsigned word = 0;
    asm volatile("s_nop 4\n\t"
                 "v_writelane_b32 %0, %1, 0\n\tv_writelane_b32 %0, %2, 1\n\tv_writelane_b32 %0, %3, 2\n\tv_writelane_b32 %0, %4, 3\n\t"
                 "v_writelane_b32 %0, %5, 4\n\tv_writelane_b32 %0, %6, 5\n\tv_writelane_b32 %0, %7, 6\n\tv_writelane_b32 %0, %8, 7\n\t"
                 "v_writelane_b32 %0, %9, 32\n\tv_writelane_b32 %0, %10, 33\n\tv_writelane_b32 %0, %11, 34\n\tv_writelane_b32 %0, %12, 35\n\t"
                 "v_writelane_b32 %0, %13, 36\n\tv_writelane_b32 %0, %14, 37\n\tv_writelane_b32 %0, %15, 38\n\tv_writelane_b32 %0, %16, 39"
                 : "+v"(word)
                 : "s"((unsigned)mk[0]), "s"((unsigned)mk[1]), "s"((unsigned)mk[2]), "s"((unsigned)mk[3]),
                   "s"((unsigned)mk[4]), "s"((unsigned)mk[5]), "s"((unsigned)mk[6]), "s"((unsigned)mk[7]),
                   "s"((unsigned)(mk[0] >> 32)), "s"((unsigned)(mk[1] >> 32)), "s"((unsigned)(mk[2] >> 32)), "s"((unsigned)(mk[3] >> 32)),
                   "s"((unsigned)(mk[4] >> 32)), "s"((unsigned)(mk[5] >> 32)), "s"((unsigned)(mk[6] >> 32)), "s"((unsigned)(mk[7] >> 32)));
#if defined(OD_STOP) && OD_STOP == 4
    if (word != 0x12345678u) return;
#endif
    if (!act) return;
    const long long o = (long long)img * ok.kp_stride + slot;
    if (hl < 8) reinterpret_cast<unsigned*>(desc_out + o * 32)[hl] = word;
    if (hl == 0) {
        gfo_keypoint q;
        q.x = (float)x;
        q.y = (float)y;
        if (level != 0) {  // :1164-1170
            q.x = q.x * L.scale;
            q.y = q.y * L.scale;
        }
        q.size = (float)L.patch_size;
        q.angle = angle;
        q.response = (float)score;
        q.octave = level;
        q.class_id = -1;
        kp_out[o] = q;
    }
}

void gfo_launch_orient_desc(gfo_ctx* c, const GfoInput& in, int nimg)
{
    const int bpi = (c->od_pairs + OD_WAVES - 1) / OD_WAVES;  // OD_WAVES waves x 2 keypoints per workgroup (the table is padded with -1)
    dim3 grid((unsigned)bpi * 8u, (unsigned)(nimg + 7) / 8u);
    gfo_prof_begin(c, ST_ORIENT_DESC);
    OdK ok{};
    ok.nlevels = c->g.nlevels; ok.kp_stride = c->g.kp_stride; ok.total_sel_cap = c->g.total_sel_cap;
    for (int l = 0; l < c->g.nlevels; l++) ok.sel_cap[l] = c->g.lv[l].sel_cap;
    GFO_LAUNCH(c, k_orient_desc, grid, dim3(64 * OD_WAVES), 0, c->stream, c->d_geom, ok, in, c->d_pyr, c->d_blur, c->d_sel,
                       c->d_sel_cnt, c->d_kp, c->d_desc, c->d_kp_cnt, c->d_flags, c->d_od_tab, nimg);
    gfo_prof_end(c);
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_orient_desc(std::vector<const void*>& v) { v.push_back((const void*)k_orient_desc); }
