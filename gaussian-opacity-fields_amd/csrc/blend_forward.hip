// blend_forward.hip -- K7, the tiled forward alpha-blend (replaces renderCUDA, reference
// forward.cu:409-612).
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; wave w covers the 8x8 pixel quadrant (w & 1, w >> 1) (tile_pixel):
// a compact footprint meets fewer splats than the reference's 16x4 strip, and per-pixel results do not depend on the map.
//
// MI355X design -- the kernel is VALU-issue bound (two fp64 divisions and an exp per contributing pair), and a
// tile list is ~4x longer than the set of splats that actually reach a given pixel (at S1M a wave scans ~420 entries,
// a pixel keeps ~89 candidates, ~72 contribute).  A lock-step loop makes all 64 pixels of a wave pay the
// exact path whenever ANY of them needs it.  Instead, per staged batch of 256 entries:
//
//   phase 1 (cull scan, lane = PIXEL): the entry's FOOTPRINT CONIC g(r) = r^T M r (preprocess_fwd: footprint_bbox; g > 0 <=> the
//     pixel ray misses the alpha >= 1/255 level-set ellipsoid, raised by the forward-error allowance of the blend's own fp32
//     arithmetic) evaluated at the pixel's ray: 5 FMAs per (pixel, entry) in Horner form, written as PACKED fp32 over two
//     consecutive entries (v_pk_fma_f32: the six coefficients are staged SoA in LDS, so one ds_read_b128 broadcast brings one
//     coefficient of FOUR entries as two ready-made operand pairs), and the candidate bit is the sign of g - margin shifted into
//     the mask word with one v_alignbit_b32: 4 VALU instructions per (pixel, entry) where the former box test + fp32 prelude +
//     error-bounded cull took ~40.  Survivors are recorded as a 256-bit mask of the pixel in LDS (s_mask[word][thread]).
//   phase 2 (per-lane ordered consumption): every lane pops the next set bit of its own mask, reads
//     that entry's record from LDS with a per-lane address and runs the exact path (fp64 t /
//     min_value, exp, blend update).  Entries are consumed in ascending list order per pixel, so
//     the per-pixel operation sequence -- and therefore every output bit -- is unchanged; the number
//     of heavy iterations of a wave drops from "#entries any pixel passes" to "max #candidates of
//     one pixel" (~97 at S1M).  The entries that really contributed are written back into the mask and stored to the
//     binning workspace (contributor masks, cmask_base): the backward visits exactly those.
//
//  * tile-list entries are staged as whole 64-byte SplatRec lines (one aligned gather per entry,
//    colour included; the reference re-reads colour from global memory per contributing pair,
//    forward.cu:561).  LDS layout [4][256] float4: staging writes of 64 consecutive lanes are
//    contiguous (conflict-free), phase-2 reads use a per-lane address.
//  * per-wave exit by ballot(done), workgroup exit by __syncthreads_and(done) (forward.cu:475-477).
//  * XCD-aware tile order (xcd_tile_id): neighbouring tiles, which gather the same records, run on
//    the same XCD and share its L2.
//
// Arithmetic: identical operation sequence to the oracle (fp32 products/sums in source order, fp64 for
// AA/BB/min_value, the mapped depth and the final distortion normalisation, forward.cu:504-557, 589):
// colour, depth, alpha, distortion, final_T and n_contrib are bit-identical to the oracle.  The one
// exception is the unit normal (hardware rsq instead of fp64 sqrt + 3 divisions): channels 3-5 agree
// to ~3e-7.
#include "gof_common.h"

namespace gof {

// scan/consume granularity inside a staged batch: smaller = finer early exit once a wave saturates,
// larger = better lane compaction in phase 2
#ifndef GOF_FW_CHUNK
#define GOF_FW_CHUNK 256     // measured A/B at S1M with the packed conic scan: 32 -> 0.932 ms, 64 -> 0.898, 128 -> 0.891, 256 -> 0.887 (the scan is cheap now:
                             // scanning past a pixel's saturation costs less than the lane compaction of a longer candidate run gains)
#endif
constexpr int FW_CHUNK = GOF_FW_CHUNK;
#ifdef GOF_STATS
// developer-only instrumentation (never in the shipped build): [0] scanned wave-entries, [1] candidate
// (lane, entry) pairs, [2] phase-2 wave iterations, [3] exact-pass pairs, [4] contributing pairs, [5] lane-iterations active,
// [6] (GOF_CULL_AUDIT) pairs the exact path accepts that the cull scan had dropped, [7] scanned wave-entries with a candidate in the wave
__device__ unsigned long long g_fw_stats[8];
// one atomic per WAVE and statement: the sum over the lanes that execute it (an atomic per lane and contributing pair -- 5e8 atomics on a
// handful of words at S1M -- made the audit build's full-size frame take tens of seconds of the GPU suite)
__device__ __forceinline__ void stat_add_wave(unsigned long long* p, unsigned long long v, bool is_one)
{
    const unsigned long long active = __ballot(true);
    unsigned long long total;
    if (is_one) total = (unsigned long long)__popcll(active);
    else {
        total = 0;
        const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
        for (unsigned long long m = active; m; m &= m - 1ull) {
            const int l = __builtin_ctzll(m);
            total += ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, l);
        }
    }
    if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(active)) atomicAdd(p, total);
}
#define STAT_ADD(i, v) stat_add_wave(&g_fw_stats[i], (unsigned long long)(v), __builtin_constant_p(v) && (v) == 1)
#else
#define STAT_ADD(i, v)
#endif

#ifdef GOF_TILE_CLOCK
// developer-only (-DGOF_TILE_CLOCK): per-tile start / end of the workgroup on the constant-rate counter (100 MHz): the scheduler's
// efficiency = sum of the durations / (concurrency x makespan), tests/devtools/dev_tile_schedule.py
__device__ unsigned long long g_fw_tile_clock[2][1 << 16];
#define TILE_CLOCK_START() const unsigned long long _t0 = wall_clock64()
#define TILE_CLOCK_END(ARR) do { if (threadIdx.x == 0 && tile < (1u << 16)) { ARR[0][tile] = _t0; ARR[1][tile] = wall_clock64(); } } while (0)
#else
#define TILE_CLOCK_START()
#define TILE_CLOCK_END(ARR)
#endif

#ifndef GOF_FW_WAVES
#define GOF_FW_WAVES 4       // LDS bounds the occupancy at 5 workgroups per CU; asking for 4 leaves the allocator more room (88 VGPR, measured -1 %)
#endif
// one tile: everything below is per tile; called by all 256 threads of the workgroup.
// EXACT = true: a pair's value path in the reference's own arithmetic, fp64 divisions included (pair_exact_cc; the mapped depth's
//   quotient) -- every output bit that of the oracle: the VERIFICATION mode (gof_set_forward_exact(1) / GOF_FW_EXACT=1), on which
//   the image's bit-exactness tests run.
// EXACT = false (default, round 4): the same arithmetic without the two fp64 DIVISIONS per pair (pair_nodiv_cc: the quotient BB / AA
//   by two fp64 corrections of the fp32 reciprocal -- faithfully rounded; the mapped depth in fp32, mapped_depth_fast).  t, alpha, T
//   and with them every decision (t <= near, alpha < 1/255, T (1 - alpha) < 1e-4, T > 0.5), n_contrib, the contributor masks and
//   the colour / depth / alpha channels are those of the exact path -- identical on every scene tested, see pair_nodiv_cc for when
//   a last bit may differ --; the distortion channel and dist1 / dist2 carry the fp32 mapped depth's 1-2 ulp (a few 1e-7).
//   (Measured first and dropped: an fp32-only value path with error-bounded decisions and a redo list for the waves that could
//   not tell -- kernel -9 %, but 2.5 % of the waves had to be rendered again and that launch, one wave per tile, cost 0.13 ms:
//   profiles/r04_ab_call1_*.txt, profiles/HISTORY.md.)
template <bool EXACT>
__device__ __forceinline__ void
blend_forward_tile(const uint32_t tile, uint32_t (*s_mask)[TILE_PIX], const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
                   const float4* __restrict__ fconic, int W, int H, float focal_x, float focal_y, const float* __restrict__ bg_color,
                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                   const MaskPool masks, uint32_t* __restrict__ mask_cursors, uint32_t gx, uint32_t* __restrict__ tile_cost)
{
    TILE_CLOCK_START();
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, lx, ly);
    const uint32_t px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;
    const float rx = (float)(((double)pixfx - W / 2.) / (double)focal_x);
    const float ry = (float)(((double)pixfy - H / 2.) / (double)focal_y);

    const uint2 range = ranges[tile];
    int toDo = (int)(range.y - range.x);
    const int rounds = (toDo + TILE_PIX - 1) / TILE_PIX;

    // LDS record of a staged entry, arranged as the operand PAIRS of the packed prelude (v = view2gaussian):
    //   q0 = {v0, v1 | v1, v3}, q1 = {v2, v4 | v2, v6}, q2 = {v4, v7 | v5, v8}:  (n0, n1) = q0.xy rx + q0.zw ry + q1.xy,
    //   (n2, BB/2) = q1.zw rx + q2.xy ry + q2.zw;   q3 = {v9 = CC, w, r, g}, s_blue = b
    __shared__ f4 s_rec[4][TILE_PIX];
    __shared__ float s_blue[TILE_PIX];
#ifdef GOF_CULL_AUDIT
    __shared__ uint32_t s_cand[TILE_PIX / 32][TILE_PIX];
#endif
    uint32_t* const s_tile = &s_mask[0][0];      // (the epilogue's scratch word: the masks are dead by then)
    // footprint conic, SoA: s_cf[c][entry], c = {m00, 2 m01, m11, 2 m02, 2 m12, m22}; read as float4 = one coefficient of 4 entries
    __shared__ f4 s_cf[6][TILE_PIX / 4];
    // this tile's entries of the contributor-mask table (gof_common.h: MaskPool), one per staged batch
    uint32_t* const mask_entry = masks.table + mask_slot0(range.x, tile);
    // evaluation-error margin of the unit-normalised conic (sum |M_ij| = 1, the doubled off-diagonal coefficients counted doubled):
    // the Horner form below is five FMAs and one add, each rounding a partial sum bounded by B = max(1, rx^2, ry^2) -- 6 eps B,
    // eps = 2^-24 -- plus the fp32 rounding of the six coefficients, <= eps B together: 7 eps B = 4.2e-7 B (the ray is the fp32 ray
    // the exact path uses: no error there).  1e-6 B is 2.4x that.  (Rounds 1-2 carried 3e-6: in min_value units the margin is worth
    // ~ margin x lambda |mu|^2, i.e. as much as the allowance of the level itself at S1M -- 5 % of the heavy trips.)
    const f2 RX = { rx, rx }, RY = { ry, ry }, RXY = { rx, ry };
#ifndef GOF_CONE_MARGIN
#define GOF_CONE_MARGIN 1e-6f
#endif
    const float cone_margin = GOF_CONE_MARGIN * fmaxf(1.0f, fmaxf(rx * rx, ry * ry));
    const f2 NEG_MARGIN = { -cone_margin, -cone_margin };

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0, max_contributor = (uint32_t)-1;
    // accumulators kept as the register pairs the packed updates work on (every element sees the reference's operation
    // sequence: packed fp32 instructions are element-wise IEEE)
    f2 C01 = { 0, 0 }, C2N2 = { 0, 0 }, N01 = { 0, 0 };      // colour 0,1 | colour 2, normal 2 | normal 0,1
    f2 D12 = { 0, 0 }, DA = { 0, 0 };                        // dist1, dist2 | distortion, alpha
    float Dp = 0;

    for (int i = 0; i < rounds; i++, toDo -= TILE_PIX) {
        if (__syncthreads_and(done)) break;
        // the batch's chunk of the mask pool (4 sub-chunks, one per wave): taken by one thread, known to all behind the staging barrier
        if (tid == 0) mask_entry[i] = pool_take(mask_cursors, masks.cap / 4u, 1u, tile);
        const uint32_t k = range.x + (uint32_t)i * TILE_PIX + tid;
        if (k < range.y) {
            const uint32_t id = point_list[k];
            const float4* src = reinterpret_cast<const float4*>(&rec[id]);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            s_rec[0][tid] = f4{ a.x, a.y, a.y, a.w };
            s_rec[1][tid] = f4{ a.z, b.x, a.z, b.z };
            s_rec[2][tid] = f4{ b.x, b.w, b.y, c.x };
            s_rec[3][tid] = f4{ c.y, c.z, c.w, d.x };
            s_blue[tid] = d.y;
            const float4 m0 = fconic[2 * (size_t)id], m1 = fconic[2 * (size_t)id + 1];     // {m00, m01, m11, m02}, {m12, m22, ., .}
            float* cf = reinterpret_cast<float*>(&s_cf[0][0]) + tid;
            cf[0 * TILE_PIX] = m0.x; cf[1 * TILE_PIX] = 2.0f * m0.y; cf[2 * TILE_PIX] = m0.z;
            cf[3 * TILE_PIX] = 2.0f * m0.w; cf[4 * TILE_PIX] = 2.0f * m1.x; cf[5 * TILE_PIX] = m1.y;
        }
        __syncthreads();
        const int nwords_batch = (min(TILE_PIX, toDo) + 31) >> 5;
        const uint32_t chunk_v = mask_entry[i];      // (written by thread 0 in front of the barrier; needed only when the batch's words are stored: the load's latency hides behind the batch)
        if (__ballot(!done) == 0ull) {             // whole wave saturated: it only helps staging (and reports "no contributors")
            const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_v);
            if (chunk != POOL_NONE) {
                uint32_t* const mask_dst = masks.pool + ((size_t)chunk * 4u + (tid >> 6)) * MASK_SUBCHUNK_WORDS + (tid & 63u);
                for (int q = 0; q < nwords_batch; q++) mask_dst[q * 64] = 0u;
            }
            continue;
        }

        const int n = min(TILE_PIX, toDo);
        const uint32_t base = (uint32_t)i * TILE_PIX;

        int words_valid = 0;                     // mask words of this batch that hold contributor bits (wave-uniform)
        for (int c0 = 0; c0 < n; c0 += FW_CHUNK) {
        if (__ballot(!done) == 0ull) break;
        const int cn = min(FW_CHUNK, n - c0);            // entries [c0, c0 + cn) of the staged batch
        const int w0 = c0 >> 5;
        const int nw = w0 + ((cn + 31) >> 5);

        // ---- phase 1: cull scan ----
        // lane = PIXEL; per mask word 8 groups of 4 consecutive entries, highest first, so that shifting the sign bits in from
        // the right leaves entry e of the word at bit e.  g = rx (m00 rx + 2 m01 ry + 2 m02) + ry (m11 ry + 2 m12) + m22 - margin;
        // candidate <=> g <= 0 <=> sign bit (g = +0 has measure zero and lies 4x inside the margin's slack).  Unbounded /
        // degenerate Gaussians carry all-zero coefficients: g = -margin, always a candidate.
        for (int w = w0; w < nw; w++) {
            uint32_t word = 0;
#pragma unroll
            for (int g4 = 7; g4 >= 0; g4--) {
                const int q = w * 8 + g4;
                const f4 m00 = s_cf[0][q], m01 = s_cf[1][q], m11 = s_cf[2][q], m02 = s_cf[3][q], m12 = s_cf[4][q], m22 = s_cf[5][q];
                const f2 a_lo = pk_fma(m00.xy, RX, pk_fma(m01.xy, RY, m02.xy));
                const f2 a_hi = pk_fma(m00.zw, RX, pk_fma(m01.zw, RY, m02.zw));
                const f2 b_lo = pk_fma(m11.xy, RY, m12.xy);
                const f2 b_hi = pk_fma(m11.zw, RY, m12.zw);
                const f2 g_lo = pk_fma(RX, a_lo, pk_fma(RY, b_lo, m22.xy + NEG_MARGIN));
                const f2 g_hi = pk_fma(RX, a_hi, pk_fma(RY, b_hi, m22.zw + NEG_MARGIN));
                word = __builtin_amdgcn_alignbit(word, __float_as_uint(g_hi.y), 31);
                word = __builtin_amdgcn_alignbit(word, __float_as_uint(g_hi.x), 31);
                word = __builtin_amdgcn_alignbit(word, __float_as_uint(g_lo.y), 31);
                word = __builtin_amdgcn_alignbit(word, __float_as_uint(g_lo.x), 31);
            }
            const int valid = n - w * 32;                                    // entries of this word the list covers (>= 1)
            if (valid < 32) word &= (1u << valid) - 1u;                      // the tail of the LDS batch holds stale entries
            if (done) word = 0u;
#ifdef GOF_CULL_AUDIT
            // developer-only audit build (with GOF_STATS): the consumption below walks EVERY entry of the list, not only the scan's
            // candidates, and counts the pairs the exact path accepts although the scan dropped them (stat [6], must stay 0)
            s_cand[w][tid] = word;
            word = done ? 0u : (valid < 32 ? (1u << valid) - 1u : 0xFFFFFFFFu);
#endif
            s_mask[w][tid] = word;
            if ((tid & 63) == 0) STAT_ADD(0, min(32, valid));
            STAT_ADD(1, __popc(word));
#ifdef GOF_STATS
            {   // [7] scanned (wave, entry) pairs in which at least one pixel of the wave is a candidate
                uint32_t any = 0;
                for (int bq = 0; bq < 32; bq++) any |= (__ballot((word >> bq) & 1u) != 0ull) ? (1u << bq) : 0u;
                if ((tid & 63) == 0) STAT_ADD(7, __popc(any));
            }
#endif
        }

        // ---- phase 2: every lane consumes its own candidates in list order ----
        int w = w0;
        uint32_t cur = s_mask[w0][tid];
        uint32_t cbits = 0;                              // contributor bits of word w (flushed when w advances)
        for (;;) {
            const bool more = !done && (cur != 0u || w + 1 < nw);
            if (__ballot(more) == 0ull) break;
            if ((tid & 63) == 0) STAT_ADD(2, 1);
            // ONE divergent region per trip -- the accumulation at its end -- instead of four exits (round 6; until then `if (!more)
            // continue`, `if (cur == 0) continue`, `if (p.skip) continue`, `if (test_T < 1e-4) { done; continue }`, each an exec-mask save /
            // restore and a branch: 186 -> 175 instructions per trip, 48 -> 37 of them scalar): a lane without a candidate runs the
            // pair's arithmetic on entry w * 32 of the batch -- a valid LDS address -- and discards it; the decisions are predicates.
            // Same per-pixel operation sequence: same bits.  Measured, interleaved (profiles/r06_ab_call4_blend_forward_flat.txt):
            // blend_forward 0.730 -> 0.713 ms at S1M, 0.963 -> 0.938 clustered; steps 2.393 -> 2.381 / 3.234 -> 3.222 ms.
            if (more && cur == 0u) { s_mask[w][tid] = cbits; cbits = 0; w++; cur = s_mask[w][tid]; }   // a consumed word becomes its contributor word
            const bool has = more && cur != 0u;
            if (has) STAT_ADD(5, 1);
            const int b = has ? __ffs((int)cur) - 1 : 0;
            if (has) cur &= cur - 1u;
            const int j = w * 32 + b;
            const uint32_t contributor = base + (uint32_t)j + 1u;   // 1-based list position (forward.cu:497)

            // fp32 prelude (forward.cu:504-513) in the reference's operation order, two values per packed instruction
            const f4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j];
            const f2 n01 = (q0.xy * RX + q0.zw * RY) + q1.xy;               // normal[0], normal[1]
            const f2 n2b = (q1.zw * RX + q2.xy * RY) + q2.zw;               // normal[2], BB / 2
            const f2 rn = RXY * n01;
            const float n2 = n2b.x;
            const float AAf = (rn.x + rn.y) + n2b.x;
            const float BBf = 2 * n2b.y;
            PairEval p;
            p.AAf = AAf; p.BBf = BBf;
            if constexpr (EXACT) pair_exact_cc(q3.x, q3.y, p);
            else pair_nodiv_cc(q3.x, q3.y, p);
            const float alpha = p.alpha, t = p.t;
            const float test_T = T * (1 - alpha);
            const bool ok = has && !p.skip;
#if defined(GOF_STATS)
            if (ok) {
                STAT_ADD(3, 1);
#ifdef GOF_CULL_AUDIT
                if (!((s_cand[w][tid] >> b) & 1u)) STAT_ADD(6, 1);
#endif
            }
#endif
            const bool saturated = ok && test_T < 0.0001f;
            done |= saturated;
            if (!ok || saturated) continue;
            float mapped_max_t;
            if constexpr (EXACT) {
                const float max_t = t;
                mapped_max_t = (float)((GOF_FAR_PLANE * max_t - GOF_FAR_PLANE * GOF_NEAR_PLANE) / ((GOF_FAR_PLANE - GOF_NEAR_PLANE) * max_t));
            } else {
                mapped_max_t = mapped_depth_fast(t);
            }
            STAT_ADD(4, 1);

            // unit normal -n / |n|: the reference takes an fp64 sqrt and three IEEE divisions (forward.cu:548-549);
            // here one v_rsq_f32 (<= 1 ulp).  Only the normal channels depend on it: they agree with the oracle to
            // ~3e-7 instead of bit for bit; every other output is unaffected.
            const f2 sq = n01 * n01;
            const float inv_len = __builtin_amdgcn_rsqf((sq.x + sq.y) + n2 * n2 + 1e-7f);
            const f2 NINV = { -inv_len, -inv_len };
            const f2 nn01 = n01 * NINV;
            const f2 bn2 = { s_blue[j], n2 * NINV.x };                    // colour 2 | unit normal 2

            const float A = 1 - T;
            const float mm = mapped_max_t * mapped_max_t;
            const float error = mm * A + D12.y - 2 * mapped_max_t * D12.x;
            const f2 ALPHA = { alpha, alpha }, TT = { T, T };
            const f2 ea = { error * alpha, alpha };
            DA += ea * TT;                                                  // distortion += error * alpha * T | alpha channel += alpha * T
            const f2 m12 = { mapped_max_t, mm };
            D12 += m12 * ALPHA * TT;                                        // dist1 += m alpha T | dist2 += m m alpha T
            C01 += q3.zw * ALPHA * TT;
            C2N2 += bn2 * ALPHA * TT;
            N01 += nn01 * ALPHA * TT;
            if (T > 0.5f) { Dp = t; max_contributor = contributor; }
            T = test_T;
            last_contributor = contributor;
            cbits |= 1u << b;
        }
        s_mask[w][tid] = cbits;
        for (int q = w + 1; q < nw; q++) s_mask[q][tid] = 0u;       // candidate words this pixel never reached (it saturated)
        words_valid = nw;
        }   // chunk
        // the batch's contributor words -> this wave's sub-chunk (a batch the pool had no room for is counted by pool_take -- the
        // caller repeats the frame's forward with more room before its backward -- and not stored)
        const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk_v);
        if (chunk != POOL_NONE) {
            uint32_t* const mask_dst = masks.pool + ((size_t)chunk * 4u + (tid >> 6)) * MASK_SUBCHUNK_WORDS + (tid & 63u);
            for (int q = 0; q < nwords_batch; q++) mask_dst[q * 64] = (q < words_valid) ? s_mask[q][tid] : 0u;
        }
    }

    if (inside) {
        const size_t HW = (size_t)W * H;
        const float dist1 = D12.x, dist2 = D12.y;
        float distortion = DA.x;
        const float distortion_before_normalized = distortion;
        distortion = (float)((double)distortion / ((double)((1 - T) * (1 - T)) + 1e-7));
        final_T[pix_id] = T;
        final_T[pix_id + HW] = dist1;
        final_T[pix_id + 2 * HW] = dist2;
        final_T[pix_id + 3 * HW] = distortion_before_normalized;
        n_contrib[pix_id] = last_contributor;
        n_contrib[pix_id + HW] = max_contributor;
        out_color[0 * HW + pix_id] = C01.x + T * bg_color[0];
        out_color[1 * HW + pix_id] = C01.y + T * bg_color[1];
        out_color[2 * HW + pix_id] = C2N2.x + T * bg_color[2];
        out_color[3 * HW + pix_id] = N01.x;
        out_color[4 * HW + pix_id] = N01.y;
        out_color[5 * HW + pix_id] = C2N2.y;
        out_color[6 * HW + pix_id] = Dp;
        out_color[7 * HW + pix_id] = DA.y;
        out_color[8 * HW + pix_id] = distortion;
    }
    // what this tile cost: the deepest list position any of its pixels blended = the entries the backward stages and walks
#ifndef GOF_NO_TILE_COST      // (developer A/B: what the epilogue costs)
    {
        uint32_t m = inside ? last_contributor : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        __syncthreads();                            // every thread has read the popped tile id / its last mask words
        if (tid == 0) *s_tile = 0u;
        __syncthreads();
        if ((tid & 63u) == 0u) atomicMax(s_tile, m);
        __syncthreads();
        if (tid == 0) tile_cost[tile] = *s_tile;
    }
#endif
    TILE_CLOCK_END(g_fw_tile_clock);
}

// one workgroup per tile; WHICH tile is decided as the workgroup starts (pop_tile, gof_common.h): heaviest first, XCD queues of equal cost
template <bool EXACT>
__device__ __forceinline__ void
blend_forward_body(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
                   const float4* __restrict__ fconic, int W, int H, float focal_x, float focal_y, const float* __restrict__ bg_color,
                   float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                   const MaskPool masks, uint32_t* __restrict__ mask_cursors, uint32_t gx, uint32_t ntiles, const uint32_t* __restrict__ tile_order,
                   uint32_t* __restrict__ tile_queue, uint32_t* __restrict__ tile_cost)
{
    // The kernel's LDS must stay at 32 000 B: 25 allocation granules of 1280 B, five workgroups per CU.  One more word -- a
    // separate slot for the popped tile id -- made it 26 granules and FOUR workgroups per CU (blend_forward 0.886 -> 0.93 ms,
    // measured), so the tile id travels through the first word of the candidate masks, which are free before and after a tile.
    __shared__ uint32_t s_mask[TILE_PIX / 32][TILE_PIX];
    const uint32_t tile = pop_tile(tile_order, tile_queue, tile_queue + NXCD, ntiles, &s_mask[0][0]);
    if (tile >= ntiles) return;
    blend_forward_tile<EXACT>(tile, s_mask, ranges, point_list, rec, fconic, W, H, focal_x, focal_y, bg_color, final_T, n_contrib, out_color, masks,
                              mask_cursors, gx, tile_cost);
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GOF_FW_WAVES, 8)))     // 30 KB of LDS allow 5 workgroups per CU: keep the registers below 512 / 5
blend_forward(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
              const float4* __restrict__ fconic, int W, int H, float focal_x, float focal_y, const float* __restrict__ bg_color,
              float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
              const MaskPool masks, uint32_t* __restrict__ mask_cursors, uint32_t gx, uint32_t ntiles, const uint32_t* __restrict__ tile_order,
              uint32_t* __restrict__ tile_queue, uint32_t* __restrict__ tile_cost)
{
    blend_forward_body<false>(ranges, point_list, rec, fconic, W, H, focal_x, focal_y, bg_color, final_T, n_contrib, out_color, masks, mask_cursors, gx, ntiles, tile_order, tile_queue, tile_cost);
}
// the verification mode (gof_set_forward_exact(1) / GOF_FW_EXACT=1): every pair in the reference's own arithmetic
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GOF_FW_WAVES, 8)))
blend_forward_exact(const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const SplatRec* __restrict__ rec,
                    const float4* __restrict__ fconic, int W, int H, float focal_x, float focal_y, const float* __restrict__ bg_color,
                    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
                    const MaskPool masks, uint32_t* __restrict__ mask_cursors, uint32_t gx, uint32_t ntiles, const uint32_t* __restrict__ tile_order,
                    uint32_t* __restrict__ tile_queue, uint32_t* __restrict__ tile_cost)
{
    blend_forward_body<true>(ranges, point_list, rec, fconic, W, H, focal_x, focal_y, bg_color, final_T, n_contrib, out_color, masks, mask_cursors, gx, ntiles, tile_order, tile_queue, tile_cost);
}

#ifdef GOF_TILE_CLOCK
extern "C" int gof_debug_fw_tile_clock(unsigned long long* out, int ntiles)      // out[2][ntiles]: start, end
{
    (void)hipDeviceSynchronize();
    if (ntiles > (1 << 16)) return -1;
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fw_tile_clock), sizeof(unsigned long long) * ntiles, 0);
    (void)hipMemcpyFromSymbol(out + ntiles, HIP_SYMBOL(g_fw_tile_clock), sizeof(unsigned long long) * ntiles, sizeof(unsigned long long) * (1 << 16));
    return 0;
}
#endif
#ifdef GOF_STATS
extern "C" int gof_debug_fw_stats(unsigned long long* out8, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fw_stats), sizeof(g_fw_stats));
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fw_stats), z, sizeof(z)); }
    return 0;
}
#endif

} // namespace gof
