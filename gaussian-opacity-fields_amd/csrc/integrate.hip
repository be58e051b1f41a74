// integrate.hip -- K14, the opacity-field level-set query (replaces integrateCUDA, reference
// forward.cu:803-1218).
//
// The reference keeps ~8 KB of per-thread arrays (contributed_ids[1024], projected_*[256],
// point_alphas/Ts[256]) -- scratch-memory traffic on any GPU -- and walks every tile list twice
// per 256-point chunk of a pixel.  MI355X redesign, same results:
//
//   per staged batch of 256 tile-list entries (64-byte SplatRec lines in LDS):
//     phase A (pixel-centric, thread = pixel): advance the 5-sub-ray blending state machine of
//       forward.cu:886-993 and record, per pixel, WHICH list positions contributed as a 256-bit
//       mask in LDS (s_used[word][pixel]: conflict-free for the writers).
//     phase B (point-centric, thread = query point): every point of the tile walks the set
//       bits of ITS pixel's mask in ascending order (the restricted re-walk of
//       forward.cu:1138-1196) and updates its own (alpha_sum, T) pair, carried across batches in
//       out_alpha_integrated / the point workspace.  No per-thread arrays, no second pass over
//       global memory, any number of points per pixel in one sweep.
//
// Observable quirks reproduced: the pass-1 walk never terminates on saturation (`continue`,
// forward.cu:951-956), only on the 1024-contributor cap (:986-990); no T termination and the
// t-clamp in pass 2 (:1172-1190); colour of a point = colour of its pixel (:1207-1208);
// channel 8 = number of points per pixel INCLUDING the re-count of the tile's last point that
// the reference's outer while-loop performs when another pixel of the tile holds more than 256
// points (:1025-1096, see oracle/gof_oracle_integrate.inc).
// The reference keeps the contributor positions of a pixel as uint16 (:879, 983) and its second pass re-finds them by comparing
// the running 32-bit list position with the truncated value (:1145): for tile lists longer than 65535 entries a contributor at
// position c > 65535 makes the second pass evaluate the entry at position c mod 65536 instead (if that lies behind the last
// matched position; otherwise matching stops for good).  Reproduced: the masks handed to the point pass are the positions the
// reference's second pass VISITS (visit_bit below) -- identical to the contributor positions for lists of up to 65535 entries.
//
// Two launches: integrate_pixels (phase A; depends on the Gaussians and the camera only -- writes the contributor
// masks into the binning workspace, layout of cmask_base, and the pixel channels 0-2, 6, 7) and integrate_points
// (phase B; reads masks + records).  The split costs one round trip of the masks (32 B per instance) and lets a
// mesh-extraction driver run phase A ONCE per view for the 9-10 point sets it queries (extract_mesh.py:23-31, 88-100;
// gof_integrate_view / gof_integrate_points).
#include "gof_common.h"

namespace gof {

#ifdef GOF_STATS
// developer-only instrumentation (never in the shipped build) of the opacity-field query's POINT pass: [0] (point, entry) pairs walked
// (set bits of the pixel's contributor mask), [1] skipped by the front-depth test, [2] evaluated, [3] accepted (alpha >= 1/255),
// [4] wave trips of the bit loop, [5] lane-trips with a bit to process; PIXEL pass: [6] candidates popped, [7] of them used by a sub-ray,
// [8] wave trips of the candidate loop (the longest lane's, per mask word), [9] (wave, entry) iterations of the cull scan,
// [10] the trips lanes advancing on their own over a batch's 8 words would take (host counts, 400k Gaussians @ 640x400: 0.83 of [8];
// lane utilisation 0.45 -> 0.54 -- the rest is pixels that have finished while their wave has not); integrate_rays: [6]-[9] per RAY, [11] (audit
// build) pairs accepted outside the scan's candidates, [12]-[14] trips by wave kind, [15] staged batches
__device__ unsigned long long g_int_stats[16];
// one atomic per WAVE and statement: the sum over the lanes that execute it (an atomic per lane and contributing pair -- 5e8 atomics on a
// handful of words at S1M -- made the audit build's full-size frame take tens of seconds of the GPU suite)
__device__ __forceinline__ void istat_add_wave(unsigned long long* p, unsigned long long v, bool is_one)
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
#define ISTAT_ADD(i, v) istat_add_wave(&g_int_stats[i], (unsigned long long)(v), __builtin_constant_p(v) && (v) == 1)
#else
#define ISTAT_ADD(i, v)
#endif


// ---- ray-centric pixel pass (round 5) ----------------------------------------------------------------------------------------
// The five sub-rays of a pixel are its centre and its four half-pixel corners (forward.cu:881-883, 920), and a corner is shared:
// pixf.x + 0.5f of pixel p and pixf.x - 0.5f of pixel p + 1 are the same float (p + 1 exactly), so the four pixels around a corner
// evaluate THE SAME ray bit for bit -- and a sub-ray's recurrence (t, alpha, T: forward.cu:918-975) depends on that ray and the tile
// list only.  What couples the sub-rays of a pixel is an OR (`used`), a max (the depth channel) and the 1024-contributor cap.  A
// tile has 16 x 16 centres + 17 x 17 corners = 545 distinct rays where the pixel-centric form evaluates 5 x 256 = 1280.
//
//   lane = RAY: 9 waves per tile -- rays 0..255 the centre rays (tile_pixel map), 256..511 the 16 x 16 corners that are the top-left
//   corner of a pixel of the tile (same map), 512..544 the 33 corners of the tile's right column / bottom row.  Per staged batch:
//     phase 1: the forward blend's cull scan (footprint conic at the lane's own ray -- exact, no half-pixel allowance --, packed fp32);
//     phase 2: per-lane ordered consumption of the ray's candidates: ONE sub-ray evaluation of forward.cu:921-975 per trip (the
//       pixel-centric form ran five, each behind its own divergent tests); accepted entries stay as the ray's 256-bit mask in LDS;
//     assembly (thread = pixel, after the batch's barrier): used = OR of the pixel's five ray masks -> contributor words, count,
//       last contributor, the uint16 matching of lists beyond 65535 entries -- exactly the old bookkeeping, now outside the hot loop.
//   A ray is finished once T (1 - 1/255) < 1e-4 (see integrate_pixels_tile: nothing can be accepted any more), the tile once all
//   its rays are.  Pixel channels: colour / alpha / final_T from the centre ray, depth = max over the five rays' deepest accepted t.
//   COMPACTION: the reference never lets T fall below 1e-4 (an entry that would is skipped, forward.cu:951-956), so a ray is only
//   finished when T lands in [1e-4, 1.0039e-4) -- most rays of a tile finish within two or three batches, a few walk the whole list
//   accepting ever smaller alphas: counted on the host, 23 of a wave's 64 lanes were still unfinished over the trips of phase 2.  A
//   ray's state is small (T, deepest t, for a centre ray colour and alpha), so whenever the unfinished rays of the tile fit into
//   fewer waves than hold rays now, every lane parks its ray's state in LDS (by ray number), the unfinished rays are numbered by a
//   workgroup prefix sum and lane k adopts the k-th of them: the waves behind them have no ray, skip both phases and wait at the
//   batch barrier.  Masks, state and results are addressed by RAY number, so nothing else knows which lane evaluates a ray.
//   The cap (forward.cu:986-990: a pixel stops for good at its 1024th used entry while its neighbours, who share its corner rays, go
//   on) cannot be honoured ray by ray: a tile in which a pixel reaches it is abandoned (tile_cost = TILE_CAPPED) and rendered by the
//   pixel-centric kernel behind this one (integrate_pixels_capped).  Every output bit is that of the pixel-centric form.
constexpr int IR_CORNER0 = TILE_PIX;            // rays 256..511: corner (i, j), i, j < 16 = top-left corner of pixel (i, j), at tile_thread(i, j)
constexpr int IR_EDGE0 = 2 * TILE_PIX;          // rays 512..528: corners (16, j), j = 0..16; rays 529..544: corners (i, 16), i = 0..15
constexpr int IR_RAYS = 2 * TILE_PIX + 33;
constexpr int IR_THREADS = 576;                 // 9 wave64
constexpr uint32_t IR_NO_RAY = IR_THREADS - 1;  // a lane without a ray: addresses a column / slot of the per-ray arrays that no ray owns
constexpr uint32_t TILE_CAPPED = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t corner_lane(uint32_t ci, uint32_t cj)
{
    if (ci < 16u && cj < 16u) return IR_CORNER0 + tile_thread(ci, cj);
    return ci == 16u ? IR_EDGE0 + cj : IR_EDGE0 + 17u + ci;
}
// ray r of tile (tx, ty): its position in pixel units -- an integer or an integer + 0.5, formed exactly as pixf + offset is
// (forward.cu:920) -- as the ray direction; false if no pixel of the tile inside the image uses the ray
__device__ __forceinline__ bool ray_of(uint32_t r, uint32_t tx, uint32_t ty, int W, int H, float focal_x, float focal_y, float& rx, float& ry)
{
    uint32_t ci, cj;
    if (r < (uint32_t)IR_EDGE0) tile_pixel(r & 255u, ci, cj);
    else { const uint32_t e = r - (uint32_t)IR_EDGE0; ci = e < 17u ? 16u : e - 17u; cj = e < 17u ? e : 16u; }
    const uint32_t px = tx * TILE_X + ci, py = ty * TILE_Y + cj;
    const bool is_centre = r < (uint32_t)TILE_PIX;
    const float posx = is_centre ? (float)px + 0.5f : (float)px, posy = is_centre ? (float)py + 0.5f : (float)py;
    rx = (float)(((double)posx - W / 2.) / (double)focal_x);
    ry = (float)(((double)posy - H / 2.) / (double)focal_y);
    // a corner inside or on the border of the image touches a pixel of this tile that lies inside
    return is_centre ? (px < (uint32_t)W && py < (uint32_t)H) : (r < (uint32_t)IR_RAYS && px <= (uint32_t)W && py <= (uint32_t)H);
}

#ifndef GOF_IR_WAVES
#define GOF_IR_WAVES 7      // keep the registers at 512 / 7 (72): measured, profiles/r05_ab_call2_integrate_rays.txt
#endif
#ifndef GOF_IR_COMPACT
#define GOF_IR_COMPACT 1    // developer A/B: 0 = rays stay on the lanes they start on
#endif
#ifndef GOF_IR_BATCH
#define GOF_IR_BATCH 128    // tile-list entries staged per batch
#endif
#ifndef GOF_IR_EXIT
#define GOF_IR_EXIT 1       // a wave beyond the stagers / pixel threads that holds no ray after a compaction leaves the kernel (its slot and registers go to the next workgroup)
#endif
constexpr int IR_BATCH = GOF_IR_BATCH;
constexpr int IR_WORDS = IR_BATCH / 32;         // mask words of a batch
constexpr int IR_KEEP_WAVES = (2 * IR_BATCH > TILE_PIX ? 2 * IR_BATCH : TILE_PIX) / 64;      // waves that stage (records: threads < IR_BATCH, conics: the next IR_BATCH) or own pixels
static_assert(IR_BATCH == 64 || IR_BATCH == 128 || IR_BATCH == 256, "the staging roles assume 64, 128 or 256 entries per batch");
__global__ void __launch_bounds__(IR_THREADS) __attribute__((amdgpu_waves_per_eu(GOF_IR_WAVES, 8)))
integrate_rays(const uint2* __restrict__ gaussian_ranges, const uint32_t* __restrict__ gaussian_list,
               const SplatRec* __restrict__ rec, const float4* __restrict__ fconic, int W, int H,
               float focal_x, float focal_y, const float* __restrict__ bg_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
               float* __restrict__ out_color, uint32_t* __restrict__ cmask, uint32_t gx, uint32_t ntiles,
               const uint32_t* __restrict__ tile_order, uint32_t* __restrict__ tile_queue, uint32_t* __restrict__ tile_cost)
{
    __shared__ uint32_t s_rmask[IR_WORDS][IR_THREADS];      // [word][ray]: candidate bits of the staged batch, rewritten in place with the accepted ones
    const uint32_t tile = pop_tile(tile_order, tile_queue, tile_queue + NXCD, ntiles, &s_rmask[0][0]);      // longest list first (gof_common.h)
    if (tile >= ntiles) return;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    const bool pixel_thread = tid < (uint32_t)TILE_PIX;      // threads 0..255: the pixels of the tile (assembly, outputs)

    // staged entries: the forward blend's layout (operand pairs of the packed prelude, blend_forward.hip) + the footprint conic, SoA
    __shared__ f4 s_rec[4][IR_BATCH];
    __shared__ float s_blue[IR_BATCH];
    __shared__ f4 s_cf[6][IR_BATCH / 4];
    // a ray's state while no lane holds it (parked at a compaction, and at the end): by ray number
    __shared__ float s_T[IR_THREADS], s_maxt[IR_THREADS];
    __shared__ float s_C[4][TILE_PIX];               // centre rays: colour, alpha
    __shared__ uint16_t s_list[IR_THREADS];          // the unfinished rays, numbered
    __shared__ uint32_t s_wcount[IR_THREADS / 64];
    __shared__ uint32_t s_live[2];                   // unfinished rays of the workgroup, by batch parity
    __shared__ uint32_t s_abort, s_cost;
#ifdef GOF_CULL_AUDIT
    __shared__ uint32_t s_cand[IR_WORDS][IR_THREADS];
#endif

    // the ray this lane holds
    uint32_t ray = tid;
    float rx, ry;
    if (!ray_of(ray, tx, ty, W, H, focal_x, focal_y, rx, ry)) ray = IR_NO_RAY;
    bool done = ray == IR_NO_RAY;
    float T = 1.0f, maxt = 0.0f;
    float C0 = 0, C1 = 0, C2 = 0, Calpha = 0;          // centre rays only

    const uint2 range = gaussian_ranges[tile];
    const int total = (int)(range.y - range.x);
    const int nbatches = (total + IR_BATCH - 1) / IR_BATCH;
    uint32_t* const cm_tile = cmask + cmask_base(range.x, tile) * TILE_PIX;

    // pixel state (threads 0..255, assembly): the bookkeeping of forward.cu:976-990
    uint32_t lx, ly;
    tile_pixel(tid & 255u, lx, ly);
    const bool inside = pixel_thread && tx * TILE_X + lx < (uint32_t)W && ty * TILE_Y + ly < (uint32_t)H;
    const uint32_t c00 = corner_lane(lx, ly), c10 = corner_lane(lx + 1u, ly), c01 = corner_lane(lx, ly + 1u), c11 = corner_lane(lx + 1u, ly + 1u);
    uint32_t last_contributor = 0, n_local = 0;
    uint32_t last_matched = 0;      // uint16 emulation (forward.cu:983, 1145), see integrate_pixels_tile
    bool match_stuck = false;
    if (tid == 0) { s_abort = 0u; s_cost = 0u; s_live[0] = 0u; s_live[1] = 0u; }
    int held_waves = IR_THREADS / 64;                  // waves that may hold a ray (workgroup-uniform)
    int alive_waves = IR_THREADS / 64;                 // waves that have not left the kernel
    __syncthreads();

    for (int b = 0; ; b++) {
        // (not __syncthreads_count: waves may have left the kernel -- the hardware's barrier counts the surviving ones, a library reduction need not)
        { const uint64_t lv = __ballot(!done); if ((tid & 63u) == 0u && lv) atomicAdd(&s_live[b & 1], (uint32_t)__popcll(lv)); }
        __syncthreads();                                   // every wave has finished batch b - 1: the ray masks are complete, s_rec / s_cf are free
        const int n_live = (int)s_live[b & 1];
        if (tid == 0) s_live[(b + 1) & 1] = 0u;            // (last read before the previous batch's staging barrier, next added to behind this batch's)
        if (b > 0 && pixel_thread) {
            // ---- assembly of batch b - 1: the pixel's contributor words from its five rays' masks ----
            const int bb = b - 1;
            const int nb = min(IR_BATCH, total - bb * IR_BATCH);
            const int nwords = (nb + 31) >> 5;
            for (int w = 0; w < nwords; w++) {
                uint32_t u = s_rmask[w][tid] | s_rmask[w][c00] | s_rmask[w][c10] | s_rmask[w][c01] | s_rmask[w][c11];
                if (!inside) u = 0u;                               // (a pixel outside the image may sit next to a live border corner)
                uint32_t word = u;
                if (u) {
                    const uint32_t pos0 = (uint32_t)bb * IR_BATCH + (uint32_t)w * 32u;      // bit k = 1-based list position pos0 + k + 1
                    if (pos0 + 32u <= 0xFFFFu) {                  // stored exactly by the reference's uint16 ids: the second pass finds them where they are
                        n_local += (uint32_t)__popc(u);
                        last_contributor = pos0 + 32u - (uint32_t)__clz((int)u);
                        last_matched = last_contributor;
                    } else {
                        word = 0u;
                        uint32_t rest = u;
                        while (rest) {
                            const int bit = __ffs((int)rest) - 1;
                            rest &= rest - 1u;
                            const uint32_t contributor = pos0 + (uint32_t)bit + 1u;
                            last_contributor = contributor;
                            if (contributor <= 0xFFFFu) {
                                word |= 1u << bit;
                                last_matched = contributor;
                            } else if (!match_stuck) {            // stored mod 2^16: matched at THAT position, if still ahead
                                const uint32_t c16 = contributor & 0xFFFFu;
                                if (c16 > last_matched) {
                                    last_matched = c16;
                                    uint32_t* vw = cm_tile + (size_t)((c16 - 1u) >> 5) * TILE_PIX + tid;   // this thread's own word of an earlier batch
                                    *vw |= 1u << ((c16 - 1u) & 31u);
                                } else match_stuck = true;
                            }
                            n_local += 1;
                        }
                    }
                }
                cm_tile[((size_t)bb * IR_WORDS + w) * TILE_PIX + tid] = word;
            }
            if (n_local >= (uint32_t)MAX_NUM_CONTRIBUTORS * 4) s_abort = 1u;      // the cap: this tile goes to the pixel-centric kernel
        }
        if (n_live == 0 || b == nbatches) break;

#if GOF_IR_COMPACT
        // ---- compaction: the unfinished rays move to the lowest lanes when that empties at least one wave ----
        if ((n_live + 63) / 64 < held_waves) {             // (workgroup-uniform)
            s_T[ray] = T; s_maxt[ray] = maxt;              // park every held ray (a lane without one writes the unowned slot)
            if (ray < (uint32_t)TILE_PIX) { s_C[0][ray] = C0; s_C[1][ray] = C1; s_C[2][ray] = C2; s_C[3][ray] = Calpha; }
            const uint64_t live = __ballot(!done);
            if ((tid & 63u) == 0u) s_wcount[tid >> 6] = (uint32_t)__popcll(live);
            __syncthreads();                               // (the assembly above has read the masks)
            // a ray nobody holds any more accepts nothing: its column stays zero from here on (all columns, by the threads that are left)
            for (uint32_t c = tid; c < (uint32_t)IR_THREADS; c += 64u * (uint32_t)alive_waves)
#pragma unroll
                for (int w = 0; w < IR_WORDS; w++) s_rmask[w][c] = 0u;
            uint32_t slot = (uint32_t)__popcll(live & ((1ull << (tid & 63u)) - 1ull));
            for (uint32_t i = 0; i < (tid >> 6); i++) slot += s_wcount[i];
            if (!done) s_list[slot] = (uint16_t)ray;
            __syncthreads();
            held_waves = (n_live + 63) / 64;
#if GOF_IR_EXIT
            // HARDWARE ASSUMPTION (gfx950, stated because HIP's own rule is stricter): the waves that stay keep executing
            // __syncthreads() after these have returned.  HIP calls a barrier that not every thread of the workgroup reaches undefined;
            // on this GPU the workgroup barrier (s_barrier) counts the waves that have not ended -- a wave that executes s_endpgm leaves
            // the barrier's membership; observed on MI355X over every run of the suite and the soaks of round 5 -- and the exits here are
            // WHOLE waves behind a barrier, wave-uniform, with nothing of theirs left in flight (their rays are parked in LDS above).
            // This file is compiled for gfx950 only (gof_common.h refuses any other target); a port has to build with -DGOF_IR_EXIT=0
            // (the idle waves then stay parked at the barriers: measured 8.23 vs 4.85 ms for the pass).  A regression
            // shows up as a HANG, not as wrong numbers: every GPU test of the query runs under a time-out
            // (tests/test_parity_gpu.py: QUERY_TIMEOUT) and tests/devtools run the kernel under `timeout`.
            alive_waves = max(IR_KEEP_WAVES, held_waves);
            if ((int)(tid >> 6) >= alive_waves) return;                         // (whole waves; every ray they held is parked)
#endif
            if (tid < (uint32_t)n_live) {
                ray = s_list[tid];
                ray_of(ray, tx, ty, W, H, focal_x, focal_y, rx, ry);
                T = s_T[ray]; maxt = s_maxt[ray];
                if (ray < (uint32_t)TILE_PIX) { C0 = s_C[0][ray]; C1 = s_C[1][ray]; C2 = s_C[2][ray]; Calpha = s_C[3][ray]; }
                done = false;
            } else {
                ray = IR_NO_RAY;
                done = true;
            }
            if (tid == 0) ISTAT_ADD(5, 1);                 // [5] compactions
        }
#endif
        const bool centre = ray < (uint32_t)TILE_PIX;
        const f2 RX = { rx, rx }, RY = { ry, ry }, RXY = { rx, ry };
        // evaluation-error margin of the unit-normalised conic in Horner form: blend_forward.hip (7 eps B derived, 1e-6 B carried)
#ifndef GOF_INT_RAY_MARGIN
#define GOF_INT_RAY_MARGIN 1e-6f
#endif
        const float cone_margin = GOF_INT_RAY_MARGIN * fmaxf(1.0f, fmaxf(rx * rx, ry * ry));
        const f2 NEG_MARGIN = { -cone_margin, -cone_margin };

        // ---- staging: threads 0 .. IR_BATCH - 1 the records, the next IR_BATCH the footprint conics ----
        const int toDo = total - b * IR_BATCH;
        const int n = min(IR_BATCH, toDo);
        if (tid < 2u * IR_BATCH) {
            const uint32_t e = tid & (uint32_t)(IR_BATCH - 1);
            const uint32_t k = range.x + (uint32_t)b * IR_BATCH + e;
            if (k < range.y) {
                const uint32_t id = gaussian_list[k];
                if (tid < (uint32_t)IR_BATCH) {
                    const float4* src = reinterpret_cast<const float4*>(&rec[id]);
                    const float4 a = src[0], bq = src[1], c = src[2], d = src[3];
                    s_rec[0][e] = f4{ a.x, a.y, a.y, a.w };
                    s_rec[1][e] = f4{ a.z, bq.x, a.z, bq.z };
                    s_rec[2][e] = f4{ bq.x, bq.w, bq.y, c.x };
                    s_rec[3][e] = f4{ c.y, c.z, c.w, d.x };
                    s_blue[e] = d.y;
                } else {
                    const float4 m0 = fconic[2 * (size_t)id], m1 = fconic[2 * (size_t)id + 1];     // {m00, m01, m11, m02}, {m12, m22, ., .}
                    float* cf = reinterpret_cast<float*>(&s_cf[0][0]) + e;
                    cf[0 * IR_BATCH] = m0.x; cf[1 * IR_BATCH] = 2.0f * m0.y; cf[2 * IR_BATCH] = m0.z;
                    cf[3 * IR_BATCH] = 2.0f * m0.w; cf[4 * IR_BATCH] = 2.0f * m1.x; cf[5 * IR_BATCH] = m1.y;
                }
            }
        }
        __syncthreads();                                   // (also: the assembly of batch b - 1 has read the masks)
        if (s_abort) break;
        if (tid == 0) ISTAT_ADD(15, 1);
        const int nw = (n + 31) >> 5;
        if (__ballot(!done) == 0ull) {                     // every ray of this wave is finished (or it holds none): "nothing accepted" for the batch
            for (int w = 0; w < nw; w++) s_rmask[w][ray] = 0u;
            continue;
        }

        // ---- phase 1: cull scan, lane = RAY (blend_forward.hip: g = r^T M r - margin in Horner form, two entries per packed
        // instruction, candidate <=> sign bit) ----
        for (int w = 0; w < nw; w++) {
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
            // developer-only audit build: the consumption walks EVERY entry and counts the pairs it accepts that the scan dropped ([11], must stay 0)
            s_cand[w][ray] = word;
            word = done ? 0u : (valid < 32 ? (1u << valid) - 1u : 0xFFFFFFFFu);
#endif
            s_rmask[w][ray] = word;
            if ((tid & 63u) == 0u) ISTAT_ADD(9, min(32, valid));             // [9] (wave, entry) pairs scanned
        }

        // ---- phase 2: every lane consumes its ray's candidates in list order: forward.cu:921-975 for ONE sub-ray.  Written without
        // branches around the arithmetic (the three `continue`s of the reference become one predicate): with ~40 rays per wave some lane
        // takes every path anyway, and each divergent exit cost an exec-mask save / restore and a branch per trip ----
        int w = 0;
        uint32_t cur = s_rmask[0][ray];
        uint32_t cbits = 0;                                // accepted bits of word w (flushed when w advances)
        for (;;) {
            const bool more = !done && (cur != 0u || w + 1 < nw);
            if (__ballot(more) == 0ull) break;
#ifdef GOF_STATS
            { const int live_now = __popcll(__ballot(!done)); if ((tid & 63u) == 0u) ISTAT_ADD(10, live_now); }       // [10] unfinished rays over the trips
#endif
            if ((tid & 63u) == 0u) { ISTAT_ADD(8, 1); ISTAT_ADD(tid < 256u ? 12 : (tid < 512u ? 13 : 14), 1); }      // [8] wave trips of the candidate loop; [12] / [13] / [14]: of waves 0-3, 4-7, 8
            if (more && cur == 0u) { s_rmask[w][ray] = cbits; cbits = 0; w++; cur = s_rmask[w][ray]; }
            const bool have = more && cur != 0u;
            const int bit = have ? __ffs((int)cur) - 1 : 0;
            cur &= cur - 1u;                               // (0 stays 0)
            const int j = w * 32 + bit;
            if (have) ISTAT_ADD(6, 1);                     // [6] candidates popped (ray, entry)
            const f4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j];
            const f2 n01 = (q0.xy * RX + q0.zw * RY) + q1.xy;               // normal[0], normal[1]
            const f2 n2b = (q1.zw * RX + q2.xy * RY) + q2.zw;               // normal[2], BB / 2
            const f2 rn = RXY * n01;
            const float AA = (rn.x + rn.y) + n2b.x;
            const float BB = 2 * n2b.y;
            const float CC = q3.x, wgt = q3.y;
            // one IEEE division: -BB/(2*AA) == -(BB/AA)/2 and BB/4 are exact power-of-two scalings (forward.cu:927-931)
            const float q = BB / AA;
            const float t = -q * 0.5f;
            const double min_value = (double)(-q) * (double)(BB * 0.25f) + (double)CC;
            float power = -0.5f * (float)min_value;
            if (power > 0.0f) power = 0.0f;
            const float alpha = fminf(0.99f, wgt * gexpf<true>(power));
            const float test_T = T * (1 - alpha);
            // the reference's three exits (forward.cu:933, 949, 952), each in its own sense of the comparison (NaN falls through as there):
            // (double)t <= NEAR_PLANE is t < 0.2f (see integrate_pixels_tile)
            const bool ok = have & !(t < 0.2f) & !(alpha < 1.0f / 255.0f) & !(test_T < 0.0001f);
#ifdef GOF_CULL_AUDIT
            if (ok && !((s_cand[w][ray] >> bit) & 1u)) ISTAT_ADD(11, 1);
#endif
            if (ok) ISTAT_ADD(7, 1);                       // [7] accepted
            if (centre) {
                const float aT = alpha * T;
                const float c0 = C0 + q3.z * alpha * T, c1 = C1 + q3.w * alpha * T, c2 = C2 + s_blue[j] * alpha * T, ca = Calpha + aT;
                C0 = ok ? c0 : C0; C1 = ok ? c1 : C1; C2 = ok ? c2 : C2; Calpha = ok ? ca : Calpha;
            }
            maxt = (ok && t > maxt) ? t : maxt;
            T = ok ? test_T : T;
            cbits |= ok ? 1u << bit : 0u;
            if (ok && T * (1 - 1.0f / 255.0f) < 0.0001f) done = true;        // nothing can be accepted any more
        }
        s_rmask[w][ray] = cbits;
        for (int q = w + 1; q < nw; q++) s_rmask[q][ray] = 0u;               // candidate words this ray never reached (it finished)
    }

    // results by ray number: every lane parks what it holds, the pixel threads collect
    s_T[ray] = T; s_maxt[ray] = maxt;
    if (ray < (uint32_t)TILE_PIX) { s_C[0][ray] = C0; s_C[1][ray] = C1; s_C[2][ray] = C2; s_C[3][ray] = Calpha; }
    __syncthreads();
    if (s_abort) {                                         // (workgroup-uniform)
        if (tid == 0) tile_cost[tile] = TILE_CAPPED;
        return;
    }
    if (inside) {
        const uint32_t px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
        const uint32_t pix_id = (uint32_t)W * py + px;
        const size_t HW = (size_t)W * H;
        const float Tc = s_T[tid];
        const float depth = fmaxf(fmaxf(s_maxt[tid], fmaxf(s_maxt[c00], s_maxt[c10])), fmaxf(s_maxt[c01], s_maxt[c11]));
        final_T[pix_id] = Tc;
        n_contrib[pix_id] = last_contributor;
        out_color[0 * HW + pix_id] = s_C[0][tid] + Tc * bg_color[0];
        out_color[1 * HW + pix_id] = s_C[1][tid] + Tc * bg_color[1];
        out_color[2 * HW + pix_id] = s_C[2][tid] + Tc * bg_color[2];
        out_color[6 * HW + pix_id] = depth;
        out_color[7 * HW + pix_id] = s_C[3][tid];
    }
    // what the point pass will walk in this tile (the deepest contributor position of its pixels): part of its dispatch cost
    if (pixel_thread) {
        uint32_t m = inside ? last_contributor : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if ((tid & 63u) == 0u) atomicMax(&s_cost, m);
    }
    __syncthreads();
    if (tid == 0) tile_cost[tile] = s_cost;
}

// ---- pixel-centric form (rounds 1-4): thread = pixel, five sub-rays per thread.  Since round 5 the FALLBACK of integrate_rays for
// tiles in which a pixel meets the reference's 1024-contributor cap (integrate_pixels_capped), and the whole pixel pass when
// gof_set_integrate_pixel_pass(1) / GOF_INT_PIXELS=1 asks for it (integrate_pixels: A/B timing, and the tests run both forms).
__device__ __forceinline__ void
integrate_pixels_tile(const uint32_t tile, uint32_t& s_tile, const uint2* __restrict__ gaussian_ranges, const uint32_t* __restrict__ gaussian_list,
                 const SplatRec* __restrict__ rec, const float4* __restrict__ bbox, const float4* __restrict__ fconic, int W, int H,
                 float focal_x, float focal_y, const float* __restrict__ bg_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                 float* __restrict__ out_color, uint32_t* __restrict__ cmask, uint32_t gx, uint32_t* __restrict__ tile_cost)
{
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, lx, ly);
    const uint32_t px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const size_t HW = (size_t)W * H;
    const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;

    const uint2 range = gaussian_ranges[tile];
    int toDo = (int)(range.y - range.x);
    const int nbatches = (toDo + TILE_PIX - 1) / TILE_PIX;

    __shared__ float4 s_rec[4][TILE_PIX];
    __shared__ uint32_t s_used[8][TILE_PIX];
    __shared__ float4 s_box[TILE_PIX];
    __shared__ float4 s_con[2][TILE_PIX];
    const float pxf = (float)px, pyf = (float)py;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    // pixel rectangle of this wave (tile_pixel: wave w covers the 8x8 quadrant (w & 1, w >> 1)), inclusive bounds
    const float wave_x0 = (float)(tx * TILE_X + 8u * (wave & 1u)), wave_x1 = wave_x0 + 7.0f;
    const float wave_y0 = (float)(ty * TILE_Y + 8u * (wave >> 1)), wave_y1 = wave_y0 + 7.0f;
    uint32_t* const cm_tile = cmask + cmask_base(range.x, tile) * TILE_PIX;

    // the 5 sub-rays: centre + 4 half-pixel corners (forward.cu:881-883, 920)
    float srx[5], sry[5];
    {
        const float offx[5] = { 0.0f, -0.5f, 0.5f, -0.5f, 0.5f };
        const float offy[5] = { 0.0f, -0.5f, -0.5f, 0.5f, 0.5f };
#pragma unroll
        for (int c = 0; c < 5; c++) {
            srx[c] = (float)(((double)(pixfx + offx[c]) - W / 2.) / (double)focal_x);
            sry[c] = (float)(((double)(pixfy + offy[c]) - H / 2.) / (double)focal_y);
        }
    }
    // footprint-conic cull: centre ray, the half-pixel steps in ray units, and the evaluation-error margin of the
    // unit-normalised conic (sum |M_ij| = 1): 8 fp32 roundings of terms bounded by max(1, |r|^2) -> 3e-6 * that bound
    const float crx = srx[0], cry = sry[0];
    const float two_hx = 2.0f * (0.5f / focal_x), two_hy = 2.0f * (0.5f / focal_y);
#ifndef GOF_INT_CONE_MARGIN
#define GOF_INT_CONE_MARGIN 3e-6f
#endif
    const float cone_margin = GOF_INT_CONE_MARGIN * fmaxf(1.0f, fmaxf((fabsf(crx) + two_hx) * (fabsf(crx) + two_hx), (fabsf(cry) + two_hy) * (fabsf(cry) + two_hy)));
    float cT[5] = { 1.0f, 1.0f, 1.0f, 1.0f, 1.0f };
    float C0 = 0, C1 = 0, C2 = 0, Cdepth = 0, Calpha = 0;
    uint32_t contributor = 0, last_contributor = 0, n_local = 0;
    uint32_t last_matched = 0;      // uint16 emulation (forward.cu:983, 1145): list position of the second pass's last match,
    bool match_stuck = false;       // and whether a truncated id fell behind it (nothing matches any more)
    bool done = !inside;

    for (int b = 0; b < nbatches; b++, toDo -= TILE_PIX) {
        // A sub-ray whose transmittance T satisfies T * (1 - 1/255) < 1e-4 (in fp32, as the test below evaluates it) is finished
        // for good: an entry is only used with alpha >= 1/255, (1 - alpha) and the product are monotone in alpha, so
        // T * (1 - alpha) < 1e-4 for every later entry -- the reference's loop keeps running there (forward.cu:951-956, the
        // `continue` leaves T untouched, so T never actually falls below 1e-4) but can never use another entry.  A pixel whose
        // five sub-rays are all finished stops here (`done`), a wave whose pixels have all stopped
        // skips its scan, and the tile stops staging once all 256 have: bit-identical results, and on scenes that saturate the
        // bulk of a long tile list is never touched (integrate_points bounds its walk by the tile's last contributor likewise).
        if (__syncthreads_and(done)) break;      // (also the barrier: the previous batch is finished with s_rec)
        const uint32_t k = range.x + (uint32_t)b * TILE_PIX + tid;
        if (k < range.y) {
            const uint32_t id = gaussian_list[k];
            const float4* src = reinterpret_cast<const float4*>(&rec[id]);
            const float4 a4 = src[0], b4 = src[1], c4 = src[2], d4 = src[3];
            s_rec[0][tid] = a4; s_rec[1][tid] = b4; s_rec[2][tid] = c4; s_rec[3][tid] = d4;
            s_box[tid] = bbox[id];
            s_con[0][tid] = fconic[2 * (size_t)id];
            s_con[1][tid] = fconic[2 * (size_t)id + 1];
        }
        __syncthreads();
        if (__ballot(!done) == 0ull) {             // the whole wave has stopped: it only helped staging; "no contributors" for this batch
            const int nz = ((toDo < TILE_PIX ? toDo : TILE_PIX) + 31) >> 5;
            for (int w = 0; w < nz; w++) cm_tile[((size_t)b * 8 + w) * TILE_PIX + tid] = 0u;
            continue;
        }

        // ---- phase A1: cull scan (wave-uniform over entries): footprint box, then the footprint conic bounded over the
        // pixel's 5 sub-rays; survivors are recorded per pixel as candidate bits in s_used ----
        const int n = toDo < 0 ? 0 : (toDo < TILE_PIX ? toDo : TILE_PIX);
        // (a) lane = ENTRY: 64 entries at a time against the wave's 8x8 pixel rectangle (widened by the same pixel as below);
        // (b) lane = PIXEL over the entries that touch the wave.
        for (int w = 0; w < 8; w += 2) {
            const int je = w * 32 + (int)lane;
            bool touch = false;
            if (je < n) {
                const float4 bx = s_box[je];
                touch = (wave_x1 + 1.0f >= bx.x) & (wave_x0 - 1.0f <= bx.y) & (wave_y1 + 1.0f >= bx.z) & (wave_y0 - 1.0f <= bx.w);
            }
            uint64_t m = __ballot(touch);
            uint32_t word_lo = 0, word_hi = 0;
            while (m) {
                const int bb = __builtin_ctzll(m);
                m &= m - 1ull;
                const int j = w * 32 + bb;
                if (lane == 0) ISTAT_ADD(9, 1);                                          // [9] (wave, touching entry) iterations of the cull scan
                // conservative footprint box: integer-rounded bounds {ceil(lo), floor(hi)}, exact for integer pixel positions; the
                // corner sub-rays sit at p +- 0.5, and p + 0.5 >= lo is implied by p + 1 >= ceil(lo): widened by one full pixel.
                const float4 bx = s_box[j];
                const bool inbox = !done & (pxf + 1.0f >= bx.x) & (pxf - 1.0f <= bx.y) & (pyf + 1.0f >= bx.z) & (pyf - 1.0f <= bx.w);
                if (__ballot(inbox) == 0ull) continue;
                // footprint conic g(r) = r^T M r (> 0 outside the alpha >= 1/255 level set, preprocess.hip): lower bound over the
                // centre and the 4 corner sub-rays, against the fp32 evaluation error of the unit-normalised form
                const float4 m0 = s_con[0][j], m1 = s_con[1][j];
                const float tx_ = fmaf(m0.x, crx, fmaf(m0.y, cry, m0.w));          // (M r)_x
                const float ty_ = fmaf(m0.y, crx, fmaf(m0.z, cry, m1.x));          // (M r)_y
                const float g0 = fmaf(crx, tx_, fmaf(cry, ty_, fmaf(m0.w, crx, fmaf(m1.x, cry, m1.y))));
                const float gc = g0 - fabsf(tx_) * two_hx - fabsf(ty_) * two_hy + m1.z;
                const uint32_t cand = (inbox & !(fminf(g0, gc) > cone_margin)) ? 1u : 0u;
                if (bb < 32) word_lo |= cand << bb; else word_hi |= cand << (bb - 32);
            }
            s_used[w][tid] = word_lo;
            s_used[w + 1][tid] = word_hi;
        }
        // ---- phase A2: per-lane ordered consumption of the candidates: the 5-sub-ray state machine of forward.cu:886-993;
        // s_used is rewritten in place with the entries that contributed ----
        // (round 4, built and removed: a per-sub-ray certain-reject pre-test -- fp32 estimate of the power with its error bound against
        // ln(1/255 w) and against ln of what the sub-ray can still take, 1 - 1e-4 / T -- in front of the IEEE division / fp64 / exp.
        // On the host it rejects 48 % of the sub-ray evaluations and 98 % of the rest are used; on the GPU the kernel's time did not
        // move (14.49 ms at the config-5 shape): the lanes of a wave sit at different candidates, one lane on the exact path keeps
        // the wave on it, and 51 % of the evaluations take it.  profiles/HISTORY.md, round 4.)
        // (word by word, the wave moving on together: letting every lane advance over the 8 mask words on its own -- the forward
        // blend's scheme -- was measured SLOWER here, 8.17 -> 8.57 ms at S1M and 15.3 -> 16.6 ms at S5M, as in integrate_points)
#ifdef GOF_STATS
        int my_batch_trips = 0;
#endif
        for (int w = 0; w < 8; w++) {
            uint32_t cand = s_used[w][tid];
            uint32_t word = 0;
#ifdef GOF_STATS
            int my_trips = 0;
#endif
            while (cand && !done) {
#ifdef GOF_STATS
                my_trips++;
#endif
                const int bit = __ffs((int)cand) - 1;
                cand &= cand - 1;
                const int j = w * 32 + bit;
                contributor = (uint32_t)b * TILE_PIX + (uint32_t)j + 1u;      // 1-based list position
                ISTAT_ADD(6, 1);
                const float4 a4 = s_rec[0][j], b4 = s_rec[1][j], c4 = s_rec[2][j];
                const float wgt = c4.z;
                const float log_thr = cull_log_threshold(wgt);
                bool used = false;
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    const float rx = srx[c], ry = sry[c];
                    const float n0 = a4.x * rx + a4.y * ry + a4.z;
                    const float n1 = a4.y * rx + a4.w * ry + b4.x;
                    const float n2 = a4.z * rx + b4.x * ry + b4.y;
                    const float AA = rx * n0 + ry * n1 + n2;
                    const float BB = 2 * (b4.z * rx + b4.w * ry + c4.x);
                    const float CC = c4.y;
                    // one IEEE division: -BB/(2*AA) == -(BB/AA)/2 and BB/4 are exact power-of-two scalings (forward.cu:927-931)
                    const float q = BB / AA;
                    const float t = -q * 0.5f;
                    // (double)t <= NEAR_PLANE (forward.cu:933) as an fp32 compare: 0.2f = 0x3E4CCCCD lies ABOVE the double 0.2, so "t <= 0.2 in
                    // double" is "t < 0.2f" for every float t (NaN: false both ways) -- the forward blend's form (gof_common.h)
                    if (t < 0.2f) continue;
                    const double min_value = (double)(-q) * (double)(BB * 0.25f) + (double)CC;
                    // (float)(-0.5 * min_value): the scaling by a power of two commutes with the rounding (a subnormal result, where it
                    // does not, has exp() == 1 either way): one fp64 multiply less per sub-ray
                    float power = -0.5f * (float)min_value;
                    if (power > 0.0f) power = 0.0f;
                    if (power < log_thr) continue;                           // w * exp(power) < 0.999/255: below the threshold for sure
                    const float alpha = fminf(0.99f, wgt * gexpf<true>(power));      // (power <= 0: clamped above)
                    if (alpha < 1.0f / 255.0f) continue;
                    const float test_T = cT[c] * (1 - alpha);
                    if (test_T < 0.0001f) continue;
                    if (c == 0) {
                        const float4 d4 = s_rec[3][j];
                        C0 += c4.w * alpha * cT[c];
                        C1 += d4.x * alpha * cT[c];
                        C2 += d4.y * alpha * cT[c];
                    }
                    if (t > Cdepth) Cdepth = t;
                    if (c == 0) Calpha += alpha * cT[c];
                    cT[c] = test_T;
                    used = true;
                }
                if (used) {
                    ISTAT_ADD(7, 1);
                    last_contributor = contributor;
                    if (contributor <= 0xFFFFu) {                 // stored exactly: the second pass finds it where it is
                        word |= 1u << bit;
                        last_matched = contributor;
                    } else if (!match_stuck) {                    // stored mod 2^16: matched at THAT position, if still ahead
                        const uint32_t c16 = contributor & 0xFFFFu;
                        if (c16 > last_matched) {
                            last_matched = c16;
                            uint32_t* vw = cm_tile + (size_t)((c16 - 1u) >> 5) * TILE_PIX + tid;   // this thread's own word of an earlier batch
                            *vw |= 1u << ((c16 - 1u) & 31u);
                        } else match_stuck = true;
                    }
                    n_local += 1;
                    if (n_local >= (uint32_t)MAX_NUM_CONTRIBUTORS * 4) done = true;
                    // nothing can be used any more (see the head of the batch loop): the largest T against the smallest usable alpha
                    if (fmaxf(fmaxf(fmaxf(cT[0], cT[1]), fmaxf(cT[2], cT[3])), cT[4]) * (1 - 1.0f / 255.0f) < 0.0001f) done = true;
                }
            }
            s_used[w][tid] = word;
#ifdef GOF_STATS
            my_batch_trips += my_trips;
            for (int o = 32; o > 0; o >>= 1) my_trips = max(my_trips, __shfl_xor(my_trips, o));      // [8] wave trips of the candidate loop = the longest lane's
            if (lane == 0) ISTAT_ADD(8, my_trips);
#endif
        }
#ifdef GOF_STATS
        for (int o = 32; o > 0; o >>= 1) my_batch_trips = max(my_batch_trips, __shfl_xor(my_batch_trips, o));   // [10] what lanes advancing on their own over the batch's 8 words would take
        if (lane == 0) ISTAT_ADD(10, my_batch_trips);
#endif
        // contributor words of this batch -> binning workspace (only the words the list covers)
        const int nwords = (n + 31) >> 5;
        for (int w = 0; w < nwords; w++)
            cm_tile[((size_t)b * 8 + w) * TILE_PIX + tid] = s_used[w][tid];
    }

    if (inside) {
        final_T[pix_id] = cT[0];
        n_contrib[pix_id] = last_contributor;
        out_color[0 * HW + pix_id] = C0 + cT[0] * bg_color[0];
        out_color[1 * HW + pix_id] = C1 + cT[0] * bg_color[1];
        out_color[2 * HW + pix_id] = C2 + cT[0] * bg_color[2];
        out_color[6 * HW + pix_id] = Cdepth;
        out_color[7 * HW + pix_id] = Calpha;
    }
    // what the point pass will walk in this tile (the deepest contributor position of its pixels): part of its dispatch cost
    {
        uint32_t m = inside ? last_contributor : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        __syncthreads();
        if (tid == 0) s_tile = 0u;
        __syncthreads();
        if (lane == 0) atomicMax(&s_tile, m);
        __syncthreads();
        if (tid == 0) tile_cost[tile] = s_tile;
    }
}

__global__ void __launch_bounds__(256)
integrate_pixels(const uint2* __restrict__ gaussian_ranges, const uint32_t* __restrict__ gaussian_list,
                 const SplatRec* __restrict__ rec, const float4* __restrict__ bbox, const float4* __restrict__ fconic, int W, int H,
                 float focal_x, float focal_y, const float* __restrict__ bg_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                 float* __restrict__ out_color, uint32_t* __restrict__ cmask, uint32_t gx, uint32_t ntiles,
                 const uint32_t* __restrict__ tile_order, uint32_t* __restrict__ tile_queue, uint32_t* __restrict__ tile_cost)
{
    __shared__ uint32_t s_tile;
    const uint32_t tile = pop_tile(tile_order, tile_queue, tile_queue + NXCD, ntiles, &s_tile);      // longest list first (gof_common.h)
    if (tile >= ntiles) return;
    integrate_pixels_tile(tile, s_tile, gaussian_ranges, gaussian_list, rec, bbox, fconic, W, H, focal_x, focal_y, bg_color, final_T, n_contrib, out_color, cmask, gx, tile_cost);
}
// one workgroup per tile in blockIdx order: only the tiles integrate_rays gave up on (tile_cost == TILE_CAPPED) are rendered
__global__ void __launch_bounds__(256)
integrate_pixels_capped(const uint2* __restrict__ gaussian_ranges, const uint32_t* __restrict__ gaussian_list,
                 const SplatRec* __restrict__ rec, const float4* __restrict__ bbox, const float4* __restrict__ fconic, int W, int H,
                 float focal_x, float focal_y, const float* __restrict__ bg_color, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                 float* __restrict__ out_color, uint32_t* __restrict__ cmask, uint32_t gx, uint32_t ntiles, uint32_t* __restrict__ tile_cost)
{
    __shared__ uint32_t s_tile;
    const uint32_t tile = blockIdx.x;
    if (tile >= ntiles || tile_cost[tile] != TILE_CAPPED) return;      // (workgroup-uniform)
    integrate_pixels_tile(tile, s_tile, gaussian_ranges, gaussian_list, rec, bbox, fconic, W, H, focal_x, focal_y, bg_color, final_T, n_contrib, out_color, cmask, gx, tile_cost);
}

// copies what the point pass needs of the geometry workspace (64-byte records, front depths) into a compact buffer
__global__ void __launch_bounds__(256)
pack_view_geometry(int P, const SplatRec* __restrict__ rec, const float4* __restrict__ fconic, SplatRec* __restrict__ rec_out, float* __restrict__ zfront_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float4* src = reinterpret_cast<const float4*>(&rec[i]);
    float4* dst = reinterpret_cast<float4*>(&rec_out[i]);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    zfront_out[i] = fconic[2 * (size_t)i + 1].w;
}

// Phase B.  base_color: the [9,H,W] image integrate_pixels wrote (channels 0-2 = the pixel colour every point of the
// pixel receives, forward.cu:1207-1208); out_color may alias it.
__global__ void __launch_bounds__(256)
integrate_points(const uint2* __restrict__ gaussian_ranges, const uint2* __restrict__ point_ranges,
                 const uint32_t* __restrict__ gaussian_list, const uint32_t* __restrict__ point_list,
                 const SplatRec* __restrict__ rec, const float* __restrict__ zfront, int zstride, const uint32_t* __restrict__ cmask, int W, int H,
                 const uint32_t* __restrict__ pt_key, const float2* __restrict__ pt_ray, const float* __restrict__ pt_depth, float* __restrict__ pt_T,
                 float* __restrict__ pt_acc, const float* __restrict__ base_color, float* __restrict__ out_color, float* __restrict__ out_alpha_integrated,
                 float* __restrict__ out_color_integrated, const uint32_t* __restrict__ n_contrib, int acc_min, uint32_t gx, uint32_t ntiles,
                 const uint32_t* __restrict__ tile_order, uint32_t* __restrict__ tile_queue)
{
    __shared__ uint32_t s_tile;
    const uint32_t tile = pop_tile(tile_order, tile_queue, tile_queue + NXCD, ntiles, &s_tile);      // most (points x walked entries) first (gof_common.h)
    if (tile >= ntiles) return;
    const uint32_t tx = tile % gx, ty = tile / gx;
    const uint32_t tid = threadIdx.x;
    uint32_t lx, ly;
    tile_pixel(tid, lx, ly);
    const uint32_t px = tx * TILE_X + lx, py = ty * TILE_Y + ly;
    const bool inside = px < (uint32_t)W && py < (uint32_t)H;
    const uint32_t pix_id = (uint32_t)W * py + px;
    const size_t HW = (size_t)W * H;

    const uint2 range = gaussian_ranges[tile];
    const uint2 prange = point_ranges[tile];
    // out_color == nullptr (accumulating calls of a mesh-extraction driver): no image is produced
    if (out_color && base_color != out_color && inside) {
#pragma unroll
        for (int c = 0; c < 8; c++) out_color[c * HW + pix_id] = base_color[c * HW + pix_id];
    }
    if (prange.y <= prange.x) {                        // no query point in this tile: channel 8 = 0
        if (out_color && inside) out_color[8 * HW + pix_id] = 0.0f;
        return;
    }
    // only list positions up to the tile's last contributor (n_contrib, written by integrate_pixels) carry mask bits -- and only
    // the batches up to there were processed by the pixel pass
    __shared__ uint32_t s_max_last;
    if (tid == 0) s_max_last = 0;
    __syncthreads();
    atomicMax(&s_max_last, inside ? n_contrib[pix_id] : 0u);
    __syncthreads();
    const int toDo0 = (int)min(range.y - range.x, s_max_last);
    const int rounds = (toDo0 + TILE_PIX - 1) / TILE_PIX;
    const int nbatches = rounds > 0 ? rounds : 1;      // a tile without Gaussians still writes alpha = 0 for its points

    __shared__ float4 s_rec[3][TILE_PIX];
    __shared__ float s_zfront[TILE_PIX];
    __shared__ uint32_t s_used[8][TILE_PIX];
    __shared__ uint32_t s_cnt[TILE_PIX];
    __shared__ uint32_t s_iter;
    const uint32_t* const cm_tile = cmask + cmask_base(range.x, tile) * TILE_PIX;

    s_cnt[tid] = 0;
    if (tid == 0) s_iter = 1;

    int toDo = toDo0;
    for (int b = 0; b < nbatches; b++, toDo -= TILE_PIX) {
        __syncthreads();   // previous batch finished with s_rec / s_used
        const uint32_t k = range.x + (uint32_t)b * TILE_PIX + tid;
        if (k < range.y) {
            const uint32_t id = gaussian_list[k];
            const float4* src = reinterpret_cast<const float4*>(&rec[id]);
            s_rec[0][tid] = src[0]; s_rec[1][tid] = src[1]; s_rec[2][tid] = src[2];
            s_zfront[tid] = zfront[(size_t)id * zstride];
        }
        const int n = toDo < 0 ? 0 : (toDo < TILE_PIX ? toDo : TILE_PIX);
        const int nwords = (n + 31) >> 5;
#pragma unroll
        for (int w = 0; w < 8; w++)
            s_used[w][tid] = (w < nwords) ? cm_tile[((size_t)b * 8 + w) * TILE_PIX + tid] : 0u;
        const bool last_batch = (b == nbatches - 1);
        __syncthreads();

        for (uint32_t base = prange.x; base < prange.y; base += TILE_PIX) {
            const uint32_t pi = base + tid;
            if (pi >= prange.y) continue;
            // list order: coalesced, L2-resident across the batches of the tile.  The ray (forward.cu:1108-1109, two fp64 divisions)
            // was formed once per call by gather_sorted_points, not here once per batch; the point's pixel of the tile is the low
            // byte of its sorted key (point_keys: tile_thread of the pixel it projects to)
            const float2 ray = pt_ray[pi];
            const float ray_depth = pt_depth[pi];
            const uint32_t lp = pt_key[pi] & 0xFFu;
            float T, acc;
            if (b == 0) { T = 1.f; acc = 0.f; atomicAdd(&s_cnt[lp], 1u); }
            else { T = pt_T[pi]; acc = pt_acc[pi]; }
            const float rx = ray.x, ry = ray.y;
            // the set bits of the pixel's 8 mask words, the wave moving from word to word together (a flat per-lane loop over all
            // words, lanes advancing independently, measured slower: 12.8 vs 12.3 ms at S5M with 45M points)
            int w = 0;
            uint32_t mask = s_used[0][lp];
            while (true) {
#ifdef GOF_STATS
                const unsigned long long with_bits = __ballot(mask != 0u);
                if (with_bits == 0ull) { if (++w >= 8) break; mask = s_used[w][lp]; continue; }
                if ((tid & 63u) == (uint32_t)__builtin_ctzll(with_bits)) ISTAT_ADD(4, 1);
#else
                if (__ballot(mask != 0u) == 0ull) { if (++w >= 8) break; mask = s_used[w][lp]; continue; }
#endif
                if (mask == 0u) continue;
                ISTAT_ADD(5, 1); ISTAT_ADD(0, 1);
                const int bit = __ffs((int)mask) - 1;
                mask &= mask - 1;
                const int j = w * 32 + bit;
                // the query point lies in front of everything this Gaussian can reach with alpha >= 1/255 (t is clamped
                // to the point's depth below): certainly skipped by the alpha test
                if (ray_depth < s_zfront[j]) { ISTAT_ADD(1, 1); continue; }
                ISTAT_ADD(2, 1);
                const float4 a4 = s_rec[0][j], b4 = s_rec[1][j], c4 = s_rec[2][j];
                const float n0 = a4.x * rx + a4.y * ry + a4.z;
                const float n1 = a4.y * rx + a4.w * ry + b4.x;
                const float n2 = a4.z * rx + b4.x * ry + b4.y;
                const float AA = rx * n0 + ry * n1 + n2;
                const float BB = 2 * (b4.z * rx + b4.w * ry + c4.x);
                const float CC = c4.y;
                float t = -BB / (2 * AA);
                if (t > ray_depth) t = ray_depth;
                const float power = -0.5f * (AA * t * t + BB * t + CC);
                const float alpha = fminf(0.99f, c4.z * gexpf(power));
                if (alpha < 1.0f / 255.0f) continue;
                ISTAT_ADD(3, 1);
                const float test_T = T * (1 - alpha);
                acc += alpha * T;
                T = test_T;
            }
            if (!last_batch) { pt_T[pi] = T; pt_acc[pi] = acc; }
            else {
                const uint32_t pid = point_list[pi];
                // acc_min: the caller's buffers hold the running minimum over the views queried so far and the colour of the view that
                // attained it -- extract_mesh.py:26-29 (final_color = where(alpha < final_alpha, color, final_color); final_alpha =
                // min(final_alpha, alpha)) fused into the store; a point outside this view keeps its values (its alpha here is 1).
                // torch.min propagates NaN, `<` does not select it: alpha becomes NaN, the colour stays.
                bool take = true, take_color = true;
                if (acc_min) {
                    const float old = out_alpha_integrated[pid];
                    take_color = acc < old;
                    take = take_color || (acc != acc);
                }
                if (take) out_alpha_integrated[pid] = acc;
                if (take_color && out_color_integrated) {
                    uint32_t plx, ply;
                    tile_pixel(lp, plx, ply);
                    const size_t ppix = (size_t)W * (ty * TILE_Y + ply) + (tx * TILE_X + plx);      // the pixel the point projects to
                    out_color_integrated[3 * (size_t)pid + 0] = base_color[0 * HW + ppix];
                    out_color_integrated[3 * (size_t)pid + 1] = base_color[1 * HW + ppix];
                    out_color_integrated[3 * (size_t)pid + 2] = base_color[2 * HW + ppix];
                }
            }
        }
    }
    __syncthreads();
    if (!out_color) return;                             // channel 8 is the only thing left to compute

    // number of outer-loop iterations the reference's block would run: max over pixels of
    // max(1, ceil(points_in_pixel / 256))
    const uint32_t m = s_cnt[tid];
    const uint32_t needed = m > (uint32_t)MAX_NUM_PROJECTED ? (m + MAX_NUM_PROJECTED - 1) / MAX_NUM_PROJECTED : 1u;
    if (inside && needed > 1) atomicMax(&s_iter, needed);
    __syncthreads();

    // The reference re-counts the tile's LAST point in its (tile, depth) order -- the deepest point, ties to the highest id --
    // once per extra outer iteration (forward.cu:1025-1096).  The list here is grouped by pixel, so find that point explicitly;
    // only tiles in which some pixel holds > 256 points get here.
    const uint32_t n_iter = s_iter;
    __shared__ unsigned long long s_deepest;
    __shared__ uint32_t s_deepest_lp;
    if (n_iter > 1) {                                   // block-uniform
        if (tid == 0) s_deepest = 0ull;
        __syncthreads();
        unsigned long long best = 0ull;
        for (uint32_t pi = prange.x + tid; pi < prange.y; pi += TILE_PIX) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(pt_depth[pi]) << 32) | point_list[pi];
            best = key > best ? key : best;
        }
        atomicMax(&s_deepest, best);
        __syncthreads();
        const unsigned long long top = s_deepest;
        for (uint32_t pi = prange.x + tid; pi < prange.y; pi += TILE_PIX) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(pt_depth[pi]) << 32) | point_list[pi];
            if (key == top) s_deepest_lp = pt_key[pi] & 0xFFu;
        }
        __syncthreads();
    }
    if (inside) {
        uint32_t total = m;
        if (n_iter > needed && s_deepest_lp == tid) total += n_iter - needed;
        out_color[8 * HW + pix_id] = (float)total;
    }
}

#ifdef GOF_STATS
extern "C" int gof_debug_int_stats(unsigned long long* out16, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_int_stats), sizeof(g_int_stats));
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_int_stats), z, sizeof(z)); }
    return 0;
}
#endif

} // namespace gof
