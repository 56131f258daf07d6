// dp_rawb.hip.h -- a batch of RawPairwiseAligner requests (praline/component/align.py:302-447 on cext_align_*,
// praline/util/cext.c:99-306): every request has its own m, g1, g2 and zero cells, so there is nothing for the lanes of a
// wave to share ACROSS requests - the parallelism is inside one request.
//
// k_rawb_fill: one workgroup per request at a time (it walks every PRALINE_RAWB_GROUPS-th request of the size-sorted list), one
// wave per strip of 64 ROWS.  Lane l owns row y0 + l and walks along it; at step
// s it computes column x = s - l, the cells of one step form an anti-diagonal:
//   - left neighbour (y, x - 1): the lane's own previous step;
//   - up neighbour (y - 1, x): lane l - 1's previous step, one `v_mov_b32_dpp wave_shr:1`; the diagonal neighbour is the up
//     neighbour of the step before;
//   - m[y][x]: a lane reads along its own row, 16 floats per 16 steps into registers, the load of the next 16 in flight
//     (columns skewed by the lane: all lanes use element i of their 16 at the same step - no LDS staging);
//   - g1[y] is a lane constant, g2[x] enters at lane 0 and moves down one lane per step (wave_shr:1 again);
//   - the tie flags of a cell are one byte (as far as the traceback reads them: see rawb_chunk); a lane stores 16 of them per
//     16 steps;
//   - zero cells: one bit per cell (k_rawb_zero);
//   - m, flags and zero bits lie in [strip][chunk][lane] order (dp_rawb.h): every wave-level access is one contiguous block.
// Lane 0 takes its up neighbour - the last row of the strip above - from an LDS ring written by lane 63 of the wave that owns
// that strip and runs >= 64 columns ahead; the waves signal progress through two LDS counters per ring, checked every 16
// steps (no barrier: a wave whose strips are done leaves).  Wave 0 is fed from memory instead, sixteen columns at a time
// through a small LDS buffer: strip 0 reads the boundary row o[0][x] (k_rawb_init), its later strips (requests of more than
// 64 x PRALINE_RAWB_WAVES rows) the row that the last wave wrote to `wrap` - a whole row, so that no ring bounds how far the
// last wave of a round may run ahead of the first wave of the next (a cycle of full rings otherwise, from ~2 600 columns on).
// Nothing but the flags, the last row and column (end cell of the global and semiglobal modes) and every lane's first maximum
// (local mode) leaves the kernel: o is never stored.  k_rawb_trace finds the end cell and walks the flags.
#pragma once
#include "dp_rawb.h"

#define RAWB_NEG_INF (-__builtin_inff())
// measurement builds only (scripts/build_variant.sh, VARIANT_RAWB=1): bit 0 no tie flags, bit 1 no waiting for the neighbour
// strips - wrong results, never in the product library
#ifndef PRALINE_RAWB_ABLATE
#define PRALINE_RAWB_ABLATE 0
#endif

__device__ __forceinline__ bool rawb_free_one(int mode) { return mode == 2 || mode == 3; }
__device__ __forceinline__ bool rawb_free_two(int mode) { return mode == 2 || mode == 4; }
// o[y,0,1] / o[0,x,2] for idx >= 1 (align.py:375,383): float64 arithmetic, one rounding
__device__ __forceinline__ float rawb_boundary(int idx, float go, float ge, bool is_free)
{
    return is_free ? 0.0f : (float)((double)(idx - 1) * (double)ge + (double)go);
}

__device__ __forceinline__ float rawb_shr1(float old, float src)   // lane l <- src of lane l - 1; lane 0 <- old
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

struct RawbRow16 { float v[16]; };

// a lane's 16 match scores of one chunk: four float4, 64 apart (one contiguous KB per wave and piece)
__device__ __forceinline__ void rawb_load16(RawbRow16 &r, const float4 *p)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 f = p[64 * q];
        r.v[4 * q] = f.x; r.v[4 * q + 1] = f.y; r.v[4 * q + 2] = f.z; r.v[4 * q + 3] = f.w;
    }
}

// wait until *flag >= need (the neighbour wave is resident: same workgroup).  A wave that has waited ~0.1 s gives up, reports
// and stops waiting for the rest of the kernel (`dead`): a broken hand-off ends in an error code, not in a hung device.
__device__ __forceinline__ void rawb_wait(volatile int *flag, int need, int32_t *error, bool &dead)
{
    if (!dead && !(PRALINE_RAWB_ABLATE & 2)) {
        int spins = 0;
        while (__builtin_amdgcn_readfirstlane(*flag) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 20)) { *error = 1; dead = true; break; }
        }
    }
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// a row entry another wave of the workgroup stored: past the vector L1 (the same addresses held the previous round's row)
__device__ __forceinline__ float4 rawb_load_fresh(const float4 *p)
{
    const unsigned *q = reinterpret_cast<const unsigned *>(p);
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = __hip_atomic_load(q + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__builtin_bit_cast(float, w[0]), __builtin_bit_cast(float, w[1]), __builtin_bit_cast(float, w[2]), __builtin_bit_cast(float, w[3]));
}

// Boundary row o[0][x] of every request (align.py:357-385), x = 0 .. L2 (+ padding the fill kernel may read)
__global__ __launch_bounds__(256) void k_rawb_init(RawBatchDev d)
{
    const RawReq rq = d.reqs[blockIdx.x];
    const float2 *g1 = d.g1 + rq.g1_off, *g2 = d.g2 + rq.g2_off;
    const bool free_one = rawb_free_one(rq.mode), free_two = rawb_free_two(rq.mode);
    float4 *top = d.top + rq.top_off;
    const float g2_00 = g2[0].x;
    for (int x = threadIdx.x; x < rq.L2 + PRALINE_RAWB_ROW_PAD; x += blockDim.x) {
        float4 v = make_float4(RAWB_NEG_INF, RAWB_NEG_INF, RAWB_NEG_INF, 0.0f);
        if (x == 0) {
            v.x = 0.0f;
            v.y = free_one ? 0.0f : g1[0].x - g1[0].y;
            v.z = free_two ? 0.0f : g2[0].x - g2[0].y;
        } else if (x <= rq.L2) {
            v.z = rawb_boundary(x, g2_00, g2[x - 1].y, free_two);
        }
        top[x] = v;
    }
}

// the caller's m (row-major, requests one after the other) -> the arena layout (dp_rawb.h): one workgroup per (strip, chunk)
// block, thread = (piece q, lane l): the four floats of row 64 k + l + 1 at columns 16 c + 4 q - l ... (zero outside the matrix)
__global__ __launch_bounds__(256) void k_rawb_stage(RawBatchDev d, const float *__restrict__ src, float4 *__restrict__ dst, const int64_t *__restrict__ block0)
{
    // block0[r]: the first workgroup of request r (requests in launch order); found by bisection
    int lo = 0, hi = d.n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (block0[mid] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1; }
    const RawReq rq = d.reqs[lo];
    const int b = (int)((int64_t)blockIdx.x - block0[lo]);     // k * ncs + c
    const int k = b / rq.ncs, c = b % rq.ncs;
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int y = 64 * k + lane;                               // 0-based row
    const int x0 = 16 * c + 4 * q - lane;                      // 0-based column of the piece's first float
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (y < rq.L1) {
        const float *sr = src + rq.src_off + (int64_t)y * rq.L2;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (x0 + e >= 0 && x0 + e < rq.L2) v[e] = sr[x0 + e];
    }
    dst[rq.m_off + (int64_t)b * 256 + q * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}

// zero cells -> bits of the skewed mask rows (cells on the boundary row / column have no effect: cext.c:141-149 starts at 1)
__global__ __launch_bounds__(256) void k_rawb_zero(RawBatchDev d, const int32_t *__restrict__ zero_req, const int32_t *__restrict__ zero_idx, int64_t n_zero)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_zero) return;
    const RawReq rq = d.reqs[zero_req[i]];
    const int y = zero_idx[2 * i], x = zero_idx[2 * i + 1];
    if (y < 1 || y > rq.L1 || x < 1 || x > rq.L2) return;
    const int p = x - 1 + ((y - 1) & 63);                      // step - 1 of the cell in its strip
    const int64_t word = rq.z_off + ((int64_t)((y - 1) >> 6) * rq.ncs + (p >> 4)) * 64 + ((y - 1) & 63);
    unsigned int *w32 = reinterpret_cast<unsigned int *>(d.z) + (word >> 1);
    atomicOr(w32, 1u << ((p & 15) + ((word & 1) ? 16 : 0)));
}

// sixteen steps of one strip.  EDGE: some lane is before its first or at / beyond its last column in these steps.  OUT: where
// the strip's last row goes - 0 nowhere, 1 the LDS ring of the wave below, 2 the `wrap` row in memory (the wave below is wave 0
// of the next round), 3 the request's last row (end cell of the global and semiglobal modes).
template <bool MASK, bool EDGE, bool LOCAL, int OUT>
__device__ __forceinline__ void rawb_chunk(const RawbRow16 &mrow, unsigned zbits, const float2 g2c, int c, int lane, int L2, int y, bool row_ok,
                                           bool last_strip, bool feeds, int L1, float base, float go1, float ge1, float bndU, const float4 *hand /* LDS, entry of step 0 */,
                                           int hand_mask, int hand_pos, float4 *out_ring, int out_pos, float4 *out_row /* or NULL */, float4 *edge_row, float4 *edge_col,
                                           float &curM, float &curU, float &curL, float &upM, float &upU, float &upL, float &go2, float &ge2,
                                           float &sbest, int &scode, float &sM, float &sU, uint4 &flags)
{
    unsigned pk[4] = {0u, 0u, 0u, 0u};
    float4 hn = hand[hand_pos & hand_mask];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int s = 16 * c + 1 + i;
        const int x = s - lane;
        // lane 0's up neighbour: the row above the strip, read one step ahead
        __asm__ volatile("" ::: "memory");   // (one step's LDS read and stores at a time: hoisted, the sixteen reads alone hold 64 registers)
        const float4 h = hn;
        if (i < 15) hn = hand[(hand_pos + i + 1) & hand_mask];
        const float nuM = rawb_shr1(h.x, curM), nuU = rawb_shr1(h.y, curU), nuL = rawb_shr1(h.z, curL);
        const float sgo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g2c.x), i));
        const float sge = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g2c.y), i));
        go2 = rawb_shr1(sgo, go2);
        ge2 = rawb_shr1(sge, ge2);
        const float ms = mrow.v[i];
        // cext.c:141-183
        const float mm = upM + ms, mu = upU + ms, ml = upL + ms;   // (upM/U/L still hold the step before: the diagonal)
        float mmax = LOCAL ? __builtin_fmaxf(__builtin_fmaxf(mm, mu), __builtin_fmaxf(ml, base)) : __builtin_fmaxf(__builtin_fmaxf(mm, mu), ml);
        const float up_open = nuM + go1, up_ext = nuU + ge1;
        float umax = __builtin_fmaxf(up_open, up_ext);
        const float lf_open = curM + go2, lf_ext = curL + ge2;
        float lmax = __builtin_fmaxf(lf_open, lf_ext);
        // The reference's flag byte (cext.c:155-183; here shifted right by one: bits 6 .. 0 = lf_ext, lf_open, up_ext, up_open, ml,
        // mu, mm) - as far as the traceback reads it: it takes the FIRST set flag of a state's group (praline/util/align.py:155-183),
        // so the last flag of every group may be set unconditionally (one of up_open / up_ext equals their maximum: if not the
        // first, then the second; the same for lf_* and - without the local mode's zero - for mm / mu / ml).  Four compares instead
        // of seven (five in local mode), the same paths.
        unsigned f = LOCAL ? 0x50u : 0x54u;
        if (!(PRALINE_RAWB_ABLATE & 1)) {
            f |= (mm == mmax ? 1u : 0u) | (mu == mmax ? 2u : 0u) | (up_open >= up_ext ? 8u : 0u) | (lf_open >= lf_ext ? 32u : 0u);
            if (LOCAL) f |= ml == mmax ? 4u : 0u;
        }
        if (MASK) {
            const bool zc = (zbits >> i) & 1u;   // a zero cell keeps what the caller initialised: zeros (align.py:362-367, cext.c:147-149)
            mmax = zc ? 0.0f : mmax; umax = zc ? 0.0f : umax; lmax = zc ? 0.0f : lmax;
            f = zc ? 0u : f;
        }
        if (EDGE && x < 1) { mmax = RAWB_NEG_INF; umax = bndU; lmax = RAWB_NEG_INF; }   // not started: the lane shows its boundary cell (y, 0)
        __asm__ volatile("" : "+v"(f));          // (the byte first, then its place in the word: folded into the selects, the shifted flag constants hold 20 registers)
        pk[i >> 2] |= f << (8 * (i & 3) + 1);
        __asm__ volatile("" : "+v"(pk[i >> 2]));   // (the flags of a step are formed in that step: deferred to the end of the chunk, their inputs spill)
        const bool in_row = !EDGE || (x >= 1 && x <= L2);
        if (LOCAL) {
            const float v3 = __builtin_fmaxf(__builtin_fmaxf(mmax, umax), lmax);
            const bool gt = in_row && v3 > sbest;
            sbest = gt ? v3 : sbest; scode = gt ? s : scode; sM = gt ? mmax : sM; sU = gt ? umax : sU;
            __asm__ volatile("" : "+v"(sbest), "+v"(scode), "+v"(sM), "+v"(sU));   // (decided in this step: deferred, the values of sixteen steps stay live)
        } else {
            if (OUT == 3 && y == L1 && in_row) edge_row[x] = make_float4(mmax, umax, lmax, 0.0f);
            if (EDGE && x == L2 && row_ok) edge_col[y] = make_float4(mmax, umax, lmax, 0.0f);
        }
        if (OUT == 1 && lane == 63 && in_row) out_ring[(out_pos + x - 1) & (PRALINE_RAWB_RING - 1)] = make_float4(mmax, umax, lmax, 0.0f);
        if (OUT == 2 && lane == 63 && in_row) out_row[x] = make_float4(mmax, umax, lmax, 0.0f);
        upM = nuM; upU = nuU; upL = nuL;
        curM = mmax; curU = umax; curL = lmax;
    }
    flags = make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

template <bool MASK, bool LOCAL>
__global__ __launch_bounds__(64 * PRALINE_RAWB_WAVES) __attribute__((amdgpu_waves_per_eu(4))) void k_rawb_fill(RawBatchDev d)
{
    __shared__ float4 ring[PRALINE_RAWB_WAVES][PRALINE_RAWB_RING];
    __shared__ float4 topbuf[32];                         // strip 0: the boundary row, two halves of 16 columns
    __shared__ int produced[PRALINE_RAWB_WAVES], consumed[PRALINE_RAWB_WAVES];
    // (the wave index as a scalar: everything derived from it - strips, neighbours, hand-off conditions - is uniform, and the
    // compiler can only branch on it with scalar instructions if it knows)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
    if (lane == 0) { produced[wave] = 0; consumed[wave] = 0; }
    __syncthreads();
    // A workgroup takes every gridDim.x-th request of the (size-sorted) list, one after the other: wave w goes from strip w of
    // one request straight to strip w of the next, so the start-up skew of the waves (64 + 16 columns each) is paid once per
    // workgroup, not once per request.  The hand-off streams simply continue: out_pos / in_pos count the columns a wave has
    // handed down / taken over since the kernel started, in the order both sides visit the strips.
    const int prod = (wave + W - 1) % W;
    bool dead = false;
    int out_pos = 0, in_pos = 0;
    for (int q = blockIdx.x; q < d.n; q += gridDim.x) {
    const RawReq rq = d.reqs[q];
    if ((rq.mode == 1) != LOCAL) continue;                // (the other instance's request)
    const int L1 = rq.L1, L2 = rq.L2, R = rq.nstrips;
    const bool local = LOCAL, free_one = rawb_free_one(rq.mode);
    const float base = local ? 0.0f : RAWB_NEG_INF;
    const float2 *g1 = d.g1 + rq.g1_off, *g2 = d.g2 + rq.g2_off;
    const float4 *top = d.top + rq.top_off;
    float4 *wrap = d.wrap + rq.top_off;                   // (same shape as the boundary row)
    float4 *edge_row = d.edge + rq.edge_off, *edge_col = edge_row + (L2 + 1);
    const float g1_00 = g1[0].x;
    const int NC = (L2 + 63 + 15) / 16;
    float bestv = RAWB_NEG_INF;
    int best_y = 0, best_x = 0, best_k = 0;
    for (int k = wave; k < R; k += W) {
        const bool feeds = k + 1 < R, fed = k > 0, last_strip = k == R - 1;
        const bool from_row = wave == 0;                  // the row above comes from memory: boundary row or `wrap`
        float4 *out_row = feeds && wave == W - 1 ? wrap : nullptr;
        const int out_kind = feeds ? (out_row ? 2 : 1) : (LOCAL ? 0 : 3);
        const float4 *feed = k == 0 ? top : wrap;
        const int y = 64 * k + 1 + lane;
        const bool row_ok = y <= L1;
        const int yc = row_ok ? y : L1;
        const float2 gv = g1[yc - 1];
        const float go1 = gv.x, ge1 = gv.y;
        const float bndU = rawb_boundary(yc, g1_00, ge1, free_one);
        float curM = RAWB_NEG_INF, curU = bndU, curL = RAWB_NEG_INF;
        // lane 0's first diagonal neighbour: the boundary cell of the row above the strip
        float upM = RAWB_NEG_INF, upU = RAWB_NEG_INF, upL = RAWB_NEG_INF;
        if (k == 0) { const float4 t0 = top[0]; upM = t0.x; upU = t0.y; upL = t0.z; }
        else upU = rawb_boundary(64 * k, g1_00, g1[64 * k - 1].y, free_one);
        float go2 = 0.0f, ge2 = 0.0f;
        // the strip's blocks in the three chunk-ordered arenas (dp_rawb.h)
        const float4 *mrow = reinterpret_cast<const float4 *>(d.m) + rq.m_off + (int64_t)k * rq.ncs * 256 + lane;
        uint4 *trow = reinterpret_cast<uint4 *>(d.t) + rq.t_off + (int64_t)k * rq.ncs * 64 + lane;
        const uint16_t *zrow = d.z + rq.z_off + (int64_t)k * rq.ncs * 64 + lane;
        float sbest = RAWB_NEG_INF, sM = 0.0f, sU = 0.0f;
        int scode = 0;
        RawbRow16 mA, mB;
        unsigned zA = 0, zB = 0;
        float2 gA, gB;
        float4 treg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        rawb_load16(mA, mrow);
        if (MASK) zA = zrow[0];
        gA = g2[lane & 15];
        if (from_row) {
            if (fed) rawb_wait(&produced[prod], in_pos + min(32, L2), d.error, dead);
            if (lane < 16) topbuf[lane] = rawb_load_fresh(feed + 1 + lane);
            treg = rawb_load_fresh(feed + 17 + (lane & 15));
        }
        const float4 *hand = from_row ? topbuf : ring[prod];
        const int hand_mask = from_row ? 31 : PRALINE_RAWB_RING - 1;
        const int hand_base = from_row ? 0 : in_pos;      // position of column 1 in the hand-off stream
        const int out_base = out_pos;
        for (int c = 0; c < NC; ++c) {
            // the neighbours: the strip above has produced this chunk's columns (wave 0 reads the row two chunks ahead); the strip
            // below has read what this chunk overwrites
            if (fed) rawb_wait(&produced[prod], in_pos + min(16 * c + (from_row ? 48 : 16), L2), d.error, dead);
            if (feeds && !out_row) rawb_wait(&consumed[wave], out_base + min(max(16 * c + 16 - 63, 0), L2) - PRALINE_RAWB_RING, d.error, dead);
            // the next chunk's inputs
            rawb_load16(mB, mrow + 256 * (c + 1));
            if (MASK) zB = zrow[64 * (c + 1)];
            gB = g2[16 * (c + 1) + (lane & 15)];
            if (from_row) {
                if (lane < 16) topbuf[16 * ((c + 1) & 1) + lane] = treg;
                treg = rawb_load_fresh(feed + 16 * (c + 2) + 1 + (lane & 15));
            }
            uint4 fl;
            const bool edge = 16 * c < 63 || 16 * c + 16 >= L2;
#define RAWB_CHUNK(E, O)                                                                                                                        \
    rawb_chunk<MASK, E, LOCAL, O>(mA, zA, gA, c, lane, L2, y, row_ok, last_strip, feeds, L1, base, go1, ge1, bndU, hand, hand_mask, hand_base + 16 * c,  \
                           ring[wave], out_base, out_row, edge_row, edge_col, curM, curU, curL, upM, upU, upL, go2, ge2, sbest, scode, sM, sU, fl)
            if (out_kind == 1) { if (edge) RAWB_CHUNK(true, 1); else RAWB_CHUNK(false, 1); }
            else if (out_kind == 2) { if (edge) RAWB_CHUNK(true, 2); else RAWB_CHUNK(false, 2); }
            else if (out_kind == 3) { if (edge) RAWB_CHUNK(true, 3); else RAWB_CHUNK(false, 3); }
            else { if (edge) RAWB_CHUNK(true, 0); else RAWB_CHUNK(false, 0); }
#undef RAWB_CHUNK
            trow[64 * c] = fl;
            // progress: this strip's last row up to column 16 c + 16 - 63, the row above read up to column 16 c + 16
            if (out_row) __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the row entries have reached the L2)
            else __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
                if (feeds) *(volatile int *)&produced[wave] = out_base + min(max(16 * c + 16 - 63, 0), L2);
                if (fed) *(volatile int *)&consumed[prod] = in_pos + min(16 * c + 16, L2);
            }
            mA = mB; zA = zB; gA = gB;
        }
        if (feeds) out_pos += L2;
        if (fed) in_pos += L2;
        if (local && row_ok && sbest > bestv) {   // rows ascend with the strips: the first maximum of the lane stays
            bestv = sbest; best_y = y; best_x = scode - lane;
            best_k = sM == sbest ? 0 : (sU == sbest ? 1 : 2);
        }
    }
    if (local && wave < R)
        d.best[rq.best_off + wave * 64 + lane] = make_float4(bestv, __builtin_bit_cast(float, best_y), __builtin_bit_cast(float, best_x), __builtin_bit_cast(float, best_k));
    }
}

// wave-wide reductions of the end-cell search (all lanes receive the result)
__device__ __forceinline__ float rawb_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int rawb_wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ unsigned long long rawb_wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}

// End cell (align.py:401-431), score and path (praline/util/align.py:144-185, 268-297) of every request: one wave each.  The
// lanes search the last row / column (or the lanes' first maxima and the boundary cells, local mode) together; the walk
// itself is serial, so the wave fetches the flags it is about to need as a tile - 64 rows x 128 columns ending at the current
// cell, one row per lane - into LDS and walks inside it until the path leaves it (a step is an LDS read instead of a dependent
// read of memory: ~15 tile fetches instead of ~800 memory round trips for a 400 x 400 alignment).
#define RAWB_TILE_RS 176   // bytes per tile row in LDS: 16 of headroom for the dword shift, 36 dwords, 16 spare (and rows 44 banks apart)
__global__ __launch_bounds__(256) void k_rawb_trace(RawBatchDev d)
{
    __shared__ uint4 tiles[4][64 * RAWB_TILE_RS / 16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the walk is uniform)
    const int r = blockIdx.x * 4 + wv;
    if (r >= d.n) return;
    const RawReq rq = d.reqs[r];
    const int L1 = rq.L1, L2 = rq.L2, mode = rq.mode;
    const bool free_one = rawb_free_one(mode), free_two = rawb_free_two(mode);
    const bool semiglobal = mode >= 2;
    const float2 *g1 = d.g1 + rq.g1_off;
    const float4 *top = d.top + rq.top_off;
    const float4 *edge_row = d.edge + rq.edge_off, *edge_col = edge_row + (L2 + 1);
    const float g1_00 = g1[0].x;
    auto row_at = [&](int x) {   // o[L1][x]
        return x == 0 ? make_float4(RAWB_NEG_INF, rawb_boundary(L1, g1_00, g1[L1 - 1].y, free_one), RAWB_NEG_INF, 0.0f) : edge_row[x];
    };
    auto col_at = [&](int y) { return y == 0 ? top[L2] : edge_col[y]; };   // o[y][L2]
    auto pick = [](const float4 &v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); };
    auto max3 = [](const float4 &v) { return __builtin_fmaxf(v.x, __builtin_fmaxf(v.y, v.z)); };
    int cy = L1, cx = L2, ck = 0;
    float score;
    if (mode == 1) {
        // first flat argmax of o (align.py:402): the largest value, among equals the smallest (y, x, k) - over the boundary row,
        // the boundary column and the lanes' first maxima of their rows
        float bv = RAWB_NEG_INF;
        unsigned long long bkey = ~0ull;
        auto offer = [&](float v, int y, int x, int k) {
            const unsigned long long key = ((unsigned long long)(unsigned)y << 34) | ((unsigned long long)(unsigned)x << 2) | (unsigned)k;
            if (v > bv || (v == bv && key < bkey)) { bv = v; bkey = key; }
        };
        for (int x = lane; x <= L2; x += 64) { const float4 v = top[x]; offer(v.x, 0, x, 0); offer(v.y, 0, x, 1); offer(v.z, 0, x, 2); }
        for (int y = 1 + lane; y <= L1; y += 64) offer(rawb_boundary(y, g1_00, g1[y - 1].y, free_one), y, 0, 1);
        const int nrec = min(rq.nstrips, PRALINE_RAWB_WAVES) * 64;
        for (int q = lane; q < nrec; q += 64) {
            const float4 b = d.best[rq.best_off + q];
            const int y = __builtin_bit_cast(int, b.y);
            if (y >= 1) offer(b.x, y, __builtin_bit_cast(int, b.z), __builtin_bit_cast(int, b.w));
        }
        score = rawb_wave_max(bv);
        const unsigned long long key = rawb_wave_min_u64(bv == score ? bkey : ~0ull);
        cy = (int)(key >> 34); cx = (int)((key >> 2) & 0xffffffffull); ck = (int)(key & 3);
    } else if (mode == 0) {
        const float4 q = row_at(L2);
        ck = 0;
        if (q.y > pick(q, ck)) ck = 1;
        if (q.z > pick(q, ck)) ck = 2;
        score = pick(q, ck);
    } else {
        float rmax = RAWB_NEG_INF, cmax = RAWB_NEG_INF;
        for (int x = lane; x <= L2; x += 64) rmax = __builtin_fmaxf(rmax, max3(row_at(x)));
        for (int y = lane; y <= L1; y += 64) cmax = __builtin_fmaxf(cmax, max3(col_at(y)));
        rmax = rawb_wave_max(rmax); cmax = rawb_wave_max(cmax);
        // the LARGEST coordinate that holds the maximum, there the smallest k (align.py:414-423)
        if (rmax > cmax && free_two) {
            int bx = -1;
            for (int x = lane; x <= L2; x += 64) { const float4 v = row_at(x); if (v.x == rmax || v.y == rmax || v.z == rmax) bx = x; }
            cx = rawb_wave_max_i(bx); cy = L1;
            const float4 v = row_at(cx);
            ck = v.x == rmax ? 0 : (v.y == rmax ? 1 : 2);
            score = rmax;
        } else {
            int by = -1;
            for (int y = lane; y <= L1; y += 64) { const float4 v = col_at(y); if (v.x == cmax || v.y == cmax || v.z == cmax) by = y; }
            cy = rawb_wave_max_i(by); cx = L2;
            const float4 v = col_at(cy);
            ck = v.x == cmax ? 0 : (v.y == cmax ? 1 : 2);
            score = cmax;
        }
    }
    cy = __builtin_amdgcn_readfirstlane(cy); cx = __builtin_amdgcn_readfirstlane(cx); ck = __builtin_amdgcn_readfirstlane(ck);
    if (lane == 0) d.scores[rq.index] = score;
    const int cap = L1 + L2 + 2;
    int32_t *path = d.paths + 2 * rq.path_off;
    // path rows are written back to front; they are collected in two registers (row j since the last flush in lane j) and
    // stored 64 at a time
    int w = cap, w_flushed = cap, held = 0, hy = 0, hx = 0;
    auto flush = [&]() {
        if (lane < held) { const int at = w_flushed - 1 - lane; path[2 * at] = hy; path[2 * at + 1] = hx; }
        w_flushed = w; held = 0;
    };
    auto emit = [&](int yy, int xx) {
        hy = lane == held ? yy : hy; hx = lane == held ? xx : hx;
        --w; ++held;
        if (held == 64) flush();
    };
    int y = cy, x = cx, k = ck;
    if (semiglobal) {
        if (y != L1) { for (int yy = L1; yy > y; --yy) emit(yy, x); }
        else if (x != L2) { for (int xx = L2; xx > x; --xx) emit(y, xx); }
    }
    emit(y, x);
    const uint4 *t = reinterpret_cast<const uint4 *>(d.t) + rq.t_off;
    // A tile: rows ty - 63 .. ty, columns xw .. xw + 127 with xw = max(1, tx - 111), fetched at the cell (ty, tx).  Lane i
    // loads the nine 16-byte pieces of row ty - i that hold those columns (a row's pieces start at step boundaries, i.e. at
    // x - 1 + lane of the row: a different byte offset per row), shifts them so that column xw comes first - the bytes with
    // v_alignbyte, the dwords through the LDS address - and the walk's address is (rows up) * RAWB_TILE_RS + (x - xw):
    // + RAWB_TILE_RS for a step up, - 1 for a step left.
    uint8_t *tile = reinterpret_cast<uint8_t *>(&tiles[wv][0]);
    int ty = -1, tx = 0, xw = 0;
    int guard = 0;
    bool done = false;
    while (!done && guard < cap) {
        if (y == 0 || x == 0) {
            // on the boundary row / column: t[1:,0,1] = insert-up-extend (align.py:377), t[0,1:,2] = insert-left-extend (align.py:385)
            unsigned f = 0;
            if (y == 0 && x == 0) f = 0;
            else if (x == 0) f = (k == 1 && !free_one) ? 32u : 0u;
            else f = (k == 2 && !free_two) ? 128u : 0u;
            if (f == 0) break;
            if (f == 32u) --y; else --x;
            emit(y, x);
            ++guard;
            continue;
        }
        if (ty < 0 || ty - y > 63 || x < xw) {
            ty = y; tx = x; xw = max(1, tx - 111);
            const int yy = ty - lane;
            unsigned wd[37];
#pragma unroll
            for (int e = 0; e < 37; ++e) wd[e] = 0u;
            int a = 0;
            if (yy >= 1) {
                const int p0 = (xw - 1) + ((yy - 1) & 63), c0 = p0 >> 4;
                a = p0 & 15;
                // (a strip of a narrow request has fewer chunk blocks than a tile is wide: never beyond them - behind the last
                // strip of the last request the arena ends)
                const uint4 *src = t + ((int64_t)((yy - 1) >> 6) * rq.ncs + c0) * 64 + ((yy - 1) & 63);
#pragma unroll
                for (int q = 0; q < 9; ++q)
                    if (c0 + q < rq.ncs) { const uint4 v = src[64 * q]; wd[4 * q] = v.x; wd[4 * q + 1] = v.y; wd[4 * q + 2] = v.z; wd[4 * q + 3] = v.w; }
            }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the reads of the tile this one replaces)
            // bytes a & 3 out with v_alignbyte, dwords a >> 2 out through the address: byte (x - xw) of the row lands at offset x - xw
            unsigned *row = reinterpret_cast<unsigned *>(tile + lane * RAWB_TILE_RS + 16) - (a >> 2);
#pragma unroll
            for (int e = 0; e < 36; ++e) row[e] = __builtin_amdgcn_alignbyte(wd[e + 1], wd[e], (unsigned)(a & 3));
            __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        // inside the tile
        int up = ty - y;
        int addr = up * RAWB_TILE_RS + 16 + (x - xw);
        while (true) {
            unsigned f = __builtin_amdgcn_readfirstlane((unsigned)tile[addr]);
            f &= (0xc0300eu >> (8 * k)) & 0xffu;
            if (f == 0) { done = true; break; }
            // the lowest set flag decides (praline/util/align.py:155-183): match from M / U / L (bits 1-3: up-left, next state
            // 0 / 1 / 2), insert-up open / extend (bits 4, 5: up, state 0 / 1), insert-left open / extend (bits 6, 7: left, state 0 / 2)
            const int bit = __builtin_ctz(f) - 1;
            const int dy = (0x1f >> bit) & 1, dx = (0x67 >> bit) & 1;
            k = (0x2124 >> (2 * bit)) & 3;
            y -= dy; x -= dx; up += dy;
            addr += dy * RAWB_TILE_RS - dx;
            emit(y, x);
            if (++guard >= cap || y == 0 || x < xw || up > 63) break;   // (x < xw covers x == 0: xw >= 1)
        }
    }
    if (semiglobal) {
        if (y != 0) { for (int yy = y - 1; yy >= 0; --yy) emit(yy, 0); }
        else if (x != 0) { for (int xx = x - 1; xx >= 0; --xx) emit(0, xx); }
    }
    flush();
    if (lane == 0) {
        d.path_info[2 * (int64_t)rq.index] = w;
        d.path_info[2 * (int64_t)rq.index + 1] = cap - w;
    }
}
