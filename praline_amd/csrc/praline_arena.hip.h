// praline_arena.hip.h -- part of praline_dp.hip (one translation unit; included there, in this order): the profile arena: creation (whole or in parts), packing and pre-multiply launches, track sets, per-position gap scores,
// reference-order tables, resident progressive alignment (clusters merged on the device and appended in place).
// --------------------------------------------------------------------------------------------
// arena
// --------------------------------------------------------------------------------------------
static const int kNstepChoices[] = {2, 8, 10, 12, 14, 16};

struct praline_arena {
    int64_t n_seqs = 0;
    int A = 0;               // alphabet size of the raw profiles
    int n_active = 0;        // symbols that can contribute to a match score
    int nstep = 0;           // MFMA steps per tile (template instance)
    int KP = 0, KS = 0;
    int64_t rows_raw = 0, rows_pad = 0;
    int max_len = 0;
    std::vector<int32_t> len, row_off_pad, row_off_raw, active;
    // host sources of the creation's asynchronous uploads (kept: praline_arena_create does not wait for them)
    bool building = false;       // between praline_arena_begin and praline_arena_finish: only praline_arena_put_rows may touch it
    unsigned inexact_bits = 0;   // some value of the first rows is not a normal float16 (praline_arena_put_rows)
    std::vector<float> h_S;
    std::vector<int32_t> h_seq_of_rowp, h_active_up;
    std::vector<unsigned char> h_slot_of;
    DevBuf<float> d_raw, d_S, d_P, d_Q;
    DevBuf<int32_t> d_len, d_row_off_pad, d_row_off_raw, d_seq_of_rowp, d_active;
    // f16 split operands for k_dp_split16 (matrix-pipe MFMA)
    int nr16 = 0;          // 16-wide k ranges (1 or 2); 0 = not available (> 32 active symbols)
    int nterm16 = 3;       // 1: every operand is exactly representable in f16, 3: hi/lo split in six MFMAs per step (NR = 2),
                           // 2: the same three terms K-packed into four MFMAs (at most 21 active symbols; dp_kernels.hip.h)
    DevBuf<char> d_P16, d_Q16;
    DevBuf<int> d_flag16;
    // one-hot arenas (ordinary sequences): active-symbol index per padded row; see k_dp_split16<.., ONEHOT>
    bool onehot = false;       // one-hot operand table in use
    bool all_onehot = false;   // every profile row is one-hot (plain sequences): required by the preprofile counting
    int s_scale_bits = -1;     // smallest k <= 8 with S * 2^k integral in every entry (-1: none); s_absmax = max |S|
    float s_absmax = 0.0f;
    DevBuf<unsigned char> d_sym8;
    // preprofile stage (k_path_counts): raw symbol of every one-hot row (255: not one-hot), int32 counts [rows_raw][A]
    DevBuf<unsigned char> d_sym_raw;
    DevBuf<int32_t> d_counts;
    int32_t *counts_ext = nullptr;   // caller-owned count buffer (praline_arena_counts_bind)
    int32_t *counts_ptr() const { return counts_ext ? counts_ext : d_counts.p; }
    // reference-order audit mode (k_match_ref): track-set partition of the alphabet axis and per-row nonzero lists
    std::vector<int32_t> set_lo;     // n_sets + 1 boundaries, default {0, A}
    DevBuf<int32_t> d_set_lo;
    DevBuf<unsigned char> d_nzidx, d_nzcnt;
    DevBuf<float> d_reft;    // T[row][i][b] (k_build_reft), ref_tb floats per (row, symbol); ref_tb = 0: not built
    int ref_tb = 0;          // nonzeros per row, rounded up to 4 / 8 / 16 / 32 (0: more)
    int reft_state = 0;      // d_reft: 0 not tried, 1 built, -1 not available (too large / too many nonzeros)
    bool ref_ready = false;
    // the same half-terms with two adjacent columns interleaved, for k_match_tile (dp_reftile.hip.h): T2[i][pair row][b][2]
    DevBuf<float> d_reft2;
    DevBuf<int64_t> d_pr_off;   // first pair row of every sequence
    int64_t pair_rows = 0;
    int reft2_state = 0;        // 0: not tried, 1: built, -1: not available for this arena (alphabet / row density)
    // resident progressive alignment (praline_arena_append_merged): integer counts of every row, capacities
    DevBuf<int32_t> d_cnt;
    bool have_cnt = false;
    int64_t cap_rows_raw = 0, cap_rows_pad = 0, cap_seqs = 0;   // 0: the buffers hold exactly what is in use
    int64_t rp_end = 0;       // padded rows taken by sequences (the zero tail follows)
    bool wide = false;       // more than 32 active symbols: no MFMA operand layouts; every plan runs the reference-order path
    // per-position gap scores (praline_arena_set_gap_scores): (open, extend) per padded row; plans created while they
    // are set read their match scores from dense tiles and can run with them (praline_plan_run_gaps)
    DevBuf<float> d_gaps;
    bool has_gaps = false;
    Arena16Dev view16() const
    {
        Arena16Dev v;
        v.sym8 = (onehot && nterm16 == 1) ? d_sym8.p : nullptr;
        // the staged stream addresses the arena with 32-bit lane offsets
        const char *ns = getenv("PRALINE_NO_STAGE");
        v.stage = (!(ns && ns[0] == '1') && (uint64_t)rows_pad * 64 * nr16 < 0xffff0000ull) ? 1 : 0;
        v.P16 = d_P16.p; v.Q16 = d_Q16.p; v.row_off = d_row_off_pad.p; v.len = d_len.p;
        v.half_bytes = 2 * nr16 * 16; v.row_bytes = 2 * v.half_bytes;
        return v;
    }
    ArenaDev view() const
    {
        ArenaDev v;
        v.P = d_P.p; v.Q = d_Q.p; v.row_off = d_row_off_pad.p; v.len = d_len.p; v.KP = KP; v.KS = KS;
        return v;
    }
};

// Every arena entry point but praline_arena_put_rows / _finish / _destroy goes through this: an arena between
// praline_arena_begin and praline_arena_finish has no tables, no operands and no lengths on the device yet.
static int arena_ready(const praline_arena *a)
{
    if (!a) return fail(PRALINE_ERR_ARG, "arena is NULL");
    if (a->building) return fail(PRALINE_ERR_ARG, "the arena is still being built (praline_arena_finish)");
    return PRALINE_OK;
}

static int arena_launch_premultiply(praline_arena *a, bool check_f16 = false)
{
    if (a->wide) return PRALINE_OK;   // no packed operands: plans on this arena read the raw profiles (k_match_ref)
    if (!check_f16) {   // the recurring call: everything in one launch
        hipLaunchKernelGGL(k_prepare_rows, dim3((unsigned)(a->rows_pad / 32)), dim3(256), 0, g_rt.stream, a->d_raw.p, a->d_S.p,
                           a->d_seq_of_rowp.p, a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p, a->d_active.p, a->n_active,
                           a->A, a->KP, a->KS, a->rows_pad, a->d_P.p, a->d_Q.p, a->nr16, (_Float16 *)a->d_P16.p,
                           (_Float16 *)a->d_Q16.p, (int64_t)0, a->nterm16 == 2 ? 1 : 0);
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }
    const int64_t total = a->rows_pad * a->KP;
    const int threads = 256;
    const int64_t blocks = (total + threads - 1) / threads;
    hipLaunchKernelGGL(k_pack_profiles, dim3((unsigned)blocks), dim3(threads), 0, g_rt.stream, a->d_raw.p,
                       a->d_seq_of_rowp.p, a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p,
                       a->d_active.p, a->n_active, a->A, a->KP, a->KS, a->rows_pad, a->d_P.p);
    dim3 grid((unsigned)((a->rows_pad + 31) / 32), (unsigned)((a->KP + 31) / 32));
    hipLaunchKernelGGL(k_premultiply, grid, dim3(64), 0, g_rt.stream, a->d_raw.p, a->d_S.p,
                       a->d_seq_of_rowp.p, a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p,
                       a->d_active.p, a->n_active, a->A, a->KP, a->KS, a->rows_pad, a->d_Q.p);
    if (a->nr16 > 0) {
        int *flag = check_f16 ? a->d_flag16.p : nullptr;
        if (check_f16) HIPCHK(hipMemsetAsync(a->d_flag16.p, 0, sizeof(int), g_rt.stream));
        praline_launch_split_f16(a->d_P.p, a->KP, a->KS, a->n_active, a->nr16, a->rows_pad, a->d_P16.p, flag, g_rt.stream);
        praline_launch_split_f16(a->d_Q.p, a->KP, a->KS, a->n_active, a->nr16, a->rows_pad, a->d_Q16.p, flag, g_rt.stream);
    }
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

// Arena creation in three steps (praline_arena_create is begin + one put + finish): begin sizes the arena and allocates
// the raw rows, put uploads a range of rows (asynchronously: a caller that concatenates per-sequence arrays into
// page-locked staging uploads the first half while it copies the second), finish scans, packs and pre-multiplies.
static int arena_begin(int64_t n_seqs, const int32_t *lens, int32_t A, praline_arena **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_seqs <= 0 || !lens) return fail(PRALINE_ERR_ARG, "NULL or empty arena input");
    // the raw (concatenated) alphabet may be wide; what the kernels bound is the number of ACTIVE symbols (<= 32)
    if (A <= 0 || A > 254) return fail(PRALINE_ERR_ARG, "alphabet size %d not in 1..254 (concatenated track sets)", A);
    RC(ensure_runtime(-1));
    praline_arena *a = new praline_arena();
    a->n_seqs = n_seqs;
    a->A = A;
    a->set_lo = {0, A};
    a->len.assign(lens, lens + n_seqs);
    a->row_off_pad.resize(n_seqs);
    a->row_off_raw.resize(n_seqs);
    int64_t rr = 0, rp = 0;
    for (int64_t s = 0; s < n_seqs; ++s) {
        if (lens[s] <= 0) { delete a; return fail(PRALINE_ERR_ARG, "sequence %lld has length %d (must be >= 1)", (long long)s, lens[s]); }
        a->row_off_raw[s] = (int32_t)rr;
        a->row_off_pad[s] = (int32_t)rp;
        rr += lens[s];
        rp += (lens[s] + 31) / 32 * 32;
        a->max_len = std::max(a->max_len, lens[s]);
        if (rp > (int64_t)1 << 30) { delete a; return fail(PRALINE_ERR_ARG, "arena too large"); }
    }
    a->rows_raw = rr;
    a->rp_end = rp;
    // tail padding: the kernels prefetch one row past the longest sequence and read whole strips
    a->rows_pad = rp + (a->max_len + 31) / 32 * 32 + 64;
    int rc0 = PRALINE_OK;
    if ((rc0 = a->d_raw.alloc((size_t)rr * A)) || (rc0 = a->d_sym_raw.alloc((size_t)rr))) { delete a; return rc0; }
    a->building = true;
    *out = a;
    return PRALINE_OK;
}

static int arena_put(praline_arena *a, int64_t row0, int64_t n_rows, const float *rows)
{
    if (!a || !a->building) return fail(PRALINE_ERR_ARG, "rows can only be put into an arena between begin and finish");
    if (!rows || row0 < 0 || n_rows < 0 || row0 + n_rows > a->rows_raw) return fail(PRALINE_ERR_ARG, "row range %lld + %lld outside the arena's %lld rows", (long long)row0, (long long)n_rows, (long long)a->rows_raw);
    if (row0 == 0) {
        // ... whether some value is NOT a normal float16 (13 low mantissa bits set, or an exponent outside -14 .. 15): such
        // an arena needs the hi/lo split whatever S is, so the device-side exactness check (and the second packing launch
        // that follows its read-back) can be skipped - float profiles, i.e. every preprofile / profile-profile stage
        // (looked for in the first 64 K values only: float profiles show one in their first rows, and arenas that show none
        // there keep the device-side check)
        unsigned inexact_bits = 0;
        for (int64_t k = 0, n = std::min<int64_t>(n_rows * a->A, 65536); k < n && !inexact_bits; ++k) {
            unsigned u;
            memcpy(&u, &rows[k], 4);
            const unsigned e = (u >> 23) & 0xffu;
            if (u & 0x7fffffffu) inexact_bits = (u & 0x1fffu) | (unsigned)(e < 113u) | (unsigned)(e > 142u);
        }
        a->inexact_bits = inexact_bits;
    }
    // (a DMA when the caller's buffer is page-locked: praline_host_alloc)
    if (n_rows > 0) HIPCHK(hipMemcpyAsync(a->d_raw.p + row0 * a->A, rows, (size_t)n_rows * a->A * sizeof(float), hipMemcpyHostToDevice, g_rt.stream));
    return PRALINE_OK;
}

// (destroys the arena when it fails)
static int arena_finish(praline_arena *a, const float *S)
{
    if (!a || !a->building) return fail(PRALINE_ERR_ARG, "the arena is not being built");
    if (!S) { (void)hipStreamSynchronize(g_rt.stream); delete a; return fail(PRALINE_ERR_ARG, "NULL score matrix"); }
    PhaseTimer pt("arena_create");
    const int64_t n_seqs = a->n_seqs, rr = a->rows_raw;
    const int A = a->A;
    const int32_t *lens = a->len.data();
    const unsigned inexact_bits = a->inexact_bits;
    // active symbols: i contributes to m = sum_i P1[y,i] * Q2[x,i] only if some profile has mass on
    // it and row i of S is not all zero; dropping the others is exact (their terms are +-0).
    // One pass over the raw profiles gathers everything the host needs from them: which symbols carry mass, and per
    // row whether it is one-hot and on which symbol (counted branch-free so that the loop vectorises).
    std::vector<char> has_mass(A, 0), has_score(A, 0);
    bool all_onehot_rows = true;
    hipStream_t st = g_rt.stream;
    // the raw profiles go up first; the scan of their rows (mass per symbol, one-hot rows) runs on the device - one
    // small read-back instead of 0.6 ms of host time for the 11 MB of C2.  Everything the host can prepare without the
    // scan's answer is done while the upload is in flight (a DMA when the caller's buffer is page-locked:
    // praline_host_alloc).
    int *const d_flags = g_rt.d_scan_flags;   // (A <= 254; owned by the runtime: nothing is freed when this call returns)
    int *flags = g_rt.h_flags;
    {
        hipError_t e0 = hipMemsetAsync(d_flags, 0, ((size_t)A + 1) * sizeof(int), st);
        if (e0 == hipSuccess) {
            hipLaunchKernelGGL(k_scan_profiles, dim3((unsigned)((rr + 255) / 256)), dim3(256), 0, st, a->d_raw.p, rr, A, a->d_sym_raw.p, d_flags);
            e0 = hipGetLastError();
        }
        if (e0 == hipSuccess) e0 = hipMemcpyAsync(flags, d_flags, ((size_t)A + 1) * sizeof(int), hipMemcpyDeviceToHost, st);
        if (e0 != hipSuccess) { (void)hipStreamSynchronize(st); delete a; return fail(PRALINE_ERR_DEVICE, "arena scan: %s", hipGetErrorString(e0)); }
    }
    pt.mark("upload + device scan enqueued");
    for (int i = 0; i < A; ++i)
        for (int j = 0; j < A; ++j)
            if (S[i * A + j] != 0.0f) has_score[i] = 1;
    for (int k = 0; k <= 8 && a->s_scale_bits < 0; ++k) {
        bool ok = true;
        for (int i = 0; i < A * A && ok; ++i) {
            const float v = S[i] * (float)(1 << k);
            ok = std::isfinite(v) && v == std::nearbyint(v);
        }
        if (ok) a->s_scale_bits = k;
    }
    for (int i = 0; i < A * A; ++i) a->s_absmax = std::max(a->s_absmax, std::fabs(S[i]));
    std::vector<int32_t> &seq_of_rowp = a->h_seq_of_rowp;
    seq_of_rowp.assign((size_t)a->rows_pad, -1);
    for (int64_t s = 0; s < n_seqs; ++s)
        std::fill(seq_of_rowp.begin() + a->row_off_pad[s], seq_of_rowp.begin() + a->row_off_pad[s] + (lens[s] + 31) / 32 * 32, (int32_t)s);
    a->h_S.assign(S, S + (size_t)A * A);
    {
        int rc0 = PRALINE_OK;
        if ((rc0 = a->d_S.alloc((size_t)A * A)) || (rc0 = a->d_S.upload(a->h_S.data(), (size_t)A * A, st)) ||
            (rc0 = a->d_len.upload(a->len, st)) || (rc0 = a->d_row_off_pad.upload(a->row_off_pad, st)) ||
            (rc0 = a->d_row_off_raw.upload(a->row_off_raw, st)) || (rc0 = a->d_seq_of_rowp.upload(seq_of_rowp, st)) ||
            (rc0 = a->d_flag16.alloc(1))) {
            (void)hipStreamSynchronize(st);
            delete a;
            return rc0;
        }
    }
    pt.mark("host tables");
    {
        const hipError_t e0 = hipStreamSynchronize(st);
        if (e0 != hipSuccess) { delete a; return fail(PRALINE_ERR_DEVICE, "arena scan: %s", hipGetErrorString(e0)); }
        for (int i = 0; i < A; ++i) has_mass[i] = (char)(flags[(size_t)i] != 0);
        all_onehot_rows = flags[(size_t)A] == 0;
    }
    pt.mark("wait for the scan");
    for (int i = 0; i < A; ++i)
        if (has_mass[i] && has_score[i]) a->active.push_back(i);
    a->n_active = (int)a->active.size();
    if (const char *env = getenv("PRALINE_NO_COMPACT")) {
        if (env[0] == '1') { a->active.resize(A); std::iota(a->active.begin(), a->active.end(), 0); a->n_active = A; }
    }
    const int need = std::max(1, (a->n_active + 1) / 2);
    a->nstep = 0;
    for (int c : kNstepChoices) if (c >= need) { a->nstep = c; break; }
    if (!a->nstep) {
        // more active symbols than the MFMA operand layouts hold (32): the arena keeps the raw profiles only and its
        // plans evaluate the match scores on the vector ALU in the reference's order (k_match_ref, any alphabet <= 254)
        a->wide = true;
        a->nstep = 2;
    }
    a->KS = (a->nstep + 3) / 4 * 4;
    a->KP = 2 * a->KS;
    a->nr16 = a->wide ? 0 : (a->n_active <= 16 ? 1 : 2);

    // one-hot arenas (every row: a single 1, zeros elsewhere): active-symbol bytes for the one-hot operand table (built
    // on the device below: k_build_sym8)
    std::vector<unsigned char> &slot_of = a->h_slot_of;
    {
        const bool want_table = a->nr16 > 0 && !(getenv("PRALINE_NO_ONEHOT") && getenv("PRALINE_NO_ONEHOT")[0] == '1');
        a->all_onehot = all_onehot_rows;
        a->onehot = all_onehot_rows && want_table;
        if (a->onehot) {
            const unsigned char none = (unsigned char)(16 * a->nr16);
            slot_of.assign(256, none);
            for (int k = 0; k < a->n_active; ++k) slot_of[a->active[k]] = (unsigned char)k;
        }
    }

    int rc = PRALINE_OK;
    a->h_active_up = a->active.empty() ? std::vector<int32_t>(1, 0) : a->active;
    if ((rc = a->d_active.upload(a->h_active_up, st)) ||
        (rc = a->d_P.alloc(a->wide ? 1 : (size_t)a->rows_pad * a->KP)) || (rc = a->d_Q.alloc(a->wide ? 1 : (size_t)a->rows_pad * a->KP)) ||
        (a->onehot && (rc = a->d_sym8.alloc((size_t)a->rows_pad + 64))) ||
        (a->nr16 > 0 && ((rc = a->d_P16.alloc((size_t)a->rows_pad * 4 * a->nr16 * 16)) ||
                         (rc = a->d_Q16.alloc((size_t)a->rows_pad * 4 * a->nr16 * 16))))) {
        (void)hipStreamSynchronize(st);
        delete a;
        return rc;
    }
    if (a->onehot) {
        // (the table is read by k_build_sym8 below from the runtime's 256-byte buffer; its host source lives in the arena)
        if (hipMemcpyAsync(g_rt.d_slot_of, slot_of.data(), 256, hipMemcpyHostToDevice, st) != hipSuccess) {
            (void)hipStreamSynchronize(st);
            delete a;
            return fail(PRALINE_ERR_DEVICE, "symbol table upload failed");
        }
        const int64_t rows_out = a->rows_pad + 64;
        hipLaunchKernelGGL(k_build_sym8, dim3((unsigned)((rows_out + 255) / 256)), dim3(256), 0, st, a->d_sym_raw.p, a->d_seq_of_rowp.p,
                           a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p, g_rt.d_slot_of, a->rows_pad, rows_out, a->d_sym8.p);
        if (hipGetLastError() != hipSuccess) { (void)hipStreamSynchronize(st); delete a; return fail(PRALINE_ERR_DEVICE, "k_build_sym8 launch failed"); }
    }
    pt.mark("allocations + uploads (async)");
    const bool host_knows_split = a->nr16 > 0 && inexact_bits != 0;
    if (host_knows_split) {
        // profiles that float16 cannot hold: three terms (K-packed into four MFMAs for at most 21 active symbols), one
        // fused pack / pre-multiply / split launch, no read-back
        const char *pk = getenv("PRALINE_PACKED3");
        a->nterm16 = (a->nr16 == 2 && 3 * a->n_active <= 63 && !(pk && pk[0] == '0')) ? 2 : 3;
        rc = arena_launch_premultiply(a);
    } else {
        rc = arena_launch_premultiply(a, true);
    }
    if (rc != PRALINE_OK) { (void)hipStreamSynchronize(st); delete a; return rc; }
    pt.mark("premultiply launches");
    // Float profiles: nothing to read back - the packing launch and the small uploads (their host sources live in the
    // arena) finish under whatever the caller does next on this stream (plan creation waits for its own uploads).
    hipError_t e = hipSuccess;
    if (!host_knows_split) {
        e = hipStreamSynchronize(st);
        pt.mark("stream sync");
    }
    if (e != hipSuccess) { delete a; return fail(PRALINE_ERR_DEVICE, "arena upload: %s", hipGetErrorString(e)); }
    if (a->nr16 > 0 && !host_knows_split) {
        int flag = 1;
        e = hipMemcpy(&flag, a->d_flag16.p, sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete a; return fail(PRALINE_ERR_DEVICE, "arena flag: %s", hipGetErrorString(e)); }
        a->nterm16 = flag ? 3 : 1;
        // at most 21 active symbols: the three terms fit the 64 k slots of four MFMAs (PRALINE_PACKED3=0: keep six)
        const char *pk = getenv("PRALINE_PACKED3");
        if (a->nterm16 == 3 && a->nr16 == 2 && 3 * a->n_active <= 63 && !(pk && pk[0] == '0')) {
            a->nterm16 = 2;
            rc = arena_launch_premultiply(a);   // re-split in the packed layout
            if (rc == PRALINE_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(PRALINE_ERR_DEVICE, "arena re-split failed");
            if (rc != PRALINE_OK) { delete a; return rc; }
        }
    }
    a->building = false;
    return PRALINE_OK;
}

extern "C" int praline_arena_create(int64_t n_seqs, const int32_t *lens, int32_t A, const float *profiles,
                                    const float *S, praline_arena **out)
{
    if (!out) return fail(PRALINE_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_seqs <= 0 || !lens || !profiles || !S) return fail(PRALINE_ERR_ARG, "NULL or empty arena input");
    praline_arena *a = nullptr;
    RC(arena_begin(n_seqs, lens, A, &a));
    int rc = arena_put(a, 0, a->rows_raw, profiles);
    if (rc != PRALINE_OK) { (void)hipStreamSynchronize(g_rt.stream); delete a; return rc; }
    RC(arena_finish(a, S));
    *out = a;
    return PRALINE_OK;
}

extern "C" int praline_arena_begin(int64_t n_seqs, const int32_t *lens, int32_t A, praline_arena **out)
{
    return arena_begin(n_seqs, lens, A, out);
}

extern "C" int praline_arena_put_rows(praline_arena *arena, int64_t row0, int64_t n_rows, const float *rows)
{
    return arena_put(arena, row0, n_rows, rows);
}

extern "C" int praline_arena_finish(praline_arena *arena, const float *S) { return arena_finish(arena, S); }

extern "C" int praline_arena_destroy(praline_arena *arena)
{
    if (!arena) return PRALINE_OK;
    if (g_rt.ready) (void)hipStreamSynchronize(g_rt.stream);
    delete arena;
    return PRALINE_OK;
}

extern "C" int praline_arena_set_track_sets(praline_arena *arena, int32_t n_sets, const int32_t *sizes)
{
    RC(arena_ready(arena));
    if (!arena || n_sets <= 0 || !sizes) return fail(PRALINE_ERR_ARG, "bad track-set arguments");
    std::vector<int32_t> lo(1, 0);
    for (int n = 0; n < n_sets; ++n) {
        if (sizes[n] <= 0) return fail(PRALINE_ERR_ARG, "track set %d has size %d", n, sizes[n]);
        lo.push_back(lo.back() + sizes[n]);
    }
    if (lo.back() != arena->A) return fail(PRALINE_ERR_ARG, "track-set sizes sum to %d, the arena alphabet is %d", lo.back(), arena->A);
    arena->set_lo.swap(lo);
    arena->ref_ready = false;
    arena->reft2_state = 0;
    return PRALINE_OK;
}

// Per-position gap scores (GapScoreModel, praline/container/score.py:45-68): g = float32 [sum of the lengths][2] =
// (open, extend) of every position of every sequence, in arena order; NULL: back to constant gap scores.
extern "C" int praline_arena_set_gap_scores(praline_arena *arena, const float *g)
{
    RC(arena_ready(arena));
    if (!arena) return fail(PRALINE_ERR_ARG, "arena is NULL");
    RC(ensure_runtime(-1));
    if (!g) { arena->has_gaps = false; arena->d_gaps.release(); return PRALINE_OK; }
    // gap rows exist for the sequences the arena holds NOW: an arena that grows (praline_arena_set_counts /
    // praline_arena_append_merged) would leave its appended sequences without any - the two are mutually exclusive
    if (arena->have_cnt || arena->cap_seqs != 0)
        return fail(PRALINE_ERR_UNSUPPORTED, "gap scores on a growing arena (praline_arena_set_counts) are not supported");
    const size_t rows = (size_t)arena->rows_pad + 64;
    std::vector<float> pad(rows * 2, 0.0f);
    for (int64_t q = 0; q < arena->n_seqs; ++q) {
        const float *src = g + (size_t)arena->row_off_raw[(size_t)q] * 2;
        const int L = arena->len[(size_t)q];
        for (int k = 0; k < 2 * L; ++k) {
            if (!(src[k] <= 0.0f)) return fail(PRALINE_ERR_UNSUPPORTED, "batched kernels need gap scores <= 0 (sequence %lld, position %d: %g)", (long long)q, k / 2, src[k]);
        }
        std::copy(src, src + 2 * (size_t)L, pad.begin() + (size_t)arena->row_off_pad[(size_t)q] * 2);
    }
    hipStream_t st = g_rt.stream;
    RC(arena->d_gaps.upload(pad, st));
    HIPCHK(hipStreamSynchronize(st));
    arena->has_gaps = true;
    return PRALINE_OK;
}

static int arena_ensure_ref(praline_arena *a);
static int arena_ensure_reft_table(praline_arena *a);

// reference-order match scores of the pairs chunk_pairs[0 .. n_chunk) into mref + m_off[pair]
static int launch_match_ref(praline_arena *a, const int32_t *d_pairs, const int32_t *d_chunk_pairs, size_t n_chunk, int max_l1,
                            const int64_t *d_m_off, float *d_mref, const TileOut &to = TileOut())
{
    RC(arena_ensure_ref(a));
    RC(arena_ensure_reft_table(a));
    hipStream_t st = g_rt.stream;
    const dim3 grid((unsigned)n_chunk, (unsigned)((max_l1 + PRALINE_REF_ROWS - 1) / PRALINE_REF_ROWS)), block(256);
    const int n_sets = (int)a->set_lo.size() - 1;
#define PRALINE_REFT(TB)                                                                                               \
    hipLaunchKernelGGL((k_match_reft<TB>), grid, block, 0, st, a->d_raw.p, a->A, a->d_reft.p, a->rows_raw, a->d_row_off_raw.p, a->d_len.p,  \
                       a->d_nzidx.p, a->d_nzcnt.p, a->d_set_lo.p, n_sets, d_pairs, d_chunk_pairs, d_m_off, d_mref, to)
    switch (a->reft_state == 1 ? a->ref_tb : 0) {
        case 4: PRALINE_REFT(4); break;
        case 8: PRALINE_REFT(8); break;
        case 16: PRALINE_REFT(16); break;
        case 32: PRALINE_REFT(32); break;
        default:
            hipLaunchKernelGGL(k_match_ref, grid, block, 0, st, a->d_raw.p, a->d_S.p, a->A, a->d_row_off_raw.p, a->d_len.p, a->d_nzidx.p,
                               a->d_nzcnt.p, a->d_set_lo.p, n_sets, d_pairs, d_chunk_pairs, d_m_off, d_mref, to);
    }
#undef PRALINE_REFT
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

// nonzero lists + set boundaries for k_match_ref, built on first use
static int arena_ensure_ref(praline_arena *a)
{
    if (a->ref_ready) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    RC(a->d_set_lo.upload(a->set_lo, st));
    RC(a->d_nzidx.alloc((size_t)a->rows_raw * a->A));
    RC(a->d_nzcnt.alloc((size_t)a->rows_raw));
    hipLaunchKernelGGL(k_build_nz, dim3((unsigned)((a->rows_raw + 255) / 256)), dim3(256), 0, st, a->d_raw.p, a->rows_raw, a->A,
                       a->d_nzidx.p, a->d_nzcnt.p);
    HIPCHK(hipGetLastError());
    // the per-row tables of k_match_reft (half of every term prepared once per arena row) when they fit
    std::vector<unsigned char> cnt((size_t)a->rows_raw);
    HIPCHK(hipMemcpyAsync(cnt.data(), a->d_nzcnt.p, cnt.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int max_nz = 1;
    for (unsigned char c : cnt) max_nz = std::max(max_nz, (int)c);
    a->ref_tb = max_nz <= 4 ? 4 : (max_nz <= 8 ? 8 : (max_nz <= 16 ? 16 : (max_nz <= 32 ? 32 : 0)));
    a->reft_state = 0;
    a->d_reft.release();
    a->ref_ready = true;
    return PRALINE_OK;
}

// the per-row tables of k_match_reft, built on the first launch that needs them (plans on the tile kernels never do)
static int arena_ensure_reft_table(praline_arena *a)
{
    if (a->reft_state != 0) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    a->reft_state = -1;
    size_t table_limit = (size_t)16 << 30;
    if (const char *env = getenv("PRALINE_REF_TABLE_MB")) table_limit = (size_t)atoll(env) << 20;
    const size_t t_elems = (size_t)a->rows_raw * a->A * (size_t)a->ref_tb;
    if (a->ref_tb > 0 && t_elems * sizeof(float) <= table_limit) {
        RC(a->d_reft.alloc(t_elems));
        const int64_t n = a->rows_raw * a->A;
        hipLaunchKernelGGL(k_build_reft, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a->d_raw.p, a->d_S.p, a->A, a->rows_raw,
                           a->d_nzidx.p, a->d_nzcnt.p, a->d_set_lo.p, (int)a->set_lo.size() - 1, a->ref_tb, a->d_reft.p);
        HIPCHK(hipGetLastError());
        a->reft_state = 1;
    }
    return PRALINE_OK;
}

// the interleaved table of k_match_tile; state -1 when the arena does not qualify (more than 32 symbols, rows with more
// than 8 nonzeros, table over the limit): such plans take their tiles from k_match_reft / k_match_ref
static int arena_ensure_reft2(praline_arena *a)
{
    if (a->reft2_state != 0) return PRALINE_OK;
    RC(arena_ensure_ref(a));
    a->reft2_state = -1;
    if (a->wide || a->nr16 <= 0 || a->ref_tb <= 0 || !praline_match_tile_supported(a->A, a->ref_tb)) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    std::vector<int64_t> pr_off((size_t)a->n_seqs);
    int64_t pr = 0;
    for (int64_t q = 0; q < a->n_seqs; ++q) { pr_off[(size_t)q] = pr; pr += (a->len[(size_t)q] + 1) / 2; }
    a->pair_rows = std::max<int64_t>(pr, 1);
    RC(a->d_pr_off.upload(pr_off, st));
    RC(a->d_reft2.alloc((size_t)a->A * (size_t)a->pair_rows * (size_t)a->ref_tb * 2));
    RC(praline_launch_build_reft2(a->d_raw.p, a->d_S.p, a->A, a->d_row_off_raw.p, a->d_len.p, a->d_pr_off.p, a->pair_rows, a->d_nzidx.p,
                                  a->d_nzcnt.p, a->d_set_lo.p, (int)a->set_lo.size() - 1, a->ref_tb, a->d_reft2.p, (int)a->n_seqs, st));
    HIPCHK(hipStreamSynchronize(st));   // (pr_off goes out of scope)
    a->reft2_state = 1;
    return PRALINE_OK;
}

// ---- resident progressive alignment: clusters merged on the device, appended to the arena in place ----------
template <typename T> static int grow_buf(DevBuf<T> &b, size_t old_n, size_t new_n, int fill_byte, hipStream_t st)
{
    if (!b.p || new_n <= old_n) return PRALINE_OK;
    DevBuf<T> nb;
    RC(nb.alloc(new_n));
    HIPCHK(hipMemcpyAsync(nb.p, b.p, old_n * sizeof(T), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(nb.p + old_n, fill_byte, (new_n - old_n) * sizeof(T), st));
    HIPCHK(hipStreamSynchronize(st));   // the old block goes back to the pool
    std::swap(b.p, nb.p);
    std::swap(b.n, nb.n);
    std::swap(b.cap_bytes, nb.cap_bytes);
    return PRALINE_OK;
}

static int arena_reserve(praline_arena *a, int64_t need_seqs, int64_t need_rows_raw, int64_t need_rows_pad)
{
    hipStream_t st = g_rt.stream;
    const int64_t cur_seqs = a->cap_seqs ? a->cap_seqs : a->n_seqs, cur_raw = a->cap_rows_raw ? a->cap_rows_raw : a->rows_raw,
                  cur_pad = a->cap_rows_pad ? a->cap_rows_pad : a->rows_pad;
    if (need_seqs > cur_seqs) {
        const int64_t n = std::max(need_seqs, 2 * cur_seqs);
        RC(grow_buf(a->d_len, (size_t)cur_seqs, (size_t)n, 0, st));
        RC(grow_buf(a->d_row_off_pad, (size_t)cur_seqs, (size_t)n, 0, st));
        RC(grow_buf(a->d_row_off_raw, (size_t)cur_seqs, (size_t)n, 0, st));
        a->cap_seqs = n;
    }
    if (need_rows_raw > cur_raw) {
        const int64_t n = std::max(need_rows_raw, 2 * cur_raw);
        RC(grow_buf(a->d_raw, (size_t)cur_raw * a->A, (size_t)n * a->A, 0, st));
        RC(grow_buf(a->d_cnt, (size_t)cur_raw * a->A, (size_t)n * a->A, 0, st));
        a->cap_rows_raw = n;
    }
    if (need_rows_pad > cur_pad) {
        const int64_t n = std::max(need_rows_pad, 2 * cur_pad);
        RC(grow_buf(a->d_seq_of_rowp, (size_t)cur_pad, (size_t)n, 0xff, st));   // -1: no sequence
        if (!a->wide) {
            RC(grow_buf(a->d_P, (size_t)cur_pad * a->KP, (size_t)n * a->KP, 0, st));
            RC(grow_buf(a->d_Q, (size_t)cur_pad * a->KP, (size_t)n * a->KP, 0, st));
            if (a->nr16 > 0) {
                RC(grow_buf(a->d_P16, (size_t)cur_pad * 4 * a->nr16 * 16, (size_t)n * 4 * a->nr16 * 16, 0, st));
                RC(grow_buf(a->d_Q16, (size_t)cur_pad * 4 * a->nr16 * 16, (size_t)n * 4 * a->nr16 * 16, 0, st));
            }
        }
        a->cap_rows_pad = n;
    }
    return PRALINE_OK;
}

extern "C" int praline_arena_set_counts(praline_arena *arena, const int32_t *counts, int64_t reserve_seqs, int64_t reserve_rows)
{
    RC(arena_ready(arena));
    if (!arena || !counts) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (arena->has_gaps)
        return fail(PRALINE_ERR_UNSUPPORTED, "the arena holds per-position gap scores (praline_arena_set_gap_scores): it cannot grow");
    RC(ensure_runtime(-1));
    praline_arena *a = arena;
    RC(a->d_cnt.alloc((size_t)a->rows_raw * a->A));
    HIPCHK(hipMemcpyAsync(a->d_cnt.p, counts, (size_t)a->rows_raw * a->A * sizeof(int32_t), hipMemcpyHostToDevice, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    a->have_cnt = true;
    if (reserve_seqs > 0 || reserve_rows > 0)
        RC(arena_reserve(a, a->n_seqs + std::max<int64_t>(reserve_seqs, 0), a->rows_raw + std::max<int64_t>(reserve_rows, 0),
                         a->rows_pad + std::max<int64_t>(reserve_rows, 0) + 32 * std::max<int64_t>(reserve_seqs, 0)));
    return PRALINE_OK;
}

extern "C" int praline_arena_premultiply(praline_arena *arena)
{
    RC(arena_ready(arena));
    if (!arena) return fail(PRALINE_ERR_ARG, "arena is NULL");
    return arena_launch_premultiply(arena);
}
