// praline_stage.hip.h -- part of praline_dp.hip (one translation unit; included there, in this order): stage helpers on device paths: preprofile counts and path bounding boxes, Waterman-Eggert masks, score and path read-back,
// timing / kernel-name queries.
// ---- preprofile stage on the device: counts and path bounding boxes (k_path_counts / k_path_bounds) ----------
extern "C" int praline_arena_counts_reset(praline_arena *arena)
{
    RC(arena_ready(arena));
    if (!arena->counts_ext && !arena->d_counts.p) RC(arena->d_counts.alloc((size_t)arena->rows_raw * arena->A));
    HIPCHK(hipMemsetAsync(arena->counts_ptr(), 0, (size_t)arena->rows_raw * arena->A * sizeof(int32_t), g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_arena_counts_bind(praline_arena *arena, void *d_counts)
{
    RC(arena_ready(arena));
    arena->counts_ext = (int32_t *)d_counts;
    return PRALINE_OK;
}

extern "C" int praline_plan_add_counts(praline_plan *plan, int use_threshold, float threshold, int local)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (!plan->want_paths) return fail(PRALINE_ERR_ARG, "plan was created without want_paths");
    if (plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "praline_plan_run has not been called");
    praline_arena &a = *plan->arena;
    if (!a.counts_ptr()) return fail(PRALINE_ERR_ARG, "praline_arena_counts_reset has not been called");
    if (!a.all_onehot)
        return fail(PRALINE_ERR_UNSUPPORTED, "preprofile counting needs one-hot profiles (plain sequences), as "
                    "ProfileBuilder needs plain tracks (praline/util/align.py:187-213)");
    if (plan->n_pairs == 0) return PRALINE_OK;
    // pair lists whose masters come in runs (the preprofile stage's order): a workgroup per run with the master's count
    // block in LDS (k_path_counts_runs).  The runs are found once per plan, from the device copy of the pair list.
    if (plan->count_runs < 0) {
        plan->count_runs = 0;
        const size_t lds_need = (size_t)a.max_len * a.A * sizeof(int32_t);
        const char *cr = getenv("PRALINE_COUNT_RUNS");   // 0: never; 1: whenever the count block fits LDS (tests); default: runs of 64 pairs and more on average
        const bool forced = cr && cr[0] == '1';
        const int64_t max_runs = forced ? plan->n_pairs : plan->n_pairs / 64;
        if (lds_need <= (size_t)64 << 10 && (plan->n_pairs >= 4096 || forced) && !(cr && cr[0] == '0')) {
            std::vector<int32_t> hp((size_t)plan->n_pairs * 2);
            HIPCHK(hipMemcpyAsync(hp.data(), plan->d_pairs.p, hp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, g_rt.stream));
            HIPCHK(hipStreamSynchronize(g_rt.stream));
            std::vector<int64_t> runs;
            for (int64_t p = 0; p < plan->n_pairs;) {
                int64_t q = p + 1;
                while (q < plan->n_pairs && hp[(size_t)(2 * q)] == hp[(size_t)(2 * p)]) ++q;
                runs.push_back(p); runs.push_back(q);
                p = q;
                if ((int64_t)runs.size() / 2 > max_runs) break;   // (short runs: one lane per pair and global atomics)
            }
            if ((int64_t)runs.size() / 2 <= max_runs) {
                RC(plan->d_count_runs.upload(runs, g_rt.stream));
                HIPCHK(hipStreamSynchronize(g_rt.stream));   // (runs goes out of scope)
                plan->count_runs = (int64_t)runs.size() / 2;
            }
        }
    }
    if (plan->count_runs > 0) {
        hipLaunchKernelGGL(k_path_counts_runs, dim3((unsigned)plan->count_runs), dim3(256), (size_t)a.max_len * a.A * sizeof(int32_t),
                           g_rt.stream, plan->d_pairs.p, plan->last_scores, plan->d_paths.p, plan->d_path_start.p, plan->d_path_rows.p,
                           plan->d_count_runs.p, use_threshold, threshold, local, a.d_row_off_raw.p, a.d_len.p, a.d_sym_raw.p, a.A,
                           a.counts_ptr());
        HIPCHK(hipGetLastError());
        return PRALINE_OK;
    }
    const int threads = 64;
    const int64_t blocks = (plan->n_pairs + threads - 1) / threads;
    hipLaunchKernelGGL(k_path_counts, dim3((unsigned)blocks), dim3(threads), 0, g_rt.stream, plan->d_pairs.p,
                       plan->last_scores, plan->d_paths.p, plan->d_path_start.p, plan->d_path_rows.p, plan->n_pairs,
                       use_threshold, threshold, local, a.d_row_off_raw.p, a.d_len.p, a.d_sym_raw.p, a.A, a.counts_ptr());
    HIPCHK(hipGetLastError());
    return PRALINE_OK;
}

extern "C" int praline_arena_counts_read(praline_arena *arena, int32_t *counts)
{
    if (!arena || !counts) return fail(PRALINE_ERR_ARG, "NULL argument");
    RC(arena_ready(arena));
    if (!arena->counts_ptr()) return fail(PRALINE_ERR_ARG, "praline_arena_counts_reset has not been called");
    HIPCHK(hipMemcpyAsync(counts, arena->counts_ptr(), (size_t)arena->rows_raw * arena->A * sizeof(int32_t),
                          hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_plan_path_bounds(praline_plan *plan, int32_t *bounds)
{
    if (!plan || !bounds) return fail(PRALINE_ERR_ARG, "NULL argument");
    if (!plan->want_paths) return fail(PRALINE_ERR_ARG, "plan was created without want_paths");
    if (plan->n_pairs == 0) return PRALINE_OK;
    DevBuf<int32_t> d_bounds;
    RC(d_bounds.alloc((size_t)plan->n_pairs * 4));
    const int threads = 256;
    const int64_t blocks = (plan->n_pairs + threads - 1) / threads;
    hipLaunchKernelGGL(k_path_bounds, dim3((unsigned)blocks), dim3(threads), 0, g_rt.stream, plan->d_paths.p,
                       plan->d_path_start.p, plan->d_path_rows.p, plan->n_pairs, d_bounds.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(bounds, d_bounds.p, (size_t)plan->n_pairs * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, g_rt.stream));
    HIPCHK(hipStreamSynchronize(g_rt.stream));
    return PRALINE_OK;
}

extern "C" int praline_arena_append_merged_many(praline_arena *arena, praline_plan *plan, int64_t n, const int64_t *pair_index,
                                                int32_t *new_index, int32_t *new_len)
{
    if (!arena || !plan || !new_index || !new_len || (n > 0 && !pair_index)) return fail(PRALINE_ERR_ARG, "NULL argument");
    RC(arena_ready(arena));
    if (plan->arena != arena) return fail(PRALINE_ERR_ARG, "the plan belongs to another arena");
    if (!plan->want_paths || plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "the plan has no paths (want_paths + praline_plan_run first)");
    if (plan->last_mode == PRALINE_MODE_LOCAL) return fail(PRALINE_ERR_UNSUPPORTED, "clusters are merged along global / semiglobal paths");
    if (n <= 0) return PRALINE_OK;
    for (int64_t q = 0; q < n; ++q)
        if (pair_index[q] < 0 || pair_index[q] >= plan->n_pairs) return fail(PRALINE_ERR_ARG, "pair index out of range");
    praline_arena *a = arena;
    if (!a->have_cnt) return fail(PRALINE_ERR_ARG, "praline_arena_set_counts has not been called");
    if (a->has_gaps) return fail(PRALINE_ERR_UNSUPPORTED, "the arena holds per-position gap scores: it cannot grow");
    hipStream_t st = g_rt.stream;
    // where the paths are: one round trip for the whole plan (a level of the guide tree is one plan)
    const int64_t np = plan->n_pairs;
    std::vector<int64_t> start((size_t)np);
    std::vector<int32_t> rows((size_t)np), pr((size_t)np * 2);
    HIPCHK(hipMemcpyAsync(start.data(), plan->d_path_start.p, (size_t)np * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(rows.data(), plan->d_path_rows.p, (size_t)np * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(pr.data(), plan->d_pairs.p, (size_t)np * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int64_t idx0 = a->n_seqs;
    int64_t rows_raw = a->rows_raw, rp = a->rp_end;
    int new_max = a->max_len;
    for (int64_t q = 0; q < n; ++q) {
        const int cols = rows[(size_t)pair_index[q]] - 1;
        if (cols <= 0) return fail(PRALINE_ERR_DEVICE, "empty alignment path");
        rows_raw += cols;
        rp += (cols + 31) / 32 * 32;
        new_max = std::max(new_max, cols);
    }
    const int64_t new_rows_pad = rp + (new_max + 31) / 32 * 32 + 64;
    RC(arena_reserve(a, idx0 + n, rows_raw, new_rows_pad));
    if (!a->d_set_lo.p) RC(a->d_set_lo.upload(a->set_lo, st));
    const int64_t rp0 = a->rp_end;
    for (int64_t q = 0; q < n; ++q) {
        const int64_t p = pair_index[q];
        const int cols = rows[(size_t)p] - 1;
        const int64_t pad = (cols + 31) / 32 * 32;
        const int32_t off_raw = (int32_t)a->rows_raw, off_pad = (int32_t)a->rp_end;
        const int64_t idx = a->n_seqs;
        hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((pad + 255) / 256)), dim3(256), 0, st, a->d_seq_of_rowp.p + a->rp_end, pad, (int32_t)idx);
        hipLaunchKernelGGL(k_merge_clusters, dim3((unsigned)cols), dim3(64), 0, st, plan->d_paths.p + 2 * start[(size_t)p], cols, a->d_cnt.p,
                           a->d_raw.p, a->A, (int64_t)a->row_off_raw[pr[(size_t)(2 * p)]], (int64_t)a->row_off_raw[pr[(size_t)(2 * p + 1)]],
                           (int64_t)off_raw, a->d_set_lo.p, (int)a->set_lo.size() - 1);
        a->len.push_back(cols);
        a->row_off_raw.push_back(off_raw);
        a->row_off_pad.push_back(off_pad);
        a->n_seqs = idx + 1;
        a->rows_raw += cols;
        a->rp_end += pad;
        new_index[q] = (int32_t)idx;
        new_len[q] = cols;
    }
    HIPCHK(hipGetLastError());
    // the descriptors of the new sequences (the host vectors are final now)
    HIPCHK(hipMemcpyAsync(a->d_len.p + idx0, a->len.data() + idx0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(a->d_row_off_raw.p + idx0, a->row_off_raw.data() + idx0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(a->d_row_off_pad.p + idx0, a->row_off_pad.data() + idx0, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    a->max_len = new_max;
    a->rows_pad = new_rows_pad;
    // a merged cluster is no plain sequence: the one-hot shortcuts of this arena end here
    a->onehot = false;
    a->all_onehot = false;
    if (a->nr16 > 0 && a->nterm16 == 1) a->nterm16 = 3;   // (an exact arena keeps the standard layout; its new rows need the lo pieces)
    a->ref_ready = false;
    a->reft2_state = 0;
    a->d_counts.release();
    a->counts_ext = nullptr;
    if (!a->wide) {   // packed operands of the new rows only (they are contiguous in the padded row space)
        hipLaunchKernelGGL(k_prepare_rows, dim3((unsigned)((a->rp_end - rp0) / 32)), dim3(256), 0, st, a->d_raw.p, a->d_S.p, a->d_seq_of_rowp.p,
                           a->d_row_off_pad.p, a->d_row_off_raw.p, a->d_len.p, a->d_active.p, a->n_active, a->A, a->KP, a->KS,
                           a->rows_pad, a->d_P.p, a->d_Q.p, a->nr16, (_Float16 *)a->d_P16.p, (_Float16 *)a->d_Q16.p,
                           (int64_t)(rp0 / 32), a->nterm16 == 2 ? 1 : 0);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(st));   // the descriptor uploads read the host vectors
    return PRALINE_OK;
}

extern "C" int praline_arena_append_merged(praline_arena *arena, praline_plan *plan, int64_t pair_index, int32_t *new_index,
                                           int32_t *new_len)
{
    return praline_arena_append_merged_many(arena, plan, 1, &pair_index, new_index, new_len);
}

extern "C" int praline_plan_mask_path_bounds(praline_plan *plan)
{
    if (!plan) return fail(PRALINE_ERR_ARG, "plan is NULL");
    if (!plan->want_paths || plan->last_mode < 0) return fail(PRALINE_ERR_ARG, "the plan has no paths (want_paths + praline_plan_run first)");
    if (plan->has_rects && plan->slot_rects < 0)
        return fail(PRALINE_ERR_UNSUPPORTED, "the plan was created with its own rectangle lists");
    if (plan->n_pairs == 0) return PRALINE_OK;
    hipStream_t st = g_rt.stream;
    if (plan->slot_rects < 0) {
        // fixed slots: PRALINE_MAX_RECTS per pair, all empty to start with
        std::vector<int32_t> ro((size_t)plan->n_pairs + 1);
        for (int64_t p = 0; p <= plan->n_pairs; ++p) ro[(size_t)p] = (int32_t)(p * PRALINE_MAX_RECTS);
        RC(plan->d_rect_off.upload(ro, st));
        RC(plan->d_rects.alloc((size_t)plan->n_pairs * PRALINE_MAX_RECTS * 4));
        const int32_t empty[4] = {1 << 30, -1, 1 << 30, -1};
        std::vector<int32_t> rv((size_t)plan->n_pairs * PRALINE_MAX_RECTS * 4);
        for (size_t i = 0; i < rv.size(); ++i) rv[i] = empty[i & 3];
        RC(plan->d_rects.upload(rv.data(), rv.size(), st));
        HIPCHK(hipStreamSynchronize(st));
        plan->slot_rects = 0;
    }
    if (plan->slot_rects >= PRALINE_MAX_RECTS)
        return fail(PRALINE_ERR_UNSUPPORTED, "more than %d rectangles per pair: create a plan with explicit rectangle lists", PRALINE_MAX_RECTS);
    const int64_t blocks = (plan->n_pairs + 255) / 256;
    hipLaunchKernelGGL(k_path_bounds_to_rects, dim3((unsigned)blocks), dim3(256), 0, st, plan->d_paths.p, plan->d_path_start.p,
                       plan->d_path_rows.p, plan->n_pairs, plan->slot_rects, plan->d_rects.p);
    HIPCHK(hipGetLastError());
    plan->slot_rects += 1;
    plan->has_rects = true;
    plan->mask_kind = 1;
    return PRALINE_OK;
}

extern "C" int praline_batch_scores(praline_arena *arena, int mode, float gap_open, float gap_extend, int64_t n_pairs,
                                    const int32_t *pairs, float *scores)
{
    praline_plan *pl = nullptr;
    RC(praline_plan_create(arena, n_pairs, pairs, 0, nullptr, nullptr, &pl));
    int rc = praline_plan_run(pl, mode, gap_open, gap_extend, nullptr);
    if (rc == PRALINE_OK) rc = praline_plan_scores(pl, scores);
    praline_plan_destroy(pl);
    return rc;
}
